"""Per-basic-block instruction mix of one kernel in a hipcc -S listing (tools: where do a kernel's wave-instructions sit?).
usage: isa_blocks.py listing.s mangled_kernel_name"""
import re, sys
lines = open(sys.argv[1]).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(sys.argv[2] + ":"))
blocks, cur = [], {"lbl": "entry", "n": 0, "valu": 0, "salu": 0, "lds": 0, "vmem": 0, "smem": 0, "br": [], "mul": 0}
for l in lines[start + 1:]:
    if l.startswith(".Lfunc_end"): break
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        blocks.append(cur); cur = {"lbl": m.group(1), "n": 0, "valu": 0, "salu": 0, "lds": 0, "vmem": 0, "smem": 0, "br": [], "mul": 0}; continue
    t = l.strip().split()
    if not t or t[0].startswith(";") or t[0].startswith("."): continue
    op = t[0]; cur["n"] += 1
    if op.startswith("v_"):
        cur["valu"] += 1
        if "mul" in op or "mad_u64" in op: cur["mul"] += 1
    elif op.startswith("s_cbranch") or op.startswith("s_branch"): cur["br"].append(op[2:] + ">" + t[1]); cur["salu"] += 1
    elif op.startswith("s_load") or op.startswith("s_buffer"): cur["smem"] += 1
    elif op.startswith("s_"): cur["salu"] += 1
    elif op.startswith("ds_"): cur["lds"] += 1
    else: cur["vmem"] += 1
blocks.append(cur)
for i, b in enumerate(blocks):
    print(f"{i:3d} {b['lbl']:12s} n={b['n']:4d} valu={b['valu']:4d} (mul {b['mul']:3d}) salu={b['salu']:3d} lds={b['lds']:3d} vmem={b['vmem']:3d} smem={b['smem']:2d} {' '.join(b['br'])}")
print("total", sum(b["n"] for b in blocks), "valu", sum(b["valu"] for b in blocks))
