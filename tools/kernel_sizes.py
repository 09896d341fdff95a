#!/usr/bin/env python3
"""Code size, registers, LDS and occupancy of every kernel of a translation unit, from the compiler's own remarks in its assembly:
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o /tmp/x.s colorid_amd/csrc/cid_kmerset.hip && python3 tools/kernel_sizes.py /tmp/x.s
(round 6: a kernel unrolled into 60 KB of code ran 15 % slower than its rolled form of 12 KB — the instruction cache is 64 KB for two CUs)"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
rows = []
for m in re.finditer(r"^(_Z[\w]+):.*?; codeLenInByte = (\d+).*?; NumVgprs: (\d+).*?; ScratchSize: (\d+).*?; Occupancy: (\d+).*?; LDSByteSize: (\d+)", txt, re.S | re.M):
    name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
    rows.append((int(m.group(2)), name[:90], m.group(3), m.group(4), m.group(5), m.group(6)))
for r in sorted(rows, reverse=True):
    print(f"{r[0]:7d} B  vgpr {r[2]:>3s} scratch {r[3]:>4s} occ {r[4]} lds {r[5]:>6s}  {r[1]}")
