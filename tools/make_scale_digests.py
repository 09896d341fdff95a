#!/usr/bin/env python3
"""tests/golden/scale_digests.json: the sha256 of the reduced per-accession counters a run of N ranks must reproduce, for N = 2, 4, 8, made
on ONE GPU by `bench.py --emulate-world N` (one rank searching the N shards / holding the N stripes one after the other — the counters a
real N-rank run all-reduces to).  `bench.py --gpus N --scale-check` compares; tests/test_gpu_bench_launch.py reproduces them on one GPU.
Workloads: the default (configs[1] at 256 colours), the striped placement at 2^27 rows per stripe (eight 64-GiB stripes do not fit one
GPU; the 8-GPU run checks itself with the same --stripe-log2-bloom 27 before it is timed at 2^30), and the launch test's toys.
Run on the GPU box: python3 tools/make_scale_digests.py > gpurun_out/scale_digests.json"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOY = ["--reads", "20000", "--bloom", "1000003", "--genome-len", "30000"]
TOY_STRIPED = ["--reads", "20000", "--placement", "striped", "--stripe-log2-bloom", "20", "--stripe-colours", "128", "--density", "0.1"]
WORKLOADS = [([], (2, 4, 8)), (["--placement", "striped", "--stripe-log2-bloom", "27"], (2, 4, 8)), (TOY, (2, 3, 4, 8)), (TOY_STRIPED, (2, 3, 4, 8))]
out = {}
for extra, ns in WORKLOADS:
    for n in ns:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--emulate-world", str(n), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-variants",
               "--scale-check"] + extra
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT)
        if p.returncode not in (0, 1):
            print(p.stderr[-2000:], file=sys.stderr)
            sys.exit(1)
        j = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
        sc = j["scale_check"]
        out.setdefault(sc["key"], {})[str(n)] = sc["got"]
        print(f"{sc['key']} N={n} {sc['got'][:16]} setup {j['config']['setup_s']} s {j['config'].get('setup_phases')}", file=sys.stderr, flush=True)
print(json.dumps(out, indent=1, sort_keys=True))
