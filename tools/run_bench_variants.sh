mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -3) > gpurun_out/pytest_gpu.log
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline"
( $B 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default', d['value']/1e9, d['roofline']['kernel_ms'])"
  $B --bloom 65536 --density 0.2134 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('L2-resident', d['value']/1e9, d['roofline']['kernel_ms'])"
  $B --density 0.9 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dense0.9', d['value']/1e9, d['roofline']['kernel_ms'])"
  $B --density 1.0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dense1.0', d['value']/1e9, d['roofline']['kernel_ms'])"
  $B --colours 1024 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C1024', d['value']/1e9, d['roofline']['kernel_ms'], d['roofline']['frac'])"
  $B --colours 64 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C64', d['value']/1e9, d['roofline']['kernel_ms'], d['roofline']['frac'])"
) > gpurun_out/variants.log 2>&1
cat gpurun_out/pytest_gpu.log gpurun_out/variants.log
