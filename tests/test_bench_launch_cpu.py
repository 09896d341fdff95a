"""bench.py --gpus N without a launcher decides BEFORE touching a GPU: with fewer devices than ranks it refuses in one line
(exit 2, no traceback, nothing on stdout).  The two-rank runs themselves are tests/test_gpu_bench_launch.py."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_n_without_devices_is_refused_in_one_line():
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("devices present")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, env=env, timeout=300)
    assert p.returncode == 2 and p.stdout == ""
    err = [ln for ln in p.stderr.splitlines() if ln.strip()]
    assert len(err) == 1 and "needs 2 visible GPUs" in err[0]
