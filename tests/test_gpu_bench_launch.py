"""bench.py's N > 1 branches, executed: `python bench.py --gpus 2` starts its own two ranks (fresh child processes through
torch.distributed.run), here with BENCH_BACKEND=gloo so that both ranks can share the one GPU of the test box (RCCL refuses two
ranks on one device; on an 8-GPU node the same code runs with the default backend nccl = RCCL over xGMI).  The real HIP path runs
in every rank; only the exchange step takes the host route.

  * read-sharded / index-replicated (the reference's only parallel boundary is the rayon map over reads,
    src/read_id_mt_pe.rs:300-302, src/main.rs:718-721): one JSON line, n_gpus 2, two per_rank records, total k-mers = the sum
    of the shards, and the all-reduced per-accession counters equal those of ONE rank searching both shards (--emulate-world 2);
  * colour-striped (§8e.2, BASELINE configs[4]): the same, the one rank then holding both stripes."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOY = ["--reads", "20000", "--bloom", "1000003", "--genome-len", "30000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-variants"]
TOY_STRIPED = TOY + ["--placement", "striped", "--stripe-log2-bloom", "20", "--stripe-colours", "128", "--density", "0.1"]


def run_bench(args, backend=None, timeout=600, extra_env=None):
    env = dict(os.environ, **(extra_env or {}))
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    if backend:
        env["BENCH_BACKEND"] = backend
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       env=env, timeout=timeout, cwd=ROOT)
    return p


def one_json_line(p):
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("extra,world", [(TOY, 2), (TOY_STRIPED, 2), (TOY, 4), (TOY_STRIPED, 3)], ids=["replicated-2", "striped-2", "replicated-4", "striped-3"])
def test_n_ranks_equal_one_rank_over_all_shards(extra, world):
    two = one_json_line(run_bench(["--gpus", str(world)] + extra, backend="gloo"))
    one = one_json_line(run_bench(["--emulate-world", str(world)] + extra))
    assert two["n_gpus"] == world and one["n_gpus"] == 1
    assert two["config"]["backend"] == "gloo"
    assert len(two["per_rank"]) == world and [r["rank"] for r in two["per_rank"]] == list(range(world))
    assert all(r["kmers"] > 0 and r["kernel_ms"] > 0 for r in two["per_rank"])
    striped = "--placement" in extra
    if striped:   # every rank sees every k-mer; the query does not grow with N
        assert {r["kmers"] for r in two["per_rank"]} == {two["config"]["total_kmers"]} == {one["config"]["total_kmers"]}
        assert two["config"]["n_colors_total"] == one["config"]["n_colors_total"] == 128 * world
        assert two["config"]["consistent"] and one["config"]["consistent"]
    else:         # the shards (reads seeded per rank) add up
        assert two["config"]["total_kmers"] == sum(r["kmers"] for r in two["per_rank"]) == one["per_rank"][0]["kmers"]
    # value = the units all ranks processed / the slowest rank's time
    assert two["value"] == pytest.approx(two["config"]["total_kmers"] * two["steps"] / (two["ms_per_step"] * 1e-3 * two["steps"]), rel=1e-9)
    assert all(v > 0 for v in two["counters"]["sums"])     # hits, unique k-mers and their multiplicities all counted
    assert two["counters"] == one["counters"]


@pytest.mark.parametrize("extra,n", [(TOY, 2), (TOY, 8), (TOY_STRIPED, 3), (TOY_STRIPED, 8), ([], 2), ([], 8), (["--placement", "striped", "--stripe-log2-bloom", "27"], 4)],
                         ids=["toy-2", "toy-8", "toy-striped-3", "toy-striped-8", "default-2", "default-8", "striped27-4"])
def test_scale_check_reproduces_the_committed_digests(extra, n):
    """tests/golden/scale_digests.json (tools/make_scale_digests.py): what `bench.py --gpus N --scale-check` must find on an N-GPU node,
    reproduced here by one rank standing in for N — at the toys' sizes, at the default workload and at the striped placement's check size"""
    fast = [] if "--no-variants" in extra else ["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-variants"]
    one = one_json_line(run_bench(["--emulate-world", str(n), "--scale-check"] + extra + fast))
    sc = one["scale_check"]
    assert sc["ok"] is True and sc["ranks"] == n and sc["expected"] == sc["got"] == one["counters"]["sha256"], sc
    assert one["config"]["setup_s"] < 60 and one["config"]["setup_phases"]


def test_scale_check_fails_loudly_on_other_counters(tmp_path):
    """two gloo ranks with --scale-check: ok; the same asked to reproduce a digest that is not theirs: exit code 1, ok false in the line"""
    good = one_json_line(run_bench(["--gpus", "2", "--scale-check"] + TOY, backend="gloo"))
    assert good["scale_check"]["ok"] is True and good["scale_check"]["ranks"] == 2
    none = one_json_line(run_bench(["--emulate-world", "2", "--scale-check", "--reads", "20001"] + TOY[2:]))
    assert none["scale_check"]["ok"] is None and none["scale_check"]["expected"] is None     # no digest committed for this workload: reported, not failed
    import json as _json
    g = _json.load(open(os.path.join(ROOT, "tests", "golden", "scale_digests.json")))
    g[good["scale_check"]["key"]]["2"] = "0" * 64
    other = tmp_path / "scale_digests.json"          # (a copy: the committed file is never written to)
    other.write_text(_json.dumps(g))
    p = run_bench(["--emulate-world", "2", "--scale-check"] + TOY, extra_env={"BENCH_SCALE_DIGESTS": str(other)})
    assert p.returncode == 1
    bad = _json.loads([ln for ln in p.stdout.splitlines() if ln.strip()][0])
    assert bad["scale_check"]["ok"] is False


def test_nccl_without_enough_devices_is_a_one_line_refusal():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two devices present: the RCCL launch itself is tests/test_gpu_multi.py's business")
    p = run_bench(["--gpus", "2"] + TOY)
    assert p.returncode == 2 and p.stdout == ""
    err = [ln for ln in p.stderr.splitlines() if ln.strip()]
    assert len(err) == 1 and "needs 2 visible GPUs" in err[0] and "Traceback" not in p.stderr
