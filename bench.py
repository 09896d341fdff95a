#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on BASELINE.json's config, measured on MI355X.

metric  : query k-mers/s through hash -> n row gathers -> AND -> per-colour count (cid_search_count_dev),
          inputs resident in HBM, hits bit-exact vs the CPU oracle on a sample.
workload: configs[1] at the metric's colour count — m = 50,000,000-bit Bloom rows, n = 4 hashes, k = 31,
          C = 256 colours; the distinct canonical 31-mers (with multiplicities) of 1,000,000 synthetic 150-bp
          reads per GPU (1 % substitutions), `search -g -f 0` semantics.
index   : synthetic: Bernoulli(p) background bits, p = 1 - exp(-n*Lg/m) for Lg = 3 Mbp genomes,
          plus an exact Bloom insert (simple_bloom.rs:19-26, on the GPU) of every error-free query k-mer into
          the colour its read was drawn from.
step    : one pass of the hot path over the rank's whole k-mer batch (+ the RCCL all-reduce of the 3*C
          per-colour counters when N > 1).  N > 1: reads are sharded over ranks, the index is replicated.

Side records of the one JSON line (N = 1 only; never `value`; the line stays under 7 KB):
  config.codes_input       the same query as 2-bit codes (cid_search_count_codes_dev), 10 steps
  config.producer_ordered  2-bit codes grouped by the 128-byte index line of their first row as a separate pass: search_ms = the
                           search alone, in_step_ms = grouping + search inside one step
  config.box               this card's clocks, sampled from its sysfs node during a pass of headline launches AFTER the timed steps
  e2e                      reads in pageable host memory -> H2D -> cid_kmerset built FOR the index (window codes + first-row keys,
                           sort on (key, code), run-length) -> cid_search_count_set_report (hits / unique / sum / mode per accession
                           on the device) -> 4*C numbers to the host; best of 6 calls; PCIe-inclusive
  e2e.code_ordered         the same without cid_kmerset_set_target_index (round 3's path)
  e2e.pinned_input         the same as e2e with the reads in page-locked host memory (what the CLI's batches are): the bus at its DMA rate
  rows128                  the headline's k-mers against m, n of the headline and 1024 colours (configs[3]'s index: 128-byte rows)
  readid                   cid_readid_count_dev on configs[2]'s shape (m = 30 M, n = 2, k = 21, 256 colours), 1 M reads and 1 M pairs
                           resident; alg bytes = n rows of 32 B per distinct k-mer of a read + its bases in + its report row out;
                           cpu_baseline = the oracle's read loop on every host core (the reference runs it under rayon)
  readid_long              cid_readid_count_resident on the same index: 150 Mbases as 10 kb reads and as a 2 kb / 10 kb / 100 kb mix
  cpu_baseline             oracle/liborc.so: orc_search_count on 1 thread (the reference's `search` is single-threaded), .all_cores,
                           .faithful_structure (a hash map of rows as in bigsi.rs:19-27, one heap clone per k-mer)

Launch: `python bench.py [--gpus N]` — for N > 1 without a launcher (WORLD_SIZE unset) this process starts N fresh children
        itself (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N`
        as a subprocess, decided before anything here touches the GPU), relays rank 0's JSON line and exits with the children's
        code; fewer than N devices => exit 2 with a one-line reason.  The driver's own torch.distributed.run launch works as before.
        BENCH_BACKEND=gloo (default nccl = RCCL over xGMI): the exchange steps go through host memory and the ranks may share
        a device (rank r uses device r mod device_count) — how tests/test_gpu_bench_launch.py runs the N > 1 branches on one GPU.
        --emulate-world W (one rank): the same W shards / stripes, searched one after the other by one rank; its `counters`
        digest must equal the W-rank run's.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec


_HIP = None


def hip_memcpy(dst, src, nbytes, kind):
    """hipMemcpy between library-owned and torch-owned memory (kind: 2 = D2H, 3 = D2D); synchronous (D2D: made so)."""
    global _HIP
    import ctypes
    if _HIP is None:
        _HIP = ctypes.CDLL("libamdhip64.so")
        _HIP.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        _HIP.hipMemcpy.restype = ctypes.c_int
    rc = _HIP.hipMemcpy(dst, src, nbytes, kind)
    if rc == 0 and kind == 3:
        # a device-to-device hipMemcpy returns before it has run: without this wait the caller's next torch op may reuse the source
        # block (the caching allocator hands it out again at once) and the copy reads the NEXT chunk's bytes — the index background
        # then differs from run to run and from rank to rank
        rc = _HIP.hipDeviceSynchronize()
    if rc != 0:
        raise RuntimeError(f"hipMemcpy failed: {rc}")


def pmc_path(kernel, C, m, n, k):
    """profiles/pmc/<kernel>_C<colours>_m<bloom>_n<hashes>_k<k>.json: ONE file per (kernel, workload), written by tools/pmc_store.py
    from the committed counter rows.  (Round 4 kept one fixed file name for every workload; the C = 1024 pass overwrote the headline's
    and the driver-run line lost its traffic figure.)"""
    return os.path.join("profiles", "pmc", f"{kernel}_C{C}_m{m}_n{n}_k{k}.json")


def profiled_traffic(K, C, m, n, k, kernel="k_search_count"):
    """(HBM bytes per launch of `kernel`, where the figure comes from) — PMC counters cannot be read from inside the timed run, so
    the bytes are those of the committed rocprofv3 --pmc passes over THIS workload (profiles/pmc/, keyed by workload); (None, reason)
    when no such pass is committed.  The source string goes into the JSON line so that a reader sees it was NOT measured in this run."""
    path = pmc_path(kernel, C, m, n, k)
    try:
        with open(os.path.join(ROOT, path)) as f:
            pj = json.load(f)
        if (pj["n_colors"], pj["bloom_size"], pj["num_hash"], pj["k_size"]) != (C, m, n, k):
            return None, f"{path} names another workload inside than in its file name"
        if pj["kmers_per_launch"] != K:
            return None, f"{path} was taken at {pj['kmers_per_launch']} k-mers per launch, this run has {K}"
        return float(pj["traffic_bytes"]), (f"{path} (lease tag {pj.get('tag', '?')}): separate rocprofv3 --pmc passes over this workload, "
                                            "TCC_EA0_RDREQ by request size + WRITE_SIZE with the guide's gfx950 corrections; not measured in this run")
    except (OSError, KeyError, ValueError) as e:
        return None, f"no committed PMC passes for this workload ({path}: {type(e).__name__})"


def traffic_fields(rec, ms, K, C, m, n, k, kernel="k_search_count"):
    """traffic / traffic_source / traffic_GBs / traffic_frac of a record whose kernel took `ms` per launch"""
    tr, src = profiled_traffic(K, C, m, n, k, kernel)
    rec["traffic"], rec["traffic_source"] = tr, src
    if tr:
        rec["traffic_GBs"] = tr / (ms * 1e-3) / 1e9
        rec["traffic_frac"] = tr / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS
    return rec


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--reads", type=int, default=1_000_000, help="synthetic reads per GPU")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--bloom", type=int, default=50_000_000)
    ap.add_argument("--hashes", type=int, default=4)
    ap.add_argument("--k", type=int, default=31)
    ap.add_argument("--colours", type=int, default=256)
    ap.add_argument("--genome-len", type=int, default=3_000_000)
    ap.add_argument("--error-rate", type=float, default=0.01)
    ap.add_argument("--density", type=float, default=None, help="override the background bit density (experiments)")
    ap.add_argument("--codes", action="store_true", help="experiment: feed 2-bit codes (cid_search_count_codes_dev) instead of ASCII")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the oracle baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true", help="only the headline launches (profiling runs: every k_search_count dispatch is a headline step)")
    ap.add_argument("--placement", choices=["replicated", "striped"], default="replicated",
                    help="replicated (default, the metric's config): reads sharded over the GPUs, index replicated; striped = BASELINE "
                         "configs[4]: every GPU holds one colour stripe of an index too large for one HBM and sees every k-mer")
    ap.add_argument("--stripe-colours", type=int, default=512)
    ap.add_argument("--stripe-log2-bloom", type=int, default=30)
    ap.add_argument("--stripe-hashes", type=int, default=3)
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="one rank stands in for W: the index holds all W shards' plants (or all W stripes) and the W shards are searched "
                         "one after the other into the same counters; the `counters` digest equals the W-rank run's (the N > 1 test)")
    ap.add_argument("--scale-check", action="store_true",
                    help="compare the run's `counters` digest with the one committed under tests/golden/scale_digests.json for this workload and "
                         "this number of ranks (made by --emulate-world runs on one GPU, tools/make_scale_digests.py): the line gets a "
                         "`scale_check` record and the exit code is 1 on a mismatch — the first thing to run on an N-GPU node")
    ap.add_argument("--only", default=None, help="profiling: run just this side record (readid_long) and print it as the JSON line")
    ap.add_argument("--traffic-bytes", type=float, default=None,
                    help="HBM bytes per launch from a separate rocprofv3 --pmc pass (profiles/), if known")
    return ap.parse_args()


def make_reads_kmers(dev, seed, n_reads, read_len, k, n_colours, err_rate, return_reads=False, return_codes=False):
    """Distinct canonical k-mers of synthetic reads: (ascii [K,k] u8, freq [K] i32, colour [K] i32; colour >= C = not planted)."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    codes = torch.randint(0, 4, (n_reads, read_len), device=dev, dtype=torch.int64, generator=g)
    genome = torch.randint(0, n_colours, (n_reads,), device=dev, dtype=torch.int64, generator=g)
    err = torch.rand((n_reads, read_len), device=dev, generator=g) < err_rate
    nw = read_len - k + 1
    fwd = torch.zeros((n_reads, nw), device=dev, dtype=torch.int64)
    rc = torch.zeros((n_reads, nw), device=dev, dtype=torch.int64)
    bad = torch.zeros((n_reads, nw), device=dev, dtype=torch.bool)
    for j in range(k):
        c = codes[:, j:j + nw]
        fwd = (fwd << 2) | c
        rc = rc | ((3 - c) << (2 * j))
        bad |= err[:, j:j + nw]
    canon = torch.minimum(fwd, rc).reshape(-1)  # A<C<G<T: numeric min == lexicographic min of the ASCII strings
    reads_ascii = torch.tensor(list(b"ACGT"), device=dev, dtype=torch.uint8)[codes] if return_reads else None
    del fwd, rc, codes, err
    colour = torch.where(bad, torch.full_like(bad, n_colours, dtype=torch.int64), genome[:, None].expand(-1, nw)).reshape(-1)
    del bad
    uniq, inverse, counts = torch.unique(canon, return_inverse=True, return_counts=True)
    del canon
    col_u = torch.full((uniq.numel(),), n_colours, device=dev, dtype=torch.int64)
    col_u.scatter_reduce_(0, inverse, colour, reduce="amin")  # any planted occurrence plants the distinct k-mer
    del inverse, colour
    # random order, as a hash map would iterate (hash order is irrelevant to the row access pattern anyway)
    perm = torch.randperm(uniq.numel(), device=dev, generator=g)
    uniq, counts, col_u = uniq[perm], counts[perm], col_u[perm]
    lut = torch.tensor(list(b"ACGT"), device=dev, dtype=torch.uint8)
    shifts = torch.arange(2 * (k - 1), -1, -2, device=dev, dtype=torch.int64)
    K = uniq.numel()
    ascii_k = torch.empty((K, k), device=dev, dtype=torch.uint8)
    step = 8_000_000
    for s in range(0, K, step):
        ascii_k[s:s + step] = lut[((uniq[s:s + step, None] >> shifts[None, :]) & 3)]
    if return_codes:
        return ascii_k.contiguous(), counts.to(torch.int32).contiguous(), col_u.to(torch.int32).contiguous(), uniq.contiguous()
    if return_reads:
        return ascii_k.contiguous(), counts.to(torch.int32).contiguous(), col_u.to(torch.int32).contiguous(), reads_ascii
    return ascii_k.contiguous(), counts.to(torch.int32).contiguous(), col_u.to(torch.int32).contiguous()


def fill_background(dev, mat_ptr, m, rs, n_colours, p, seed):
    """Bernoulli(p) bits for colours < n_colours, zero padding elsewhere, written into the index matrix
    (m rows x rs u64 words at device address mat_ptr)."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    w64 = (n_colours + 63) // 64
    weights = (torch.ones(64, dtype=torch.int64, device=dev) << torch.arange(64, device=dev, dtype=torch.int64))
    chunk = 1 << 20
    for r0 in range(0, m, chunk):
        r1 = min(m, r0 + chunk)
        bits = torch.rand((r1 - r0, w64 * 64), device=dev, generator=g) < p
        if n_colours < w64 * 64:
            bits[:, n_colours:] = False
        words = torch.zeros((r1 - r0, rs), dtype=torch.int64, device=dev)
        words[:, :w64] = (bits.view(r1 - r0, w64, 64).to(torch.int64) * weights).sum(dim=2)
        torch.cuda.synchronize()
        hip_memcpy(mat_ptr + r0 * rs * 8, words.data_ptr(), words.numel() * 8, 3)


def fill_background_fast(dev, mat_ptr, m, rs, n_colours, p, seed, digits=8):
    """Like fill_background for matrices of tens of GiB: every u64 word is folded from `digits` uniform random words by the
    AND/OR recurrence over the binary expansion of p (bit probability q / 2^digits, q = round(p * 2^digits)) instead of one
    float per bit.  Returns the density actually used."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    w64 = (n_colours + 63) // 64
    q = max(1, round(p * (1 << digits)))
    tail = n_colours % 64
    tail_mask = ((1 << tail) - 1) if tail else -1
    if tail_mask >= 1 << 63:
        tail_mask -= 1 << 64
    chunk = max(1, (1 << 27) // max(rs, w64))  # <= 1 GiB per temporary
    for r0 in range(0, m, chunk):
        nr = min(chunk, m - r0)
        acc = torch.zeros((nr, w64), dtype=torch.int64, device=dev)
        for i in range(digits):  # least significant digit first: P <- (digit + P) / 2
            w = torch.randint(-2**31, 2**31, (nr, 2 * w64), device=dev, dtype=torch.int32, generator=g).view(torch.int64)
            acc = (w | acc) if (q >> i) & 1 else (w & acc)
            del w
        if tail:
            acc[:, w64 - 1] &= tail_mask
        if rs != w64:
            words = torch.zeros((nr, rs), dtype=torch.int64, device=dev)
            words[:, :w64] = acc
        else:
            words = acc
        torch.cuda.synchronize()
        hip_memcpy(mat_ptr + r0 * rs * 8, words.data_ptr(), words.numel() * 8, 3)
        del words, acc
    return q / (1 << digits)


def gpu_sysfs_dir(device_index):
    """/sys/bus/pci/devices/<bus id> of HIP device `device_index` (hipDeviceGetPCIBusId): a box shows the sysfs nodes of all of the
    host's cards, but only the one this process was given is under load — `card0` is usually somebody else's idle GPU."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    buf = ctypes.create_string_buffer(64)
    if hip.hipDeviceGetPCIBusId(buf, 64, int(device_index)) != 0:
        return None
    path = os.path.join("/sys/bus/pci/devices", buf.value.decode().lower())
    return path if os.path.isdir(path) else None


def box_clocks(sysfs_dir):
    """Current sclk / mclk / fclk (MHz) of the card at `sysfs_dir`: the hwmon frequency inputs where the driver has them, else the
    starred level of pp_dpm_*.  No child process: this process has initialised the GPU (and may run under rocprofv3's preload),
    where spawning `rocm-smi` — a `#!/usr/bin/env python3` script — is the exec hop the GPU boxes forbid."""
    import glob
    import re
    out = {}
    for path in sorted(glob.glob(os.path.join(sysfs_dir, "hwmon/hwmon*/freq*_input"))):
        try:
            with open(path) as f:
                hz = int(f.read().strip())
            with open(path.replace("_input", "_label")) as f:
                label = f.read().strip()
        except (OSError, ValueError):
            continue
        out.setdefault(label, round(hz / 1e6))
    for name in ("sclk", "mclk", "fclk"):
        if name in out:
            continue
        try:
            with open(os.path.join(sysfs_dir, f"pp_dpm_{name}")) as f:
                cur = [ln for ln in f.read().splitlines() if ln.rstrip().endswith("*")]
        except OSError:
            continue
        if cur:
            mhz = re.search(r"(\d+)\s*mhz", cur[0], re.I)
            if mhz:
                out[name] = int(mhz.group(1))
    return out


class ClockSampler:
    """Samples box_clocks() of THIS process's card from a thread while the timed steps run (an idle GPU reports its parked clocks,
    and the management firmware's reading lags the load by some tens of milliseconds: one reading after the loop says nothing).
    result(): per clock the median and the maximum over the samples taken under load, and how many there were."""

    def __init__(self, device_index, period_s=0.01):
        import threading
        self.dir = None
        try:
            self.dir = gpu_sysfs_dir(device_index)
        except OSError:
            pass
        self.period, self.samples, self._stop = period_s, [], threading.Event()
        self._t = threading.Thread(target=self._run, daemon=True) if self.dir else None

    def _run(self):
        while not self._stop.is_set():
            c = box_clocks(self.dir)
            if c:
                self.samples.append(c)
            self._stop.wait(self.period)

    def start(self):
        if self._t:
            self._t.start()
        return self

    def stop(self):
        self._stop.set()
        if self._t:
            self._t.join()

    def result(self):
        if not self.dir:
            return {"error": "no sysfs node for this HIP device"}
        if not self.samples:
            return {"error": f"no clock files under {self.dir}"}
        out = {"sysfs": self.dir, "samples_under_load": len(self.samples)}
        for name in sorted({k for c in self.samples for k in c}):
            v = sorted(c[name] for c in self.samples if name in c)
            out[f"{name}_mhz"] = {"median": v[len(v) // 2], "max": v[-1]}
        return out


def clocks_under_load(device_index, launch, steps=12):
    """The card's clocks while the headline kernel runs — in a pass of its own AFTER the timed steps (local launches only, no
    collective: the other ranks are not in it).  Round 4 ran the sampler thread inside the timed region of rank 0: a Python thread
    that takes the GIL and queries the SMU through sysfs competes with the launch loop of exactly the rank whose clock is reported."""
    cs = ClockSampler(device_index, period_s=0.02).start()
    for _ in range(steps):
        launch()
    torch.cuda.synchronize()
    cs.stop()
    return cs.result()


def host_collective(dev):
    """The device small bookkeeping tensors of the collectives live on: RCCL ("nccl") wants device tensors; gloo (BENCH_BACKEND=gloo,
    the one-GPU rehearsal of the N > 1 branches) reduces / gathers host tensors."""
    return torch.device("cpu") if dist.is_initialized() and dist.get_backend() == "gloo" else dev


def per_rank_times(dev, world, kernel_ms, collective_ms, step_ms, kmers):
    """[{rank, kmers, kernel_ms, collective_ms, step_ms}] over all ranks (rank 0 prints them): a scaling loss can be attributed to
    the kernel (slower box, HBM placement), to the collective (xGMI / RCCL) or to neither (launch gaps, the barrier)."""
    mine = torch.tensor([kernel_ms, collective_ms, step_ms, float(kmers)], dtype=torch.float64, device=host_collective(dev))
    if world > 1:
        allr = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
    else:
        allr = [mine]
    return [{"rank": r, "kmers": int(t[3]), "kernel_ms": round(float(t[0]), 4), "collective_ms": round(float(t[1]), 4),
             "step_ms": round(float(t[2]), 4)} for r, t in enumerate(allr)]


def max_over_ranks(dev, world, seconds):
    el = torch.tensor([seconds], dtype=torch.float64, device=host_collective(dev))
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    return float(el.item())


def counters_digest(*tensors):
    """sha256 over the per-accession counters of the last step + their sums: a W-rank run and `--emulate-world W` must agree."""
    import hashlib
    h = hashlib.sha256()
    sums = []
    for t in tensors:
        a = t.detach().cpu().numpy().astype(np.int64)
        h.update(a.tobytes())
        sums.append(int(a.sum()))
    return {"sha256": h.hexdigest(), "sums": sums}


def workload_key(a):
    """the parameters that decide the counters, as one string: the key of tests/golden/scale_digests.json"""
    if a.placement == "striped":
        return (f"striped:reads={a.reads}:len={a.read_len}:k={a.k}:colours={a.stripe_colours}:log2bloom={a.stripe_log2_bloom}:hashes={a.stripe_hashes}:"
                f"err={a.error_rate}:density={a.density}")
    return (f"replicated:reads={a.reads}:len={a.read_len}:bloom={a.bloom}:hashes={a.hashes}:k={a.k}:colours={a.colours}:genome={a.genome_len}:"
            f"err={a.error_rate}:density={a.density}:codes={int(a.codes)}")


def scale_check(a, n_ranks, counters):
    """{"key", "ranks", "expected", "got", "ok"}: ok None = no digest is committed for this workload and rank count"""
    rec = {"key": workload_key(a), "ranks": n_ranks, "got": counters["sha256"], "expected": None, "ok": None}
    try:
        with open(os.environ.get("BENCH_SCALE_DIGESTS") or os.path.join(ROOT, "tests", "golden", "scale_digests.json")) as f:   # (tests point it at a copy)
            rec["expected"] = json.load(f).get(rec["key"], {}).get(str(n_ranks))
    except (OSError, ValueError):
        pass
    if rec["expected"] is not None:
        rec["ok"] = rec["expected"] == rec["got"]
    return rec


def self_launch(a):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks as a CHILD `torch.distributed.run` (this process has
    not touched the GPU: torch.cuda.device_count() does not initialise it), pass rank 0's JSON line through, return the exit code."""
    import socket
    import subprocess
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if ndev < 1 or (backend == "nccl" and ndev < a.gpus):
        print(f"bench.py: --gpus {a.gpus} needs {a.gpus} visible GPUs, this box has {ndev} (RCCL puts one rank on one device; "
              "BENCH_BACKEND=gloo lets ranks share a device for a functional rehearsal)", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, env=env, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    for ln in proc.stdout.splitlines():
        if not ln.startswith("{"):
            print(ln, file=sys.stderr)
    if proc.returncode == 0 and len(lines) != 1:
        print(f"bench.py: expected one JSON line from rank 0, got {len(lines)}", file=sys.stderr)
        return 1
    for ln in lines:
        print(ln, flush=True)
    return proc.returncode


def bench_striped(a, json_out, world, rank, device_index, dev, ctx, stream):
    """BASELINE configs[4]: m = 2^30, n = 3, 512 colours per GPU (64 GiB resident per rank; 4096 colours = 512 GiB over 8 GPUs).
    Every rank sees every k-mer (the distinct canonical 31-mers of the same 1 M reads), searches its own stripe
    (cid_search_count_stripe_dev), then ONE all-reduce(SUM) of the packed per-k-mer facts (4 B per k-mer, RCCL over xGMI) + the tiny
    per-colour vector, then the exactly-one-colour rule on every rank (cid_search_unique_finalize_dev).  value = distinct query
    k-mers answered per second against the whole 512*N-colour index (weak scaling: the index grows with N, the query does not)."""
    import colorid_amd
    from colorid_amd.striped import StripedIndex, reduce_stripe_facts
    Cs, n, k, m = a.stripe_colours, a.stripe_hashes, a.k, 1 << a.stripe_log2_bloom
    n_stripes = a.emulate_world or world
    my_stripes = list(range(n_stripes)) if a.emulate_world else [rank]
    C_total = Cs * n_stripes
    t_setup = time.time()
    kmers, freq, colour = make_reads_kmers(dev, 42, a.reads, a.read_len, k, C_total, a.error_rate)   # the same k-mers on every rank
    K = kmers.shape[0]
    held = []
    setup_phases = {"kmers_s": round(time.time() - t_setup, 2), "stripes_s": []}
    for r in my_stripes:
        t_stripe = time.time()
        hx = colorid_amd.Index(ctx, m, n, k, Cs)
        ptr, rs = hx.device_matrix()
        p_bg = fill_background_fast(dev, ptr, m, rs, Cs, a.density if a.density is not None else 1.0 - math.exp(-n * 5_000_000 / m), seed=7 + r)
        base = r * Cs
        mine = torch.where((colour >= base) & (colour < base + Cs), colour - base, torch.full_like(colour, Cs)).contiguous()
        torch.cuda.synchronize()
        hx.insert_kmers_dev(kmers.data_ptr(), mine.data_ptr(), K)
        ctx.synchronize()
        hx.finalize()
        held.append((hx, base))
        setup_phases["stripes_s"].append(round(time.time() - t_stripe, 2))
        del mine
    del colour
    si = StripedIndex(ctx, held, C_total)
    fact = torch.zeros(K, dtype=torch.int32, device=dev)
    hits = torch.zeros(C_total, dtype=torch.int64, device=dev)
    nu = torch.zeros(C_total, dtype=torch.int64, device=dev)
    sf = torch.zeros(C_total, dtype=torch.int64, device=dev)
    uc = torch.empty(K, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    t_setup = time.time() - t_setup
    ev = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(4)) for _ in range(a.steps)]

    def step(i=None):
        fact.zero_(); hits.zero_(); nu.zero_(); sf.zero_()
        if i is not None:
            ev[i][0].record(stream)
        si.search_count_local(kmers, fact, hits)
        if i is not None:
            ev[i][1].record(stream)
        reduce_stripe_facts(fact, hits)      # (torch's RCCL stream; the current stream waits for it before the next record)
        if i is not None:
            ev[i][2].record(stream)
        si.unique_finalize(fact, freq, nu, sf, uc)
        if i is not None:
            ev[i][3].record(stream)

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    digest = counters_digest(hits, nu, sf)   # the timed steps' result, before the clock pass adds further launches into `hits`
    n_hits = int(hits.sum().item())
    clocks = clocks_under_load(device_index, lambda: si.search_count_local(kmers, fact, hits)) if rank == 0 else None
    kern_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
    coll_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in ev]))
    fin_ms = float(np.mean([e[2].elapsed_time(e[3]) for e in ev]))
    ranks = per_rank_times(dev, world, kern_ms, coll_ms, elapsed / a.steps * 1e3, K)
    elapsed = max_over_ranks(dev, world, elapsed)
    if rank == 0:
        w64 = (Cs + 63) // 64
        alg = n * w64 * 8 + k + 4 + 4   # rows + k-mer bytes + the k-mer's packed fact read and written
        achieved = alg * K / (kern_ms * 1e-3) / 1e9
        # consistency: every unique k-mer is counted once, and the planted k-mers are found
        ok = int(nu.sum().item()) == int((uc != -1).sum().item()) and n_hits >= int(0.6 * K)
        result = {
            "metric": "query k-mers/s on 50M-bit n=4 256-colour BIGSI; bit-exact hits vs CPU",
            "value": K * a.steps / elapsed, "unit": "k-mers/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"configs[4] (NOT the metric's config): m=2^{a.stripe_log2_bloom} n={n} k={k}, {Cs} colours per stripe x {n_stripes} stripe(s) on {world} GPU(s) = "
                                   f"{C_total} colours colour-striped, the {K} distinct canonical k-mers of {a.reads} synthetic {a.read_len}bp reads seen by every GPU",
                       "placement": "striped", "kmers": K, "bloom_size": m, "num_hash": n, "k_size": k, "n_colors_total": C_total,
                       "stripe_bytes": m * rs * 8, "background_density": p_bg,
                       "parallelism": f"colour stripes over {world} GPU(s); one all-reduce(SUM) of 4 B per k-mer + 8*C_total bytes per step",
                       "collective_bytes_per_step": 4 * K + 8 * C_total, "setup_s": round(t_setup, 1), "setup_phases": setup_phases, "consistent": bool(ok),
                       "backend": (dist.get_backend() if dist.is_initialized() else None), "emulate_world": a.emulate_world or None,
                       "total_kmers": K, "box": clocks},
            "counters": digest,
            "per_rank": ranks, "kernel_ms": kern_ms, "collective_ms": coll_ms, "finalize_ms": fin_ms,
            "roofline": {"bound": "hbm", "kernel": "k_search_count (stripe mode)", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None, "alg_bytes_per_kmer": alg, "kernel_ms": kern_ms, "kmers_per_launch": K},
            "cpu_baseline": None,
        }
        traffic_fields(result["roofline"], kern_ms, K, Cs, m, n, k, kernel="k_search_count_stripe")
        if a.scale_check:
            result["scale_check"] = scale_check(a, n_stripes, digest)
        json_out.write(json.dumps(result) + "\n")
        json_out.flush()
        exit_code = 1 if a.scale_check and result["scale_check"]["ok"] is False else 0
    for hx, _ in held:
        hx.close()
    ctx.close()
    if dist.is_initialized():
        dist.destroy_process_group()
    return exit_code if rank == 0 else 0


def main():
    a = parse_args()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(self_launch(a))       # before any GPU call: the ranks are fresh child processes
    # stdout carries exactly one line, the JSON: RCCL prints a version banner to C stdout (late, when that is a pipe), so
    # everything else this process or its libraries write to fd 1 goes to stderr
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    a.gpus = world
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    if backend not in ("nccl", "gloo"):
        raise SystemExit(f"BENCH_BACKEND={backend}: nccl (RCCL) or gloo")
    if a.emulate_world and world != 1:
        raise SystemExit("--emulate-world is a one-rank run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product has no CPU path (the oracle is only the cpu_baseline leg)")
    device_index = local_rank % torch.cuda.device_count() if backend == "gloo" else local_rank
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    if world > 1 or os.environ.get("BENCH_FORCE_DIST"):  # BENCH_FORCE_DIST: exercise the collective path with one rank
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    import colorid_amd
    ctx = colorid_amd.Context(device_index)
    # One explicit (non-null) stream for torch ops, RCCL and our kernels: torch.cuda.Event then brackets the
    # kernel on the stream it is launched on.
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    assert stream.cuda_stream != 0
    ctx.set_stream(stream.cuda_stream)

    if a.only:          # a side record alone (profiling passes: every dispatch of the run belongs to it); not the driver's line
        fn = {"readid_long": side_readid_long, "readid": side_readid}[a.only]
        json_out.write(json.dumps({"only": a.only, a.only: fn(a, dev, ctx, stream, with_oracle=not a.no_cpu_baseline)}) + "\n")
        json_out.flush()
        ctx.close()
        return
    if a.placement == "striped":
        return bench_striped(a, json_out, world, rank, device_index, dev, ctx, stream)
    C, n, k, m = a.colours, a.hashes, a.k, a.bloom
    t_setup = time.time()
    hx = colorid_amd.Index(ctx, m, n, k, C)
    ptr, rs = hx.device_matrix()
    p_bg = a.density if a.density is not None else 1.0 - math.exp(-n * a.genome_len / m)
    fill_background(dev, ptr, m, rs, C, p_bg, seed=7)
    torch.cuda.synchronize()
    setup_phases = {"background_s": round(time.time() - t_setup, 2), "shards_s": []}   # every rank plants ALL shards: N of these per rank
    n_shards = a.emulate_world or world
    my_shards = list(range(n_shards)) if a.emulate_world else [rank]
    shards = []                      # (ascii k-mers, multiplicities, 2-bit codes, planted colour) of the shards this rank searches
    for r in range(n_shards):  # the replicated index holds every shard's planted k-mers
        t_shard = time.time()
        kk, ff, cc, codes = make_reads_kmers(dev, 42 + r, a.reads, a.read_len, k, C, a.error_rate, return_codes=True)
        torch.cuda.synchronize()
        hx.insert_kmers_dev(kk.data_ptr(), cc.data_ptr(), kk.shape[0])
        ctx.synchronize()
        if r in my_shards:
            shards.append((kk, ff, codes, cc))
        del kk, ff, cc, codes
        setup_phases["shards_s"].append(round(time.time() - t_shard, 2))
    hx.finalize()
    kmers, freq, codes, planted = shards[0]
    K = sum(sh[0].shape[0] for sh in shards)
    out = torch.zeros(3 * C, dtype=torch.int64, device=dev)  # hits | n_unique | sum_unique_freq (u64 on the device)
    ucs = [torch.empty(sh[0].shape[0], dtype=torch.int32, device=dev) for sh in shards]
    uc = ucs[0]
    torch.cuda.synchronize()
    t_setup = time.time() - t_setup

    from colorid_amd._lib import check, vp
    from colorid_amd.dist import allreduce_counts

    # cid_search_count*_dev overwrites its counters: under --emulate-world every further shard gets its own and is added in
    extra = [torch.zeros(3 * C, dtype=torch.int64, device=dev) for _ in shards[1:]]

    def launch():                    # one launch per shard held (one, except under --emulate-world)
        for (kk, ff, cd, _), u, o in zip(shards, ucs, [out] + extra):
            if a.codes:
                check(hx.lib.cid_search_count_codes_dev(ctx.h, hx.h, vp(cd.data_ptr()), vp(ff.data_ptr()), kk.shape[0], vp(o.data_ptr()),
                                                        vp(o.data_ptr() + 8 * C), vp(o.data_ptr() + 16 * C), vp(u.data_ptr())))
            else:
                hx.search_count_dev(kk.data_ptr(), ff.data_ptr(), kk.shape[0], o.data_ptr(), o.data_ptr() + 8 * C,
                                    o.data_ptr() + 16 * C, u.data_ptr())
        for o in extra:
            out.add_(o)

    def step():
        out.zero_()            # every step is one whole query: fresh counters, search, reduction
        launch()
        allreduce_counts(out)  # RCCL over xGMI when N > 1: sum of the per-accession counters (24*C bytes)

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    ev = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(3)) for _ in range(a.steps)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        out.zero_()
        ev[i][0].record(stream)
        launch()
        ev[i][1].record(stream)
        allreduce_counts(out)        # (torch's RCCL stream; the current stream waits for it before the next record)
        ev[i][2].record(stream)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    final_counters = out.clone()     # the timed steps' result (the clock pass below launches without the reduction)
    clocks = clocks_under_load(device_index, launch) if rank == 0 else None
    kern_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
    coll_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in ev]))
    ranks = per_rank_times(dev, world, kern_ms, coll_ms, elapsed / a.steps * 1e3, K)
    total_kmers = sum(r["kmers"] for r in ranks)
    elapsed = max_over_ranks(dev, world, elapsed)

    result = None
    if rank == 0:
        w64 = (C + 63) // 64
        alg_bytes_per_kmer = n * w64 * 8 + (8 if a.codes else k) + 4 + 4  # rows + k-mer bytes + freq in + unique-colour out
        achieved = alg_bytes_per_kmer * K / (kern_ms * 1e-3) / 1e9
        result = {
            "metric": "query k-mers/s on 50M-bit n=4 256-colour BIGSI; bit-exact hits vs CPU",
            "value": total_kmers * a.steps / elapsed,
            "unit": "k-mers/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"configs[1] @ C={C}: m={m} n={n} k={k}, distinct canonical k-mers of "
                                   f"{a.reads} synthetic {a.read_len}bp reads per GPU (search -g -f 0)",
                       "kmers_per_gpu": K, "bloom_size": m, "num_hash": n, "k_size": k, "n_colors": C,
                       "row_bytes": rs * 8, "index_bytes": m * rs * 8, "background_density": p_bg,
                       "parallelism": f"reads sharded over {world} GPU(s), index replicated, all-reduce(3C u64)",
                       "backend": (dist.get_backend() if dist.is_initialized() else None), "emulate_world": a.emulate_world or None,
                       "total_kmers": total_kmers, "setup_s": round(t_setup, 1), "setup_phases": setup_phases, "box": clocks},
            "counters": counters_digest(final_counters[:C], final_counters[C:2 * C], final_counters[2 * C:]),
            "roofline": {"bound": "hbm", "kernel": "k_search_count", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": None, "traffic_source": None,
                         "alg_bytes_per_kmer": alg_bytes_per_kmer,
                         "kernel_ms": kern_ms, "kmers_per_launch": K},
            "per_rank": ranks, "kernel_ms": kern_ms, "collective_ms": coll_ms,
        }
        # the PMC traffic of profiles/pmc/ at this run's kernel time: what the kernel really moves (every 32-byte row costs a 128-byte line)
        if a.traffic_bytes is not None:
            result["roofline"].update({"traffic": a.traffic_bytes, "traffic_source": "--traffic-bytes (a rocprofv3 --pmc pass of the caller)",
                                       "traffic_GBs": a.traffic_bytes / (kern_ms * 1e-3) / 1e9,
                                       "traffic_frac": a.traffic_bytes / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS})
        else:
            traffic_fields(result["roofline"], kern_ms, shards[0][0].shape[0], C, m, n, k)
        if world == 1 and not a.codes and not a.no_variants and not a.emulate_world:
            # for the record, not the headline: the same query with the k-mers as the 2-bit codes that GPU k-mer counting
            # produces (what `colorid search` feeds the kernel for k <= 32): 8 instead of k input bytes per k-mer
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for i in range(12):
                if i == 2:
                    e0.record(stream)
                out.zero_()
                check(hx.lib.cid_search_count_codes_dev(ctx.h, hx.h, vp(codes.data_ptr()), vp(freq.data_ptr()), K, vp(out.data_ptr()),
                                                        vp(out.data_ptr() + 8 * C), vp(out.data_ptr() + 16 * C), vp(uc.data_ptr())))
            e1.record(stream)
            torch.cuda.synchronize()
            ms_codes = e0.elapsed_time(e1) / 10
            result["config"]["codes_input"] = {"ms_per_step": ms_codes, "kmers_per_s": K / ms_codes * 1e3}
            # and with the k-mers grouped by the index line of their first row (cid_order_codes_for_index_dev): the grouping timed on
            # its own, then the search over the grouped k-mers — a quarter of the row fetches become hits in lines just fetched
            oc, of = torch.empty_like(codes), torch.empty_like(freq)
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            for i in range(4):
                if i == 1:
                    e[0].record(stream)
                check(hx.lib.cid_order_codes_for_index_dev(ctx.h, hx.h, vp(codes.data_ptr()), vp(freq.data_ptr()), K, vp(oc.data_ptr()), vp(of.data_ptr())))
            e[1].record(stream)
            for i in range(10):
                out.zero_()
                check(hx.lib.cid_search_count_codes_dev(ctx.h, hx.h, vp(oc.data_ptr()), vp(of.data_ptr()), K, vp(out.data_ptr()),
                                                        vp(out.data_ptr() + 8 * C), vp(out.data_ptr() + 16 * C), vp(uc.data_ptr())))
            e[2].record(stream)
            torch.cuda.synchronize()
            ordered_counts = out.clone()
            ms_order, ms_search = e[0].elapsed_time(e[1]) / 3, e[1].elapsed_time(e[2]) / 10
            out.zero_()
            check(hx.lib.cid_search_count_codes_dev(ctx.h, hx.h, vp(codes.data_ptr()), vp(freq.data_ptr()), K, vp(out.data_ptr()),
                                                    vp(out.data_ptr() + 8 * C), vp(out.data_ptr() + 16 * C), vp(uc.data_ptr())))
            torch.cuda.synchronize()
            result["config"]["producer_ordered"] = {
                "search_ms": ms_search, "kmers_per_s": K / ms_search * 1e3, "grouping_ms": ms_order,
                "in_step_ms": ms_order + ms_search, "same_counts": bool(torch.equal(ordered_counts, out))}
            del oc, of
            result["e2e"] = e2e_reads_to_report(a, dev, ctx, hx, out, C, k)
            # the other kernels of the path in the same driver-run line (side records, never `value`)
            result["rows128"] = side_rows128(a, dev, ctx, stream, kmers, freq, planted)
            result["readid"] = side_readid(a, dev, ctx, stream, with_oracle=not a.no_cpu_baseline)
            result["readid_long"] = side_readid_long(a, dev, ctx, stream, with_oracle=not a.no_cpu_baseline)
        if world == 1 and not a.no_cpu_baseline and not a.emulate_world:
            result["cpu_baseline"], result["bit_exact"] = cpu_baseline(a, hx, ptr, kmers, freq, C, n, k, m, rs)
    exit_code = 0
    if result is not None:
        if a.scale_check:
            result["scale_check"] = scale_check(a, n_shards, result["counters"])
            exit_code = 1 if result["scale_check"]["ok"] is False else 0
        json_out.write(json.dumps(result) + "\n")
        json_out.flush()
    hx.close()
    ctx.close()
    if dist.is_initialized():
        dist.destroy_process_group()
    return exit_code


def e2e_reads_to_report(a, dev, ctx, hx, counters, C, k):
    """What a user of `colorid search` sees of the GPU, PCIe included (never `value`): the step's reads start in HOST memory, go up,
    are counted into the distinct canonical k-mers on the device (cid_kmerset), searched (the headline kernel on 2-bit codes) and
    reduced to the per-accession report — hits, unique k-mers, their multiplicities' sum and MODE — which is all that comes back."""
    import colorid_amd
    from colorid_amd._lib import check, vp
    _, _, _, reads = make_reads_kmers(dev, 42, a.reads, a.read_len, k, C, a.error_rate, return_reads=True)
    host_reads = reads.cpu().numpy()
    del reads
    so = (np.arange(host_reads.shape[0] + 1, dtype=np.uint64) * a.read_len)
    pinned = torch.from_numpy(host_reads).pin_memory()   # the same reads where a caller keeps them in page-locked memory (the CLI's batches are)
    def run(targeted, src=host_reads):
        times, parts, hits, nd = [], [], None, 0
        for _ in range(6):
            ks = colorid_amd.KmerSet(ctx, k)
            if targeted:
                ks.set_target_index(hx)   # what `colorid search` does: the set is ordered for the index by its own sort
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            check(ks.lib.cid_kmerset_add_seqs(ks.h, vp(src.ctypes.data), vp(so.ctypes.data), host_reads.shape[0], 0))
            t1 = time.perf_counter()
            nd = ks.finalize()
            t2 = time.perf_counter()
            rep = ks.search_count_report(hx)
            t3 = time.perf_counter()
            times.append(t3 - t0)
            parts.append((t1 - t0, t2 - t1, t3 - t2))
            hits = rep[0]
            ks.close()
        return times, parts, hits, nd

    c_times, c_parts, c_hits, _ = run(False)
    times, parts, hits, nd = run(True)
    p_times, p_parts, p_hits, _ = run(True, pinned.numpy())
    p_best = int(np.argmin(p_times))
    c_best = int(np.argmin(c_times))
    best = int(np.argmin(times))
    same = bool(np.array_equal(hits.astype(np.int64), counters[:C].cpu().numpy()))   # the headline step's hits (same reads, same index)
    return {"reads": int(host_reads.shape[0]), "distinct_kmers": int(nd), "ms": times[best] * 1e3,
            "reads_per_s": host_reads.shape[0] / times[best], "kmers_per_s": nd / times[best],
            "phases_ms": {"upload_and_window_codes": parts[best][0] * 1e3, "sort_and_count": parts[best][1] * 1e3,
                          "search_and_report": parts[best][2] * 1e3},
            "all_ms": [round(t * 1e3, 2) for t in times], "same_hits_as_headline": same,
            "pinned_input": {"ms": round(p_times[p_best] * 1e3, 3), "upload_and_window_codes_ms": round(p_parts[p_best][0] * 1e3, 3),
                             "same_hits": bool(np.array_equal(hits, p_hits))},
            "code_ordered": {"ms": c_times[c_best] * 1e3, "same_hits": bool(np.array_equal(hits, c_hits)),
                             "phases_ms": {"upload_and_window_codes": c_parts[c_best][0] * 1e3, "sort_and_count": c_parts[c_best][1] * 1e3,
                                           "search_and_report": c_parts[c_best][2] * 1e3}}}


def side_rows128(a, dev, ctx, stream, kmers, freq, planted, C=1024):
    """The headline batch against 128-byte rows: BASELINE configs[3]'s index (m, n, k of the headline, C = 1024: 6.4 GB), the same
    k-mers planted into the same colours.  At 32-byte rows every row costs a 128-byte HBM line (the headline's 0.25); here a row IS a
    line, so the same kernel family shows what the gather does when nothing is over-fetched."""
    import colorid_amd
    n, k, m = a.hashes, a.k, a.bloom
    hx = colorid_amd.Index(ctx, m, n, k, C)
    ptr, rs = hx.device_matrix()
    p_bg = fill_background_fast(dev, ptr, m, rs, C, 1.0 - math.exp(-n * a.genome_len / m), seed=11)
    K = kmers.shape[0]
    torch.cuda.synchronize()
    hx.insert_kmers_dev(kmers.data_ptr(), planted.data_ptr(), K)
    ctx.synchronize()
    hx.finalize()
    out = torch.zeros(3 * C, dtype=torch.int64, device=dev)
    uc = torch.empty(K, dtype=torch.int32, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    steps = 10
    for i in range(steps + 2):
        if i == 2:
            e0.record(stream)
        out.zero_()
        hx.search_count_dev(kmers.data_ptr(), freq.data_ptr(), K, out.data_ptr(), out.data_ptr() + 8 * C, out.data_ptr() + 16 * C, uc.data_ptr())
    e1.record(stream)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    alg = n * rs * 8 + k + 4 + 4
    # the planted k-mers are found in their colour, and every unique k-mer is counted once
    found = int(out[:C].sum().item()) >= int((planted < a.colours).sum().item())
    ok = found and int(out[C:2 * C].sum().item()) == int((uc != -1).sum().item())
    hx.close()
    del out, uc
    rec = {"kernel": "k_search_count (128-byte rows)", "n_colors": C, "row_bytes": rs * 8, "index_bytes": m * rs * 8, "kmers": K, "ms": ms,
           "kmers_per_s": K / ms * 1e3, "alg_bytes_per_kmer": alg, "achieved_GBs": alg * K / ms / 1e6, "frac": alg * K / ms / 1e6 / HBM_PEAK_GBS,
           "background_density": p_bg, "consistent": bool(ok)}
    return traffic_fields(rec, ms, K, C, m, n, k)


def side_readid(a, dev, ctx, stream, with_oracle, reads=1_000_000, L=150):
    """cid_readid_count_dev (`read_id -d 1 -B 3`, read_id_mt_pe.rs:104-165,282-363) on BASELINE configs[2]'s shape: m = 30,000,000, n = 2,
    k = 21, 256 colours, 1 M single-end reads and 1 M pairs of 150 bp resident in HBM; per-read rows of a sample checked against
    the oracle, which is also timed on every host core (the reference runs this loop under rayon)."""
    import colorid_amd
    C, n, k, m, d, B = 256, 2, 21, 30_000_000, 1, 3
    hx = colorid_amd.Index(ctx, m, n, k, C)
    ptr, rs = hx.device_matrix()
    p_bg = 1.0 - math.exp(-n * 5_000_000 / m)
    fill_background(dev, ptr, m, rs, C, p_bg, seed=7)
    kk, _, cc, seqs = make_reads_kmers(dev, 42, 2 * reads, L, k, C, a.error_rate, return_reads=True)
    torch.cuda.synchronize()
    hx.insert_kmers_dev(kk.data_ptr(), cc.data_ptr(), kk.shape[0])
    ctx.synchronize()
    hx.finalize()
    del kk, cc
    bases = seqs.reshape(-1).contiguous()
    rec = {"config": {"bloom_size": m, "num_hash": n, "k_size": k, "n_colors": C, "row_bytes": rs * 8, "stride_d": d, "start_sample": B,
                      "background_density": p_bg, "read_len": L}}
    oix = None
    for name, mates in (("single_end", 1), ("paired", 2)):
        n_seq = reads * mates
        seq_off = (torch.arange(n_seq + 1, device=dev, dtype=torch.int64) * L).contiguous()
        read0 = (torch.arange(reads + 1, device=dev, dtype=torch.int64) * mates).contiguous()
        report = torch.empty((reads, C + 1), dtype=torch.int32, device=dev)
        nk = torch.empty(reads, dtype=torch.int32, device=dev)
        st = torch.empty(reads, dtype=torch.uint8, device=dev)
        max_bytes, max_win = L * mates, ((L - k) // d + 1) * mates
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        steps = 10
        for i in range(steps + 2):
            if i == 2:
                e0.record(stream)
            hx.readid_count_dev(bases.data_ptr(), seq_off.data_ptr(), read0.data_ptr(), reads, d, B, max_bytes, max_win,
                                report.data_ptr(), nk.data_ptr(), st.data_ptr())
        e1.record(stream)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / steps
        nk_sum = int(nk.to(torch.int64).sum().item())
        alg = nk_sum * n * rs * 8 + n_seq * L + reads * (C + 1) * 4
        r = {"reads": reads, "mates": mates, "ms": ms, "reads_per_s": reads / ms * 1e3, "distinct_kmers_per_s": nk_sum / ms * 1e3,
             "row_gathers_per_s": nk_sum * n / ms * 1e3, "alg_bytes": alg, "achieved_GBs": alg / ms / 1e6, "frac": alg / ms / 1e6 / HBM_PEAK_GBS}
        if with_oracle:
            if oix is None:
                oix = cpu_baseline_readid_index(ptr, m, n, k, C, rs)
            r.update(cpu_baseline_readid(oix, bases, seq_off, read0, report, nk, reads, mates, max_bytes, d, B))
        rec[name] = traffic_fields(r, ms, reads, C, m, n, k, kernel="k_readid_pe" if mates == 2 else "k_readid_se")   # (units per launch: reads)
        del report, nk, st, seq_off, read0
    hx.close()
    return rec


def side_readid_long(a, dev, ctx, stream, with_oracle, total_bases=150_000_000):
    """Long reads through cid_readid_count_resident (read_id_mt_pe.rs:104-165,282-363 takes any read length; kmer.rs:221-243): configs[2]'s
    index, 150 Mbases resident in HBM as 10 kb reads and as a 2 kb / 10 kb / 100 kb mix (a third of the bases each; the 2 kb reads are
    routed to the per-wave LDS kernel inside the same batch).  ms = wall time of one call, host side included (the offsets are host
    arrays: routing and the work lists are made on the host), mean of 5 after 2; rows of a sample of reads against the oracle."""
    import colorid_amd
    C, n, k, m, d, B = 256, 2, 21, 30_000_000, 1, 3
    hx = colorid_amd.Index(ctx, m, n, k, C)
    ptr, rs = hx.device_matrix()
    p_bg = 1.0 - math.exp(-n * 5_000_000 / m)
    fill_background(dev, ptr, m, rs, C, p_bg, seed=7)
    hx.finalize()
    g = torch.Generator(device=dev)
    g.manual_seed(4242)
    lut = torch.tensor(list(b"ACGT"), device=dev, dtype=torch.uint8)
    bases = lut[torch.randint(0, 4, (total_bases,), device=dev, generator=g)].contiguous()
    rec = {"config": {"bloom_size": m, "num_hash": n, "k_size": k, "n_colors": C, "row_bytes": rs * 8, "stride_d": d, "start_sample": B,
                      "background_density": p_bg, "bases": total_bases}}
    third = total_bases // 3
    shapes = {"reads_10kb": [(10_000, total_bases // 10_000)],
              "mix_2k_10k_100k": [(2_000, third // 2_000), (10_000, third // 10_000), (100_000, third // 100_000)]}
    oix = None
    for name, parts in shapes.items():
        lens = np.concatenate([np.full(cnt, L, np.uint64) for L, cnt in parts])
        np.random.default_rng(7).shuffle(lens)          # the mix: lengths interleaved, as a sequencer delivers them
        reads = int(lens.size)
        seq_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        read0 = np.arange(reads + 1, dtype=np.uint64)
        report = torch.empty((reads, C + 1), dtype=torch.int32, device=dev)
        nk = torch.empty(reads, dtype=torch.int32, device=dev)
        st = torch.empty(reads, dtype=torch.uint8, device=dev)
        times = []
        for i in range(7):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            hx.readid_count_resident(bases.data_ptr(), seq_off, read0, d, B, report.data_ptr(), nk.data_ptr(), st.data_ptr())
            torch.cuda.synchronize()
            times.append((time.perf_counter() - t0) * 1e3)
        ms = float(np.mean(times[2:]))
        nk_sum = int(nk.to(torch.int64).sum().item())
        nb = int(seq_off[-1])
        alg = nk_sum * n * rs * 8 + nb + reads * (C + 1) * 4
        r = {"reads": reads, "lengths": [[int(L), int(c)] for L, c in parts], "ms": ms, "all_ms": [round(t, 2) for t in times],
             "bases_per_s": nb / ms * 1e3, "distinct_kmers": nk_sum,
             "row_gathers_per_s": nk_sum * n / ms * 1e3, "alg_bytes": alg, "achieved_GBs": alg / ms / 1e6, "frac": alg / ms / 1e6 / HBM_PEAK_GBS}
        if with_oracle:
            if oix is None:
                oix = cpu_baseline_readid_index(ptr, m, n, k, C, rs)
            r.update(cpu_baseline_readid_long(oix, bases, seq_off, read0, report, nk, d, B))
        rec[name] = r
        if name == "reads_10kb":
            traffic_fields(r, ms, reads, C, m, n, k, kernel="k_readid_slices")   # the search kernel's fetched bytes over the WHOLE call's time
            # Soft-masked reads (a lower-case stretch: its case is kept, SURVEY App. B Q2 — the byte-string path): ONE such read in the batch, and
            # 1 % of the reads.  Until round 5 one lower-case base sent the whole batch through round 1's global sort (2.5 x the time).
            soft = {}
            for tag, every in (("one_read", reads), ("one_pct", 100)):
                b2 = bases.clone()
                for r_i in range(7, reads, every):
                    a = int(seq_off[r_i]) + 4_000
                    b2[a:a + 300] |= 0x20
                ts = []
                for i in range(6):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    hx.readid_count_resident(b2.data_ptr(), seq_off, read0, d, B, report.data_ptr(), nk.data_ptr(), st.data_ptr())
                    torch.cuda.synchronize()
                    ts.append((time.perf_counter() - t0) * 1e3)
                soft[tag] = {"soft_masked_reads": len(range(7, reads, every)), "ms": float(np.mean(ts[2:])), "all_ms": [round(t, 2) for t in ts],
                             "vs_clean": float(np.mean(ts[2:])) / ms}
                if with_oracle:
                    soft[tag].update(cpu_baseline_readid_long(oix, b2, seq_off, read0, report, nk, d, B, sample_bases=4_000_000))
                del b2
            rec["soft_masked"] = soft
        del report, nk, st
    hx.close()
    return rec


def cpu_baseline_readid_long(oix, bases, seq_off, read0, report, nk, d, B, sample_bases=12_000_000):
    """the oracle's read loop (orc_readid_counts) on every host core over the first reads of the batch (about sample_bases bases:
    the oracle keeps a hash set per read, a few hundred thousand windows per second and core); bit-exactness of the device rows"""
    ncpu = os.cpu_count() or 1
    S = int(max(1, min(len(read0) - 1, np.searchsorted(seq_off, sample_bases))))
    hb = bases[:int(seq_off[S])].cpu().numpy()
    t = time.perf_counter()
    want = oix.readid_counts(hb, seq_off[:S + 1].copy(), read0[:S + 1].copy(), d, B, n_threads=min(ncpu, S))
    dt = time.perf_counter() - t
    exact = bool(np.array_equal(want[0], report[:S].cpu().numpy().view(np.uint32)) and
                 np.array_equal(want[1], nk[:S].cpu().numpy().view(np.uint32)))
    return {"bit_exact": exact,
            "cpu_baseline": {"value": int(seq_off[S]) / dt, "unit": "bases/s", "cores": min(ncpu, S), "kind": "port",
                             "sample": f"first {S} reads ({int(seq_off[S])} bases), orc_readid_counts, {dt:.2f}s"}}


def cpu_baseline_readid_index(mat_ptr, m, n, k, C, rs):
    """host copy of the device matrix as the oracle's index (the checker's side of side_readid)"""
    from oracle import orc
    rows = np.empty((m, rs * 2), np.uint32)
    hip_memcpy(rows.ctypes.data, mat_ptr, rows.nbytes, 2)
    oix = orc.Index(m, n, k, C)
    oix.rows()[:] = rows[:, :oix.w32]
    return oix


def cpu_baseline_readid(oix, bases, seq_off, read0, report, nk, reads, mates, max_bytes, d, B):
    """The oracle's per-read loop (orc_readid_counts: read_id_mt_pe.rs:66-165,282-363 restated) on every host core — the reference
    runs it under rayon — over a bounded sample of the same reads; also the bit-exactness check of the device rows."""
    ncpu = os.cpu_count() or 1
    S = min(reads, 4000 * min(ncpu, 64) // mates)
    hb = bases[:S * max_bytes].cpu().numpy()
    so = seq_off[:S * mates + 1].cpu().numpy().astype(np.uint64)
    r0 = read0[:S + 1].cpu().numpy().astype(np.uint64)
    t = time.perf_counter()
    want = oix.readid_counts(hb, so, r0, d, B, n_threads=ncpu)
    dt = time.perf_counter() - t
    exact = bool(np.array_equal(want[0], report[:S].cpu().numpy().view(np.uint32)) and
                 np.array_equal(want[1], nk[:S].cpu().numpy().view(np.uint32)))
    return {"bit_exact": exact,
            "cpu_baseline": {"value": S / dt, "unit": "reads/s", "cores": ncpu, "kind": "port",
                             "sample": f"first {S} reads, orc_readid_counts, {dt:.2f}s"}}


def cpu_baseline(a, hx, mat_ptr, kmers, freq, C, n, k, m, rs):
    """The oracle (kind "port": plain-C restatement, 1 thread like the reference's `search`) on a bounded sample of
    the same k-mers against a host copy of the same index; also the bit-exactness check of the GPU result."""
    from oracle import orc
    w32 = (C + 31) // 32
    oix = orc.Index(m, n, k, C)
    rows = oix.rows()
    step = 4_000_000
    torch.cuda.synchronize()
    for r0 in range(0, m, step):  # device u64 rows -> BitVec<u32> rows (same little-endian bytes)
        nr = min(step, m - r0)
        blk = np.empty((nr, rs * 2), np.uint32)
        hip_memcpy(blk.ctypes.data, mat_ptr + r0 * rs * 8, blk.nbytes, 2)
        rows[r0:r0 + nr, :] = blk[:, :w32]
    for c in range(C):
        oix.set_color(c, f"genome_{c:04d}", a.genome_len - k + 1)
    K = kmers.shape[0]
    probe = min(K, 100_000)
    hk = kmers[:probe].cpu().numpy()
    hf = freq[:probe].cpu().numpy().astype(np.uint64)
    t = time.perf_counter()
    oix.search_count(hk, hf)
    rate = probe / (time.perf_counter() - t)
    S = int(min(K, max(probe, rate * a.cpu_seconds)))
    hk = kmers[:S].cpu().numpy()
    hf = freq[:S].cpu().numpy()
    t = time.perf_counter()
    want = oix.search_count(hk, hf.astype(np.uint64))
    dt = time.perf_counter() - t
    got = hx.search_count(hk, hf.astype(np.uint32))
    exact = all(np.array_equal(w, g) for w, g in zip(want, got))
    # second figure, for honesty about the hardware rather than the reference: the same loop on every host core
    ncpu = os.cpu_count() or 1
    S2 = int(min(K, S * min(ncpu, 64)))
    hk2 = kmers[:S2].cpu().numpy()
    hf2 = freq[:S2].cpu().numpy().astype(np.uint64)
    t = time.perf_counter()
    want2 = oix.search_count_mt(hk2, hf2, ncpu)
    dt2 = time.perf_counter() - t
    got2 = hx.search_count(hk2, hf2.astype(np.uint32))
    exact = exact and all(np.array_equal(w, g) for w, g in zip(want2, got2))
    # third figure: the reference's own data structure (FNV-hashed map row -> bit vector, a heap clone per k-mer), 1 thread
    t = time.perf_counter()
    sparse = oix.sparse_map()
    dt_map = time.perf_counter() - t
    S3 = int(min(S, max(probe, rate * 5.0)))
    t = time.perf_counter()
    want3 = oix.search_count_sparse(sparse, hk[:S3], hf[:S3].astype(np.uint64))
    dt3 = time.perf_counter() - t
    orc.sparse_free(sparse)
    exact = exact and all(np.array_equal(w, g) for w, g in zip(want3, hx.search_count(hk[:S3], hf[:S3].astype(np.uint32))[:3]))
    base = {"value": S / dt, "unit": "k-mers/s", "cores": 1, "kind": "port",
            "sample": f"first {S} of the {K} query k-mers, orc_search_count on a host copy of the index, {dt:.1f}s; host has {ncpu} cores",
            "all_cores": {"value": S2 / dt2, "cores": ncpu, "sample": f"first {S2} k-mers, orc_search_count_mt, {dt2:.1f}s"},
            "faithful_structure": {"value": S3 / dt3, "cores": 1,
                                   "sample": f"first {S3} k-mers, orc_search_count_sparse, {dt3:.1f}s after {dt_map:.1f}s building the map"}}
    return base, bool(exact)


if __name__ == "__main__":
    sys.exit(main() or 0)
