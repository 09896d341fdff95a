# Where the wall time of a one-GPU CLI run goes (run on the GPU box from the repo root, after `E2E_GENOMES=256 E2E_GROUPS=0 python3 tools/e2e_demo.py`:
# 2.8 GB .bxi of 256 colours, 1 M reads as a single-stream fastq.gz and as block gzip).  Per run: the whole-process wall clock, the CLI's own
# phase lines (COLORID_TIMING=1: GPU context, index load, classification, counts file, release), what precedes main (dynamic loading, measured
# with LD_DEBUG=statistics) and what follows the subcommand (the difference).  Each case three times, with the orderly teardown
# (COLORID_FULL_TEARDOWN=1: round 3's behaviour) and without it (COLORID_FAST_EXIT=1: what a run does by default when nothing in the process writes at
# exit — the GPU boxes of this pool preload a guard library, which makes the default the orderly way there).
W=/tmp/cid_e2e
BIN=colorid_amd/bin/colorid
run() {  # run <label> <env...> -- <args...>
  local label=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  for rep in 1 2 3; do
    local s=$(date +%s.%N)
    env COLORID_TIMING=1 "${envs[@]}" $BIN "$@" > /dev/null 2> /tmp/cid_e2e/breakdown.err
    local e=$(date +%s.%N)
    python3 - "$label" "$s" "$e" <<'PY'
import re, sys
label, s, e = sys.argv[1], float(sys.argv[2]), float(sys.argv[3])
err = open("/tmp/cid_e2e/breakdown.err", errors="replace").read().replace("\r", "\n")
ph = re.findall(r"timing: ([A-Za-z ]+?) (\d+) ms \(at (\d+) ms\)", err)
last = int(ph[-1][2]) if ph else 0
print(f"{label}: wall {e - s:.2f} s | " + ", ".join(f"{n} {ms}" for n, ms, _ in ph) + f" | after the subcommand returned + before main: {(e - s) * 1e3 - last:.0f} ms")
PY
  done
}
echo "== read_id, block gzip (device front end)"
run "read_id bgzf  fast exit     " COLORID_FAST_EXIT=1 -- read_id -b $W/idx.bxi -q $W/reads.bgzf.fastq.gz -n $W/rid_bd
run "read_id bgzf  full teardown " COLORID_FULL_TEARDOWN=1 -- read_id -b $W/idx.bxi -q $W/reads.bgzf.fastq.gz -n $W/rid_bd
echo "== read_id, single-stream gzip (host front end)"
run "read_id gz    fast exit     " COLORID_FAST_EXIT=1 -- read_id -b $W/idx.bxi -q $W/reads.fastq.gz -n $W/rid_bd
run "read_id gz    full teardown " COLORID_FULL_TEARDOWN=1 -- read_id -b $W/idx.bxi -q $W/reads.fastq.gz -n $W/rid_bd
echo "== search (default report), block gzip"
run "search bgzf   fast exit     " COLORID_FAST_EXIT=1 -- search -b $W/idx.bxi -q $W/reads.bgzf.fastq.gz -f 0 -p 0.005
run "search bgzf   full teardown " COLORID_FULL_TEARDOWN=1 -- search -b $W/idx.bxi -q $W/reads.bgzf.fastq.gz -f 0 -p 0.005
echo "== info (no index upload: context + code objects only)"
run "info          fast exit     " COLORID_FAST_EXIT=1 -- info -b $W/idx.bxi
echo "== dynamic loading before main (LD_DEBUG=statistics, info)"
LD_DEBUG=statistics $BIN info -b $W/idx.bxi 2>&1 >/dev/null | grep -E "total startup time|time needed for relocation|number of relocations:|time needed to load objects" | head -8
ls -la colorid_amd/libcolorid_hip.so colorid_amd/bin/colorid
nm -D colorid_amd/libcolorid_hip.so | grep -c " T cid_"; nm -D colorid_amd/libcolorid_hip.so | grep -c " [TW] _Z"
echo "== the index uploaded from a mapping of the file (default) against the buffered reader (COLORID_INDEX_MMAP=0)"
run "read_id bgzf  mmap          " -- read_id -b $W/idx.bxi -q $W/reads.bgzf.fastq.gz -n $W/rid_bd
run "read_id bgzf  buffered read " COLORID_INDEX_MMAP=0 -- read_id -b $W/idx.bxi -q $W/reads.bgzf.fastq.gz -n $W/rid_bd
cmp $W/rid_bd_reads.txt $W/rid_b_reads.txt && echo "rows identical to e2e_demo's run"
