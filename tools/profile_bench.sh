# Collect the round's rocprofv3 evidence for bench.py's dominant kernel (run on the GPU box via gpurun), in the SAME lease as a plain
# bench run, with the box's clocks recorded before and after.  Kernel timing and PMC counters are separate runs (guide:
# MI355X_MICROARCH.md §HBM / rocprofv3 PMC slots); one derived metric per pass; counters only for k_search_count; every pass under
# its own timeout (a two-metric pass once aborted and hung the profiler for 24 minutes).
TAG=${1:-r02}; shift   # remaining arguments go to bench.py (e.g. --colours 1024 for configs[3]'s row width)
EXTRA="$@"
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/$TAG; export TMPDIR=/tmp
BENCH="python3 bench.py --no-cpu-baseline --no-variants $EXTRA"
rocm-smi --showclocks > gpurun_out/$TAG/clocks_before.txt 2>&1
timeout 300 python3 bench.py $EXTRA --steps 20 --warmup 5 > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err
rocm-smi --showclocks > gpurun_out/$TAG/clocks_after_bench.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/stats -- $BENCH --steps 20 --warmup 3 > gpurun_out/$TAG/bench_stats.log 2>&1
pmc() {  # pmc <dir> <counters...>
  d=$1; shift
  timeout 240 rocprofv3 --pmc "$@" --kernel-include-regex "k_search_count" --output-format csv -d gpurun_out/$TAG/$d -- $BENCH --steps 3 --warmup 1 > gpurun_out/$TAG/$d.log 2>&1
  f=$(find gpurun_out/$TAG/$d -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && (head -1 $f; grep "k_search_count" $f) > gpurun_out/$TAG/$d.csv
  rm -rf gpurun_out/$TAG/$d
}
pmc pmc_rdreq TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
pmc pmc_fetch FETCH_SIZE
pmc pmc_write WRITE_SIZE
pmc pmc_sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS
pmc pmc_l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_DRAM_sum
rocm-smi --showclocks > gpurun_out/$TAG/clocks_after_profile.txt 2>&1
f=$(find gpurun_out/$TAG/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/$TAG/kernel_stats.csv
f=$(find gpurun_out/$TAG/stats -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && (head -1 $f; grep "k_search_count\|k_insert" $f) > gpurun_out/$TAG/kernel_trace_cid.csv
rm -rf gpurun_out/$TAG/stats
cut -c1-600 gpurun_out/$TAG/bench.json
ls -la gpurun_out/$TAG
