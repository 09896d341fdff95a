# Collect the round's rocprofv3 evidence for bench.py's dominant kernel (run on the GPU box via gpurun).
# Kernel timing and PMC counters are separate runs (guide: MI355X_MICROARCH.md §HBM / rocprofv3 PMC slots).
TAG=${1:-r01}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/$TAG; export TMPDIR=/tmp
BENCH="python3 bench.py --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/stats -- $BENCH --steps 20 --warmup 3 > gpurun_out/$TAG/bench_stats.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d gpurun_out/$TAG/pmc_rdreq -- $BENCH --steps 3 --warmup 1 > gpurun_out/$TAG/pmc_rdreq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/$TAG/pmc_fetch -- $BENCH --steps 3 --warmup 1 > gpurun_out/$TAG/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/$TAG/pmc_write -- $BENCH --steps 3 --warmup 1 > gpurun_out/$TAG/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d gpurun_out/$TAG/pmc_sq -- $BENCH --steps 3 --warmup 1 > gpurun_out/$TAG/pmc_sq.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_DRAM_sum --output-format csv -d gpurun_out/$TAG/pmc_l2 -- $BENCH --steps 3 --warmup 1 > gpurun_out/$TAG/pmc_l2.log 2>&1
# keep only the rows of our kernels from the big CSVs (gpurun_out is capped at 64 MiB)
for d in pmc_rdreq pmc_fetch pmc_write pmc_sq pmc_l2; do
  f=$(find gpurun_out/$TAG/$d -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && (head -1 $f; grep "k_search_count" $f) > gpurun_out/$TAG/$d.csv && rm -rf gpurun_out/$TAG/$d
done
f=$(find gpurun_out/$TAG/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/$TAG/kernel_stats.csv
f=$(find gpurun_out/$TAG/stats -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && (head -1 $f; grep "k_search_count\|k_insert" $f) > gpurun_out/$TAG/kernel_trace_cid.csv
rm -rf gpurun_out/$TAG/stats
tail -1 gpurun_out/$TAG/bench_stats.log | cut -c1-400
ls -la gpurun_out/$TAG
