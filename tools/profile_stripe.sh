# rocprofv3 evidence for the colour-striped step on one GPU (BASELINE configs[4]'s per-GPU share: m = 2^30, n = 3, one 512-colour stripe of
# 64 GiB; bench.py --placement striped): kernel stats and the fabric read requests of k_search_count in stripe mode.  Run on the GPU box.
TAG=${TAG:-r04}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/${TAG}_stripe; export TMPDIR=/tmp
O=gpurun_out/${TAG}_stripe
B="python3 bench.py --placement striped --no-cpu-baseline"
timeout 400 python3 bench.py --placement striped --steps 10 --warmup 2 > $O/bench.json 2> $O/bench.err
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B --steps 10 --warmup 2 > $O/bench_stats.log 2>&1
timeout 400 rocprofv3 --kernel-include-regex "k_search_count" --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $O/pmc_rdreq -- $B --steps 2 --warmup 1 > $O/pmc_rdreq.log 2>&1
timeout 400 rocprofv3 --kernel-include-regex "k_search_count" --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $B --steps 2 --warmup 1 > $O/pmc_write.log 2>&1
for d in pmc_rdreq pmc_write; do
  f=$(find $O/$d -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && (head -1 $f; grep "k_search_count" $f) > $O/$d.csv && rm -rf $O/$d
done
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats.csv
rm -rf $O/stats
cut -c1-400 $O/bench.json
ls $O
