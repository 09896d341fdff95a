# rocprofv3 evidence for the long-read read_id path (bench.py --only readid_long: cid_readid_count_resident, 150 Mbases as 10 kb reads
# and as a 2 kb / 10 kb / 100 kb mix on configs[2]'s index): plain run, --kernel-trace --stats, then PMC passes of the search kernel.
# Run on the GPU box: bash tools/profile_readid_long.sh r05_readid_long
TAG=${1:-r05_readid_long}
cd $GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
B="python3 bench.py --only readid_long --no-cpu-baseline"
timeout 400 python3 bench.py --only readid_long > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B > $O/bench_stats.log 2>&1
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats.csv
rm -rf $O/stats
if [ -n "$PMC_KERNEL" ]; then
  pmc() { d=$1; shift
    timeout 300 rocprofv3 --pmc "$@" --kernel-include-regex "$PMC_KERNEL" --output-format csv -d $O/$d -- $B > $O/$d.log 2>&1
    f=$(find $O/$d -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $O/$d.csv; rm -rf $O/$d; }
  pmc pmc_rdreq TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
  pmc pmc_write WRITE_SIZE
fi
cut -c1-1500 $O/bench.json
head -30 $O/kernel_stats.csv | cut -c1-200
