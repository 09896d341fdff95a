# rocprofv3 evidence for k_readid (tools/bench_readid.py: m=30M n=2 k=21 C=256, 1M synthetic 150-bp reads resident in HBM)
TAG=${1:-r01}; shift   # remaining arguments go to tools/bench_readid.py (e.g. --paired)
EXTRA="$@"
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/${TAG}_readid; export TMPDIR=/tmp
O=gpurun_out/${TAG}_readid
B="python3 tools/bench_readid.py --check 0 $EXTRA"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B --steps 10 > $O/bench_stats.log 2>&1
timeout 240 rocprofv3 --kernel-include-regex "k_readid<" --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $O/pmc_rdreq -- $B --steps 2 > $O/pmc_rdreq.log 2>&1
timeout 240 rocprofv3 --kernel-include-regex "k_readid<" --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $B --steps 2 > $O/pmc_write.log 2>&1
timeout 240 rocprofv3 --kernel-include-regex "k_readid<" --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $O/pmc_sq -- $B --steps 2 > $O/pmc_sq.log 2>&1
timeout 240 rocprofv3 --kernel-include-regex "k_readid<" --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $O/pmc_sq2 -- $B --steps 2 > $O/pmc_sq2.log 2>&1
for d in pmc_rdreq pmc_write pmc_sq pmc_sq2; do
  f=$(find $O/$d -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && (head -1 $f; grep "k_readid<" $f) > $O/$d.csv && rm -rf $O/$d
done
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats.csv
rm -rf $O/stats
grep "^{" $O/bench_stats.log | tail -1 | cut -c1-300
ls $O
