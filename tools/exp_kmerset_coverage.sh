# k-mer counting of reads at coverage (1 M x 150 bp reads of ONE random genome, 1 % substitutions): finalize with the crowded runs' tier
# (k_run_dedupe_sort, cid_rundedupe.hpp) and without it (CID_KMERSET_DEDUPE=0: round 4's kernels), code order and built for an index; the
# sets' digests must agree.  Run on the GPU box from the repo root: bash tools/exp_kmerset_coverage.sh > gpurun_out/r05_kmerset_coverage.txt
for G in 3000000 300000; do
  for T in "" 1; do
    for D in 1 0; do
      echo "genome $G bases (coverage $((150000000 / G))x), $([ -n "$T" ] && echo 'built for an index' || echo 'code order'), CID_KMERSET_DEDUPE=$D:"
      EXP_DIGEST=1 EXP_ITERS=4 EXP_TARGET=$T CID_KMERSET_DEDUPE=$D EXP_GENOME=$G python3 tools/exp_kmerset.py 2>/dev/null
    done
  done
done
echo "all distinct (random reads):"
EXP_ITERS=4 python3 tools/exp_kmerset.py 2>/dev/null
EXP_ITERS=4 EXP_TARGET=1 python3 tools/exp_kmerset.py 2>/dev/null
export TMPDIR=/tmp
for T in "" 1; do
  EXP_TARGET=$T EXP_GENOME=3000000 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/cov_final$T -- python3 tools/exp_kmerset.py > gpurun_out/cov_final$T.log 2>&1
  f=$(find gpurun_out/cov_final$T -name "*kernel_stats.csv" | head -1)
  echo "kernels, 50x, $([ -n "$T" ] && echo 'built for an index' || echo 'code order') (5 calls each; average us):"
  python3 -c "
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    print('  %-64s %4s x %9.1f' % (r['Name'][:64], r['Calls'], float(r['AverageNs'])/1e3))
" $f
done
