# SQ-level counters of k_search_count for one bench.py shape: bash tools/profile_sq.sh <tag> <bench args...>
TAG=$1; shift
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/$TAG; export TMPDIR=/tmp
BENCH="python3 bench.py --no-cpu-baseline $@"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d gpurun_out/$TAG/a -- $BENCH --steps 2 --warmup 1 > gpurun_out/$TAG/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d gpurun_out/$TAG/b -- $BENCH --steps 2 --warmup 1 > gpurun_out/$TAG/b.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum --output-format csv -d gpurun_out/$TAG/c -- $BENCH --steps 2 --warmup 1 > gpurun_out/$TAG/c.log 2>&1
for d in a b c; do
  f=$(find gpurun_out/$TAG/$d -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then (head -1 $f; grep "k_search_count" $f) > gpurun_out/$TAG/$d.csv; fi
  rm -rf gpurun_out/$TAG/$d
done
python3 - <<PY
import csv, collections, glob
acc = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/$TAG/?.csv")):
    for r in csv.DictReader(open(f)):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
for k in sorted(m): print(f"{k:32s} {m[k]:.4g}")
wc = m.get("SQ_WAVE_CYCLES", 1)
print("ACTIVE_ANY/WAVE_CYCLES", m.get("SQ_ACTIVE_INST_ANY", 0) / wc, "WAIT_ANY/WAVE_CYCLES", m.get("SQ_WAIT_ANY", 0) / wc)
PY
