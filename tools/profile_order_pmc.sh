# L2 / fabric counters of k_search_count for three input orders (code order, 2^12 slices, exact line) with the persistent XCD-queue
# kernel: rocprofv3 --pmc passes over tools/exp_order_one.py.  Run on the GPU box from the repo root.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02_order; export TMPDIR=/tmp
for ORDER in none 12 0; do
  for PASS in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE WRITE_SIZE"; do
    T=$(echo $PASS | cut -d' ' -f1)
    rocprofv3 --pmc $PASS --output-format csv -d gpurun_out/r02_order/p_${ORDER}_$T -- python3 tools/exp_order_one.py $ORDER 1 > gpurun_out/r02_order/p_${ORDER}_$T.log 2>&1
    f=$(find gpurun_out/r02_order/p_${ORDER}_$T -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && (head -1 $f; grep "k_search_count" $f) > gpurun_out/r02_order/pmc_${ORDER}_$T.csv
    rm -rf gpurun_out/r02_order/p_${ORDER}_$T
  done
done
python3 - <<'PY'
import csv, glob, collections, json
out = {}
for f in sorted(glob.glob("gpurun_out/r02_order/pmc_*.csv")):
    order = f.split("pmc_")[1].split("_")[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    out.setdefault(order, {}).update({k: sum(v) / len(v) for k, v in acc.items()})
print(json.dumps(out, indent=1))
json.dump(out, open("gpurun_out/r02_order/summary.json", "w"), indent=1)
PY
