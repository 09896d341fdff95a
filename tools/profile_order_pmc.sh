# L2 / fabric counters of k_search_count for two input orders (code order, exact first-row line) — rocprofv3 --pmc passes over
# tools/exp_order_one.py, counters for k_search_count only, one pass per counter group, every pass under its own timeout.
# Run on the GPU box from the repo root.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02_order; export TMPDIR=/tmp
for ORDER in none 0; do
  i=0
  for PASS in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout 240 rocprofv3 --pmc $PASS --kernel-include-regex "k_search_count" --output-format csv -d gpurun_out/r02_order/p_${ORDER}_$i -- python3 tools/exp_order_one.py $ORDER 0 > gpurun_out/r02_order/p_${ORDER}_$i.log 2>&1
    f=$(find gpurun_out/r02_order/p_${ORDER}_$i -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && (head -1 $f; grep "k_search_count" $f) > gpurun_out/r02_order/pmc_${ORDER}_$i.csv
    rm -rf gpurun_out/r02_order/p_${ORDER}_$i
  done
done
python3 - <<'PY'
import csv, glob, collections, json
out = {}
for f in sorted(glob.glob("gpurun_out/r02_order/pmc_*_?.csv")):
    order = f.split("pmc_")[1].split("_")[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    out.setdefault({"none": "code order", "0": "grouped by first-row line"}[order], {}).update({k: sum(v) / len(v) for k, v in acc.items()})
for k, v in out.items():
    v["read_GB"] = 128 * v.get("TCC_EA0_RDREQ_128B_sum", 0) / 1e9
    v["read_GB_fetch_size_x2"] = 2 * 1024 * v.get("FETCH_SIZE", 0) / 1e9
    v["write_GB"] = 1024 * v.get("WRITE_SIZE", 0) / 1e9
    v["l2_hit_rate"] = v.get("TCC_HIT_sum", 0) / max(1.0, v.get("TCC_HIT_sum", 0) + v.get("TCC_MISS_sum", 0))
print(json.dumps(out, indent=1))
json.dump(out, open("gpurun_out/r02_order/summary.json", "w"), indent=1)
PY
