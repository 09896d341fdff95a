# HBM line requests of k_search_count over a k-mer set in code order and over the set built FOR the index (rocprofv3 --pmc passes over
# tools/exp_set_search.py; counters for k_search_count only; every pass under its own timeout).  Run on the GPU box from the repo root.
TAG=${TAG:-r04}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/${TAG}_set; export TMPDIR=/tmp
for HOW in code target; do
  python3 tools/exp_set_search.py $HOW 10 > gpurun_out/${TAG}_set/wall_$HOW.json 2> gpurun_out/${TAG}_set/wall_$HOW.err
  i=0
  for PASS in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout 240 rocprofv3 --pmc $PASS --kernel-include-regex "k_search_count" --output-format csv -d gpurun_out/${TAG}_set/p_${HOW}_$i -- python3 tools/exp_set_search.py $HOW 3 > gpurun_out/${TAG}_set/p_${HOW}_$i.log 2>&1
    f=$(find gpurun_out/${TAG}_set/p_${HOW}_$i -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && (head -1 $f; grep "k_search_count" $f) > gpurun_out/${TAG}_set/pmc_${HOW}_$i.csv
    rm -rf gpurun_out/${TAG}_set/p_${HOW}_$i
  done
done
python3 - <<PY
import csv, glob, collections, json
out = {}
for f in sorted(glob.glob("gpurun_out/${TAG}_set/pmc_*_?.csv")):
    how = f.split("pmc_")[1].split("_")[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    out.setdefault(how, {}).update({k: sum(v) / len(v) for k, v in acc.items()})
for how, v in out.items():
    v["read_GB"] = 128 * v.get("TCC_EA0_RDREQ_128B_sum", 0) / 1e9
    v["read_GB_fetch_size_x2"] = 2 * 1024 * v.get("FETCH_SIZE", 0) / 1e9
    v["write_GB"] = 1024 * v.get("WRITE_SIZE", 0) / 1e9
    v["l2_hit_rate"] = v.get("TCC_HIT_sum", 0) / max(1.0, v.get("TCC_HIT_sum", 0) + v.get("TCC_MISS_sum", 0))
    try:
        v["wall"] = json.load(open("gpurun_out/${TAG}_set/wall_%s.json" % how))
    except Exception as e:
        v["wall"] = str(e)
print(json.dumps(out, indent=1))
json.dump(out, open("gpurun_out/${TAG}_set_search_pmc.json", "w"), indent=1)
PY
