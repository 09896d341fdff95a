# TLB / L2 counters of k_search_count against two identical indexes in different allocations (tools/exp_alias3_one.py).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02_alias; export TMPDIR=/tmp
i=0
for PASS in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum" "GRBM_GUI_ACTIVE GRBM_UTCL2_BUSY" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $PASS --kernel-include-regex "k_search_count" --output-format csv -d gpurun_out/r02_alias/p_$i -- python3 tools/exp_alias3_one.py > gpurun_out/r02_alias/p_$i.log 2>&1
  f=$(find gpurun_out/r02_alias/p_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
by = collections.defaultdict(dict)
for r in rows: by[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
for d in sorted(by): print(d, by[d])
PY
  tail -6 gpurun_out/r02_alias/p_$i.log | cut -c1-160
  rm -rf gpurun_out/r02_alias/p_$i
done
