#!/bin/bash
# one profiled run of the device front end on the 16 M-read block-gzip file (after tools/e2e_demo.py + tools/exp_frontend.sh left it in /tmp/cid_e2e)
W=/tmp/cid_e2e
BIN=colorid_amd/bin/colorid
export TMPDIR=/tmp
for rep in 1 2; do COLORID_TIMING=1 $BIN read_id -b $W/idx.bxi -q $W/reads16.bgzf.fastq.gz -n $W/rid_p 2>&1 >/dev/null | tr '\r' '\n' | grep "timing:" | cut -c1-250; done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03_frontend_prof -- $BIN read_id -b $W/idx.bxi -q $W/reads16.bgzf.fastq.gz -n $W/rid_p > /dev/null 2> gpurun_out/r03_frontend_prof.err
f=$(ls gpurun_out/r03_frontend_prof/*/*kernel_stats.csv | head -1); python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    print(f"{r['Name'][:80]:80s} calls={r['Calls']:>4s} total_ms={float(r['TotalDurationNs'])/1e6:8.2f}")
PY
