#!/bin/bash
# k_bgzf_inflate with 1 / 2 / 4 / 8 members per wave (CID_INFLATE_LANES): kernel times of tools/exp_inflate.py's launches under rocprofv3
export TMPDIR=/tmp
for l in 1 2 4 8; do
  rm -rf /tmp/prof
  CID_INFLATE_LANES=$l timeout 250 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -- python3 tools/exp_inflate.py > /dev/null 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob("/tmp/prof/**/*kernel_trace.csv", recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "bgzf" in r["Kernel_Name"]]
d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6 for r in rows]
print("lanes $l:", len(d), "launches; whole 1M-read batch:", [round(x,2) for x in d[-3:]], "batches of 1024:", [round(x,2) for x in d[-18:-15]], "batches of 256:", [round(x,2) for x in d[5:8]])
PY
done
