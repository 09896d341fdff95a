#!/bin/bash
# classification-phase time of `colorid read_id` (COLORID_TIMING=1) for host thread settings, 3 runs each, on the files
# tools/e2e_demo.py leaves in /tmp/cid_e2e (run that first)
W=/tmp/cid_e2e
BIN=colorid_amd/bin/colorid
f=${1:-reads.bgzf.fastq.gz}
for cfg in "COLORID_GZ_THREADS=8 COLORID_PARSE_THREADS=2" "COLORID_GPU_INFLATE=1 COLORID_PARSE_THREADS=2" "COLORID_GPU_INFLATE=1 COLORID_PARSE_THREADS=4" "COLORID_GPU_INFLATE=1 COLORID_PARSE_THREADS=6" "COLORID_GPU_INFLATE=1 COLORID_PARSE_THREADS=8"; do
  line="$cfg :"
  for rep in 1 2 3; do
    t=$(env COLORID_TIMING=1 $cfg $BIN read_id -b $W/idx.bxi -q $W/$f -n $W/rid_x 2>&1 >/dev/null | grep -o "timing: total [0-9]* ms" | grep -o "[0-9]*")
    line="$line $t"
  done
  echo "$line"
  cmp $W/rid_x_reads.txt $W/rid_b_reads.txt && echo "  same rows"
done
env COLORID_TIMING=1 COLORID_GPU_INFLATE=1 COLORID_PARSE_THREADS=4 $BIN read_id -b $W/idx.bxi -q $W/$f -n $W/rid_x 2>&1 >/dev/null | grep "timing:" | cut -c1-330
