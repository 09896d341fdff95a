#!/bin/bash
# classification-phase time of `colorid read_id` (COLORID_TIMING=1) on 4 M reads (the BGZF file of tools/e2e_demo.py four times over; run
# that first): members inflated by the host's threads (default) or on the GPU (COLORID_GPU_INFLATE=1), 3 runs each
W=/tmp/cid_e2e
BIN=colorid_amd/bin/colorid
cat $W/reads.bgzf.fastq.gz $W/reads.bgzf.fastq.gz $W/reads.bgzf.fastq.gz $W/reads.bgzf.fastq.gz > $W/reads4.bgzf.fastq.gz
for cfg in "A=1" "COLORID_GPU_INFLATE=1" "COLORID_GPU_INFLATE=1 COLORID_GPU_INFLATE_MB=256" "COLORID_GPU_INFLATE=1 COLORID_PARSE_THREADS=4" "COLORID_GPU_INFLATE=1 COLORID_GPU_INFLATE_MB=64 COLORID_PARSE_THREADS=4"; do
  line="$cfg :"
  for rep in 1 2 3; do
    t=$(env COLORID_TIMING=1 $cfg $BIN read_id -b $W/idx.bxi -q $W/reads4.bgzf.fastq.gz -n $W/rid_x 2>&1 >/dev/null | grep -o "timing: total [0-9]* ms" | grep -o "[0-9]*")
    line="$line $t"
  done
  echo "$line"
done
