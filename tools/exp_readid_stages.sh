#!/bin/bash
# stage times of `colorid read_id` (COLORID_TIMING=1) on 4 M reads (the BGZF file of tools/e2e_demo.py four times over; run that first)
W=/tmp/cid_e2e
BIN=colorid_amd/bin/colorid
cat $W/reads.bgzf.fastq.gz $W/reads.bgzf.fastq.gz $W/reads.bgzf.fastq.gz $W/reads.bgzf.fastq.gz > $W/reads4.bgzf.fastq.gz
for cfg in "A=1" "A=2" "COLORID_POLL_THREADS=2" "COLORID_POLL_THREADS=3" "COLORID_POLL_THREADS=2 COLORID_PARSE_THREADS=4" "COLORID_GPU_INFLATE=1 COLORID_POLL_THREADS=2 COLORID_PARSE_THREADS=4"; do
  echo "== $cfg"
  env COLORID_TIMING=1 $cfg $BIN read_id -b $W/idx.bxi -q $W/reads4.bgzf.fastq.gz -n $W/rid_x 2>&1 >/dev/null | grep "timing: total" | sed 's/of the GPU calls.*of poll/of poll/' | cut -c1-260
done
cmp $W/rid_x_reads.txt <(cat $W/rid_b_reads.txt $W/rid_b_reads.txt $W/rid_b_reads.txt $W/rid_b_reads.txt) && echo same rows
