#!/bin/bash
# classification-phase time of `colorid read_id` (COLORID_TIMING=1) on 1 M and 4 M reads (the BGZF file of tools/e2e_demo.py, once and
# four times over; run that first): the default pool sizes (from the CPU quota) against fixed ones, 4 runs each
W=/tmp/cid_e2e
BIN=colorid_amd/bin/colorid
cat $W/reads.bgzf.fastq.gz $W/reads.bgzf.fastq.gz $W/reads.bgzf.fastq.gz $W/reads.bgzf.fastq.gz > $W/reads4.bgzf.fastq.gz
for f in reads.bgzf.fastq.gz reads4.bgzf.fastq.gz reads.fastq.gz; do
for cfg in "DEFAULT=1" "COLORID_GZ_THREADS=8 COLORID_PARSE_THREADS=2 COLORID_POLL_THREADS=8"; do
  line="$f $cfg :"
  for rep in 1 2 3 4; do
    t=$(env COLORID_TIMING=1 $cfg $BIN read_id -b $W/idx.bxi -q $W/$f -n $W/rid_x 2>&1 >/dev/null | grep -o "timing: total [0-9]* ms" | grep -o "[0-9]*")
    line="$line $t"
  done
  echo "$line"
done; done
