#!/bin/bash
# classification-phase time of `colorid read_id` (COLORID_TIMING=1) with the default pools on 1 M and 4 M reads (the BGZF and plain-gzip files of
# tools/e2e_demo.py, once and four times over; run that first), 4 runs each
W=/tmp/cid_e2e
BIN=colorid_amd/bin/colorid
cat $W/reads.bgzf.fastq.gz $W/reads.bgzf.fastq.gz $W/reads.bgzf.fastq.gz $W/reads.bgzf.fastq.gz > $W/reads4.bgzf.fastq.gz
for f in reads4.bgzf.fastq.gz reads.bgzf.fastq.gz reads.fastq.gz; do
  line="$f :"
  for rep in 1 2 3 4; do
    t=$(env COLORID_TIMING=1 $BIN read_id -b $W/idx.bxi -q $W/$f -n $W/rid_x 2>&1 >/dev/null | grep -o "timing: total [0-9]* ms" | grep -o "[0-9]*")
    line="$line $t"
  done
  echo "$line"
done
