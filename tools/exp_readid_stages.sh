#!/bin/bash
# classification-phase time of `colorid read_id` (COLORID_TIMING=1) on the plain-gzip file of tools/e2e_demo.py (run that first):
# zlib raw inflate + libdeflate CRC-32 (default when libdeflate is there) against gzread (COLORID_LIBDEFLATE=0), alternating runs;
# then the same file four times over (multi-member) and the paired case
W=/tmp/cid_e2e
BIN=colorid_amd/bin/colorid
cat $W/reads.fastq.gz $W/reads.fastq.gz $W/reads.fastq.gz $W/reads.fastq.gz > $W/reads4.fastq.gz
for f in reads.fastq.gz reads4.fastq.gz; do
for cfg in "A=1" "COLORID_LIBDEFLATE=0" "A=1" "COLORID_LIBDEFLATE=0" "A=1" "COLORID_LIBDEFLATE=0"; do
  t=$(env COLORID_TIMING=1 $cfg $BIN read_id -b $W/idx.bxi -q $W/$f -n $W/rid_x 2>&1 >/dev/null | grep -o "timing: total [0-9]* ms\|classification [0-9]* ms" | tr '\n' ' ')
  echo "$f $cfg : $t"
done
done
cmp $W/rid_x_reads.txt <(cat $W/rid_reads.txt $W/rid_reads.txt $W/rid_reads.txt $W/rid_reads.txt) && echo same rows
