#!/bin/bash
# `colorid read_id` on block-gzip input: the device front end (cid_fastq_*, default on one GPU) against the host front end
# (COLORID_DEVICE_FASTQ=0: inflating + packing threads), 1 M and 4 M reads, single-end and paired, COLORID_TIMING=1 phase lines.
# Run tools/e2e_demo.py first (E2E_GENOMES=256 E2E_GROUPS=0): it leaves the index and the reads under /tmp/cid_e2e.
W=/tmp/cid_e2e
BIN=colorid_amd/bin/colorid
[ -f $W/reads4.bgzf.fastq.gz ] || cat $W/reads.bgzf.fastq.gz $W/reads.bgzf.fastq.gz $W/reads.bgzf.fastq.gz $W/reads.bgzf.fastq.gz > $W/reads4.bgzf.fastq.gz
run() {   # label, env, files...
  label=$1; shift; envs=$1; shift
  for rep in 1 2 3; do
    s=$(date +%s.%N)
    env COLORID_TIMING=1 $envs $BIN read_id -b $W/idx.bxi -q "$@" -n $W/rid_$label 2> $W/err_$label >/dev/null
    e=$(date +%s.%N)
    echo "$label wall $(python3 -c "print(round($e-$s,3))") s | $(grep 'timing: classification' $W/err_$label | tr -d '\r') | $(grep 'timing: total' $W/err_$label | sed 's/; of the GPU calls.*//' | cut -c1-200)"
  done
}
run host1 "COLORID_DEVICE_FASTQ=0" $W/reads.bgzf.fastq.gz
run dev1 "A=1" $W/reads.bgzf.fastq.gz
run host4 "COLORID_DEVICE_FASTQ=0" $W/reads4.bgzf.fastq.gz
run dev4 "A=1" $W/reads4.bgzf.fastq.gz
run dev4_96mb "COLORID_DEVICE_FASTQ_MB=96" $W/reads4.bgzf.fastq.gz
run dev4_24mb "COLORID_DEVICE_FASTQ_MB=24" $W/reads4.bgzf.fastq.gz
run host4pe "COLORID_DEVICE_FASTQ=0" $W/reads4.bgzf.fastq.gz $W/reads4.bgzf.fastq.gz
run dev4pe "A=1" $W/reads4.bgzf.fastq.gz $W/reads4.bgzf.fastq.gz
cmp $W/rid_host4_reads.txt $W/rid_dev4_reads.txt && echo "same rows (4 M single-end)"
cmp $W/rid_host4pe_reads.txt $W/rid_dev4pe_reads.txt && echo "same rows (4 M pairs)"
