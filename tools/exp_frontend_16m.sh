# steady state of the device front end against the host front end: 16 M reads (the block-gzip file of tools/e2e_demo.py sixteen times over),
# single-end and paired; COLORID_TIMING=1 phase lines.  Run tools/e2e_demo.py (E2E_GENOMES=256 E2E_GROUPS=0) first.
W=/tmp/cid_e2e
BIN=colorid_amd/bin/colorid
[ -f $W/reads4.bgzf.fastq.gz ] || cat $W/reads.bgzf.fastq.gz $W/reads.bgzf.fastq.gz $W/reads.bgzf.fastq.gz $W/reads.bgzf.fastq.gz > $W/reads4.bgzf.fastq.gz
cat $W/reads4.bgzf.fastq.gz $W/reads4.bgzf.fastq.gz $W/reads4.bgzf.fastq.gz $W/reads4.bgzf.fastq.gz > $W/reads16.bgzf.fastq.gz
run() { cfg=$1; shift; for rep in 1 2 3; do echo "$cfg [$#]: $(env $cfg COLORID_TIMING=1 $BIN read_id -b $W/idx.bxi -q "$@" -n $W/rid_p$# 2>&1 >/dev/null | tr '\r' '\n' | grep -E "timing: (device|classification|total)" | sed 's/; of the GPU calls.*//; s/timing: //; s/waits: parser on a full queue//' | tr '\n' '|' | cut -c1-700)"; done; }
run "COLORID_DEVICE_FASTQ=0" $W/reads16.bgzf.fastq.gz
cp $W/rid_p1_reads.txt $W/rid_host16_reads.txt
run "A=default" $W/reads16.bgzf.fastq.gz
cmp $W/rid_p1_reads.txt $W/rid_host16_reads.txt && echo "same rows (16 M single-end)"
run "COLORID_DEVICE_FASTQ_HOST_SHARE=0" $W/reads16.bgzf.fastq.gz
run "COLORID_DEVICE_FASTQ_HOST_SHARE=0.5" $W/reads16.bgzf.fastq.gz
run "COLORID_DEVICE_FASTQ=0" $W/reads4.bgzf.fastq.gz $W/reads4.bgzf.fastq.gz
cp $W/rid_p2_reads.txt $W/rid_host4pe_reads.txt
run "COLORID_DEVICE_FASTQ=1" $W/reads4.bgzf.fastq.gz $W/reads4.bgzf.fastq.gz
cmp $W/rid_p2_reads.txt $W/rid_host4pe_reads.txt && echo "same rows (4 M pairs)"
run "COLORID_DEVICE_FASTQ=0" $W/reads.bgzf.fastq.gz
run "A=default" $W/reads.bgzf.fastq.gz
