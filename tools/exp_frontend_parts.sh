# where the host thread of the device front end waits inside cid_fastq_classify (CID_FASTQ_TIMING=1), 16 M reads
# after tools/e2e_demo.py (E2E_GENOMES=256 E2E_GROUPS=0) + tools/exp_batch_id.sh (writes reads.bgzf.fastq.gz)
W=/tmp/cid_e2e
BIN=colorid_amd/bin/colorid
[ -f $W/reads16.bgzf.fastq.gz ] || { cat $W/reads.bgzf.fastq.gz $W/reads.bgzf.fastq.gz $W/reads.bgzf.fastq.gz $W/reads.bgzf.fastq.gz > $W/reads4.bgzf.fastq.gz; cat $W/reads4.bgzf.fastq.gz $W/reads4.bgzf.fastq.gz $W/reads4.bgzf.fastq.gz $W/reads4.bgzf.fastq.gz > $W/reads16.bgzf.fastq.gz; }
run() { cfg="$1"; for rep in 1 2 3; do echo "$cfg:"; env $cfg CID_FASTQ_TIMING=1 COLORID_TIMING=1 $BIN read_id -b $W/idx.bxi -q $W/reads16.bgzf.fastq.gz -n $W/rid_pt 2>&1 >/dev/null | tr '\r' '\n' | grep -E "cid_fastq:|timing: (device|classification)" | cut -c1-330; done; }
run "A=default"
run "CID_INFLATE_PRIORITY=0"
run "COLORID_DEVICE_FASTQ_HOST_SHARE=0"
