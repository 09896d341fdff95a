# 16 M reads through the device front end: the inflate launches on a highest-priority queue (default) against an ordinary one, and
# members per wave.  After tools/e2e_demo.py (E2E_GENOMES=256 E2E_GROUPS=0) + tools/exp_batch_id.sh (writes reads.bgzf.fastq.gz).
W=/tmp/cid_e2e
BIN=colorid_amd/bin/colorid
[ -f $W/reads16.bgzf.fastq.gz ] || { cat $W/reads.bgzf.fastq.gz $W/reads.bgzf.fastq.gz $W/reads.bgzf.fastq.gz $W/reads.bgzf.fastq.gz > $W/reads4.bgzf.fastq.gz; cat $W/reads4.bgzf.fastq.gz $W/reads4.bgzf.fastq.gz $W/reads4.bgzf.fastq.gz $W/reads4.bgzf.fastq.gz > $W/reads16.bgzf.fastq.gz; }
run() { cfg="$1"; for rep in 1 2 3; do echo "$cfg: $(env $cfg COLORID_TIMING=1 $BIN read_id -b $W/idx.bxi -q $W/reads16.bgzf.fastq.gz -n $W/rid_pr 2>&1 >/dev/null | tr '\r' '\n' | grep -E "timing: (device|classification)" | sed 's/timing: //' | tr '\n' '|' | cut -c1-300)"; done; }
run "CID_INFLATE_PRIORITY=0"
cp $W/rid_pr_reads.txt $W/rid_pr0_reads.txt
run "A=default"
cmp $W/rid_pr_reads.txt $W/rid_pr0_reads.txt && echo "same rows"
run "CID_INFLATE_LANES=1"
run "COLORID_DEVICE_FASTQ_AHEAD=2"
run "COLORID_DEVICE_FASTQ_HOST_SHARE=0"
run "COLORID_DEVICE_FASTQ_HOST_SHARE=0 CID_INFLATE_PRIORITY=0"
