W=/tmp/cid_e2e
BIN=colorid_amd/bin/colorid
[ -f $W/reads4.bgzf.fastq.gz ] || cat $W/reads.bgzf.fastq.gz $W/reads.bgzf.fastq.gz $W/reads.bgzf.fastq.gz $W/reads.bgzf.fastq.gz > $W/reads4.bgzf.fastq.gz
cat $W/reads4.bgzf.fastq.gz $W/reads4.bgzf.fastq.gz $W/reads4.bgzf.fastq.gz $W/reads4.bgzf.fastq.gz > $W/reads16.bgzf.fastq.gz
run() { cfg=$1; shift; for rep in 1 2; do echo "$cfg [$#]: $(env $cfg COLORID_TIMING=1 $BIN read_id -b $W/idx.bxi -q "$@" -n $W/rid_p$# 2>&1 >/dev/null | tr '\r' '\n' | grep -E "timing: (device|classification)" | sed 's/timing: //' | tr '\n' '|' | cut -c1-250)"; done; }
for s in 0.6 0.7 0.8 0.9; do run "COLORID_DEVICE_FASTQ_HOST_SHARE=$s" $W/reads16.bgzf.fastq.gz; done
for s in 0.5 0.7 0.85 1; do run "COLORID_DEVICE_FASTQ_HOST_SHARE=$s" $W/reads4.bgzf.fastq.gz $W/reads4.bgzf.fastq.gz; done
