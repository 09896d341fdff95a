# 16 M reads through the device front end: host share of the inflate x threads of the poll (alternating runs).  Needs tools/e2e_demo.py's files.
W=/tmp/cid_e2e
BIN=colorid_amd/bin/colorid
[ -f $W/reads4.bgzf.fastq.gz ] || cat $W/reads.bgzf.fastq.gz $W/reads.bgzf.fastq.gz $W/reads.bgzf.fastq.gz $W/reads.bgzf.fastq.gz > $W/reads4.bgzf.fastq.gz
[ -f $W/reads16.bgzf.fastq.gz ] || cat $W/reads4.bgzf.fastq.gz $W/reads4.bgzf.fastq.gz $W/reads4.bgzf.fastq.gz $W/reads4.bgzf.fastq.gz > $W/reads16.bgzf.fastq.gz
run() { cfg=$1; shift; echo "$cfg [$#]: $(env $cfg COLORID_TIMING=1 $BIN read_id -b $W/idx.bxi -q "$@" -n $W/rid_s$# 2>&1 >/dev/null | tr '\r' '\n' | grep -E "timing: (device|classification|total)" | sed 's/timing: device front end: waiting for the file reader/reader/; s/(H2D of the members) //; s/ (+ \([0-9]*\) ms sizing its buffers)/ + \1/; s/handing the rows to the poll/handover/; s/until the first stretch was pushed/to first push/; s/; waits: parser on a full queue.*of poll + write:/;/; s/timing: //' | tr '\n' '|' | cut -c1-400)"; }
for rep in 1 2 3; do for cfg in ${CFGS:-"A=default" "COLORID_DEVICE_FASTQ_HOST_SHARE=0 COLORID_POLL_THREADS=10" "COLORID_DEVICE_FASTQ_HOST_SHARE=0 COLORID_POLL_THREADS=12" "COLORID_DEVICE_FASTQ_HOST_SHARE=0.15 COLORID_POLL_THREADS=8"}; do
  run "$cfg" $W/reads16.bgzf.fastq.gz
done; done
