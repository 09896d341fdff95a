# polling threads x inflating host threads for the 16 M-read device front end (the poll of a stretch takes 18 ms on 4 threads, the GPU side
# 13 ms since the two-halves step): interleaved rounds.  After tools/e2e_demo.py + tools/exp_batch_id.sh.
W=/tmp/cid_e2e
BIN=colorid_amd/bin/colorid
[ -f $W/reads16.bgzf.fastq.gz ] || { cat $W/reads.bgzf.fastq.gz $W/reads.bgzf.fastq.gz $W/reads.bgzf.fastq.gz $W/reads.bgzf.fastq.gz > $W/reads4.bgzf.fastq.gz; cat $W/reads4.bgzf.fastq.gz $W/reads4.bgzf.fastq.gz $W/reads4.bgzf.fastq.gz $W/reads4.bgzf.fastq.gz > $W/reads16.bgzf.fastq.gz; }
one() { rm -f $W/rid_po_reads.txt $W/rid_po_counts.txt; echo "$1: $(env $1 COLORID_TIMING=1 $BIN read_id -b $W/idx.bxi -q $W/reads16.bgzf.fastq.gz -n $W/rid_po 2>&1 >/dev/null | tr '\r' '\n' | grep -E "timing: (device|total|classification)" | sed 's/; of the GPU calls.*//; s/timing: //; s/device front end: //; s/waiting for the file reader/reader/; s/push (H2D of the members)/push/; s/ until the first stretch was pushed/ to first push/; s/GPU calls (copies + kernels)/GPU/; s/waits: parser on a full queue [0-9]* ms, //; s/GPU stage idle [0-9]* ms, //; s/ (+ [0-9]* ms sizing its buffers)//' | tr '\n' '|' | cut -c1-330)"; }
for round in 1 2 3 4; do
  for cfg in ${CFGS:-A=default COLORID_POLL_THREADS=6 COLORID_POLL_THREADS=8}; do
    one "$(echo $cfg | tr ',' ' ')"
  done
done
