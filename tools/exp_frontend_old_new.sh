# the build before the two-halves step (tools/ab_old/: libcolorid_hip.so + bin/colorid of the commit before) against the current one,
# interleaved on the same box: 1 M reads, 16 M reads, 4 M pairs.  After tools/e2e_demo.py + tools/exp_batch_id.sh.
W=/tmp/cid_e2e
[ -f $W/reads16.bgzf.fastq.gz ] || { cat $W/reads.bgzf.fastq.gz $W/reads.bgzf.fastq.gz $W/reads.bgzf.fastq.gz $W/reads.bgzf.fastq.gz > $W/reads4.bgzf.fastq.gz; cat $W/reads4.bgzf.fastq.gz $W/reads4.bgzf.fastq.gz $W/reads4.bgzf.fastq.gz $W/reads4.bgzf.fastq.gz > $W/reads16.bgzf.fastq.gz; }
one() { bin=$2; shift 2; rm -f $W/rid_ab_reads.txt $W/rid_ab_counts.txt; echo "$LABEL $(COLORID_TIMING=1 $bin read_id -b $W/idx.bxi -q "$@" -n $W/rid_ab 2>&1 >/dev/null | tr '\r' '\n' | grep -E "timing: (device|classification)" | sed 's/timing: //; s/device front end: //; s/waiting for the file reader/reader/; s/push (H2D of the members)/push/; s/ until the first stretch was pushed/ to first push/' | tr '\n' '|' | cut -c1-260)"; }
for round in 1 2 3 4 5 6; do
  for which in old new; do
    [ $which = old ] && B=tools/ab_old/bin/colorid || B=colorid_amd/bin/colorid
    LABEL="1M  $which:" one x $B $W/reads.bgzf.fastq.gz
    LABEL="16M $which:" one x $B $W/reads16.bgzf.fastq.gz
    LABEL="4Mpe $which:" one x $B $W/reads4.bgzf.fastq.gz $W/reads4.bgzf.fastq.gz
  done
done
