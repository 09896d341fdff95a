# where a 16 M-read device-front-end run spends its wall time (full COLORID_TIMING lines + the library's own parts, CID_FASTQ_TIMING);
# after tools/e2e_demo.py + tools/exp_batch_id.sh (reads.bgzf.fastq.gz).  The output files are removed first: truncating the 600 MB
# file of the run before costs 50 ms inside fopen.
W=/tmp/cid_e2e
BIN=colorid_amd/bin/colorid
[ -f $W/reads16.bgzf.fastq.gz ] || { cat $W/reads.bgzf.fastq.gz $W/reads.bgzf.fastq.gz $W/reads.bgzf.fastq.gz $W/reads.bgzf.fastq.gz > $W/reads4.bgzf.fastq.gz; cat $W/reads4.bgzf.fastq.gz $W/reads4.bgzf.fastq.gz $W/reads4.bgzf.fastq.gz $W/reads4.bgzf.fastq.gz > $W/reads16.bgzf.fastq.gz; }
for cfg in ${CFGS:-A=default}; do
for rep in 1 2 3; do rm -f $W/rid_d_reads.txt $W/rid_d_counts.txt; echo "$cfg"; T0=$(grep -E "nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat | tr "\n" " "); env $cfg CID_FASTQ_TIMING=1 COLORID_TIMING=1 $BIN read_id -b $W/idx.bxi -q $W/reads16.bgzf.fastq.gz -n $W/rid_d 2>&1 >/dev/null | tr '\r' '\n' | grep -E "timing:|cid_fastq:" | grep -v "counts file\|release\|set-up" | cut -c1-420; echo "cpu.stat before: $T0 after: $(grep -E "nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat | tr "\n" " ")"; echo; done
done
