# kernel timeline of a 16 M-read `read_id` through the device front end (rocprofv3 --kernel-trace, csv) -> tools/timeline_gaps.py
# after tools/e2e_demo.py (E2E_GENOMES=256 E2E_GROUPS=0), tools/exp_batch_id.sh (writes reads.bgzf.fastq.gz)
W=/tmp/cid_e2e
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/frontend_timeline; rm -rf $O; mkdir -p $O
cat $W/reads.bgzf.fastq.gz $W/reads.bgzf.fastq.gz $W/reads.bgzf.fastq.gz $W/reads.bgzf.fastq.gz > $W/reads4.bgzf.fastq.gz
cat $W/reads4.bgzf.fastq.gz $W/reads4.bgzf.fastq.gz $W/reads4.bgzf.fastq.gz $W/reads4.bgzf.fastq.gz > $W/reads16.bgzf.fastq.gz
COLORID_TIMING=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- colorid_amd/bin/colorid read_id -b $W/idx.bxi -q $W/reads16.bgzf.fastq.gz -n $W/rid_tl > $O/run.log 2>&1
f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python3 tools/timeline_gaps.py $f > $O/timeline.txt 2>&1
tr '\r' '\n' < $O/run.log | grep "timing:" | cut -c1-300 >> $O/timeline.txt
rm -rf $O/trace
cat $O/timeline.txt
