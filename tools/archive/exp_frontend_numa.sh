# Run-to-run spread of the 16 M-read front end (0.5 s, sometimes 0.7-1.0 s with the polling threads twice as slow): is it where the
# threads run?  Interleaved rounds (what the neighbours on the host do then hits every variant alike): free, and bound (taskset) to the
# CPUs of either NUMA node.  After tools/e2e_demo.py + tools/exp_batch_id.sh.
W=/tmp/cid_e2e
BIN=colorid_amd/bin/colorid
[ -f $W/reads16.bgzf.fastq.gz ] || { cat $W/reads.bgzf.fastq.gz $W/reads.bgzf.fastq.gz $W/reads.bgzf.fastq.gz $W/reads.bgzf.fastq.gz > $W/reads4.bgzf.fastq.gz; cat $W/reads4.bgzf.fastq.gz $W/reads4.bgzf.fastq.gz $W/reads4.bgzf.fastq.gz $W/reads4.bgzf.fastq.gz > $W/reads16.bgzf.fastq.gz; }
for n in /sys/devices/system/node/node*; do echo "$(basename $n): $(cat $n/cpulist)"; done
one() { rm -f $W/rid_n_reads.txt $W/rid_n_counts.txt; echo "$1: $(COLORID_TIMING=1 $2 $BIN read_id -b $W/idx.bxi -q $W/reads16.bgzf.fastq.gz -n $W/rid_n 2>&1 >/dev/null | tr '\r' '\n' | grep -E "timing: (total|classification)" | sed 's/; of the GPU calls.*//; s/timing: //; s/GPU calls (copies + kernels)/GPU/; s/waits: parser on a full queue [0-9]* ms, //; s/GPU stage idle [0-9]* ms, //' | tr '\n' '|' | cut -c1-200)"; }
for round in ${ROUNDS:-1 2 3 4 5 6 7 8}; do
  one "free " ""
  one "node0" "taskset -c $(cat /sys/devices/system/node/node0/cpulist)"
  one "node1" "taskset -c $(cat /sys/devices/system/node/node1/cpulist)"
done
