# The device FASTQ front end on a SMALL block-gzip sample (1 M reads, 310 MB of text): stretch size against the classification phase.
# With one 256 MiB stretch nothing overlaps (inflate + classify, then poll + write).  After tools/e2e_demo.py and tools/exp_batch_id.sh.
W=/tmp/cid_e2e
BIN=$PWD/colorid_amd/bin/colorid
for mb in 256 128 64 32 16; do
  for rep in 1 2 3; do
    echo "MB=$mb: $(COLORID_DEVICE_FASTQ_MB=$mb COLORID_TIMING=1 $BIN read_id -b $W/idx.bxi -q $W/reads.bgzf.fastq.gz -n $W/small_$mb 2>&1 >/dev/null | tr '\r' '\n' | grep -E "timing: (classification|total|device front end|index load)" | sed 's/; waits.*//; s/timing: //' | tr '\n' '|' | cut -c1-420)"
  done
  cmp $W/small_${mb}_reads.txt $W/small_256_reads.txt || echo "MB=$mb DIFFERS"
done
