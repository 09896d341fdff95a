# what the host front end waits for on a single-stream fastq.gz with real quality strings (4 M reads, forty letters): the CLI's own timing
# lines.  After tools/exp_gz_device_bound.sh's data (it writes /tmp/cid_e2e/reads4.forty.fastq.gz).
W=/tmp/cid_e2e
BIN=colorid_amd/bin/colorid
for cfg in "A=default" "COLORID_GZ_THREADS=12" "COLORID_PARSE_THREADS=6" "COLORID_POLL_THREADS=4"; do
  for rep in 1 2; do
    echo "== $cfg"
    env $cfg COLORID_TIMING=1 $BIN read_id -b $W/idx.bxi -q $W/reads4.forty.fastq.gz -n $W/rid_q 2>&1 >/dev/null | tr '\r' '\n' | grep "timing:" | cut -c1-330
  done
done
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null
