#!/bin/bash
# classification-phase time of `colorid read_id` (COLORID_TIMING=1) for host thread settings, 3 runs each, on the files
# tools/e2e_demo.py leaves in /tmp/cid_e2e (run that first)
W=/tmp/cid_e2e
BIN=colorid_amd/bin/colorid
f=${1:-reads.bgzf.fastq.gz}
for gz in 8 4; do for parse in 1 2 4; do for poll in 2 4 8; do
  line="gz=$gz parse=$parse poll=$poll :"
  for rep in 1 2 3; do
    t=$(env COLORID_TIMING=1 COLORID_GZ_THREADS=$gz COLORID_PARSE_THREADS=$parse COLORID_POLL_THREADS=$poll $BIN read_id -b $W/idx.bxi -q $W/$f -n $W/rid_x 2>&1 >/dev/null | grep -o "timing: total [0-9]* ms" | grep -o "[0-9]*")
    line="$line $t"
  done
  echo "$line"
done; done; done
