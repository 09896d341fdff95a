# The device front end with and without a share of every stretch inflated on the host (COLORID_DEVICE_FASTQ_HOST_SHARE), alternating runs:
# 16 M reads single-end, 4 M pairs, 1 M reads.  Needs tools/e2e_demo.py's files (E2E_GENOMES=256 E2E_GROUPS=0).  On the GPU box:
#   E2E_GENOMES=256 E2E_GROUPS=0 python3 tools/e2e_demo.py > /dev/null && bash tools/exp_frontend_share.sh > gpurun_out/r06_frontend_share.txt
W=/tmp/cid_e2e
BIN=colorid_amd/bin/colorid
[ -f $W/reads4.bgzf.fastq.gz ] || cat $W/reads.bgzf.fastq.gz $W/reads.bgzf.fastq.gz $W/reads.bgzf.fastq.gz $W/reads.bgzf.fastq.gz > $W/reads4.bgzf.fastq.gz
[ -f $W/reads16.bgzf.fastq.gz ] || cat $W/reads4.bgzf.fastq.gz $W/reads4.bgzf.fastq.gz $W/reads4.bgzf.fastq.gz $W/reads4.bgzf.fastq.gz > $W/reads16.bgzf.fastq.gz
run() { cfg=$1; shift; echo "$cfg [$#]: $(env $cfg COLORID_TIMING=1 $BIN read_id -b $W/idx.bxi -q "$@" -n $W/rid_s$# 2>&1 >/dev/null | tr '\r' '\n' | grep -E "timing: (device|classification|total)" | sed 's/; of the GPU calls.*//; s/timing: //; s/waits: parser on a full queue//' | tr '\n' '|' | cut -c1-600)"; }
for rep in 1 2 3; do for sh in ${SHARES:-0 0.3}; do
  run "COLORID_DEVICE_FASTQ_HOST_SHARE=$sh" $W/reads16.bgzf.fastq.gz
  run "COLORID_DEVICE_FASTQ_HOST_SHARE=$sh" $W/reads4.bgzf.fastq.gz $W/reads4.bgzf.fastq.gz
  run "COLORID_DEVICE_FASTQ_HOST_SHARE=$sh" $W/reads.bgzf.fastq.gz
done; done
