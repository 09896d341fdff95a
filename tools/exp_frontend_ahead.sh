# the device front end's look-ahead (stretches pushed, i.e. inflating, ahead of the one being classified) x the host's share of the
# inflate work, 16 M single-end reads; run tools/e2e_demo.py (E2E_GENOMES=256 E2E_GROUPS=0) and tools/exp_frontend_16m.sh first
W=/tmp/cid_e2e
BIN=colorid_amd/bin/colorid
run() { cfg=$1; shift; for rep in 1 2 3; do echo "$cfg: $(env $cfg COLORID_TIMING=1 $BIN read_id -b $W/idx.bxi -q "$@" -n $W/rid_a 2>&1 >/dev/null | tr '\r' '\n' | grep -E "timing: (device|classification|total)" | sed 's/; of the GPU calls.*//; s/timing: //; s/waits: parser on a full queue//' | tr '\n' '|' | cut -c1-330)"; done; }
for ahead in 1 2 3; do for share in 0.5 0.375 0.25; do
  run "COLORID_DEVICE_FASTQ_AHEAD=$ahead COLORID_DEVICE_FASTQ_HOST_SHARE=$share" $W/reads16.bgzf.fastq.gz
done; done
cmp $W/rid_a_reads.txt $W/rid_host16_reads.txt && echo "same rows (16 M single-end)"
