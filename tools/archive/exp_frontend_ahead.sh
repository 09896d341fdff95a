# the device front end's knobs on 16 M single-end reads: look-ahead (stretches inflating ahead of the one being classified) x the host's share
# of the inflate work x the stretch size; run tools/e2e_demo.py (E2E_GENOMES=256 E2E_GROUPS=0) and tools/exp_frontend_16m.sh first
W=/tmp/cid_e2e
BIN=colorid_amd/bin/colorid
run() { cfg=$1; shift; for rep in 1 2 3; do echo "$cfg: $(env $cfg COLORID_TIMING=1 $BIN read_id -b $W/idx.bxi -q "$@" -n $W/rid_a 2>&1 >/dev/null | tr '\r' '\n' | grep -E "timing: (device|total)" | sed 's/; of the GPU calls.*//; s/timing: //; s/waits: parser on a full queue//; s/device front end: //; s/push (H2D of the members)/push/; s/waiting for the file reader/reader/' | tr '\n' '|' | cut -c1-230)"; done; }
for mb in ${EXP_MB:-128 256}; do for ahead in ${EXP_AHEAD:-1 2}; do for share in ${EXP_SHARE:-0 0.25 0.5}; do
  run "COLORID_DEVICE_FASTQ_MB=$mb COLORID_DEVICE_FASTQ_AHEAD=$ahead COLORID_DEVICE_FASTQ_HOST_SHARE=$share" $W/reads16.bgzf.fastq.gz
done; done; done
cmp $W/rid_a_reads.txt $W/rid_host16_reads.txt && echo "same rows (16 M single-end)"
