# device front end: stretch size x the share of every stretch the reader's host threads inflate (after tools/e2e_demo.py left the files in /tmp/cid_e2e)
W=/tmp/cid_e2e
BIN=colorid_amd/bin/colorid
for cfg in "COLORID_DEVICE_FASTQ_HOST_SHARE=0" "COLORID_DEVICE_FASTQ_HOST_SHARE=0.5" "COLORID_DEVICE_FASTQ_HOST_SHARE=1" "COLORID_DEVICE_FASTQ_HOST_SHARE=0.5 COLORID_DEVICE_FASTQ_MB=128" "COLORID_DEVICE_FASTQ_HOST_SHARE=1 COLORID_DEVICE_FASTQ_MB=64" "COLORID_DEVICE_FASTQ_HOST_SHARE=1 COLORID_DEVICE_FASTQ_MB=128" "COLORID_DEVICE_FASTQ=0"; do
  for rep in 1 2; do echo "$cfg: $(env $cfg COLORID_TIMING=1 $BIN read_id -b $W/idx.bxi -q $W/reads4.bgzf.fastq.gz -n $W/rid_p 2>&1 >/dev/null | tr '\r' '\n' | grep -E "timing: (device|classification|total)" | sed 's/; of the GPU calls.*//; s/timing: //' | tr '\n' '|' | cut -c1-420)"; done
done
