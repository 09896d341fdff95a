# where a 16 M-read device-front-end run spends its wall time (full COLORID_TIMING lines); after tools/e2e_demo.py + tools/exp_frontend_16m.sh
W=/tmp/cid_e2e
BIN=colorid_amd/bin/colorid
for rep in 1 2 3; do COLORID_TIMING=1 $BIN read_id -b $W/idx.bxi -q $W/reads16.bgzf.fastq.gz -n $W/rid_d 2>&1 >/dev/null | tr '\r' '\n' | grep "timing:" | cut -c1-400; echo; done
