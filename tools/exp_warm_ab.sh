# What the helper thread beside the index load should load ahead (colorid read_id on 150 Mbases of 10 kb reads, block gzip): the default mask
# (READID | INFLATE | FASTQ = 13), + CID_WARM_PIPES (45), + CID_WARM_COLD (29), both (61), nothing (0).  Run tools/exp_cli_long.py first (it makes /tmp/cid_long).
W=/tmp/cid_long
BIN=colorid_amd/bin/colorid
run() { cfg="$1"; for rep in 1 2 3; do env $cfg COLORID_TIMING=1 $BIN read_id -b $W/idx.bxi -q $W/10kb.fastq.gz -n $W/o -Q 0 2>&1 >/dev/null | tr '\r' '\n' | grep -E "timing: (index load|classification|GPU context)" | sed 's/timing: //' | tr '\n' '|' ; echo " <- $cfg"; done; }
run "A=default"
run "COLORID_WARM=45"
run "COLORID_WARM=29"
run "COLORID_WARM=61"
run "COLORID_WARM=0"
