# `colorid search` with a block-gzip fastq query of 16 M (and 1 M) reads: the k-mer map counted through the device front end (default) against the
# host front end (COLORID_DEVICE_FASTQ=0); after tools/e2e_demo.py (E2E_GENOMES=256 E2E_GROUPS=0) + tools/exp_frontend_16m.sh
W=/tmp/cid_e2e
BIN=colorid_amd/bin/colorid
run() { cfg=$1; shift; for rep in 1 2 3; do env $cfg COLORID_TIMING=1 $BIN search -b $W/idx.bxi -q $W/$1 -f 0 -p 0.005 > $W/search_$2.txt 2> $W/search_$2.err; echo "$cfg [$1]: $(tr '\r' '\n' < $W/search_$2.err | grep -E "timing: (search|query|index|device)|k-mers in query" | sed 's/timing: //' | tr '\n' '|' | cut -c1-300)"; done; }
run "COLORID_DEVICE_FASTQ=0" reads16.bgzf.fastq.gz host
run "A=default" reads16.bgzf.fastq.gz dev
cmp <(sort $W/search_host.txt) <(sort $W/search_dev.txt) && echo "same report (16 M reads)"
run "COLORID_DEVICE_FASTQ=0" reads.bgzf.fastq.gz host1
run "A=default" reads.bgzf.fastq.gz dev1
cmp <(sort $W/search_host1.txt) <(sort $W/search_dev1.txt) && echo "same report (1 M reads)"
