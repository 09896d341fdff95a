# read pairs from two gzip streams (4 M pairs; one quality letter, and forty): both files are decoded at once — serial decoders, the chunked
# decoder sharing half the CPU share between the mates (default), and eight threads each.  After tools/e2e_demo.py + tools/exp_gz_stream.sh +
# tools/exp_gz_stream_quals.sh (they write the files).
W=/tmp/cid_e2e
BIN=colorid_amd/bin/colorid
run() { cfg=$1; shift; for rep in 1 2 3; do echo "$cfg [$(basename $1)]: $(env $cfg COLORID_TIMING=1 $BIN read_id -b $W/idx.bxi -q "$@" -n $W/rid_gp 2>&1 >/dev/null | tr '\r' '\n' | grep -E "timing: (classification|gzip member)" | sed 's/timing: //; s/gzip member of [0-9]* bytes of text decoded on //; s/ threads:.*/ threads/' | tr '\n' '|' | cut -c1-160)"; done; }
for f in reads4.l6.fastq.gz reads4.forty.fastq.gz; do
  run "COLORID_PAR_GZIP=0" $W/$f $W/$f
  cp $W/rid_gp_reads.txt $W/rid_gp_serial.txt
  run "A=default" $W/$f $W/$f
  cmp $W/rid_gp_reads.txt $W/rid_gp_serial.txt && echo "same rows ($f)"
  run "COLORID_GZ_THREADS=8" $W/$f $W/$f
done
