# `colorid read_id` / `search` on single-stream gzip (the commonest fastq.gz): the reader's own DEFLATE decoder (fast_inflate.hpp, default)
# against zlib's raw inflate (COLORID_FAST_INFLATE=0); 4 M reads, gzip levels 1 and 6.  After tools/e2e_demo.py (E2E_GENOMES=256 E2E_GROUPS=0).
W=/tmp/cid_e2e
BIN=colorid_amd/bin/colorid
python3 - <<'PY'
import gzip
W="/tmp/cid_e2e"
text=gzip.open(f"{W}/reads.fastq.gz","rb").read()
for lvl in (1,6):
    with gzip.open(f"{W}/reads4.l{lvl}.fastq.gz","wb",compresslevel=lvl) as f:
        for _ in range(4): f.write(text)
PY
ls -la $W/reads4.l1.fastq.gz $W/reads4.l6.fastq.gz
run() { cfg=$1; shift; for rep in 1 2 3; do echo "$cfg [$(basename $1)]: $(env $cfg COLORID_TIMING=1 $BIN read_id -b $W/idx.bxi -q "$@" -n $W/rid_g 2>&1 >/dev/null | tr '\r' '\n' | grep -E "timing: (classification|total|the input|gzip member)" | sed 's/; of the GPU calls.*//; s/timing: //' | tr '\n' '|' | cut -c1-260)"; done; }
for lvl in 1 6; do
  run "COLORID_FAST_INFLATE=0" $W/reads4.l$lvl.fastq.gz
  cp $W/rid_g_reads.txt $W/rid_gz_zlib.txt
  run "COLORID_PAR_GZIP=0" $W/reads4.l$lvl.fastq.gz
  run "A=default" $W/reads4.l$lvl.fastq.gz
  cmp $W/rid_g_reads.txt $W/rid_gz_zlib.txt && echo "same rows (level $lvl)"
  run "COLORID_GZ_THREADS=8" $W/reads4.l$lvl.fastq.gz
done
for rep in 1 2; do for cfg in COLORID_FAST_INFLATE=0 COLORID_PAR_GZIP=0 A=default; do echo "search $cfg: $(env $cfg COLORID_TIMING=1 $BIN search -b $W/idx.bxi -q $W/reads4.l6.fastq.gz -f 0 -p 0.005 2>&1 >/dev/null | grep -E "timing: search" | tr '\n' ' ')"; done; done
