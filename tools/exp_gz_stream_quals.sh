# single-stream gzip with quality strings that do not compress to nothing (tools/e2e_demo.py writes 'I' everywhere: such a stream decodes at
# 3 GB/s and what follows the decoder is the limit).  The same 4 M reads with (a) four binned quality letters, 90 / 6 / 3 / 1 % (current
# Illumina instruments), (b) forty letters, falling along the read (older instruments): serial decoder against the chunked one.
# After tools/e2e_demo.py (E2E_GENOMES=256 E2E_GROUPS=0).
W=/tmp/cid_e2e
BIN=colorid_amd/bin/colorid
python3 - <<'PY'
import gzip, numpy as np
W="/tmp/cid_e2e"
rng=np.random.default_rng(4)
lines=gzip.open(f"{W}/reads.fastq.gz","rb").read().split(b"\n")
n=len(lines)//4
for tag in ("binned","forty"):
    if tag=="binned":
        q=rng.choice(np.frombuffer(b"F:,#",np.uint8),(n,150),p=[.90,.06,.03,.01])
    else:
        base=np.linspace(72,52,150)[None,:]
        q=np.clip(base+rng.normal(0,4,(n,150)),35,74).astype(np.uint8)
    with gzip.open(f"{W}/reads4.{tag}.fastq.gz","wb",compresslevel=6) as f:
        for rep in range(4):
            out=[]
            for i in range(n):
                out.append(lines[4*i]+(b".%d"%rep)+b"\n"+lines[4*i+1]+b"\n+\n"+q[i].tobytes()+b"\n")
            f.write(b"".join(out))
PY
ls -la $W/reads4.binned.fastq.gz $W/reads4.forty.fastq.gz
run() { cfg=$1; shift; for rep in 1 2 3; do echo "$cfg [$(basename $1)]: $(env $cfg COLORID_TIMING=1 $BIN read_id -b $W/idx.bxi -q "$@" -n $W/rid_q 2>&1 >/dev/null | tr '\r' '\n' | grep -E "timing: (classification|gzip member)" | sed 's/timing: //; s/gzip member of [0-9]* bytes of text decoded on //; s/ started inside the stream and were taken,/ taken,/; s/ stretches decoded again serially/ serial/' | tr '\n' '|' | cut -c1-200)"; done; }
for tag in binned forty; do
  [ -n "$QUICK" ] || run "COLORID_FAST_INFLATE=0" $W/reads4.$tag.fastq.gz
  run "COLORID_PAR_GZIP=0" $W/reads4.$tag.fastq.gz
  cp $W/rid_q_reads.txt $W/rid_q_zlib.txt
  run "A=default" $W/reads4.$tag.fastq.gz
  cmp $W/rid_q_reads.txt $W/rid_q_zlib.txt && echo "same rows as from the serial decoder ($tag)"
  run "COLORID_GZ_THREADS=8" $W/reads4.$tag.fastq.gz
done
