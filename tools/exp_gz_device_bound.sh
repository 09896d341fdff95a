# A bound for "single-stream gzip decoded on the device" (round-3 verdict, stretch item) that needs no new kernel: the same 4 M reads with
# forty quality letters (a) as ONE gzip stream, decoded on the host's threads as the CLI does today, and (b) cut into block-gzip members
# and inflated by the device front end — what a device decoder of the single stream could reach AT BEST: it has the same DEFLATE chains
# to walk, plus block starts to guess, 16-bit marker symbols to write and resolve, and every chunk's first block decoded twice.
# After tools/e2e_demo.py (E2E_GENOMES=256 E2E_GROUPS=0).
W=/tmp/cid_e2e
BIN=colorid_amd/bin/colorid
python3 - <<'PY'
import gzip, struct, zlib, numpy as np
W="/tmp/cid_e2e"
rng=np.random.default_rng(4)
lines=gzip.open(f"{W}/reads.fastq.gz","rb").read().split(b"\n")
n=len(lines)//4
base=np.linspace(72,52,150)[None,:]
q=np.clip(base+rng.normal(0,4,(n,150)),35,74).astype(np.uint8)
parts=[]
for rep in range(4):
    parts.append(b"".join(lines[4*i]+(b".%d"%rep)+b"\n"+lines[4*i+1]+b"\n+\n"+q[i].tobytes()+b"\n" for i in range(n)))
text=b"".join(parts)
with gzip.open(f"{W}/reads4.forty.fastq.gz","wb",compresslevel=6) as f:
    f.write(text)
with open(f"{W}/reads4.forty.bgzf.fastq.gz","wb") as f:
    for i in range(0,len(text),65280):
        c=text[i:i+65280]
        co=zlib.compressobj(6,zlib.DEFLATED,-15); body=co.compress(c)+co.flush()
        f.write(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0"+struct.pack("<H",len(body)+25)+body+struct.pack("<II",zlib.crc32(c),len(c)))
    f.write(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
PY
ls -la $W/reads4.forty.fastq.gz $W/reads4.forty.bgzf.fastq.gz
run() { cfg=$1; shift; for rep in 1 2 3; do echo "$cfg [$(basename $1)]: $(env $cfg COLORID_TIMING=1 $BIN read_id -b $W/idx.bxi -q "$@" -n $W/rid_q 2>&1 >/dev/null | tr '\r' '\n' | grep -E "timing: classification" | sed 's/timing: //' | tr '\n' '|' | cut -c1-120)"; done; }
run "A=host_threads_default" $W/reads4.forty.fastq.gz
cp $W/rid_q_reads.txt $W/rid_q_gz.txt
run "COLORID_DEVICE_FASTQ_HOST_SHARE=0" $W/reads4.forty.bgzf.fastq.gz
cmp $W/rid_q_reads.txt $W/rid_q_gz.txt && echo "same rows"
run "A=device_front_end_default_share" $W/reads4.forty.bgzf.fastq.gz
run "COLORID_DEVICE_FASTQ=0" $W/reads4.forty.bgzf.fastq.gz
nproc
