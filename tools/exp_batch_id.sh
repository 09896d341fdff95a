# `colorid batch_id` (one GPU context + one index upload for a whole sample sheet) against one `read_id` per sample: 8 samples of
# 1 M reads (single-stream gzip and block-gzip alternate), the 256-genome demo index.  After tools/e2e_demo.py (E2E_GENOMES=256 E2E_GROUPS=0).
W=/tmp/cid_e2e
BIN=$PWD/colorid_amd/bin/colorid
python3 - <<'PY'
import gzip, struct, zlib
W="/tmp/cid_e2e"
text=gzip.open(f"{W}/reads.fastq.gz","rb").read()
def bgzf(path, data, block=65280):
    with open(path,"wb") as f:
        for i in range(0,len(data),block):
            chunk=data[i:i+block]
            c=zlib.compressobj(1,zlib.DEFLATED,-15); body=c.compress(chunk)+c.flush()
            f.write(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0"+struct.pack("<H",len(body)+25)+body+struct.pack("<II",zlib.crc32(chunk),len(chunk)))
        f.write(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
bgzf(f"{W}/reads.bgzf.fastq.gz", text)
with open(f"{W}/samples.tsv","w") as t:
    for i in range(8):
        t.write(f"sample{i}\t{W}/" + ("reads.fastq.gz" if i % 2 == 0 else "reads.bgzf.fastq.gz") + "\n")
PY
cd $W
now() { date +%s.%N; }
for rep in 1 2 3; do
  t0=$(now)
  $BIN batch_id -b $W/idx.bxi -q $W/samples.tsv -T batch >/dev/null 2>$W/batch.err
  t1=$(now)
  for i in 0 1 2 3 4 5 6 7; do
    f=$(awk -v n=sample$i '$1==n{print $2}' $W/samples.tsv)
    $BIN read_id -b $W/idx.bxi -q $f -n $W/single$i >/dev/null 2>>$W/single.err
  done
  t2=$(now)
  python3 -c "print('rep $rep: batch_id of 8 samples %.2f s; 8 x read_id %.2f s' % ($t1-$t0, $t2-$t1))"
done
for i in 0 1 2 3 4 5 6 7; do cmp $W/sample${i}_batch_reads.txt $W/single${i}_reads.txt && cmp $W/sample${i}_batch_counts.txt $W/single${i}_counts.txt || echo "sample $i DIFFERS"; done
echo "rows compared"
COLORID_TIMING=1 $BIN batch_id -b $W/idx.bxi -q $W/samples.tsv -T batch 2>&1 >/dev/null | tr '\r' '\n' | grep -E "timing: (GPU context|index load|classification|counts file|release|total)" | cut -c1-200
