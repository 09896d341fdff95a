# Host-side sanitizer runs of the CLI on a GPU box (GPU ASan is not available on this pool): build tools/bin/colorid_asan and
# tools/bin/colorid_tsan with g++ -fsanitize=address,undefined / -fsanitize=thread from colorid_amd/csrc/host/*.cpp first
# (`make -C tools sanitizers`, in the build container) and take their two lines out of .gpurunignore for the call (60 MB that the other
# calls need not carry).
set -x
cd $GRAFT_REPO_ROOT
W=/tmp/san; mkdir -p $W
python3 - <<'PY'
import gzip, numpy as np, os
rng=np.random.default_rng(1); W="/tmp/san"
ac=np.frombuffer(b"ACGT",np.uint8)
with open(f"{W}/refs.tsv","w") as t:
    gs=[]
    for g in range(6):
        s=ac[rng.integers(0,4,200000)].tobytes(); gs.append(s)
        open(f"{W}/g{g}.fasta","wb").write(b">g%d\n"%g + b"\n".join(s[i:i+70] for i in range(0,len(s),70))+b"\n")
        t.write(f"genome{g}\t{W}/g{g}.fasta\n")
with open(f"{W}/refs130.tsv","w") as t:      # 130 accessions = 3 words of 64 colours: an index that can be striped over 2-3 ranks
    for g in range(130):
        s=gs[g%6][(g//6)*8000:(g//6)*8000+20000]
        open(f"{W}/h{g}.fasta","wb").write(b">h%d\n"%g + s + b"\n")
        t.write(f"acc{g:03d}\t{W}/h{g}.fasta\n")
def fq(path, n, mate):
    r=np.random.default_rng(5)
    with gzip.open(path,"wb",compresslevel=1) as f:
        for i in range(n):
            g=gs[r.integers(0,6)]; p=int(r.integers(0,len(g)-300)); L=int(r.integers(30,151))
            s=g[p:p+L] if mate==0 else g[p+100:p+100+L]
            f.write(b"@r%d/%d\n"%(i,mate+1)+s+b"\n+\n"+b"I"*len(s)+b"\n")
fq(f"{W}/r_1.fastq.gz",60000,0); fq(f"{W}/r_2.fastq.gz",60000,1)
# the same reads as block-gzip files (round 3: the device FASTQ front end of read_id / search / build takes these)
import struct, zlib
def bgzf(src, dst):
    text = gzip.open(src, "rb").read()
    with open(dst, "wb") as f:
        for i in list(range(0, len(text), 65280)) + [None]:
            c = b"" if i is None else text[i:i + 65280]
            co = zlib.compressobj(6, zlib.DEFLATED, -15); body = co.compress(c) + co.flush()
            f.write(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(body) + 8 - 1) + body +
                    struct.pack("<II", zlib.crc32(c) & 0xFFFFFFFF, len(c)))
bgzf(f"{W}/r_1.fastq.gz", f"{W}/b_1.fastq.gz"); bgzf(f"{W}/r_2.fastq.gz", f"{W}/b_2.fastq.gz")
PY
for B in tools/bin/colorid_asan "setarch x86_64 -R tools/bin/colorid_tsan"; do
  echo "=== $B"
  export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 TSAN_OPTIONS="report_signal_unsafe=0 second_deadlock_stack=1"
  $B build -s 2000000 -n 3 -k 27 -b $W/ix -r $W/refs.tsv > $W/build.out 2> $W/build.err; echo "build rc=$?"; tail -3 $W/build.err
  $B search -b $W/ix.bxi -q $W/r_1.fastq.gz -r $W/r_2.fastq.gz -g -f 0 > $W/s.out 2> $W/s.err; echo "search rc=$?"; grep -c . $W/s.out; grep -i "sanitizer\|ERROR\|WARNING: Thread" $W/s.err | head -5
  $B read_id -b $W/ix.bxi -q $W/r_1.fastq.gz $W/r_2.fastq.gz -n $W/rid -c 5000 > $W/r.out 2> $W/r.err; echo "read_id rc=$?"; wc -l $W/rid_reads.txt; grep -i "sanitizer\|ERROR\|WARNING: Thread" $W/r.err | head -5
  $B read_id -b $W/ix.bxi -q $W/r_1.fastq.gz -n $W/rid1 -c 777 > $W/r1.out 2> $W/r1.err; echo "read_id SE rc=$?"; grep -i "sanitizer\|ERROR\|WARNING: Thread" $W/r1.err | head -5
  # round 2: the default report (device-side mode), several ranks on the one GPU (cid_group: one host thread per rank), hashcheck
  $B search -b $W/ix.bxi -q $W/r_1.fastq.gz -f 0 -p 0.01 > $W/s2.out 2> $W/s2.err; echo "search default rc=$?"; grep -c . $W/s2.out; grep -i "sanitizer\|ERROR\|WARNING: Thread" $W/s2.err | head -5
  $B search -b $W/ix.bxi -q $W/r_1.fastq.gz -r $W/r_2.fastq.gz -f 0 -p 0.01 --devices 0,0,0 > $W/s3.out 2> $W/s3.err; echo "search 3 ranks rc=$?"; grep -c . $W/s3.out; grep -i "sanitizer\|ERROR\|WARNING: Thread" $W/s3.err | head -5
  $B read_id -b $W/ix.bxi -q $W/r_1.fastq.gz $W/r_2.fastq.gz -n $W/rid3 -c 5000 --devices 0,0 > $W/r3.out 2> $W/r3.err; echo "read_id 2 ranks rc=$?"; cmp $W/rid_reads.txt $W/rid3_reads.txt && echo "same rows as one rank"; grep -i "sanitizer\|ERROR\|WARNING: Thread" $W/r3.err | head -5
  # colour stripes over the ranks (cid_group_stripes_*): loader, searches, read_id and the per-read splice of the ranks' lists
  $B build -s 3000000 -n 3 -k 27 -b $W/ix130 -r $W/refs130.tsv > $W/b130.out 2> $W/b130.err; echo "build 130 rc=$?"
  $B search -b $W/ix130.bxi -q $W/r_1.fastq.gz -r $W/r_2.fastq.gz -f 0 -p 0.01 > $W/s4a.out 2> $W/s4a.err; echo "search 130 one GPU rc=$?"
  $B search -b $W/ix130.bxi -q $W/r_1.fastq.gz -r $W/r_2.fastq.gz -f 0 -p 0.01 --devices 0,0,0 --placement striped > $W/s4.out 2> $W/s4.err; echo "search striped 3 ranks rc=$?"; grep -c . $W/s4.out; sort $W/s4a.out > $W/s4a.sorted; sort $W/s4.out | cmp - $W/s4a.sorted && echo "same rows as one GPU"; grep -i "sanitizer\|ERROR\|WARNING: Thread" $W/s4.err | head -5
  $B read_id -b $W/ix130.bxi -q $W/r_1.fastq.gz $W/r_2.fastq.gz -n $W/rid4a -c 5000 > $W/r4a.out 2> $W/r4a.err; echo "read_id 130 one GPU rc=$?"
  $B read_id -b $W/ix130.bxi -q $W/r_1.fastq.gz $W/r_2.fastq.gz -n $W/rid4 -c 5000 --devices 0,0 --placement striped > $W/r4.out 2> $W/r4.err; echo "read_id striped 2 ranks rc=$?"; cmp $W/rid4a_reads.txt $W/rid4_reads.txt && echo "same rows as one GPU"; grep -i "sanitizer\|ERROR\|WARNING: Thread" $W/r4.err | head -5
  # round 3: block-gzip input through the device front end (reader thread + its inflating pool, pinned buffer pool, pushes one stretch ahead, the
  # poll threads behind it), small stretches so that several are in flight; search's and build's k-mer maps through the same reader
  export COLORID_DEVICE_FASTQ_MB=1
  $B read_id -b $W/ix.bxi -q $W/b_1.fastq.gz -n $W/ridb1 -c 777 > $W/rb1.out 2> $W/rb1.err; echo "read_id SE block gzip rc=$?"; cmp $W/rid1_reads.txt $W/ridb1_reads.txt && echo "same rows as from the gzip stream"; grep -i "sanitizer\|ERROR\|WARNING: Thread" $W/rb1.err | head -5
  COLORID_DEVICE_FASTQ_AHEAD=3 $B read_id -b $W/ix.bxi -q $W/b_1.fastq.gz $W/b_2.fastq.gz -n $W/ridb -c 5000 > $W/rb.out 2> $W/rb.err; echo "read_id PE block gzip rc=$?"; cmp $W/rid_reads.txt $W/ridb_reads.txt && echo "same rows as from the gzip stream"; grep -i "sanitizer\|ERROR\|WARNING: Thread" $W/rb.err | head -5
  # round 6: the stretches in page-locked memory, their copies beside the loop (CID_FASTQ_KEEP on cid_fastq_push_bgzf), the inflate behind the classifier
  COLORID_DEVICE_FASTQ_PINNED=1 COLORID_DEVICE_FASTQ_HOST_SHARE=0 CID_FASTQ_INFLATE_BESIDE=0 $B read_id -b $W/ix.bxi -q $W/b_1.fastq.gz $W/b_2.fastq.gz -n $W/ridp -c 5000 > $W/rp.out 2> $W/rp.err; echo "read_id PE block gzip, page-locked rc=$?"; cmp $W/rid_reads.txt $W/ridp_reads.txt && echo "same rows as from the gzip stream"; grep -i "sanitizer\|ERROR\|WARNING: Thread" $W/rp.err | head -5
  $B search -b $W/ix.bxi -q $W/b_1.fastq.gz -r $W/b_2.fastq.gz -f 0 -p 0.01 > $W/sb.out 2> $W/sb.err; echo "search block gzip rc=$?"; cut -f2- $W/sb.out | sort > $W/sb.sorted; $B search -b $W/ix.bxi -q $W/r_1.fastq.gz -r $W/r_2.fastq.gz -f 0 -p 0.01 2> /dev/null | cut -f2- | sort | cmp - $W/sb.sorted && echo "same report as from the gzip stream"; grep -i "sanitizer\|ERROR\|WARNING: Thread" $W/sb.err | head -5
  printf "reads\t$W/b_1.fastq.gz\t$W/b_2.fastq.gz\ngenome0\t$W/g0.fasta\n" > $W/refs_fq.tsv
  $B build -s 2000000 -n 3 -k 27 -b $W/ixfq -r $W/refs_fq.tsv > $W/bfq.out 2> $W/bfq.err; echo "build from block gzip rc=$?"; grep -i "sanitizer\|ERROR\|WARNING: Thread" $W/bfq.err | head -5
  # batch_id: a sheet of three samples (gzip stream, block-gzip pair through the device front end with the next sample read ahead, FASTA), one index load
  printf "$W/s_gz\t$W/r_1.fastq.gz\n$W/s_bgzf\t$W/b_1.fastq.gz\t$W/b_2.fastq.gz\n$W/s_fa\t$W/g0.fasta\n" > $W/sheet.tsv   # (a sample's name is the prefix of its files)
  $B batch_id -b $W/ix.bxi -q $W/sheet.tsv -T san -c 5000 > $W/bid.out 2> $W/bid.err; echo "batch_id rc=$?"; cmp $W/rid_reads.txt $W/s_bgzf_san_reads.txt && echo "batch_id: the pair's rows as read_id wrote them"; grep -i "sanitizer\|ERROR\|WARNING: Thread" $W/bid.err | head -5
  # round 4: the device front end giving way mid-run (classifier drained, output emptied, the whole input again through the host front end),
  # the orderly teardown (the default leaves with _exit once the outputs are closed), the index uploaded from a mapping (ix130: 132 MB)
  CID_FASTQ_REFUSE_AT_STEP=2 $B read_id -b $W/ix.bxi -q $W/b_1.fastq.gz $W/b_2.fastq.gz -n $W/ridf -c 5000 > $W/rf.out 2> $W/rf.err; echo "read_id PE restart rc=$?"; cmp $W/rid_reads.txt $W/ridf_reads.txt && echo "restart: same rows"; grep -c "starting over" $W/rf.err; grep -i "sanitizer\|ERROR\|WARNING: Thread" $W/rf.err | head -5
  COLORID_FULL_TEARDOWN=1 $B read_id -b $W/ix130.bxi -q $W/b_1.fastq.gz -n $W/ridt -c 5000 > $W/rt.out 2> $W/rt.err; echo "read_id orderly teardown rc=$?"; grep -i "sanitizer\|ERROR\|WARNING: Thread" $W/rt.err | head -5
  COLORID_FULL_TEARDOWN=1 COLORID_INDEX_MMAP=0 $B read_id -b $W/ix130.bxi -q $W/b_1.fastq.gz -n $W/ridu -c 5000 > $W/ru.out 2> $W/ru.err; echo "read_id buffered index rc=$?"; cmp $W/ridt_reads.txt $W/ridu_reads.txt && echo "mapped index: same rows as buffered"
  unset COLORID_DEVICE_FASTQ_MB
  $B hashcheck -b $W/ix.bxi -r $W/refs.tsv > $W/h.out 2> $W/h.err; echo "hashcheck rc=$?"; grep verdict $W/h.out; grep -i "sanitizer\|ERROR\|WARNING: Thread" $W/h.err | head -5
done
# TSan: the HIP/HSA runtime is not instrumented and reports races between its own threads (objects it allocates inside an API call
# on the caller's thread and frees on its own).  What matters: is the racing access itself — the innermost frame outside the
# sanitizer runtime — in THIS repository's code (colorid_amd/csrc/host/*.cpp, libcolorid_hip.so)?
python3 - <<'PY'
import glob, re
tot = ours = 0
for f in sorted(glob.glob("/tmp/san/*.err")):
    blocks = open(f, errors="replace").read().split("WARNING: ThreadSanitizer")[1:]
    for b in blocks:
        tot += 1
        head = re.split(r"\n\s*(?:Location is|Thread T\d+ |Mutex M\d+)", b)[0]
        sections = re.split(r"\n\s*\n", head)          # "Write of size ..." / "Previous write ..." stacks
        mine = False
        for sec in sections:
            frames = [l for l in sec.splitlines() if re.match(r"\s*#\d+ ", l)]
            inner = next((l for l in frames if "libtsan" not in l), None)
            if inner and re.search(r"csrc/host/|libcolorid_hip\.so|colorid_tsan", inner):
                mine = True
        if mine:
            ours += 1
            print("RACING ACCESS IN THIS REPOSITORY'S CODE (", f, "):\n", head[:1500])
print(f"ThreadSanitizer reports in the TSan binary's runs: {tot}; with the racing access itself in this repository's code: {ours}")
PY
