# kernels and copies of `colorid read_id` on 16 M reads of block gzip (device front end): who occupies the GPU while the loop runs.
# Needs tools/e2e_demo.py's files.  On the GPU box: bash tools/prof_cli_16m.sh [tag]
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
W=/tmp/cid_e2e; TAG=${1:-cli16}; O=gpurun_out/prof_$TAG; rm -rf $O
[ -f $W/reads4.bgzf.fastq.gz ] || cat $W/reads.bgzf.fastq.gz $W/reads.bgzf.fastq.gz $W/reads.bgzf.fastq.gz $W/reads.bgzf.fastq.gz > $W/reads4.bgzf.fastq.gz
[ -f $W/reads16.bgzf.fastq.gz ] || cat $W/reads4.bgzf.fastq.gz $W/reads4.bgzf.fastq.gz $W/reads4.bgzf.fastq.gz $W/reads4.bgzf.fastq.gz > $W/reads16.bgzf.fastq.gz
COLORID_TIMING=1 CID_FASTQ_TIMING=1 timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $O -- colorid_amd/bin/colorid read_id -b $W/idx.bxi -q $W/reads16.bgzf.fastq.gz -n $W/rid_prof > $O.out 2> $O.err
tr '\r' '\n' < $O.err | grep -E "timing: (device|total)|cid_fastq:" | cut -c1-700
python3 - $O <<'PY'
import csv, glob, sys
O = sys.argv[1]
f = glob.glob(O + "/**/*kernel_stats.csv", recursive=True)[0]
tot = 0
for r in list(csv.DictReader(open(f)))[:14]:
    print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>5s}  total {float(r["TotalDurationNs"]) / 1e6:8.1f} ms  avg {float(r["AverageNs"]) / 1e3:9.1f} us')
for r in csv.DictReader(open(f)): tot += float(r["TotalDurationNs"])
print(f"all kernels: {tot / 1e6:.1f} ms")
m = glob.glob(O + "/**/*memory_copy_stats.csv", recursive=True)
if m:
    for r in csv.DictReader(open(m[0])): print("copies", r["Name"], r["Calls"], f'{float(r["TotalDurationNs"]) / 1e6:.1f} ms')
# busy spans of the device: union of kernel intervals after the index load
k = glob.glob(O + "/**/*kernel_trace.csv", recursive=True)[0]
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(k)))
t_first = iv[0][0]; busy = 0; cur_s, cur_e = iv[0]
for s, e in iv[1:]:
    if s > cur_e: busy += cur_e - cur_s; cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"kernels: first start to last end {(iv[-1][1] - t_first) / 1e6:.1f} ms, union of kernel time {busy / 1e6:.1f} ms")
PY
