#!/bin/bash
# kernel times of the long-read path at ONE read length (150 Mbases; tools/exp_readlen_route.py): rocprofv3 --kernel-trace --stats.
#   bash tools/prof_readlen.sh 1000000 [tag]    -> gpurun_out/prof_len_<tag>/…kernel_stats.csv and the first rows on stdout
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
L=${1:-1000000}; TAG=${2:-$L}; O=gpurun_out/prof_len_$TAG
rm -rf $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 tools/exp_readlen_route.py $L > $O.jsonl 2> $O.err || exit 1
python3 - $O <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print(f'{r["Name"][:72]:72s} calls {r["Calls"]:>4s}  avg {float(r["AverageNs"]) / 1e3:9.1f} us  {r["Percentage"]:>6s} %')
PY
cat $O.jsonl
