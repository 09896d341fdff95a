# SQ counters of ONE kernel of the long-read path at one read length (tools/exp_readlen_route.py): where its waves' cycles go.
#   bash tools/pmc_kernel.sh <kernel regex> <read length> [tag]      (on the GPU box)
RE=${1:-k_long_first_flags}; L=${2:-100000}; TAG=${3:-pmc_$RE}
cd $GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
B="python3 tools/exp_readlen_route.py $L"
pmc() { d=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-include-regex "$RE" --output-format csv -d $O/$d -- $B > $O/$d.log 2>&1
  f=$(find $O/$d -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $O/$d.csv; rm -rf $O/$d; }
pmc a SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT
pmc b SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES
pmc c SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_ATOMIC_RETURN SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_WAVES GRBM_GUI_ACTIVE
python3 - $O <<'PY'
import csv, sys, collections
O = sys.argv[1]
for f in ("a", "b", "c"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    try: rows = list(csv.DictReader(open(f"{O}/{f}.csv")))
    except Exception as e: print(f, e); continue
    for r in rows:
        k = r["Kernel_Name"][:40] + " grid " + r.get("Grid_Size", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, d in acc.items():
        calls = len({r["Dispatch_Id"] for r in rows if (r["Kernel_Name"][:40] + " grid " + r.get("Grid_Size", "")) == k})
        print(k, "dispatches", calls, {c: round(v / calls) for c, v in d.items()})
PY
