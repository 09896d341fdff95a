# Counters of the k-mer set's sort kernels (tools/exp_kmerset.py, EXP_TARGET=1: the set built for an index): fabric reads / writes by size,
# and where the waves' cycles go.  Run on the GPU box: bash tools/pmc_kmerset.sh <tag>
TAG=${1:-r06_kmerset_pmc}
cd $GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
export EXP_TARGET=${EXP_TARGET-1} EXP_ITERS=3
B="python3 tools/exp_kmerset.py"
RE="k_part_scatter|k_run_bucket_sort|k_rle|k_part_hist"
pmc() { d=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-include-regex "$RE" --output-format csv -d $O/$d -- $B > $O/$d.log 2>&1
  f=$(find $O/$d -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $O/$d.csv; rm -rf $O/$d; }
pmc a TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
pmc b SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT
pmc c SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
python3 - $O <<'PY'
import csv, sys, collections
O = sys.argv[1]
for f in ("a", "b", "c"):
    try: rows = list(csv.DictReader(open(f"{O}/{f}.csv")))
    except Exception as e: print(f, e); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
    for r in rows:
        k = r["Kernel_Name"].split("(")[0][:48]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
    for k, d in acc.items():
        print(f"{k:50s} x{len(disp[k]):3d}", {c: round(v / len(disp[k])) for c, v in d.items()})
PY
