# per-kernel time of GPU k-mer counting (run on the GPU box from the repo root): rocprofv3 kernel trace of tools/exp_kmerset.py
# TAG (default r04): file prefix under gpurun_out/;  EXP_TARGET=1: the set built for an index
TAG=${TAG:-r04}
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python3 tools/exp_kmerset.py > gpurun_out/${TAG}_kmerset_wall.json 2> gpurun_out/${TAG}_kmerset_wall.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_kmerset_prof -- python3 tools/exp_kmerset.py > gpurun_out/${TAG}_kmerset_prof.log 2>&1
cat gpurun_out/${TAG}_kmerset_wall.json
f=$(ls gpurun_out/${TAG}_kmerset_prof/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" gpurun_out/${TAG}_kmerset_kernel_stats.csv && head -25 "$f" | cut -c1-150
