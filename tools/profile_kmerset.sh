# per-kernel time of GPU k-mer counting (run on the GPU box from the repo root): rocprofv3 kernel trace of tools/exp_kmerset.py
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python3 tools/exp_kmerset.py > gpurun_out/r03_kmerset_wall.json 2> gpurun_out/r03_kmerset_wall.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03_kmerset_prof -- python3 tools/exp_kmerset.py > gpurun_out/r03_kmerset_prof.log 2>&1
cat gpurun_out/r03_kmerset_wall.json
f=$(ls gpurun_out/r03_kmerset_prof/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -25 "$f"
