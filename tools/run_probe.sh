# build first (in the container): make -C tools
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
(timeout 120 ./tools/bin/gather_probe 11 1526 120; timeout 120 ./tools/bin/gather_probe 12 1526 120; timeout 120 ./tools/bin/gather_probe 9 1526 120) > gpurun_out/probe3.log 2>&1
for v in 11 12; do
  rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d gpurun_out/pmcC_v$v -- ./tools/bin/gather_probe $v 1526 120 > gpurun_out/pmcC_v$v.log 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum --output-format csv -d gpurun_out/pmcD_v$v -- ./tools/bin/gather_probe $v 1526 120 > gpurun_out/pmcD_v$v.log 2>&1
done
cat gpurun_out/probe3.log
