mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
(timeout 120 ./tools/gather_probe 8 1526 120 0; timeout 120 ./tools/gather_probe 0 1526 120 1; timeout 120 ./tools/gather_probe 0 1526 120 2; timeout 120 ./tools/gather_probe 3 1526 120 2; timeout 120 ./tools/gather_probe 8 1526 120 2 ) > gpurun_out/probe2.log 2>&1
run_pmc() { # name variant alloc
  rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d gpurun_out/pmcA_$1 -- ./tools/gather_probe $2 1526 120 $3 > gpurun_out/pmcA_$1.log 2>&1
  rocprofv3 --pmc TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RD_UNCACHED_32B_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum --output-format csv -d gpurun_out/pmcB_$1 -- ./tools/gather_probe $2 1526 120 $3 > gpurun_out/pmcB_$1.log 2>&1
}
run_pmc v0a0 0 0
run_pmc v8a0 8 0
run_pmc v0a2 0 2
run_pmc v0a1 0 1
cat gpurun_out/probe2.log
