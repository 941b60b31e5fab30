cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03b
python scripts/host_e2e.py > gpurun_out/r03b/host_e2e.txt 2> gpurun_out/r03b/host_e2e.err
cat gpurun_out/r03b/host_e2e.txt; tail -5 gpurun_out/r03b/host_e2e.err
python scripts/wg_hostprof.py wg_10kb_union 2>&1 | tail -40
