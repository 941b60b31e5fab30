cd $GRAFT_REPO_ROOT
{
  timeout 1500 python scripts/gpu_fuzz.py 4000 500000 2>&1 | tail -3
  HPK_FUZZ_BIG=1 timeout 1500 python scripts/gpu_fuzz.py 200 600000 2>&1 | tail -3
  HPK_FUZZ_WIDE=1 timeout 900 python scripts/gpu_fuzz.py 15 700000 2>&1 | tail -3
} | tee gpurun_out/fuzz2.txt
