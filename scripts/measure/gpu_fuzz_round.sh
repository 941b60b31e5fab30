#!/bin/bash
# The randomised parity run of the round: small / big / wide slices of scripts/gpu_fuzz.py, tally -> gpurun_out/fuzz.txt
cd $GRAFT_REPO_ROOT
{
  echo "# scripts/gpu_fuzz.py on $(python -c 'import torch;print(torch.cuda.get_device_name(0))' 2>/dev/null), libhpk.so of this tree; HIP path vs numpy oracle"
  timeout 1500 python scripts/gpu_fuzz.py ${NSMALL:-3000} ${SBASE:-100000} 2>&1 | tail -5
  HPK_FUZZ_BIG=1 timeout 1500 python scripts/gpu_fuzz.py ${NBIG:-150} ${BBASE:-200000} 2>&1 | tail -5
  HPK_FUZZ_WIDE=1 timeout 2400 python scripts/gpu_fuzz.py ${NWIDE:-10} ${WBASE:-300000} 2>&1 | tail -5
} | tee gpurun_out/fuzz.txt
