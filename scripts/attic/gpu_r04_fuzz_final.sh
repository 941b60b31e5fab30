#!/bin/bash
# the randomised parity run on the round's final kernels (two-kernel path, then a slice scored inside the stencil), then the slow tests
cd $GRAFT_REPO_ROOT
NSMALL=5000 NBIG=200 NWIDE=8 SBASE=110000 BBASE=210000 WBASE=310000 bash scripts/gpu_fuzz_round.sh > /dev/null 2>&1
cp gpurun_out/fuzz.txt gpurun_out/fuzz_two_kernel_final.txt
HPK_FUSE=1 NSMALL=1500 NBIG=60 NWIDE=2 SBASE=510000 BBASE=610000 WBASE=710000 bash scripts/gpu_fuzz_round.sh > /dev/null 2>&1
cp gpurun_out/fuzz.txt gpurun_out/fuzz_fused_final.txt
grep "^fuzz" gpurun_out/fuzz_two_kernel_final.txt gpurun_out/fuzz_fused_final.txt
HPK_SLOW=1 timeout 2400 python -m pytest tests -m gpu -q --show-capture=no -k "slow or full_size or wide_band or reference_at_size" 2>&1 | tail -8 | tee gpurun_out/slow_tests_final.txt
