#!/bin/bash
# second measurement call of round 4: the scan A/B micro-benchmark, the fused kernel's ablation and counters on the final kernels,
# the randomised parity run (two-kernel path, then a slice scored inside the stencil on fresh seeds)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
  echo "# scripts/ubench/scan_ab.hip on $(python -c 'import torch;print(torch.cuda.get_device_name(0))' 2>/dev/null)"
  hipcc -O3 --offload-arch=gfx950 scripts/ubench/scan_ab.hip -o /tmp/scan_ab && timeout 120 /tmp/scan_ab && timeout 120 /tmp/scan_ab | tail -5
} > gpurun_out/scan_ab.txt 2>&1
bash scripts/gpu_r04_ablate.sh > /dev/null 2>&1
rm -f gpurun_out/pmc_fused.txt
bash scripts/gpu_r04_pmc_fused.sh > gpurun_out/pmc_fused.txt 2>&1
rm -rf gpurun_out/pmc_fused
NSMALL=5000 NBIG=200 NWIDE=8 bash scripts/gpu_fuzz_round.sh > /dev/null 2>&1
cp gpurun_out/fuzz.txt gpurun_out/fuzz_two_kernel.txt
HPK_FUSE=1 NSMALL=1500 NBIG=60 NWIDE=2 SBASE=500000 BBASE=600000 WBASE=700000 bash scripts/gpu_fuzz_round.sh > /dev/null 2>&1
cp gpurun_out/fuzz.txt gpurun_out/fuzz_fused.txt
cat gpurun_out/scan_ab.txt gpurun_out/r04c/ablate.txt gpurun_out/fuzz_two_kernel.txt gpurun_out/fuzz_fused.txt
