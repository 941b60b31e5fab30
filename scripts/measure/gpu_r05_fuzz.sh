#!/bin/bash
# round 5: the randomised parity run on the round's kernels (hpk_stencil_lean + hpk_stencil_s + the redo queue, per-band tile geometry): the
# slices of scripts/measure/gpu_fuzz_round.sh (a quarter of whose cases carry structure, every weight-input case once more with all chunks lean) and
# a slice with structure in every case -> gpurun_out/fuzz.txt
cd $GRAFT_REPO_ROOT
NSMALL=${NSMALL:-4200} NBIG=${NBIG:-200} NWIDE=${NWIDE:-6} SBASE=800000 BBASE=900000 WBASE=950000 bash scripts/measure/gpu_fuzz_round.sh > /dev/null 2>&1
{
  cat gpurun_out/fuzz.txt
  echo "# every case with structure (HPK_FUZZ_STRUCT=1)"
  HPK_FUZZ_STRUCT=1 timeout 1500 python scripts/gpu_fuzz.py ${NSTRUCT:-800} 1000000 2>&1 | tail -5
  HPK_FUZZ_STRUCT=1 HPK_FUZZ_BIG=1 timeout 1500 python scripts/gpu_fuzz.py ${NSTRUCTBIG:-60} 1100000 2>&1 | tail -5
} > gpurun_out/fuzz_r05.txt
cat gpurun_out/fuzz_r05.txt
