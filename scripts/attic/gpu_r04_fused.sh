#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_parity.py -q --show-capture=no -x -k "fused or inside_the_stencil" 2>&1 | tail -40 | tee $O/pytest_fused.txt
bash scripts/gpu_r04_ablate.sh
