#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04b; mkdir -p $O
timeout 900 python -m pytest "tests/test_gpu_fused.py::test_fused_equals_two_kernel_path[pw1-ww1]" "tests/test_gpu_ref_big.py::test_wide_band_vs_oracle[wide_p4w7]" -q --show-capture=no 2>&1 | tail -120 | tee $O/pytest_two.txt
