#!/bin/bash
# The slow full-size parity test (chr1 @5 kb, (4,7), n = 49 792, num = 2011 against the oracle): ~10 GB of host memory
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
HPK_SLOW=1 timeout 1500 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -k "vs_oracle" --durations=5 2>&1 | tail -15 | tee gpurun_out/slow_tests.txt
