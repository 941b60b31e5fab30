#!/bin/bash
# Round-4 first check: the new GPU tests (reference-pinned big fixtures, N-process paths, deterministic mode), the whole GPU
# suite as the driver runs it, the default bench line on the new workload (64 distinct mixed-depth bands)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04a; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_ref_big.py tests/test_gpu_multiproc.py -x -q 2>&1 | tail -25 | tee $O/pytest_new.txt
timeout 1500 python -m pytest tests -m gpu -x -q --durations=12 2>&1 | tail -30 | tee $O/pytest.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err
cut -c1-3000 $O/bench.json
timeout 300 python bench.py --depths 60 --no-extra --cpu-rows 0 > $O/bench_depth60.json 2>> $O/bench.err
cut -c1-1500 $O/bench_depth60.json
