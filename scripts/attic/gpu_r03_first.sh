#!/bin/bash
# Round-3 first check: GPU parity tests, default bench line, group sweep
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03a
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee gpurun_out/r03a/pytest.txt
python bench.py > gpurun_out/r03a/bench.json 2> gpurun_out/r03a/bench.err
cut -c1-1500 gpurun_out/r03a/bench.json
GRPS="1 4 32" STEPS=5 BATCH=128 bash scripts/gpu_groups.sh | tee gpurun_out/r03a/groups.txt
for c in chr1_10kb_union chr1_5kb wg_10kb_union wg_5kb; do
  timeout 600 python bench.py --config $c --steps 5 --warmup 1 --cpu-rows 0 2>/dev/null | tail -1 > gpurun_out/r03a/bench_$c.json
  cut -c1-700 gpurun_out/r03a/bench_$c.json
done
