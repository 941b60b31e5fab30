#!/bin/bash
# the kernel-trace pass of scripts/gpu_profile_round.sh alone -> gpurun_out/r04/kernel_stats.csv
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --steps 2 --warmup 1 --cpu-rows 0 --no-probes --no-extra 2>/dev/null | python -c "import sys,json; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('untraced kernel_ms', o['roofline']['kernel_ms'])"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o k --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --cpu-rows 0 --no-probes --no-extra > $OUT/trace.log 2>&1
grep -o '"kernel_ms": [0-9.]*' $OUT/trace.log | head -1
cp $(find $OUT/trace -name 'k_kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
grep "hpk_stencil_s" $OUT/kernel_stats.csv | cut -d, -f2-8
tail -5 $OUT/trace.log > $OUT/trace.log.tail; rm -rf $OUT/trace $OUT/trace.log
