#!/bin/bash
# the row-per-DPP-row stencil: parity first (stop at the first failure), then the bench lines it is judged on
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --show-capture=no 2>&1 | tail -15 | tee $O/parity.txt
if grep -q "failed\|error" $O/parity.txt; then echo "PARITY FAILED"; [ -z "$FORCE" ] && exit 1; fi
P='import sys,json
o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%s ms/chrom %.4f  stencil/chrom %.4f frac %.3f value %.4g" % (sys.argv[1], o["config"].get("ms_per_chromosome") or 0, o["roofline"]["kernel_ms_per_chromosome"], o["roofline"]["frac"], o["value"]))'
timeout 600 python bench.py --no-extra --cpu-rows 0 --depths 60 --steps 5 --warmup 2 2>>$O/bench.err | python -c "$P" d60 | tee -a $O/bench.txt
timeout 600 python bench.py --no-extra --cpu-rows 0 --steps 5 --warmup 2 2>>$O/bench.err | python -c "$P" mixed | tee -a $O/bench.txt
timeout 600 python bench.py --no-extra --cpu-rows 0 --config chr1_10kb_union --steps 5 --warmup 2 2>>$O/bench.err | python -c "$P" union | tee -a $O/bench.txt
timeout 600 python bench.py --no-extra --cpu-rows 0 --config chr1_5kb --steps 20 --warmup 3 2>>$O/bench.err | python -c "$P" 5kb | tee -a $O/bench.txt
timeout 600 python bench.py --no-extra --cpu-rows 0 --balanced-f64 --steps 20 --warmup 3 2>>$O/bench.err | python -c "$P" balf64 | tee -a $O/bench.txt
if [ -n "$FULL" ]; then timeout 1800 python -m pytest tests -m gpu -q --show-capture=no 2>&1 | tail -25 | tee $O/pytest.txt; fi
tail -5 $O/bench.err
