#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04d; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --show-capture=no 2>&1 | tail -40 | tee $O/pytest.txt
timeout 600 python bench.py --no-extra --cpu-rows 0 > $O/bench.json 2>$O/bench.err; cut -c1-1500 $O/bench.json
timeout 600 python bench.py --no-extra --cpu-rows 0 --depths 60 > $O/bench_d60.json 2>>$O/bench.err; cut -c1-1500 $O/bench_d60.json
timeout 600 python bench.py --no-extra --cpu-rows 0 --config chr1_10kb_union --steps 5 > $O/bench_union.json 2>>$O/bench.err; cut -c1-1200 $O/bench_union.json
