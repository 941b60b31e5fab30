#!/bin/bash
# the shader clock while the default bench workload runs (rocm-smi sampled beside it), and while the instruction-cost microbenchmark runs
cd $GRAFT_REPO_ROOT
echo "== idle"; rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4
( python bench.py --cpu-rows 0 --no-extra --no-probes --steps 60 > /tmp/b.json 2>/dev/null ) &
BP=$!
sleep 12
echo "== bench running"
for i in 1 2 3 4 5 6 7 8; do rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Average Graphics Package Power\|Socket Graphics Package Power" | tr '\n' ' '; echo; sleep 0.4; done
wait $BP
python -c "import json; d=json.load(open('/tmp/b.json')); print('bench value %.4g stencil %.4f' % (d['value'], d['roofline']['kernel_ms_per_chromosome']))"
