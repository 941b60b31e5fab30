#!/bin/bash
# record bound per chromosome by depth class (option spec_class): the tests around the bounds, then the default workload with and without
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_ref_big.py tests/test_gpu_multiproc.py -m gpu -x -q --show-capture=no 2>&1 | tail -4
P='import sys,json
o=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=o["config"]; print("%s ms/chrom %.4f stencil %.4f score %.4f value %.4g redone %s rescored %s" % (sys.argv[1], c["ms_per_chromosome"], o["roofline"]["kernel_ms_per_chromosome"], o["phases_ms"]["score"], o["value"], c["passes_redone_in_full"], c["passes_rescored"]))'
for cl in 1 0; do
  HPK_SPEC_CLASS=$cl python bench.py --no-extra --cpu-rows 0 --steps 5 --warmup 3 2>/dev/null | python -c "$P" "class $cl mixed"
  HPK_SPEC_CLASS=$cl python bench.py --no-extra --cpu-rows 0 --depths 60 --steps 5 --warmup 3 2>/dev/null | python -c "$P" "class $cl d60  "
done
