#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c; mkdir -p $O; rm -f $O/ablate.txt
P='import sys,json
o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("ms/chrom %.4f  stencil/chrom %.4f  kernel %s" % (o["config"]["ms_per_chromosome"], o["roofline"]["kernel_ms_per_chromosome"], o["roofline"]["kernel"]))'
for stop in 0 8 9; do
  echo "== fused, dbg_stop=$stop" | tee -a $O/ablate.txt
  HPK_FUSE=1 HPK_DBG_STOP=$stop timeout 600 python bench.py --no-extra --cpu-rows 0 --depths 60 --steps 5 --warmup 2 --no-probes 2>/dev/null | python -c "$P" | tee -a $O/ablate.txt
done
echo "== two-kernel" | tee -a $O/ablate.txt
HPK_FUSE=0 timeout 600 python bench.py --no-extra --cpu-rows 0 --depths 60 --steps 5 --warmup 2 --no-probes 2>/dev/null | python -c "$P" | tee -a $O/ablate.txt
