#!/bin/bash
# round 5: is the wait for the prefetched rows DRAM latency?  lean tiles alone, 8 distinct bands (408 MB) against one band (51 MB: Infinity Cache)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/clk; mkdir -p $OUT
for dist in 8 1; do
  for stop in 10 0; do
    echo "== --depths 60 dbg_stop $stop distinct $dist"
    HPK_SPEC_FORCE=6 HPK_DBG_STOP=$stop HPK_LIB=$PWD/hicpeaks_amd/libhpk_clk1.so HPK_CLK_DUMP=$OUT/ab.bin python bench.py --depths 60 --no-extra --steps 2 --warmup 1 --batch 8 --group 8 --distinct $dist --cpu-rows 0 --pipeline-depth 1 --no-probes 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('tiles/chrom %.0f lean/chrom %.0f' % (c['tiles']/16., c['lean_tiles']/16.))"
    python scripts/clk_summary.py $OUT/ab.bin 8 | head -9
    python - <<'PY'
import numpy as np
raw=np.fromfile('gpurun_out/clk/ab.bin',dtype=np.uint64).reshape(-1,16,8)
raw=raw[raw[:,:,:7].sum(axis=(1,2))>0]
raw[:,:,6]&=np.uint64((1<<40)-1); raw[:,:,7]&=np.uint64((1<<40)-1)
a=raw.astype(float)/8
print('per wave: wait | cells | list | prefix | phase2+barriers | batches | end')
for w in range(16):
    print(w, ' '.join('%7.0f'%a[:,w,i].mean() for i in range(7)))
PY
  done
done
