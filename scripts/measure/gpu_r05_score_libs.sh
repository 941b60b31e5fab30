#!/bin/bash
# round 5: hpk_score variants, one library each (hicpeaks_amd/libhpk_exp*.so next to libhpk.so), same box, interleaved repeats
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for lib in hicpeaks_amd/libhpk.so hicpeaks_amd/libhpk_exp*.so; do
  echo "== $lib"
  HPK_LIB=$PWD/$lib bash scripts/measure/gpu_r05_score.sh
done
done
