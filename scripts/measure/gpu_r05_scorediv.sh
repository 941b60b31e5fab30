#!/bin/bash
# round 5: tiles per scoring workgroup (HPK_SCORE_DIV) now that the record bound leaves a quarter of the records: same box, per-phase times
cd $GRAFT_REPO_ROOT
for div in ${DIVS:-8 4 16 32 64 128 8}; do
  echo "== HPK_SCORE_DIV=$div"
  HPK_SCORE_DIV=$div STEPS=${STEPS:-8} bash scripts/measure/gpu_r05_score.sh
done
