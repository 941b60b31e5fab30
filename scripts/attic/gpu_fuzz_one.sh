#!/bin/bash
# one fuzz seed under the library's A/B switches
cd $GRAFT_REPO_ROOT
SEED=${1:-1300317}
for envs in "" "HPK_FREEZE_TICKET=1" "HPK_FREEZE_KERNEL=1" "HPK_OLD_STENCIL=1" "HPK_NO_SINGLE=1" "HPK_SPEC=0"; do
  echo "== $envs: $(env $envs HPK_FUZZ_VERBOSE=1 timeout 120 python scripts/gpu_fuzz.py 1 $SEED 2>&1 | grep -a "^ok\|MISMATCH\|CRASH\|both-raise" | cut -c1-160)"
done
