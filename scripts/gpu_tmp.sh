cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
G=32 STEPS=5 CFGS="chr1_10kb chr1_10kb_union chr1_5kb" bash scripts/gpu_exp.sh
G=32 STEPS=5 CFGS="chr1_10kb chr1_10kb_union" bash scripts/gpu_exp.sh
