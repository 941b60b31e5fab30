cd $GRAFT_REPO_ROOT
HPK_BENCH_FORCE_DIST=1 python bench.py --steps 2 --warmup 1 --cpu-rows 0 2>/dev/null | grep metric | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash scripts/gpu_cli_wg.sh 2>&1 | tail -12
