cd $GRAFT_REPO_ROOT
for r in 16 13 12 11; do
 echo "== risk 2^-$r"
 HPK_RISK_LOG2=$r timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "dynamic_range" 2>&1 | grep -E "passed|failed|Max rel|Mismatch" | head -6
 for cfg in chr1_10kb chr1_5kb; do
  HPK_RISK_LOG2=$r python bench.py --config $cfg --steps 3 --warmup 1 --batch 50 --cpu-rows 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg stencil_ms %.4f ms/chrom %.4f freeze %.4f' % (d['roofline']['kernel_ms'], d['config']['ms_per_chromosome'], d['phases_ms']['freeze']))"
  HPK_RISK_LOG2=$r HPK_LIB=$PWD/hicpeaks_amd/libhpk_clk.so HPK_CLK_DUMP=gpurun_out/clk_r.bin python bench.py --config $cfg --steps 1 --warmup 1 --batch 2 --cpu-rows 0 --pipeline-depth 1 > /dev/null 2>&1
  python scripts/clk_summary.py gpurun_out/clk_r.bin | grep -E "redone|ticks per wave"
 done
done
HPK_ROUNDS=1 python bench.py --steps 3 --warmup 1 --batch 50 --cpu-rows 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('rounds=1: ms/chrom %.4f copied back %d' % (d['config']['ms_per_chromosome'], d['config']['records_copied_back']), d['phases_ms'])"
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_seam.py -m gpu -q 2>&1 | tail -3
