cd /tmp && export TMPDIR=/tmp
for lib in libhpk.so libhpk_exp_ir.so; do
rm -rf /tmp/wg
HPK_LIB=$GRAFT_REPO_ROOT/hicpeaks_amd/$lib rocprofv3 --kernel-trace --stats -d /tmp/wg -o k --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --config wg_5kb --steps 4 --warmup 1 --cpu-rows 0 > /tmp/wg.log 2>&1
f=$(find /tmp/wg -name "*kernel_stats.csv" | head -1); echo $lib; grep "hpk_ir_partial" $f | cut -c1-140
done
