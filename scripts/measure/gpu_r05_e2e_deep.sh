#!/bin/bash
# round 5: the reader's ceiling.  The deep whole-genome 5 kb .mcool of scripts/attic/gpu_r04_e2e_deep.sh (1.07 G pixels); the command line's wall
# time on one GPU with the chunks decoded (a) by a Python thread pool (round 4: HPK_READ_PYTHON=1) and (b) by libhpk's host threads
# (hpk_decode_chunks), by decoding threads; then the stages per chromosome with (b).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05e; mkdir -p $O
DEPTH=${DEPTH:-500}
F=/tmp/hpk_deep.mcool
CH="1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16 17 18 19 20 21 22 X"
{
echo "# host: $(nproc) cores, $(free -g | awk '/Mem:/{print $2}') GiB RAM"
cd /tmp && export TMPDIR=/tmp
PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 $R/scripts/make_cool_deep.py $F --res 5000 --num 2011 --depth $DEPTH --far --threads $(( $(nproc) < 48 ? $(nproc) : 48 )) 2>/dev/null | tail -1
run() {   # label, env...
  local label=$1; shift
  for rep in 1 2; do
    t0=$(date +%s.%N)
    env "$@" python $R/scripts/pyHICCUPS -p $F::/resolutions/5000 -O /tmp/deep_$label.bedpe --pw 4 --ww 7 --maxapart 10000000 --logFile /tmp/deep.log > /dev/null 2>&1
    t1=$(date +%s.%N)
    echo "$label  wall $(python -c "print('%.2f' % ($t1 - $t0))") s  lines $(wc -l < /tmp/deep_$label.bedpe)"
  done
}
run python_pool_64 HPK_READ_PYTHON=1 HPK_READ_THREADS=64
run python_pool_16 HPK_READ_PYTHON=1 HPK_READ_THREADS=16
run native_16 HPK_READ_THREADS=16
run native_64 HPK_READ_THREADS=64
run native_128 HPK_READ_THREADS=128
run native_nopread_64 HPK_READ_NO_PREAD=1 HPK_READ_THREADS=64
cmp /tmp/deep_python_pool_64.bedpe /tmp/deep_native_64.bedpe && echo "identical output (Python pool / native decoder)"
python $R/scripts/host_e2e.py --deep --depth $DEPTH --file $F --chroms $CH 2>&1 | grep -v "^## cold" | tail -32
} 2>&1 | tee $O/host_e2e_deep.txt
