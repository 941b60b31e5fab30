#!/bin/bash
# round 6: the randomised parity run on the round's kernels (walk state in LDS, constant table addresses) and on back-end #0 (HPK_FUZZ_CPU: every
# case through hpk_create(-1) as well): the slices of round 5 on new seeds, PAR processes side by side on the one GPU (the run is bound by the oracle,
# one core per process) -> gpurun_out/fuzz_r06.txt
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fz
PAR=${PAR:-12}
export HPK_FUZZ_CPU=${HPK_FUZZ_CPU:-4}
run_slice() {   # name, cases, first seed, env...
  local name=$1 n=$2 base=$3; shift 3
  local per=$(( (n + PAR - 1) / PAR ))
  for i in $(seq 0 $((PAR - 1))); do
    local lo=$((base + i * per)); local cnt=$per
    [ $((i * per + per)) -gt $n ] && cnt=$((n - i * per))
    [ $cnt -le 0 ] && continue
    ( env "$@" timeout 2400 python scripts/gpu_fuzz.py $cnt $lo > gpurun_out/fz/${name}_$i.txt 2>&1 ) &
  done
  wait
  python - $name <<'PY'
import ast, glob, re, sys
tot, secs, cases, bad = {}, 0.0, 0, []
for f in sorted(glob.glob('gpurun_out/fz/%s_*.txt' % sys.argv[1])):
    for l in open(f):
        m = re.match(r'fuzz\[(\w+)\]: seeds (\d+)\.\.(\d+), (\d+) cases in (\d+) s: (\{.*\})', l)
        if m:
            cases += int(m.group(4)); secs = max(secs, float(m.group(5)))
            for k, v in ast.literal_eval(m.group(6)).items():
                tot[k] = tot.get(k, 0) + v
        elif l.startswith(('MISMATCH', 'CRASH')):
            bad.append(l.strip()[:400])
print('fuzz[%s]: %d cases, slowest process %.0f s: %s' % (sys.argv[1], cases, secs, tot))
for b in bad[:20]:
    print('  ' + b)
PY
}
{
  echo "# scripts/measure/gpu_r06_fuzz.sh (seed offset ${SEED_OFF:-0}): scripts/gpu_fuzz.py, HIP path AND back-end #0 (HPK_FUZZ_CPU=$HPK_FUZZ_CPU threads) vs the numpy oracle, $PAR processes side by side"
  run_slice small ${NSMALL:-4800} $((1200000 + ${SEED_OFF:-0}))
  run_slice big ${NBIG:-240} $((1300000 + ${SEED_OFF:-0})) HPK_FUZZ_BIG=1
  run_slice wide ${NWIDE:-12} $((1350000 + ${SEED_OFF:-0})) HPK_FUZZ_WIDE=1
  echo "# every case with structure (HPK_FUZZ_STRUCT=1)"
  run_slice struct ${NSTRUCT:-960} $((1400000 + ${SEED_OFF:-0})) HPK_FUZZ_STRUCT=1
  run_slice structbig ${NSTRUCTBIG:-72} $((1500000 + ${SEED_OFF:-0})) HPK_FUZZ_STRUCT=1 HPK_FUZZ_BIG=1
} 2>&1 | tee gpurun_out/fuzz_r06${SEED_OFF:+_off$SEED_OFF}.txt
