#!/bin/bash
# round 5: what values that depend on the chromosome alone cost - HPK_SPEC_HALO = 1 (the library's default: the inherited layout, lean tiles),
# 2 (the command lines' default: every chromosome ends up under the layout of its own frozen width, computed once more if need be) and 0 (round
# 4's --deterministic: the plan's own layout for everybody); same box
cd $GRAFT_REPO_ROOT
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; c=d["config"]; print("%-22s spec_halo %s  value %.4g  ms/chrom %.4f  stencil/chrom %.4f  redone %d of %d" % (sys.argv[1], sys.argv[2], d["value"], c["ms_per_chromosome"], r["kernel_ms_per_chromosome"], c["passes_redone_in_full"], c["chromosomes_per_step"] * d["steps"]))'
for sh in 1 2 0; do
  HPK_SPEC_HALO=$sh python bench.py --cpu-rows 0 --no-extra --no-probes --steps 10 2>/dev/null | python -c "$P" mixed $sh
  HPK_SPEC_HALO=$sh python bench.py --cpu-rows 0 --no-extra --no-probes --steps 10 --structure 2>/dev/null | python -c "$P" mixed_structure $sh
  HPK_SPEC_HALO=$sh python bench.py --config chr1_10kb_union --cpu-rows 0 --no-extra --no-probes --steps 5 --warmup 2 2>/dev/null | python -c "$P" union $sh
  HPK_SPEC_HALO=$sh python bench.py --config chr1_5kb --cpu-rows 0 --no-extra --no-probes --steps 5 --warmup 2 2>/dev/null | python -c "$P" chr1_5kb $sh
  HPK_SPEC_HALO=$sh python bench.py --config wg_10kb_union --cpu-rows 0 --no-extra --no-probes --steps 5 --warmup 2 2>/dev/null | python -c "$P" wg_10kb_union $sh
done
