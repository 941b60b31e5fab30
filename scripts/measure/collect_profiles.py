#!/usr/bin/env python
"""gpurun_out/<tag>/ (scripts/measure/profile_round.sh) -> the summaries kept under profiles/: <tag>_bench.json,
<tag>_bench_other_configs.jsonl, <tag>_kernel_stats.csv, <tag>_pmc_summary.txt and traffic.json (HBM bytes per stencil
launch = 2 x FETCH_SIZE + WRITE_SIZE, FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md / the calibration in
profiles/r01_pmc_summary.txt).  usage: collect_profiles.py <tag> [--no-copy]"""
import csv, glob, json, os, shutil, sys, collections

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tag = sys.argv[1]
copy = '--no-copy' not in sys.argv
out = os.path.join(REPO, 'gpurun_out', tag)
prof = os.path.join(REPO, 'profiles')


def pmc_means(d, pat):
    """counter -> (mean per launch, launches).  The stencil is up to three kernels per launch of a batch since round 5
    (hpk_stencil_lean, hpk_stencil_s, hpk_stencil_s over the redo queue): `hpk_stencil` = the sum of their means."""
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(os.path.join(d, '**', '*_counter_collection.csv'), recursive=True)):
        for row in csv.DictReader(open(f)):
            if pat in row['Kernel_Name']:
                acc[row['Counter_Name']][row['Kernel_Name']].append(float(row['Counter_Value']))
    return {k: (sum(sum(v) / len(v) for v in byk.values()), max(len(v) for v in byk.values())) for k, byk in acc.items()}


have_raw = bool(glob.glob(os.path.join(out, '**', '*_counter_collection.csv'), recursive=True))
lines = ['# rocprofv3 --pmc passes of scripts/measure/profile_round.sh %s, mean per dispatch' % tag]
traffic = {'_comment': 'HBM traffic per stencil launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes; KiB; FETCH_SIZE '
                       'doubled per the gfx950 note in MI355X_MICROARCH.md, calibrated in profiles/r01_pmc_summary.txt); bench.py copies '
                       'the entry of the configuration it runs into roofline.traffic'}
def group_of(name):
    try:
        return int(open(os.path.join(out, 'pmc_group_%s.txt' % name)).read())
    except Exception:
        return 1


for cfg in ('chr1_10kb', 'chr1_10kb_union', 'chr1_5kb', 'deep_1kb'):
    vals = {}
    G = group_of(cfg)           # chromosomes per launch of the counter passes: figures below are per chromosome
    for cnt in ('FETCH_SIZE', 'WRITE_SIZE', 'SQ_INSTS_VALU', 'SQ_INSTS_SALU'):
        for kern in ('hpk_stencil', 'hpk_score'):
            m = pmc_means(os.path.join(out, 'pmc_%s_%s' % (cfg, cnt)), kern)
            if cnt in m:
                vals[(kern, cnt)] = (m[cnt][0] / G, m[cnt][1])
                lines.append('%-16s %-12s %-13s launches=%d of %d chromosomes, per chromosome %.5g %s' % (cfg, kern, cnt, m[cnt][1], G, m[cnt][0] / G,
                                                                                                      'wave-instructions' if cnt.startswith('SQ_') else 'KiB'))
    if ('hpk_stencil', 'FETCH_SIZE') in vals and ('hpk_stencil', 'WRITE_SIZE') in vals:
        tb = int((2 * vals[('hpk_stencil', 'FETCH_SIZE')][0] + vals[('hpk_stencil', 'WRITE_SIZE')][0]) * 1024)
        traffic[cfg] = {'traffic_bytes': tb, 'source': 'profiles/%s_pmc_summary.txt' % tag,
                        'note': 'hpk_stencil*, per chromosome of a launch: 2 x FETCH_SIZE + WRITE_SIZE'}
        lines.append('%-16s stencil HBM traffic per chromosome: %.1f MB' % (cfg, tb / 1e6))
    if cfg in traffic and ('hpk_score', 'FETCH_SIZE') in vals and ('hpk_score', 'WRITE_SIZE') in vals:
        traffic[cfg]['score_traffic_bytes'] = int((2 * vals[('hpk_score', 'FETCH_SIZE')][0] + vals[('hpk_score', 'WRITE_SIZE')][0]) * 1024)
    # vector instructions per chromosome (wave-instructions: bench.py's roofline_valu prices them at 4 cycles on 1 024 SIMDs)
    if cfg in traffic and ('hpk_stencil', 'SQ_INSTS_VALU') in vals:
        traffic[cfg]['valu_insts'] = int(vals[('hpk_stencil', 'SQ_INSTS_VALU')][0])
    if cfg in traffic and ('hpk_score', 'SQ_INSTS_VALU') in vals:
        traffic[cfg]['score_valu_insts'] = int(vals[('hpk_score', 'SQ_INSTS_VALU')][0])
    # ... and scalar ones (one scalar unit per CU: bench.py prices them at the measured 2.07 cycles each, profiles/r06_instruction_cost.txt)
    if cfg in traffic and ('hpk_stencil', 'SQ_INSTS_SALU') in vals:
        traffic[cfg]['salu_insts'] = int(vals[('hpk_stencil', 'SQ_INSTS_SALU')][0])
    if cfg in traffic and ('hpk_score', 'SQ_INSTS_SALU') in vals:
        traffic[cfg]['score_salu_insts'] = int(vals[('hpk_score', 'SQ_INSTS_SALU')][0])
Gs = group_of('sq')
for kern in ('hpk_stencil', 'hpk_score'):
    lines.append('## %s (chr1_10kb), SQ / TCC counters per launch of %d chromosomes' % (kern, Gs))
    for d in sorted(glob.glob(os.path.join(out, 'pmc_sq_*/'))):
        for k, (v, n) in sorted(pmc_means(d, kern).items()):
            lines.append('%-28s n=%d mean=%.4g' % (k, n, v))
# (the GPU box deletes the raw counter files after this step: a later run here, without them, only copies)
if have_raw:
    open(os.path.join(out, 'pmc_summary.txt'), 'w').write('\n'.join(lines) + '\n')
    if len(traffic) > 1:
        json.dump(traffic, open(os.path.join(out, 'traffic.json'), 'w'), indent=1)
ks = glob.glob(os.path.join(out, 'trace', '**', '*kernel_stats.csv'), recursive=True)
if ks:
    # the product's kernels only (the bench's band generator runs torch kernels of its own: 40 % of round 5's file), percentages
    # recomputed over what is kept
    rows = [r for r in csv.DictReader(open(ks[0])) if 'hpk_' in r['Name']]
    tot = sum(float(r['TotalDurationNs']) for r in rows) or 1.0
    with open(os.path.join(out, 'kernel_stats.csv'), 'w', newline='') as f:
        wr = csv.DictWriter(f, fieldnames=list(rows[0].keys()) if rows else ['Name'])
        wr.writeheader()
        for r in rows:
            if 'Percentage' in r:
                r['Percentage'] = '%.6f' % (100.0 * float(r['TotalDurationNs']) / tot)
            wr.writerow(r)
print('\n'.join(lines[:80]))
if copy:
    shutil.copy(os.path.join(out, 'bench.json'), os.path.join(prof, '%s_bench.json' % tag))
    with open(os.path.join(prof, '%s_bench_other_configs.jsonl' % tag), 'w') as f:
        for p in sorted(glob.glob(os.path.join(out, 'bench_*.json'))):
            t = open(p).read().strip()
            if t:
                try:            # which run of the round script the line is (file name minus "bench_")
                    dd = json.loads(t)
                    dd['config']['run'] = os.path.basename(p)[len('bench_'):-len('.json')]
                    t = json.dumps(dd)
                except Exception:
                    pass
                f.write(t + '\n')
    if os.path.exists(os.path.join(out, 'kernel_stats.csv')):
        shutil.copy(os.path.join(out, 'kernel_stats.csv'), os.path.join(prof, '%s_kernel_stats.csv' % tag))
    shutil.copy(os.path.join(out, 'pmc_summary.txt'), os.path.join(prof, '%s_pmc_summary.txt' % tag))
    if os.path.exists(os.path.join(out, 'traffic.json')):
        shutil.copy(os.path.join(out, 'traffic.json'), os.path.join(prof, 'traffic.json'))
    print('copied to profiles/')
