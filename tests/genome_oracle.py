"""Host-only half of tests/test_gpu_genome.py: the oracle on one chromosome of a genome configuration, run in worker processes
(importable without a GPU and without pytest)."""
import numpy as np

from hicpeaks_amd import synthetic

SIG, MIN_READS, MAXWW = 0.05, 16, 10
GENOMES = {
    'wg_10kb_union': dict(res=10000, maxapart=5000000, pw=[1, 2, 4], ww=[3, 5, 7], depth=60.0, nloops=400, n_ref=24896,
                          oracle=['19', '20', '21', '22', 'X'], min_sig=10, min_final=5),
    'wg_5kb': dict(res=5000, maxapart=10000000, pw=[4], ww=[7], depth=25.0, nloops=800, n_ref=49792, oracle=['21', '22'],
                   min_sig=3, min_final=2),      # (the two smallest chromosomes at depth 25: a handful of pixels pass)
}


def _oracle_job(job):
    """(worker process, host only) the chromosome's band as the SURVEY 8-D2 recipe gives it, and the oracle's verdict on it"""
    name, chrom, n, seed = job
    from oracle import hiccups_oracle as orc
    cfg = GENOMES[name]
    num = cfg['maxapart'] // cfg['res'] + MAXWW + 1
    raw, weight, _ = synthetic.synth_band(n, num, depth=cfg['depth'], nloops=max(1, cfg['nloops'] * n // cfg['n_ref']), seed=seed)
    IR, cband, biases = orc.prep_from_band(raw, weight, min(cfg['ww']))
    det = {}
    want = orc.hiccups(raw, cband, biases, biases, IR, n, num, detail=det, pw=cfg['pw'], ww=cfg['ww'], maxww=MAXWW, sig=SIG,
                       maxapart=cfg['maxapart'], res=cfg['res'], min_local_reads=MIN_READS, min_marginal_peaks=2, onlyanchor=False)
    return name, chrom, raw.astype(np.float32), weight, det, want
