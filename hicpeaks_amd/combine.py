"""Multi-resolution merge of peak calls: the host-side consumer of this package's BEDPE output.

Counterpart of hicpeaks/utilities.py:442-467 (`_parse_peakfile`), 469-552 (`combine_annotations`) and the
scripts/combine-resolutions command line (flags and defaults of scripts/combine-resolutions:12-48).  Pure host
code: a few thousand peaks per resolution, nothing here belongs on the GPU.

Rule, as the reference applies it: walk the resolutions from fine to coarse.  A peak at a finer resolution that has a
neighbour (Euclidean distance between the (start1, start2) corners) at a coarser one is kept and "claims" those
coarser peaks, which are then never emitted; the radius is 2*max_res when both resolutions are below 2*max_res and
5*max_res otherwise.  A peak with no coarser neighbour survives only if its own resolution is <= max_res and it is
either at a resolution >= good_res or short-range (start2 - start1 <= mindis).
"""
import numpy as np


def find_chrom_pre(chromlabels):
    """'chr' if the first label carries the prefix, else '' (utilities.py:432-440)."""
    return 'chr' if chromlabels[0].startswith('chr') else ''


def _parse_peakfile(filpath, skip=1):
    """{chrom label without prefix: [(start1, end1, start2, end2), ...]} from a BEDPE-like file
    (utilities.py:442-467).  The prefix is removed with str.lstrip semantics, like the reference."""
    table = {}
    with open(filpath, 'r') as source:
        for i, line in enumerate(source):
            if i < skip:
                continue
            f = line.rstrip().split()
            table.setdefault(f[0], []).append((int(f[1]), int(f[2]), int(f[4]), int(f[5])))
    pre = find_chrom_pre(list(table.keys()))     # IndexError on an empty file, as upstream
    return {chrom.lstrip(pre): peaks for chrom, peaks in table.items()}


def _keys(chrom, peaks):
    return [(chrom,) + tuple(p[:2]) + (chrom,) + tuple(p[2:]) for p in peaks]


def combine_annotations(byres, good_res=10000, mindis=100000, max_res=10000):
    """Sorted list of (chrom, s1, e1, chrom, s2, e2) (utilities.py:469-552)."""
    if len(byres) == 1:
        out = []
        for r in byres:
            for c in byres[r]:
                out.extend(_keys(c, byres[r][c]))
        return out

    near, far = 2 * max_res, 5 * max_res
    reslist = sorted(byres)
    kept, claimed = set(), set()

    def stands_alone(res, peaks):
        """Mask of peaks that may be emitted without support from a coarser resolution."""
        if res > max_res:
            return np.zeros(len(peaks), dtype=bool)
        if res >= good_res:
            return np.ones(len(peaks), dtype=bool)
        return (peaks[:, 2] - peaks[:, 0]) <= mindis

    for i, fine in enumerate(reslist[:-1]):
        for coarse in reslist[i + 1:]:
            radius = near if (fine < near and coarse < near) else far
            for c, plist in byres[fine].items():
                keys = _keys(c, plist)
                # peaks claimed by a finer resolution earlier on are gone for good; `claimed` only ever receives
                # peaks of resolutions coarser than `fine` inside this pass, so the mask can be taken up front
                live = np.array([k not in claimed for k in keys], dtype=bool)
                if not live.any():
                    continue
                peaks = np.asarray(plist, dtype=np.int64).reshape(-1, 4)
                alone = stands_alone(fine, peaks)
                ref = byres[coarse].get(c, [])
                if len(ref):
                    refa = np.asarray(ref, dtype=np.int64).reshape(-1, 4)
                    dx = (peaks[:, None, 0] - refa[None, :, 0]).astype(np.float64)
                    dy = (peaks[:, None, 2] - refa[None, :, 2]).astype(np.float64)
                    hit = np.sqrt(dx * dx + dy * dy) <= radius
                    hit &= live[:, None]
                    supported = hit.any(axis=1)
                    refkeys = _keys(c, ref)
                    for j in np.nonzero(hit.any(axis=0))[0]:
                        claimed.add(refkeys[j])
                else:
                    supported = np.zeros(len(keys), dtype=bool)
                for k in np.nonzero(live & (supported | alone))[0]:
                    kept.add(keys[k])

    last = reslist[-1]
    for c, plist in byres[last].items():
        peaks = np.asarray(plist, dtype=np.int64).reshape(-1, 4)
        alone = stands_alone(last, peaks)
        for k, key in enumerate(_keys(c, plist)):
            if alone[k] and key not in claimed:
                kept.add(key)
    return sorted(kept)


def format_combined(peak_list):
    """Six-column lines of scripts/combine-resolutions:68-71 ('chr' is prepended unconditionally)."""
    return ''.join('\t'.join(('chr' + t[0], str(t[1]), str(t[2]), 'chr' + t[3], str(t[4]), str(t[5]))) + '\n'
                   for t in peak_list)


def _parser():
    import argparse
    from . import __version__
    p = argparse.ArgumentParser(usage='%(prog)s <-O output> [options]',
                                description='Combine loop calls from different resolutions.',
                                formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument('-v', '--version', action='version', version=' '.join(['%(prog)s', __version__]),
                   help='Print version number and exit.')
    p.add_argument('-O', '--output', help='Output peak file name.')
    p.add_argument('-p', '--paths', nargs='+', help='List of peak file paths at different resolutions.')
    p.add_argument('-R', '--resolutions', type=int, nargs='+',
                   help='List of resolutions corresponding to the input peak files.')
    p.add_argument('-S', '--skip-rows', type=int, default=0, help='Number of leading lines to skip.')
    p.add_argument('-G', '--good-res', type=int, default=20000,
                   help='Peaks found only at resolutions finer than this are kept only when short-range '
                   '(see --min-dis).')
    p.add_argument('-M', '--min-dis', type=int, default=200000, help='See --good-res.')
    p.add_argument('--max-res', type=int, default=10000,
                   help='Only peaks originally called at this or a finer resolution are written.')
    return p


def main_combine(argv=None):
    import sys
    argv = list(sys.argv[1:] if argv is None else argv) or ['-h']
    args = _parser().parse_args(argv)
    byres = {res: _parse_peakfile(path, args.skip_rows) for res, path in zip(args.resolutions, args.paths)}
    peaks = combine_annotations(byres, good_res=args.good_res, mindis=args.min_dis, max_res=args.max_res)
    with open(args.output, 'w') as out:
        out.write(format_combined(peaks))
    return 0
