"""Text output of the two command lines (scripts/pyHICCUPS:200-210, scripts/pyBHFDR:169-176) and their
argument parsers; the executables in scripts/ are thin wrappers around `main_hiccups` / `main_bhfdr`."""

HICCUPS_FMT = ('{0}\t{1}\t{2}\t{3}\t{4}\t{5}\t{6}\t{7:.3g}\t{8}\t{9}\t{10:.3g}\t{11:.3g}\t{12:.3g}'
               '\t{13:.3g}\t{14:.3g}\t{15:.3g}\n')
BHFDR_FMT = '{0}\t{1}\t{2}\t{3}\t{4}\t{5}\t{6}\t{7:.3g}\t{8}\t{9}\t{10:.3g}\t{11:.3g}\t{12:.3g}\n'


def _format(fmt, chrom, table, res, sort):
    c = 'chr' + chrom.lstrip('chr')
    keys = sorted(table) if sort else list(table)
    out = []
    for px in keys:
        tmp = table[px]
        out.append(fmt.format(*((c, px[0], px[0] + res, c, px[1], px[1] + res, '.', tmp[3], '.', '.') + tuple(tmp[4:]))))
    return ''.join(out)


def format_hiccups(chrom, table, res, sort=False):
    """16 columns: chrom x x+res chrom y y+res . O . . fold_K p_K q_K fold_Y p_Y q_Y (pyHICCUPS:202-205)."""
    return _format(HICCUPS_FMT, chrom, table, res, sort)


def format_bhfdr(chrom, table, res, sort=False):
    """13 columns (pyBHFDR:171)."""
    return _format(BHFDR_FMT, chrom, table, res, sort)
