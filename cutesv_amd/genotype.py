"""Host side of the genotype step: the likelihood table behind `gl_idx`.

cal_GL (cuteSV_genotype.py:33-56) is a pure function of (DR, DV) whose inputs collapse, after
its two hard-coded cases (:34-37) and rescale_read_counts (:25-31), onto c0 + c1 <= 100.  The
kernels therefore only produce the table index (csv_gl_index in include/cutesv_hip.h); the
strings that go into the VCF rows are looked up here.  The table is built once per process in
float64 with the same libm/numpy operations the reference applies (pow, log10, np.around), so
the looked-up fields are bit-identical, not merely within the 1e-6 north_star tolerates.
"""
from math import log10

import numpy as np

ERR = 0.1                      # cuteSV_genotype.py:10
PRIOR = float(1 / 3)           # :11
GENOTYPES = ("0/0", "0/1", "1/1")
TABLE_SIZE = 101 * 101 + 2
_SPECIAL = {101 * 101: ("0/1", "3,3,24", "3", "3.0"), 101 * 101 + 1: ("0/1", "3,3,45", "3", "3.0")}
_table = {}


def gl_index(c0, c1):
    """Python twin of csv_gl_index: (DR, DV) -> table index (cuteSV_genotype.py:25-39)."""
    if c0 == 3 and c1 == 1:
        return 101 * 101
    if c0 == 6 and c1 == 2:
        return 101 * 101 + 1
    total = c0 + c1
    if total > 100:
        c0 = int(100 * float(c0 / total))
        c1 = 100 - c0
    return c0 * 101 + c1


def _likelihood_fields(c0, c1):
    """(GT, PL, GQ, QUAL) strings for already-rescaled counts (cuteSV_genotype.py:45-56)."""
    g00 = np.float64(pow((1 - ERR), c0) * pow(ERR, c1) * (1 - PRIOR) / 2)
    g11 = np.float64(pow(ERR, c0) * pow((1 - ERR), c1) * (1 - PRIOR) / 2)
    g01 = np.float64(pow(0.5, c0 + c1) * PRIOR)
    logs = np.array([log10(g00), log10(g01), log10(g11)])
    top = max(logs)
    lse = top + log10(sum(pow(10.0, x - top) for x in logs))
    norm = list(np.minimum(logs - lse, 0.0))
    p = [pow(10, x) for x in norm]
    pl = [int(np.around(-10 * log10(x))) for x in p]
    gq = [int(-10 * log10(p[1] + p[2])), int(-10 * log10(p[0] + p[2])), int(-10 * log10(p[0] + p[1]))]
    qual = abs(np.around(-10 * log10(p[0]), 1))
    return GENOTYPES[norm.index(max(norm))], "%d,%d,%d" % (pl[0], pl[1], pl[2]), str(max(gq)), str(qual)


def gl_fields(idx):
    """table lookup: gl_idx -> (GT, 'PL0,PL1,PL2', GQ, QUAL) as the strings of the reference's rows."""
    idx = int(idx)
    hit = _table.get(idx)
    if hit is None:
        hit = _SPECIAL.get(idx) or _likelihood_fields(idx // 101, idx % 101)
        _table[idx] = hit
    return hit


def fill_table():
    """every row of the table now (~0.15 s) instead of on first use: resolve.warm_up() calls it before the pool forks"""
    for c0 in range(101):
        for c1 in range(101 - c0):
            gl_fields(c0 * 101 + c1)
    for k in _SPECIAL:
        gl_fields(k)


def gl_table_blob(keys):
    """the strings of the table rows in `keys` as csv_rows_emit wants them: (blob, offsets[TABLE_SIZE + 1]) with
    'GT\tPL\tGQ\tQUAL' per listed row and empty entries elsewhere"""
    off = np.zeros(TABLE_SIZE + 1, np.int64)
    parts, lens = [], np.zeros(TABLE_SIZE, np.int64)
    for k in sorted(int(x) for x in keys):
        sfx = "\t".join(gl_fields(k)).encode()
        parts.append(sfx)
        lens[k] = len(sfx)
    np.cumsum(lens, out=off[1:])
    return b"".join(parts), off
