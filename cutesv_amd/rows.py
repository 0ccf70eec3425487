"""Output side of the hot path: the structure-of-arrays result -> the reference's row lists.

Row layouts are the output contract of the stage (SURVEY.md §8b "Return rows"), consumed
unchanged by the reference's VCF emitter (cuteSV_genotype.py:263-458):
    DEL 13 fields / INS 14   cuteSV_resolveINDEL.py:207-219, 419-432 (no genotype), :464-478 (genotype)
    DUP 11                   cuteSV_resolveDUP.py:121-131, 170-180
    INV 12                   cuteSV_resolveINV.py:145-156, 240-251
    TRA 12                   cuteSV_resolveTRA.py:171-182 (genotype fields from call_gt, :258-309)
All numeric fields are `str`; read names are joined by ','.
"""
from . import _abi
from .genotype import gl_fields

_TRA_ALT = ("N[%s[", "N]%s]", "[%s[N", "]%s]N")      # cuteSV_resolveTRA.py:142-153


def _ci(v):
    return "-%d,%d" % (v, v)                          # cal_CIPOS, cuteSV_genotype.py:60


def materialise(store, segments, res, chrom_of_seg=None):
    """res: dict of trimmed result arrays (HostResult.trimmed()) -> list of (segment index, row).

    The call order of `res` is already the reference's emission order; rows come back in it.
    """
    n = len(res["bp1"])
    if n == 0:
        return []
    sup_off = res["support_off"]
    sup_sig = res["support_sig"]
    # one vectorised gather for all read names
    names = store.names.take(store.read_id[sup_sig]) if len(sup_sig) else []
    call_seg = res["call_seg"].tolist()
    bp1 = res["bp1"].tolist(); bp2 = res["bp2"].tolist(); support = res["support"].tolist()
    cipos = res["cipos"].tolist(); cilen = res["cilen"].tolist()
    seq_pick = res["seq_pick"].tolist(); aux = res["call_aux"].tolist()
    dr = res["dr"].tolist(); gl = res["gl_idx"].tolist()
    off = sup_off.tolist()
    seg_type = segments["svtype"].tolist()
    seg_chrom = segments["chrom"].tolist()
    seg_gt = segments["genotype"].tolist()
    out = []
    for c in range(n):
        k = call_seg[c]
        t = seg_type[k]
        chrom = store.chroms[seg_chrom[k]]
        reads = ",".join(names[off[c]:off[c + 1]])
        if seg_gt[k] and gl[c] >= 0:          # TRA: count_coverage may give up -> '.' fields (cuteSV_resolveTRA.py:276-281)
            gt, pl, gq, qual = gl_fields(gl[c])
            g = (str(dr[c]), gt, pl, gq, qual)
        else:
            g = (".", "./.", ".,.,.", ".", ".")
        if t == _abi.DEL:
            row = [chrom, "DEL", str(bp1[c]), str(-bp2[c]), str(support[c]), _ci(cipos[c]), _ci(cilen[c]),
                   g[0], g[1], g[2], g[3], g[4], reads]
        elif t == _abi.INS:
            row = [chrom, "INS", str(bp1[c]), str(bp2[c]), str(support[c]), _ci(cipos[c]), _ci(cilen[c]),
                   g[0], g[1], g[2], g[3], g[4], reads, store.sequence(seq_pick[c])[:bp2[c]]]
        elif t == _abi.DUP:
            row = [chrom, "DUP", str(bp1[c]), str(bp2[c] - bp1[c]), str(support[c]), g[0], g[1], g[2], g[3], g[4], reads]
        elif t == _abi.INV:
            row = [chrom, "INV", str(bp1[c]), str(bp2[c] - bp1[c]), str(support[c]), g[0], g[1],
                   store.strands[aux[c]], g[2], g[3], g[4], reads]
        else:
            code = aux[c] & 7
            chr2 = store.chroms[aux[c] >> 3]
            mate = bp2[c] + (1 if code in (0, 2) else 0)              # types A/C, cuteSV_resolveTRA.py:140
            row = [chrom, _TRA_ALT[code] % ("%s:%s" % (chr2, mate)), str(bp1[c]), chr2, str(bp2[c]), str(support[c]),
                   g[0], g[1], g[2], g[3], g[4], reads]
        out.append((k, row))
    return out
