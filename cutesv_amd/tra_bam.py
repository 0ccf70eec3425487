"""BAM-faithful genotyping of TRA calls on the host (the default of the `run_tra` drop-in).

The reference genotypes translocations by re-opening the BAM (call_gt, cuteSV_resolveTRA.py:258-309; count_coverage,
cuteSV_genotype.py:72-93): every alignment `fetch` returns counts towards `iteration`, including the secondary and
low-mapq ones that never reach the reads table (main script :711-733).  The GPU variant (k_genotype_tra) walks the reads
table instead and is therefore only identical when the windows hold no such alignment; it stays an explicit opt-in
(CUTESV_AMD_TRA_GT=reads_table, or Params.genotype_tra in the batched stage).  This module is the faithful path: the
same decision as the reference's loop, taken over chunks of `fetch`'s alignments with array gates (window_status), for the few
hundred BND calls of a genome.  It needs pysam, like cuteSV itself.
"""
from .genotype import gl_fields, gl_index


import itertools

import numpy as np

_UP_BOUND_STEPS = ((2, 20), (5, 9), (15, 7))          # support <= 2 / 5 / 15 -> x20 / x9 / x7, beyond x5 (cuteSV_genotype.py:62-70)


def threshold_ref_count(num):
    for limit, factor in _UP_BOUND_STEPS:
        if num <= limit:
            return factor * num
    return 5 * num


def window_status(alignments, s, e, names, up_bound, itround, chunk=2048):
    """What count_coverage (cuteSV_genotype.py:72-93) decides for one window, evaluated a chunk of alignments at a time with
    array gates instead of a per-alignment loop.  `alignments` yields pysam-like records in fetch order; `names` (a set) is
    extended with the spanning primary names up to the alignment at which the reference stops.  Returns 0 / 1 / -1.

    Per alignment i (1-based position k in the stream): primary = flag is 0 or 16; spanning = primary, starts before s and
    ends after e.  The walk stops at the first alignment where either (A) it is spanning and the number of distinct
    spanning names seen so far reaches up_bound -> 1, or (B) it is primary and k >= itround -> 1 when at most a fifth of
    the k alignments were primary, else -1.  (A is tested before B on the same alignment.)"""
    seen = primary_seen = 0
    it = iter(alignments)
    while True:
        block = list(itertools.islice(it, chunk))
        if not block:
            return 0
        flag = np.fromiter((a.flag for a in block), np.int64, len(block))
        primary = (flag == 0) | (flag == 16)
        # coordinates of the primary records only: fetch() also yields records without an end (an unmapped mate placed at its
        # partner's position has reference_end None), and the reference never looks at theirs either (`flag not in (0, 16)`)
        start = np.zeros(len(block), np.int64)
        end = np.zeros(len(block), np.int64)
        for i in np.flatnonzero(primary).tolist():
            start[i] = block[i].reference_start or 0
            end[i] = block[i].reference_end or 0
        spanning = primary & (start < s) & (end > e)
        k = seen + 1 + np.arange(len(block))                                # position in the stream
        n_primary = primary_seen + np.cumsum(primary)
        # distinct spanning names so far: a name counts where it first appears (and was not in `names` before the window)
        fresh = np.zeros(len(block), bool)
        local = set()
        for i in np.flatnonzero(spanning).tolist():                         # (only the spanning records: a handful to a few dozen)
            q = block[i].query_name
            if q not in names and q not in local:
                local.add(q)
                fresh[i] = True
        n_names = len(names) + np.cumsum(fresh)
        stop_a = spanning & (n_names >= up_bound)
        stop_b = primary & (k >= itround)
        stops = np.flatnonzero(stop_a | stop_b)
        last = int(stops[0]) if len(stops) else len(block) - 1
        names.update(block[i].query_name for i in np.flatnonzero(spanning[:last + 1]).tolist())
        if len(stops):
            if stop_a[last]:
                return 1
            return 1 if float(n_primary[last] / k[last]) <= 0.2 else -1
        seen += len(block)
        primary_seen = int(n_primary[-1])


def count_coverage(chrom, s, e, bam, names, up_bound, itround):
    return window_status(bam.fetch(chrom, s, e), s, e, names, up_bound, itround)


def call_gt(bam, pos_1, pos_2, chr_1, chr_2, read_ids, max_cluster_bias, gt_round):
    """cuteSV_resolveTRA.py:258-309 on an open pysam.AlignmentFile -> (DV, DR, GT, PL, GQ, QUAL) as row strings"""
    names = set()
    reads = set(read_ids)
    up_bound = threshold_ref_count(len(reads))
    s = max(int(pos_1) - max_cluster_bias, 0)
    e = min(int(pos_1) + max_cluster_bias, bam.get_reference_length(chr_1))
    status = count_coverage(chr_1, s, e, bam, names, up_bound, gt_round)
    if status == -1:
        return str(len(reads)), ".", "./.", ".,.,.", ".", "."
    if status == 0:
        s = max(int(pos_2) - max_cluster_bias, 0)
        e = min(int(pos_2) + max_cluster_bias, bam.get_reference_length(chr_2))
        count_coverage(chr_2, s, e, bam, names, up_bound, gt_round)           # (its status is not looked at, :290-299)
    dr = sum(1 for q in names if q not in reads)
    gt, pl, gq, qual = gl_fields(gl_index(dr, len(reads)))
    return str(len(reads)), str(dr), gt, pl, gq, qual


def genotype_rows(rows, bam_path, max_cluster_bias, gt_round):
    """TRA rows of the clustering stage (cuteSV_resolveTRA.py:171-182; genotype fields '.') -> the same rows with the
    fields call_gt computes from the BAM.  Row layout: [chr1, ALT, pos1, chr2, pos2, support, DR, GT, PL, GQ, QUAL, reads]."""
    if not rows:
        return rows
    import pysam                                     # as the reference does inside call_gt (:259)
    bam = pysam.AlignmentFile(bam_path)
    try:
        out = []
        for r in rows:
            dv, dr, gt, pl, gq, qual = call_gt(bam, r[2], r[4], r[0], r[3], r[11].split(","), max_cluster_bias, gt_round)
            out.append([r[0], r[1], r[2], r[3], r[4], dv, dr, gt, pl, gq, qual, r[11]])
        return out
    finally:
        bam.close()
