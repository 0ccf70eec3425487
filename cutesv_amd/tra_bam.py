"""BAM-faithful genotyping of TRA calls on the host (the default of the `run_tra` drop-in).

The reference genotypes translocations by re-opening the BAM (call_gt, cuteSV_resolveTRA.py:258-309; count_coverage,
cuteSV_genotype.py:72-93): every alignment `fetch` returns counts towards `iteration`, including the secondary and
low-mapq ones that never reach the reads table (main script :711-733).  The GPU variant (k_genotype_tra) walks the reads
table instead and is therefore only identical when the windows hold no such alignment; it stays an explicit opt-in
(CUTESV_AMD_TRA_GT=reads_table, or Params.genotype_tra in the batched stage).  This module is the faithful path: the
same loop as the reference, over pysam, for the few hundred BND calls of a genome.  It needs pysam, like cuteSV itself.
"""
from .genotype import gl_fields, gl_index


def threshold_ref_count(num):                       # cuteSV_genotype.py:62-70
    if num <= 2:
        return 20 * num
    if num <= 5:
        return 9 * num
    if num <= 15:
        return 7 * num
    return 5 * num


def count_coverage(chrom, s, e, bam, names, up_bound, itround):
    """cuteSV_genotype.py:72-93; returns the status 0 / 1 / -1"""
    iteration = 0
    primary_num = 0
    for aln in bam.fetch(chrom, s, e):
        iteration += 1
        if aln.flag not in (0, 16):
            continue
        primary_num += 1
        if aln.reference_start < s and aln.reference_end > e:
            names.add(aln.query_name)
            if len(names) >= up_bound:
                return 1
        if iteration >= itround:
            return 1 if float(primary_num / iteration) <= 0.2 else -1
    return 0


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
