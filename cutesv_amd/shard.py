"""Sharding of the clustering stage over the GPUs of a node.

(chromosome, type) segments are independent — the reference already treats them as separate pool
tasks (main script :1116-1189) and TRA signatures of a (chr1, chr2) pair live entirely in chr1's block
(:801) — so there is NO data-path collective: each rank (one process per GPU) clusters its own
chromosomes and the host concatenates the per-chromosome row lists exactly as main script :1191-1197
does.  All types of one chromosome stay on one rank so its reads block is shipped and scanned once.
"""
from .columns import TYPES


def chromosome_cost(store, chrom, genotype=False):
    n = sum(e - b for (t, c), (b, e) in store.seg_index.items() if c == chrom)
    if genotype and store.reads_off is not None:
        i = store.chroms.index(chrom)
        n += int(store.reads_off[i + 1] - store.reads_off[i])
    return n


def assign(store, world_size, genotype=False):
    """Longest-processing-time-first assignment of chromosomes to ranks -> list (per rank) of chromosomes.
    Deterministic: every rank computes the same table without communicating."""
    chroms = sorted({c for (_, c) in store.seg_index}, key=lambda c: (-chromosome_cost(store, c, genotype), c))
    load = [0] * world_size
    out = [[] for _ in range(world_size)]
    for c in chroms:
        r = min(range(world_size), key=lambda i: (load[i], i))
        out[r].append(c)
        load[r] += chromosome_cost(store, c, genotype)
    return out


def tasks_of_rank(store, rank, world_size, genotype=False, types=TYPES):
    mine = set(assign(store, world_size, genotype)[rank])
    return [(t, c) for (t, c) in store.tasks(types=types) if c in mine]


def merge_results(per_rank):
    """{chr: rows} dicts of all ranks -> one dict (chromosomes are disjoint across ranks)."""
    out = {}
    for d in per_rank:
        for c, rows in d.items():
            assert c not in out, "chromosome %s on two ranks" % c
            out[c] = rows
    return out
