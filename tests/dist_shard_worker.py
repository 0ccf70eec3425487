"""Worker of test_two_rank_gloo_sharded_stage_matches_single_rank (CPU, gloo)."""
import hashlib
import os
import sys

import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from cutesv_amd import synth, shard, rows as rows_mod          # noqa: E402
from cutesv_amd.columns import Params, TYPES                   # noqa: E402
from oracle import oracle                                       # noqa: E402


def stage(store, params, units):
    """one rank's units (whole chromosomes or pieces of one, shard.plan) through the engine -> {(type, chrom, piece): rows}"""
    hb, keys = shard.host_batch(store, params, units)
    res = oracle.cluster_batch(hb)
    per_seg = rows_mod.rows_by_segment(store, hb.segments, res)
    return {k: per_seg[i] for i, k in enumerate(keys)}


def digest(rows):
    return hashlib.sha256("\n".join("\t".join(r) for r in rows).encode()).hexdigest()


def main():
    rank, world = int(sys.argv[1]), int(sys.argv[2])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    st = synth.small_mixed(seed=11, genotype=True)
    p = Params.ont(genotype=True)
    # max_imbalance 0 forces the largest chromosomes to be cut into pieces: the piece path is exercised at world size 2 too
    plan = shard.plan(st, world, p, genotype=True, max_imbalance=0.0)
    assert any(u[2] > 1 for us in plan for u in us), "no chromosome was cut"
    mine = stage(st, p, plan[rank])
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)                 # (rows are small here; bench.py sends digests)
    dist.barrier()
    if rank == 0:
        merged = {c: digest(r) for c, r in shard.merge_rows(gathered).items()}
        full = {c: digest(r) for c, r in shard.merge_rows([stage(st, p, shard.plan(st, 1, p, genotype=True)[0])]).items()}
        assert merged == full, (merged, full)
        print("SHARD-OK", len(full))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
