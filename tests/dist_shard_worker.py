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


def stage(store, params, tasks):
    hb = store.host_batch(tasks, params)
    res = oracle.cluster_batch(hb)
    per_seg = rows_mod.rows_by_segment(store, hb.segments, res)
    by = {t: per_seg[k] for k, t in enumerate(tasks)}
    out = {}
    for t in TYPES:
        for (tt, ch) in tasks:
            if tt == t:
                out.setdefault(ch, []).extend(by[(tt, ch)])
    return out


def digest(rows):
    return hashlib.sha256("\n".join("\t".join(r) for r in rows).encode()).hexdigest()


def main():
    rank, world = int(sys.argv[1]), int(sys.argv[2])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    st = synth.small_mixed(seed=11, genotype=True)
    p = Params.ont(genotype=True)
    mine = stage(st, p, shard.tasks_of_rank(st, rank, world, genotype=True))
    gathered = [None] * world
    dist.all_gather_object(gathered, {c: digest(r) for c, r in mine.items()})
    dist.barrier()
    if rank == 0:
        merged = shard.merge_results(gathered)
        full = {c: digest(r) for c, r in stage(st, p, st.tasks()).items()}
        assert merged == full, (merged, full)
        print("SHARD-OK", len(full))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
