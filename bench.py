#!/usr/bin/env python3
"""bench.py — signatures clustered per second on MI355X (BASELINE.json's metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg3|cfg2|cfg4|cfg5|rebuild|extract] [--scale S] [--mode replica|shard]

A "step" is one pass of the whole hot path (chain -> refine -> order [-> reads order -> genotype]) over one synthetic
signature batch that is already resident in HBM when the timed region starts, AND the delivery of its result - the calls'
structure-of-arrays and the support lists - into the caller's host memory: the region of SURVEY.md 8(d) ("flat arrays ->
rows SoA in host RAM") with the inputs resident as the bench contract asks.  `value` = signatures / that step.  The same
line carries the two neighbouring regions: `kernel_only` (the launch sequence alone, results left in HBM: what r01-r03
reported as value) and `boundary.one_shot_call_ms` (page-locked host columns -> host SoA, PCIe both ways).
At N = 1 the workload is BASELINE config 3 (synthetic HG002-shaped ONT 30x, ~2.8 M INS+DEL signatures, ONT preset): the
metric is quoted "on 30x WGS", and this is the 30x whole-genome configuration that names one MI355X; the other configs
(cfg2, cfg4, cfg5) are measured in the same run in compact form (`other_workloads`).  The reads table of the genotyping
workloads (cfg4 / cfg5) is fed in the order cuteSV's extraction leaves it (synth.extraction_order), so the device-side
reads ordering is inside every step.

N > 1: one process per GPU.  Launched by torch.distributed.run the ranks come from the environment; plain
`python bench.py --gpus N` spawns the N ranks itself (127.0.0.1 rendezvous).  The path shards with no data-path
collective; the only communication is the timing barrier / max-reduce (gloo, CPU tensors).
  --mode replica (default for cfg3)  every rank clusters its own genome of the workload's shape (seed + rank): "weak"
                            scaling; the line ALSO carries a `sharded` object: BASELINE config 4 (one HiFi genome with
                            --genotype) split over the N ranks, measured right after the replica loop.
  --mode shard              ONE genome (BASELINE configs 4 and 5: "chromosomes sharded over 8 MI355X"): the ranks split
                            its chromosomes with shard.plan; a step is the rank's whole csv_cluster_batch call
                            (H2D + kernels + D2H); rank 0 then merges the ranks' rows and checks them against the
                            unsharded run: "strong" scaling.

Prints ONE JSON line on rank 0.  Extra objects:
  roofline      dominant kernel (largest average HIP-event duration over the K steps of a second, event-instrumented
                pass on the library's own stream): algorithmic bytes (32 B per signature the launch
                processes + 64 B per call; genotype 21 B per read + 32 B per call) / duration vs the 8 TB/s HBM peak.
                `traffic` comes from the rocprofv3 PMC passes committed under profiles/ for the same command, or null.
                `cold` repeats the figure with the caches evicted before every step (csv_cache_flush): the columns of
                cfg3 fit the 256 MiB Infinity Cache, so the warm loop is a MALL number and the cold one an HBM number.
  boundary      the stage as a drop-in sees it: one_shot_call_ms = csv_cluster_batch from page-locked host columns
                (H2D + kernels + D2H), rows_ms = the native row builder, stage_wall_ms = flat columns in host RAM ->
                the reference's {chr: rows} (resolve.cluster_stage), stage_speedup = cpu_baseline wall / stage_wall.
  cpu_baseline  oracle/py_restatement.py (the reference's execution model: Python loops + numpy scalars in a
                multiprocessing pool at all host cores) timed on rank 0 at N = 1 on a bounded sample.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from cutesv_amd import synth, engine, _abi, rows as rows_mod, resolve, shard     # noqa: E402
from cutesv_amd.columns import Params                                            # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); the copy ceiling is measured per run (~5.3 TB/s)
PCIE_PEAK_GBS = 63.0           # PCIe Gen5 x16, spec (same guide)


def make_workload(name, scale, rank):
    if name == "cfg3":
        return synth.ont30(seed=20260103 + rank, scale=scale), Params.ont(), \
            "cfg3: synthetic HG002-shaped ONT 30x, INS+DEL, ONT preset (synth.ont30, scale %g)" % scale
    if name == "cfg2":
        sites = dict(np.load(os.path.join(ROOT, "tests", "golden", "sim_sites.npz")))
        return synth.sim_all_types(sites, seed=20260102 + rank), Params.ont(), "cfg2: all five simulation beds, whole genome, ONT preset"
    if name == "cfg4":
        st, _ = synth.extraction_order(synth.hifi30_gt(seed=20260104 + rank, scale=scale))
        return st, Params.hifi(genotype=True, min_support=3), \
            "cfg4: synthetic HiFi 30x with --genotype, reads in extraction order (synth.hifi30_gt, scale %g)" % scale
    if name == "cfg5":
        st, _ = synth.extraction_order(synth.ont90_all(seed=20260105 + rank, scale=scale))
        return st, Params.ont(genotype=True), \
            "cfg5: synthetic ONT 90x, all five types, --genotype for INS/DEL/DUP/INV, reads in extraction order (synth.ont90_all, scale %g)" % scale
    raise SystemExit("unknown workload " + name)


def kernel_units(store, hb, res, stats, per_sig_step):
    """Algorithmic bytes per launch of every kernel: SURVEY.md 8(d)'s per-unit figures PARTITIONED over the kernels, every byte
    charged once, to the kernel that first needs it:
      clustering 32 B per signature = a 8 -> the chain kernel; b 8 + read_id 4 + aux 4 -> the refine kernel that handles the
      signature's cluster (only clusters that pass the size gate are ever refined: signatures of the others are never charged
      to a refine kernel, and aux only counts for INS); cluster_id 4 + allele_id 4 -> k_chain_ids, which only runs for
      CSV_IN_PER_SIG batches; INV chain rows also read b and aux (12 B), TRA rows aux (4 B);
      64 B per emitted call + 8 B per support entry -> k_emit;
      genotype: 21 B per read + 32 B per genotyped call -> the genotype stage (reads order + prefix max + stabbing queries).
    The whole-step figure is the contract's: 24 B per signature in (+ 8 B out when the per-signature outputs are produced in
    the timed loop) + 64 B per call + the genotype bytes."""
    t = res.trimmed()
    cid = t["cluster_id"]
    cid = cid[cid >= 0]
    W = int(cid.shape[0])
    sizes = np.bincount(cid, minlength=t["n_clusters"])
    first = np.flatnonzero(np.r_[True, cid[1:] != cid[:-1]])
    segs = hb.segments
    seglen = segs["sig_end"] - segs["sig_begin"]
    woff = np.r_[0, np.cumsum(seglen)]
    seg_of_cluster = np.searchsorted(woff, first, side="right") - 1
    valid = sizes >= segs["read_count"][seg_of_cluster]
    ctype = segs["svtype"][seg_of_cluster]
    indel = ctype <= _abi.INS
    ins = ctype == _abi.INS

    def refine_bytes(mask):
        return int(16 * sizes[valid & mask].sum() + 4 * sizes[valid & mask & ins].sum())
    n_iw = int(sizes[valid & (sizes <= 64) & indel].sum())             # k_refine_indel_wave
    n_pw = int(sizes[valid & (sizes <= 64) & ~indel].sum())            # k_refine<64,64>
    n_mid = int(sizes[valid & (sizes > 64) & (sizes <= 256)].sum())    # k_refine<64,256>
    n_blk = int(sizes[valid & (sizes > 256)].sum())                    # k_refine<256,2048>
    calls, sup = res.n_calls, res.n_support
    R = 0 if hb.r_start is None else int(hb.r_start.shape[0])
    gt_calls = int((t["gl_idx"] >= 0).sum())
    w_inv = int(seglen[segs["svtype"] == _abi.INV].sum())
    w_tra = int(seglen[segs["svtype"] == _abi.TRA].sum())
    b = {
        "k_chain_count": 8 * W + 12 * w_inv + 4 * w_tra,
        "k_chain_apply": 16 * int(stats.n_work_wave + stats.n_work_block),       # one item record per gated cluster (bookkeeping, not in the contract)
        "k_refine_indel_wave": refine_bytes((sizes <= 64) & indel), "k_refine_wave": refine_bytes((sizes <= 64) & ~indel),
        "k_refine_mid": refine_bytes((sizes > 64) & (sizes <= 256)), "k_refine_block": refine_bytes(sizes > 256),
        "k_emit": 64 * calls + 8 * sup, "k_items_scan": 8 * int(stats.n_work_wave + stats.n_work_block),
        # genotyping is judged as ONE stage (reads ordering + prefix max over the reads table + the per-call stabbing
        # queries): 21 B per read + 32 B per genotyped call over the summed duration of its kernels (see main())
        "genotype_stage": 21 * R + 32 * gt_calls,
    }
    total = (32 if per_sig_step else 24) * W + 64 * calls + (21 * R + 32 * gt_calls if R else 0)
    return b, total, dict(signatures=W, sig_in_gated_clusters=n_iw + n_pw + n_mid + n_blk, sig_refined_indel_wave=n_iw, sig_refined_pair_wave=n_pw,
                          sig_refined_mid=n_mid, sig_refined_block=n_blk, calls=calls, supports=sup,
                          reads=R, genotyped_calls=gt_calls, clusters=int(t["n_clusters"]), gated_clusters=int(valid.sum()))


def bench_rebuild(a):
    """--workload rebuild: SURVEY.md 8f row 2 (main script :750-857): the cfg3 genome's rows in random order + 5 % duplicates
    -> the order contract, on the device.  Roofline (composite-key sort, sort.hip.h): pack reads the 28-byte row and writes a
    16-byte element; a radix pass reads the elements twice (histogram, scatter) and writes them once = 48 B per row and pass;
    the tail reads the elements and writes the six output columns (16 + 32 B per row)."""
    from cutesv_amd import rebuild
    store, params, _ = make_workload("cfg3", a.scale, 0)
    per = synth.unsorted_rows(store, seed=1, dup_frac=0.05)
    n_in = sum(len(d["a"]) for d in per.values())
    ctx = engine.Context(0)
    dev_ms, wall_host, wall_dev, info = [], [], [], None
    def seg_of(t, ci, beg, end):
        rec = store.segment(t, store.chroms[ci], params).copy()
        rec["sig_begin"], rec["sig_end"] = beg, end
        return rec
    # (one loop per form: the device form keeps page-locked staging and result arrays in the context, which the through-host form's
    # pageable copies of 80 MB each way were seen to suffer from when the two alternated)
    for _ in range(a.warmup + min(a.steps, 10)):
        t0 = time.perf_counter()
        got, info = rebuild.store_from_unsorted(ctx, store.chroms, per)
        hb = got.host_batch(got.tasks(), params)
        r1 = ctx.cluster_batch(hb)
        wall_host.append(time.perf_counter() - t0)
        dev_ms.append(info["ms_device"])
    for _ in range(a.warmup + min(a.steps, 10)):
        t0 = time.perf_counter()
        batch, tasks, src_row = rebuild.rebuild_to_device_batch(ctx, store.chroms, per, seg_of)
        r2 = ctx.cluster_batch(batch, reuse=True)         # (the context's recycled page-locked result arrays)
        wall_dev.append(time.perf_counter() - t0)
    same = all(np.array_equal(r1.trimmed()[k], r2.trimmed()[k]) for k in ("bp1", "bp2", "support", "cipos", "cilen", "call_seg"))
    ms = float(np.median(dev_ms[a.warmup:]))
    t0 = time.perf_counter()
    for t, d in per.items():
        np.lexsort((d["read_id"], d["b"], d["a"], d["chrom"]))
    t_np = time.perf_counter() - t0
    # the reference's own sort is Python's list.sort with a tuple key (main script :764-802): timed on a bounded sample
    smp = 200_000
    d = per["DEL"]
    tl = list(zip(d["chrom"][:smp].tolist(), d["a"][:smp].tolist(), d["b"][:smp].tolist(), d["read_id"][:smp].tolist()))
    t0 = time.perf_counter()
    tl.sort(key=lambda x: (x[0], x[1], x[2], x[3]))
    t_py = time.perf_counter() - t0
    # algorithmic bytes: the row in (28 B: seg 4 + a 8 + b 8 + read 4 + aux 4) and the row out (28 B).  The radix passes are
    # TRAFFIC, not algorithmic bytes: 16 B packed + per pass 16 read (histogram) + 16 read + 16 written (scatter)
    bytes_alg = n_in * 56
    bytes_traffic = n_in * (28 + 16 + 48 * info["n_passes"] + 16 + 32)
    out = {"metric": "signature rows rebuilt/sec (sort + de-duplication, main script :750-857)", "value": n_in / (ms * 1e-3), "unit": "rows/s", "n_gpus": 1,
           "steps": min(a.steps, 10), "warmup": a.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "int64 keys", "data": "synthetic",
           "config": {"workload": "rebuild: cfg3 rows in random order + 5 %% exact duplicates (%d rows in, %d out, %d radix passes)" % (n_in, got.n_sig, info["n_passes"])},
           "roofline": {"bound": "hbm", "kernel": "k_sort_* + k_rebuild_*", "achieved": bytes_alg / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": bytes_alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": bytes_traffic, "algorithmic_bytes": bytes_alg,
                        "traffic_note": "computed, not measured: pack 28 + 16 B, %d radix passes x 48 B, tail 16 + 32 B per row" % info["n_passes"],
                        "traffic_gbs": bytes_traffic / (ms * 1e-3) / 1e9},
           "cpu_baseline": {"value": smp / t_py, "unit": "rows/s", "cores": 1, "kind": "port",
                            "sample": "Python list.sort with the reference's tuple key on %d DEL rows (%.2f s): the reference's rebuild is this per type" % (smp, t_py)},
           "cpu_baseline_numpy": {"value": n_in / t_np, "unit": "rows/s", "cores": 1, "sample": "numpy lexsort of all rows, %.3f s" % t_np},
           "chain": {"rebuild_to_host_then_cluster_ms": float(np.median(wall_host[a.warmup:])) * 1e3,
                     "rebuild_on_device_then_cluster_ms": float(np.median(wall_dev[a.warmup:])) * 1e3,
                     "same_calls": bool(same),
                     "note": "wall clock of rebuild + csv_cluster_batch from unsorted host rows; second form: CSV_RB_KEEP_ON_DEVICE + CSV_IN_DEVICE_COLUMNS, page-locked staging and result arrays"}}
    emit(out)
    ctx.close()


def bench_extract(a):
    """--workload extract: SURVEY.md 8f row 4 (main script :606-681, :50-513): the CIGAR scan of 10^5 long reads and the
    split-read analysis of 10^5 reads with SA entries.  Roofline of the CIGAR scan: 4 B per CIGAR operation, read twice
    (count + emit)."""
    from cutesv_amd import extract
    from oracle import oracle
    n = int(100_000 * a.scale)
    off, cigar, start, use = synth.cigar_reads(n)
    enc = synth.split_reads(n)
    ctx = engine.Context(0)
    cg, sp = [], []
    for _ in range(a.warmup + min(a.steps, 10)):
        cg.append(extract.cigar_signatures(ctx, off, cigar, start, use)["ms_device"])
        sp.append(extract.split_signatures(ctx, enc)["ms_device"])
    ms_c, ms_s = float(np.median(cg[a.warmup:])), float(np.median(sp[a.warmup:]))
    got = extract.cigar_signatures(ctx, off, cigar, start, use)
    t0 = time.perf_counter(); want = oracle.cigar_signatures(off, cigar, start, use); t_c = time.perf_counter() - t0
    t0 = time.perf_counter(); want_s = oracle.split_signatures(enc); t_s = time.perf_counter() - t0
    got_s = extract.split_signatures(ctx, enc)
    parity = all(np.array_equal(got[k], want[k]) for k in ("ins_read", "ins_pos", "ins_len", "del_read", "del_pos", "del_len")) and \
        all(np.array_equal(got_s[k], want_s[k]) for k in ("kind", "read", "a", "b"))
    nops = int(off[-1])
    # the hand-off extraction -> rebuild: the signatures of 10 tasks through host memory (D2H of the scan's result, rows built
    # there, H2D into the rebuild, D2H of the sorted columns) and through the context's pool (nothing but src_row comes back)
    from cutesv_amd import rebuild
    n_task = 10
    rank = np.random.default_rng(5).permutation(n * n_task).astype(np.int32)
    major = np.zeros(2, np.uint8); nodedup = np.array([0, 1], np.uint8)

    def via_host():
        rows = {k: [] for k in ("seg", "a", "b", "read", "aux")}
        for k in range(n_task):
            g = extract.cigar_signatures(ctx, off, cigar, start, use)
            rows["seg"] += [np.ones(len(g["ins_pos"]), np.int32), np.zeros(len(g["del_pos"]), np.int32)]
            rows["a"] += [g["ins_pos"] + k, g["del_pos"] + k]; rows["b"] += [g["ins_len"], g["del_len"]]
            rows["read"] += [k * n + g["ins_read"], k * n + g["del_read"]]; rows["aux"] += [g["ins_len"], np.zeros(len(g["del_pos"]), np.int64)]
        cat = {k: np.concatenate(v) for k, v in rows.items()}
        return rebuild.rebuild_columns(ctx, cat["seg"], cat["a"], cat["b"], rank[cat["read"]], cat["aux"], major, nodedup)

    def via_pool():
        rebuild.pool_reset(ctx)
        for k in range(n_task):
            extract.cigar_signatures(ctx, off, cigar, start + k, use, pool=dict(seg_ins=1, seg_del=0, read_base=k * n), host_outputs=False)
        return rebuild.rebuild_pool(ctx, rank, major, nodedup, keep_on_device=True)
    # ... and the same with the reads' CIGAR words in page-locked memory (a driver decodes the BAM records into such a buffer):
    # the hand-off's floor is then the words themselves over PCIe (n_task x 73 MB at 55 GB/s)
    p_off, p_cigar = engine.pinned_copy(off), engine.pinned_copy(cigar)

    def via_pool_pinned():
        rebuild.pool_reset(ctx)
        for k in range(n_task):
            extract.cigar_signatures(ctx, p_off, p_cigar, start + k, use, pool=dict(seg_ins=1, seg_del=0, read_base=k * n), host_outputs=False)
        return rebuild.rebuild_pool(ctx, rank, major, nodedup, keep_on_device=True)
    chain = {"cigar_bytes_per_task": int(cigar.nbytes), "pcie_floor_ms": n_task * cigar.nbytes / 55e9 * 1e3}
    for name, fn in (("through_host_ms", via_host), ("on_device_ms", via_pool), ("on_device_pinned_cigars_ms", via_pool_pinned)):
        fn()
        t0 = time.perf_counter(); r = fn(); chain[name] = (time.perf_counter() - t0) * 1e3
        chain[name.replace("_ms", "_rows")] = int(r["n_out"])
    chain["note"] = "%d tasks of %d reads each: CIGAR scan -> rows -> rebuild (sort + de-duplication); second form: CSV_CG_TO_POOL + CSV_RB_FROM_POOL + CSV_RB_KEEP_ON_DEVICE" % (n_task, n)
    rebuild.pool_reset(ctx)
    out = {"metric": "reads scanned/sec (CIGAR scan of parse_read, main script :606-655)", "value": n / (ms_c * 1e-3), "unit": "reads/s", "n_gpus": 1,
           "steps": min(a.steps, 10), "warmup": a.warmup, "ms_per_step": ms_c, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "u32 CIGAR words", "data": "synthetic",
           "config": {"workload": "extract: %d long reads, %d CIGAR operations (synth.cigar_reads); %d reads with %d alignments for the split-read analysis" %
                                  (n, nops, n, int(enc["ent_off"][-1]))},
           "roofline": {"bound": "hbm", "kernel": "k_cigar_count + k_cigar_emit", "achieved": 8.0 * nops / (ms_c * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": 8.0 * nops / (ms_c * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes": 8 * nops},
           "cpu_baseline": {"value": n / t_c, "unit": "reads/s", "cores": 1, "kind": "port", "sample": "oracle/cutesv_oracle.c csvo_cigar_signatures, all %d reads, %.3f s" % (n, t_c)},
           "split_reads": {"ms_device": ms_s, "reads_per_s": n / (ms_s * 1e-3), "alignments": int(enc["ent_off"][-1]), "candidates": int(len(got_s["kind"])),
                           "us_per_read_vs_cigar_scan": ms_s / ms_c,
                           "cpu_c_oracle_reads_per_s": n / t_s},
           "signatures": {"ins": int(len(got["ins_pos"])), "del": int(len(got["del_pos"]))}, "chain_extract_rebuild": chain, "parity_vs_oracle": bool(parity)}
    emit(out)
    ctx.close()


def spawn_ranks(a):
    """`python bench.py --gpus N` without a launcher: start the N ranks and relay rank 0's line"""
    port = 29500 + (os.getpid() % 2000)
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rc = procs[0].returncode
    for p in procs[1:]:
        rc = p.wait() or rc
    sys.stdout.write(out.decode())
    sys.exit(rc)


def digest_rows(rows_by_chr):
    import hashlib
    return {c: hashlib.sha256("\n".join("\t".join(r) for r in rows).encode()).hexdigest() for c, rows in rows_by_chr.items()}


_REF = {}


def synthetic_reference(store):
    """A synthetic reference genome ON DISK for the VCF leg (a FASTA file with 60-base lines + its .fai, in /dev/shm when there is
    room): contigs as long as the workload's coordinates need.  The emitter reads REF / ALT bases out of the memory-mapped file
    (cutesv_amd/fasta.py Reference) exactly as it would out of a real hg38.fa.  Built once per process, removed at exit."""
    import atexit
    import shutil
    import tempfile
    from cutesv_amd.fasta import Reference
    lens = []
    for i, c in enumerate(store.chroms):
        need = 0
        for t in ("DEL", "INS", "DUP", "INV", "TRA"):
            if (t, c) in store.seg_index:
                b0, e0 = store.seg_index[(t, c)]
                if e0 > b0:
                    need = max(need, int(store.a[b0:e0].max()), int(store.b[b0:e0].max()) if t in ("DUP", "INV") else 0)
        if store.contig_len is not None:
            need = max(need, int(store.contig_len[i]))
        lens.append(need + 200_000)
    if store.chroms and "TRA" in {t for t, _ in store.seg_index}:       # mate positions live on other contigs
        m = max(lens)
        lens = [m] * len(lens)
    key = (tuple(store.chroms), tuple(lens))
    if key in _REF:
        return _REF[key]
    t0 = time.perf_counter()
    base = "/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > sum(lens) * 1.1 + (1 << 28) else None
    d = tempfile.mkdtemp(prefix="cutesv_amd_ref_", dir=base)
    atexit.register(shutil.rmtree, d, ignore_errors=True)
    path = os.path.join(d, "ref.fa")
    line = b"ACGTTGCAAGCTTAGCCATGGATCCGTAACGTTAGCATGCCGATTACGGCTAAGTCCATG\n"     # 60 bases
    assert len(line) == 61
    block = line * 17189                                                         # ~1 MiB
    fai = []
    with open(path, "wb") as f:
        off = 0
        for c, n in zip(store.chroms, lens):
            hdr = (">%s synthetic\n" % c).encode()
            f.write(hdr); off += len(hdr)
            full, rem = divmod(n, 60)
            fai.append("%s\t%d\t%d\t60\t61\n" % (c, n, off))
            k = full
            while k > 0:
                w = min(k, 17189)
                f.write(block[:61 * w]); k -= w
            if rem:
                f.write(line[:rem] + b"\n")
            off += full * 61 + (rem + 1 if rem else 0)
    with open(path + ".fai", "w") as f:
        f.writelines(fai)
    ref = Reference(path)
    _REF[key] = (ref, dict(path=path, bytes=os.path.getsize(path), contigs=len(lens), build_s=round(time.perf_counter() - t0, 2)))
    return _REF[key]


def vcf_leg(ctx, pstore, params, tasks):
    """native VCF record emit straight from the SoA (no Python rows): EVERY type of the workload, REF / ALT bases fetched from an
    on-disk reference through fasta.Reference (mmap + .fai) as generate_output does through pysam (GT:254-262, 297-309, 334,
    360-365, 431-448).  (r03 timed it with ignore_sequence and without the pair types.)"""
    from cutesv_amd import vcf as vcf_mod
    try:
        ref, ref_info = synthetic_reference(pstore)
        hb2 = pstore.host_batch(tasks, params)
        r3 = ctx.cluster_batch(hb2)
        kw = dict(min_size=params.min_size, max_size=params.max_size, genotype=params.genotype, ignore_sequence=False, as_view=True)
        tv = []
        text = b""
        for _ in range(5):
            t0 = time.perf_counter()
            text, _ = vcf_mod.emit_records(pstore, hb2.segments, r3, ref, **kw)
            tv.append(time.perf_counter() - t0)
        n_rec, n_bytes = int(np.count_nonzero(np.frombuffer(text, np.uint8) == 10)), len(text)      # (the view is this thread's buffer: read it now)
        # pinned columns -> VCF text: the boundary call + the native emitter, no Python rows in between
        slim = SLIM if hb2.a.dtype == np.int32 else {}      # (the emitter reads no support list unless --report_readid is on)
        ts = timed(lambda: vcf_mod.emit_records(pstore, hb2.segments, ctx.cluster_batch(hb2, reuse=True, **slim), ref, **kw), 5)
        tn = []
        for _ in range(3):
            t0 = time.perf_counter()
            vcf_mod.emit_records(pstore, hb2.segments, r3, ref, **dict(kw, ignore_sequence=True))      # (pair types look one base up regardless)
            tn.append(time.perf_counter() - t0)
        types = sorted({t for t, _ in tasks})
        return dict(ms=float(np.median(tv)) * 1e3, records=n_rec, bytes=n_bytes, threads=min(16, os.cpu_count() or 1),
                    stage_wall_vcf_ms=float(np.median(ts)) * 1e3, ignore_sequence=False, types=types, reference=ref_info,
                    ms_ignore_sequence=float(np.median(tn)) * 1e3)
    except Exception as e:          # noqa: BLE001  (never let the optional leg break the benchmark line)
        import traceback
        return dict(error=repr(e), trace=traceback.format_exc()[-600:])


def per_task_leg(ctx, store, params, tasks):
    """INTEGRATION.md mode 1: the reference's own task interface, one (chromosome, type) task per call from ITS files
    (<TYPE>.pickle at sigs_index offsets): the pickle -> columns (walked in C) + the boundary call + rows, for the largest task"""
    try:
        import pickle
        import tempfile
        ins_tasks = [(t, c) for (t, c) in tasks if t == "INS"] or tasks
        tt, tc = max(ins_tasks, key=lambda k: store.seg_index[k][1] - store.seg_index[k][0])
        b0, e0 = store.seg_index[(tt, tc)]
        nm = store.names.take(store.read_id[b0:e0])
        if tt == "INS":
            lst = [(int(store.a[i]), int(store.b[i]), nm[i - b0], store.sequence(i), "INS", tc) for i in range(b0, e0)]
        else:
            lst = [(int(store.a[i]), int(store.b[i]), nm[i - b0], tt, tc) for i in range(b0, e0)]
        with tempfile.TemporaryDirectory() as wd:
            wd += "/"
            with open(wd + tt + ".pickle", "wb") as f:
                pickle.dump(lst, f)
            idx = {t_: {} for t_ in ("DEL", "INS", "INV", "DUP", "TRA")}
            idx[tt][tc] = 0
            bias = params.max_cluster_bias_INS if tt == "INS" else params.max_cluster_bias_DEL
            ratio = params.diff_ratio_merging_INS if tt == "INS" else params.diff_ratio_merging_DEL
            args = (wd, tc, tt, params.min_support, ratio, bias, min(params.min_support, 5), "bam", False, params.gt_round, params.remain_reads_ratio, idx)
            fn = resolve.run_ins if tt == "INS" else resolve.run_del
            resolve._ctx = ctx
            ts_ = timed(lambda: fn(args), 4)
            t0 = time.perf_counter()
            with open(wd + tt + ".pickle", "rb") as f:
                pickle.load(f)
            t_unpickle = time.perf_counter() - t0
        return dict(task="%s chr%s" % (tt, tc), signatures=e0 - b0, ms=float(np.median(ts_)) * 1e3, unpickle_ms=t_unpickle * 1e3,
                    signatures_per_s=(e0 - b0) / float(np.median(ts_)),
                    note="resolve.run_ins(args) with the reference's argument tuple on its pickle layout: the pickle walked in C out of the mapped file "
                         "(_cols_native.pickle_table: no Python object per tuple) + csv_cluster_batch + rows; unpickle_ms = pickle.load of the same block alone")
    except Exception as e:           # noqa: BLE001  (optional leg)
        return dict(error=repr(e))


PARITY_FIELDS = ("call_seg", "call_cluster", "bp1", "bp2", "support", "cipos", "cilen", "search_pos", "seq_pick", "dr", "dv", "gl_idx",
                 "support_off", "support_sig", "cluster_id", "allele_id")
GT_PARTS = ("k_reads_order", "k_reads_gather", "k_reads_maxlen", "k_genotype")        # stage slots (HIP events)
GT_KERNELS = ("k_reads_runs", "k_reads_plan", "k_reads_gather", "k_reads_maxlen", "k_genotype")   # kernel names (PMC)


def timed(fn, reps):
    """wall time of fn() to the moment its result exists (the result is released after the clock stops: tearing down
    the previous call's 350 k row strings is the consumer's time, not the producer's)"""
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn()
        ts.append(time.perf_counter() - t0)
        del out
    return ts


def per_kernel_us(v, names):
    """A stage slot is the time between two hipEventRecord calls on the library's stream; the records themselves occupy the
    stream (a slot that launches nothing still reads 4-5 us).  The library records one slot that is empty BY CONSTRUCTION
    ("event_floor", after the last kernel); that reading is subtracted, so that kernel_us is the kernel's own duration (it
    then agrees with rocprofv3's begin-to-end average: profiles/).  Raw slots are kept in `_raw`."""
    raw = {names[i]: float(v[i]) * 1e3 for i in range(_abi.N_STAGES) if names[i]}
    gap = raw.pop("event_floor", 0.0)
    if gap <= 0.5:                                           # (a library without the explicit slot)
        gap = min([x for x in raw.values() if x > 0.5] or [0.0])
    d = {k: round(max(0.0, x - gap), 2) if x > 0.5 else 0.0 for k, x in raw.items()}
    d["genotype_stage"] = round(sum(d.get(k, 0.0) for k in GT_PARTS), 2)
    d["_event_floor"] = round(gap, 2)
    d["_raw"] = {k: round(x, 2) for k, x in raw.items() if x > 0.5}
    return d


def cpu_legs(name, store, params, tasks, hb, n_sig, procs, py_pool, full_pool):
    """CPU baselines of one workload (rank 0, N = 1; before any HIP state exists in the process: the pool forks).
    py_pool: oracle/py_restatement.py under a fork Pool (the reference's execution model); full_pool: on every task of the
    workload (else the segments of every 3rd chromosome when the workload has > 4 M signatures)."""
    from oracle import oracle, py_restatement as pr
    cpu = None
    if py_pool:
        sample = tasks if (full_pool or n_sig <= 4_000_000) else [t for i, t in enumerate(tasks) if (i % 24) % 3 == 0]
        tl = pr.tasks_from_store(store, params, sample)
        ns = sum(len(t[2]) for t in tl)
        t0 = time.perf_counter()
        r = pr.run_pool_forked(tl, procs)
        dt = time.perf_counter() - t0
        # what that wall is made of (r05 review: > 90 % of it was the pool itself): the same pool with no-op tasks; the pool at
        # one worker per task (no idle forks); the largest task alone, in this process (the compute critical path)
        fit = min(len(tl), procs)
        t0 = time.perf_counter()
        pr.run_pool_forked(tl, fit)
        dt_fit = time.perf_counter() - t0
        # (the no-op pools are forked from THIS process, whose size sets their cost - tens of seconds at 256 workers once the
        # 90x workload's tuple lists are in memory: measured for the workloads up to 4 M signatures, once for the others' fit)
        small = n_sig <= 4_000_000
        dt_noop = min(pr.pool_startup_seconds(procs, len(tl)) for _ in range(2 if name == "cfg3" else 1)) if small else None
        dt_noop_fit = min(pr.pool_startup_seconds(fit, len(tl)) for _ in range(2 if name == "cfg3" else 1))
        big = max(tl, key=lambda t: len(t[2]) + (len(t[3]) if t[3] else 0))
        t0 = time.perf_counter()
        pr.run_task(big)
        dt_crit = time.perf_counter() - t0
        cpu = dict(value=ns / dt, unit="signatures/s", cores=procs, kind="port", wall_s=dt, full_workload=len(sample) == len(tasks),
                   wall_s_pool_fit=dt_fit, pool_fit_workers=fit, pool_startup_s=dt_noop, pool_fit_startup_s=dt_noop_fit, critical_path_s=dt_crit,
                   critical_task="%s chr%s, %d signatures" % (big[0], big[1], len(big[2])),
                   sample="%d of %d (chr,type) tasks, %d signatures, %.2f s wall; oracle/py_restatement.py in a fork "
                          "Pool(%d): the reference's pool model (Python per-signature loop, numpy scalar calls); %.2f s at Pool(%d); "
                          "the pools alone (no-op tasks) %s / %.2f s; largest task alone %.3f s"
                          % (len(sample), len(tasks), ns, dt, procs, dt_fit, fit, "-" if dt_noop is None else "%.2f" % dt_noop, dt_noop_fit, dt_crit),
                   rows=sum(len(x[1]) for x in r))
        del tl, r
    t0 = time.perf_counter()
    ores = oracle.cluster_batch(hb, per_sig=True)
    dtc = time.perf_counter() - t0
    cpu_c = dict(value=n_sig / dtc, unit="signatures/s", cores=1, kind="port",
                 sample="full workload, oracle/cutesv_oracle.c single thread, %.3f s" % dtc)
    # the same C restatement over the (chr, type) tasks on all host cores, one task per call like the reference's pool:
    # the parallelism a task pool can reach is bounded by the task count and by the largest task (chr1 INS)
    nthr = min(len(tasks), procs)
    dtm, mt_calls = min((oracle.cluster_tasks_mt(store, tasks, params, nthr) for _ in range(3)), key=lambda x: x[0])
    cpu_c_mt = dict(value=n_sig / dtm, unit="signatures/s", cores=nthr, kind="port", wall_s=dtm, calls=mt_calls,
                    sample="full workload, oracle/cutesv_oracle.c, one (chr,type) task per call on %d threads (host has %d cores; "
                           "%d tasks), largest first, best of 3: %.4f s" % (nthr, os.cpu_count() or 1, len(tasks), dtm))
    return cpu, cpu_c, cpu_c_mt, ores


def stage_leg(name, store, params):
    """bench_stage.mode1_stage in a process of its own, never allowed to break the line.  A cuteSV main process is small when it
    forks its phase-3 pool; this one holds gigabytes of workloads and oracle results by then, and every fork copies its page
    tables (measured: the 32-worker cfg4 stage 322 ms from a fresh process, 915-1130 ms forked from here)."""
    import subprocess
    try:
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench_stage.py"), "--workload", name], stdout=subprocess.PIPE, stderr=None, timeout=900)
        lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
        if p.returncode != 0 or not lines:
            return dict(error="bench_stage.py exited with %d" % p.returncode)
        return json.loads(lines[-1])
    except Exception as e:          # noqa: BLE001
        import traceback
        return dict(error=repr(e), trace=traceback.format_exc()[-400:])


SLIM = dict(no_support=True, coord32=True, fields=("call_aux", "cipos", "cilen", "seq_pick", "dr", "gl_idx"))


def resident_loops(ctx, phb, steps, warmup, dist=None):
    """The resident timed loops of an uploaded batch.  Returns (seconds of `steps` steps WITH the full result - calls + support
    lists - delivered to page-locked host arrays each step, seconds of the launch sequences alone, the last delivered result,
    seconds of `steps` steps with the SLIM result delivered: what a run without --report_readid consumes, ABI v7)."""
    ctx.upload(phb, per_sig=False)
    # every step does the whole stage, the ordering / packing of the reads table included (the library would keep the
    # ordered table of an upload across runs: CSV_OPT_REUSE_READS_ORDER, measured separately)
    ctx.option(1, 0)
    ctx.run(); ctx.sync()
    probe = ctx.download()
    # (the full result: every call field + the support lists; coordinates as int32 when the columns are - CSV_OUT_COORD_I32: the same
    # numbers in 60 instead of 76 bytes per call)
    c32 = phb.a.dtype == np.int32
    res = ctx.result_buffers(cap_calls=probe.n_calls + 64, cap_support=probe.n_support + 64, coord32=c32)
    for _ in range(warmup):
        ctx.run(); ctx.download(into=res)
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.run()
        ctx.download(into=res)                               # k_publish into the page-locked arrays + ONE stream synchronisation
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    for _ in range(warmup):
        ctx.run()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.run()
    ctx.sync()
    dt_k = time.perf_counter() - t0
    dt_slim = None
    if phb.a.dtype == np.int32:
        slim = ctx.result_buffers(cap_calls=probe.n_calls + 64, cap_support=64, **SLIM)
        for _ in range(warmup):
            ctx.run(); ctx.download(into=slim)
        t0 = time.perf_counter()
        for _ in range(steps):
            ctx.run()
            ctx.download(into=slim)
        dt_slim = time.perf_counter() - t0
    # pipelined delivery (csv_batch_publish_async): run k's result crosses PCIe while run k + 1 computes, two result arenas on
    # the device, two sets of page-locked arrays on the host; every step's full result is in host memory when the clock stops
    def pipelined(bufs):
        for _ in range(max(2, warmup)):
            ctx.run(); ctx.publish_async(bufs[0]); ctx.publish_wait()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        ctx.run(); ctx.publish_async(bufs[0])
        for k in range(1, steps):
            ctx.run(); ctx.publish_async(bufs[k & 1]); ctx.publish_wait()
        ctx.publish_wait()
        if dist is not None:
            dist.barrier()
        return time.perf_counter() - t0
    res_b = ctx.result_buffers(cap_calls=probe.n_calls + 64, cap_support=probe.n_support + 64, coord32=c32)
    dt_pipe = pipelined([res, res_b])
    dt_pipe_slim = None
    if dt_slim is not None:
        slim_b = ctx.result_buffers(cap_calls=probe.n_calls + 64, cap_support=64, **SLIM)
        dt_pipe_slim = pipelined([slim, slim_b])
    return dt, dt_k, res, dt_slim, dt_pipe, dt_pipe_slim


def instrumented(ctx, phb, steps):
    """per-kernel HIP-event durations on the library's stream (per_sig on: kernel_units needs cluster_id); returns
    (mean stage ms with the per-signature outputs, the same without them, the run's stats, the downloaded result)"""
    names = engine.stage_names()
    ctx.upload(phb, per_sig=True)
    acc = np.zeros(_abi.N_STAGES)
    st = None
    for _ in range(steps):
        st = ctx.run(stats=True)
        acc += np.array(list(st.ms_stage))
    acc /= steps
    res = ctx.download(per_sig=True)
    ctx.upload(phb, per_sig=False)
    plain = np.zeros(_abi.N_STAGES)
    tot = 0.0
    n2 = max(3, min(steps, 10))
    for _ in range(n2):
        s2 = ctx.run(stats=True)
        plain += np.array(list(s2.ms_stage))
        tot += s2.ms_total
    plain /= n2
    return per_kernel_us(acc, names), per_kernel_us(plain, names), st, res, tot / n2 * 1e3


def dominant(per_kernel_nps, kbytes):
    """the longest kernel of the plain pass; kernels within 5 % of the longest count as tied and the one that moves the most
    algorithmic bytes is named (so that the choice does not flip between runs on a 0.1 us difference)"""
    cand = [n for n in per_kernel_nps if n in kbytes and kbytes[n] > 0 and per_kernel_nps[n] > 0 and n not in ("k_chain_apply", "k_items_scan")]
    longest = max(per_kernel_nps[n] for n in cand)
    return max((n for n in cand if per_kernel_nps[n] >= 0.95 * longest), key=lambda n: kbytes[n])


def traffic_of(workload, scale):
    """L2 <-> fabric bytes per launch from the rocprofv3 PMC passes of the SAME command committed under profiles/
    (scripts/refresh_profiles.sh; counters calibrated in profiles/r04_pmc_calibration.txt)"""
    tf = os.path.join(ROOT, "profiles", "traffic_%s.json" % workload)
    if not os.path.exists(tf) or scale != 1.0:
        return None
    with open(tf) as f:
        return {k: v for k, v in json.load(f).items() if k.startswith("k_") and "." not in k and "[" not in k}


def valu_issue_of(workload, scale, kernel, kernel_us, n_cu=256, clock_mhz=2400.0):
    """How busy the kernel keeps the vector ALUs, from the SQ counter passes of the SAME command committed under profiles/
    (scripts/profile_insts.sh -> profiles/insts_<workload>.json).  SQ_ACTIVE_INST_VALU counts, summed over all wavefronts, the
    quad-cycles (4 clocks) in which a wavefront's VALU instruction occupies its SIMD; busy = that / (SIMDs x the kernel's own
    duration).  The HBM roofline is what the bench contract asks for; THIS is the ceiling these integer kernels run against
    (DESIGN.md section 5): a fraction near 1 means only fewer instructions make the kernel faster."""
    f = os.path.join(ROOT, "profiles", "insts_%s.json" % workload)
    if not os.path.exists(f) or scale != 1.0 or not kernel_us:
        return None
    with open(f) as fh:
        r = json.load(fh).get(kernel)
    if not r or not r.get("SQ_ACTIVE_INST_VALU"):
        return None
    # busy share of the chip's SIMDs as a pure counter ratio: SQ_BUSY_CYCLES is summed over the 32 shader engines, so the
    # kernel lasted SQ_BUSY_CYCLES / 32 clocks IN THE COUNTER RUN; 4 * ACTIVE / (4 * n_cu * BUSY / 32) = 32 * ACTIVE / (n_cu * BUSY)
    # (scripts/rocprof_counters.py prints the same column; no clock is assumed - r04 priced the counter run's instructions
    # on the bench run's duration at an assumed 2.4 GHz)
    busy = r.get("SQ_BUSY_CYCLES")
    return {"kernel": kernel, "valu_insts_per_launch": r["SQ_INSTS_VALU"], "salu_insts_per_launch": r["SQ_INSTS_SALU"], "lds_insts_per_launch": r["SQ_INSTS_LDS"],
            "waves": r["SQ_WAVES"], "valu_active_quad_cycles": r["SQ_ACTIVE_INST_VALU"],
            "valu_busy_frac": (32.0 * r["SQ_ACTIVE_INST_VALU"] / (n_cu * busy)) if busy else None, "kernel_us": kernel_us,
            "kernel_us_in_the_counter_run_at_2p4_ghz": (busy / 32.0 / 2400.0) if busy else None,
            "wave_share_issuing_valu": r["SQ_ACTIVE_INST_VALU"] / r["SQ_WAVE_CYCLES"] if r.get("SQ_WAVE_CYCLES") else None,
            "wave_share_issue_stalled": r["SQ_WAIT_INST_ANY"] / r["SQ_WAVE_CYCLES"] if r.get("SQ_WAVE_CYCLES") else None,
            "source": "profiles/insts_%s.json (rocprofv3 --pmc, two passes)" % workload}


def compact_workload(ctx, name, a, cpu):
    """cfg2 / cfg4 / cfg5 in the N = 1 line, compact: the same two resident loops, the dominant kernel and its roofline
    fraction, parity of the int32-column device path against the C oracle at full size, the boundary call, the C all-threads
    baseline (and, where it was run, the Python pool on the FULL workload)."""
    store, params, wl_name, tasks, hb, n_sig = cpu["store"], cpu["params"], cpu["wl_name"], cpu["tasks"], cpu["hb"], cpu["n_sig"]
    pstore = store.pinned()
    phb = pstore.host_batch(tasks, params)
    steps = max(5, min(a.steps, 20))
    dt, dt_k, _, dt_s, dt_p, dt_ps = resident_loops(ctx, phb, steps, 3)
    pk_sig, pk, st, res, _ = instrumented(ctx, phb, 5)
    kbytes, total_bytes, units = kernel_units(store, hb, res, st, per_sig_step=False)
    if units["reads"]:
        pk["genotype_stage"] = round(sum(pk.get(k, 0.0) for k in GT_PARTS), 2)
    dom = dominant(pk, kbytes)
    w, g = cpu["ores"].trimmed(), res.trimmed()
    parity = all(np.array_equal(g[k], w[k]) for k in PARITY_FIELDS)
    t_one = timed(lambda: ctx.cluster_batch(phb, reuse=True), 4)
    t_one_slim = timed(lambda: ctx.cluster_batch(phb, reuse=True, **SLIM), 4)
    tr = traffic_of(name, 1.0)
    dom_traffic = None if tr is None else (sum(tr.get(k, 0) for k in GT_KERNELS) if dom == "genotype_stage" else tr.get(dom))
    ms, ms_k, one = dt_p / steps * 1e3, dt_k / steps * 1e3, float(np.min(t_one)) * 1e3
    out = {"workload": wl_name, "signatures": n_sig, "reads": units["reads"], "calls": units["calls"],
           "ms_per_step": ms, "value": n_sig / (ms * 1e-3), "serial_ms_per_step": dt / steps * 1e3,
           "pipelined_slim_ms_per_step": None if dt_ps is None else dt_ps / steps * 1e3, "kernel_only_ms_per_step": ms_k, "kernel_only_value": n_sig / (ms_k * 1e-3),
           "one_shot_call_ms": one, "one_shot_slim_ms": float(np.min(t_one_slim)) * 1e3,
           "slim_ms_per_step": None if dt_s is None else dt_s / steps * 1e3,
           "dominant_kernel": {"kernel": dom, "us": pk[dom], "algorithmic_bytes": kbytes[dom],
                               "frac": round(kbytes[dom] / (pk[dom] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4), "traffic": dom_traffic},
           "pipeline": {"algorithmic_bytes": total_bytes, "frac": round(total_bytes / (ms_k * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
           "kernel_us": {k: v for k, v in pk.items() if isinstance(v, float) and v > 0.3 and not k.startswith("_")},
           "parity_vs_oracle": bool(parity), "parity_columns": "int32 (SigStore.pinned())",
           "cpu_baseline_c_mt": {k: cpu["cpu_c_mt"][k] for k in ("value", "cores", "wall_s")},
           "one_shot_speedup_vs_c_mt": cpu["cpu_c_mt"]["wall_s"] * 1e3 / one}
    if name == "cfg5":
        out["vcf_emit_native"] = vcf_leg(ctx, pstore, params, tasks)
    if cpu.get("mode1") is not None:
        out["mode1_stage"] = cpu["mode1"]
    if cpu["cpu"] is not None:
        out["cpu_baseline"] = {k: cpu["cpu"][k] for k in ("value", "cores", "wall_s", "full_workload", "sample", "wall_s_pool_fit", "pool_fit_workers",
                                                          "pool_startup_s", "pool_fit_startup_s", "critical_path_s") if k in cpu["cpu"]}
        out["step_speedup_vs_cpu_baseline"] = cpu["cpu"]["wall_s"] * 1e3 / ms if cpu["cpu"]["full_workload"] else None
        out["one_shot_speedup_vs_cpu_baseline"] = cpu["cpu"]["wall_s"] * 1e3 / one if cpu["cpu"]["full_workload"] else None
    return out


def measure_sharded(ctx, dist, rank, world, workload, scale, steps, warmup):
    """ONE genome over `world` ranks (BASELINE configs 4 / 5): every rank computes the same plan without communicating, a step
    is the rank's whole boundary call (H2D + kernels + D2H); rank 0 merges the ranks' rows exactly as main_ctrl concatenates
    task results and compares them with the unsharded run.  Returns (dict on rank 0 / None, seconds, signatures of the genome)."""
    from cutesv_amd.columns import SigStore
    shared = "/dev/shm/cutesv_amd_bench_%s_%s" % (os.environ.get("MASTER_PORT", "0"), workload)
    if world > 1:
        if rank == 0:
            store, params, wl_name = make_workload(workload, scale, 0)
            store.save(shared)
        dist.barrier()
        if rank != 0:
            _, params, wl_name = make_workload(workload, 0.001, 0)
            store = SigStore.load(shared)
        dist.barrier()
        if rank == 0:
            import shutil
            shutil.rmtree(shared, ignore_errors=True)          # (the mappings stay valid)
    else:
        store, params, wl_name = make_workload(workload, scale, 0)
    all_tasks = store.tasks()
    plan = shard.plan(store, world, params, genotype=params.genotype)
    units = plan[rank]
    pstore = store.pinned()
    phb, unit_keys = shard.host_batch(pstore, params, units, pin=engine.pinned_copy)
    n_sig = int((phb.segments["sig_end"] - phb.segments["sig_begin"]).sum())
    for _ in range(warmup):
        ctx.cluster_batch(phb, reuse=True)
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.cluster_batch(phb, reuse=True)
    t_rank = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    loads = [n_sig]
    total_sig = n_sig
    if dist is not None:
        import torch
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt[0])
        gl = [None] * world
        dist.all_gather_object(gl, (n_sig, t_rank / steps * 1e3, 0 if phb.r_start is None else int(phb.r_start.shape[0])))
        loads = gl
        total_sig = sum(x[0] for x in gl)
    per_seg = rows_mod.rows_by_segment(pstore, phb.segments, ctx.cluster_batch(phb))
    mine = {k: per_seg[i] for i, k in enumerate(unit_keys)}
    gathered = [mine]
    if dist is not None:
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
    out = None
    if rank == 0:
        merged = digest_rows(shard.merge_rows(gathered))
        full = digest_rows(resolve.cluster_stage(pstore, params, tasks=all_tasks, ctx=ctx))
        ok = merged == full
        assert ok, "sharded rows differ from the unsharded run"
        ms = dt / steps * 1e3
        out = {"workload": wl_name, "mode": "one genome split over the ranks: chromosomes longest-first, the largest cut at gaps wider than "
                                            "max_cluster_bias until the heaviest rank is within 3 % of the mean; no collective; a step is the rank's "
                                            "whole boundary call (H2D + kernels + D2H)",
               "ranks": world, "signatures_total": total_sig, "ms_per_step": ms, "value": total_sig / (ms * 1e-3), "scaling": "strong",
               "shard_merge_equals_unsharded": bool(ok), "pieces": sum(len(u) for u in plan),
               "per_rank": [{"signatures": x[0], "ms_per_step": round(x[1], 4), "reads": x[2]} for x in loads] if dist is not None else None,
               "signature_load_max_over_mean": (max(x[0] for x in loads) / (total_sig / world)) if dist is not None else 1.0}
    tasks = list(dict.fromkeys((t, c) for (t, c, _) in unit_keys))
    hb_plain, _ = shard.host_batch(store, params, units)          # (the same pieces from the pageable store: the one-shot-pageable leg, kernel_units)
    return out, dt, total_sig, store, params, wl_name, pstore, phb, tasks, hb_plain


_OUT_FD = None


def quiet_stdout():
    """Everything a library writes to file descriptor 1 from here on goes to stderr (gloo announces "[Gloo] Rank 0 is connected
    to N peer ranks" on stdout); the benchmark line itself is written to the real stdout by emit()."""
    global _OUT_FD
    if _OUT_FD is None:
        sys.stdout.flush()
        _OUT_FD = os.dup(1)
        os.dup2(2, 1)


LINE_LIMIT = 7600        # bytes: the driver keeps an 8 000-byte tail of stdout (BENCH_r04.json): the whole line must fit it
FULL_LINE = False         # --full: print every object (the scripts under scripts/ read kernel_us_cold, traffic_per_kernel, ...)


def _num(x, nd=4):
    return round(x, nd) if isinstance(x, float) else x


def compact(out):
    """The line the driver parses: the contract's keys in full, everything else as short named numbers.  The full objects go
    to stderr and to gpurun_out/bench_detail_<workload>.json (`detail`)."""
    if "roofline" not in out or "config" not in out or not isinstance(out.get("roofline"), dict):
        return out
    r = out["roofline"]
    c = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data") if k in out}
    c["config"] = {k: (v[:200] if isinstance(v, str) else v) for k, v in out["config"].items()}
    keep = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes", "kernel_us", "kernel_us_net", "boundary_us", "copy_ceiling", "frac_of_copy_ceiling")
    c["roofline"] = {k: _num(r.get(k)) for k in keep if k in r}
    if isinstance(r.get("valu_issue"), dict):
        c["roofline"]["valu_busy_frac"] = _num(r["valu_issue"].get("valu_busy_frac"))
        c["roofline"]["valu_insts_per_wave"] = _num(r["valu_issue"]["valu_insts_per_launch"] / max(1, r["valu_issue"]["waves"]), 1)
    if isinstance(r.get("cold"), dict):
        c["roofline"]["cold_frac"] = _num(r["cold"].get("frac"))
    if isinstance(out.get("roofline_pipeline"), dict):
        c["roofline"]["pipeline_frac"] = _num(out["roofline_pipeline"].get("frac"))
    h = out.get("host_to_host")
    if isinstance(h, dict):
        c["roofline"]["pcie"] = {k: _num(v) for k, v in h["pcie"].items()}
        c["host_to_host"] = {k: _num(h.get(k)) for k in ("gate_first", "delta16", "ms", "value", "slim_ms", "slim_value", "bulk_upload_ms", "pageable_columns_ms", "bytes_all_columns")}
        c["host_to_host"]["region"] = "page-locked host columns -> host SoA (H2D + kernels + D2H), SURVEY 8d (ii)"
    cb = out.get("cpu_baseline")
    if isinstance(cb, dict):
        c["cpu_baseline"] = {k: (cb[k][:120] if isinstance(cb[k], str) else _num(cb[k])) for k in ("value", "unit", "cores", "kind", "wall_s", "wall_s_pool_fit", "pool_fit_workers", "pool_startup_s",
                                                                                                     "pool_fit_startup_s", "critical_path_s", "full_workload", "sample", "rows") if k in cb}
    else:
        c["cpu_baseline"] = cb
    b = out.get("boundary") or {}
    vcf = b.get("vcf_emit_native") or {}
    c["regions_ms"] = {"kernel_only": _num((out.get("kernel_only") or {}).get("ms_per_step")), "resident_delivered_pipelined": _num(out.get("ms_per_step")),
                       "resident_delivered_serial": _num((out.get("resident_delivered_serial") or {}).get("ms_per_step")),
                       "resident_delivered_slim_serial": _num((out.get("resident_delivered_slim") or {}).get("ms_per_step")),
                       "resident_delivered_slim_pipelined": _num((out.get("resident_delivered_slim") or {}).get("pipelined_ms_per_step")),
                       "host_to_host": _num((h or {}).get("ms")), "host_to_host_slim": _num((h or {}).get("slim_ms")),
                       "stage_wall_rows": _num(b.get("stage_wall_ms")), "stage_wall_lazy_rows": _num(b.get("stage_wall_lazy_ms")),
                       "stage_wall_vcf_text": _num(vcf.get("stage_wall_vcf_ms")),
                       "per_task_drop_in": _num((b.get("per_task_drop_in") or {}).get("ms"))}
    if isinstance(out.get("kernel_us"), dict):
        c["kernel_us"] = {k[2:] if k.startswith("k_") else k: v for k, v in out["kernel_us"].items() if isinstance(v, float) and v > 0.3 and not k.startswith("_")}
    if isinstance(out.get("cpu_baseline_c_mt"), dict):
        c["cpu_baseline_c_mt"] = {k: _num(out["cpu_baseline_c_mt"][k]) for k in ("value", "cores", "wall_s")}
    if isinstance(out.get("speedups"), dict):
        c["speedups"] = {k: _num(v, 1) for k, v in out["speedups"].items() if k != "note"}
    c["parity_vs_oracle"] = out.get("parity_vs_oracle")
    c["value_region"] = out.get("value_region")
    if isinstance(out.get("mode1_stage"), dict):
        import bench_stage
        c["mode1_stage"] = {str(out["config"]["workload"]).split(":")[0].split(" ")[0]: bench_stage.compact(out["mode1_stage"])}
    ow = out.get("other_workloads")
    if isinstance(ow, dict):
        c["other_workloads"] = {}
        for name, o in ow.items():
            if "error" in o:
                c["other_workloads"][name] = {"error": o["error"][:120]}
                continue
            d = o["dominant_kernel"]
            e = {"signatures": o["signatures"], "reads": o["reads"], "calls": o["calls"], "ms_per_step": _num(o["ms_per_step"]),
                 "serial_ms_per_step": _num(o.get("serial_ms_per_step")), "slim_ms_per_step": _num(o.get("slim_ms_per_step")),
                 "pipelined_slim_ms_per_step": _num(o.get("pipelined_slim_ms_per_step")), "kernel_only_ms": _num(o["kernel_only_ms_per_step"]), "host_to_host_ms": _num(o["one_shot_call_ms"]),
                 "host_to_host_slim_ms": _num(o.get("one_shot_slim_ms")), "dominant": [d["kernel"], d["us"], d["frac"]],
                 "pipeline_frac": o["pipeline"]["frac"], "parity_vs_oracle": o["parity_vs_oracle"], "c_mt_wall_ms": _num(o["cpu_baseline_c_mt"]["wall_s"] * 1e3, 2)}
            if "cpu_baseline" in o:
                cbo = o["cpu_baseline"]
                e["py_pool"] = [_num(cbo.get(k), 3) for k in ("wall_s", "wall_s_pool_fit", "pool_startup_s", "critical_path_s")]
            if isinstance(o.get("mode1_stage"), dict):
                import bench_stage
                c.setdefault("mode1_stage", {})[name] = bench_stage.compact(o["mode1_stage"])
            if isinstance(o.get("vcf_emit_native"), dict) and "stage_wall_vcf_ms" in o["vcf_emit_native"]:
                e["stage_wall_vcf_ms"] = _num(o["vcf_emit_native"]["stage_wall_vcf_ms"], 2)
            c["other_workloads"][name] = e
    sh = out.get("sharded")
    if isinstance(sh, dict):
        c["sharded"] = {k: _num(sh.get(k)) for k in ("ranks", "signatures_total", "ms_per_step", "value", "scaling", "shard_merge_equals_unsharded", "pieces", "signature_load_max_over_mean", "error") if k in sh}
        c["sharded"]["workload"] = (sh.get("workload") or "")[:60]
        if sh.get("per_rank"):
            c["sharded"]["per_rank_ms"] = [x["ms_per_step"] for x in sh["per_rank"]]
    for k in ("per_rank", "shard_merge_equals_unsharded", "timed_region", "detail"):
        if out.get(k) is not None:
            c[k] = out[k]
    c["keys"] = "other_workloads.*.py_pool = [wall_s at Pool(all cores), at Pool(min(tasks, cores)), the pool alone (no-op tasks), largest task alone]"
    for k in ("per_rank", "other_workloads", "kernel_us", "speedups"):            # (a last resort: never needed at N <= 8)
        if len(json.dumps(c)) + 1 <= LINE_LIMIT:
            break
        c.pop(k, None)
        c.setdefault("trimmed_to_fit_the_line", []).append(k)
    return c


def emit(out):
    if not FULL_LINE and isinstance(out, dict) and "roofline" in out and "kernel_us" in out:
        try:
            d = os.path.join(ROOT, "gpurun_out")
            os.makedirs(d, exist_ok=True)
            path = os.path.join(d, "bench_detail_%s_n%s.json" % (str(out["config"]["workload"]).split(":")[0], out.get("n_gpus")))
            with open(path, "w") as f:
                json.dump(out, f)
            out["detail"] = os.path.relpath(path, ROOT)
        except OSError:
            pass
        sys.stderr.write("[bench detail] " + json.dumps(out) + "\n")
        out = compact(out)
    line = (json.dumps(out) + "\n").encode()
    if _OUT_FD is None:
        sys.stdout.write(line.decode()); sys.stdout.flush()
    else:
        sys.stdout.flush()
        os.write(_OUT_FD, line)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="cfg3")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--mode", default="auto", choices=["auto", "replica", "shard"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-others", action="store_true", help="skip the compact cfg2 / cfg4 / cfg5 objects of the default N = 1 line")
    ap.add_argument("--cpu-procs", type=int, default=0)
    ap.add_argument("--no-mode1", action="store_true", help="skip the mode1_stage legs (the drop-in under the reference's forked Pool, bench_stage.py)")
    ap.add_argument("--full", action="store_true", help="print every object on the one line (default: the compact line, full objects on stderr and in gpurun_out/)")
    a = ap.parse_args()
    global FULL_LINE
    FULL_LINE = a.full
    if "WORLD_SIZE" in os.environ or a.gpus == 1:
        quiet_stdout()

    if a.workload == "rebuild":
        return bench_rebuild(a)
    if a.workload == "extract":
        return bench_extract(a)
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        spawn_ranks(a)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    # BASELINE.json: configs 4 and 5 are ONE genome "sharded over 8 MI355X"; config 3 names one GPU (more GPUs = more genomes)
    shard_mode = a.mode == "shard" or (a.mode == "auto" and world > 1 and a.workload in ("cfg4", "cfg5"))
    if world > 1:
        # one rank per GPU: keep the rank's host threads (page-locked copies, the row builder, the VCF emitter) on its own
        # share of the cores - on an 8-GPU node that is also the socket its GPU hangs off
        try:
            cores = sorted(os.sched_getaffinity(0))
            per = max(1, len(cores) // world)
            os.sched_setaffinity(0, cores[(local_rank % world) * per:(local_rank % world + 1) * per] or cores)
        except (AttributeError, OSError):
            pass
    ctx = None
    if world > 1:
        # (the library and the HIP runtime it links are loaded before torch is imported)
        ndev = max(1, engine.device_count())
        ctx = engine.Context(local_rank % ndev)    # (more ranks than devices: they share; used to rehearse N > 1 on one GPU)
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)     # timing barrier only: no data-path collective

    sharded_obj = None
    if shard_mode:
        if ctx is None:
            ctx = engine.Context(local_rank % max(1, engine.device_count()))
        sharded_obj, dt, total_sig, store, params, wl_name, pstore, phb, tasks, hb = measure_sharded(ctx, dist, rank, world, a.workload, a.scale, a.steps, a.warmup)
        n_sig = int((phb.segments["sig_end"] - phb.segments["sig_begin"]).sum())
        dt_k = dt_slim = dt_serial = dt_pipe_slim = None
        t_upload = t_pin = None
    else:
        store, params, wl_name = make_workload(a.workload, a.scale, rank)
        tasks = store.tasks()
        hb = store.host_batch(tasks, params)
        n_sig = int((hb.segments["sig_end"] - hb.segments["sig_begin"]).sum())

    # ---------------- CPU baselines first (fork pools before any HIP state exists in this process)
    cpu = cpu_c = cpu_c_mt = ores = mode1 = None
    others = {}
    if rank == 0 and world == 1 and not a.no_cpu_baseline and not shard_mode:
        procs = a.cpu_procs or os.cpu_count() or 1
        cpu, cpu_c, cpu_c_mt, ores = cpu_legs(a.workload, store, params, tasks, hb, n_sig, procs, py_pool=True, full_pool=False)
        cal = os.path.join(ROOT, "profiles", "calibration_py_restatement.json")
        if os.path.exists(cal):
            with open(cal) as f:
                cpu["calibration_vs_reference"] = json.load(f)
        if a.workload in ("cfg3", "cfg4") and a.scale == 1.0 and not a.no_mode1:
            mode1 = stage_leg(a.workload, store, params)
        if a.workload == "cfg3" and a.scale == 1.0 and not a.no_others:
            for name in ("cfg2", "cfg4", "cfg5"):
                try:
                    st_o, p_o, wl_o = make_workload(name, 1.0, 0)
                    tk_o = st_o.tasks()
                    hb_o = st_o.host_batch(tk_o, p_o)
                    ns_o = int((hb_o.segments["sig_end"] - hb_o.segments["sig_begin"]).sum())
                    # the Python pool (the reference's execution model) on the FULL 90x workload; cfg2 / cfg4: the C legs only
                    c0, c1, c2, o_o = cpu_legs(name, st_o, p_o, tk_o, hb_o, ns_o, procs, py_pool=(name in ("cfg4", "cfg5")), full_pool=True)
                    others[name] = dict(store=st_o, params=p_o, wl_name=wl_o, tasks=tk_o, hb=hb_o, n_sig=ns_o, cpu=c0, cpu_c=c1, cpu_c_mt=c2, ores=o_o)
                    if name == "cfg4" and not a.no_mode1:
                        others[name]["mode1"] = stage_leg(name, st_o, p_o)
                except Exception as e:          # noqa: BLE001  (never let a compact leg break the benchmark line)
                    others[name] = dict(error=repr(e))

    # ---------------- GPU
    if ctx is None:
        ctx = engine.Context(local_rank % max(1, engine.device_count()))
    if not shard_mode:
        # the columns as a worker process holds them: in page-locked host memory
        t0 = time.perf_counter()
        pstore = store.pinned()
        t_pin = time.perf_counter() - t0
        phb = pstore.host_batch(tasks, params)
        t0 = time.perf_counter()
        ctx.upload(phb, per_sig=False)
        t_upload = time.perf_counter() - t0
        dt_serial, dt_k, _, dt_slim, dt, dt_pipe_slim = resident_loops(ctx, phb, a.steps, a.warmup, dist)
        total_sig = n_sig
        dt_own = dt
        if dist is not None:
            import torch
            tt = torch.tensor([dt, dt_k], dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt, dt_k = float(tt[0]), float(tt[1])
            ts = torch.tensor([n_sig], dtype=torch.float64)
            dist.all_reduce(ts, op=dist.ReduceOp.SUM)
            total_sig = int(ts[0])
    # which device every rank sat on (a SCALE record can then show N distinct GPUs) and what it measured by itself
    try:
        bus, ncu = engine.device_info(ctx.device)
    except Exception:            # noqa: BLE001
        bus, ncu = None, None
    own_ms = (dt_own if not shard_mode else dt) / a.steps * 1e3
    per_rank = [dict(rank=rank, device=ctx.device, devices_visible=engine.device_count(), pci_bus_id=bus, compute_units=ncu,
                     signatures=n_sig, ms_per_step=round(own_ms, 5), value=round(n_sig / (own_ms * 1e-3)))]
    if dist is not None:
        gl = [None] * world
        dist.all_gather_object(gl, per_rank[0])
        per_rank = gl
    ms_reads_kept = None
    if not shard_mode and phb.r_start is not None:
        ctx.option(1, 1)
        for _ in range(3):
            ctx.run()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            ctx.run()
        ctx.sync()
        ms_reads_kept = (time.perf_counter() - t0) / a.steps * 1e3
        ctx.option(1, 0)
    ms_per_step = dt / a.steps * 1e3
    value = total_sig * a.steps / dt
    ms_kernel_only = None if dt_k is None else dt_k / a.steps * 1e3

    # world > 1, replica mode: the un-parameterised command also measures ONE genome sharded over the ranks (BASELINE config 4)
    if world > 1 and not shard_mode and a.workload == "cfg3":
        try:
            sharded_obj = measure_sharded(ctx, dist, rank, world, "cfg4", a.scale, max(5, min(a.steps, 20)), 2)[0]
        except Exception as e:          # noqa: BLE001
            sharded_obj = {"error": repr(e)}
            if dist is not None:
                raise

    out = None
    if rank == 0:
        names = engine.stage_names()
        per_kernel, per_kernel_nps, st, res, kernel_time_us = instrumented(ctx, phb, a.steps)
        # cold: caches evicted before every step (the flush is outside the event-timed region of csv_batch_run)
        cold_acc = np.zeros(_abi.N_STAGES)
        cold_tot = 0.0
        ncold = min(a.steps, 10)
        for _ in range(ncold):
            ctx.cache_flush(1 << 30)
            s2 = ctx.run(stats=True)
            cold_acc += np.array(list(s2.ms_stage))
            cold_tot += s2.ms_total
        per_kernel_cold = per_kernel_us(cold_acc / ncold, names)

        # ---------------- the boundary as a drop-in sees it
        t_one = timed(lambda: ctx.cluster_batch(phb, reuse=True), 9)          # (caller-owned result arrays, allocated once)
        gate_first, bytes_not_sent = ctx.lazy_info()
        delta16 = ctx.delta16_info()
        rbits = ctx.reads_delta_info() if phb.r_start is not None else 0        # (of THIS call: later legs go through other faces)
        t_one_slim = timed(lambda: ctx.cluster_batch(phb, reuse=True, **SLIM), 9) if phb.a.dtype == np.int32 else None
        os.environ["CSV_NO_LAZY"] = "1"                                       # the whole columns in one piece (r04's form)
        t_one_bulk = timed(lambda: ctx.cluster_batch(phb, reuse=True), 5)
        del os.environ["CSV_NO_LAZY"]
        t_one_pageable = timed(lambda: ctx.cluster_batch(hb, reuse=True), 5)
        r2 = ctx.cluster_batch(phb)
        t_rows = timed(lambda: rows_mod.rows_by_segment(pstore, phb.segments, r2), 3)
        t_stage = timed(lambda: resolve.cluster_stage(pstore, params, tasks=tasks, ctx=ctx), 5)
        # the same stage handing out lazy row sequences (rows.LazyRows: a row's strings are created when somebody reads it)
        t_stage_lazy = timed(lambda: resolve.cluster_stage(pstore, params, tasks=tasks, ctx=ctx, lazy=True), 7)
        n_rows = sum(len(v) for v in resolve.cluster_stage(pstore, params, tasks=tasks, ctx=ctx).values())
        # (bytes per signature / read as the ABI defines the columns; positions and lengths travel as int32 when the store
        # keeps narrow twins - CSV_IN_SIG_I32 / CSV_IN_READS_I32)
        # (reads: CSV_IN_READS_DELTA16 sends the starts as 16-bit gaps / the ends as 16-bit lengths, r_idp the id and the primary flag
        # as one word - whichever of them the last call took: csv_batch_info 3)
        per_read = (2 if rbits & 1 else phb.r_start.dtype.itemsize) + (2 if rbits & 2 else phb.r_start.dtype.itemsize) + (4 if rbits & 4 else 5) if phb.r_start is not None else 0
        h2d_bytes = (2 * phb.a.dtype.itemsize + 8) * n_sig + (per_read * int(phb.r_start.shape[0]) if phb.r_start is not None else 0)
        t_vcf = vcf_leg(ctx, pstore, params, tasks)
        t_task = per_task_leg(ctx, store, params, tasks)

        # measured device-to-device copy ceiling of this box (report the fraction of the vendor peak AND of
        # the copy ceiling): 512 MiB hipMemcpy device to device, read + write bytes over the best of 10 runs
        copy_gbs = None
        try:
            copy_gbs = ctx.copy_bandwidth(512 << 20, 10)
        except Exception as e:           # noqa: BLE001  (optional leg)
            print("copy ceiling not measured: %r" % (e,), file=sys.stderr)
        try:
            kbytes, total_bytes, units = kernel_units(store, hb, res, st, per_sig_step=False)
        except Exception:                # noqa: BLE001  (timing experiments with ablated kernels: CSV_BENCH_LENIENT=1)
            if not os.environ.get("CSV_BENCH_LENIENT"):
                raise
            kbytes, total_bytes, units = {n: 1 for n in names if n}, 1, {}
        dom = dominant(per_kernel_nps, kbytes)
        # kernel_us (net) is the kernel's own time between two event records; what rocprofv3 --kernel-trace calls its duration -
        # and what the kernel costs the step - also holds one kernel boundary (dispatch to dispatch, ~1.6 us here).  When the
        # whole step is one chain on one stream (no side streams: no pair types, no reads stage) the boundary follows from the
        # plain kernel-only loop itself: (step - sum of the kernels) / kernels launched, and the roofline is priced on kernel +
        # boundary, so that it agrees with the committed trace (profiles/*_kernel_trace.txt).
        # (r06: a resident run queues the two tiers above 64 signatures whether they have work or not; an empty launch is all
        # boundary, so its whole slot is charged to the boundaries of the kernels that do work - which lands within 1-2 % of the
        # dominant kernel's rocprofv3 average: 14.5 vs 14.46 us on cfg3)
        launched = [n for n in per_kernel_nps if n.startswith("k_") and isinstance(per_kernel_nps[n], float) and per_kernel_nps[n] > 0.6
                    and not (n in ("k_refine_mid", "k_refine_block", "k_refine_wave") and per_kernel_nps[n] < 2.0)]
        one_chain = not shard_mode and all(per_kernel_nps.get(n, 0.0) < 0.6 for n in ("k_refine_wave", "k_reads_order", "k_reads_gather", "k_genotype_tra"))
        boundary_us = None
        if one_chain and launched and ms_kernel_only is not None:
            boundary_us = max(0.0, (ms_kernel_only * 1e3 - sum(per_kernel_nps[n] for n in launched)) / len(launched))
        dom_us = per_kernel_nps[dom] + (boundary_us or 0.0)
        achieved = kbytes[dom] / (dom_us * 1e-6) / 1e9
        cold_achieved = kbytes[dom] / (per_kernel_cold[dom] * 1e-6) / 1e9 if per_kernel_cold[dom] > 0 else None
        traffic_all = traffic_of(a.workload, a.scale)
        traffic = None if traffic_all is None else (sum(traffic_all.get(k, 0) for k in GT_KERNELS) if dom == "genotype_stage" else traffic_all.get(dom))
        roof_all = {n: {"us": per_kernel_nps[n], "algorithmic_bytes": kbytes[n], "gbs": round(kbytes[n] / (per_kernel_nps[n] * 1e-6) / 1e9, 1),
                        "frac": round(kbytes[n] / (per_kernel_nps[n] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)}
                    for n in per_kernel_nps if n in kbytes and kbytes[n] > 0 and isinstance(per_kernel_nps[n], float) and per_kernel_nps[n] > 0}
        parity = None
        if ores is not None:
            w, g = ores.trimmed(), res.trimmed()
            parity = all(np.array_equal(g[k], w[k]) for k in PARITY_FIELDS)
        stage_ms = float(np.median(t_stage)) * 1e3
        one_ms = float(np.min(t_one)) * 1e3
        one_slim_ms = None if t_one_slim is None else float(np.min(t_one_slim)) * 1e3
        one_bulk_ms = float(np.min(t_one_bulk)) * 1e3
        # what crosses PCIe in the host -> host call: the bulk copies (all the columns, or the position column alone in the
        # gate-first form), the rows the device then reads out of the caller's columns (useful bytes: 12 B per signature of a
        # gated cluster + 4 B for INS; the link moves whole 64-byte lines of them), the reads table, and the result coming back
        down_full = 76 * r2.n_calls + 4 * r2.n_support + 8
        down_slim = 4 * (4 + len(SLIM["fields"])) * r2.n_calls
        refine_alg = sum(kbytes.get(k, 0) for k in ("k_refine_indel_wave", "k_refine_wave", "k_refine_mid", "k_refine_block"))
        fetched_useful = (refine_alg - (4 if phb.a.dtype == np.int32 else 0) * units.get("sig_in_gated_clusters", 0)) if gate_first else 0
        up_bulk = h2d_bytes - bytes_not_sent                    # (gate-first: b / read_id / aux stay behind; CSV_IN_SIG_DELTA16: half of the position column)

        def pcie(ms, down):
            moved = up_bulk + fetched_useful + down
            return {"bytes_up_bulk": int(up_bulk), "bytes_up_fetched_useful": int(fetched_useful), "bytes_down": int(down), "ms": ms,
                    "achieved_gbs": moved / (ms * 1e-3) / 1e9, "frac_of_63": moved / (ms * 1e-3) / 1e9 / PCIE_PEAK_GBS}
        host_to_host = {
            "region": "csv_cluster_batch: page-locked host columns -> kernels -> result SoA in page-locked host memory (SURVEY 8d region (ii); the "
                      "reference's timed region MAIN:1113-1199 minus the rows)",
            "gate_first": bool(gate_first), "delta16": bool(delta16), "reads_16bit_forms": int(rbits),
            "ms": one_ms, "value": n_sig / (one_ms * 1e-3), "ms_all": [round(x * 1e3, 3) for x in t_one],
            "slim_ms": one_slim_ms, "slim_value": None if one_slim_ms is None else n_sig / (one_slim_ms * 1e-3),
            "bulk_upload_ms": one_bulk_ms, "pageable_columns_ms": float(np.min(t_one_pageable)) * 1e3,
            "pcie": pcie(one_ms, down_full), "pcie_slim": None if one_slim_ms is None else pcie(one_slim_ms, down_slim),
            "bytes_all_columns": int(h2d_bytes),
            "note": "gate-first: only the position column (+ the reads table) is copied in bulk; the device reads b / read_id / aux of the clusters "
                    "that pass the size gate out of the caller's columns (whole 64-byte lines cross the link: bytes_up_fetched_useful is the "
                    "rows' own bytes).  slim = CSV_OUT_NO_SUPPORT_LIST | CSV_OUT_COORD_I32, fields %s: what the VCF emitter reads without "
                    "--report_readid" % (SLIM["fields"],)}
        ko_ms = ms_kernel_only if ms_kernel_only is not None else None
        other_out = {}
        for name, cpu_o in others.items():
            if "error" in cpu_o:
                other_out[name] = cpu_o
                continue
            try:
                other_out[name] = compact_workload(ctx, name, a, cpu_o)
            except Exception as e:      # noqa: BLE001
                other_out[name] = {"error": repr(e)}
            cpu_o.clear()
        out = {
            "metric": "SV signatures clustered/sec (whole node)", "value": value, "unit": "signatures/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step,
            "timed_region": ("the rank's whole boundary call: page-locked host columns -> kernels -> host SoA" if shard_mode else
                             "inputs resident in HBM -> all kernels -> every call field + the int32 support lists delivered into page-locked host arrays (int32 coordinates with int32 columns), every step's full "
                             "result; the delivery of step k (one copy-engine transfer of the result block, written as a device image behind the run) runs under the kernels of step k + 1 (csv_batch_publish_async)"),
            "kernel_only": None if ko_ms is None else {"ms_per_step": ko_ms, "value": total_sig / (ko_ms * 1e-3),
                                                       "note": "the launch sequence alone, results left in HBM (the region r01-r03 reported as value)"},
            "ms_per_step_reads_order_kept": ms_reads_kept,
            "higher_is_better": True, "scaling": "strong" if shard_mode else "weak", "vs_baseline": None, "dtype": "int32 columns, int64+f64 arithmetic" if phb.a.dtype == np.int32 else "int64+f64", "data": "synthetic",
            "config": {"workload": wl_name, "signatures_per_gpu": n_sig, "signatures_total": total_sig, "segments": len(tasks),
                       "preset": "ONT" if a.workload in ("cfg2", "cfg3", "cfg5") else "HiFi",
                       "genotype": bool(params.genotype), "mode": a.mode, "columns": "int32 positions / lengths (SigStore.pinned())",
                       "sharding": (sharded_obj["mode"] if (shard_mode and sharded_obj) else "one genome per GPU, no collective")},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_note": "L2 <-> fabric bytes per launch (2 x FETCH_SIZE + WRITE_SIZE, rocprofv3 PMC passes of this command, calibrated on "
                                         "known byte counts: profiles/r04_pmc_calibration.txt); includes requests the Infinity Cache answers",
                         "algorithmic_bytes": kbytes[dom], "kernel_us": round(dom_us, 2), "kernel_us_net": per_kernel_nps[dom],
                         "boundary_us": None if boundary_us is None else round(boundary_us, 2),
                         "valu_issue": valu_issue_of(a.workload, a.scale, "k_genotype" if dom == "genotype_stage" else dom,
                                                     per_kernel_nps.get("k_genotype") if dom == "genotype_stage" else per_kernel_nps[dom]),
                         "copy_ceiling": copy_gbs, "frac_of_copy_ceiling": (achieved / copy_gbs) if copy_gbs else None,
                         "cold": {"kernel_us": per_kernel_cold[dom], "achieved": cold_achieved, "frac": None if cold_achieved is None else cold_achieved / HBM_PEAK_GBS,
                                  "pipeline_us": round(cold_tot / ncold * 1e3, 2),
                                  "note": "L2 + Infinity Cache evicted before every step (csv_cache_flush, 1 GiB)"}},
            # one denominator: the kernel-only loop (the algorithmic bytes are the kernels'; the host delivery is not HBM traffic)
            "roofline_pipeline": {"algorithmic_bytes": total_bytes, "step_us": None if ko_ms is None else round(ko_ms * 1e3, 2),
                                  "achieved": total_bytes / (ko_ms * 1e-3) / 1e9 if ko_ms else None, "unit": "GB/s",
                                  "frac": total_bytes / (ko_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if ko_ms else None,
                                  "instrumented_pass_us": round(kernel_time_us, 2),
                                  "note": "achieved / frac on the plain kernel-only loop; instrumented_pass_us is the event-per-kernel pass (each record occupies the stream) and is NOT a denominator"},
            "roofline_per_kernel": roof_all, "traffic_per_kernel": traffic_all,
            "kernel_us": per_kernel_nps, "kernel_us_per_sig_outputs": per_kernel, "kernel_us_cold": per_kernel_cold, "units": units,
            "cpu_baseline": cpu, "cpu_baseline_c": cpu_c, "cpu_baseline_c_mt": cpu_c_mt,
            # speed-ups, each over ONE named pair of regions (r03's speedup_vs_cpu_baseline divided the kernel-only loop by a Python pool)
            "mode1_stage": mode1,
            "value_region": "sharded_boundary_call" if shard_mode else "resident_delivered_pipelined",
            "speedups": {
                "stage_wall_vs_reference_model": (cpu["wall_s"] * 1e3 / stage_ms) if (cpu and cpu.get("full_workload")) else None,
                "stage_wall_vs_reference_model_pool_fit": (cpu["wall_s_pool_fit"] * 1e3 / stage_ms) if (cpu and cpu.get("full_workload") and cpu.get("wall_s_pool_fit")) else None,
                "stage_wall_vs_reference_critical_path": (cpu["critical_path_s"] * 1e3 / stage_ms) if (cpu and cpu.get("critical_path_s")) else None,
                "host_to_host_vs_c_all_threads": (cpu_c_mt["wall_s"] * 1e3 / one_ms) if cpu_c_mt else None,
                "resident_step_vs_c_all_threads": (cpu_c_mt["wall_s"] * 1e3 / ms_per_step) if (cpu_c_mt and world == 1) else None,
                "note": "reference_model = oracle/py_restatement.py in a fork Pool at all host cores (cuteSV's execution model, the north_star target), "
                        "whose wall is mostly the pool itself (cpu_baseline.pool_startup_s); _pool_fit = the same at one worker per task; "
                        "_critical_path = its largest task alone (what no number of cores gets under); "
                        "c_all_threads = the C oracle, one (chr,type) task per thread; stage_wall = page-locked columns -> the reference's row lists. "
                        "The like-for-like stage comparison (both sides under the same forked pool on the same pickles) is mode1_stage"},
            "host_to_host": host_to_host,
            "resident_delivered_serial": None if dt_serial is None else {"ms_per_step": dt_serial / a.steps * 1e3, "value": n_sig * a.steps / dt_serial,
                                                                         "note": "run, download, run, download ...: what r04 reported as value"},
            "resident_delivered_slim": None if dt_slim is None else {"ms_per_step": dt_slim / a.steps * 1e3, "value": n_sig * a.steps / dt_slim,
                                                                     "pipelined_ms_per_step": None if dt_pipe_slim is None else dt_pipe_slim / a.steps * 1e3,
                                                                     "note": "the slim result (no support lists, int32 coordinates): serial and pipelined"},
            "boundary": {"pin_ms": None if t_pin is None else t_pin * 1e3,
                         "one_shot_call_ms": one_ms,
                         "rows_ms": float(np.median(t_rows)) * 1e3, "rows": n_rows,
                         "stage_wall_ms": stage_ms, "stage_wall_ms_all": [round(x * 1e3, 3) for x in t_stage],
                         "stage_wall_lazy_ms": float(np.median(t_stage_lazy)) * 1e3,
                         "vcf_emit_native": t_vcf, "per_task_drop_in": t_task,
                         "stage_signatures_per_s": n_sig / (stage_ms * 1e-3)},
            "other_workloads": other_out or None,
            "per_rank": per_rank,
            "sharded": sharded_obj if not shard_mode else None,
            "parity_vs_oracle": parity, "shard_merge_equals_unsharded": (sharded_obj or {}).get("shard_merge_equals_unsharded") if shard_mode else None,
        }
        if not shard_mode:
            out["boundary"]["upload_ms"] = t_upload * 1e3
        emit(out)
    if dist is not None:
        dist.barrier()
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()
    if os.environ.get("CSV_BENCH_EXIT_ALARM"):     # under rocprofv3 the process was seen to hang AFTER the tool had
        import signal                              # written its output; let teardown run, but not forever
        sys.stdout.flush()
        signal.alarm(int(os.environ["CSV_BENCH_EXIT_ALARM"]))


if __name__ == "__main__":
    main()
