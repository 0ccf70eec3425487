#!/usr/bin/env python3
"""bench.py — signatures clustered per second on MI355X (BASELINE.json's metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg3|cfg2|cfg4|cfg5] [--scale S]

A "step" is one pass of the whole hot path (chain -> select -> refine -> order [-> genotype]) over one
synthetic signature batch that is already resident in HBM when the timed region starts.  At N = 1 the
workload is BASELINE config 3 (synthetic HG002-shaped ONT 30x, ~2.8 M INS+DEL signatures, ONT preset):
the metric is quoted "on 30x WGS", and this is the 30x whole-genome configuration that names one MI355X.
For N > 1 every rank (one process per GPU, launched by torch.distributed.run) clusters its own genome
of that shape (seed + rank): chromosomes / samples shard with no data-path collective,
so scaling is "weak" and the only communication is the timing barrier (gloo, CPU tensors).

Prints ONE JSON line on rank 0.  Extra objects:
  roofline      dominant kernel (largest average HIP-event duration over the K steps of a second, event-
                instrumented pass on the library's own stream): algorithmic bytes (32 B per
                signature the launch processes + 64 B per call; genotype 21 B per read + 32 B per call)
                / duration vs the 8 TB/s HBM peak.  `traffic` comes from the rocprofv3 PMC passes
                committed under profiles/ for the same command, or null.
  cpu_baseline  oracle/py_restatement.py (the reference's execution model: Python loops + numpy scalars in
                a multiprocessing pool at all host cores) timed on rank 0 at N = 1 on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from cutesv_amd import synth, engine, _abi, rows as rows_mod     # noqa: E402
from cutesv_amd.columns import Params                            # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); the copy ceiling is measured per run (~5.3 TB/s)


def make_workload(name, scale, rank):
    if name == "cfg3":
        return synth.ont30(seed=20260103 + rank, scale=scale), Params.ont(), \
            "cfg3: synthetic HG002-shaped ONT 30x, INS+DEL, ONT preset (synth.ont30, scale %g)" % scale
    if name == "cfg2":
        sites = dict(np.load(os.path.join(ROOT, "tests", "golden", "sim_sites.npz")))
        return synth.sim_all_types(sites, seed=20260102 + rank), Params.ont(), "cfg2: all five simulation beds, whole genome, ONT preset"
    if name == "cfg4":
        return synth.hifi30_gt(seed=20260104 + rank, scale=scale), Params.hifi(genotype=True, min_support=3), \
            "cfg4: synthetic HiFi 30x with --genotype (synth.hifi30_gt, scale %g)" % scale
    if name == "cfg5":
        return synth.ont90_all(seed=20260105 + rank, scale=scale), Params.ont(genotype=True), \
            "cfg5: synthetic ONT 90x, all five types, --genotype for INS/DEL/DUP/INV (synth.ont90_all, scale %g)" % scale
    raise SystemExit("unknown workload " + name)


def kernel_units(store, hb, res, stats):
    """signatures / reads / calls each kernel processes in one launch -> algorithmic bytes"""
    t = res.trimmed()
    cid = t["cluster_id"]
    cid = cid[cid >= 0]
    W = int(cid.shape[0])
    sizes = np.bincount(cid, minlength=t["n_clusters"])
    first = np.flatnonzero(np.r_[True, cid[1:] != cid[:-1]])
    segs = hb.segments
    woff = np.r_[0, np.cumsum(segs["sig_end"] - segs["sig_begin"])]
    seg_of_cluster = np.searchsorted(woff, first, side="right") - 1
    valid = sizes >= segs["read_count"][seg_of_cluster]
    indel = segs["svtype"][seg_of_cluster] <= _abi.INS
    n_iw = int(sizes[valid & (sizes <= 64) & indel].sum())             # k_refine_indel_wave
    n_pw = int(sizes[valid & (sizes <= 64) & ~indel].sum())            # k_refine<64,64>
    n_mid = int(sizes[valid & (sizes > 64) & (sizes <= 256)].sum())    # k_refine<64,256>
    n_blk = int(sizes[valid & (sizes > 256)].sum())                    # k_refine<256,2048>
    n_small, n_big = n_iw + n_pw, n_mid + n_blk
    calls, sup = res.n_calls, res.n_support
    R = 0 if hb.r_start is None else int(hb.r_start.shape[0])
    gt_calls = int((t["gl_idx"] >= 0).sum())
    per_sig, per_call = 32, 64
    n_ref = max(1, n_small + n_big)
    share = lambda n: per_sig * n + per_call * calls * n // n_ref       # calls attributed in proportion to the signatures refined
    b = {
        "k_chain_count": per_sig * W, "k_chain_apply": per_sig * W,
        "k_refine_indel_wave": share(n_iw), "k_refine_wave": share(n_pw), "k_refine_mid": share(n_mid), "k_refine_block": share(n_blk),
        "k_emit": per_call * calls + 8 * sup, "k_items_scan": 8 * int(stats.n_work_wave + stats.n_work_block),
        # genotyping is judged as ONE stage (prefix max over the reads table + the per-call stabbing queries): 21 B per
        # read + 32 B per genotyped call over the summed duration of its four kernels (see main())
        "genotype_stage": 21 * R + 32 * gt_calls,
    }
    total = per_sig * W + per_call * calls + (21 * R + 32 * gt_calls if R else 0)
    return b, total, dict(signatures=W, sig_in_wave_items=n_small, sig_in_block_items=n_big, calls=calls, supports=sup,
                          reads=R, clusters=int(t["n_clusters"]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="cfg3")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-procs", type=int, default=0)
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None

    store, params, wl_name = make_workload(a.workload, a.scale, rank)
    tasks = store.tasks()
    hb = store.host_batch(tasks, params)
    n_sig = int((hb.segments["sig_end"] - hb.segments["sig_begin"]).sum())

    # ---------------- CPU baseline first (fork pool before any HIP state exists in this process)
    cpu = None
    cpu_c = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        from oracle import oracle, py_restatement as pr
        procs = a.cpu_procs or os.cpu_count() or 1
        # bounded sample: all tasks of the workload when it has <= 4 M signatures (a few seconds of pool time),
        # else the segments of every 3rd chromosome (same flags)
        sample = tasks if n_sig <= 4_000_000 else [t for i, t in enumerate(tasks) if (i % 24) % 3 == 0]
        tl = pr.tasks_from_store(store, params, sample)
        ns = sum(len(t[2]) for t in tl)
        t0 = time.perf_counter()
        r = pr.run_pool_forked(tl, procs)
        dt = time.perf_counter() - t0
        cpu = dict(value=ns / dt, unit="signatures/s", cores=procs, kind="port",
                   sample="%d of %d (chr,type) tasks, %d signatures, %.2f s wall; oracle/py_restatement.py in a fork "
                          "Pool(%d): the reference's pool model (Python per-signature loop, numpy scalar calls)"
                          % (len(sample), len(tasks), ns, dt, procs),
                   rows=sum(len(x[1]) for x in r))
        del tl, r
        t0 = time.perf_counter()
        ores = oracle.cluster_batch(hb, per_sig=True)
        dtc = time.perf_counter() - t0
        cpu_c = dict(value=n_sig / dtc, unit="signatures/s", cores=1, kind="port",
                     sample="full workload, oracle/cutesv_oracle.c single thread, %.3f s" % dtc)

    # ---------------- GPU (the library and the HIP runtime it links are loaded before torch is imported)
    ctx = engine.Context(local_rank)
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)     # timing barrier only: no data-path collective
    t0 = time.perf_counter()
    ctx.upload(hb)
    t_upload = time.perf_counter() - t0
    for _ in range(a.warmup):
        ctx.run()
    ctx.sync()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        ctx.run()
    ctx.sync()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    total_sig = n_sig
    if dist is not None:
        import torch
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt[0])
        ts = torch.tensor([n_sig], dtype=torch.float64)
        dist.all_reduce(ts, op=dist.ReduceOp.SUM)
        total_sig = int(ts[0])
    ms_per_step = dt / a.steps * 1e3
    value = total_sig * a.steps / dt

    # ---------------- instrumented pass: per-kernel HIP-event durations on the library's stream
    acc = np.zeros(_abi.N_STAGES)
    tot = 0.0
    st = None
    for _ in range(a.steps):
        st = ctx.run(stats=True)
        acc += np.array(list(st.ms_stage))
        tot += st.ms_total
    acc /= a.steps
    names = engine.stage_names()
    t0 = time.perf_counter()
    res = ctx.download(per_sig=True)
    t_download = time.perf_counter() - t0
    t0 = time.perf_counter()
    row_list = rows_mod.materialise(store, hb.segments, res.trimmed())
    t_rows = time.perf_counter() - t0
    t0 = time.perf_counter()
    r2 = ctx.cluster_batch(hb)          # the one-shot C-ABI call: H2D + kernels + D2H
    t_boundary = time.perf_counter() - t0
    # native VCF record emit straight from the SoA (no Python rows).  ignore_sequence: a 3.1 Gbp synthetic reference is
    # not materialised for the benchmark, so REF/ALT are 'N' / '<TYPE>' as with cuteSV's --ignore_sequence; pair types
    # (which always look up one base) are left out of this timing
    t_vcf = None
    if rank == 0 and not params.genotype or rank == 0:
        from cutesv_amd import vcf as vcf_mod
        try:
            keep = [i for i, (t, c) in enumerate(tasks) if t in ("DEL", "INS")]
            hb2 = store.host_batch([tasks[i] for i in keep], params)
            r3 = ctx.cluster_batch(hb2)
            t0 = time.perf_counter()
            text, _ = vcf_mod.emit_records(store, hb2.segments, r3, None, min_size=params.min_size, max_size=params.max_size,
                                           genotype=params.genotype, ignore_sequence=True)
            t_vcf = dict(ms=(time.perf_counter() - t0) * 1e3, records=text.count("\n"), bytes=len(text))
        except Exception as e:          # never let the optional leg break the benchmark line
            t_vcf = dict(error=str(e))

    out = None
    if rank == 0:
        # measured device-to-device copy ceiling of this box (report the fraction of the vendor peak AND of
        # the copy ceiling): 512 MiB hipMemcpy device to device, read + write bytes over the best of 10 runs
        copy_gbs = None
        try:
            copy_gbs = ctx.copy_bandwidth(512 << 20, 10)
        except Exception as e:           # noqa: BLE001  (optional leg)
            print("copy ceiling not measured: %r" % (e,), file=sys.stderr)
            copy_gbs = None
        kbytes, total_bytes, units = kernel_units(store, hb, res, st)
        per_kernel = {names[i]: round(float(acc[i]) * 1e3, 2) for i in range(_abi.N_STAGES) if names[i]}   # microseconds
        per_kernel["genotype_stage"] = round(sum(per_kernel.get(k, 0.0) for k in ("k_pmax_count", "k_pmax_apply", "k_genotype")), 2)
        dom = max((n for n in per_kernel if n in kbytes and kbytes[n] > 0), key=lambda n: per_kernel[n])
        dom_s = per_kernel[dom] * 1e-6
        achieved = kbytes[dom] / dom_s / 1e9
        traffic = None
        tf = os.path.join(ROOT, "profiles", "traffic_%s.json" % a.workload)
        if os.path.exists(tf) and a.scale == 1.0:
            with open(tf) as f:
                tj = json.load(f)
                traffic = sum(tj.get(k, 0) for k in ("k_pmax_count", "k_pmax_apply", "k_genotype")) if dom == "genotype_stage" else tj.get(dom)
        parity = None
        if cpu_c is not None:
            w, g = ores.trimmed(), res.trimmed()
            parity = all(np.array_equal(g[k], w[k]) for k in ("call_seg", "call_cluster", "bp1", "bp2", "support", "cipos", "cilen",
                                                             "search_pos", "seq_pick", "dr", "dv", "gl_idx", "support_off",
                                                             "support_sig", "cluster_id", "allele_id"))
        out = {
            "metric": "SV signatures clustered/sec (whole node)", "value": value, "unit": "signatures/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int64+f64", "data": "synthetic",
            "config": {"workload": wl_name, "signatures_per_gpu": n_sig, "segments": len(tasks), "preset": "ONT" if a.workload in ("cfg2", "cfg3", "cfg5") else "HiFi",
                       "genotype": bool(params.genotype), "sharding": "one genome per GPU, no collective"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes": kbytes[dom], "kernel_us": per_kernel[dom],
                         "copy_ceiling": copy_gbs, "frac_of_copy_ceiling": (achieved / copy_gbs) if copy_gbs else None},
            "roofline_pipeline": {"algorithmic_bytes": total_bytes, "kernel_time_us": round(tot / a.steps * 1e3, 2),
                                  "achieved": total_bytes / (ms_per_step * 1e-3) / 1e9, "unit": "GB/s",
                                  "frac": total_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS},
            "kernel_us": per_kernel, "units": units,
            "cpu_baseline": cpu, "cpu_baseline_c": cpu_c,
            "speedup_vs_cpu_baseline": (value / world / cpu["value"]) if cpu else None,
            "boundary": {"upload_ms": t_upload * 1e3, "download_ms": t_download * 1e3, "rows_ms": t_rows * 1e3,
                         "one_shot_call_ms": t_boundary * 1e3, "rows": len(row_list), "vcf_emit_native": t_vcf,
                         "pcie_inclusive_signatures_per_s": n_sig / t_boundary},
            "parity_vs_oracle": parity,
        }
        print(json.dumps(out))
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()
    if os.environ.get("CSV_BENCH_EXIT_ALARM"):     # under rocprofv3 the process was seen to hang AFTER the tool had
        import signal                              # written its output; let teardown run, but not forever
        sys.stdout.flush()
        signal.alarm(int(os.environ["CSV_BENCH_EXIT_ALARM"]))


if __name__ == "__main__":
    main()
