"""split_sigs.json.gz: the reference's organize_split_signal / analysis_split_read (main script :50-513) driven with
synthetic primary alignments and SA-tag entries (see make_golden_main.py, which imports the main script and calls
split_golden)."""
import gzip
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from cutesv_amd import synth                                  # noqa: E402  (pseudo_sequence: the fixtures store its key, not the bases)

CHROMS = ["1", "10", "2", "X"]                                # (string order != numeric order: analysis_bnd compares names)


def sa_cigar(rng, clip0, span, clip1):
    """a CIGAR text whose leading / trailing clips and reference span are the given numbers; hard clips, insertions and
    skipped regions are thrown in (acquire_clip_pos counts only S clips and M / D / = / X spans)"""
    ops = []
    hard = rng.random() < 0.1
    if clip0:
        ops.append("%d%s" % (clip0, "H" if hard else "S"))
    left = span
    while left > 0:
        k = int(min(left, rng.integers(1, 4000)))
        ops.append("%d%s" % (k, rng.choice(["M", "M", "M", "=", "X", "D"])))
        left -= k
        if left > 0 and rng.random() < 0.3:
            ops.append("%d%s" % (int(rng.integers(1, 50)), rng.choice(["I", "N"])))
    if clip1:
        ops.append("%d%s" % (clip1, "H" if hard else "S"))
    return "".join(ops), (0 if hard else clip0), (0 if hard else clip1)


def random_split_read(rng, name, key, kind):
    """segments that roughly tile the read; `kind` steers how consecutive segments relate on the reference"""
    L = int(rng.integers(800, 20000))
    n = int(rng.choice([1, 2, 2, 2, 3, 3, 3, 4, 5, 6, 9]))
    cuts = np.sort(rng.integers(0, L, size=2 * n))
    segs = []
    chrom = str(rng.choice(CHROMS)); strand = str(rng.choice(["+", "-"])); ref = int(rng.integers(10_000, 5_000_000))
    for k in range(n):
        rs, re = int(cuts[2 * k]), int(cuts[2 * k + 1])
        if rng.random() < 0.35:                               # abutting / overlapping on the read
            rs = max(0, (segs[-1][1] if segs else rs) + int(rng.integers(-40, 120)))
            re = max(rs + 1, re)
        span = max(1, re - rs + int(rng.integers(-30, 30)))
        u = rng.random()
        if kind == "tra" and u < 0.5 or u < 0.12:
            chrom = str(rng.choice(CHROMS)); ref = int(rng.integers(10_000, 5_000_000))
        if kind == "inv" and u < 0.6 or u > 0.9:
            strand = "-" if strand == "+" else "+"
        step = {"del": int(rng.choice([0, 40, 300, 5000, 200000])), "ins": int(rng.integers(-20, 20)), "dup": -int(rng.choice([0, 50, 800, 20000])),
                "inv": int(rng.integers(-3000, 3000)), "tra": int(rng.integers(-100, 100)), "mix": int(rng.integers(-6000, 6000))}[kind]
        fs = max(0, ref + step)
        segs.append([rs, re, fs, fs + span, chrom, strand])
        ref = fs + span
    order = rng.permutation(n)                                # the SA tag lists the other alignments in no particular order
    segs = [segs[i] for i in order]
    mapq = [int(rng.choice([0, 5, 20, 60, 60])) for _ in segs]
    # the first one is the record itself (primary_info when its mapq passes)
    p = segs[0]
    primary = list(p) if mapq[0] >= 20 else []
    sa = ""
    for s, q in zip(segs[1:], mapq[1:]):
        rs, re, fs, fe, ch, st = s
        c0, c1 = (rs, L - re) if st == "+" else (L - re, rs)
        text, _, _ = sa_cigar(rng, c0, fe - fs, c1)
        sa += "%s,%d,%s,%s,%d,%d;" % (ch, fs + 1, st, text, q, int(rng.integers(0, 50)))
    return dict(name=name, primary=primary, sa=sa, qlen=L, key=key)


def ins_inside_tra_read(rng, name, key):
    """first and last segment continue each other on one chromosome, the middle of the read maps elsewhere: the rule at the
    end of analysis_split_read (main script :436-460) that reports the middle as an insertion"""
    L = int(rng.integers(3000, 12000))
    a = int(rng.integers(300, L // 3)); b = int(rng.integers(2 * L // 3, L - 300))
    mid = [(a + int(rng.integers(0, 30)), b - int(rng.integers(0, 30)))]
    if rng.random() < 0.4:                                    # two middle pieces
        m = (mid[0][0] + mid[0][1]) // 2
        mid = [(mid[0][0], m), (m + int(rng.integers(0, 20)), mid[0][1])]
    chrom = str(rng.choice(CHROMS)); other = str(rng.choice([c for c in CHROMS if c != chrom])); strand = str(rng.choice(["+", "-"]))
    ref = int(rng.integers(100_000, 5_000_000))
    gap = int(rng.choice([0, 3, -3, 25, -25, 200, -400, -5000]))
    first = [0, a, ref, ref + a, chrom, strand]
    last = [b, L, ref + a + gap, ref + a + gap + (L - b), chrom, strand]
    if strand == "-":                                         # on the reverse strand the read runs down the reference
        first, last = [0, a, ref + a + gap, ref + a + gap + a, chrom, strand], [b, L, ref - (L - b) + a, ref + a, chrom, strand]
    segs = [first] + [[rs, re, int(rng.integers(10_000, 5_000_000)), 0, other, str(rng.choice(["+", "-"]))] for rs, re in mid] + [last]
    for s in segs[1:-1]:
        s[3] = s[2] + (s[1] - s[0])
    order = rng.permutation(len(segs))
    segs = [segs[i] for i in order]
    mapq = [60] + [int(rng.choice([20, 60, 60, 5])) for _ in segs[1:]]
    sa = ""
    for s, q in zip(segs[1:], mapq[1:]):
        rs, re, fs, fe, ch, st = s
        c0, c1 = (rs, L - re) if st == "+" else (L - re, rs)
        text, _, _ = sa_cigar(rng, c0, fe - fs, c1)
        sa += "%s,%d,%s,%s,%d,%d;" % (ch, fs + 1, st, text, q, 0)
    return dict(name=name, primary=list(segs[0]), sa=sa, qlen=L, key=key)


def split_golden(main):
    cases = []
    for name, seed, n, kind, params in (("deletions", 21, 300, "del", dict(sv=30, max_size=100000, min_mapq=20, parts=7)),
                                        ("insertions", 22, 300, "ins", dict(sv=30, max_size=100000, min_mapq=20, parts=7)),
                                        ("duplications", 23, 300, "dup", dict(sv=30, max_size=100000, min_mapq=20, parts=7)),
                                        ("inversions", 24, 300, "inv", dict(sv=30, max_size=100000, min_mapq=20, parts=7)),
                                        ("translocations", 25, 300, "tra", dict(sv=30, max_size=100000, min_mapq=20, parts=7)),
                                        ("mixture", 26, 600, "mix", dict(sv=30, max_size=100000, min_mapq=20, parts=7)),
                                        ("no_limits", 27, 300, "mix", dict(sv=1, max_size=-1, min_mapq=0, parts=-1)),
                                        ("tight", 28, 300, "del", dict(sv=500, max_size=3000, min_mapq=30, parts=3)),
                                        ("ins_inside_tra", 29, 200, "instra", dict(sv=30, max_size=100000, min_mapq=20, parts=7))):
        rng = np.random.default_rng(seed)
        reads = [(ins_inside_tra_read(rng, "sr%05d" % i, seed * 100000 + i) if kind == "instra" else
                  random_split_read(rng, "sr%05d" % i, seed * 100000 + i, kind)) for i in range(n)]
        cand = {t: [] for t in ("DEL", "INS", "DUP", "INV", "TRA")}
        for r in reads:
            query = synth.pseudo_sequence(r["qlen"], r["key"])
            main.organize_split_signal(list(r["primary"]), r["sa"].split(";")[:-1], r["qlen"], params["sv"], params["min_mapq"], params["parts"],
                                       r["name"], cand, params["max_size"], query)
        cases.append(dict(name=name, params=params, reads=reads, **{t: [list(x) for x in cand[t]] for t in cand}))
    with gzip.open(os.path.join(HERE, "split_sigs.json.gz"), "wt") as f:
        json.dump(cases, f)
    print("split_sigs.json.gz: %d cases; candidates %s" % (len(cases), {t: sum(len(c[t]) for c in cases) for t in ("DEL", "INS", "DUP", "INV", "TRA")}))
