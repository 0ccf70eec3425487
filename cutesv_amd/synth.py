"""Seeded synthetic signature sets of the shapes BASELINE.json's configs name.

The reference ships no signature data (only VISOR truth beds), so every workload here is
builder-defined and documented: a set of true SV sites with read support drawn around them,
uniform singleton noise, small noisy loci, and (for genotyping) a reads table.  Everything is
produced directly in the flat column layout of `columns.SigStore`; numpy's PCG64 stream makes
the arrays identical in the build container and on the GPU box.

Contig lengths are hg19's (the values in the reference's simulation/LASeR.bed:1-24).
"""
import numpy as np

from .columns import SigStore, NameTable, Params, TYPES

CONTIGS = [("1", 249250621), ("2", 243199373), ("3", 198022430), ("4", 191154276), ("5", 180915260),
           ("6", 171115067), ("7", 159138663), ("8", 146364022), ("9", 141213431), ("10", 135534747),
           ("11", 135006516), ("12", 133851895), ("13", 115169878), ("14", 107349540), ("15", 102531392),
           ("16", 90354753), ("17", 81195210), ("18", 78077248), ("19", 59128983), ("20", 63025520),
           ("21", 48129895), ("22", 51304566), ("X", 155270560), ("Y", 59373566)]
# allele fraction per contig in LASeR.bed column 5 (50 -> het, 100 -> hom), same order
CONTIG_AF = [0.5, 1.0, 0.5, 0.5, 1.0, 0.5, 1.0, 0.5, 1.0, 0.5, 1.0, 0.5, 1.0, 0.5, 1.0, 0.5, 1.0, 0.5, 1.0, 0.5,
             1.0, 0.5, 1.0, 1.0]


def _site_lengths(rng, n, lo=30, hi=6000):
    """log-uniform [lo, hi] with +10 % mass at 300±15 (Alu) and +3 % at 6000±100 (L1)."""
    ln = np.exp(rng.uniform(np.log(lo), np.log(hi), n))
    u = rng.random(n)
    alu = u < 0.10 / 1.13
    l1 = (u >= 0.10 / 1.13) & (u < 0.13 / 1.13)
    ln[alu] = rng.normal(300, 15, alu.sum())
    ln[l1] = rng.normal(6000, 100, l1.sum())
    return np.maximum(np.rint(ln), lo).astype(np.int64)


def _place(rng, n, contigs):
    """n positions placed proportionally to contig length -> (chrom index, position)."""
    lens = np.array([l for _, l in contigs], dtype=np.float64)
    ch = rng.choice(len(contigs), size=n, p=lens / lens.sum())
    pos = (rng.random(n) * (lens[ch] - 200000)).astype(np.int64) + 100000
    return ch.astype(np.int64), pos


class _Cols:
    """accumulates unsorted signatures of one SV type"""

    def __init__(self):
        self.ch, self.a, self.b, self.rid, self.aux = [], [], [], [], []

    def add(self, ch, a, b, rid, aux=None):
        n = len(a)
        self.ch.append(np.asarray(ch, np.int64)); self.a.append(np.asarray(a, np.int64))
        self.b.append(np.asarray(b, np.int64)); self.rid.append(np.asarray(rid, np.int64))
        self.aux.append(np.zeros(n, np.int64) if aux is None else np.asarray(aux, np.int64))

    def cat(self):
        if not self.a:
            z = np.zeros(0, np.int64)
            return z, z, z, z, z
        return tuple(np.concatenate(x) for x in (self.ch, self.a, self.b, self.rid, self.aux))


def _indel_type(rng, contigs, n_sites, coverage, pos_sigma, len_sigma, n_noise, n_loci, next_read,
                dup_frac=0.03, min_len=10, sites=None, noise_len=(10, 40)):
    """Signatures of one INDEL type: true sites + singleton noise + small noisy loci."""
    cols = _Cols()
    if sites is None:
        s_ch, s_pos = _place(rng, n_sites, contigs)
        s_len = _site_lengths(rng, n_sites)
        hom = rng.random(n_sites) < 1.0 / 3.0
        af = np.where(hom, 0.95, 0.5)
    else:
        s_ch, s_pos, s_len, af = sites
        n_sites = len(s_pos)
    k = rng.binomial(coverage, af)
    tot = int(k.sum())
    site_of = np.repeat(np.arange(n_sites), k)
    pos = s_pos[site_of] + np.rint(rng.normal(0, pos_sigma, tot)).astype(np.int64)
    ln = np.maximum(min_len, np.rint(s_len[site_of] * (1 + rng.normal(0, len_sigma, tot)))).astype(np.int64)
    rid = next_read + np.arange(tot)
    cols.add(s_ch[site_of], np.maximum(pos, 1), ln, rid, ln)
    # a few reads carry a second, shorter signature of the same event (exercises the per-read de-duplication)
    nd = int(tot * dup_frac)
    if nd:
        pick = rng.choice(tot, nd, replace=False)
        cols.add(s_ch[site_of][pick], np.maximum(pos[pick] + rng.integers(-20, 21, nd), 1),
                 np.maximum(min_len, (ln[pick] * rng.uniform(0.3, 0.9, nd)).astype(np.int64)), rid[pick],
                 np.maximum(min_len, (ln[pick] * 0.5).astype(np.int64)))
    next_read += tot
    if n_noise:
        ch, p = _place(rng, n_noise, contigs)
        ln = rng.integers(noise_len[0], noise_len[1] + 1, n_noise)
        cols.add(ch, p, ln, next_read + np.arange(n_noise), ln)
        next_read += n_noise
    if n_loci:
        ch, p = _place(rng, n_loci, contigs)
        kk = rng.integers(3, 9, n_loci)
        t = int(kk.sum())
        lo = np.repeat(np.arange(n_loci), kk)
        ln = rng.integers(10, 81, t)
        cols.add(ch[lo], np.maximum(p[lo] + np.rint(rng.normal(0, 40, t)).astype(np.int64), 1), ln,
                 next_read + np.arange(t), ln)
        next_read += t
    return cols, next_read


def _finish(contigs, per_type, n_reads_total, rng, reads=None, shuffle_ids=True):
    """sort each type into the reference's file order and assemble the SigStore"""
    chroms = [c for c, _ in contigs]
    perm = rng.permutation(n_reads_total).astype(np.int64) if shuffle_ids else np.arange(n_reads_total)
    A, B, R, X, seg_index = [], [], [], [], {}
    n = 0
    for t in TYPES:
        if t not in per_type:
            continue
        ch, a, b, rid, aux = per_type[t].cat()
        if len(a) == 0:
            continue
        rid = perm[rid]
        if t == "INV":      # (chr, strand, int(p1), p2, read)   main script :792
            order = np.lexsort((rid, b, a, aux, ch))
        elif t == "TRA":    # (chr1, chr2, type, int(p1), p2, read)  main script :801 ; aux = chr2*8 + type
            order = np.lexsort((rid, b, a, aux, ch))
        else:               # (chr, int(pos), len, read)  main script :764-783
            order = np.lexsort((rid, b, a, ch))
        ch, a, b, rid, aux = ch[order], a[order], b[order], rid[order], aux[order]
        keep = np.ones(len(a), bool)                      # adjacent exact duplicates (main script :958-969)
        keep[1:] = (ch[1:] != ch[:-1]) | (a[1:] != a[:-1]) | (b[1:] != b[:-1]) | (rid[1:] != rid[:-1]) | (aux[1:] != aux[:-1])
        ch, a, b, rid, aux = ch[keep], a[keep], b[keep], rid[keep], aux[keep]
        bounds = np.flatnonzero(np.r_[True, ch[1:] != ch[:-1], True])
        for i in range(len(bounds) - 1):
            seg_index[(t, chroms[int(ch[bounds[i]])])] = (n + int(bounds[i]), n + int(bounds[i + 1]))
        A.append(a); B.append(b); R.append(rid); X.append(aux)
        n += len(a)
    cat = lambda xs, dt: np.concatenate(xs).astype(dt) if xs else np.zeros(0, dt)
    kw = {}
    if reads is not None:
        r_ch, r_s, r_e, r_p, r_i = reads
        order = np.lexsort((r_s, r_ch))
        r_ch, r_s, r_e, r_p, r_i = r_ch[order], r_s[order], r_e[order], r_p[order], perm[r_i[order]]
        off = np.searchsorted(r_ch, np.arange(len(chroms) + 1)).astype(np.int64)
        kw = dict(reads_off=off, r_start=r_s.astype(np.int64), r_end=r_e.astype(np.int64),
                  r_primary=r_p.astype(np.uint8), r_id=r_i.astype(np.int32))
    return SigStore(chroms=chroms, a=cat(A, np.int64), b=cat(B, np.int64), read_id=cat(R, np.int32),
                    aux=cat(X, np.int32), seg_index=seg_index, names=NameTable(),
                    contig_len=np.array([l for _, l in contigs], np.int64), **kw)


def ont30(seed=20260103, scale=1.0, contigs=CONTIGS, coverage=30, ins_ratio=1.0):
    """cfg-3: HG002-shaped ONT 30x, INS + DEL, ~2.8 M signatures at scale 1."""
    rng = np.random.default_rng(seed)
    per, nr = {}, 0
    for t, mult in (("DEL", 1.0), ("INS", ins_ratio)):
        per[t], nr = _indel_type(rng, contigs, int(12000 * scale * mult), coverage, 12.0, 0.04,
                                 int(600000 * scale * mult), int(100000 * scale * mult), nr)
    return _finish(contigs, per, nr, rng)


def _reads_table(rng, contigs, coverage, mean_len, sd_len, lo, hi, lognormal=None):
    chs, starts, ends = [], [], []
    for ci, (_, L) in enumerate(contigs):
        n = int(coverage * L / mean_len)
        s = np.sort((rng.random(n) * L).astype(np.int64))
        if lognormal is None:
            ln = np.clip(rng.normal(mean_len, sd_len, n), lo, hi).astype(np.int64)
        else:
            ln = np.clip(rng.lognormal(np.log(lognormal[0]), lognormal[1], n), lo, hi).astype(np.int64)
        chs.append(np.full(n, ci, np.int64)); starts.append(s); ends.append(np.minimum(s + ln, L))
    return np.concatenate(chs), np.concatenate(starts), np.concatenate(ends)


def _support_from_reads(rng, r_ch, r_s, r_e, off, s_ch, s_pos, s_len, k, window):
    """for each site pick up to k reads spanning [pos - window, pos + len + window] -> (site index, read index)"""
    site_l, read_l = [], []
    for i in range(len(s_pos)):
        c = int(s_ch[i]); lo, hi = int(off[c]), int(off[c + 1])
        p0, p1 = int(s_pos[i]) - window, int(s_pos[i]) + int(s_len[i]) + window
        j1 = lo + int(np.searchsorted(r_s[lo:hi], p0, side="right"))
        j0 = max(lo, j1 - 400)
        cand = np.flatnonzero(r_e[j0:j1] >= p1) + j0
        if len(cand) == 0:
            continue
        kk = min(int(k[i]), len(cand))
        if kk <= 0:
            continue
        sel = rng.choice(cand, kk, replace=False)
        site_l.append(np.full(kk, i, np.int64)); read_l.append(sel)
    if not site_l:
        z = np.zeros(0, np.int64)
        return z, z
    return np.concatenate(site_l), np.concatenate(read_l)


def hifi30_gt(seed=20260104, scale=1.0, contigs=CONTIGS, coverage=30):
    """cfg-4: PacBio HiFi 30x with a reads table for --genotype: ~0.6 M signatures, ~6.2 M reads."""
    rng = np.random.default_rng(seed)
    if scale != 1.0:
        contigs = [(c, max(400000, int(l * scale))) for c, l in contigs]
    r_ch, r_s, r_e = _reads_table(rng, contigs, coverage, 15000, 2500, 5000, 25000)
    nR = len(r_s)
    off = np.searchsorted(r_ch, np.arange(len(contigs) + 1))
    per = {}
    n_sites = int(12000 * scale)
    for t in ("DEL", "INS"):
        s_ch, s_pos = _place(rng, n_sites, contigs)
        s_len = _site_lengths(rng, n_sites)
        hom = rng.random(n_sites) < 1.0 / 3.0
        k = rng.binomial(coverage, np.where(hom, 0.95, 0.5))
        si, ri = _support_from_reads(rng, r_ch, r_s, r_e, off, s_ch, s_pos, np.where(t == "DEL", s_len, 0), k, 50)
        cols = _Cols()
        tot = len(si)
        pos = s_pos[si] + np.rint(rng.normal(0, 2, tot)).astype(np.int64)
        ln = np.maximum(10, np.rint(s_len[si] * (1 + rng.normal(0, 0.005, tot)))).astype(np.int64)
        cols.add(s_ch[si], np.maximum(pos, 1), ln, ri, ln)
        nn = int(30000 * scale / 2)
        ch, p = _place(rng, nn, contigs)
        # noise signatures come from a read that starts shortly before the position
        j = np.array([int(off[c]) + max(0, int(np.searchsorted(r_s[off[c]:off[c + 1]], pp)) - 1) for c, pp in zip(ch, p)],
                     dtype=np.int64)
        lnn = rng.integers(30, 60, nn)
        cols.add(ch, p, lnn, j, lnn)
        per[t] = cols
    prim = (rng.random(nR) >= 0.03).astype(np.int64)       # 3 % supplementary records
    reads = (r_ch, r_s, r_e, prim, np.arange(nR, dtype=np.int64))
    return _finish(contigs, per, nR, rng, reads=reads)


def _pair_sites(rng, contigs, n_sites, size_lo, size_hi):
    ch, p = _place(rng, n_sites, contigs)
    size = np.exp(rng.uniform(np.log(size_lo), np.log(size_hi), n_sites)).astype(np.int64)
    return ch, p, p + size


def _pair_types(rng, contigs, coverage, next_read, n_dup, n_inv, n_tra, sites=None):
    """DUP / INV / TRA signatures around true sites, plus a little noise (cfg-2 / cfg-5 recipe)."""
    per = {}
    nch = len(contigs)
    # DUP: (start + N(0,20), end + N(0,20))
    if sites is not None and "DUP" in sites:
        ch, p1, p2, af = sites["DUP"]
    else:
        ch, p1, p2 = _pair_sites(rng, contigs, n_dup, 500, 50000); af = np.full(n_dup, 0.5)
    k = rng.binomial(coverage, af); tot = int(k.sum()); so = np.repeat(np.arange(len(p1)), k)
    c = _Cols()
    c.add(ch[so], np.maximum(p1[so] + np.rint(rng.normal(0, 20, tot)).astype(np.int64), 1),
          p2[so] + np.rint(rng.normal(0, 20, tot)).astype(np.int64), next_read + np.arange(tot))
    next_read += tot
    nn = max(10, tot // 10)
    nc, np1 = _place(rng, nn, contigs)
    c.add(nc, np1, np1 + rng.integers(100, 20000, nn), next_read + np.arange(nn)); next_read += nn
    per["DUP"] = c
    # INV: two strands ++/-- alternating per read
    if sites is not None and "INV" in sites:
        ch, p1, p2, af = sites["INV"]
    else:
        ch, p1, p2 = _pair_sites(rng, contigs, n_inv, 500, 50000); af = np.full(n_inv, 0.5)
    k = rng.binomial(coverage, af); tot = int(k.sum()); so = np.repeat(np.arange(len(p1)), k)
    c = _Cols()
    c.add(ch[so], np.maximum(p1[so] + np.rint(rng.normal(0, 15, tot)).astype(np.int64), 1),
          p2[so] + np.rint(rng.normal(0, 15, tot)).astype(np.int64), next_read + np.arange(tot),
          np.arange(tot) % 2)
    next_read += tot
    per["INV"] = c
    # TRA: aux = chr2*8 + type code
    if sites is not None and "TRA" in sites:
        ch, p1, ch2, p2, ty, af = sites["TRA"]
    else:
        ch, p1 = _place(rng, n_tra, contigs)
        ch2 = (ch + 1 + rng.integers(0, nch - 1, n_tra)) % nch
        p2 = (rng.random(n_tra) * (np.array([l for _, l in contigs])[ch2] - 200000)).astype(np.int64) + 100000
        ty = rng.integers(0, 4, n_tra); af = np.full(n_tra, 0.5)
    k = rng.binomial(coverage, af); tot = int(k.sum()); so = np.repeat(np.arange(len(p1)), k)
    c = _Cols()
    c.add(ch[so], np.maximum(p1[so] + np.rint(rng.normal(0, 8, tot)).astype(np.int64), 1),
          np.maximum(p2[so] + np.rint(rng.normal(0, 8, tot)).astype(np.int64), 1), next_read + np.arange(tot),
          ch2[so] * 8 + ty[so])
    next_read += tot
    per["TRA"] = c
    return per, next_read


def ont90_all(seed=20260105, scale=1.0, contigs=CONTIGS, genotype_reads=True):
    """cfg-5: ONT ultra-long 90x, all five types, INS:DEL = 2:1, ~10 M signatures at scale 1."""
    rng = np.random.default_rng(seed)
    if scale != 1.0:
        contigs = [(c, max(400000, int(l * scale))) for c, l in contigs]
    per, nr = {}, 0
    for t, mult in (("DEL", 1.0), ("INS", 2.0)):
        per[t], nr = _indel_type(rng, contigs, int(12000 * scale * mult), 90, 12.0, 0.04,
                                 int(1500000 * scale * mult), int(250000 * scale * mult), nr)
    pp, nr = _pair_types(rng, contigs, 90, nr, int(3712 * scale), max(4, int(44 * scale)), max(8, int(380 * scale)))
    per.update(pp)
    reads = None
    if genotype_reads:
        r_ch, r_s, r_e = _reads_table(rng, contigs, 90, 100000, 0, 1000, 1000000,
                                      lognormal=(50000, 0.6))
        nR = len(r_s)
        # read ids of the reads table are fresh ids appended after the signature reads
        reads = (r_ch, r_s, r_e, np.ones(nR, np.int64), nr + np.arange(nR, dtype=np.int64))
        nr += nR
    return _finish(contigs, per, nr, rng, reads=reads)


def sim_all_types(sites, seed=20260102, contigs=CONTIGS, coverage=30, chroms=None):
    """cfg-2 (and cfg-1 with chroms=['1'], types DEL only): signatures synthesised around the truth
    sites of the reference's simulation/sim_*.bed.gz (committed as data in tests/golden/sim_sites.npz).

    sites: dict with arrays  del_ch, del_start, del_end / ins_ch, ins_pos, ins_len / dup_ch, dup_start, dup_end /
           inv_ch, inv_start, inv_end / tra_ch, tra_pos, tra_ch2, tra_pos2, tra_type   (chrom = index into CONTIGS)
    """
    rng = np.random.default_rng(seed)
    af = np.array(CONTIG_AF)
    per, nr = {}, 0

    def sel(prefix):
        ch = np.asarray(sites[prefix + "_ch"], np.int64)
        m = np.ones(len(ch), bool) if chroms is None else np.isin(ch, [i for i, (c, _) in enumerate(contigs) if c in chroms])
        return ch, m

    ch, m = sel("del")
    st = np.asarray(sites["del_start"], np.int64)[m]; ln = (np.asarray(sites["del_end"], np.int64) - np.asarray(sites["del_start"], np.int64))[m]
    per["DEL"], nr = _indel_type(rng, contigs, 0, coverage, 10.0, 0.03, 5000 if chroms else 50000, 0, nr,
                                 dup_frac=0.0, sites=(ch[m], st, ln, af[ch[m]]))
    if "ins_ch" in sites and (chroms is None):
        ch, m = sel("ins")
        per["INS"], nr = _indel_type(rng, contigs, 0, coverage, 10.0, 0.03, 50000, 0, nr, dup_frac=0.0,
                                     sites=(ch[m], np.asarray(sites["ins_pos"], np.int64)[m],
                                            np.asarray(sites["ins_len"], np.int64)[m], af[ch[m]]))
        ps = {}
        c1 = np.asarray(sites["dup_ch"], np.int64)
        ps["DUP"] = (c1, np.asarray(sites["dup_start"], np.int64), np.asarray(sites["dup_end"], np.int64), af[c1])
        c1 = np.asarray(sites["inv_ch"], np.int64)
        ps["INV"] = (c1, np.asarray(sites["inv_start"], np.int64), np.asarray(sites["inv_end"], np.int64), af[c1])
        c1 = np.asarray(sites["tra_ch"], np.int64)
        ps["TRA"] = (c1, np.asarray(sites["tra_pos"], np.int64), np.asarray(sites["tra_ch2"], np.int64),
                     np.asarray(sites["tra_pos2"], np.int64), np.asarray(sites["tra_type"], np.int64), af[c1])
        pp, nr = _pair_types(rng, contigs, coverage, nr, 0, 0, 0, sites=ps)
        per.update(pp)
    return _finish(contigs, per, nr, rng)


def small_mixed(seed, n_sites=40, coverage=30, genotype=True, n_contigs=3, contig_len=3_000_000, pos_sigma=12.0,
                len_sigma=0.04, dup_frac=0.1, n_noise=600, n_loci=60):
    """Small all-type store for parity tests (seconds on the CPU oracle, reference-runnable)."""
    rng = np.random.default_rng(seed)
    contigs = [(str(i + 1), contig_len) for i in range(n_contigs)]
    per, nr = {}, 0
    for t in ("DEL", "INS"):
        per[t], nr = _indel_type(rng, contigs, n_sites, coverage, pos_sigma, len_sigma, n_noise, n_loci, nr,
                                 dup_frac=dup_frac)
    pp, nr = _pair_types(rng, contigs, coverage, nr, max(4, n_sites // 3), max(3, n_sites // 4), max(4, n_sites // 3))
    per.update(pp)
    reads = None
    if genotype:
        r_ch, r_s, r_e = _reads_table(rng, contigs, coverage, 15000, 4000, 2000, 40000)
        nR = len(r_s)
        # half of the reads table reuses signature read ids so that support ∩ cover is non-trivial
        ids = np.where(rng.random(nR) < 0.5, rng.integers(0, max(nr, 1), nR), nr + np.arange(nR))
        prim = (rng.random(nR) >= 0.1).astype(np.int64)
        reads = (r_ch, r_s, r_e, prim, ids.astype(np.int64))
        nr += nR
    return _finish(contigs, per, nr, rng, reads=reads)


def reference_sequence(length, seed=0):
    """deterministic pseudo-reference for the VCF-emit tests: mostly ACGT with a sprinkle of IUPAC codes and N"""
    rng = np.random.default_rng(seed)
    alphabet = np.frombuffer(b"ACGT" * 12 + b"RYSWKMBDHVN", dtype=np.uint8)
    return alphabet[rng.integers(0, len(alphabet), length)].tobytes().decode()


def extraction_order(store, seed=7, region=10_000_000, workers=16):
    """The same store with its reads table in the order cuteSV's extraction + rebuild steps leave it (main script
    :697-735, :810, :1040-1093): the genome is cut into `region`-sized tasks, each pool worker appends the batches it
    happened to process to its own file (a batch = the reads that START in the region, in BAM order), and the rebuild
    step concatenates the workers' files and sorts stably by chromosome only.  Inside a chromosome the block is thus a
    permutation of disjoint, start-sorted runs.  Returns (store, row permutation applied)."""
    import dataclasses
    if store.reads_off is None:
        return store, None
    rng = np.random.default_rng(seed)
    off = store.reads_off
    perm = np.arange(store.n_reads, dtype=np.int64)
    for c in range(len(store.chroms)):
        lo, hi = int(off[c]), int(off[c + 1])
        if hi - lo < 2:
            continue
        task = store.r_start[lo:hi] // region                   # batch of every read
        n_task = int(task.max()) + 1
        worker = rng.integers(0, workers, n_task)                # which worker took the batch ...
        when = rng.permutation(n_task)                           # ... and when (order inside its file)
        key = worker[task] * n_task + when[task]
        perm[lo:hi] = lo + np.argsort(key, kind="stable")
    st = dataclasses.replace(store, r_start=store.r_start[perm], r_end=store.r_end[perm],
                             r_primary=store.r_primary[perm], r_id=store.r_id[perm])
    return st, perm


def concat_stores(x, y, suffix="b"):
    """Two synthetic stores as one genome: y's chromosomes are renamed (name + suffix) and its read ids shifted past
    x's.  Used to put a pathological locus (e.g. a 10 000x pile-up) next to ordinary ones in one batch."""
    shift = int(max(x.read_id.max(initial=-1), -1 if x.r_id is None else x.r_id.max(initial=-1))) + 1
    chroms = list(x.chroms) + [c + suffix for c in y.chroms]
    nx = len(x.chroms)
    y_aux = y.aux.copy()
    seg_index = dict(x.seg_index)
    for (t, c), (b, e) in y.seg_index.items():
        seg_index[(t, c + suffix)] = (b + x.n_sig, e + x.n_sig)
        if t == "TRA":                                   # aux = chr2 * 8 + type: chr2 moves with the renaming
            y_aux[b:e] = ((y.aux[b:e] >> 3) + nx) * 8 + (y.aux[b:e] & 7)
    kw = {}
    if x.reads_off is not None and y.reads_off is not None:
        kw = dict(reads_off=np.concatenate([x.reads_off, x.reads_off[-1] + y.reads_off[1:]]),
                  r_start=np.concatenate([x.r_start, y.r_start]), r_end=np.concatenate([x.r_end, y.r_end]),
                  r_primary=np.concatenate([x.r_primary, y.r_primary]), r_id=np.concatenate([x.r_id, y.r_id + shift]).astype(np.int32))
    cl = None if x.contig_len is None or y.contig_len is None else np.concatenate([x.contig_len, y.contig_len])
    return SigStore(chroms=chroms, a=np.concatenate([x.a, y.a]), b=np.concatenate([x.b, y.b]),
                    read_id=np.concatenate([x.read_id, y.read_id + shift]).astype(np.int32), aux=np.concatenate([x.aux, y_aux]).astype(np.int32),
                    seg_index=seg_index, names=NameTable(), contig_len=cl, **kw)


def pseudo_sequence(length, key):
    """deterministic pseudo-random ACGT string (fixtures store `key` instead of the bases)"""
    j = np.arange(length, dtype=np.uint64)
    h = (j * np.uint64(2654435761) + np.uint64(key) * np.uint64(40503) + (j >> np.uint64(3)) * np.uint64(97)) >> np.uint64(7)
    return np.array(list("ACGT"))[(h & np.uint64(3)).astype(np.int64)].astype("U1").tobytes().decode("utf-32-le") if length else ""


# ------------------------------------------------------------------------------------ inputs of the neighbouring steps (bench.py)
def unsorted_rows(store, seed=1, dup_frac=0.05):
    """The signatures of `store` as the extraction step leaves them for the rebuild step (main script :750-857): per type,
    rows in random order with `dup_frac` exact duplicates (overlapping extraction windows make them; INS rows are not
    duplicated: their tie groups are the host's business).  -> per_type dict for rebuild.store_from_unsorted"""
    rng = np.random.default_rng(seed)
    per = {}
    for (t, ch), (b, e) in store.seg_index.items():
        d = per.setdefault(t, dict(chrom=[], a=[], b=[], read_id=[], aux=[]))
        d["chrom"].append(np.full(e - b, store.chroms.index(ch))); d["a"].append(store.a[b:e]); d["b"].append(store.b[b:e])
        d["read_id"].append(store.read_id[b:e])
        d["aux"].append(store.aux[b:e] if t in ("INS", "INV", "TRA") else np.zeros(e - b, np.int32))
    for t, d in per.items():
        cols = {k: np.concatenate(v) for k, v in d.items()}
        n = len(cols["a"])
        if t == "INS" or dup_frac <= 0:
            perm = rng.permutation(n)
            per[t] = {k: v[perm] for k, v in cols.items()}
        else:
            dup = rng.integers(0, n, max(1, int(n * dup_frac)))
            perm = rng.permutation(n + len(dup))
            per[t] = {k: np.concatenate([v, v[dup]])[perm] for k, v in cols.items()}
    return per


def cigar_reads(n, seed=77, mean_ops=180):
    """BAM-encoded CIGARs of n long reads with ONT-like statistics: ~mean_ops operations per read (a 15-20 kb read at ~90 %
    identity has a few hundred), mostly matches and 1-9 base indels, a few per cent of indels of 30 bases or more, clips at
    the ends.  -> (cig_off, cigar, ref_start, use)"""
    rng = np.random.default_rng(seed)
    nops = np.minimum(rng.geometric(1.0 / mean_ops, n), 5000).astype(np.int64) + 2
    off = np.zeros(n + 1, np.int64); np.cumsum(nops, out=off[1:])
    tot = int(off[-1])
    op = rng.choice(np.array([0, 1, 2, 7, 8], np.uint32), tot, p=[0.50, 0.17, 0.17, 0.10, 0.06])
    ln = np.where(op == 0, rng.integers(5, 200, tot), np.where(rng.random(tot) < 0.03, rng.integers(30, 400, tot), rng.integers(1, 10, tot))).astype(np.uint32)
    first, last = off[:-1], off[1:] - 1
    op[first] = 4; ln[first] = rng.integers(0, 500, n); op[last] = 4; ln[last] = rng.integers(0, 500, n)      # soft clips
    cigar = ((ln << 4) | op).astype(np.uint32)
    start = rng.integers(0, 200_000_000, n).astype(np.int64)
    use = (rng.random(n) < 0.95).astype(np.uint8)
    return off, cigar, start, use


def split_reads(n, seed=78, n_chrom=6):
    """flat csv_split_in arrays of n reads with 0 .. 12 alignments each (a primary and SA entries that tile the read loosely,
    mostly on one chromosome and strand): every rule of analysis_split_read fires somewhere"""
    rng = np.random.default_rng(seed)
    n_ent = rng.choice(np.array([0, 1, 2, 2, 2, 3, 3, 3, 4, 4, 5, 6, 8, 12]), n)
    ent_off = np.zeros(n + 1, np.int64); np.cumsum(n_ent, out=ent_off[1:])
    ne = int(ent_off[-1])
    read = np.repeat(np.arange(n), n_ent)
    first = np.zeros(ne, bool); first[ent_off[:-1][n_ent > 0]] = True
    L = rng.integers(500, 30000, n).astype(np.int64)
    primary = (first & (rng.random(ne) < 0.8)).astype(np.uint8)
    base_st = np.repeat((rng.random(n) < 0.5).astype(np.uint8), n_ent)
    strand = np.where(rng.random(ne) < 0.7, base_st, (rng.random(ne) < 0.35).astype(np.uint8)).astype(np.uint8)
    base_ch = np.repeat(rng.integers(0, n_chrom, n), n_ent)
    chrom = np.where(rng.random(ne) < 0.8, base_ch, rng.integers(0, n_chrom, ne)).astype(np.int32)
    Lr = L[read]
    lo = (rng.random(ne) * Lr).astype(np.int64); hi = np.minimum(Lr, lo + 1 + (rng.random(ne) * Lr * 0.5).astype(np.int64))
    base_ref = np.repeat(rng.integers(100_000, 3_000_000, n), n_ent)
    ref = np.maximum(0, base_ref + np.where(rng.random(ne) < 0.7, lo + rng.integers(-3000, 3000, ne), rng.integers(-2_000_000, 2_000_000, ne))).astype(np.int64)
    span = np.maximum(1, hi - lo + rng.integers(-50, 50, ne)).astype(np.int64)
    c0 = np.where(primary == 1, lo, np.where(strand == 0, lo, Lr - hi)); c1 = np.where(primary == 1, hi, np.where(strand == 0, Lr - hi, lo))
    f1 = np.where(primary == 1, ref + span, span)
    mapq = rng.choice(np.array([0, 3, 20, 30, 60], np.int32), ne)
    return dict(ent_off=ent_off, read_len=L, c0=c0.astype(np.int64), c1=c1.astype(np.int64), f0=ref, f1=f1.astype(np.int64), chr=chrom, mapq=mapq,
                strand=strand, primary=primary)
