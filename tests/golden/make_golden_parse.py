"""parse_reads.json.gz: the reference's parse_read (main script :606-681) called read after read on stub records that
carry CIGARs AND SA tags (see make_golden_main.py, which imports the main script and calls parse_golden): the whole
extraction of a task minus BAM decode."""
import gzip
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from cutesv_amd import synth                                  # noqa: E402
from make_golden_split import CHROMS, sa_cigar                # noqa: E402


class _Read:
    def __init__(self, d):
        self.query_name, self.flag, self.mapq, self.reference_start = d["name"], d["flag"], d["mapq"], d["start"]
        self.cigartuples = self.cigar = [tuple(x) for x in d["cigar"]]
        self.query_sequence = synth.pseudo_sequence(d["seq_len"], d["seq_key"])
        self.query_length = d["seq_len"]
        self.reference_end = d["start"] + sum(l for op, l in self.cigartuples if op in (0, 2, 3, 7, 8))
        self._tags = [tuple(t) for t in d["tags"]]

    def get_tags(self):
        return self._tags


def random_record(rng, name, key):
    """a primary / supplementary / secondary record: clips, a few large and many small indels, and (for most primary
    records) an SA tag naming one to four other alignments of the read"""
    flag = int(rng.choice([0, 0, 0, 16, 16, 2048, 2064, 256, 4]))
    left = int(rng.integers(0, 3000)) if rng.random() < 0.7 else 0
    right = int(rng.integers(0, 3000)) if rng.random() < 0.7 else 0
    hard = rng.random() < 0.15
    ops = []
    if left:
        ops.append((5 if hard else 4, left))
    for _ in range(int(rng.integers(1, 40))):
        u = rng.random()
        if u < 0.5:
            ops.append((int(rng.choice([0, 7, 8])), int(rng.integers(1, 600))))
        elif u < 0.75:
            ops.append((1, int(rng.integers(30, 300)) if rng.random() < 0.2 else int(rng.integers(1, 10))))
        else:
            ops.append((2, int(rng.integers(30, 300)) if rng.random() < 0.2 else int(rng.integers(1, 10))))
    if right:
        ops.append((5 if hard else 4, right))
    qlen = sum(l for op, l in ops if op in (0, 1, 4, 7, 8))               # what pysam reports as query_length
    seq_len = qlen + (sum(l for op, l in ops if op == 5) if hard else 0) + 8  # (long enough for the reference's shift counter)
    tags = [("NM", int(rng.integers(0, 50)))]
    if flag in (0, 16) and rng.random() < 0.8 or rng.random() < 0.1:
        sa = ""
        for _ in range(int(rng.choice([1, 1, 2, 3, 4, 8]))):
            ch = "7" if rng.random() < 0.6 else str(rng.choice(CHROMS))
            c0 = int(rng.integers(0, max(1, seq_len - 10))); span = int(rng.integers(50, 4000)); c1 = max(0, seq_len - c0 - span + int(rng.integers(-40, 40)))
            text, _, _ = sa_cigar(rng, c0, span, c1)
            sa += "%s,%d,%s,%s,%d,%d;" % (ch, int(rng.integers(1, 3_000_000)), str(rng.choice(["+", "-"])), text, int(rng.choice([0, 10, 20, 60])), 3)
        tags.append(("SA", sa))
    return dict(name=name, flag=flag, mapq=int(rng.choice([0, 10, 20, 60, 60])), start=int(rng.integers(0, 3_000_000)), cigar=[list(x) for x in ops],
                seq_len=seq_len, seq_key=key, tags=[list(t) for t in tags])


def parse_golden(main):
    cases = []
    for name, seed, n, params in (("defaults", 31, 500, dict(sv=30, min_mapq=20, parts=7, min_read_len=500, min_siglength=10, md=0, mi=100, max_size=100000)),
                                  ("loose", 32, 400, dict(sv=1, min_mapq=0, parts=-1, min_read_len=0, min_siglength=1, md=50, mi=50, max_size=-1)),
                                  ("strict", 33, 400, dict(sv=100, min_mapq=30, parts=3, min_read_len=3000, min_siglength=30, md=500, mi=500, max_size=2000))):
        rng = np.random.default_rng(seed)
        recs = [random_record(rng, "pr%05d" % i, seed * 100000 + i) for i in range(n)]
        cand = {t: [] for t in ("DEL", "INS", "DUP", "INV", "TRA")}
        for d in recs:
            main.parse_read(_Read(d), cand, "7", params["sv"], params["min_mapq"], params["parts"], params["min_read_len"], params["min_siglength"],
                            params["md"], params["mi"], params["max_size"])
        cases.append(dict(name=name, params=params, chrom="7", chroms=sorted(set(CHROMS) | {"7"}), reads=recs,
                          **{t: [list(x) for x in cand[t]] for t in cand}))
    with gzip.open(os.path.join(HERE, "parse_reads.json.gz"), "wt") as f:
        json.dump(cases, f)
    print("parse_reads.json.gz: %d cases; candidates %s" % (len(cases), {t: sum(len(c[t]) for c in cases) for t in ("DEL", "INS", "DUP", "INV", "TRA")}))


def single_pipe_golden(main):
    """single_pipe.json.gz: the reference's single_pipe (main script :697-743) on a stub alignment file - the gates of the task
    loop (secondary records, the task's start coordinate, --include_bed regions) and the reads table rows next to the
    candidates.  The function pickles into <temp>/signatures/<pid><TYPE>.pickle: read back here."""
    import pickle
    import tempfile

    class _Sam:
        def __init__(self, reads):
            self.reads = reads

        def fetch(self, chrom, s, e):
            return iter(self.reads)

    cases = []
    for name, seed, n, task, bed, params in (("plain", 41, 300, ["7", 1_000_000, 2_000_000], None, dict(sv=30, min_mapq=20, parts=7, min_read_len=500, min_siglength=10, md=0, mi=100, max_size=100000)),
                                             ("bed", 42, 300, ["7", 500_000, 3_000_000], [[600_000, 900_000], [1_500_000, 2_200_000]],
                                              dict(sv=30, min_mapq=10, parts=-1, min_read_len=0, min_siglength=10, md=50, mi=50, max_size=-1))):
        rng = np.random.default_rng(seed)
        recs = sorted((random_record(rng, "sp%05d" % i, seed * 100000 + i) for i in range(n)), key=lambda d: d["start"])
        main.samfile = _Sam([_Read(d) for d in recs])
        with tempfile.TemporaryDirectory() as tmp:
            tmp += "/"
            os.mkdir(tmp + "signatures")
            main.single_pipe("stub.bam", params["sv"], params["min_mapq"], params["parts"], params["min_read_len"], tmp, task, params["min_siglength"],
                             params["md"], params["mi"], params["max_size"], bed)
            out = {}
            for fn in os.listdir(tmp + "signatures"):
                for t in ("DEL", "INS", "DUP", "INV", "TRA", "reads"):
                    if fn.endswith(t + ".pickle"):
                        with open(tmp + "signatures/" + fn, "rb") as f:
                            out["reads_table" if t == "reads" else t] = [list(x) for x in pickle.load(f)]
        main.samfile = None
        cases.append(dict(name=name, params=params, task=task, bed=bed, chroms=sorted(set(CHROMS) | {"7"}), reads=recs, **out))
    with gzip.open(os.path.join(HERE, "single_pipe.json.gz"), "wt") as f:
        json.dump(cases, f)
    print("single_pipe.json.gz: %d cases; reads rows %s, candidates %s" % (len(cases), [len(c["reads_table"]) if "reads_table" in c else len(c["reads"]) for c in cases],
                                                                            {t: sum(len(c[t]) for c in cases) for t in ("DEL", "INS", "DUP", "INV", "TRA")}))
