"""cigar_sigs.json.gz: the reference's parse_read (main script :606-681) driven with stub read objects (see
make_golden_main.py, which imports the main script and calls cigar_golden)."""
import gzip
import json
import os

import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from cutesv_amd import synth                                  # noqa: E402  (pseudo_sequence: the fixtures store its key, not the bases)


class _Read:
    """what parse_read touches of a pysam.AlignedSegment"""

    def __init__(self, name, flag, mapq, start, cig, seq):
        self.query_name, self.flag, self.mapq, self.reference_start = name, flag, mapq, start
        self.cigartuples = cig
        self.cigar = cig
        self.query_sequence = seq
        self.query_length = len(seq)
        self.reference_end = start + sum(l for op, l in cig if op in (0, 2, 3, 7, 8))

    def get_tags(self):
        return []                                      # no SA tag: the split-read analysis is not on this path


def random_read(rng, name, kind, key):
    """kind: shapes the operation mix (ordinary / bursts of large indels / clips and exotic operations)"""
    ops = []
    n = int(rng.integers(1, 60 if kind != "long" else 400))
    if kind == "clips" and rng.random() < 0.7:
        ops.append((int(rng.choice([4, 5])), int(rng.integers(1, 300))))
    for _ in range(n):
        u = rng.random()
        if u < 0.45:
            ops.append((int(rng.choice([0, 7, 8])), int(rng.integers(1, 400))))
        elif u < 0.70:
            big = rng.random() < (0.5 if kind == "bursts" else 0.15)
            ops.append((1, int(rng.integers(10, 400)) if big else int(rng.integers(1, 10))))
        elif u < 0.95:
            big = rng.random() < (0.5 if kind == "bursts" else 0.15)
            ops.append((2, int(rng.integers(10, 400)) if big else int(rng.integers(1, 10))))
        elif kind == "clips":
            ops.append((int(rng.choice([3, 6])), int(rng.integers(1, 50))))       # N and P: the quirks of the shift counter
        else:
            ops.append((0, int(rng.integers(1, 30))))
    if kind == "clips" and rng.random() < 0.7:
        ops.append((int(rng.choice([4, 5])), int(rng.integers(1, 300))))
    # the query holds every operation that "consumes" the reference's shift counter so that slices never run off its end
    qlen = sum(l for op, l in ops if op != 2) + 8
    seq = synth.pseudo_sequence(qlen, key)
    return _Read(name, int(rng.choice([0, 16, 2048, 256])), int(rng.choice([0, 10, 20, 60])), int(rng.integers(0, 10_000_000)), ops, seq)


def cigar_golden(main):
    cases = []
    for name, seed, n, kind, params in (("ordinary", 11, 300, "plain", dict(min_siglength=10, mi=100, md=0)),
                                        ("bursts", 12, 300, "bursts", dict(min_siglength=10, mi=100, md=0)),
                                        ("bursts_md500", 13, 300, "bursts", dict(min_siglength=10, mi=1000, md=500)),
                                        ("clips_and_skips", 14, 300, "clips", dict(min_siglength=10, mi=100, md=100)),
                                        ("long_reads", 15, 40, "long", dict(min_siglength=30, mi=50, md=50)),
                                        ("siglength_1", 16, 100, "bursts", dict(min_siglength=1, mi=0, md=0))):
        rng = np.random.default_rng(seed)
        reads = [random_read(rng, "rd%05d" % i, kind, seed * 100000 + i) for i in range(n)]
        min_mapq, min_read_len = 20, 500
        cand = {t: [] for t in ("DEL", "INS", "DUP", "INV", "TRA")}
        for r in reads:
            main.parse_read(r, cand, "chr7", 30, min_mapq, 7, min_read_len, params["min_siglength"], params["md"], params["mi"], 100000)
        assert not cand["DUP"] and not cand["INV"] and not cand["TRA"]
        cases.append(dict(name=name, params=dict(params, min_mapq=min_mapq, min_read_len=min_read_len),
                          reads=[dict(name=r.query_name, flag=r.flag, mapq=r.mapq, start=r.reference_start, cigar=[list(x) for x in r.cigartuples],
                                      seq_len=len(r.query_sequence), seq_key=seed * 100000 + i) for i, r in enumerate(reads)],
                          INS=[list(x) for x in cand["INS"]], DEL=[list(x) for x in cand["DEL"]]))
    with gzip.open(os.path.join(HERE, "cigar_sigs.json.gz"), "wt") as f:
        json.dump(cases, f)
    print("cigar_sigs.json.gz: %d cases, %d INS + %d DEL signatures" % (len(cases), sum(len(c["INS"]) for c in cases), sum(len(c["DEL"]) for c in cases)))
