#!/usr/bin/env python3
"""Golden vectors from the reference's MAIN SCRIPT (/root/reference/src/cuteSV/cuteSV), produced by running it here.

    python tests/golden/make_golden_main.py        # needs /root/reference (read-only mount)

The extension-less main script is imported through SourceFileLoader with stub `pysam`, `cigar` and `Bio.Seq` modules
(nothing on the two paths below touches them beyond attribute look-ups on the objects we hand in).  Nothing of the
reference is copied: the outputs are data - the inputs we synthesise and what the reference's functions return.

Outputs
    rebuild_order.json.gz   process_process_sigs_type (main script :750-857) + remove_duplicates_sorted (:958-969) driven on
                            per-worker signature pickles with duplicates across workers, adversarial read-name orders,
                            x.5 INS positions and same-key INS rows that differ only in their sequence: the per-chromosome
                            lists the rebuild step writes, in its order
    cigar_sigs.json.gz      parse_read (main script :606-681) / generate_combine_sigs (:515-575) driven with stub read
                            objects carrying BAM-encoded CIGARs: the INS / DEL signatures each read yields (8f row 4)
    parse_reads.json.gz     parse_read (main script :606-681) called read after read on stub records with CIGARs and SA tags:
                            what a whole task's extraction appends to the five candidate lists (8f row 4)
    split_sigs.json.gz      organize_split_signal / analysis_split_read (main script :50-513) driven with synthetic primary
                            alignments and SA-tag texts: the candidates of all five SV types each read yields (8f row 4)
"""
import gzip
import importlib.machinery
import importlib.util
import json
import os
import pickle
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def load_main():
    for name in ("pysam", "cigar", "Bio", "Bio.Seq"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    # BAM CIGAR operation codes (SAM specification section 4.2; pysam exports them under these names)
    for code, name in enumerate(("CMATCH", "CINS", "CDEL", "CREF_SKIP", "CSOFT_CLIP", "CHARD_CLIP", "CPAD", "CEQUAL", "CDIFF", "CBACK")):
        setattr(sys.modules["pysam"], name, code)

    class Seq(str):                                          # Bio.Seq.Seq stand-in (reverse_complement is not on these paths)
        def reverse_complement(self):
            return Seq(self[::-1].translate(str.maketrans("ACGTacgt", "TGCAtgca")))
    sys.modules["Bio.Seq"].Seq = Seq
    sys.modules["Bio"].Seq = sys.modules["Bio.Seq"]

    class Cigar:                                             # cigar.Cigar stand-in: items() -> (length, operation) of a CIGAR text
        def __init__(self, text):
            self.text = text

        def items(self):
            import re
            return ((int(n), op) for n, op in re.findall(r"(\d+)([MIDNSHP=XB])", self.text))
    sys.modules["cigar"].Cigar = Cigar
    sys.path.insert(0, os.path.join(REF, "src"))
    loader = importlib.machinery.SourceFileLoader("cutesv_main", os.path.join(REF, "src", "cuteSV", "cuteSV"))
    spec = importlib.util.spec_from_loader("cutesv_main", loader)
    mod = importlib.util.module_from_spec(spec)
    loader.exec_module(mod)
    return mod


# ------------------------------------------------------------------------------------------------ rebuild order
NAMES = ["r9", "r10", "r100", "R10", "read/1", "read/10", "read/2", "a", "a0", "a_", "a-", "aa", "A", "Z", "z",
         "m64011_190830_220126/1/ccs", "m64011_190830_220126/10/ccs", "m64011_190830_220126/2/ccs",
         "0b1", "00b1", "~tilde", "!bang", "dup", "dup.1"]


def rebuild_case(seed, n_each, chroms, n_pid=3):
    rng = np.random.default_rng(seed)
    names = NAMES + ["q%04d" % i for i in range(60)]
    pick = lambda: names[int(rng.integers(0, len(names)))]
    seqs = ["ACGT" * 5, "ACGA" * 5, "TTTT" * 5, "ACGT" * 5 + "A", "A" * 20]
    cand = {t: [] for t in ("DEL", "INS", "DUP", "INV", "TRA", "reads")}
    for _ in range(n_each):
        ch = chroms[int(rng.integers(0, len(chroms)))]
        pos = int(rng.integers(0, 40)) * 5
        cand["DEL"].append((pos, int(rng.choice([30, 30, 45, 60])), pick(), "DEL", ch))
        ipos = pos + (0.5 if rng.random() < 0.3 else 0.0) if rng.random() < 0.6 else pos      # split-read INS: (a + b) / 2
        cand["INS"].append((ipos, int(rng.choice([50, 50, 20])), pick(), seqs[int(rng.integers(0, len(seqs)))], "INS", ch))
        cand["DUP"].append((pos, pos + int(rng.choice([500, 500, 900])), pick(), "DUP", ch))
        cand["INV"].append((str(rng.choice(["++", "--"])), pos, pos + int(rng.choice([700, 700, 1200])), pick(), "INV", ch))
        cand["TRA"].append((str(rng.choice(["A", "B", "C", "D"])), pos, chroms[int(rng.integers(0, len(chroms)))],
                            int(rng.integers(0, 30)) * 10, pick(), "TRA", ch))
        cand["reads"].append((pos, pos + int(rng.integers(100, 3000)), int(rng.random() < 0.8), pick(), ch))
    # same (chr, int(pos), len, read) INS rows that differ only in the sequence / in the half position
    for k in range(6):
        ch, nm = chroms[k % len(chroms)], names[k]
        cand["INS"] += [(77, 50, nm, "TTTT" * 5, "INS", ch), (77.5, 50, nm, "ACGT" * 5, "INS", ch), (77, 50, nm, "ACGT" * 5, "INS", ch),
                        (77.0, 50, nm, "ACGT" * 5, "INS", ch)]
    # overlapping extraction windows: exact duplicates, in the same and in other workers' files
    for t in cand:
        if t == "reads":
            continue
        d = [cand[t][int(i)] for i in rng.integers(0, len(cand[t]), max(2, len(cand[t]) // 6))]
        cand[t] += d + d[: len(d) // 2]
    files = []                                               # per worker: per type: list of batches (one pickle.dump each)
    for p in range(n_pid):
        files.append({t: [] for t in cand})
    for t, lst in cand.items():
        order = rng.permutation(len(lst))
        cuts = np.sort(rng.integers(0, len(lst) + 1, 2 * n_pid - 1))
        parts = np.split(order, cuts)
        for bi, part in enumerate(parts):
            files[bi % n_pid][t].append([lst[int(i)] for i in part])
    return files


def run_rebuild(main, files):
    pids = [1000 + i for i in range(len(files))]
    out = {}
    with tempfile.TemporaryDirectory() as d:
        d = d + "/"
        os.mkdir(d + "signatures")
        for pid, f in zip(pids, files):
            for t, batches in f.items():
                with open("%ssignatures/%s%s.pickle" % (d, pid, t), "wb") as fh:
                    for b in batches:
                        pickle.dump(b, fh)
        for t in ("DEL", "INS", "DUP", "INV", "TRA", "reads"):
            sv, index, reads_count = main.process_process_sigs_type((t, d, pids, False))
            per_chr = []
            with open("%s%s.pickle" % (d, t), "rb") as fh:
                for ch, off in sorted(index.items(), key=lambda kv: kv[1]):
                    fh.seek(off)
                    per_chr.append([ch, [list(x) for x in pickle.load(fh)]])
            out[t] = per_chr
            if t == "reads":
                out["reads_count"] = reads_count
    return out


def rebuild_golden(main):
    cases = []
    for name, seed, n, chroms in (("three_chroms", 1, 120, ["1", "10", "2"]), ("one_chrom_dense", 2, 300, ["chrX"]),
                                  ("names_mix", 3, 200, ["chr1", "chr10", "chr2", "chrM"])):
        files = rebuild_case(seed, n, chroms)
        cases.append(dict(name=name, files=[{t: [[list(x) for x in b] for b in bs] for t, bs in f.items()} for f in files],
                          out=run_rebuild(main, files)))
    with gzip.open(os.path.join(HERE, "rebuild_order.json.gz"), "wt") as f:
        json.dump(cases, f)
    print("rebuild_order.json.gz: %d cases, %d rows out" % (len(cases), sum(len(r) for c in cases for t in c["out"] if t != "reads_count" for _, r in c["out"][t])))


def main_():
    main = load_main()
    rebuild_golden(main)
    if "--cigar" in sys.argv or True:
        try:
            from make_golden_cigar import cigar_golden
            cigar_golden(main)
            from make_golden_split import split_golden
            split_golden(main)
            from make_golden_parse import parse_golden, single_pipe_golden
            parse_golden(main)
            single_pipe_golden(main)
        except ImportError:
            pass


if __name__ == "__main__":
    sys.path.insert(0, HERE)
    main_()
