"""The reference genome for the VCF emit step, without pysam (cuteSV_genotype.py:254-259: generate_output opens
`pysam.FastaFile(ref)` and fetches whole chromosomes).

`Reference(path)` maps the FASTA file into memory (mmap: nothing is read until a base is looked at, nothing is copied into
Python strings - a human reference is 3.1 GB) and indexes it like `samtools faidx`: from `<path>.fai` when it sits next to
the file, else with one pass of the native indexer (`csv_fasta_index`, cutesv_amd/csrc/vcf_emit.cpp).  `csv_vcf_emit` then
reads REF / ALT bases straight out of the mapping through (line_bases, line_width) arithmetic (include/cutesv_hip.h
csv_vcf_in.chrom_line_bases).  gzip-compressed FASTA cannot be mapped: `read_fasta` (whole contigs as str, round 1's reader)
remains for it and for tests.
"""
import ctypes as C
import gzip
import mmap
import os

import numpy as np

from ._lib import lib


class Reference:
    """contig name -> (offset, length, line_bases, line_width) over a memory-mapped FASTA file"""

    def __init__(self, path, fai=None, write_fai=False):
        self.path = str(path)
        if self.path.endswith(".gz"):
            raise ValueError("a gzip-compressed FASTA cannot be memory-mapped: decompress it (or bgzip + faidx upstream), or use fasta.read_fasta")
        self._f = open(self.path, "rb")
        self.size = os.fstat(self._f.fileno()).st_size
        self._mm = mmap.mmap(self._f.fileno(), 0, access=mmap.ACCESS_READ) if self.size else None
        self._buf = np.frombuffer(self._mm, np.uint8) if self._mm is not None else np.zeros(0, np.uint8)
        self.base_addr = int(self._buf.ctypes.data) if self.size else 0
        fai = fai or self.path + ".fai"
        if os.path.exists(fai) and os.path.getmtime(fai) >= os.path.getmtime(self.path):
            self._read_fai(fai)
        else:
            self._build()
            if write_fai:
                self.write_fai(fai)
        self.index = {n: i for i, n in enumerate(self.names)}

    # ---- index
    def _read_fai(self, fai):
        names, rows = [], []
        with open(fai) as f:
            for line in f:
                p = line.rstrip("\n").split("\t")
                if len(p) < 5:
                    continue
                names.append(p[0])
                rows.append([int(x) for x in p[1:5]])
        t = np.array(rows, np.int64).reshape(-1, 4)
        self.names = names
        self.length, self.offset = t[:, 0].copy(), t[:, 1].copy()
        self.line_bases, self.line_width = t[:, 2].astype(np.int32), t[:, 3].astype(np.int32)
        # a stale or foreign index must not send the emitter past the mapping
        last = self.offset + np.where(self.line_bases > 0, (np.maximum(self.length, 1) - 1) // np.maximum(self.line_bases, 1) * self.line_width
                                      + (np.maximum(self.length, 1) - 1) % np.maximum(self.line_bases, 1), 0)
        # (a contig without sequence - a header at the end of the file, length 0: samtools writes the same entry - has nothing to
        # read: its offset may equal the file size)
        has = self.length > 0
        if len(t) and ((has.any() and int(last[has].max()) >= self.size) or int(self.offset.min()) < 0 or int(self.offset.max()) > self.size):
            raise ValueError("%s does not describe %s (offsets beyond the file)" % (fai, self.path))

    def _build(self):
        cap = 1024
        while True:
            name_off, name_len = np.zeros(cap, np.int64), np.zeros(cap, np.int32)
            length, offset = np.zeros(cap, np.int64), np.zeros(cap, np.int64)
            lb, lw = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
            n = lib().csv_fasta_index(C.c_void_p(self.base_addr), self.size, cap, name_off.ctypes.data, name_len.ctypes.data,
                                      length.ctypes.data, offset.ctypes.data, lb.ctypes.data, lw.ctypes.data)
            if n < 0:
                raise ValueError("%s cannot be indexed (sequence before a header, or lines of unequal length inside a contig)" % self.path)
            if n <= cap:
                break
            cap = int(n)
        self.names = [bytes(self._buf[int(name_off[i]):int(name_off[i]) + int(name_len[i])]).decode() for i in range(n)]
        self.length, self.offset, self.line_bases, self.line_width = length[:n], offset[:n], lb[:n], lw[:n]

    def write_fai(self, fai=None):
        with open(fai or self.path + ".fai", "w") as f:
            for i, nm in enumerate(self.names):
                f.write("%s\t%d\t%d\t%d\t%d\n" % (nm, self.length[i], self.offset[i], self.line_bases[i], self.line_width[i]))

    # ---- access
    def __contains__(self, name):
        return name in self.index

    def contig(self, name):
        """(address of the first base, length, line_bases, line_width) - what csv_vcf_in takes per chromosome"""
        i = self.index[name]
        return self.base_addr + int(self.offset[i]), int(self.length[i]), int(self.line_bases[i]), int(self.line_width[i])

    def fetch(self, name, start=0, end=None):
        """bases [start, end) of a contig as str (pysam.FastaFile.fetch); for tests and small slices"""
        i = self.index[name]
        n, lb, lw, off = int(self.length[i]), int(self.line_bases[i]), int(self.line_width[i]), int(self.offset[i])
        end = n if end is None else min(end, n)
        start = max(0, start)
        if end <= start:
            return ""
        lo = off + (start // lb) * lw + start % lb
        hi = off + ((end - 1) // lb) * lw + (end - 1) % lb + 1
        raw = bytes(self._buf[lo:hi])
        return raw.replace(b"\r", b"").replace(b"\n", b"").decode()

    def close(self):
        self._buf = None
        if self._mm is not None:
            try:
                self._mm.close()
            except BufferError:          # (numpy views still alive: released with them)
                pass
        self._f.close()


def read_fasta(path, only=None):
    """{contig name: sequence} with every contig as one str (plain or gzip): the round-1 reader, for gzip files and tests"""
    opener = gzip.open if str(path).endswith(".gz") else open
    out, name, parts = {}, None, []
    with opener(path, "rt") as f:
        for line in f:
            if line.startswith(">"):
                if name is not None and (only is None or name in only):
                    out[name] = "".join(parts)
                name, parts = line[1:].split()[0], []
            elif name is not None and (only is None or name in only):
                parts.append(line.strip())
    if name is not None and (only is None or name in only):
        out[name] = "".join(parts)
    return out
