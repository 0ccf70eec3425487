"""The CIGAR scan of cuteSV's extraction step on the GPU.

`cigar_signatures` is the face of `csv_cigar_signatures` (cutesv_amd/csrc/cigar.hip.h): the flat BAM-encoded CIGAR array
of a batch of reads -> the INS / DEL signatures parse_read + generate_combine_sigs (main script :606-655, :515-575) make
of them.  `candidates` turns the flat result into the reference's candidate tuples (the inserted sequence is cut out of
the reads' query sequences here: the bases never travel to the GPU).  BAM decode, the SA-tag split-read analysis and
everything else of the extraction stay in the Python driver with pysam, as north_star has it: a driver would collect
`read.cigartuples`, `read.reference_start`, `read.mapq >= min_mapq and read.query_length >= min_read_len` for a task's
reads, make one call here, and extend candidate["INS"] / candidate["DEL"] with the result.
"""
import ctypes as C

import numpy as np

from . import _abi
from ._lib import lib


class CigarIn(C.Structure):
    _fields_ = [("n_reads", C.c_int64), ("cig_off", C.c_void_p), ("cigar", C.c_void_p), ("ref_start", C.c_void_p), ("use", C.c_void_p),
                ("min_siglength", C.c_int32), ("reserved", C.c_int32), ("merge_ins_threshold", C.c_int64), ("merge_del_threshold", C.c_int64)]


_OUT = [("ins_read", np.int32, "i"), ("ins_pos", np.int64, "i"), ("ins_len", np.int64, "i"), ("ins_piece0", np.int64, "i"), ("ins_npiece", np.int32, "i"),
        ("piece_qoff", np.int32, "p"), ("piece_len", np.int32, "p"), ("del_read", np.int32, "d"), ("del_pos", np.int64, "d"), ("del_len", np.int64, "d")]


class CigarOut(C.Structure):
    _fields_ = ([("cap_sig_ins", C.c_int64), ("cap_piece_ins", C.c_int64), ("cap_sig_del", C.c_int64),
                 ("n_sig_ins", C.c_int64), ("n_piece_ins", C.c_int64), ("n_sig_del", C.c_int64)]
                + [(n, C.c_void_p) for n, _, _ in _OUT] + [("ms_device", C.c_float), ("reserved", C.c_int32)])


def encode_cigars(cigartuples_per_read):
    """[[(op, oplen), ...] per read] (pysam's read.cigartuples) -> (cig_off int64[n + 1], cigar uint32[n_ops]) in the BAM
    encoding oplen << 4 | op"""
    lens = np.fromiter((len(c) for c in cigartuples_per_read), np.int64, len(cigartuples_per_read))
    off = np.zeros(len(lens) + 1, np.int64)
    np.cumsum(lens, out=off[1:])
    flat = np.empty(int(off[-1]), np.uint32)
    k = 0
    for c in cigartuples_per_read:
        for op, ln in c:
            flat[k] = (int(ln) << 4) | int(op)
            k += 1
    return off, flat


def _run(fn, handle, cig_off, cigar, ref_start, use, min_siglength, merge_ins_threshold, merge_del_threshold, check):
    cig_off = np.ascontiguousarray(cig_off, np.int64); cigar = np.ascontiguousarray(cigar, np.uint32)
    ref_start = np.ascontiguousarray(ref_start, np.int64)
    use = None if use is None else np.ascontiguousarray(use, np.uint8)
    n = len(ref_start)
    cin = CigarIn(n_reads=n, cig_off=cig_off.ctypes.data, cigar=cigar.ctypes.data if len(cigar) else None, ref_start=ref_start.ctypes.data,
                  use=None if use is None else use.ctypes.data, min_siglength=int(min_siglength),
                  merge_ins_threshold=int(merge_ins_threshold), merge_del_threshold=int(merge_del_threshold))
    caps = dict(i=max(16, n // 4), p=max(16, n // 4), d=max(16, n // 4))
    for _ in range(2):
        arrs = {name: np.zeros(caps[k], dt) for name, dt, k in _OUT}
        cout = CigarOut(cap_sig_ins=caps["i"], cap_piece_ins=caps["p"], cap_sig_del=caps["d"], **{k: v.ctypes.data for k, v in arrs.items()})
        rc = fn(handle, C.byref(cin), C.byref(cout)) if handle is not None else fn(C.byref(cin), C.byref(cout))
        if rc == _abi.E_CAPACITY:
            caps = dict(i=int(cout.n_sig_ins) + 1, p=int(cout.n_piece_ins) + 1, d=int(cout.n_sig_del) + 1)
            continue
        check(rc)
        cut = dict(i=int(cout.n_sig_ins), p=int(cout.n_piece_ins), d=int(cout.n_sig_del))
        out = {name: arrs[name][:cut[k]] for name, _, k in _OUT}
        out["ms_device"] = float(cout.ms_device)
        return out
    raise RuntimeError("csv_cigar_signatures: capacity retry failed")


def cigar_signatures(ctx, cig_off, cigar, ref_start, use=None, min_siglength=10, merge_ins_threshold=100, merge_del_threshold=0):
    """flat CIGARs of a batch of reads -> dict of the signature arrays of csv_cigar_out (defaults: cuteSV_Description.py:123-152)"""
    L = lib()
    L.csv_cigar_signatures.restype = C.c_int
    L.csv_cigar_signatures.argtypes = [C.c_void_p, C.POINTER(CigarIn), C.POINTER(CigarOut)]
    return _run(L.csv_cigar_signatures, ctx._h, cig_off, cigar, ref_start, use, min_siglength, merge_ins_threshold, merge_del_threshold, ctx._check)


def candidates(sig, read_names, query_sequences, chrom):
    """signature arrays -> the reference's candidate tuples (main script :520-575):
    INS (pos, len, read, seq, "INS", chr), DEL (pos, len, read, "DEL", chr), in read order"""
    ins, dele = [], []
    qo, ql = sig["piece_qoff"].tolist(), sig["piece_len"].tolist()
    for r, pos, ln, p0, npc in zip(sig["ins_read"].tolist(), sig["ins_pos"].tolist(), sig["ins_len"].tolist(),
                                    sig["ins_piece0"].tolist(), sig["ins_npiece"].tolist()):
        q = query_sequences[r]
        seq = "".join(str(q[qo[p] : qo[p] + ql[p]]) for p in range(p0, p0 + npc))      # read.query_sequence[shift - oplen : shift] (:639-640)
        ins.append((pos, ln, read_names[r], seq, "INS", chrom))
    for r, pos, ln in zip(sig["del_read"].tolist(), sig["del_pos"].tolist(), sig["del_len"].tolist()):
        dele.append((pos, ln, read_names[r], "DEL", chrom))
    return ins, dele
