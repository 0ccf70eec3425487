"""VCF record emit for the clustering stage's calls.

`emit_records` hands the structure-of-arrays result straight to the native emitter (`csv_vcf_emit`,
cutesv_amd/csrc/vcf_emit.cpp), which restates generate_output (cuteSV_genotype.py:242-467) and main_ctrl's
SVID numbering (cuteSV main script :1208-1237) — no Python row lists in between.  Only the strings a call
needs from Python-side objects are gathered here: the sliced INS sequence (cuteSV_resolveINDEL.py:402), the
read names when --report_readid is on, and the genotype strings of the gl_idx values that occur.
"""
import ctypes as C

import numpy as np

from . import _abi
from ._lib import lib
from .genotype import gl_fields


class VcfIn(C.Structure):
    _fields_ = [
        ("res", C.POINTER(_abi.BatchOut)), ("seg", C.c_void_p), ("n_seg", C.c_int32), ("n_chrom", C.c_int32),
        ("chrom_name", C.POINTER(C.c_char_p)), ("chrom_seq", C.POINTER(C.c_char_p)), ("chrom_len", C.c_void_p),
        ("chrom_rank", C.c_void_p),
        ("ins_alt", C.c_char_p), ("ins_alt_off", C.c_void_p), ("rnames", C.c_char_p), ("rnames_off", C.c_void_p),
        ("strand_name", C.POINTER(C.c_char_p)),
        ("gl_key", C.c_void_p), ("gl_str", C.POINTER(C.c_char_p)), ("n_gl", C.c_int32),
        ("min_size", C.c_int64), ("max_size", C.c_int64),
        ("genotype", C.c_int32), ("report_readid", C.c_int32), ("ignore_sequence", C.c_int32), ("reserved", C.c_int32),
        ("chrom_line_bases", C.c_void_p), ("chrom_line_width", C.c_void_p),
    ]


def _csr(strings):
    blob = "".join(strings).encode()
    off = np.zeros(len(strings) + 1, np.int64)
    if strings:
        off[1:] = np.cumsum([len(s) for s in strings])
    return blob, off


import threading

_TLS = threading.local()          # the recycled output buffer is per thread: csv_vcf_emit runs with the GIL released and supports
                                  # concurrent callers (a second caller gets a worker team of its own), so must its caller's buffer


def emit_records(store, segments, res, reference, min_size=30, max_size=100000, genotype=False, report_readid=False,
                 ignore_sequence=False, svid=None, as_bytes=False, as_view=False):
    """calls of one batch -> (VCF body text, svid counters).

    segments   the csv_segment records the batch was run with (HostBatch.segments)
    res        _abi.HostResult of that batch
    reference  a fasta.Reference (memory-mapped FASTA + .fai: bases are read in C straight from the mapping), or
               {chromosome name: sequence (str or bytes)}; may miss chromosomes without calls
    svid       running counters [INS, DEL, BND, DUP, INV] (main script :1209-1213), advanced in place
    as_bytes   return the text as `bytes` (what a file is written from) instead of decoding it to `str`
    as_view    return a memoryview of this thread's output buffer instead (valid until the thread's next emit_records call):
               `f.write(view)` needs no copy of the text - with real REF sequences a 30x genome's records are ~30 MB
    """
    L = lib()
    L.csv_vcf_emit.restype = C.c_int
    L.csv_vcf_emit.argtypes = [C.POINTER(VcfIn), C.c_char_p, C.c_int64, C.POINTER(C.c_int64), C.c_void_p]
    t = res.trimmed()
    n = res.n_calls
    segs = np.ascontiguousarray(segments, dtype=_abi.SEGMENT_DTYPE)
    call_type = segs["svtype"][t["call_seg"]] if n else np.zeros(0, np.int32)
    nch = len(store.chroms)
    names = (C.c_char_p * nch)(*[c.encode() for c in store.chroms])
    from .fasta import Reference
    line_bases = line_width = None
    if isinstance(reference, Reference):
        ent = [reference.contig(c) if c in reference else (0, 0, 0, 0) for c in store.chroms]
        seqs = (C.c_void_p * nch)(*[e[0] or None for e in ent])
        clen = np.array([e[1] for e in ent], np.int64)
        line_bases = np.array([e[2] for e in ent], np.int32)
        line_width = np.array([e[3] for e in ent], np.int32)
        seq_bytes = reference                              # (kept alive below)
    else:
        seq_bytes = [None if reference is None or c not in reference else
                     (reference[c] if isinstance(reference[c], bytes) else reference[c].encode()) for c in store.chroms]
        seqs = (C.c_char_p * nch)(*seq_bytes)
        clen = np.array([0 if s is None else len(s) for s in seq_bytes], np.int64)
    order = sorted(range(nch), key=lambda i: store.chroms[i])
    rank = np.zeros(nch, np.int32)
    rank[order] = np.arange(nch, dtype=np.int32)
    # inserted sequences of INS calls, sliced to SVLEN
    alt_blob, alt_off = None, None
    if n and not ignore_sequence and (call_type == _abi.INS).any() and t["seq_pick"] is None:
        raise ValueError("emit_records: the result carries no seq_pick (a slim result without that field): INS records need it unless ignore_sequence")
    if n and report_readid and (t["support_sig"] is None or t["support_off"] is None):
        raise ValueError("emit_records: report_readid needs the support lists (the result was made with CSV_OUT_NO_SUPPORT_LIST)")
    if n and not ignore_sequence and (call_type == _abi.INS).any():
        from . import _cols_native as cn                     # (built by the same make as the library; no Python fallback)
        ins = np.flatnonzero(call_type == _abi.INS)
        pick = np.ascontiguousarray(t["seq_pick"][ins], np.int64)
        ln = np.ascontiguousarray(t["bp2"][ins], np.int64)
        table = store.ins_seq
        if table is None:
            if store.names.names is not None:
                store.sequence(0)                             # (raises: real read names but no inserted sequences)
            ln = np.minimum(store.aux[pick].astype(np.int64), ln)      # synthetic stores: 'ACGT' repeated to the aux length
        from .columns import SpanList
        if isinstance(table, SpanList):                       # (a task store built from the reference's pickle: spans of the mapped file)
            alt_blob, took = table.join(pick, ln)
        else:
            took = np.empty(len(ins), np.int64)
            alt_blob = cn.clip_join(table, pick, ln, took)    # b"".join(sequence(pick)[:SVLEN]) in C (GT:297-309)
        alt_off = np.zeros(n + 1, np.int64)
        alt_off[ins + 1] = took
        np.cumsum(alt_off, out=alt_off)
    rn_blob, rn_off = None, None
    if n and report_readid:
        nm = store.names.take(store.read_id[t["support_sig"]])
        so = t["support_off"].tolist()
        rn_blob, rn_off = _csr([",".join(nm[so[c]:so[c + 1]]) for c in range(n)])
    gl = t["gl_idx"]                                      # (a slim result may leave it out: no call is genotyped then)
    keys = np.unique(gl[gl >= 0]).astype(np.int32) if (n and gl is not None) else np.zeros(0, np.int32)
    gl_strs = (C.c_char_p * max(1, len(keys)))(*["\t".join(gl_fields(int(k))).encode() for k in keys])
    strands = (C.c_char_p * len(store.strands))(*[s.encode() for s in store.strands])
    if svid is None:
        svid = np.zeros(5, np.int64)
    vin = VcfIn(res=C.pointer(res.c), seg=segs.ctypes.data, n_seg=len(segs), n_chrom=nch,
                chrom_name=names, chrom_seq=C.cast(seqs, C.POINTER(C.c_char_p)), chrom_len=clen.ctypes.data, chrom_rank=rank.ctypes.data,
                chrom_line_bases=None if line_bases is None else line_bases.ctypes.data,
                chrom_line_width=None if line_width is None else line_width.ctypes.data,
                ins_alt=alt_blob, ins_alt_off=None if alt_off is None else alt_off.ctypes.data,
                rnames=rn_blob, rnames_off=None if rn_off is None else rn_off.ctypes.data,
                strand_name=strands, gl_key=keys.ctypes.data, gl_str=gl_strs, n_gl=len(keys),
                min_size=min_size, max_size=max_size, genotype=int(bool(genotype)), report_readid=int(bool(report_readid)),
                ignore_sequence=int(bool(ignore_sequence)))
    cap = 256 * max(n, 1) + (len(alt_blob) if alt_blob else 0) + (len(rn_blob) if rn_blob else 0) + 4096
    for _ in range(2):
        buf = getattr(_TLS, "buf", None)
        if buf is None or len(buf) < cap:                 # (recycled: a fresh zero-filled buffer per call costs as much as the emitter)
            buf = _TLS.buf = np.empty(cap, np.uint8)
        need = C.c_int64(0)
        sv = svid.copy()
        rc = L.csv_vcf_emit(C.byref(vin), C.cast(buf.ctypes.data, C.c_char_p), len(buf), C.byref(need), sv.ctypes.data)
        if rc == _abi.E_CAPACITY:
            cap = need.value + 16
            continue
        if rc != _abi.OK:
            raise RuntimeError("csv_vcf_emit: %s (a reference sequence is missing or too short?)" % _abi.ERR_NAME.get(rc, rc))
        svid[:] = sv
        if as_view:
            return memoryview(buf)[:need.value], svid
        raw = buf[:need.value].tobytes()
        return (raw if as_bytes else raw.decode()), svid
    raise RuntimeError("csv_vcf_emit: capacity retry failed")


def emit_stage(results, reference, **kw):
    """VCF body text of a `resolve.cluster_stage(..., lazy=True)` result ({chr: rows.LazyRows}): the native emitter reads the
    structure of arrays the lazy rows are backed by - no row strings are created on the way (GT:242-467, main script
    :1208-1237).  Keyword arguments as emit_records."""
    backs = {id(b): b for b in (v.backing() if hasattr(v, "backing") else None for v in results.values()) if b is not None}
    if len(backs) != 1 or any(not hasattr(v, "backing") or v.backing() is None for v in results.values()):
        raise ValueError("emit_stage needs the untouched result of ONE cluster_stage(lazy=True) call")
    b = next(iter(backs.values()))
    # "untouched" is checked, not assumed (advisor, r05): the emitter writes EVERY call of the backing's result, so the rows
    # on offer must be exactly those calls - each once, none dropped with a chromosome, a task or a filtered row
    idx = np.concatenate([v.call_indices() for v in results.values()]) if results else np.zeros(0, np.int64)
    n = b.res.n_calls
    if len(idx) != n or (n and not np.array_equal(np.sort(idx), np.arange(n))):
        raise ValueError("emit_stage: the rows on offer are not exactly the calls of the stage's result (%d rows, %d calls): rows were "
                         "removed, repeated or filtered - write those with emit_records on a result of their own, or from the row lists" % (len(idx), n))
    return emit_records(b.store, b.segments, b.res, reference, **kw)
