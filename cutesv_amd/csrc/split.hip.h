// split.hip.h — SURVEY.md §8f row 4, second half: the split-read (SA tag) analysis of cuteSV's extraction step on the GPU.
//
// Restates organize_split_signal (cuteSV main script :483-513) and analysis_split_read with analysis_inv / analysis_bnd
// (:50-464).  A read contributes its primary alignment and the entries of its SA tag; each becomes a segment
// [read_start, read_end, ref_start, ref_end, chr, strand]; the segments are sorted by read_start (stable) and walked with
// the reference's two-segment and sliding three-segment rules, which emit INV / TRA / DUP / INS / DEL candidates.  The
// text of the SA tag (and acquire_clip_pos on its CIGAR strings, :466-481) stays with pysam in the Python driver
// (north_star): the input is flat per-entry numbers.
//
// One THREAD per read: a read has a handful of segments (--max_split_parts 7 by default) and the rules are scalar
// branches on a few integers, so the parallelism is across the reads of a batch (a task region holds 10^4 - 10^5).  Two
// passes (count, prefix over reads - the scan kernels of cigar.hip.h -, emit) because the output is dense and in read
// order; segments are sorted in the read's own slice of a global scratch array.  Every comparison the reference makes in
// floating point (x + 0.5 * d >= y;  x < max(SV_size, delta / 5)) is made on the same values: the first in exact integer
// form, the second in float64 like Python.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace csv {

struct SpSeg { i64 rs, re, fs, fe; int chr, st; };           // st: 0 '+', 1 '-'

struct SplitArgs {
    i64 n_reads;
    const i64* ent_off; const i64* read_len;
    const i64* c0; const i64* c1; const i64* f0; const i64* f1;
    const int* chr; const int* mapq; const uint8_t* strand; const uint8_t* primary;
    i64 sv, max_size;
    int min_mapq, parts;
    SpSeg* seg;                     // scratch: one slot per entry
    int4* cnt;                      // per read: x = candidates (pass 1) / exclusive offset (after the scan)
    i64 cap;
    uint8_t* kind; int* read; int* o_chr; int* aux; i64* a; i64* b; i64* c; i64* d;
};

template <bool EMIT> struct SpSink {
    const SplitArgs& A; i64 base; int n; int read;
    __device__ __forceinline__ void put(int kind, int chr, int aux, i64 a, i64 b, i64 c, i64 d)
    {
        if (EMIT) {
            const i64 k = base + n;
            if (k < A.cap) { A.kind[k] = (uint8_t)kind; A.read[k] = read; A.o_chr[k] = chr; A.aux[k] = aux; A.a[k] = a; A.b[k] = b; A.c[k] = c; A.d[k] = d; }
        }
        n++;
    }
};

__device__ __forceinline__ SpSeg sp_flip(SpSeg x, i64 L) { SpSeg y = x; y.rs = L - x.re; y.re = L - x.rs; return y; }    // [RLength - x[1], RLength - x[0]] + x[2:]
__device__ __forceinline__ double sp_max(i64 a, i64 delta) { const double q = (double)delta / 5.0; return (double)a > q ? (double)a : q; }   // max(a, delta / 5)

template <bool EMIT> __device__ __forceinline__ void sp_inv(SpSink<EMIT>& S, const SpSeg& e1, const SpSeg& e2, i64 SV)      // analysis_inv :50-95
{
    if (e1.st == 0) {
        if (e1.fe - e2.fe >= SV && 2 * e2.rs + (e1.fe - e2.fe) >= 2 * e1.re) S.put(3, e1.chr, 0, e2.fe, e1.fe, 0, 0);
        if (e2.fe - e1.fe >= SV && 2 * e2.rs + (e2.fe - e1.fe) >= 2 * e1.re) S.put(3, e1.chr, 0, e1.fe, e2.fe, 0, 0);
    } else {
        if (e2.fs - e1.fs >= SV && 2 * e2.rs + (e2.fs - e1.fs) >= 2 * e1.re) S.put(3, e1.chr, 1, e1.fs, e2.fs, 0, 0);
        if (e1.fs - e2.fs >= SV && 2 * e2.rs + (e1.fs - e2.fs) >= 2 * e1.re) S.put(3, e1.chr, 1, e2.fs, e1.fs, 0, 0);
    }
}
template <bool EMIT> __device__ __forceinline__ void sp_bnd(SpSink<EMIT>& S, const SpSeg& e1, const SpSeg& e2)                // analysis_bnd :97-188
{
    if (e2.rs - e1.re > 100) return;
    const bool lt = e1.chr < e2.chr;                   // (chromosome ids are ranks in Python string order)
    // type codes 0..3 = 'A'..'D'
    if (e1.st == 0 && e2.st == 0) { if (lt) S.put(4, e1.chr, 0, e1.fe, e2.fs, e2.chr, 0); else S.put(4, e2.chr, 3, e2.fs, e1.fe, e1.chr, 0); }
    else if (e1.st == 0)          { if (lt) S.put(4, e1.chr, 1, e1.fe, e2.fe, e2.chr, 0); else S.put(4, e2.chr, 1, e2.fe, e1.fe, e1.chr, 0); }
    else if (e2.st == 0)          { if (lt) S.put(4, e1.chr, 2, e1.fs, e2.fs, e2.chr, 0); else S.put(4, e2.chr, 2, e2.fs, e1.fs, e1.chr, 0); }
    else                          { if (lt) S.put(4, e1.chr, 3, e1.fs, e2.fe, e2.chr, 0); else S.put(4, e2.chr, 0, e2.fe, e1.fs, e1.chr, 0); }
}
// the INS / DEL pair of rules on two consecutive same-strand segments (:241-259, :358-376, :382-399, :411-428); `gate` is
// the extra ele_3[2] >= ele_2[3] test of :361 / :371 (true where the reference has none)
template <bool EMIT> __device__ __forceinline__ void sp_indel(SpSink<EMIT>& S, const SpSeg& e1, const SpSeg& e2, i64 SV, i64 Max, int rc, bool gate)
{
    i64 delta = e2.rs + e1.fe - e2.fs - e1.re;
    if ((double)(e1.fe - e2.fs) < sp_max(SV, delta) && delta >= SV)
        if ((double)(e2.fs - e1.fe) <= sp_max(100, delta) && (delta <= Max || Max == -1))
            if (gate) S.put(1, e2.chr, 2 | rc, e2.fs + e1.fe, delta, e1.re + (e2.fs - e1.fe) / 2, e2.rs - (e2.fs - e1.fe) / 2);
    delta = e2.fs - e2.rs + e1.re - e1.fe;
    if ((double)(e1.fe - e2.fs) < sp_max(SV, delta) && delta >= SV)
        if ((double)(e2.rs - e1.re) <= sp_max(100, delta) && (delta <= Max || Max == -1))
            if (gate) S.put(0, e2.chr, 0, e1.fe, delta, 0, 0);
}

template <bool EMIT> __device__ __forceinline__ int split_read(const SplitArgs& A, i64 r, i64 out_base)
{
    const i64 e0 = A.ent_off[r], e1o = A.ent_off[r + 1], L = A.read_len[r];
    const i64 SV = A.sv, Max = A.max_size;
    SpSeg* SP = A.seg + e0;
    // organize_split_signal: the primary alignment (it lifts the mapq gate, :486-488), then the SA entries that pass it;
    // inserted in read_start order as they come (== the stable sort of :195)
    int n = 0, min_mapq = A.min_mapq;
    for (i64 k = e0; k < e1o; k++) {
        SpSeg x; x.chr = A.chr[k]; x.st = A.strand[k];
        if (A.primary[k]) { x.rs = A.c0[k]; x.re = A.c1[k]; x.fs = A.f0[k]; x.fe = A.f1[k]; min_mapq = 0; }
        else {
            if (A.mapq[k] < min_mapq) continue;                                                      // :501
            // (both words loaded, then chosen: written as two branches that read c0 / c1 crosswise, the compiler builds a table
            // of the two POINTERS in scratch and indexes it by the strand - 24 bytes of scratch per lane for a select)
            const i64 v0 = A.c0[k], v1 = A.c1[k];
            x.rs = x.st == 0 ? v0 : v1; x.re = L - (x.st == 0 ? v1 : v0);                             // :503-510
            x.fs = A.f0[k]; x.fe = A.f0[k] + A.f1[k];
        }
        int p = n++;
        while (p > 0 && SP[p - 1].rs > x.rs) { SP[p] = SP[p - 1]; p--; }
        SP[p] = x;
    }
    SpSink<EMIT> S{A, out_base, 0, (int)r};
    if (!(n <= A.parts || A.parts == -1)) return 0;                                                  // :512
    bool trigger = false;
    if (n == 2) {
        SpSeg a1 = SP[0], a2 = SP[1];
        if (a1.chr == a2.chr) {
            if (a1.st != a2.st) sp_inv(S, a1, a2, SV);
            else {
                int rc = 0;
                if (a1.st == 1) { a1 = sp_flip(SP[1], L); a2 = sp_flip(SP[0], L); rc = 1; }           // :219-223
                if (a1.fe - a2.fs >= SV) {                                                           // :225-241
                    if (a2.rs - a1.re >= a1.fe - a2.fs)
                        S.put(1, a2.chr, 2 | rc, a1.fe + a2.fs, a2.rs + a1.fe - a2.fs - a1.re, a1.re + (a2.fs - a1.fe) / 2, a2.rs - (a2.fs - a1.fe) / 2);
                    else S.put(2, a2.chr, 0, a2.fs, a1.fe, 0, 0);
                }
                sp_indel(S, a1, a2, SV, Max, rc, true);
            }
        } else sp_bnd(S, a1, a2);
    } else {
        for (int a = 0; a + 2 < n; a++) {
            SpSeg a1 = SP[a], a2 = SP[a + 1], a3 = SP[a + 2];
            bool a3_none = false;
            const bool last = (n - 3 == a);
            if (a1.chr == a2.chr) {
                if (a2.chr == a3.chr) {
                    if (a1.st == a3.st && a1.st != a2.st) {                                          // :270-314: + - +  /  - + -
                        if (a2.st == 1) {
                            const i64 d = a3.fs - a1.fe;
                            if (2 * a2.rs + d >= 2 * a1.re && 2 * a3.rs + d >= 2 * a2.re)
                                if (a2.fs >= a1.fe && a3.fs >= a2.fe) { S.put(3, a1.chr, 0, a1.fe, a2.fe, 0, 0); S.put(3, a1.chr, 1, a2.fs, a3.fs, 0, 0); }
                        } else {
                            const i64 d = a1.fs - a3.fe;
                            if (2 * a1.re <= 2 * a2.rs + d && 2 * a3.rs + d >= 2 * a2.re)
                                if (a2.fs - a3.fe >= -50 && a1.fs - a2.fe >= -50) { S.put(3, a1.chr, 0, a3.fe, a2.fe, 0, 0); S.put(3, a1.chr, 1, a2.fs, a1.fs, 0, 0); }
                        }
                    }
                    if (last && a1.st != a3.st) {                                                    // :316-331
                        if (a2.st == a1.st) sp_inv(S, a2, a3, SV); else sp_inv(S, a1, a2, SV);
                    }
                    if (a1.st == a3.st && a1.st == a2.st) {                                          // :333-399: dup & ins & del
                        int rc = 0;
                        if (a1.st == 1) { a1 = sp_flip(SP[a + 2], L); a2 = sp_flip(SP[a + 1], L); a3 = sp_flip(SP[a], L); rc = 1; }
                        if (a2.fe - a3.fs >= SV && a2.fs < a3.fe) S.put(2, a2.chr, 0, a3.fs, a2.fe, 0, 0);
                        if (a == 0 && a1.fe - a2.fs >= SV) S.put(2, a2.chr, 0, a2.fs, a1.fe, 0, 0);
                        sp_indel(S, a1, a2, SV, Max, rc, a3.fs >= a2.fe);
                        if (last) { a1 = a2; a2 = a3; sp_indel(S, a1, a2, SV, Max, rc, true); }
                    }
                    if (last && a1.st != a2.st && a2.st == a3.st) { a1 = a2; a2 = a3; a3_none = true; }                 // :401-404
                    if (a3_none || (a1.st == a2.st && a2.st != a3.st)) {                             // :405-428 (the flip takes SP[a + 1], SP[a] as written)
                        int rc = 0;
                        if (a1.st == 1) { a1 = sp_flip(SP[a + 1], L); a2 = sp_flip(SP[a], L); rc = 1; }
                        sp_indel(S, a1, a2, SV, Max, rc, true);
                    }
                }
            } else {                                                                                 // :431-437
                trigger = true;
                sp_bnd(S, a1, a2);
                if (last && a2.chr != a3.chr) sp_bnd(S, a2, a3);
            }
        }
    }
    if (n >= 3 && trigger && SP[0].chr == SP[n - 1].chr && SP[0].st == SP[n - 1].st) {               // :439-464: an insertion inside a translocation
        SpSeg a1, a2; int rc = 0;
        if (SP[0].st == 0) { a1 = SP[0]; a2 = SP[n - 1]; } else { a1 = sp_flip(SP[n - 1], L); a2 = sp_flip(SP[0], L); rc = 1; }
        const i64 dis_ref = a2.fs - a1.fe, dis_read = a2.rs - a1.re, dl = dis_read - dis_ref;
        const i64 ad = dis_ref < 0 ? -dis_ref : dis_ref;
        if ((double)ad < sp_max(SV, dl) && dl >= SV && (dl <= Max || Max == -1))
            S.put(1, a2.chr, rc, a2.fs < a1.fe ? a2.fs : a1.fe, dl, a1.re + dis_ref / 2, a2.rs - dis_ref / 2);
        if (dis_ref <= -SV) S.put(2, a2.chr, 0, a2.fs, a1.fe, 0, 0);
    }
    return S.n;
}

__global__ __launch_bounds__(256) void k_split_count(SplitArgs A)
{
    const i64 r = (i64)blockIdx.x * 256 + threadIdx.x;
    if (r < A.n_reads) A.cnt[r] = make_int4(split_read<false>(A, r, 0), 0, 0, 0);
}
__global__ __launch_bounds__(256) void k_split_emit(SplitArgs A)
{
    const i64 r = (i64)blockIdx.x * 256 + threadIdx.x;
    if (r < A.n_reads) split_read<true>(A, r, A.cnt[r].x);
}

// CSV_CG_TO_POOL: the candidates as rows of the device-resident signature pool (cigar.hip.h PoolCols), in the columns the rebuild
// sorts on (cutesv_amd/columns.py from_tuple_lists has the same encoding: INV aux = strand code, TRA aux = chr2 * 8 + type)
struct PoolSegBase { int b[5]; };
__global__ __launch_bounds__(256) void k_pool_from_split(PoolCols P, i64 base, SplitArgs A, i64 n, PoolSegBase sb, i64 read_base, const int* query_len)
{
    const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int kind = A.kind[i], r = A.read[i], aux = A.aux[i];
    i64 a = A.a[i], b = A.b[i];
    int ax = 0;
    if (kind == 1) {
        if (aux & 2) a >>= 1;                               // an x.5 position travels doubled: its integer part sorts
        const i64 ql = query_len ? (i64)query_len[r] : A.read_len[r], c = A.c[i], d = A.d[i];
        const i64 lo = c < ql ? c : ql, hi = d < ql ? d : ql;
        ax = (int)(hi > lo ? hi - lo : 0);
    } else if (kind == 3) ax = aux;
    else if (kind == 4) ax = (int)(A.c[i] * 8 + aux);
    const i64 o = base + i;
    P.seg[o] = sb.b[kind] + A.o_chr[i]; P.a[o] = a; P.b[o] = b; P.read[o] = (int)(read_base + r); P.aux[o] = ax;
}

}  // namespace csv
