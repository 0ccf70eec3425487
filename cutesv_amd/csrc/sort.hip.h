// sort.hip.h — the rebuild step on the GPU: stable LSD radix sort of row
// permutations on multi-column keys, then gather + adjacent de-duplication.
//
// Reproduces the ORDER CONTRACT of cuteSV's process_process_sigs_type (main script :750-857): signatures
// sorted by (chromosome, [strand | chr2, type], int(pos), len | pos2, read name) (:764-802) with adjacent
// exact duplicates removed (:958-969).  Read names are interned ids whose order is the names' string order,
// so the whole key is integer columns.
//
// One pass = one 8-bit digit of one key column: k_sort_hist (per-wavefront-tile digit histograms),
// k_sort_rowsum / k_sort_rowscan (digit-major exclusive scan), k_sort_scatter (stable scatter of the permutation).  The
// permutation moves, the columns stay in place and are gathered by index; digits whose byte is zero for
// every row are skipped by the host.  Intra-tile stability uses the bit-sliced ballot trick: eight
// __ballot()s tell each lane which lanes of its 64-row chunk carry the same digit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace csv {

// 64-row chunks per wavefront tile.  Measured on the 2.85 M rows of a 30x genome (composite-key sort, one box, ms per rebuild):
// 8: 1.09, 16: 0.71, 32: 0.58, 64: 0.53, 128: 0.66 - a longer tile keeps more rows of a digit together, so the scatter writes
// longer runs (it is bound by its scattered 16-byte stores, not by the ~700 wavefronts' latency); beyond 64 the chip runs dry.
#ifndef CSV_SORT_CHUNKS
#define CSV_SORT_CHUNKS 64
#endif
constexpr int SORT_CHUNKS = CSV_SORT_CHUNKS;
constexpr int SORT_WTILE = 64 * SORT_CHUNKS;         // rows per wavefront tile (4096)

struct SortPass {
    const void* col;       // key column (device)
    int         elem64;    // 1: int64 column, 0: int32
    int         shift;     // bit offset of the digit
    i64         n;
    int         nunits;    // wavefront tiles
    const int*  perm_in;   // nullptr on the first pass (identity)
    int*        perm_out;
    int*        hist;      // [256][nunits]
};

__device__ __forceinline__ int sort_digit(const SortPass& P, i64 row)
{
    const i64 src = P.perm_in ? P.perm_in[row] : row;
    const u64 v = P.elem64 ? (u64)((const i64*)P.col)[src] : (u64)(unsigned)((const int*)P.col)[src];
    return (int)((v >> P.shift) & 255);
}

// one wavefront per tile of 2048 rows: LDS histogram per wavefront, written digit-major
__global__ __launch_bounds__(256) void k_sort_hist(SortPass P)
{
    __shared__ int h[4][256];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int unit = blockIdx.x * 4 + wv;
    for (int d = lane; d < 256; d += 64) h[wv][d] = 0;
    __syncthreads();
    if (unit < P.nunits) {
        const i64 base = (i64)unit * SORT_WTILE;
        for (int c = 0; c < SORT_CHUNKS; c++) {
            const i64 row = base + c * 64 + lane;
            if (row < P.n) atomicAdd(&h[wv][sort_digit(P, row)], 1);
        }
    }
    __syncthreads();
    if (unit < P.nunits)
        for (int d = lane; d < 256; d += 64) P.hist[(i64)d * P.nunits + unit] = h[wv][d];
}

// exclusive scan of hist in (digit, unit) order, two launches of 256 workgroups (one per digit):
//   k_sort_rowsum   total of digit d's row -> tot[d]
//   k_sort_rowscan  base(d) = sum of tot[d' < d]; row-local exclusive scan + base, in place
__global__ __launch_bounds__(256) void k_sort_rowsum(const int* hist, int nunits, int* tot)
{
    const int* row = hist + (i64)blockIdx.x * nunits;
    i64 v = 0;
    for (int u = threadIdx.x; u < nunits; u += 256) v += row[u];
    v = wave_sum_i64(v);
    __shared__ i64 s[4];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) tot[blockIdx.x] = (int)(s[0] + s[1] + s[2] + s[3]);
}

__global__ __launch_bounds__(256) void k_sort_rowscan(int* hist, int nunits, const int* tot)
{
    __shared__ i64 sh[4];
    __shared__ int wsum[4];
    __shared__ int carry_s;
    int* row = hist + (i64)blockIdx.x * nunits;
    const int base = (int)block_prefix_of(tot, blockIdx.x, sh);
    if (threadIdx.x == 0) carry_s = base;
    __syncthreads();
    for (int b0 = 0; b0 < nunits; b0 += 256) {
        const int u = b0 + threadIdx.x;
        const int v = u < nunits ? row[u] : 0;
        const int inc = wave_incl_scan_i32(v);
        if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = inc;
        __syncthreads();
        int off = carry_s;
        for (int k = 0; k < (int)(threadIdx.x >> 6); k++) off += wsum[k];
        if (u < nunits) row[u] = off + inc - v;
        __syncthreads();
        if (threadIdx.x == 255) carry_s = off + inc;
        __syncthreads();
    }
}

// stable scatter: every wavefront walks its tile chunk by chunk, in row order
__global__ __launch_bounds__(256) void k_sort_scatter(SortPass P)
{
    __shared__ int cnt[4][256];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int unit = blockIdx.x * 4 + wv;
    if (unit < P.nunits)
        for (int d = lane; d < 256; d += 64) cnt[wv][d] = P.hist[(i64)d * P.nunits + unit];
    __syncthreads();
    if (unit >= P.nunits) return;
    const i64 base = (i64)unit * SORT_WTILE;
    const u64 lt = (1ull << lane) - 1ull;
    for (int c = 0; c < SORT_CHUNKS; c++) {
        const i64 row = base + c * 64 + lane;
        const bool in = row < P.n;
        const int dig = in ? sort_digit(P, row) : 0;
        const int src = in ? (P.perm_in ? P.perm_in[row] : (int)row) : 0;
        u64 match = __ballot(in);
#pragma unroll
        for (int bit = 0; bit < 8; bit++) {
            const bool set = (dig >> bit) & 1;
            const u64 mk = __ballot(set);
            match &= set ? mk : ~mk;
        }
        if (in) {
            const int before = cnt[wv][dig];                         // same value for the whole match group
            const int rank = __popcll(match & lt);
            P.perm_out[before + rank] = src;
        }
        // the highest lane of each group advances the group's counter (after everyone read it: LDS ops of a
        // wavefront execute in order)
        if (in && (match >> lane) <= 1ull) cnt[wv][dig] += __popcll(match);
        __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------------------------------- composite-key sort (round 4)
// The sort above moves a permutation and GATHERS the digit of every row through it in both kernels of every pass: 64 lanes,
// 64 cache lines, 8 useful bytes each - 173 us per pass on the 2.85 M rows of a 30x genome, 10 passes, and the final gather
// + de-duplication chase the permutation through five columns again (r03: 1.73 ms in all, 6 % of the HBM roofline).
// Here the whole order contract is ONE integer per row.  Field widths come from the columns' maxima (host pass or
// k_pool_to_rows): key = seg | aux of aux-major segments | a | b | read id, most significant first, T bits; the row index
// (ib bits) and the row's aux word ride along.  When idx + key + aux fit 128 bits (a 30x genome: 22 + 70 + 13) an element
// is 16 bytes {lo, hi} = aux << (ib + T) | key << ib | idx; otherwise 32 bytes {key lo, key hi, idx | aux << 32, 0}
// (keys up to 128 bits; beyond that the permutation sort above takes the batch).  A pass is an LSD step on a 10-bit digit of
// the key bits - 7 passes for 70 bits - that MOVES the elements: every load is coalesced, the hist kernel reads 16 bytes per
// row, the scatter kernel reads 16 and writes 16.  The tail needs no gather either: duplicates are equal keys with equal aux
// words, ties of the keep-every-row segments are equal keys, and the output columns are fields of the key.
constexpr int RS_BITS = 10, RS_RADIX = 1 << RS_BITS;
typedef unsigned __int128 u128;
struct KeyLayout {
    int ib, rb, bb, ab, xb, sb, pb;      // bits of: row index, read id, b, a, aux-as-key, segment, aux payload (compact form)
    int T;                               // key bits = rb + bb + ab + xb + sb
    int wide;                            // 1: 32-byte elements
};
template <bool WIDE> struct RsElem;
template <> struct RsElem<false> { ulonglong2 v; };
template <> struct RsElem<true>  { ulonglong4 v; };
__device__ __forceinline__ u128 rs_u128(u64 lo, u64 hi) { return ((u128)hi << 64) | lo; }
__device__ __forceinline__ u64 rs_mask(int bits) { return bits >= 64 ? ~0ull : ((1ull << bits) - 1ull); }
// the key (T bits, right-aligned), the row index and the aux word of an element
template <bool WIDE> __device__ __forceinline__ u128 rs_key(const RsElem<WIDE>& e, const KeyLayout& L)
{
    if constexpr (WIDE) return rs_u128(e.v.x, e.v.y);
    else { const u128 w = rs_u128(e.v.x, e.v.y) >> L.ib; return L.T >= 128 ? w : (w & (((u128)1 << L.T) - 1)); }
}
template <bool WIDE> __device__ __forceinline__ int rs_idx(const RsElem<WIDE>& e, const KeyLayout& L)
{
    if constexpr (WIDE) return (int)(e.v.z & 0xffffffffull); else return (int)(e.v.x & rs_mask(L.ib));
}
template <bool WIDE> __device__ __forceinline__ int rs_aux(const RsElem<WIDE>& e, const KeyLayout& L)
{
    if constexpr (WIDE) return (int)(e.v.z >> 32); else return (int)(u64)(rs_u128(e.v.x, e.v.y) >> (L.ib + L.T));
}
template <bool WIDE> __device__ __forceinline__ void rs_set_row(RsElem<WIDE>& e, const KeyLayout& L, int idx, int aux)
{
    if constexpr (WIDE) e.v.z = (u64)(unsigned)idx | ((u64)(unsigned)aux << 32);
    else {
        u128 w = rs_u128(e.v.x, e.v.y);
        const u128 keep = (((u128)1 << L.T) - 1) << L.ib;
        w = (w & keep) | (u128)(unsigned)idx | ((u128)(unsigned)aux << (L.ib + L.T));
        e.v.x = (u64)w; e.v.y = (u64)(w >> 64);
    }
}
// digit of `dbits` key bits at `shift` (inside the key; the last digit of a key is narrower: the bits above it are not key)
template <bool WIDE> __device__ __forceinline__ int rs_digit(const RsElem<WIDE>& e, const KeyLayout& L, int shift, int dbits)
{
    const int s = shift + (WIDE ? 0 : L.ib);
    const u64 lo = e.v.x, hi = e.v.y;
    const u64 w = s >= 64 ? (hi >> (s - 64)) : (s == 0 ? lo : ((lo >> s) | (hi << (64 - s))));
    return (int)(w & ((1u << dbits) - 1u));
}
struct RsCols { const int* seg; const i64* a; const i64* b; const int* rid; const int* aux; const uint8_t* major; };

template <bool WIDE> __global__ __launch_bounds__(256) void k_rs_pack(RsCols C, i64 n, KeyLayout L, RsElem<WIDE>* out)
{
    const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int sg = C.seg[i], ax = C.aux[i];
    u128 k = (u128)(unsigned)sg;
    k = (k << L.xb) | (u128)(unsigned)(C.major[sg] ? ax : 0);
    k = (k << L.ab) | (u128)(u64)C.a[i];
    k = (k << L.bb) | (u128)(u64)C.b[i];
    k = (k << L.rb) | (u128)(unsigned)C.rid[i];
    RsElem<WIDE> e;
    if constexpr (WIDE) { e.v.x = (u64)k; e.v.y = (u64)(k >> 64); e.v.z = (u64)(unsigned)i | ((u64)(unsigned)ax << 32); e.v.w = 0; }
    else { const u128 w = ((u128)(unsigned)ax << (L.ib + L.T)) | (k << L.ib) | (u128)(u64)i; e.v.x = (u64)w; e.v.y = (u64)(w >> 64); }
    out[i] = e;
}
// one wavefront per tile of 2048 elements: LDS histogram per wavefront, written digit-major
template <bool WIDE> __global__ __launch_bounds__(256) void k_rs_hist(const RsElem<WIDE>* in, i64 n, int nunits, KeyLayout L, int shift, int dbits, int* hist)
{
    __shared__ int h[4][RS_RADIX];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int unit = blockIdx.x * 4 + wv;
    for (int d = lane; d < RS_RADIX; d += 64) h[wv][d] = 0;
    __syncthreads();
    if (unit < nunits) {
        // (the tile's loads eight chunks at a time: with ~1.4 wavefronts per SIMD - 2.85 M rows are 1392 tiles - a loop of
        // dependent load -> atomic steps is a chain of 32 round trips)
        const i64 base = (i64)unit * SORT_WTILE;
        for (int c0 = 0; c0 < SORT_CHUNKS; c0 += 8) {
            RsElem<WIDE> e[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { const i64 row = base + (c0 + u) * 64 + lane; e[u] = in[row < n ? row : n - 1]; }
#pragma unroll
            for (int u = 0; u < 8; u++) { const i64 row = base + (c0 + u) * 64 + lane; if (row < n) atomicAdd(&h[wv][rs_digit<WIDE>(e[u], L, shift, dbits)], 1); }
        }
    }
    __syncthreads();
    if (unit < nunits)
        for (int d = lane; d < RS_RADIX; d += 64) hist[(i64)d * nunits + unit] = h[wv][d];
}
// stable scatter of the elements: every wavefront walks its tile chunk by chunk, in row order
template <bool WIDE> __global__ __launch_bounds__(256) void k_rs_scatter(const RsElem<WIDE>* in, RsElem<WIDE>* out, i64 n, int nunits, KeyLayout L, int shift, int dbits, const int* hist)
{
    __shared__ int cnt[4][RS_RADIX];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int unit = blockIdx.x * 4 + wv;
    if (unit < nunits)
        for (int d = lane; d < RS_RADIX; d += 64) cnt[wv][d] = hist[(i64)d * nunits + unit];
    __syncthreads();
    if (unit >= nunits) return;
    const i64 base = (i64)unit * SORT_WTILE;
    const u64 lt = (1ull << lane) - 1ull;
    constexpr int PF = WIDE ? 4 : 8;                                  // chunks loaded ahead (registers: 4 words per 16-byte element; 8 per 32-byte one)
    for (int c0 = 0; c0 < SORT_CHUNKS; c0 += PF) {
        RsElem<WIDE> pre[PF];
#pragma unroll
        for (int u = 0; u < PF; u++) {
            const i64 r2 = base + (c0 + u) * 64 + lane;
            // (a 32-byte element as two 16-byte words: copied as one aggregate it went through a stack slot - 64 B of scratch per lane)
            if constexpr (WIDE) { const ulonglong2* q = (const ulonglong2*)&in[r2 < n ? r2 : n - 1]; const ulonglong2 q0 = q[0], q1 = q[1]; pre[u].v.x = q0.x; pre[u].v.y = q0.y; pre[u].v.z = q1.x; pre[u].v.w = q1.y; }
            else pre[u] = in[r2 < n ? r2 : n - 1];
        }
#pragma unroll
        for (int u = 0; u < PF; u++) {
            const i64 row = base + (c0 + u) * 64 + lane;
            const bool in_ = row < n;
            const RsElem<WIDE> e = pre[u];
            const int dig = in_ ? rs_digit<WIDE>(e, L, shift, dbits) : 0;
            u64 differ = ~__ballot(in_);
#pragma unroll
            for (int bit = 0; bit < RS_BITS; bit++) {
                const int ones = (int)((unsigned)dig << (31 - bit)) >> 31;
                const u64 mk = __ballot(ones != 0);
                differ |= mk ^ (u64)(i64)ones;
            }
            const u64 match = ~differ;                                    // lanes of this chunk with my digit
            if (in_) {
                const int before = cnt[wv][dig];                          // same value for the whole match group
                const i64 dst = before + __popcll(match & lt);
                if constexpr (WIDE) { ulonglong2* o = (ulonglong2*)&out[dst]; o[0] = make_ulonglong2(e.v.x, e.v.y); o[1] = make_ulonglong2(e.v.z, e.v.w); }
                else out[dst] = e;
            }
            // the highest lane of each group advances the group's counter (after everyone read it: LDS ops of a wavefront execute in order)
            if (in_ && (match >> lane) <= 1ull) cnt[wv][dig] += __popcll(match);
            __builtin_amdgcn_wave_barrier();
        }
    }
}

struct RsTail {
    i64 n;
    KeyLayout L;
    const uint8_t* nodedup;     // per segment: keep every row (nullable)
    const uint8_t* drop;        // per sorted position: the caller's tie_order said "duplicate" (nullable)
    int* partial;               // per 2048-row tile counts
    int* o_seg; i64* o_a; i64* o_b; int* o_rid; int* o_aux; int* o_src;
    int* n_out;
};
template <bool WIDE> __device__ __forceinline__ int rs_seg_of(u128 key, const KeyLayout& L) { return (int)(u64)(key >> (L.rb + L.bb + L.ab + L.xb)); }
template <bool WIDE> __device__ __forceinline__ int rs_keep(const RsTail& R, const RsElem<WIDE>* el, i64 i)
{
    if (i >= R.n) return 0;
    if (i == 0) return 1;
    const RsElem<WIDE> e = el[i], q = el[i - 1];
    const u128 k = rs_key<WIDE>(e, R.L);
    if (R.nodedup && R.nodedup[rs_seg_of<WIDE>(k, R.L)]) return R.drop ? !R.drop[i] : 1;
    return !(k == rs_key<WIDE>(q, R.L) && rs_aux<WIDE>(e, R.L) == rs_aux<WIDE>(q, R.L));
}
template <bool WIDE> __global__ __launch_bounds__(256) void k_rs_count(RsTail R, const RsElem<WIDE>* el)
{
    const i64 base = (i64)blockIdx.x * 2048 + (threadIdx.x >> 6) * 512;
    int cnt = 0;
    for (int r = 0; r < 8; r++) cnt += __popcll(__ballot(rs_keep<WIDE>(R, el, base + r * 64 + (threadIdx.x & 63))));
    __shared__ int s[4];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) R.partial[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}
template <bool WIDE> __global__ __launch_bounds__(256) void k_rs_apply(RsTail R, const RsElem<WIDE>* el)
{
    __shared__ i64 sh[4];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const i64 base = (i64)blockIdx.x * 2048 + wv * 512;
    u64 masks[8]; int cnt = 0;
    for (int r = 0; r < 8; r++) { masks[r] = __ballot(rs_keep<WIDE>(R, el, base + r * 64 + lane)); cnt += __popcll(masks[r]); }
    int run = (int)block_prefix_of(R.partial, blockIdx.x, sh);
    __shared__ int s[4];
    if (lane == 0) s[wv] = cnt;
    __syncthreads();
    for (int k = 0; k < wv; k++) run += s[k];
    const KeyLayout& L = R.L;
    for (int r = 0; r < 8; r++) {
        const i64 i = base + r * 64 + lane;
        const u64 m = masks[r];
        if ((m >> lane) & 1) {
            const int o = run + __popcll(m & ((1ull << lane) - 1ull));
            const RsElem<WIDE> e = el[i];
            const u128 k = rs_key<WIDE>(e, L);
            R.o_rid[o] = (int)((u64)k & rs_mask(L.rb));
            R.o_b[o] = (i64)((u64)(k >> L.rb) & rs_mask(L.bb));
            R.o_a[o] = (i64)((u64)(k >> (L.rb + L.bb)) & rs_mask(L.ab));
            R.o_seg[o] = rs_seg_of<WIDE>(k, L);
            R.o_aux[o] = rs_aux<WIDE>(e, L);
            R.o_src[o] = rs_idx<WIDE>(e, L);
        }
        run += __popcll(m);
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 255) *R.n_out = run;
}
// tie groups of the keep-every-row segments (equal keys), as {position | continues << 31, source row}
template <bool WIDE> __global__ __launch_bounds__(256) void k_rs_ties(RsTail R, const RsElem<WIDE>* el, int2* list, int* n_list)
{
    const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    if (i >= R.n) return;
    const u128 k = rs_key<WIDE>(el[i], R.L);
    if (!R.nodedup[rs_seg_of<WIDE>(k, R.L)]) return;
    const bool cont = i > 0 && rs_key<WIDE>(el[i - 1], R.L) == k;
    const bool head = !cont && i + 1 < R.n && rs_key<WIDE>(el[i + 1], R.L) == k;
    if (cont || head) list[atomicAdd(n_list, 1)] = make_int2((int)i | (cont ? (int)0x80000000 : 0), rs_idx<WIDE>(el[i], R.L));
}
// the caller's answer: the element at pos[k] becomes source row src[k] (same key: only the row and its aux word change)
template <bool WIDE> __global__ __launch_bounds__(256) void k_rs_tie_apply(int n, const int* pos, const int* src, const uint8_t* flag, const int* aux_col,
                                                                           KeyLayout L, RsElem<WIDE>* el, uint8_t* drop)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    RsElem<WIDE> e = el[pos[k]];
    rs_set_row<WIDE>(e, L, src[k], aux_col[src[k]]);
    el[pos[k]] = e;
    drop[pos[k]] = flag[k];
}

// ---------------------------------------------------------------------------------------- gather + de-dup
struct RebuildArgs {
    i64 n;
    const int* perm;
    const int* seg; const i64* a; const i64* b; const int* rid; const int* aux; const int* auxk;
    const uint8_t* nodedup;     // per segment: keep every row (nullable)
    const uint8_t* drop;        // per sorted position, keep-every-row segments only: the caller's tie_order said "duplicate" (nullable)
    int* keep;             // 1 when the sorted row differs from its predecessor
    int* partial;          // per 2048-row tile counts
    int* o_seg; i64* o_a; i64* o_b; int* o_rid; int* o_aux; int* o_src;
    int* n_out;
};

__device__ __forceinline__ int rebuild_keep(const RebuildArgs& R, i64 i)
{
    if (i >= R.n) return 0;
    if (i == 0) return 1;
    const int p = R.perm ? R.perm[i] : (int)i, q = R.perm ? R.perm[i - 1] : (int)(i - 1);
    if (R.nodedup && R.nodedup[R.seg[p]]) return R.drop ? !R.drop[i] : 1;
    return !(R.seg[p] == R.seg[q] && R.a[p] == R.a[q] && R.b[p] == R.b[q] && R.rid[p] == R.rid[q] && R.aux[p] == R.aux[q]);
}

__global__ __launch_bounds__(256) void k_rebuild_count(RebuildArgs R)
{
    const i64 base = (i64)blockIdx.x * 2048 + (threadIdx.x >> 6) * 512;
    int cnt = 0;
    for (int r = 0; r < 8; r++) cnt += __popcll(__ballot(rebuild_keep(R, base + r * 64 + (threadIdx.x & 63))));
    __shared__ int s[4];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) R.partial[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}

__global__ __launch_bounds__(256) void k_rebuild_apply(RebuildArgs R)
{
    __shared__ i64 sh[4];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const i64 base = (i64)blockIdx.x * 2048 + wv * 512;
    u64 masks[8]; int cnt = 0;
    for (int r = 0; r < 8; r++) { masks[r] = __ballot(rebuild_keep(R, base + r * 64 + lane)); cnt += __popcll(masks[r]); }
    int run = (int)block_prefix_of(R.partial, blockIdx.x, sh);
    __shared__ int s[4];
    if (lane == 0) s[wv] = cnt;
    __syncthreads();
    for (int k = 0; k < wv; k++) run += s[k];
    for (int r = 0; r < 8; r++) {
        const i64 i = base + r * 64 + lane;
        const u64 m = masks[r];
        if ((m >> lane) & 1) {
            const int o = run + __popcll(m & ((1ull << lane) - 1ull));
            const int p = R.perm ? R.perm[i] : (int)i;
            R.o_seg[o] = R.seg[p]; R.o_a[o] = R.a[p]; R.o_b[o] = R.b[p]; R.o_rid[o] = R.rid[p]; R.o_aux[o] = R.aux[p]; R.o_src[o] = p;
        }
        run += __popcll(m);
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 255) *R.n_out = run;
}

// The tie groups of the keep-every-row segments in the sorted permutation: position i ties with i - 1 when (segment, a, b, read
// id) agree.  Every position that belongs to a group is appended to `list` as {i | continues << 31, source row}; the order of
// the appends does not matter (a few rows per genome: the host sorts them by position).
__global__ __launch_bounds__(256) void k_rebuild_ties(RebuildArgs R, int2* list, int* n_list)
{
    const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    if (i >= R.n) return;
    auto ties = [&](i64 x) -> bool {                        // position x continues the group of x - 1
        if (x <= 0 || x >= R.n) return false;
        const int p = R.perm ? R.perm[x] : (int)x, q = R.perm ? R.perm[x - 1] : (int)(x - 1);
        return R.nodedup[R.seg[p]] && R.seg[p] == R.seg[q] && R.a[p] == R.a[q] && R.b[p] == R.b[q] && R.rid[p] == R.rid[q];
    };
    const bool cont = ties(i), head = !cont && ties(i + 1);
    if (cont || head) {
        const int k = atomicAdd(n_list, 1);
        list[k] = make_int2((int)i | (cont ? (int)0x80000000 : 0), R.perm ? R.perm[i] : (int)i);
    }
}
// the caller's answer: perm[pos[k]] = src[k], drop[pos[k]] = flag[k]
__global__ __launch_bounds__(256) void k_rebuild_tie_apply(int n, const int* pos, const int* src, const uint8_t* flag, int* perm, uint8_t* drop)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k < n) { perm[pos[k]] = src[k]; drop[pos[k]] = flag[k]; }
}

// rows per segment of the sorted, de-duplicated output (o_seg ascends): one thread per segment, two binary searches; and the
// number of rows in keep-every-row segments that agree with their predecessor in (segment, a, b, read id) - the INS tie
// groups whose order depends on the inserted sequences, which only the host has
__global__ __launch_bounds__(256) void k_rebuild_segcount(RebuildArgs R, int n_seg, i64* seg_count, i64* n_ties)
{
    const int n = *R.n_out;
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s < n_seg) {
        int lo = 0, hi = n;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (R.o_seg[mid] < s) lo = mid + 1; else hi = mid; }
        const int first = lo;
        hi = n;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (R.o_seg[mid] <= s) lo = mid + 1; else hi = mid; }
        seg_count[s] = lo - first;
    }
    if (R.nodedup) {
        int ties = 0;
        for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x + 1; i < n; i += (i64)gridDim.x * 256)
            ties += R.nodedup[R.o_seg[i]] && R.o_seg[i] == R.o_seg[i - 1] && R.o_a[i] == R.o_a[i - 1] && R.o_b[i] == R.o_b[i - 1] && R.o_rid[i] == R.o_rid[i - 1];
        ties = wave_sum_i32(ties);
        if ((threadIdx.x & 63) == 0 && ties) atomicAdd((unsigned long long*)n_ties, (unsigned long long)ties);
    }
}

// effective aux key: aux sorts before pos only for INV (strand) and TRA (chr2, type) segments
__global__ __launch_bounds__(256) void k_rebuild_auxkey(i64 n, const int* seg, const int* aux, const uint8_t* seg_aux_major, int* auxk)
{
    const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    if (i < n) auxk[i] = seg_aux_major[seg[i]] ? aux[i] : 0;
}

// CSV_RB_FROM_POOL: the pool's rows become the rebuild's input columns; the read index of a row is replaced by the rank of the
// read's NAME (read_rank, the caller's: string order is the host's business), and the largest value of every key column is
// collected for the radix passes (mx: a, b, read id, aux of aux-major segments, segment; [5] != 0: a row the sort cannot take).
struct PoolCols;
__global__ __launch_bounds__(256) void k_pool_to_rows(const int* p_seg, const i64* p_a, const i64* p_b, const int* p_read, const int* p_aux, i64 n,
                                                      const int* rank, i64 n_rank, int n_seg, const uint8_t* seg_aux_major,
                                                      int* o_seg, i64* o_a, i64* o_b, int* o_rid, int* o_aux, unsigned long long* mx)
{
    __shared__ unsigned long long sh[4][7];
    unsigned long long m[7] = {0, 0, 0, 0, 0, 0, 0};                 // [6]: aux of every row (rides along with the composite key)
    for (i64 i = (i64)blockIdx.x * 2048 + threadIdx.x; i < n && i < (i64)(blockIdx.x + 1) * 2048; i += 256) {
        const int sg = p_seg[i], rd = p_read[i], ax = p_aux[i];
        const i64 a = p_a[i], b = p_b[i];
        int rk = -1;
        if (rank) { if (rd >= 0 && rd < n_rank) rk = rank[rd]; }
        else rk = rd;                                       // (rows from the host: the read column is the rank already)
        const bool bad = sg < 0 || sg >= n_seg || a < 0 || b < 0 || rk < 0 || ax < 0;
        o_seg[i] = bad ? 0 : sg; o_a[i] = a; o_b[i] = b; o_rid[i] = rk; o_aux[i] = ax;
        if (bad) m[5] = 1;
        else {
            m[0] = (unsigned long long)a > m[0] ? (unsigned long long)a : m[0];
            m[1] = (unsigned long long)b > m[1] ? (unsigned long long)b : m[1];
            m[2] = (unsigned long long)rk > m[2] ? (unsigned long long)rk : m[2];
            if (seg_aux_major[sg]) m[3] = (unsigned long long)ax > m[3] ? (unsigned long long)ax : m[3];
            m[4] = (unsigned long long)sg > m[4] ? (unsigned long long)sg : m[4];
            m[6] = (unsigned long long)ax > m[6] ? (unsigned long long)ax : m[6];
        }
    }
#pragma unroll
    for (int k = 0; k < 7; k++) {
        for (int d = 32; d > 0; d >>= 1) {
            const unsigned lo = __shfl_xor((unsigned)(m[k] & 0xffffffffull), d), hi = __shfl_xor((unsigned)(m[k] >> 32), d);
            const unsigned long long o = ((unsigned long long)hi << 32) | lo;
            m[k] = o > m[k] ? o : m[k];
        }
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][k] = m[k];
    }
    __syncthreads();
    if (threadIdx.x < 7) {
        unsigned long long v = sh[0][threadIdx.x];
        for (int w = 1; w < 4; w++) v = sh[w][threadIdx.x] > v ? sh[w][threadIdx.x] : v;
        if (v) atomicMax(&mx[threadIdx.x], v);
    }
}

}  // namespace csv
