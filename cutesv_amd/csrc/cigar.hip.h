// cigar.hip.h — SURVEY.md §8f row 4: the per-read CIGAR scan of cuteSV's extraction step on the GPU.
//
// Restates the CIGAR part of parse_read (cuteSV main script :606-655) and generate_combine_sigs (:515-575): every
// insertion / deletion operation of at least min_siglength bases is a piece; pieces of one type that lie within
// merge_ins_threshold / merge_del_threshold of each other inside a read are merged into one signature.  BAM decode and
// the sequences stay with pysam in the Python driver (north_star; the split-read analysis is split.hip.h): the input
// is the flat, BAM-encoded CIGAR array of a batch of reads, the output the INS / DEL signatures (+ the query slices an
// INS sequence is made of), in read order.
//
//   reference loop (:629-643)                                   here
//   shift_ins_read += oplen unless op == D   (also N, H, P)      q-advance per op, wave inclusive scan
//   sig_start += oplen for M, D, N, =, X                         r-advance per op, wave exclusive scan
//   oplen >= min_siglength and op in {I, D} -> a piece           ballot; pieces walked in order by the wavefront
//   generate_combine_sigs: INS distance = pos - previous piece's pos; DEL distance = pos - (end of the previous piece
//   when that piece was merged or was the read's first, else its START: `temp_sig.append(i[0])`, :569)
//
// One wavefront per read, 64 operations per step (a 256-byte coalesced load); two passes (count, prefix over reads,
// emit) because the outputs are dense and in read order.  HBM-bound: 4 bytes per operation, read twice.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace csv {

struct CigarArgs {
    i64 n_reads;
    const i64* cig_off;
    const unsigned* cigar;          // BAM encoding: oplen << 4 | op  (M I D N S H P = X B: 0..9)
    const i64* ref_start;
    const uint8_t* use;             // nullable
    int min_siglength;
    i64 merge_ins, merge_del;
    // per read counts (pass 1) / exclusive offsets (after the scan): x = INS signatures, y = INS pieces, z = DEL signatures
    int4* cnt;
    i64* tile_sum;                  // 3 per tile of CG_TILE reads
    i64* totals;                    // 3
    // outputs
    int* ins_read; i64* ins_pos; i64* ins_len; i64* ins_piece0; int* ins_npiece;
    int* piece_qoff; int* piece_len;
    int* del_read; i64* del_pos; i64* del_len;
};

constexpr int CG_TILE = 1024;       // reads per scan tile

// the walk over one read; EMIT = false: count only.  Returns {INS signatures, INS pieces, DEL signatures}.
template <bool EMIT> __device__ __forceinline__ int4 cigar_read(const CigarArgs& A, i64 read, i64 o_isig, i64 o_piece, i64 o_dsig)
{
    const i64 c0 = A.cig_off[read], c1 = A.cig_off[read + 1];
    if (c1 <= c0 || (A.use && !A.use[read])) return make_int4(0, 0, 0, 0);
    const int lane = lane_id();
    const unsigned first = A.cigar[c0];
    i64 refpos = A.ref_start[read];                          // sig_start before the chunk
    i64 shift = (first & 15u) == 5u ? -(i64)(first >> 4) : 0; // shift_ins_read = -hardclip_left (:621-627)
    int n_isig = 0, n_piece = 0, n_dsig = 0;
    // running state of generate_combine_sigs, wave-uniform
    bool i_open = false, d_open = false;
    i64 i_last = 0, i_pos = 0, i_len = 0, i_p0 = 0, d_last = 0, d_pos = 0, d_len = 0;
    int i_np = 0;
    for (i64 base = c0; base < c1; base += 64) {
        const i64 idx = base + lane;
        const unsigned w = idx < c1 ? A.cigar[idx] : 0u;
        const int op = (int)(w & 15u);
        const i64 len = (i64)(w >> 4);
        const bool in = idx < c1;
        const bool refch = in && (op == 0 || op == 2 || op == 3 || op == 7 || op == 8);     // CHANGETABLE[...][1] (:589-600)
        const i64 radv = refch ? len : 0, qadv = (in && op != 2) ? len : 0;
        const i64 rinc = wave_incl_scan_i64(radv), qinc = wave_incl_scan_i64(qadv);
        const i64 my_ref = refpos + rinc - radv;             // sig_start when the op is reached
        const i64 my_shift = shift + qinc;                   // shift_ins_read after the op
        const bool piece = in && len >= A.min_siglength && (op == 1 || op == 2);
        for (u64 mk = __ballot(piece); mk; mk &= mk - 1) {   // pieces in order; everything below is wave-uniform
            const int l = __ffsll((long long)mk) - 1;
            const int pop = __builtin_amdgcn_readlane(op, l);
            const i64 ppos = readlane_i64x(my_ref, l), plen = readlane_i64x(len, l);
            if (pop == 1) {
                const i64 qoff = readlane_i64x(my_shift, l) - plen;
                if (i_open && ppos - i_last <= A.merge_ins) { i_len += plen; i_np++; }       // (:535-538)
                else {
                    if (i_open) {
                        if (EMIT && lane == 0) { const i64 k = o_isig + n_isig; A.ins_read[k] = (int)read; A.ins_pos[k] = i_pos; A.ins_len[k] = i_len; A.ins_piece0[k] = i_p0; A.ins_npiece[k] = i_np; }
                        n_isig++;
                    }
                    i_open = true; i_pos = ppos; i_len = plen; i_p0 = o_piece + n_piece; i_np = 1;
                }
                i_last = ppos;
                if (EMIT && lane == 0) { A.piece_qoff[o_piece + n_piece] = (int)qoff; A.piece_len[o_piece + n_piece] = (int)plen; }
                n_piece++;
            } else {
                if (d_open && ppos - d_last <= A.merge_del) { d_len += plen; d_last = ppos + plen; }   // (:558-560)
                else {
                    const bool was_open = d_open;
                    if (d_open) {
                        if (EMIT && lane == 0) { const i64 k = o_dsig + n_dsig; A.del_read[k] = (int)read; A.del_pos[k] = d_pos; A.del_len[k] = d_len; }
                        n_dsig++;
                    }
                    d_open = true; d_pos = ppos; d_len = plen;
                    d_last = was_open ? ppos : ppos + plen;  // first piece: sum(sigs[0]) (:555); after a cut: i[0] (:569)
                }
            }
        }
        refpos += readlane_i64x(rinc, 63);
        shift += readlane_i64x(qinc, 63);
    }
    if (i_open) {
        if (EMIT && lane == 0) { const i64 k = o_isig + n_isig; A.ins_read[k] = (int)read; A.ins_pos[k] = i_pos; A.ins_len[k] = i_len; A.ins_piece0[k] = i_p0; A.ins_npiece[k] = i_np; }
        n_isig++;
    }
    if (d_open) {
        if (EMIT && lane == 0) { const i64 k = o_dsig + n_dsig; A.del_read[k] = (int)read; A.del_pos[k] = d_pos; A.del_len[k] = d_len; }
        n_dsig++;
    }
    return make_int4(n_isig, n_piece, n_dsig, 0);
}

__global__ __launch_bounds__(256) void k_cigar_count(CigarArgs A)
{
    const i64 wave = ((i64)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = ((i64)gridDim.x * 256) >> 6;
    for (i64 r = wave; r < A.n_reads; r += nwaves) {
        const int4 c = cigar_read<false>(A, r, 0, 0, 0);
        if (lane_id() == 0) A.cnt[r] = c;
    }
}

// per tile of CG_TILE reads: sums of the three counts
__global__ __launch_bounds__(256) void k_cigar_tiles(CigarArgs A)
{
    const i64 base = (i64)blockIdx.x * CG_TILE;
    i64 s0 = 0, s1 = 0, s2 = 0;
    for (int i = threadIdx.x; i < CG_TILE; i += 256) {
        const i64 r = base + i;
        if (r < A.n_reads) { const int4 c = A.cnt[r]; s0 += c.x; s1 += c.y; s2 += c.z; }
    }
    s0 = wave_sum_i64(s0); s1 = wave_sum_i64(s1); s2 = wave_sum_i64(s2);
    __shared__ i64 sh[12];
    if (lane_id() == 0) { sh[(threadIdx.x >> 6)] = s0; sh[4 + (threadIdx.x >> 6)] = s1; sh[8 + (threadIdx.x >> 6)] = s2; }
    __syncthreads();
    if (threadIdx.x < 3) A.tile_sum[(i64)blockIdx.x * 3 + threadIdx.x] = sh[4 * threadIdx.x] + sh[4 * threadIdx.x + 1] + sh[4 * threadIdx.x + 2] + sh[4 * threadIdx.x + 3];
}

// counts -> exclusive offsets, in place (the tile's prefix is recomputed from the tile sums: a few thousand values)
__global__ __launch_bounds__(256) void k_cigar_offsets(CigarArgs A)
{
    __shared__ i64 sh[12], carry[3];
    __shared__ i64 ws[3][4];
    i64 p[3] = {0, 0, 0};
    for (int t = threadIdx.x; t < (int)blockIdx.x; t += 256)
        for (int k = 0; k < 3; k++) p[k] += A.tile_sum[(i64)t * 3 + k];
    for (int k = 0; k < 3; k++) { p[k] = wave_sum_i64(p[k]); if (lane_id() == 0) sh[4 * k + (threadIdx.x >> 6)] = p[k]; }
    __syncthreads();
    if (threadIdx.x < 3) carry[threadIdx.x] = sh[4 * threadIdx.x] + sh[4 * threadIdx.x + 1] + sh[4 * threadIdx.x + 2] + sh[4 * threadIdx.x + 3];
    __syncthreads();
    const i64 base = (i64)blockIdx.x * CG_TILE;
    for (int b0 = 0; b0 < CG_TILE; b0 += 256) {
        const i64 r = base + b0 + threadIdx.x;
        int4 c = make_int4(0, 0, 0, 0);
        if (r < A.n_reads) c = A.cnt[r];
        const i64 v[3] = {c.x, c.y, c.z};
        i64 inc[3];
        for (int k = 0; k < 3; k++) { inc[k] = wave_incl_scan_i64(v[k]); if (lane_id() == 63) ws[k][threadIdx.x >> 6] = inc[k]; }
        __syncthreads();
        i64 off[3];
        for (int k = 0; k < 3; k++) {
            off[k] = carry[k];
            for (int q = 0; q < (int)(threadIdx.x >> 6); q++) off[k] += ws[k][q];
            off[k] += inc[k] - v[k];
        }
        // offsets can exceed 32 bits only beyond 2^31 signatures per batch: rejected by the host
        if (r < A.n_reads) A.cnt[r] = make_int4((int)off[0], (int)off[1], (int)off[2], 0);
        __syncthreads();
        if (threadIdx.x == 255) for (int k = 0; k < 3; k++) carry[k] = off[k] + v[k];
        __syncthreads();
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) for (int k = 0; k < 3; k++) A.totals[k] = carry[k];
}

__global__ __launch_bounds__(256) void k_cigar_emit(CigarArgs A)
{
    const i64 wave = ((i64)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = ((i64)gridDim.x * 256) >> 6;
    for (i64 r = wave; r < A.n_reads; r += nwaves) {
        const int4 o = A.cnt[r];
        cigar_read<true>(A, r, o.x, o.y, o.z);
    }
}

// ------------------------------------------------------------------------------------ device-resident signature pool
// CSV_CG_TO_POOL: the INS / DEL signatures of the batch become rows of the context's pool (segment, position, length, global
// read index, aux) - what the rebuild step sorts (main script :750-857) - without a trip through host memory.  aux of an
// INS row is the length of the inserted sequence the caller would cut out of the read (candidates(): the pieces, each
// clipped to the query sequence like a Python slice).
struct PoolCols { int* seg; i64* a; i64* b; int* read; int* aux; };
__global__ __launch_bounds__(256) void k_pool_from_cigar(PoolCols P, i64 base, CigarArgs A, i64 n_ins, i64 n_del, int seg_ins, int seg_del, i64 read_base,
                                                         const int* query_len)
{
    const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    if (i < n_ins) {
        const int r = A.ins_read[i];
        i64 seq = 0;
        if (query_len) {
            const i64 ql = query_len[r], p0 = A.ins_piece0[i];
            for (int k = 0; k < A.ins_npiece[i]; k++) {
                const i64 q = A.piece_qoff[p0 + k], l = A.piece_len[p0 + k];
                const i64 lo = q < ql ? q : ql, hi = q + l < ql ? q + l : ql;     // query_sequence[q : q + l]
                seq += hi > lo ? hi - lo : 0;
            }
        } else seq = A.ins_len[i];
        const i64 o = base + i;
        P.seg[o] = seg_ins; P.a[o] = A.ins_pos[i]; P.b[o] = A.ins_len[i]; P.read[o] = (int)(read_base + r); P.aux[o] = (int)seq;
    } else if (i < n_ins + n_del) {
        const i64 j = i - n_ins, o = base + i;
        P.seg[o] = seg_del; P.a[o] = A.del_pos[j]; P.b[o] = A.del_len[j]; P.read[o] = (int)(read_base + A.del_read[j]); P.aux[o] = 0;
    }
}

}  // namespace csv
