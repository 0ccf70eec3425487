// kernels.hip.h — device code of libcutesv_hip.so (gfx950 / CDNA4 only, wave64).
//
// Pipeline of one csv_batch_run (no host round trip in between; side streams for the pair tier, the big tiers and the reads):
//   chain     k_chain_count                    break flags, cluster starts per wavefront in LDS, the size gate (once) ->
//                                              item records per wavefront
//             k_chain_apply                    compaction: ordered work list + three tier lists; tier counts to the host
//             k_chain_ids                      per-signature cluster ids (CSV_IN_PER_SIG batches only)
//   refine    k_refine_indel_wave              DEL/INS clusters of m <= 64 in registers: four (m <= 16), two (m <= 32) or
//                                              one per wavefront; cross-lane ops only
//             k_refine<64,64>                  one wavefront per DUP/INV/TRA cluster (m <= 64), arrays in LDS
//             k_refine<64,256>                 one wavefront per cluster of 64 < m <= 256, arrays in LDS
//             k_refine<256,2048>               one workgroup per cluster; LDS up to 2048 padded elements, global scratch above
//   order     k_items_scan, k_emit             per-item counts -> two-level prefix -> dense, ordered call records + supports
//   reads     k_reads_runs / _plan / _gather   start order of every reads block (whole sorted runs move)
//             k_pmax_count / _scan             prefix max of read ends, span-local + one scan over the span maxima
//   genotype  k_genotype<1024,4>, <8192,4>     one wavefront per call: 64-ary search, backwards stabbing scan, LDS hash
//                                              set (4 KB, then 32 KB, then a global pool); k_genotype_tra for TRA calls
//   results   k_publish                        calls + supports straight into page-locked caller arrays
//
// Reference semantics and file:line citations are in oracle/cutesv_oracle.c (the CPU restatement these
// kernels are tested against) and include/cutesv_hip.h.  The path is integer sort/scan plus a little
// float64; there is no MFMA work in it.  Compile with -ffp-contract=off: float64 results must equal numpy's.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "../../include/cutesv_hip.h"

// Measurement aid (scripts/ablate.sh): -DCSV_ABLATE=<bit mask> drops a section of a kernel so that its share of the
// kernel time can be read off a same-box A/B run.  The results are wrong then; never defined in a product build.
#ifdef CSV_ABLATE
#define CSV_ABL(bit) (((CSV_ABLATE) >> (bit)) & 1)
#else
#define CSV_ABL(bit) 0
#endif
#define CSV_LIKELY(x) __builtin_expect(!!(x), 1)
#define CSV_UNLIKELY(x) __builtin_expect(!!(x), 0)

namespace csv {

typedef unsigned long long u64;
typedef long long i64;

constexpr int WAVE = 64;
constexpr u64 PAD_KEY = ~0ull;
constexpr int IDX_BITS = 21;                       // local-index bits inside a sort key: (value << 21 | index), values < 2^42 ...
constexpr int IDX_BITS_HUGE = 31;                  // ... and (value << 31 | index), values < 2^32, for chained clusters above 2^21 signatures
constexpr i64 MAX_CLUSTER = (1ll << 30);           // (padded size must fit an int; a batch holds < 2^31 signatures anyway)
constexpr int SQRT_TAB = 65536;                    // pow(n, 0.5) as glibc computes it: the table holds at least n < SQRT_TAB and always every
                                                   // n up to the longest segment of the batch (an allele cannot be larger): no sqrt() stand-in
constexpr int ARR_PAD = 8;                         // group-start arrays need P + 1 entries

// device error bits (DevCounters::error)
enum { ERR_CLUSTER_TOO_BIG = 1, ERR_READS_UNSORTED = 2, ERR_COVER_OVERFLOW = 4, ERR_KEY_RANGE = 8, ERR_TMP_OVERFLOW = 16, ERR_SIG_ORDER = 32, ERR_TRA_CHROM = 64 };

// One call as the device keeps it: written by k_emit with six 16-byte stores, read back by the genotype kernels with
// four, copied to the host in one piece.  (Fourteen separate arrays cost k_emit 9.6x its algorithmic bytes in
// scattered 4/8-byte stores and every consumer one dependent load per field.)
struct __attribute__((aligned(16))) CallRec {
    i64 bp1, bp2;                     //  0
    i64 search, pick;                 // 16
    i64 supoff; int support, cipos;   // 32
    int cilen, seg, cluster, aux;     // 48
    int dr, dv, gl, pad;              // 64
    int chrom, type_gt, bias_lo, bias_hi;   // 80: {chromosome, svtype | genotype << 8, gt_bias}: k_genotype reads no segment record
};
static_assert(sizeof(CallRec) == 96, "CallRec layout");

// A temp call record (one per allele / sub-cluster): written by the refine kernels with four 16-byte stores (the whole
// 64-byte line: nothing to merge), read by k_emit as one line.  Slot 0 of item j lives in the DENSE, item-indexed array
// t_rec0[j] - 99 % of the items have exactly one slot, so neighbouring items' records share 128-byte lines on the write and
// on the read side, and k_emit's load of it depends on nothing but the item index (r03: every record sat alone in its
// cluster's own signature range, t_rec[first w + slot]: one 64-byte line touched per 128-byte fetch, and a dependent round
// trip behind the item record).  Slots >= 1 (a second allele) stay at t_rec[first w + slot].  Slot 0 also carries what
// k_emit needs to know about the ITEM: its slot count and the aux word of the cluster's first signature.
// (Nine separate arrays made k_emit fetch nine lines per slot: 9.6x its algorithmic bytes.)
struct __attribute__((aligned(16))) TmpRec {
    i64 bp1, bp2;                     //  0
    i64 search, pick;                 // 16
    int support, cipos, cilen, supoff;   // 32
    int valid, nslots, aux0, pad2;    // 48 (valid: bit 0 the call exists, bit 1 `pick` is a row in w space - k_emit adds the segment's offset; nslots / aux0: slot 0 only)
};
static_assert(sizeof(TmpRec) == 64, "TmpRec layout");
__device__ __forceinline__ void tmp_write(TmpRec* t, i64 bp1, i64 bp2, i64 search, i64 pick, int support, int cipos, int cilen, int supoff, int valid,
                                          int nslots = 0, int aux0 = 0)
{
    int4* r = (int4*)t;
    r[0] = make_int4((int)(bp1 & 0xffffffffll), (int)(bp1 >> 32), (int)(bp2 & 0xffffffffll), (int)(bp2 >> 32));
    r[1] = make_int4((int)(search & 0xffffffffll), (int)(search >> 32), (int)(pick & 0xffffffffll), (int)(pick >> 32));
    r[2] = make_int4(support, cipos, cilen, supoff);
    r[3] = make_int4(valid, nslots, aux0, 0);
}

struct DevCounters {          // one small struct in device memory, zeroed at the start of every run
    int n_clusters;
    int n_items;              // valid clusters (work items)
    int n_items_big;          // of those, handled by the workgroup tier
    int n_tmp_calls;          // allocated temp call slots
    int n_calls;
    int error;
    i64 n_support;
    int n_gt_over;            // calls whose support + cover did not fit the small hash set
    int n_items_tiny;         // DEL/INS work items of at most tiny_max signatures (four per wavefront)
    int n_items_wide;         // DEL/INS work items of 33 .. 64 signatures (they are in the small list too): one per wavefront
    int n_runs;               // reads_order: sorted runs found in the reads table
    int ro_state;             // reads_order: RO_*
    int n_gt_huge;            // genotype: calls that need the whole global pool
    int gt_ticket;            // workgroups of the second genotype pass that have finished
    int n_tra_huge;           // the same two for k_genotype_tra
    int tra_ticket;
    int n_items_huge;         // work items above 256 signatures (the workgroup tier's; counted by k_chain_apply)
};
enum { RO_REORDER = 0, RO_IDENTITY = 1, RO_NEED_GENERAL = 2 };
// What the reads-order stage found out about the reads table of the UPLOAD: unlike DevCounters it is not zeroed by every run (a
// resident re-run that keeps the packed table - CSV_OPT_REUSE_READS_ORDER - must still see that the table needed the general
// sort); zeroed at upload and whenever the stage is run again.
struct ReadsState { int n_runs; int ro_state; int error; int pad; };      // error: ERR_* bits of the table's validation

// A position / length column as the caller sent it: int64, or int32 (CSV_IN_SIG_I32 / CSV_IN_READS_I32: a genome's coordinates
// fit 31 bits) - consumed as it is, never widened into a second copy.  The pointer test is wave-uniform (kernel argument).
// The hot kernels are instantiated per width (NARROW) and go through col_at<> instead.
struct Col {
    const i64* p64;
    const int* p32;
    __device__ __forceinline__ i64 operator[](i64 i) const { return p32 ? (i64)p32[i] : p64[i]; }
};
// coordinates of the packed reads table and its indices: the caller's width
template <bool RN> using rd_t = typename std::conditional<RN, int, i64>::type;
template <bool NARROW> __device__ __forceinline__ i64 col_at(const Col& c, i64 i) { if constexpr (NARROW) return (i64)c.p32[i]; else return c.p64[i]; }

// Everything the kernels need, passed by value.
struct DevBatch {
    int            n_seg;
    int            n_chrom;
    i64            W;                // signatures in the batch (compact "w" space)
    const csv_segment* seg;          // device copy of the segment table
    const i64*     woff;             // n_seg + 1 prefix of segment lengths
    const uint8_t* seg_drop;         // genotype requested but no reads block (INDEL:443-444)
    Col            a;                // int(pos) / pos1
    Col            b;                // length / pos2
    const int*     rid;
    const int*     aux;
    // chain / select
    int*           cluster_id;       // W
    int*           partial;          // cluster starts per chain tile
    int4*          tile_cnt;         // per chain tile: {cluster starts, work items, items above 64 signatures, tiny | wide << 16}
    int4*          item_rec;         // ordered work list: item -> {cluster id, segment, first w, size}
    int4*          list_small;       // wavefront tier (ordered): {item, segment | svtype << 24, first w, size} - everything the
                                     // refine kernels need to issue their row loads straight after this ONE load
    int*           list_big;
    int4*          list_tiny;        // DEL/INS items with m <= tiny_max (same entries): k_refine_indel_wave packs four per wavefront
    int4*          list_wide;        // DEL/INS items with 32 < m <= 64 (same entries, any order): their own units, so that no
                                     // wavefront runs a pair and then its wide members one after the other
    int            tiny_max;         // 16 (0 switches the class off)
    int            pair_in_mid;      // this run's k_refine<64,256> launch also refines the DUP / INV / TRA clusters of at most 64 signatures
                                     // (second phase; k_refine<64,64> is not launched then)
    u64*           ch_masks;         // per chain tile: its 32 flag words (written only for CSV_IN_PER_SIG: k_chain_ids)
    int4*          tile_items;       // per chain tile, TI_STRIDE slots: the clusters that passed the size gate, in order:
                                     // {first w, size, segment | svtype << 24 | tier << 28, cluster index inside the tile}
    int            per_sig;          // CSV_IN_PER_SIG: cluster_id / allele_id are produced
    int            end_z;            // the batch's last signature is a (0,0) element (looked up by the host at upload)
    int*           host_flag;        // page-locked host word {run sequence << 32 | work items above 64 signatures}: later runs of
    int            run_seq;          // the same upload launch only the tiers that have work
    int*           seg_err;          // n_seg words of CSV_SEG_* bits
    const int4*    seg_gate;         // per segment {read_count, dropped, svtype, -}: one 16-byte load for the size gate
    const int4*    tile_info;        // per chain tile, built by the host: TILE_REC int4 words (see TILE_REC below)
    // gate-first one-shot call (csv_cluster_batch from page-locked columns): only the position column crosses PCIe in bulk; what
    // the refine kernels read of the OTHER columns - the rows of the clusters that pass the size gate, ~18 % of a 30x genome -
    // is fetched by k_lazy_fetch straight out of the caller's host columns (device-visible addresses, indexed by the caller's
    // global signature index) after the chain kernels have said which rows those are.  NULL: the columns were copied whole.
    const void*    h_b;              // int32 / int64 like `b`
    const int*     h_rid;
    const int*     h_aux;
    const int2*    h_rows8;          // {b, read_id} interleaved (csv_batch_in.rows8) or NULL: then h_b / h_rid
    int*           tile_lead;        // per chain tile: tile-relative position of its first cluster start (CH_TILE: none) - the rows
                                     // before it belong to a cluster that started in an earlier tile
    // refine outputs
    TmpRec*        t_rec0;           // cap_items: slot 0 of every item (+ its slot count and first aux word), item-indexed
    i64*           item_cnt;         // packed (valid calls << 32 | supports of valid calls)
    i64*           item_base;        // exclusive prefix of item_cnt per tile of EM_TILE items, inside the tile's chunk of IS_CHUNK items
    i64*           item_chunk;       // total of item_cnt per chunk
    int*           sup_tmp;          // W: support lists (signature w), stored inside the cluster's own [s, e) range.  (r03 carried the read id along:
                                     // 8 B per entry written and read back for a word only genotyped segments need - k_emit gathers it there)
    TmpRec*        t_rec;            // W temp call records: slots >= 1 of an item live in its own signature range, t_rec[first w + slot]
    int            cap_tmp;
    int            cap_items;
    // big-cluster scratch (2 * W + 16 elements each)
    u64*           sc_k; i64* sc_x; int* sc_v1; int* sc_v2; int* sc_v3; int* sc_v4; int* sc_v5;
    // final outputs: one 96-byte record per call (the host unpacks it into csv_batch_out's arrays after ONE copy)
    CallRec*       o_rec;
    int*           o_supsig;         // global signature index (a batch holds fewer than 2^31 signatures)
    int*           o_suprid;         // read id of the support (genotype)
    int*           allele_id;        // W
    // reads
    const i64*     reads_off;
    i64            n_reads;
    Col            r_start; Col r_end; const uint8_t* r_primary; const int* r_id;            // as uploaded (any order inside a block; int32 or int64)
    // the packed, start-ordered table k_reads_gather writes (columns in the caller's width): what the genotype kernels read
    i64*           s_start64; i64* s_end64; int* s_start32; int* s_end32;
    int*           s_idp;            // read id | primary << 31
    void*          cmax;             // per chunk of 64 reads: the largest end                                   } in the table's own
    void*          cfirst;           // per chunk: the start of its first read; bfirst[k] = cfirst[64 k] (per block of 4096 reads)  } width (rd_t)
    void*          bfirst;
    i64*           span_len;         // per span of 512 reads: the longest read ...
    i64*           maxlen;           // ... and per chromosome (k_reads_maxlen): bounds how far before a window a covering read can start
    int            ro_mode;          // 0: caller promised sorted blocks; 1: run-level reorder on the device; 2: general radix sort (fallback)
    int*           ro_tcnt;          // per tile of RO_TILE rows: run starts found by k_reads_runs ...
    const int*     ro_tblk;          // per tile: the first chromosome block that begins at or after its first row (host-built)
    int4*          ro_ent;           // ... and their records {row, its start, the start of the row before, out of range}, RO_TCAP per tile
    int4*          ro_table;         // runs in start order: {source begin, length, destination begin, chromosome}
    int            ro_cap;           // capacity of both
    i64            ro_gap;           // a jump of more than this many bases between neighbours also starts a run
    const int*     ro_perm;          // mode 2: sorted position -> uploaded row
    int*           gt_huge;          // calls whose sets overflow the 32 KB tables AND one wavefront's slice of the global pool
    int*           gt_pool;          // global hash pool: gt_pool_n ints, power of two
    i64            gt_pool_n;
    int*           gt_over;          // overflow list of the first genotype pass
    const i64*     contig_len;       // reference lengths (TRA genotyping windows)
    const double*  sqrt_tab;         // pow(n, 0.5) by the host's libm, n < the longest segment + 2
    const double*  rcp_tab;          // 1.0 / n, correctly rounded (same range): exact division by small integers through div_by (below)
    const float*   cipk_tab;         // 1.96 / (n * pow(n, 0.5)) as float: the approximate cal_CIPOS of the register tier
    DevCounters*   cnt;
    ReadsState*    rs;
};

// ------------------------------------------------------------------------------------ small helpers
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ u64 lanemask_lt() { return (1ull << lane_id()) - 1ull; }

__device__ __forceinline__ i64 shfl_i64(i64 v, int src)
{
    int lo = __shfl((int)(v & 0xffffffffll), src), hi = __shfl((int)(v >> 32), src);
    return ((i64)hi << 32) | (unsigned)lo;
}
__device__ __forceinline__ i64 shfl_up_i64(i64 v, int d)
{
    int lo = __shfl_up((int)(v & 0xffffffffll), d), hi = __shfl_up((int)(v >> 32), d);
    return ((i64)hi << 32) | (unsigned)lo;
}
__device__ __forceinline__ i64 shfl_xor_i64(i64 v, int m)
{
    int lo = __shfl_xor((int)(v & 0xffffffffll), m), hi = __shfl_xor((int)(v >> 32), m);
    return ((i64)hi << 32) | (unsigned)lo;
}
__device__ __forceinline__ double shfl_xor_f64(double v, int m) { return __longlong_as_double(shfl_xor_i64(__double_as_longlong(v), m)); }
__device__ __forceinline__ i64 readlane_i64x(i64 v, int l)
{
    const int lo = __builtin_amdgcn_readlane((int)(v & 0xffffffffll), l);
    const int hi = __builtin_amdgcn_readlane((int)(v >> 32), l);
    return ((i64)hi << 32) | (unsigned)lo;
}
// a / b for a double a and a divisor b whose correctly rounded reciprocal y = RN(1 / b) is at hand (rcp_tab): q = a y is
// within an ulp or two of the quotient, the residual r = a - b q is exact in one FMA, and RN(q + r y) is the correctly rounded
// quotient - bit for bit what the float64 division of numpy / CPython returns (Markstein's theorem: it only fails for a
// divisor whose significand is all ones; checked against hardware division on 8.6e8 cases incl. quotients next to rounding
// midpoints).  Three instructions instead of the ~35 of v_div_scale / v_rcp / Newton / v_div_fmas / v_div_fixup.
__device__ __forceinline__ double div_by(double a, double b, double y)
{
    const double q = a * y;
    const double r = __builtin_fma(-b, q, a);
    return __builtin_fma(r, y, q);
}
// ---- wave64 scans / reductions on DPP (data-parallel primitives: row_shr inside rows of 16 lanes, then
// row_bcast:15 / row_bcast:31 across rows): 6 VALU moves per 32-bit word, no LDS round trips, no waitcnt.
template <int CTRL, int RM> __device__ __forceinline__ int dpp_i32(int old, int v)
{
    return __builtin_amdgcn_update_dpp(old, v, CTRL, RM, 0xf, false);       // lanes without a source keep `old`
}
template <int CTRL, int RM> __device__ __forceinline__ i64 dpp_i64(i64 old, i64 v)
{
    const int lo = dpp_i32<CTRL, RM>((int)(old & 0xffffffffll), (int)(v & 0xffffffffll));
    const int hi = dpp_i32<CTRL, RM>((int)(old >> 32), (int)(v >> 32));
    return ((i64)hi << 32) | (unsigned)lo;
}
// One step of a 64-bit DPP scan: v + (v moved by the pattern; lanes without a source and rows outside the mask add nothing),
// as the carry pair the hardware has for it - v_add_co_u32_dpp / v_addc_co_u32_dpp.  Written through dpp_i64 the compiler
// builds the moved value as two 64-bit numbers (lo | 0 and 0 | hi), each from a zeroed register and a v_mov_b32_dpp, and adds
// them with two v_lshl_add_u64: seven vector instructions a step where two do - and these kernels run at 60 - 86 % of the vector
// ALUs' issue rate (profiles/r04_cfg3_insts.txt), so the count is the time.  s_nop 4: a DPP read needs two wait states after
// the VALU write of its source and FIVE after a VALU write of EXEC (v_cmpx), and the hazard recogniser does not look inside an
// asm block: the wait covers the longer of the two whatever precedes the block.
template <int CTRL, int RM> __device__ __forceinline__ i64 dpp_add_i64(i64 v);
#define CSV_DPP_ADD64(CTRL, RM, TXT)                                                                                              \
    template <> __device__ __forceinline__ i64 dpp_add_i64<CTRL, RM>(i64 v)                                                       \
    {                                                                                                                             \
        unsigned lo = (unsigned)((u64)v & 0xffffffffull), hi = (unsigned)((u64)v >> 32);                                          \
        asm volatile("s_nop 4\n\tv_add_co_u32_dpp %0, vcc, %0, %0 " TXT "\n\tv_addc_co_u32_dpp %1, vcc, %1, %1, vcc " TXT         \
                     : "+v"(lo), "+v"(hi) : : "vcc");                                                                             \
        return (i64)(((u64)hi << 32) | lo);                                                                                       \
    }
CSV_DPP_ADD64(0x111, 0xf, "row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")
CSV_DPP_ADD64(0x112, 0xf, "row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1")
CSV_DPP_ADD64(0x114, 0xf, "row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1")
CSV_DPP_ADD64(0x118, 0xf, "row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1")
CSV_DPP_ADD64(0x142, 0xa, "row_bcast:15 row_mask:0xa bank_mask:0xf")
CSV_DPP_ADD64(0x143, 0xc, "row_bcast:31 row_mask:0xc bank_mask:0xf")
#undef CSV_DPP_ADD64
__device__ __forceinline__ i64 wave_incl_scan_i64(i64 v)
{
    v = dpp_add_i64<0x111, 0xf>(v);     // row_shr:1
    v = dpp_add_i64<0x112, 0xf>(v);     // row_shr:2
    v = dpp_add_i64<0x114, 0xf>(v);     // row_shr:4
    v = dpp_add_i64<0x118, 0xf>(v);     // row_shr:8
    v = dpp_add_i64<0x142, 0xa>(v);     // row_bcast:15 into rows 1 and 3
    v = dpp_add_i64<0x143, 0xc>(v);     // row_bcast:31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ int wave_incl_scan_i32(int v)
{
    v += dpp_i32<0x111, 0xf>(0, v);
    v += dpp_i32<0x112, 0xf>(0, v);
    v += dpp_i32<0x114, 0xf>(0, v);
    v += dpp_i32<0x118, 0xf>(0, v);
    v += dpp_i32<0x142, 0xa>(0, v);
    v += dpp_i32<0x143, 0xc>(0, v);
    return v;
}
__device__ __forceinline__ i64 wave_incl_max_i64(i64 v)
{
    i64 t;
    t = dpp_i64<0x111, 0xf>(INT64_MIN, v); v = t > v ? t : v;
    t = dpp_i64<0x112, 0xf>(INT64_MIN, v); v = t > v ? t : v;
    t = dpp_i64<0x114, 0xf>(INT64_MIN, v); v = t > v ? t : v;
    t = dpp_i64<0x118, 0xf>(INT64_MIN, v); v = t > v ? t : v;
    t = dpp_i64<0x142, 0xa>(INT64_MIN, v); v = t > v ? t : v;
    t = dpp_i64<0x143, 0xc>(INT64_MIN, v); v = t > v ? t : v;
    return v;
}
// totals come back through v_readlane of lane 63: wave-uniform (SGPR) results
__device__ __forceinline__ i64 wave_sum_i64(i64 v)
{
    v = wave_incl_scan_i64(v);
    const int lo = __builtin_amdgcn_readlane((int)(v & 0xffffffffll), 63), hi = __builtin_amdgcn_readlane((int)(v >> 32), 63);
    return ((i64)hi << 32) | (unsigned)lo;
}
__device__ __forceinline__ int wave_sum_i32(int v)
{
    return __builtin_amdgcn_readlane(wave_incl_scan_i32(v), 63);
}
__device__ __forceinline__ i64 lane63_i64(i64 v)
{
    const int lo = __builtin_amdgcn_readlane((int)(v & 0xffffffffll), 63), hi = __builtin_amdgcn_readlane((int)(v >> 32), 63);
    return ((i64)hi << 32) | (unsigned)lo;
}
// value of the lane to the left (lane 0 gets 0): DPP wave_shr:1
__device__ __forceinline__ i64 wave_shr1_i64(i64 v) { return dpp_i64<0x138, 0xf>(0, v); }

// segment of compact index w (woff is tiny; lives in L1/L2)
__device__ __forceinline__ int seg_of(const DevBatch& B, i64 w)
{
    int lo = 0, hi = B.n_seg;               // last k with woff[k] <= w (empty segments are skipped naturally)
    while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (B.woff[mid] <= w) lo = mid; else hi = mid; }
    return lo;
}

// ------------------------------------------------------------------------------------ chain
// flag[w] = 1 when w starts a new chained cluster: first of its segment, the type's break predicate
// against the previous signature, or the previous signature is a (0,0) look-alike of the reference's
// sentinel (see oracle csvo_cluster_batch).
constexpr int CH_ITEMS = 8;                         // rows of 64 per wavefront
constexpr int CH_TILE = 256 * CH_ITEMS;             // signatures per workgroup
constexpr int MID_CAP = 256;                        // largest cluster of the one-wavefront tiers (k_refine<64,256>); above: the workgroup tier
constexpr int CT_WORDS = CH_TILE / WAVE;            // flag words (64 signatures each) per tile
constexpr int TI_STRIDE = CH_TILE + 8;              // work-item slots per tile (at most one per signature)

// What the chain kernels need to know about a tile's segment(s).  The host knows the segment table when it uploads a
// batch, so it leaves one 128-byte record per tile (TILE_REC int4 words) that holds the chain / gate scalars of up to
// three NON-EMPTY segments overlapping the tile INLINE.  With it a tile's flags need no dependent look-up (segment search
// -> segment record -> rows) at all: record and rows are loaded together, ONE round trip.  (A kernel lasts as long as its
// slowest wavefront: in the first forms of this kernel the ~50 spans of a 30x genome that cross a segment boundary walked a
// chain of 4-5 dependent loads, and that chain - not the 5 400 fast wavefronts - was 6.5 us of the kernel's 12.)
//   [0]          {first segment k0, last segment k1 (overlapping the tile), inline count nin (0: more than three, or a bias that
//                does not fit 32 bits: the general form reads the segment table), -}
//   [1 + i]      segment i: {first w, read_count, segment index | svtype << 24 | dropped << 28, max_cluster_bias}
constexpr int TILE_REC = 4;
struct TileSegI { int sf, type, drop, rc, k, bias32; i64 bias; };
__device__ __forceinline__ TileSegI tile_seg(const int4 a)
{
    TileSegI g;
    g.sf = a.x; g.rc = a.y; g.k = a.z & 0xffffff; g.type = (a.z >> 24) & 7; g.drop = (a.z >> 28) & 1;
    g.bias32 = a.w; g.bias = (i64)a.w;
    return g;
}
__device__ __forceinline__ bool pair_type(int t) { return t == CSV_INV || t == CSV_TRA; }

// The general form of the break flags, one signature at a time, every lane on its own: any type, any number of segment
// boundaries, the batch's first and last spans.  What a lane needs of its segment: the chain scalars and where the segment
// and its successor begin.  All loads of a lane are independent of each other except segment search -> segment scalars, so a
// span that crosses a boundary costs a few dependent round trips, not a few per row (the first form of this path walked the
// rows one after the other and, with 47 boundary spans in a 30x genome, its latency chain WAS the kernel's duration).
struct SegScal { i64 bias, sf, next; int type; };
__device__ __forceinline__ void seg_scal_load(const DevBatch& B, int k, SegScal& S)
{
    S.sf = B.woff[k]; S.next = B.woff[k + 1];
    S.bias = B.seg[k].max_cluster_bias; S.type = B.seg[k].svtype;
}
// last segment of [k0, k1] that begins at or before w (binary search; empty segments are skipped naturally)
__device__ __forceinline__ int seg_in_tile(const DevBatch& B, i64 w, int k0, int k1)
{
    int lo = k0, hi = k1 + 1;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (B.woff[mid] <= w) lo = mid; else hi = mid; }
    return lo;
}
// break flag f and (0,0) mark z of signature w > 0's pair (left neighbour values *0, own values *1)
__device__ __forceinline__ void sig_flag(i64 w, const SegScal& S, i64 a1, i64 a0, i64 b1, i64 b0, int x1, int x0, bool& f, bool& z)
{
    z = w > 0 && a0 == 0 && b0 == 0;
    f = w == S.sf || a1 - a0 > S.bias || z;
    if (!f) {
        if (S.type == CSV_INV) f = (b1 - b0 > S.bias) || (x1 != x0);
        else if (S.type == CSV_TRA) f = x1 != x0;
    }
}
// the 8 signatures [w0, w0 + 8) of one lane (those below W): flag byte and mark byte
template <bool NARROW> __device__ __forceinline__ void chain_bytes_general(const DevBatch& B, i64 w0, int k0, int k1, unsigned& byte_out, unsigned& zbyte_out)
{
    typedef typename std::conditional<NARROW, int, i64>::type raw_t;      // (values stay in the column's width: 27 instead of 45 registers)
    unsigned byte = 0, zb = 0;
    if (w0 < B.W) {
        raw_t a[9], b[9]; int x[9];
        const i64 wl = w0 > 0 ? w0 - 1 : 0;
        if constexpr (NARROW) { a[0] = B.a.p32[wl]; b[0] = B.b.p32[wl]; } else { a[0] = B.a.p64[wl]; b[0] = B.b.p64[wl]; }
        x[0] = B.aux[wl];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const i64 w = w0 + j < B.W ? w0 + j : B.W - 1;
            if constexpr (NARROW) { a[j + 1] = B.a.p32[w]; b[j + 1] = B.b.p32[w]; } else { a[j + 1] = B.a.p64[w]; b[j + 1] = B.b.p64[w]; }
            x[j + 1] = B.aux[w];
        }
        int k = seg_in_tile(B, w0, k0, k1);
        SegScal S;
        seg_scal_load(B, k, S);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const i64 w = w0 + j;
            if (w < B.W) {
                while (k < k1 && w >= S.next) { k++; seg_scal_load(B, k, S); }
                bool f, z;
                sig_flag(w, S, (i64)a[j + 1], (i64)a[j], (i64)b[j + 1], (i64)b[j], x[j + 1], x[j], f, z);
                byte |= (unsigned)f << j; zb |= (unsigned)(f && z) << j;
            }
        }
    }
    byte_out = byte; zbyte_out = zb;
}
constexpr int EM_TILE = 8;                          // items per emit wavefront
constexpr int EM_SUPER = 512;                       // items per second-level sum

// sum of p[0 .. n) over the workgroup (256 threads); every thread gets the result
__device__ __forceinline__ i64 block_prefix_of(const int* p, int n, i64* sh)
{
    i64 v = 0;
    for (int i = threadIdx.x; i < n; i += 256) v += p[i];
    v = wave_sum_i64(v);
    if (lane_id() == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    const i64 t = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    return t;
}
__device__ __forceinline__ i64 block_prefix_of64(const i64* p, int n, i64* sh)
{
    i64 v = 0;
    for (int i = threadIdx.x; i < n; i += 256) v += p[i];
    v = wave_sum_i64(v);
    if (lane_id() == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    const i64 t = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    return t;
}

// ------------------------------------------------------------------------------------ chain + work list in two kernels
// k_chain_count: one workgroup per tile of 2048 signatures.  A chained cluster belongs to the tile it STARTS in.
//   1. rows -> break flags, kept as bit masks only: 32 words of 64 flags in LDS (+ one word for the 64 signatures after the
//      tile, so that the tile's last cluster finds its end; the end of the batch counts as one more start).  No list of
//      cluster starts is ever written: half of all signatures of a 30x genome start a cluster (singletons), and the first
//      version of this kernel spent most of its ~146 instructions per row of 64 on listing them and on a gate per start.
//   2. every thread owns 8 flags: the next start after its byte (a find-first-set over the rest of its word, the following
//      words only when that is empty), then its own starts from the top down: size = next start - this start.  The size gate
//      (a cluster is a work item when it has >= read_count signatures, does not end in a (0,0) element and its segment is not
//      dropped; INDEL:62-64) is a compare per start; the few per cent that pass are counted.
//   3. one packed wave scan (starts | work items << 16) + four wave totals order the tile's work items; their 16-byte
//      records {first w, size, segment | svtype << 24 | tier << 28, cluster index inside the tile} go to the tile's region.
// k_chain_apply turns the per-tile counts into global bases and compacts the records into the ordered work list and the
// tier lists.  (Fusing the two through a look-back costs an agent-scope release + acquire per tile, ~3.5 us against the
// ~1.7 us of a kernel boundary - MI355X_MICROARCH.md price list - and was measured slower in round 1.)

// gate of the cluster of m signatures in a segment with gate scalars g = {read_count, dropped, svtype}: bit 0 work item,
// bit 1 workgroup tier, bit 2 tiny, bit 3 wide (a DEL/INS cluster of 33 .. 64 signatures)
__device__ __forceinline__ int close_gate(int tiny_max, const int4 g, int m, int endz)
{
    if (endz || m < g.x || g.y) return 0;
    return 1 | ((m > 64) ? 2 : 0) | ((m <= tiny_max && g.z <= CSV_INS) ? 4 : 0) | ((m > 32 && m <= 64 && m > tiny_max && g.z <= CSV_INS) ? 8 : 0);
}

// starts among the 64 signatures [cb, cb + 64), cb <= W (bit i = signature cb + i), the end of the batch counting as one
// more start (bit W - cb when it falls inside the row); *zm = which of them follow a (0,0) element.  Used to look ahead
// from a tile; one signature per lane, the general form.
__device__ __forceinline__ u64 chain_flag_row64(const DevBatch& B, i64 cb, u64* zm)
{
    const i64 w = cb + lane_id();
    u64 f_end = 0, z_end = 0;                          // the batch's end
    if (B.W - cb < 64) {
        f_end = 1ull << (B.W - cb);
        if (B.end_z) z_end = f_end;
    }
    if (cb >= B.W) { *zm = z_end; return f_end; }
    bool f = false, z = false;
    if (w < B.W) {
        // (the 64 signatures lie in one chain tile - tiles are multiples of 64 - whose segment range bounds the search)
        const int4 tr = B.tile_info[TILE_REC * (int)(cb / CH_TILE)];
        const i64 wl = w > 0 ? w - 1 : 0;
        const i64 a1 = B.a[w], a0 = B.a[wl], b1 = B.b[w], b0 = B.b[wl];
        const int x1 = B.aux[w], x0 = B.aux[wl];
        SegScal S;
        seg_scal_load(B, seg_in_tile(B, w, tr.x, tr.y), S);
        sig_flag(w, S, a1, a0, b1, b0, x1, x0, f, z);
    }
    *zm = __ballot(f && z) | z_end;
    return __ballot(f) | f_end;
}

// The fast form of a wavefront's 512 flags: the tile's segments (at most three, none of them INV / TRA) are inline in its
// record.  Lane l owns the 8 consecutive signatures [base + 8 l, base + 8 l + 8): 32 (int32 columns) or 64 contiguous bytes per
// lane, loaded 16 bytes at a time; seven of its eight neighbour differences are in-lane, the eighth comes from the lane to the
// left with one DPP move, and the lane's 8 flags are a byte in a register - no ballots, no masks in SGPRs.
// The loads depend on nothing but the block index, so they are issued BEFORE the tile record is known (SpanRows; the position
// column is padded by a tile of positive values, so any span that begins inside the batch can be read), together with the
// tile record and the row after the tile: the kernel's common path is ONE memory round trip.
template <bool NARROW> struct SpanRows;
template <> struct SpanRows<true>  { int4 v0, v1; int left; };
template <> struct SpanRows<false> { longlong2 v0, v1, v2, v3; i64 left; };
template <bool NARROW> __device__ __forceinline__ void span_rows_load(const DevBatch& B, i64 base, SpanRows<NARROW>& R)
{
    const i64 lw = base > 0 ? base - 1 : 0;                     // (every lane, one address: no divergent block around a load)
    if constexpr (NARROW) {
        const int4* p = (const int4*)(B.a.p32 + base) + 2 * lane_id();
        R.v0 = p[0]; R.v1 = p[1];
        R.left = B.a.p32[lw];
    } else {
        const longlong2* p = (const longlong2*)(B.a.p64 + base) + 4 * lane_id();
        R.v0 = p[0]; R.v1 = p[1]; R.v2 = p[2]; R.v3 = p[3];
        R.left = B.a.p64[lw];
    }
}
// Returns false (wave-uniform) when a value of the span is 0 or negative: position 0 may be a (0,0) look-alike of the
// reference's sentinel and negative values would overflow the 32-bit differences; the caller then takes the general form.
// s0 .. s2: the tile's inline segments (nin of them, wave-uniform).
template <bool NARROW> __device__ __forceinline__ bool chain_bytes_fast(const SpanRows<NARROW>& R, i64 base, int nin, const TileSegI& s0, const TileSegI& s1,
                                                                        const TileSegI& s2, unsigned& byte_out)
{
    const int lane = lane_id();
    const int w0 = (int)base + 8 * lane;
    unsigned byte = 0;
    if constexpr (NARROW) {
        const int4 v0 = R.v0, v1 = R.v1;
        int prev = dpp_i32<0x138, 0xf>(0, v1.w);               // wave_shr:1
        if (lane == 0) prev = R.left;
        // all values (and the neighbour) positive?  unsigned minimum / maximum of the nine
        const unsigned mn = min(min(min((unsigned)v0.x, (unsigned)v0.y), min((unsigned)v0.z, (unsigned)v0.w)),
                                min(min((unsigned)v1.x, (unsigned)v1.y), min(min((unsigned)v1.z, (unsigned)v1.w), (unsigned)prev)));
        const unsigned mx = max(max(max((unsigned)v0.x, (unsigned)v0.y), max((unsigned)v0.z, (unsigned)v0.w)),
                                max(max((unsigned)v1.x, (unsigned)v1.y), max(max((unsigned)v1.z, (unsigned)v1.w), (unsigned)prev)));
        if (CSV_UNLIKELY(__ballot(mn == 0 || mx > 0x7fffffffu) != 0)) return false;
        // differences of non-negative 31-bit values cannot overflow; a bias beyond 31 bits can never be exceeded (bias32)
        const int d0 = v0.x - prev, d1 = v0.y - v0.x, d2 = v0.z - v0.y, d3 = v0.w - v0.z, d4 = v1.x - v0.w, d5 = v1.y - v1.x, d6 = v1.z - v1.y, d7 = v1.w - v1.z;
        if (CSV_LIKELY(nin == 1)) {
            const int bias = s0.bias32;
            byte = (unsigned)(d0 > bias) | ((unsigned)(d1 > bias) << 1) | ((unsigned)(d2 > bias) << 2) | ((unsigned)(d3 > bias) << 3) |
                   ((unsigned)(d4 > bias) << 4) | ((unsigned)(d5 > bias) << 5) | ((unsigned)(d6 > bias) << 6) | ((unsigned)(d7 > bias) << 7);
        } else {
            // a tile on a segment boundary: the bias of the signature's own segment
            const int sf1 = s1.sf, sf2 = nin > 2 ? s2.sf : 0x7fffffff, b0 = s0.bias32, b1 = s1.bias32, b2 = s2.bias32;
            const int d[8] = {d0, d1, d2, d3, d4, d5, d6, d7};
#pragma unroll
            for (int j = 0; j < 8; j++) { const int w = w0 + j, bias = w >= sf2 ? b2 : (w >= sf1 ? b1 : b0); byte |= (unsigned)(d[j] > bias) << j; }
        }
    } else {
        const longlong2 v0 = R.v0, v1 = R.v1, v2 = R.v2, v3 = R.v3;
        i64 prev = wave_shr1_i64(v3.y);
        if (lane == 0) prev = R.left;
        const bool zero = v0.x == 0 || v0.y == 0 || v1.x == 0 || v1.y == 0 || v2.x == 0 || v2.y == 0 || v3.x == 0 || v3.y == 0 || prev == 0;
        if (CSV_UNLIKELY(__ballot(zero) != 0)) return false;
        const i64 d[8] = {v0.x - prev, v0.y - v0.x, v1.x - v0.y, v1.y - v1.x, v2.x - v1.y, v2.y - v2.x, v3.x - v2.y, v3.y - v3.x};
        const int sf1 = nin > 1 ? s1.sf : 0x7fffffff, sf2 = nin > 2 ? s2.sf : 0x7fffffff;
#pragma unroll
        for (int j = 0; j < 8; j++) { const int w = w0 + j; const i64 bias = w >= sf2 ? s2.bias : (w >= sf1 ? s1.bias : s0.bias); byte |= (unsigned)(d[j] > bias) << j; }
    }
    // the first signature of a segment is a start
    { const int rel = s0.sf - w0; if (rel >= 0 && rel < 8) byte |= 1u << rel; }
    if (nin > 1) { const int rel = s1.sf - w0; if (rel >= 0 && rel < 8) byte |= 1u << rel; }
    if (nin > 2) { const int rel = s2.sf - w0; if (rel >= 0 && rel < 8) byte |= 1u << rel; }
    byte_out = byte;
    return true;
}

// first start after bit i of the 64-flag window `win` that begins at tile-relative position p0 (word wi, bit offset sh8); the
// look-ahead word and the far end included
__device__ __forceinline__ int chain_next_start(const u64* s_F, u64 win, u64 w1, int wi, int sh8, int p0, int i, int far)
{
    const u64 above = (i < 63) ? (win & ~((2ull << i) - 1ull)) : 0ull;
    if (above) return p0 + __ffsll((long long)above) - 1;
    const u64 rest = sh8 ? (w1 >> sh8) : w1;                // what the window did not cover of word wi + 1
    if (rest) return p0 + 64 + __ffsll((long long)rest) - 1;
    for (int wj = wi + 2; wj <= CT_WORDS; wj++) { const u64 f2 = s_F[wj]; if (f2) return wj * 64 + __ffsll((long long)f2) - 1; }
    return far;
}
// gate scalars {read_count, dropped, svtype} and segment of the cluster that starts at signature w
__device__ __forceinline__ int4 chain_gate_of(const i64* woff, const int4* seg_gate, int w, int nin, int k0, int k1, const TileSegI& g0, const TileSegI& g1,
                                              const TileSegI& g2, int& kseg)
{
    if (CSV_LIKELY(nin >= 1)) {
        const bool i2 = nin > 2 && w >= g2.sf, i1 = nin > 1 && w >= g1.sf;
        kseg = i2 ? g2.k : (i1 ? g1.k : g0.k);
        return make_int4(i2 ? g2.rc : (i1 ? g1.rc : g0.rc), i2 ? g2.drop : (i1 ? g1.drop : g0.drop), i2 ? g2.type : (i1 ? g1.type : g0.type), 0);
    }
    int lo = k0, hi = k1 + 1;                               // more than three segments in the tile: the segment table
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (woff[mid] <= w) lo = mid; else hi = mid; }
    kseg = lo;
    return seg_gate[lo];
}

template <bool NARROW> __global__ __launch_bounds__(256, NARROW ? 6 : 5) void k_chain_count(DevBatch B)
{
    // flags / (0,0)-predecessor marks of the tile, one bit per signature (word w = signatures 64 w .. 64 w + 63, also addressed
    // as bytes of 8 signatures); [32] = the row after the tile
    __shared__ __attribute__((aligned(16))) u64 s_F[CT_WORDS + 2], s_Z[CT_WORDS + 2];
    __shared__ int s_far[2], s_w[4], s_t[4], s_rc[4];
    const int wv = threadIdx.x >> 6, lane = lane_id();
    const i64 tile0 = (i64)blockIdx.x * CH_TILE;
    const int nvalid = (int)(B.W - tile0 < CH_TILE ? B.W - tile0 : CH_TILE);     // signatures of this tile
    // ---- everything the common path reads, issued at once: the tile's record, this wavefront's rows, and (wavefront 0) the
    // next tile's record and the row after the tile.  (Raw values: nothing is converted or compared before every load of
    // this block has been issued.)
    const i64 base = tile0 + wv * (WAVE * CH_ITEMS);
    const bool in_batch = base < B.W;                                      // (wave-uniform, known without any load)
    const bool halo_in = wv == 0 && tile0 + CH_TILE < B.W;                 // a row after the tile exists (padded: always readable)
    typedef typename std::conditional<NARROW, int, i64>::type raw_t;
    const int4* trp = B.tile_info + (i64)TILE_REC * blockIdx.x;
    const int4 t0 = trp[0], t1 = trp[1], t2 = trp[2], t3 = trp[3];
    SpanRows<NARROW> R;
    if (in_batch) span_rows_load<NARROW>(B, base, R);
    raw_t h_raw = 0, h_lraw = 0;
    int4 n0 = make_int4(0, 0, 0, 0), n1 = n0, n2 = n0, n3 = n0;
    if (halo_in) {
        if constexpr (NARROW) { h_raw = B.a.p32[tile0 + CH_TILE + lane]; h_lraw = B.a.p32[tile0 + CH_TILE - 1]; }
        else { h_raw = B.a.p64[tile0 + CH_TILE + lane]; h_lraw = B.a.p64[tile0 + CH_TILE - 1]; }
        n0 = trp[TILE_REC]; n1 = trp[TILE_REC + 1]; n2 = trp[TILE_REC + 2]; n3 = trp[TILE_REC + 3];
    }
    const int k0 = __builtin_amdgcn_readfirstlane(t0.x), k1 = __builtin_amdgcn_readfirstlane(t0.y), nin = __builtin_amdgcn_readfirstlane(t0.z);
    const TileSegI g0 = tile_seg(t1), g1 = tile_seg(t2), g2 = tile_seg(t3);
    const bool pairs = pair_type(g0.type) || (nin > 1 && pair_type(g1.type)) || (nin > 2 && pair_type(g2.type));     // (uniform values in VGPRs)
    const bool fastable = nin >= 1 && !__ballot(pairs);
    if (CSV_ABL(8)) {                                                                   // loads only (every loaded value consumed)
        i64 acc = (i64)h_raw + g0.rc;
        if (in_batch) { if constexpr (NARROW) acc += R.left + R.v0.x + R.v0.y + R.v0.z + R.v0.w + R.v1.x + R.v1.y + R.v1.z + R.v1.w; else acc += R.left + R.v0.x + R.v0.y + R.v1.x + R.v1.y + R.v2.x + R.v2.y + R.v3.x + R.v3.y; }
        if (acc == 12345) B.partial[blockIdx.x] = 1;
        return;
    }
    unsigned bits = 0;                                      // this thread's 8 flags (signatures tile0 + 8 t ...)
    const int p0 = 8 * (int)threadIdx.x;
    {
        bool have_bits = false;
        if (CSV_LIKELY(in_batch && fastable)) have_bits = chain_bytes_fast<NARROW>(R, base, nin, g0, g1, g2, bits);
        unsigned zb = 0;
        if (CSV_UNLIKELY(!have_bits) && !CSV_ABL(14)) chain_bytes_general<NARROW>(B, base + 8 * lane, k0, k1, bits, zb);
        // real signatures only (the padding behind the batch is not data) ...
        bits = (p0 + 8 <= nvalid) ? bits : (p0 >= nvalid ? 0u : (bits & ((1u << (nvalid - p0)) - 1u)));
        // ... and, in the tile that ends the batch, the end of the batch closes the last cluster: a virtual start at position
        // nvalid (position 2048 lives in the look-ahead word)
        if (nvalid < CH_TILE && p0 <= nvalid && nvalid < p0 + 8) {
            bits |= 1u << (nvalid - p0);
            if (B.end_z) zb |= 1u << (nvalid - p0);
        }
        ((unsigned char*)s_F)[threadIdx.x] = (unsigned char)bits;
        ((unsigned char*)s_Z)[threadIdx.x] = (unsigned char)zb;
    }
    if (wv == 0 && !CSV_ABL(12)) {
        // the row after the tile (the tile's last cluster ends at the first start in it); should it hold none - a cluster of
        // more than 64 signatures crossing the tile's end - walk on, 64 signatures at a time
        u64 zm = 0, fm = 0;
        bool done = nvalid < CH_TILE;                       // (the batch ends inside the tile: the virtual start is in its own words)
        if (CSV_LIKELY(!done && halo_in)) {
            // common case: the row's segments are inline in the next tile's record, none INV / TRA, all positions positive
            const int nn = __builtin_amdgcn_readfirstlane(n0.z);
            const TileSegI h0 = tile_seg(n1), h1 = tile_seg(n2), h2 = tile_seg(n3);
            const bool hp = pair_type(h0.type) || (nn > 1 && pair_type(h1.type)) || (nn > 2 && pair_type(h2.type));
            const i64 h_a = (i64)h_raw;
            i64 a0 = wave_shr1_i64(h_a);
            if (lane == 0) a0 = (i64)h_lraw;
            const int w = (int)tile0 + CH_TILE + lane;
            const bool in = w < (int)B.W;
            if (CSV_LIKELY(nn >= 1 && !__ballot(hp || (in && (a0 <= 0 || h_a <= 0))))) {
                const i64 bias = (nn > 2 && w >= h2.sf) ? h2.bias : ((nn > 1 && w >= h1.sf) ? h1.bias : h0.bias);
                fm = __ballot(in && (w == h0.sf || (nn > 1 && w == h1.sf) || (nn > 2 && w == h2.sf) || h_a - a0 > bias));
                if (B.W - (tile0 + CH_TILE) < 64) {         // the batch ends inside the row
                    fm |= 1ull << (B.W - (tile0 + CH_TILE));
                    if (B.end_z) zm = 1ull << (B.W - (tile0 + CH_TILE));
                }
                done = true;
            }
        }
        if (CSV_UNLIKELY(!done)) fm = chain_flag_row64(B, tile0 + CH_TILE, &zm);
        int far = -1, farz = 0;
        if (CSV_UNLIKELY(nvalid == CH_TILE && fm == 0)) {
            for (i64 cb = tile0 + CH_TILE + 64;; cb += 64) {
                u64 z2;
                const u64 f2 = chain_flag_row64(B, cb, &z2);
                if (f2) { const int l = __ffsll((long long)f2) - 1; far = (int)(cb - tile0) + l; farz = (int)((z2 >> l) & 1); break; }
            }
        }
        if (lane == 0) { s_F[CT_WORDS] = fm; s_Z[CT_WORDS] = zm; s_F[CT_WORDS + 1] = 0; s_far[0] = far; s_far[1] = farz; }
    }
    if (CSV_ABL(11)) { if (bits == 0x1234567u) B.partial[blockIdx.x] = 1; return; }                     // + LDS writes and the look-ahead row
    // the smallest read_count among the tile's segments bounds the candidate filter below
    int rcmin = g0.rc;
    if (nin > 1) rcmin = min(rcmin, g1.rc);
    if (nin > 2) rcmin = min(rcmin, g2.rc);
    rcmin = __builtin_amdgcn_readfirstlane(rcmin);
    if (CSV_UNLIKELY(nin == 0)) {                           // more than three segments in the tile: from the segment table
        int v = 0x7fffffff;
        for (int k = k0 + (int)threadIdx.x; k <= k1; k += 256) { const int x = B.seg_gate[k].x; v = x < v ? x : v; }
        for (int d = 32; d > 0; d >>= 1) { const int o = __shfl_xor(v, d); v = o < v ? o : v; }
        if (lane == 0) s_rc[wv] = v;
    }
    __syncthreads();
    if (CSV_UNLIKELY(nin == 0)) rcmin = min(min(s_rc[0], s_rc[1]), min(s_rc[2], s_rc[3]));
    if (B.tile_lead && wv == 1) {                           // (gate-first calls only) where the tile's first cluster starts
        const u64 nzw = __ballot(lane < CT_WORDS && s_F[lane < CT_WORDS ? lane : 0] != 0);
        int lead = CH_TILE;
        if (nzw) { const int fw = __ffsll((long long)nzw) - 1; lead = fw * 64 + __ffsll((long long)s_F[fw]) - 1; }
        if (lane == 0) B.tile_lead[blockIdx.x] = lead < nvalid ? lead : CH_TILE;
    }
    if (CSV_ABL(9)) { if (threadIdx.x == 0) B.partial[blockIdx.x] = (int)s_F[3]; return; }       // flags + barrier only
    // ---- every thread: the 8 flags [8 t, 8 t + 8) of the tile and the 64-flag window that begins with them
    const int t = threadIdx.x, wi = t >> 3, sh8 = (t & 7) * 8;
    const u64 w0 = s_F[wi], w1 = s_F[wi + 1];
    if (B.per_sig && t < CT_WORDS) B.ch_masks[(i64)blockIdx.x * CT_WORDS + t] = s_F[t];      // the flags, for k_chain_ids
    const u64 win = sh8 ? ((w0 >> sh8) | (w1 << (64 - sh8))) : w0;
    // own starts: real signatures only (the virtual start at nvalid only ENDS a cluster)
    const unsigned own = (p0 + 8 <= nvalid) ? bits : (p0 >= nvalid ? 0u : (bits & ((1u << (nvalid - p0)) - 1u)));
    // candidates: a start followed by at least read_count - 1 signatures that do not start a cluster (half of all signatures
    // of a 30x genome start one, a few per cent of those pass): runs of zeros by shift-and-AND over the window
    unsigned cand = own;
    {
        int kk = rcmin - 1;
        if (kk > 56) kk = 56;                               // (the window holds 56 flags behind a byte's last one: a longer demand is verified below)
        if (kk > 0) {
            u64 r = ~win;
            for (int have = 1; have < kk;) { const int s = have < kk - have ? have : kk - have; r &= r >> s; have += s; }
            cand &= (unsigned)(r >> 1);
        }
    }
    const int far_pos = s_far[0];
    // pass 1: the exact gate for the candidates (sizes are recomputed in pass 2 for those that pass: no per-bit arrays)
    unsigned passbits = 0;
    int n_big = 0, n_tiny = 0, n_wide = 0;
    if (!CSV_ABL(7) && __ballot(cand != 0)) {
        for (unsigned b = cand; b;) {
            const int lb = __ffs((int)b) - 1;
            b &= b - 1;
            const int pos = p0 + lb, nxt = chain_next_start(s_F, win, w1, wi, sh8, p0, lb, far_pos), m = nxt - pos;
            // (0,0) look-alike before the cluster's end: marks are kept for the tile and the row after it; a far end carries its own
            const int endz = nxt <= CT_WORDS * 64 + 63 ? (int)((s_Z[nxt >> 6] >> (nxt & 63)) & 1) : s_far[1];
            int kseg;
            const int fl = close_gate(B.tiny_max, chain_gate_of(B.woff, B.seg_gate, (int)tile0 + pos, nin, k0, k1, g0, g1, g2, kseg), m, endz);
            if (fl & 1) { passbits |= 1u << lb; n_big += (fl >> 1) & 1; n_tiny += (fl >> 2) & 1; n_wide += (fl >> 3) & 1; }
        }
    }
    // ---- order: starts | work items << 16, one wave scan + the wave totals
    const int v = __popc(own) | (__popc(passbits) << 16);
    const int inc = wave_incl_scan_i32(v);
    const int tiers = wave_sum_i32(n_tiny | (n_wide << 12) | (n_big << 20));
    if (lane == 63) { s_w[wv] = inc; s_t[wv] = tiers; }
    __syncthreads();
    int before = inc - v;
    for (int q = 0; q < wv; q++) before += s_w[q];
    if (passbits) {                                      // pass 2: the records of the starts that passed, in order
        const int c_before = (before & 0xffff), j_before = before >> 16;
        for (unsigned b = passbits; b;) {
            const int lb = __ffs((int)b) - 1;
            b &= b - 1;
            const int pos = p0 + lb, m = chain_next_start(s_F, win, w1, wi, sh8, p0, lb, far_pos) - pos;
            int kseg;
            const int4 g = chain_gate_of(B.woff, B.seg_gate, (int)tile0 + pos, nin, k0, k1, g0, g1, g2, kseg);
            const int fl = close_gate(B.tiny_max, g, m, 0);
            const int ci = c_before + __popc(own & ((1u << lb) - 1u)), ji = j_before + __popc(passbits & ((1u << lb) - 1u));
            B.tile_items[(i64)blockIdx.x * TI_STRIDE + ji] = make_int4((int)tile0 + pos, m, kseg | (g.z << 24) | ((fl >> 1) << 28), ci);
        }
    }
    // first kernel of a run: nothing in this kernel reads the counters, every later kernel is stream-ordered behind it.  (At the
    // END: a store in front of the loads made every load of the kernel "possibly clobbered", so the tile records - one address
    // per workgroup - came through vector loads into 32 VGPRs instead of scalar loads into SGPRs.)
    if (blockIdx.x == 0 && threadIdx.x < (int)(sizeof(DevCounters) / 4)) ((int*)B.cnt)[threadIdx.x] = 0;
    if (threadIdx.x == 0) {
        const int tot = s_w[0] + s_w[1] + s_w[2] + s_w[3], tt = s_t[0] + s_t[1] + s_t[2] + s_t[3];
        B.partial[blockIdx.x] = tot & 0xffff;                 // (k_chain_ids reads the starts alone)
        B.tile_cnt[blockIdx.x] = make_int4(tot & 0xffff, tot >> 16, tt >> 20, (tt & 0xfff) | (((tt >> 12) & 0xff) << 16));   // starts, items, above 64, tiny | wide << 16
    }
}

// Compaction of the tiles' item records into the ordered work list and the three tier lists.  One wavefront per chain
// tile, four tiles per workgroup (the prefix over the earlier tiles is recomputed from the per-tile counts: a few thousand
// L2-resident values).  Reads ~16 bytes per work item; no signature column, no flag, no gate.
__global__ __launch_bounds__(256) void k_chain_apply(DevBatch B)
{
    __shared__ int sh[4][5];
    const int wv = threadIdx.x >> 6;
    const int tile_b = blockIdx.x * 4, tile = tile_b + wv, ntile = (int)((B.W + CH_TILE - 1) / CH_TILE);
    // ONE round trip: the counts of the earlier tiles (one 16-byte record per tile, eight per thread issued together), the
    // workgroup's own four, and - before its count is known - the tile's first 64 item records
    int4 c8[8];
#pragma unroll
    for (int u = 0; u < 8; u++) { const int i = threadIdx.x + 256 * u; c8[u] = i < tile_b ? B.tile_cnt[i] : make_int4(0, 0, 0, 0); }
    int4 w4[4];
#pragma unroll
    for (int q = 0; q < 4; q++) w4[q] = tile_b + q < ntile ? B.tile_cnt[tile_b + q] : make_int4(0, 0, 0, 0);
    const int4 rec0 = B.tile_items[(i64)(tile < ntile ? tile : 0) * TI_STRIDE + lane_id()];
    int p0 = 0, p1 = 0, p2 = 0, p3 = 0, p4 = 0;              // cluster starts, items, items above 64, tiny, wide
#pragma unroll
    for (int u = 0; u < 8; u++) { p0 += c8[u].x; p1 += c8[u].y; p2 += c8[u].z; p3 += c8[u].w & 0xffff; p4 += c8[u].w >> 16; }
    for (int i = threadIdx.x + 2048; i < tile_b; i += 256) { const int4 c = B.tile_cnt[i]; p0 += c.x; p1 += c.y; p2 += c.z; p3 += c.w & 0xffff; p4 += c.w >> 16; }
    p0 = wave_sum_i32(p0); p1 = wave_sum_i32(p1); p2 = wave_sum_i32(p2); p3 = wave_sum_i32(p3); p4 = wave_sum_i32(p4);
    if (lane_id() == 0) { sh[wv][0] = p0; sh[wv][1] = p1; sh[wv][2] = p2; sh[wv][3] = p3; sh[wv][4] = p4; }
    int b0 = 0, b1 = 0, b2 = 0, b3 = 0, b4 = 0;               // ... of the workgroup's earlier tiles
#pragma unroll
    for (int q = 0; q < 4; q++) if (q < wv) { b0 += w4[q].x; b1 += w4[q].y; b2 += w4[q].z; b3 += w4[q].w & 0xffff; b4 += w4[q].w >> 16; }
    int4 own = make_int4(0, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 4; q++) if (q == wv) own = w4[q];
    __syncthreads();
    if (tile >= ntile) return;
    const int run = sh[0][0] + sh[1][0] + sh[2][0] + sh[3][0] + b0;          // id of the first cluster that STARTS in this tile
    const int bj = sh[0][1] + sh[1][1] + sh[2][1] + sh[3][1] + b1;
    int bb = sh[0][2] + sh[1][2] + sh[2][2] + sh[3][2] + b2, bt = sh[0][3] + sh[1][3] + sh[2][3] + sh[3][3] + b3, bw = sh[0][4] + sh[1][4] + sh[2][4] + sh[3][4] + b4;
    const i64 own0 = own.x;
    const int n_it = __builtin_amdgcn_readfirstlane(own.y);
    for (int base = 0; base < n_it; base += 64) {
        const int idx = base + lane_id();
        const bool act = idx < n_it;
        int4 rec = make_int4(0, 0, 0, 0);
        if (act) rec = base == 0 ? rec0 : B.tile_items[(i64)tile * TI_STRIDE + idx];
        const int tier = (rec.z >> 28) & 3;
        const bool wide = act && ((rec.z >> 30) & 1);
        const u64 m_big = __ballot(act && (tier & 1)), m_tiny = __ballot(act && (tier & 2)), m_wide = __ballot(wide);
        if (act) {
            const int j = bj + idx, jb = bb + __popcll(m_big & lanemask_lt()), jt = bt + __popcll(m_tiny & lanemask_lt());
            B.item_rec[j] = make_int4(run + rec.w, rec.z & 0xffffff, rec.x, rec.y);
            const int4 ent = make_int4(j, rec.z & 0x0fffffff, rec.x, rec.y);
            if (tier & 1) B.list_big[jb] = j;
            else if (tier & 2) B.list_tiny[jt] = ent;
            else B.list_small[j - jb - jt] = ent;
            // DEL/INS items of 33 .. 64 signatures are ALSO listed on their own: k_refine_indel_wave gives each a wavefront
            // instead of running them after the pair they sit in
            if (wide) B.list_wide[bw + __popcll(m_wide & lanemask_lt())] = ent;
        }
        bb += __popcll(m_big); bt += __popcll(m_tiny); bw += __popcll(m_wide);
        // (rare: the workgroup tier looks at this count before it walks the list of the items above 64 - on a 90x genome that
        // list has 15 k entries, none of them its own, and the walk alone was 10 us)
        const u64 m_huge = __ballot(act && (tier & 1) && rec.y > MID_CAP);
        if (m_huge && lane_id() == 0) atomicAdd(&B.cnt->n_items_huge, __popcll(m_huge));
    }
    if (tile == ntile - 1 && lane_id() == 63) {             // its running counts now cover the whole batch
        B.cnt->n_clusters = run + (int)own0;
        B.cnt->n_items = bj + n_it; B.cnt->n_items_big = bb; B.cnt->n_items_tiny = bt; B.cnt->n_items_wide = bw;
        // tell the host whether the tiers above 64 signatures have any work - one page-locked word {run sequence, items}: a
        // later run of the SAME upload (same columns, same parameters: same tiers) launches k_refine<64,256> /
        // k_refine<256,2048> only then (no answer yet just means that they are launched)
        if (B.host_flag)
            __hip_atomic_store((unsigned long long*)B.host_flag, ((unsigned long long)(unsigned)B.run_seq << 32) | (unsigned)bb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// CSV_IN_PER_SIG only: dense cluster id of every signature (and allele_id = -1 until k_emit says otherwise), from the
// flag masks k_chain_count published
__global__ __launch_bounds__(256) void k_chain_ids(DevBatch B)
{
    __shared__ i64 sh[4];
    const int wv = threadIdx.x >> 6;
    const i64 base = (i64)blockIdx.x * CH_TILE + wv * (WAVE * CH_ITEMS);
    i64 p0 = 0;
    for (int i = threadIdx.x; i < (int)blockIdx.x; i += 256) p0 += B.partial[i];
    p0 = wave_sum_i64(p0);
    if (lane_id() == 0) sh[wv] = p0;
    // the tile's 32 flag words: lane l holds word l; this wavefront's rows are words 8 wv .. 8 wv + 7
    const u64 mw = lane_id() < CT_WORDS ? B.ch_masks[(i64)blockIdx.x * CT_WORDS + lane_id()] : 0;
    const int pc_inc = wave_incl_scan_i32(__popcll(mw));
    int run = wv ? __builtin_amdgcn_readlane(pc_inc, (wv * CH_ITEMS - 1) & 63) : 0;      // starts in the tile's earlier wavefronts
    __syncthreads();
    run += (int)(sh[0] + sh[1] + sh[2] + sh[3]);
#pragma unroll
    for (int r = 0; r < CH_ITEMS; r++) {
        const u64 m = (u64)readlane_i64x((i64)mw, wv * CH_ITEMS + r);
        const i64 w = base + r * WAVE + lane_id();
        if (w < B.W) { B.cluster_id[w] = run + __popcll(m & (lanemask_lt() | (1ull << lane_id()))) - 1; B.allele_id[w] = -1; }
        run += __popcll(m);
    }
}

// ------------------------------------------------------------------------------------ the position column from 16-bit gaps
// CSV_IN_SIG_DELTA16 (ABI v8): the rebuild order (main script :764-802) makes the position column non-decreasing inside a
// segment and a genome's neighbouring signatures lie ~1 kb apart, so the column - the largest transfer of a gate-first call -
// crosses the link as 16-bit gaps and is rebuilt here, in front of the chain kernels.  Wherever a gap does not exist or does
// not fit (the first row of a chain tile, the first row of a segment, the caller's escape rows) the host side lists an
// ANCHOR {w, a[w]}; a row's position is its last anchor plus the gaps since: a segmented inclusive scan, one workgroup per
// chain tile, eight consecutive rows per thread (one 16-byte load).  Every tile begins with an anchor, so tiles are independent.
struct UnpackArgs {
    const uint16_t* d;          // gaps in w space (+ a tile of slack)
    int*            a;          // the int32 position column
    i64             W;
    const int*      anc_off;    // per chain tile: first anchor; [ntiles]: the end
    const int*      anc_w;      // ascending
    const int*      anc_val;
    int             ntile;      // chain tiles; workgroup `ntile` only writes padding
    int             zero_b;     // gate-first call: also fetch `b` of the rows at position 0 (what k_lazy_zero does, below)
    int             pad;        // 1: the column is followed by a tile + 64 rows of padding (ones); 0: nothing is written past W (the
                                // reads table's start column, CSV_IN_READS_DELTA16)
};
__device__ __forceinline__ i64 lazy_src(const DevBatch& B, int k, i64 w);
__global__ __launch_bounds__(256) void k_unpack_a16(UnpackArgs A, DevBatch B)
{
    __shared__ int s_f[4], s_s[4];
    const int tile = blockIdx.x, t = threadIdx.x, lane = lane_id(), wv = t >> 6;
    // the padding behind the column (positive values: the chain kernel reads any span that begins inside the batch without a
    // range test) is written here too - a DMA copy of a block of ones in front of the column was a blit kernel and an engine
    // switch of its own (5 + 8 us of the one-shot call's timeline)
    const i64 pad_end = A.pad ? A.W + CH_TILE + 64 : A.W;
    if (tile >= A.ntile) {
        for (i64 w = (i64)A.ntile * CH_TILE + t; w < pad_end; w += 256) A.a[w] = 1;
        return;
    }
    const i64 w0 = (i64)tile * CH_TILE + t * 8;
    const uint4 raw = *(const uint4*)(A.d + w0);
    const unsigned g[8] = {raw.x & 0xffffu, raw.x >> 16, raw.y & 0xffffu, raw.y >> 16, raw.z & 0xffffu, raw.z >> 16, raw.w & 0xffffu, raw.w >> 16};
    const int o1 = A.anc_off[tile + 1];
    int lo = A.anc_off[tile], hi = o1;                      // first anchor at or behind this thread's first row
    while (lo < hi) { const int mid = (lo + hi) >> 1; if ((i64)A.anc_w[mid] < w0) lo = mid + 1; else hi = mid; }
    int ai = lo;
    i64 aw = ai < o1 ? (i64)A.anc_w[ai] : (i64)INT64_MAX;
    int loc[8];
    unsigned since = 0;                                     // bit r: an anchor at or before row r inside this thread
    int sum = 0, F = 0;
#pragma unroll
    for (int r = 0; r < 8; r++) {
        if (w0 + r == aw) { sum = A.anc_val[ai]; F = 1; ai++; aw = ai < o1 ? (i64)A.anc_w[ai] : (i64)INT64_MAX; }
        else sum += (int)g[r];
        loc[r] = sum;
        if (F) since |= 1u << r;
    }
    // segmented inclusive scan of the threads' (F, sum): (F1, S1) + (F2, S2) = (F1 | F2, F2 ? S2 : S1 + S2)
    int fs = F, ss = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int fo = __shfl_up(fs, off), so = __shfl_up(ss, off);
        if (lane >= off) { if (!fs) ss += so; fs |= fo; }
    }
    if (lane == 63) { s_f[wv] = fs; s_s[wv] = ss; }
    __syncthreads();
    int ps = 0;                                             // what the earlier wavefronts carry in
    for (int q = 0; q < wv; q++) { if (s_f[q]) ps = s_s[q]; else ps += s_s[q]; }
    int ef = __shfl_up(fs, 1), es = __shfl_up(ss, 1);       // the lanes before this one, inside the wavefront
    if (lane == 0) { ef = 0; es = 0; }
    const int carry = ef ? es : ps + es;                    // (a tile's first row is an anchor: the carry is always defined by one)
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const i64 w = w0 + r;
        const int v = ((since >> r) & 1) ? loc[r] : loc[r] + carry;
        if (w < A.W) {
            A.a[w] = v;
            // (gate-first: the (0,0) look-alike rule reads the length of a row at position 0 - see k_lazy_zero)
            if (CSV_UNLIKELY(A.zero_b && v == 0)) {
                const i64 src = lazy_src(B, seg_of(B, w), w);
                const_cast<int*>(B.b.p32)[w] = B.h_rows8 ? B.h_rows8[src].x : ((const int*)B.h_b)[src];
            }
        } else if (w < pad_end) A.a[w] = 1;
    }
}

// CSV_IN_READS_DELTA16: the reads table's end column from 16-bit lengths (behind k_unpack_a16 for the starts, or on the starts
// as they were copied), then the few rows whose length does not fit - their ends travel as they are
__global__ __launch_bounds__(256) void k_reads_end16(const int* start, const uint16_t* len16, int* end, i64 n)
{
    const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    if (i < n) end[i] = start[i] + (int)len16[i];
}
// csv_batch_in.r_idp: the read id and the primary flag arrive as one word; the kernels of the reads stage read two columns
__global__ __launch_bounds__(256) void k_reads_split_idp(const unsigned* idp, int* id, uint8_t* primary, i64 n)
{
    const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { const unsigned v = idp[i]; id[i] = (int)(v & 0x7fffffffu); primary[i] = (uint8_t)(v >> 31); }
}
__global__ __launch_bounds__(256) void k_scatter_rows_i32(const i64* row, const int* val, int* out, i64 n_esc, i64 n)
{
    const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    if (i < n_esc) { const i64 r = row[i]; if (r >= 0 && r < n) out[r] = val[i]; }
}

// ------------------------------------------------------------------------------------ gate-first fetch
// A one-shot call from page-locked columns (csv_cluster_batch) sends only the POSITION column across PCIe in bulk (plus b / aux
// of INV / TRA segments, whose chain predicates read them): 82 % of a 30x genome's signatures sit in clusters that fail the size
// gate (INDEL:62-64, 86) and no kernel ever reads their lengths, read ids or sequence lengths - 27 of the 44.5 MB a bulk upload
// moves.  Once k_chain_count / k_chain_apply have listed the clusters that pass, k_lazy_fetch reads exactly their rows out of the
// caller's host columns (the device sees page-locked host memory; a load from it is a PCIe read) into the device columns, at
// the positions a bulk copy would have put them: every later kernel is unchanged.
// source row (the caller's global signature index) of compact row w of segment k
__device__ __forceinline__ i64 lazy_src(const DevBatch& B, int k, i64 w) { return B.seg[k].sig_begin + (w - B.woff[k]); }

// Before the chain kernels: the (0,0) look-alike rule (INDEL:62-64: an element equal to the reference's [0,0,''] sentinel ends
// its cluster) is the one place where a chain predicate of a DEL / INS / DUP segment looks at a length - and only where the
// position is 0.  Those rows (a handful per genome, at the head of a segment) get their `b` now.
template <bool NARROW> __global__ __launch_bounds__(256) void k_lazy_zero(DevBatch B)
{
    for (i64 w = (i64)blockIdx.x * 256 + threadIdx.x; w < B.W; w += (i64)gridDim.x * 256) {
        const bool z = NARROW ? (B.a.p32[w] == 0) : (B.a.p64[w] == 0);
        if (CSV_UNLIKELY(z)) {
            const i64 src = lazy_src(B, seg_of(B, w), w);
            if constexpr (NARROW) const_cast<int*>(B.b.p32)[w] = B.h_rows8 ? B.h_rows8[src].x : ((const int*)B.h_b)[src];
            else const_cast<i64*>(B.b.p64)[w] = ((const i64*)B.h_b)[src];
        }
    }
}

// One workgroup per chain tile.  Which rows: those of the tile's work items (tile_items, clipped to the tile) and the rows
// before the tile's first cluster start, which continue a cluster of an earlier tile - fetched whether that cluster passed
// the gate or not (two rows per tile boundary on average; a pile-up of a million signatures is fetched by all the tiles it
// covers at once, not by the one wavefront that owns the item).  The rows become a bit mask in LDS, then every thread takes 8
// rows, all loads of a thread issued before its first store: consecutive lanes, consecutive rows, consecutive host addresses.
template <bool NARROW> __global__ __launch_bounds__(256) void k_lazy_fetch(DevBatch B)
{
    __shared__ u64 s_G[CT_WORDS];
    const int tile = blockIdx.x, t = threadIdx.x;
    const i64 tile0 = (i64)tile * CH_TILE;
    const int4 tc = B.tile_cnt[tile];
    const int lead = B.tile_lead[tile];
    const int4* trp = B.tile_info + (i64)TILE_REC * tile;
    const int4 t0 = trp[0], t1 = trp[1], t2 = trp[2], t3 = trp[3];
    const int k0 = __builtin_amdgcn_readfirstlane(t0.x), k1 = __builtin_amdgcn_readfirstlane(t0.y), nin = __builtin_amdgcn_readfirstlane(t0.z);
    const TileSegI g0 = tile_seg(t1), g1 = tile_seg(t2), g2 = tile_seg(t3);
    // source offsets of the inline segments (w -> the caller's row): issued with the first round
    i64 d0 = 0, d1 = 0, d2 = 0;
    if (nin >= 1) d0 = B.seg[g0.k].sig_begin - B.woff[g0.k];
    if (nin >= 2) d1 = B.seg[g1.k].sig_begin - B.woff[g1.k];
    if (nin >= 3) d2 = B.seg[g2.k].sig_begin - B.woff[g2.k];
    if (t < CT_WORDS) {
        const int lo = t * 64;
        s_G[t] = lead >= lo + 64 ? ~0ull : (lead > lo ? (1ull << (lead - lo)) - 1ull : 0ull);
    }
    __syncthreads();
    const int n_it = tc.y;
    for (int i = t; i < n_it; i += 256) {
        const int4 it = B.tile_items[(i64)tile * TI_STRIDE + i];
        const int p0 = it.x - (int)tile0;
        const int p1 = it.y > CH_TILE - p0 ? CH_TILE : p0 + it.y;
        for (int wd = p0 >> 6; wd <= (p1 - 1) >> 6; wd++) {
            const int lo = wd * 64, x = p0 > lo ? p0 - lo : 0, y = p1 - lo < 64 ? p1 - lo : 64;
            atomicOr(&s_G[wd], (y >= 64 ? ~0ull : (1ull << y) - 1ull) & ~((1ull << x) - 1ull));
        }
    }
    __syncthreads();
    typedef typename std::conditional<NARROW, int, i64>::type raw_t;
    raw_t vb[CH_ITEMS]; int vr[CH_ITEMS], vx[CH_ITEMS];
    unsigned on = 0, wb = 0, wx = 0;                        // per row: fetched at all / b fetched / aux fetched
#pragma unroll
    for (int r = 0; r < CH_ITEMS; r++) {
        const int p = r * 256 + t;
        const i64 w = tile0 + p;
        vb[r] = 0; vr[r] = 0; vx[r] = 0;
        if (((s_G[p >> 6] >> (p & 63)) & 1ull) && w < B.W) {
            int type; i64 src;
            if (CSV_LIKELY(nin >= 1)) {
                const bool i2 = nin > 2 && (int)w >= g2.sf, i1 = nin > 1 && (int)w >= g1.sf;
                type = i2 ? g2.type : (i1 ? g1.type : g0.type);
                src = w + (i2 ? d2 : (i1 ? d1 : d0));
            } else {
                const int k = seg_in_tile(B, w, k0, k1);
                type = B.seg[k].svtype; src = lazy_src(B, k, w);
            }
            on |= 1u << r;
            bool have_b = false;
            if constexpr (NARROW) {
                if (B.h_rows8) {                            // one 8-byte load: {b, read_id} of the row (b is simply not stored for INV / TRA)
                    const int2 br = B.h_rows8[src];
                    vr[r] = br.y; vb[r] = br.x; have_b = true;
                }
            }
            if (!have_b) vr[r] = B.h_rid[src];
            if (!pair_type(type)) {                         // (INV / TRA: b and aux came with the bulk copy - the chain kernels read them)
                wb |= 1u << r;
                if (!have_b) vb[r] = ((const raw_t*)B.h_b)[src];
                wx |= 1u << r;                              // (aux of a DEL / DUP row is zero on the device: include/cutesv_hip.h)
                if (type == CSV_INS) vx[r] = B.h_aux[src];
            }
        }
    }
#pragma unroll
    for (int r = 0; r < CH_ITEMS; r++) {
        const i64 w = tile0 + r * 256 + t;
        if ((on >> r) & 1) const_cast<int*>(B.rid)[w] = vr[r];
        if ((wb >> r) & 1) { if constexpr (NARROW) const_cast<int*>(B.b.p32)[w] = vb[r]; else const_cast<i64*>(B.b.p64)[w] = vb[r]; }
        if ((wx >> r) & 1) const_cast<int*>(B.aux)[w] = vx[r];
    }
}

// ------------------------------------------------------------------------------------ input order contract
// Inside a segment the rows must be in the order the reference's rebuild step leaves them (main script
// :764-802 sort keys, :958-969 adjacent de-duplication): strictly increasing in (a, b, read_id) for DEL / INS /
// DUP and in (aux, a, b, read_id) for INV (aux = strand) and TRA (aux = chr2, type).  A violation means the
// caller's columns are not what phase 3 of cuteSV would see; csv_batch_validate reports it.
__global__ __launch_bounds__(256) void k_validate_order(DevBatch B)
{
    const i64 w = (i64)blockIdx.x * 256 + threadIdx.x;
    if (w >= B.W || w == 0) return;
    const int k = seg_of(B, w);
    if (w == B.woff[k]) return;
    const int type = B.seg[k].svtype;
    const i64 a0 = B.a[w - 1], a1 = B.a[w], b0 = B.b[w - 1], b1 = B.b[w];
    const int r0 = B.rid[w - 1], r1 = B.rid[w];
    int c = 0;                                          // sign of key(w) - key(w - 1)
    if (type == CSV_INV || type == CSV_TRA) { const int x0 = B.aux[w - 1], x1 = B.aux[w]; c = (x1 > x0) - (x1 < x0); }
    if (c == 0) c = (a1 > a0) - (a1 < a0);
    if (c == 0) c = (b1 > b0) - (b1 < b0);
    if (c == 0) c = (r1 > r0) - (r1 < r0);
    if (c == 0 && type == CSV_INS) c = 1;               // equal (pos, len, read): the sequences may still differ
    if (c <= 0) atomicOr(&B.cnt->error, ERR_SIG_ORDER);
}

// ------------------------------------------------------------------------------------ refine
// Working arrays of one cluster (P = padded power of two >= m; every array has P + ARR_PAD entries):
//   K  u64   sort keys, later the a-values (pos / pos1) in rank order
//   X  i64   b-values (len / pos2) in rank order
//   V1 int   local index of rank r (bit 31: first-seen flag for DUP/INV/TRA)
//   V2 int   local indices in (read id, index) order, later first rank of each group (allele / sub-cluster)
//   V3 int   kept-signature-of-first-appearance (INDEL) / rank of local index, later per-group values
//   V4 int   group of rank r, later per-group values
//   V5 int   per-group values
// The pointers are flat: LDS for P <= CAP, the cluster's own slice of the global scratch above.
// LDS = true: real local-address-space pointers (ds_read / ds_write); LDS = false: global scratch.  (A first
// version reached both through flat pointers: every LDS access then took the slow FLAT path and the mid
// tier spent ~25 cycles per instruction waiting on ~700 flat LDS round trips per cluster.)
#define CSV_LDS __attribute__((address_space(3)))
template <bool LDS> struct MemT;
template <> struct MemT<true>  { typedef CSV_LDS u64* u64p; typedef CSV_LDS i64* i64p; typedef CSV_LDS const i64* ci64p; typedef CSV_LDS int* intp; };
template <> struct MemT<false> { typedef u64* u64p; typedef i64* i64p; typedef const i64* ci64p; typedef int* intp; };
template <bool LDS> struct ArraysT {
    typename MemT<LDS>::u64p K; typename MemT<LDS>::i64p X;
    typename MemT<LDS>::intp V1; typename MemT<LDS>::intp V2; typename MemT<LDS>::intp V3; typename MemT<LDS>::intp V4; typename MemT<LDS>::intp V5;
};

struct ItemCtx {
    int j, cid, k, s, m, P;
    int ib;                // index bits of this cluster's sort keys
    i64 gsig0;             // global signature index of w = s
};

// One-wavefront tiers: the whole sort runs in registers.  Lane l holds elements l, l + 64, ... (E per lane);
// partners at distance >= 64 are in the same lane, partners at distance < 64 sit in lane l ^ j and are
// exchanged with DPP (quad_perm for 1 and 2, row_ror:8 for 8) or ds_bpermute — no LDS arrays, no barriers
// between the stages.
template <int J> __device__ __forceinline__ u64 xor_lane_u64(u64 v)
{
    int lo = (int)(v & 0xffffffffull), hi = (int)(v >> 32);
    if (J == 1) { lo = __builtin_amdgcn_update_dpp(lo, lo, 0xB1, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0xB1, 0xf, 0xf, false); }
    else if (J == 2) { lo = __builtin_amdgcn_update_dpp(lo, lo, 0x4E, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x4E, 0xf, 0xf, false); }
    else if (J == 8) { lo = __builtin_amdgcn_update_dpp(lo, lo, 0x128, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x128, 0xf, 0xf, false); }
    else { lo = __shfl_xor(lo, J); hi = __shfl_xor(hi, J); }
    return ((u64)(unsigned)hi << 32) | (unsigned)lo;
}
template <int J> __device__ __forceinline__ unsigned xor_lane_u32(unsigned v)
{
    if (J == 1) return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xf, 0xf, false);
    if (J == 2) return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xf, 0xf, false);
    if (J == 8) return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x128, 0xf, 0xf, false);
    return (unsigned)__shfl_xor((int)v, J);
}
// Which lanes of register e keep the SMALLER key of a compare-exchange with the partner J lanes away in stage KK of the network
// over 64 * E elements (element index = e * 64 + lane): a compile-time constant.  Written with the lane id at run time -
// (lane & J) == 0, ((e * 64 + lane) & kk) == 0 - the compiler computed every one of these masks once, in front of the kernel's
// loop, and held them in SGPR pairs for the whole kernel: 49 SGPR spills (v_writelane / v_readlane pairs) in k_refine<64,256>,
// 22 in k_refine_indel_wave.  As literals they cost one s_mov_b64 where they are used.
constexpr u64 keepmin_mask_e(int e, int KK, int J)
{
    u64 m = 0;
    for (int l = 0; l < 64; l++) { const bool lower = (l & J) == 0, asc = ((e * 64 + l) & KK) == 0; if (lower == asc) m |= 1ull << l; }
    return m;
}
template <int e, int E, int J, int KK, class T> __device__ __forceinline__ void bitonic_lanes_e(T (&k)[E])
{
    if constexpr (e < E) {
        constexpr u64 KEEPMIN = keepmin_mask_e(e, KK, J);
        T other;
        if constexpr (sizeof(T) == 8) other = (T)xor_lane_u64<J>((u64)k[e]); else other = (T)xor_lane_u32<J>((unsigned)k[e]);
        const T mn = k[e] < other ? k[e] : other, mx = k[e] < other ? other : k[e];
        k[e] = __builtin_amdgcn_inverse_ballot_w64(KEEPMIN) ? mn : mx;
        bitonic_lanes_e<e + 1, E, J, KK, T>(k);
    }
}
// same-lane partners (distance j >= 64): registers e and e | de, direction from the element index - compile-time per register
template <int e, int E, int DE, int KK, class T> __device__ __forceinline__ void bitonic_regs_e(T (&k)[E])
{
    if constexpr (e < E) {
        if constexpr ((e & DE) == 0 && (e | DE) < E) {
            constexpr bool asc = ((e * 64) & KK) == 0;      // (KK >= 128 here: the lane bits do not reach it)
            const T x = k[e], y = k[e | DE];
            const bool sw = (x > y) == asc;
            k[e] = sw ? y : x; k[e | DE] = sw ? x : y;
        }
        bitonic_regs_e<e + 1, E, DE, KK, T>(k);
    }
}
template <int E, int KK, class T> __device__ __forceinline__ void bitonic_stage(T (&k)[E])
{
    if constexpr (KK <= 64 * E) {
        if constexpr (KK >= 512) bitonic_regs_e<0, E, 4, KK, T>(k);
        if constexpr (KK >= 256) bitonic_regs_e<0, E, 2, KK, T>(k);
        if constexpr (KK >= 128) bitonic_regs_e<0, E, 1, KK, T>(k);
        if constexpr (KK >= 64) bitonic_lanes_e<0, E, 32, KK, T>(k);
        if constexpr (KK >= 32) bitonic_lanes_e<0, E, 16, KK, T>(k);
        if constexpr (KK >= 16) bitonic_lanes_e<0, E, 8, KK, T>(k);
        if constexpr (KK >= 8) bitonic_lanes_e<0, E, 4, KK, T>(k);
        if constexpr (KK >= 4) bitonic_lanes_e<0, E, 2, KK, T>(k);
        bitonic_lanes_e<0, E, 1, KK, T>(k);
        bitonic_stage<E, KK * 2, T>(k);
    }
}
template <int E> __device__ __forceinline__ void bitonic_wave(u64 (&k)[E])
{
    static_assert(E <= 8, "register pairs up to distance 4");
    bitonic_stage<E, 2, u64>(k);
}
template <int E, class KP> __device__ __forceinline__ void bitonic_wave_mem(KP K, int P)
{
    const int lane = lane_id();
    u64 k[E];
#pragma unroll
    for (int e = 0; e < E; e++) k[e] = (e * 64 + lane < P) ? K[e * 64 + lane] : PAD_KEY;
    bitonic_wave<E>(k);
#pragma unroll
    for (int e = 0; e < E; e++) if (e * 64 + lane < P) K[e * 64 + lane] = k[e];
}

// The same network on 32-bit keys: half the instructions per compare-exchange (one DPP / bpermute, one compare, one select
// instead of two, three and two).  Both sort keys of the refine kernels are (value << S) | small index: when every value of a
// cluster fits 32 - LB bits (read ids below 2^24, lengths below 2^24: nearly always) the keys are packed, sorted and unpacked
// in registers.  Returns false (nothing done) when they do not fit.
template <int E> __device__ __forceinline__ void bitonic_wave32(unsigned (&k)[E])
{
    static_assert(E <= 8, "register pairs up to distance 4");
    bitonic_stage<E, 2, unsigned>(k);
}
template <int E, class KP> __device__ __forceinline__ bool bitonic_wave_mem32(KP K, int P, int S)
{
    constexpr int LB = 8;                                   // the index of at most 256 elements
    const int lane = lane_id();
    u64 k[E];
    bool fits = true;
#pragma unroll
    for (int e = 0; e < E; e++) {
        k[e] = (e * 64 + lane < P) ? K[e * 64 + lane] : PAD_KEY;
        if (k[e] != PAD_KEY && (((k[e] >> S) + 1) >> (32 - LB)) != 0) fits = false;      // (the all-ones word is the padding key)
    }
    u64 base = 0;
    if (__ballot(!fits)) {
        // not as they are: relative to the smallest value of the cluster (second positions of DUP / INV / TRA signatures are
        // genome coordinates, but those of one cluster lie close together)
        u64 mn = ~0ull;
#pragma unroll
        for (int e = 0; e < E; e++) if (k[e] != PAD_KEY && (k[e] >> S) < mn) mn = k[e] >> S;
        for (int d = 32; d > 0; d >>= 1) { const u64 o = (u64)shfl_xor_i64((i64)mn, d); mn = o < mn ? o : mn; }
        fits = true;
#pragma unroll
        for (int e = 0; e < E; e++) if (k[e] != PAD_KEY && ((((k[e] >> S) - mn) + 1) >> (32 - LB)) != 0) fits = false;
        if (__ballot(!fits)) return false;
        base = mn;
    }
    unsigned k32[E];
#pragma unroll
    for (int e = 0; e < E; e++) k32[e] = k[e] == PAD_KEY ? 0xffffffffu : (((unsigned)((k[e] >> S) - base) << LB) | ((unsigned)k[e] & 0xffu));
    bitonic_wave32<E>(k32);
#pragma unroll
    for (int e = 0; e < E; e++)
        if (e * 64 + lane < P) K[e * 64 + lane] = k32[e] == 0xffffffffu ? PAD_KEY : ((((u64)(k32[e] >> LB) + base) << S) | (u64)(k32[e] & 0xffu));
    return true;
}

// S: the keys are (value << S) | index, index < P (S = 0: unknown layout, always the 64-bit network)
// ME: the largest P / 64 the caller can have (one-wavefront tiers: their CAP / 64) - the wider instantiations are not compiled in
template <int BLOCK, class KP, int ME = 4> __device__ void bitonic_sort(KP K, int P, int S = 0)
{
    if (BLOCK == 64 && P <= 256) {                    // callers synchronised before; the results are visible after this barrier
        bool done = false;
        if (S > 0 && !CSV_ABL(19)) {
            if (P <= 64) done = bitonic_wave_mem32<1>(K, P, S);
            else if (ME >= 2 && P <= 128) done = bitonic_wave_mem32<(ME >= 2 ? 2 : 1)>(K, P, S);
            else if (ME >= 4) done = bitonic_wave_mem32<(ME >= 4 ? 4 : 1)>(K, P, S);
        }
        if (!done) {
            if (P <= 64) bitonic_wave_mem<1>(K, P);
            else if (ME >= 2 && P <= 128) bitonic_wave_mem<(ME >= 2 ? 2 : 1)>(K, P);
            else if (ME >= 4) bitonic_wave_mem<(ME >= 4 ? 4 : 1)>(K, P);
        }
        __syncthreads();
        return;
    }
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < P; i += BLOCK) {
                const int x = i ^ j;
                if (x > i) {
                    const u64 ki = K[i], kx = K[x];
                    const bool asc = (i & k) == 0;
                    if ((ki > kx) == asc) { K[i] = kx; K[x] = ki; }
                }
            }
            __syncthreads();
        }
    }
}

// sum over the workgroup; `red` is a small LDS array; every thread gets the total
template <int BLOCK> __device__ i64 block_sum_i64(i64 v, i64* red)
{
    v = wave_sum_i64(v);
    if (BLOCK == 64) return v;
    __syncthreads();
    if (lane_id() == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    i64 t = 0;
    for (int k = 0; k < BLOCK / 64; k++) t += red[k];
    return t;
}

// inclusive scan over the workgroup of one int per thread; total via *tot
template <int BLOCK> __device__ int block_incl_scan(int v, int* tot, i64* red)
{
    const int inc = wave_incl_scan_i32(v);
    if (BLOCK == 64) { *tot = __builtin_amdgcn_readlane(inc, 63); return inc; }
    __syncthreads();
    if (lane_id() == 63) red[threadIdx.x >> 6] = inc;
    __syncthreads();
    int off = 0, t = 0;
    for (int k = 0; k < BLOCK / 64; k++) { if (k < (int)(threadIdx.x >> 6)) off += (int)red[k]; t += (int)red[k]; }
    *tot = t;
    return inc + off;
}

// numpy's pairwise summation over sq(i) = (v[i] - mean)^2, evaluated by ONE lane in numpy's exact
// association order (oracle: pairwise_f64 / csvo_np_sum_f64).
template <class VP> __device__ double np_sumsq_leaf(VP v, int n, double mean)
{
    if (n < 8) {
        double r = 0.0;
        for (int i = 0; i < n; i++) { const double x = (double)v[i] - mean; r += x * x; }
        return r;
    }
    double r[8];
    for (int j = 0; j < 8; j++) { const double x = (double)v[j] - mean; r[j] = x * x; }
    int i;
    for (i = 8; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; j++) { const double x = (double)v[i + j] - mean; r[j] += x * x; }
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) { const double x = (double)v[i] - mean; res += x * x; }
    return res;
}

// explicit-stack form of  pw(a, n) = n <= 128 ? leaf : pw(a, n2) + pw(a + n2, n - n2),  n2 = n/2 - (n/2) % 8.
// The stack (5 ints per level: offset, length, state, partial sum lo/hi) lives in the cluster's own working
// memory (LDS or global scratch), not in private arrays, so the kernel needs no scratch segment.
template <class VP, class IP> __device__ double np_sumsq_chunk(VP v, int n, double mean, IP stk)
{
    if (n <= 128) return np_sumsq_leaf(v, n, mean);
    int sp = 0;
    stk[0] = 0; stk[1] = n; stk[2] = 0;
    double ret = 0.0;
    while (sp >= 0) {
        const int off = stk[5 * sp], len = stk[5 * sp + 1], state = stk[5 * sp + 2];
        if (len <= 128) { ret = np_sumsq_leaf(v + off, len, mean); sp--; continue; }
        int n2 = len / 2; n2 -= n2 % 8;
        if (state == 0) {
            stk[5 * sp + 2] = 1;
            stk[5 * sp + 5] = off; stk[5 * sp + 6] = n2; stk[5 * sp + 7] = 0; sp++;
        } else if (state == 1) {
            const i64 bits = __double_as_longlong(ret);
            stk[5 * sp + 3] = (int)(bits & 0xffffffffll); stk[5 * sp + 4] = (int)(bits >> 32); stk[5 * sp + 2] = 2;
            stk[5 * sp + 5] = off + n2; stk[5 * sp + 6] = len - n2; stk[5 * sp + 7] = 0; sp++;
        } else {
            const i64 bits = ((i64)stk[5 * sp + 4] << 32) | (unsigned)stk[5 * sp + 3];
            ret = __longlong_as_double(bits) + ret; sp--;
        }
    }
    return ret;
}

// np.std of the int64 values v[0..n) -> the integer of cal_CIPOS (GT:58-60); one lane, any n
constexpr int CIPOS_STACK_INTS = 5 * 16;
template <class VP, class IP> __device__ int cipos_of(VP v, int n, i64 sum, const double* sqrt_tab, IP stk)
{
    const double mean = (double)sum / (double)n;
    double acc = 0.0;
    for (int off = 0; off < n; off += 8192) {
        const int c = n - off < 8192 ? n - off : 8192;
        acc += np_sumsq_chunk(v + off, c, mean, stk);
    }
    const double sd = sqrt(acc / (double)n);
    return (int)(1.96 * sd / sqrt_tab[n]);            // (n <= the segment's length < the table's size, see SQRT_TAB)
}

// one thread publishes an item's result: slot count and packed (valid calls << 32 | their supports).
// (Accumulating tile sums here with atomics was tried: even non-returning adds on ~n/8 words doubled
// the kernel time, so the prefix is a separate, single-workgroup sweep: k_items_scan.)
__device__ __forceinline__ void item_done(const DevBatch& B, int j, int ncalls, int nsup)
{
    B.item_cnt[j] = ((i64)ncalls << 32) + (i64)nsup;
}
// where slot `slot` of item j (first signature s) keeps its temp record
__device__ __forceinline__ TmpRec* tmp_slot(const DevBatch& B, int j, int s, int slot) { return slot == 0 ? &B.t_rec0[j] : &B.t_rec[s + slot]; }
// The same np.std / cal_CIPOS for n <= 256, computed by a whole wavefront from values held in LDS: numpy's
// recursion has at most three leaves there (n -> n2 = (n/2) & ~7 and n - n2 <= 135 -> 64 + <= 71); each leaf is
// summed by an 8-lane group (lane j = accumulator j of numpy's 8-way unrolled loop), then numpy's fixed combine
// tree and sequential tail.  Every lane returns the result.  No private arrays, no call: the one-wavefront tiers
// stay free of scratch.
// cal_CIPOS from the exact integer variance, as in the register tier (indel_unit): int(1.96 * std / n ** 0.5) through
// N = n * sum(d^2) - sum(d)^2 (d = x - x[0]), one float32 square root and the table factor 1.96 / (n * n ** 0.5).  false: the
// values are too far apart for the integers, or the result lies within 2e-6 of an integer - the caller replays numpy then.
template <class VP> __device__ __forceinline__ bool cipos_fast(VP v, int n, i64 sum, const float* cipk_tab, int& out)
{
    const i64 x0 = v[0];
    i64 s2 = 0;
    bool small = (u64)x0 < (1ull << 31);
    for (int i = lane_id(); i < n; i += 64) {
        const i64 d = v[i] - x0;
        small = small && d > -(1 << 20) && d < (1 << 20);
        s2 += d * d;
    }
    if (__ballot(!small)) return false;
    s2 = wave_sum_i64(s2);
    const i64 s1 = sum - (i64)n * x0;
    const i64 N = (i64)n * s2 - s1 * s1;                    // n^2 * variance, exact (< 2^57)
    const float vf = __builtin_amdgcn_sqrtf((float)(double)N) * cipk_tab[n & (SQRT_TAB - 1)];
    if (N < 0 || (vf > 0.5f && fabsf(vf - rintf(vf)) <= 2e-6f * vf)) return false;
    out = (int)vf;
    return true;
}
template <class VP> __device__ __forceinline__ int cipos_wave(VP v, int n, i64 sum, const double* sqrt_tab)
{
    const int lane = lane_id(), g = lane >> 3, j = lane & 7;
    const double mean = (double)sum / (double)n;
    int off0 = 0, len0 = n, off1 = 0, len1 = 0, off2 = 0, len2 = 0, cnt = 1;
    if (n > 128) {
        int n2 = n / 2; n2 -= n2 % 8;
        len0 = n2; off1 = n2; len1 = n - n2; cnt = 2;
        if (len1 > 128) { int m2 = len1 / 2; m2 -= m2 % 8; off2 = off1 + m2; len2 = len1 - m2; len1 = m2; cnt = 3; }
    }
    const int off = g == 0 ? off0 : (g == 1 ? off1 : off2);
    const int len = g == 0 ? len0 : (g == 1 ? len1 : (g == 2 ? len2 : 0));
    const bool act = g < cnt;
    const int nfull = len - (len & 7);
    double acc = 0.0;
    if (act && len >= 8) { const double x = (double)v[off + j] - mean; acc = x * x; }
    for (int t = 1; t < 16; t++) {                                   // leaves hold <= 128 values
        if (!__ballot(act && len >= 8 && 8 * t < nfull)) break;
        if (act && len >= 8 && 8 * t < nfull) { const double x = (double)v[off + 8 * t + j] - mean; acc += x * x; }
    }
    const double t1 = acc + shfl_xor_f64(acc, 1);
    const double t2 = t1 + shfl_xor_f64(t1, 2);
    const double t3 = t2 + shfl_xor_f64(t2, 4);
    double res = (len >= 8) ? t3 : 0.0;
    const int start = (len >= 8) ? nfull : 0;
    for (int e = 0; e < 7; e++) {
        if (!__ballot(act && j == 0 && start + e < len)) break;
        if (act && j == 0 && start + e < len) { const double x = (double)v[off + start + e] - mean; res += x * x; }
    }
    const double l0 = __longlong_as_double(readlane_i64x(__double_as_longlong(res), 0));
    const double l1 = __longlong_as_double(readlane_i64x(__double_as_longlong(res), 8));
    const double l2 = __longlong_as_double(readlane_i64x(__double_as_longlong(res), 16));
    const double tot = cnt == 1 ? l0 : (cnt == 2 ? l0 + l1 : l0 + (l1 + l2));
    const double sd = sqrt(tot / (double)n);
    return (int)(1.96 * sd / sqrt_tab[n]);
}

__device__ __forceinline__ void item_none(const DevBatch& B, int j)
{
    if (threadIdx.x == 0) item_done(B, j, 0, 0);
}

// Temp call slots need no allocation: a cluster of m signatures yields at most m calls (every allele /
// sub-cluster holds >= 1 signature; TRA's two calls need two sub-clusters), so its records live at
// t_*[s + slot], inside its own [s, e) range of W-sized arrays.  (A shared atomic bump pointer was
// measured at ~88 allocations/us on one word and dominated the kernel.)

// phase A: sort by (read id, local index); returns the number of distinct reads.
// Leaves K sorted and V2[q] = local index at sorted position q.
template <int BLOCK, bool LDS, int ME = 4> __device__ int sort_by_read(const DevBatch& B, const ItemCtx& it, const ArraysT<LDS>& A, i64* red)
{
    for (int i = threadIdx.x; i < it.P; i += BLOCK)
        A.K[i] = i < it.m ? (((u64)(unsigned)B.rid[it.s + i]) << 32) | (unsigned)i : PAD_KEY;
    __syncthreads();
    bitonic_sort<BLOCK, decltype(A.K), ME>(A.K, it.P, 32);
    int runs = 0;
    for (int q = threadIdx.x; q < it.m; q += BLOCK) {
        const u64 k = A.K[q];
        A.V2[q] = (int)(k & 0xffffffffull);
        runs += (q == 0) || ((A.K[q - 1] >> 32) != (k >> 32));
    }
    const int U = (int)block_sum_i64<BLOCK>(runs, red);
    __syncthreads();
    return U;
}

// A length / pos2 value outside [0, 2^(63 - ib)) cannot be packed into a sort key: the cluster emits nothing and its
// segment is flagged (csv_batch_out.seg_status); every other cluster of the batch is unaffected.
template <int BLOCK> __device__ bool keys_out_of_range(const DevBatch& B, const ItemCtx& it, i64* red)
{
    int bad = 0;
    for (int i = threadIdx.x; i < it.m; i += BLOCK) bad |= (((u64)B.b[it.s + i]) >> (63 - it.ib)) != 0;
    if (block_sum_i64<BLOCK>(bad, red) == 0) return false;
    if (threadIdx.x == 0) atomicOr(&B.seg_err[it.k], CSV_SEG_KEY_RANGE);
    __syncthreads();
    return true;
}

// ---- DEL / INS: generate_del_cluster / generate_ins_cluster (INDEL:110-219, 319-432)
// STAGED (the one-wavefront tier, m <= 256): the cluster's four columns are read ONCE, in one round trip at the head - lengths
// into X, sequence lengths into V5 (free on this path), positions kept in registers until the lengths are dead, then into X -
// and every later step indexes LDS.  The unstaged form goes back to global memory for the lengths (twice), the positions and
// the sequence lengths, each a dependent round trip of a kernel that is a chain of ~10 of them per cluster.
template <int BLOCK, bool LDS, bool SMALLN, bool STAGED = false, int ME = 4> __device__ void refine_indel(const DevBatch& B, const ItemCtx& it, const ArraysT<LDS>& A, i64* red, int* ired)
{
    static_assert(!STAGED || (BLOCK == 64 && LDS && SMALLN), "the staged form is the one-wavefront LDS tier");
    constexpr int E = ME;                                                   // elements per lane of the staged form (m <= 64 ME: the tier's capacity)
    const csv_segment& sg = B.seg[it.k];
    const int m = it.m, P = it.P, s = it.s, ib = it.ib;
    const u64 imask = (1ull << ib) - 1ull;
    i64 st_a[E];
    int U;
    if constexpr (STAGED) {
        i64 st_b[E]; int st_r[E], st_x[E];
#pragma unroll
        for (int e = 0; e < E; e++) {
            st_a[e] = 0; st_b[e] = 0; st_r[e] = 0; st_x[e] = 0;
            if (64 * e < m) {                                               // (wave-uniform)
                const int i = 64 * e + lane_id();
                const i64 w = s + (i < m ? i : 0);
                st_b[e] = B.b[w]; st_r[e] = B.rid[w]; st_a[e] = B.a[w]; st_x[e] = B.aux[w];
            }
        }
        int bad = 0;
#pragma unroll
        for (int e = 0; e < E; e++) bad |= (64 * e + lane_id() < m) && ((((u64)st_b[e]) >> (63 - ib)) != 0);
        if (__ballot(bad)) {                                                // keys_out_of_range
            if (threadIdx.x == 0) atomicOr(&B.seg_err[it.k], CSV_SEG_KEY_RANGE);
            item_none(B, it.j);
            return;
        }
#pragma unroll
        for (int e = 0; e < E; e++) {
            const int i = 64 * e + lane_id();
            if (i < P) A.K[i] = i < m ? (((u64)(unsigned)st_r[e]) << 32) | (unsigned)i : PAD_KEY;
            if (i < m) { A.X[i] = st_b[e]; A.V5[i] = st_x[e]; }
        }
        __syncthreads();
        bitonic_sort<BLOCK, decltype(A.K), ME>(A.K, P, 32);
        int runs = 0;
        for (int q = threadIdx.x; q < m; q += BLOCK) {
            const u64 k = A.K[q];
            A.V2[q] = (int)(k & 0xffffffffull);
            runs += (q == 0) || ((A.K[q - 1] >> 32) != (k >> 32));
        }
        U = wave_sum_i32(runs);
        __syncthreads();
    } else {
        if (keys_out_of_range<BLOCK>(B, it, red)) { item_none(B, it.j); return; }
        U = sort_by_read<BLOCK, LDS, ME>(B, it, A, red);
    }
    if (U < sg.read_count) { item_none(B, it.j); return; }                  // INDEL:133-134

    // per run of equal read id: first appearance F (smallest index) and the kept signature
    // (strictly longest, earliest among equals)  INDEL:125-131.  New key = (len, F), staged in X.
    i64 lsum = 0;
    if constexpr (STAGED) {
        u64 nk[E];
#pragma unroll
        for (int e = 0; e < E; e++) {
            const int q = 64 * e + lane_id();
            u64 key = PAD_KEY;
            if (q < m) {
                const unsigned r = (unsigned)(A.K[q] >> 32);
                if (q == 0 || (unsigned)(A.K[q - 1] >> 32) != r) {
                    const int F = A.V2[q];
                    int best = F; i64 bl = A.X[F];
                    for (int t = q + 1; t < m && (unsigned)(A.K[t] >> 32) == r; t++) {
                        const int i2 = A.V2[t]; const i64 l2 = A.X[i2];
                        if (l2 > bl) { bl = l2; best = i2; }
                    }
                    A.V3[F] = best;
                    key = ((u64)bl << ib) | (u64)F;
                }
            }
            nk[e] = key;
        }
        __syncthreads();                                                    // the read-ordered keys and the lengths are dead
#pragma unroll
        for (int e = 0; e < E; e++) {
            const int q = 64 * e + lane_id();
            if (q < P) A.K[q] = nk[e];
            if (q < m) A.X[q] = st_a[e];                                    // positions by local index
        }
        __syncthreads();
        bitonic_sort<BLOCK, decltype(A.K), ME>(A.K, P, ib);    // == stable sort by length over first-appearance order (INDEL:136)
        // rank order: a-values -> K, lengths -> X, kept local index -> V1
        i64 av[E], lv[E]; int cv[E];
#pragma unroll
        for (int e = 0; e < E; e++) {
            const int r = 64 * e + lane_id();
            av[e] = 0; lv[e] = 0; cv[e] = 0;
            if (r < U) {
                const u64 key = A.K[r];
                cv[e] = A.V3[(int)(key & imask)];
                lv[e] = (i64)(key >> ib);
                av[e] = A.X[cv[e]];
                lsum += lv[e];
            }
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < E; e++) {
            const int r = 64 * e + lane_id();
            if (r < U) { A.K[r] = (u64)av[e]; A.X[r] = lv[e]; A.V1[r] = cv[e]; }
        }
        lsum = wave_sum_i64(lsum);
        __syncthreads();
    } else {
    for (int q = threadIdx.x; q < P; q += BLOCK) {
        u64 key = PAD_KEY;
        if (q < m) {
            const unsigned r = (unsigned)(A.K[q] >> 32);
            if (q == 0 || (unsigned)(A.K[q - 1] >> 32) != r) {
                const int F = A.V2[q];
                int best = F; i64 bl = B.b[s + F];
                for (int t = q + 1; t < m && (unsigned)(A.K[t] >> 32) == r; t++) {
                    const int i2 = A.V2[t]; const i64 l2 = B.b[s + i2];
                    if (l2 > bl) { bl = l2; best = i2; }
                }
                A.V3[F] = best;
                key = ((u64)bl << ib) | (u64)F;
            }
        }
        A.X[q] = (i64)key;
    }
    __syncthreads();
    for (int q = threadIdx.x; q < P; q += BLOCK) A.K[q] = (u64)A.X[q];
    __syncthreads();
    bitonic_sort<BLOCK, decltype(A.K), ME>(A.K, P, ib);    // == stable sort by length over first-appearance order (INDEL:136)

    // rank order: a-values -> K, lengths -> X, kept local index -> V1
    for (int r = threadIdx.x; r < U; r += BLOCK) {
        const u64 key = A.K[r];
        const int ch = A.V3[(int)(key & imask)];
        const i64 len = (i64)(key >> ib);
        A.K[r] = (u64)B.a[s + ch]; A.X[r] = len; A.V1[r] = ch;
        lsum += len;
    }
    lsum = block_sum_i64<BLOCK>(lsum, red);
    __syncthreads();
    }
    const double thr = sg.diff_ratio * div_by((double)lsum, (double)U, B.rcp_tab[U]);          // INDEL:138 (exact division by table reciprocal: div_by)

    // allele split on consecutive length gaps (INDEL:153-162): V2[a] = first rank of allele a
    int carry = 0;
    for (int base = 0; base < U; base += BLOCK) {
        const int r = base + threadIdx.x;
        int f = 0;
        if (r < U && r > 0) f = ((double)(A.X[r] - A.X[r - 1]) > thr) ? 1 : 0;
        int tot;
        const int inc = block_incl_scan<BLOCK>(f, &tot, red);
        if (r < U && (f || r == 0)) A.V2[carry + inc] = r;
        carry += tot;
    }
    const int nA = carry + 1;
    if (threadIdx.x == 0) A.V2[nA] = U;
    __syncthreads();

    // emission order: stable ascending by support among alleles with cnt >= minimum_support_reads
    // (INDEL:163-166): V3[a] = slot (-1: filtered), V4[a] = offset of its supports
    const int msr = sg.min_support_reads;
    int npass = 0;
    for (int a = threadIdx.x; a < nA; a += BLOCK) {
        const int cnt = A.V2[a + 1] - A.V2[a];
        int rank = -1, soff = 0;
        if (cnt >= msr) {
            rank = 0;
            for (int a2 = 0; a2 < nA; a2++) {
                const int c2 = A.V2[a2 + 1] - A.V2[a2];
                if (c2 >= msr && (c2 < cnt || (c2 == cnt && a2 < a))) { rank++; soff += c2; }
            }
            npass++;
        }
        A.V3[a] = rank; A.V4[a] = soff;
    }
    npass = (int)block_sum_i64<BLOCK>(npass, red);
    int aux0;
    if constexpr (STAGED) aux0 = A.V5[0]; else aux0 = B.aux[s];
    const int tbase = s;

    // per allele statistics: one wavefront per allele
    double rr = sg.remain_reads_ratio; if (rr > 1) rr = 1;                  // INDEL:46-47
    const int is_ins = sg.svtype == CSV_INS;
    int ncalls = 0, nsup = 0;
    const typename MemT<LDS>::ci64p PA = (typename MemT<LDS>::ci64p)A.K; const typename MemT<LDS>::ci64p PB = A.X;
    for (int a = threadIdx.x >> 6; a < nA; a += BLOCK / 64) {
        const int rank = A.V3[a];
        if (rank < 0) continue;
        const int r0 = A.V2[a], n = A.V2[a + 1] - r0, soff = A.V4[a];
        i64 sp = 0, sl = 0;
        for (int i = lane_id(); i < n; i += 64) { sp += PA[r0 + i]; sl += PB[r0 + i]; }
        sp = wave_sum_i64(sp); sl = wave_sum_i64(sl);
        int keep = (int)(rr * (double)n); if (keep < 1) keep = 1;           // INDEL:169
        const double rcp_n = B.rcp_tab[n];
        const double pmean = div_by((double)sp, (double)n, rcp_n), lmean = div_by((double)sl, (double)n, rcp_n);
        double bp, siglen; i64 search;
        if (keep >= n) {
            // every member kept: mean of the kept values == mean of all (exact integer sums);
            // search_threshold = first member with the smallest |pos - mean| (INDEL:171-177)
            double bd = 1e300; int bi = 0x7fffffff;
            for (int i = lane_id(); i < n; i += 64) {
                const double d = fabs((double)PA[r0 + i] - pmean);
                if (d < bd) { bd = d; bi = i; }
            }
            for (int msk = 32; msk > 0; msk >>= 1) {
                const double od = shfl_xor_f64(bd, msk); const int oi = __shfl_xor(bi, msk);
                if (od < bd || (od == bd && oi < bi)) { bd = od; bi = oi; }
            }
            search = PA[r0 + bi];
            bp = pmean; siglen = lmean;
        } else {
            // keep the `keep` members closest to the mean; ties in allele order (stable sort on |x - mean|)
            i64 ks = 0, kl = 0, sr = 0;
            for (int i = lane_id(); i < n; i += 64) {
                const double dp = fabs((double)PA[r0 + i] - pmean), dl = fabs((double)PB[r0 + i] - lmean);
                int rp = 0, rl = 0;
                for (int t = 0; t < n; t++) {
                    const double tp = fabs((double)PA[r0 + t] - pmean), tl = fabs((double)PB[r0 + t] - lmean);
                    rp += (tp < dp) || (tp == dp && t < i);
                    rl += (tl < dl) || (tl == dl && t < i);
                }
                if (rp < keep) ks += PA[r0 + i];
                if (rl < keep) kl += PB[r0 + i];
                if (rp == 0) sr = PA[r0 + i];
            }
            ks = wave_sum_i64(ks); kl = wave_sum_i64(kl); sr = wave_sum_i64(sr);
            bp = (double)ks / (double)keep; siglen = (double)kl / (double)keep; search = sr;   // INDEL:176-177,187
        }
        int cip = 0, cil = 0;
        if (SMALLN) {                                                       // n <= 256: whole wavefront, no scratch
            if (CSV_ABL(20) || !cipos_fast(PA + r0, n, sp, B.cipk_tab, cip)) cip = cipos_wave(PA + r0, n, sp, B.sqrt_tab);      // INDEL:191
            if (CSV_ABL(20) || !cipos_fast(PB + r0, n, sl, B.cipk_tab, cil)) cil = cipos_wave(PB + r0, n, sl, B.sqrt_tab);      // INDEL:194
        } else if (lane_id() == 0) {                                          // V5 is free on the DEL/INS path: per-wavefront stack
            const typename MemT<LDS>::intp stk = A.V5 + (threadIdx.x >> 6) * CIPOS_STACK_INTS;
            cip = cipos_of(PA + r0, n, sp, B.sqrt_tab, stk);
            cil = cipos_of(PB + r0, n, sl, B.sqrt_tab, stk);
        }
        i64 pick = -1; int valid = 1;
        if (is_ins) {                                                       // INDEL:398-405
            const i64 want = (i64)siglen;
            valid = 0;
            for (int base = 0; base < n; base += 64) {
                const int i = base + lane_id();
                int sl;                                                      // len(seq) of the allele's i-th member
                if constexpr (STAGED) sl = A.V5[A.V1[r0 + (i < n ? i : 0)]]; else sl = B.aux[s + A.V1[r0 + (i < n ? i : 0)]];
                const int ok = (i < n) && ((i64)sl >= want);
                const u64 mk = __ballot(ok);
                if (mk) {
                    const int i0 = base + __ffsll((long long)mk) - 1;
                    pick = it.gsig0 + A.V1[r0 + i0];
                    bp = (double)PA[r0 + i0];
                    valid = 1;
                    break;
                }
            }
            search = (i64)bp;                                               // INDEL:415
        }
        // supports: the allele's kept signatures in allele order (INDEL:205, 416)
        for (int i = lane_id(); i < n; i += 64) { const int w = s + A.V1[r0 + i]; B.sup_tmp[s + soff + i] = w; }
        if (lane_id() == 0) {
            tmp_write(tmp_slot(B, it.j, tbase, rank), (i64)bp, (i64)siglen, search, pick, n, cip, cil, soff, valid, npass, aux0);
            if (valid) { ncalls++; nsup += n; }
        }
    }
    ncalls = (int)block_sum_i64<BLOCK>(ncalls, red);
    nsup = (int)block_sum_i64<BLOCK>(nsup, red);
    if (threadIdx.x == 0) item_done(B, it.j, ncalls, nsup);
}

// compacted write of the first-seen signatures of ranks [r0, r1) to sup_tmp[dst...]; one wavefront
template <bool LDS> __device__ __forceinline__ void write_first_seen(const DevBatch& B, const ArraysT<LDS>& A, int s, int r0, int r1, int dst)
{
    int run = 0;
    for (int base = r0; base < r1; base += 64) {
        const int r = base + lane_id();
        const int f = (r < r1) && (A.V1[r < r1 ? r : r0] < 0);
        const u64 mk = __ballot(f);
        if (f) { const int w = s + (A.V1[r] & 0x7fffffff); B.sup_tmp[dst + run + __popcll(mk & lanemask_lt())] = w; }
        run += __popcll(mk);
    }
}

// ---- DUP / INV / TRA: generate_dup_cluster (DUP:79-131), generate_semi_inv_cluster (INV:101-203),
//      generate_semi_tra_cluster (TRA:106-254)
template <int BLOCK, bool LDS, int ME = 4> __device__ void refine_pair(const DevBatch& B, const ItemCtx& it, const ArraysT<LDS>& A, i64* red, int* ired)
{
    const csv_segment& sg = B.seg[it.k];
    const int m = it.m, P = it.P, s = it.s, type = sg.svtype, ib = it.ib;
    const u64 imask = (1ull << ib) - 1ull;
    if (keys_out_of_range<BLOCK>(B, it, red)) { item_none(B, it.j); return; }
    const int U = sort_by_read<BLOCK, LDS, ME>(B, it, A, red);                   // V2 = (read id, index) order
    if (U < sg.read_count) { item_none(B, it.j); return; }                  // DUP:82-84, INV:106-109, TRA:128-129

    // stable sort by pos2 (DUP:86, INV:111, TRA:109): key = (pos2, local index)
    for (int i = threadIdx.x; i < P; i += BLOCK) {
        u64 key = PAD_KEY;
        if (i < m) {
            const i64 p2 = B.b[s + i];
            key = ((u64)p2 << ib) | (u64)i;
        }
        A.K[i] = key;
    }
    __syncthreads();
    bitonic_sort<BLOCK, decltype(A.K), ME>(A.K, P, ib);
    for (int r = threadIdx.x; r < m; r += BLOCK) {
        const u64 key = A.K[r];
        const int i = (int)(key & imask);
        A.K[r] = (u64)B.a[s + i]; A.X[r] = (i64)(key >> ib); A.V1[r] = i; A.V3[i] = r;
    }
    __syncthreads();
    // sub-clusters on pos2 gaps > bias (DUP:91, INV:125, TRA:117): V4[r] = sub of rank r
    const i64 bias = sg.max_cluster_bias;
    int carry = 0;
    for (int base = 0; base < m; base += BLOCK) {
        const int r = base + threadIdx.x;
        int f = 0;
        if (r < m && r > 0) f = (A.X[r] - A.X[r - 1] > bias) ? 1 : 0;
        int tot;
        const int inc = block_incl_scan<BLOCK>(f, &tot, red);
        if (r < m) A.V4[r] = carry + inc;
        carry += tot;
    }
    const int nsub = carry + 1;
    __syncthreads();
    // first-seen flag: rank r is the first occurrence of its read inside its sub-cluster.  The other
    // signatures of the same read are neighbours in V2 (read-id order).
    for (int q = threadIdx.x; q < m; q += BLOCK) {
        const int i = A.V2[q];
        const int r = A.V3[i], k = A.V4[r];
        const int id = B.rid[s + i];
        int first = 1;
        for (int t = q - 1; t >= 0 && first; t--) {
            const int i2 = A.V2[t];
            if (B.rid[s + i2] != id) break;
            const int r2 = A.V3[i2];
            if (A.V4[r2] == k && r2 < r) first = 0;
        }
        for (int t = q + 1; t < m && first; t++) {
            const int i2 = A.V2[t];
            if (B.rid[s + i2] != id) break;
            const int r2 = A.V3[i2];
            if (A.V4[r2] == k && r2 < r) first = 0;
        }
        if (first) A.V1[r] |= (int)0x80000000;
    }
    __syncthreads();
    // group starts: V2[k] = first rank of sub k (the read-id order is dead now)
    for (int r = threadIdx.x; r < m; r += BLOCK)
        if (r == 0 || A.V4[r] != A.V4[r - 1]) A.V2[A.V4[r]] = r;
    if (threadIdx.x == 0) A.V2[nsub] = m;
    __syncthreads();
    const typename MemT<LDS>::ci64p PA = (typename MemT<LDS>::ci64p)A.K; const typename MemT<LDS>::ci64p PB = A.X;
    // unique reads per sub-cluster -> V3[k]
    for (int k = threadIdx.x >> 6; k < nsub; k += BLOCK / 64) {
        const int r0 = A.V2[k], r1 = A.V2[k + 1];
        int u = 0;
        for (int r = r0 + lane_id(); r < r1; r += 64) u += (A.V1[r] < 0);
        u = wave_sum_i32(u);
        if (lane_id() == 0) A.V3[k] = u;
    }
    __syncthreads();

    if (type == CSV_TRA) {
        if ((threadIdx.x >> 6) != 0) return;                                // one wavefront finishes the item
        // sorted(temp, key=-unique) is stable: best = first maximum, second = first maximum of the rest (TRA:131)
        int bu = -1, bk = 0x7fffffff;
        for (int k = lane_id(); k < nsub; k += 64) { const int u = A.V3[k]; if (u > bu) { bu = u; bk = k; } }
        for (int msk = 32; msk > 0; msk >>= 1) {
            const int ou = __shfl_xor(bu, msk), ok = __shfl_xor(bk, msk);
            if (ou > bu || (ou == bu && ok < bk)) { bu = ou; bk = ok; }
        }
        int su = -1, sk = 0x7fffffff;
        for (int k = lane_id(); k < nsub; k += 64) { if (k == bk) continue; const int u = A.V3[k]; if (u > su) { su = u; sk = k; } }
        for (int msk = 32; msk > 0; msk >>= 1) {
            const int ou = __shfl_xor(su, msk), ok = __shfl_xor(sk, msk);
            if (ou > su || (ou == su && ok < sk)) { su = ou; sk = ok; }
        }
        const int tcode = B.aux[s] & 7;
        int emit0 = -1, emit1 = -1, ne = 0;
        if (tcode <= 3) {                                                   // TRA:154-155, 226-227
            if (nsub > 1 && (double)su >= 0.5 * (double)sg.read_count) {    // TRA:133
                if ((double)(bu + su) >= (double)m * sg.diff_ratio) { emit0 = bk; emit1 = sk; ne = 2; }       // TRA:134
            } else if ((double)bu >= (double)m * sg.diff_ratio) { emit0 = bk; ne = 1; }                      // TRA:211
        }
        const int tbase = s;
        int soff = 0;
        for (int q = 0; q < ne; q++) {
            const int k = q ? emit1 : emit0, r0 = A.V2[k], r1 = A.V2[k + 1], u = A.V3[k];
            i64 s1 = 0, s2 = 0;
            for (int r = r0 + lane_id(); r < r1; r += 64) { s1 += PA[r]; s2 += PB[r]; }
            s1 = wave_sum_i64(s1); s2 = wave_sum_i64(s2);
            int cnt = r1 - r0;
            if (k == 0) { s1 += PA[0]; s2 += PB[0]; cnt += 1; }             // element 0 is visited twice, TRA:114-124
            write_first_seen<LDS>(B, A, s, r0, r1, s + soff);
            if (lane_id() == 0)
                tmp_write(tmp_slot(B, it.j, tbase, q), (i64)((double)s1 / (double)cnt), (i64)((double)s2 / (double)cnt), 0, -1, u, 0, 0, soff, 1, ne, B.aux[s]);   // TRA:173, 175
            soff += u;
        }
        if (lane_id() == 0) item_done(B, it.j, ne, soff);
        return;
    }

    // DUP / INV: slots for sub-clusters with enough reads, in sub order: V4[k] = slot, V5[k] = support offset
    const int rc = sg.read_count;
    int carry_slot = 0, carry_sup = 0;
    for (int base = 0; base < nsub; base += BLOCK) {
        const int k = base + threadIdx.x;
        int pass = 0, u = 0;
        if (k < nsub) {
            u = A.V3[k];
            const int n = A.V2[k + 1] - A.V2[k];
            pass = (u >= rc) && (type == CSV_DUP || n >= rc);               // DUP:96-98; INV:126,132
        }
        int tot, tots;
        const int inc = block_incl_scan<BLOCK>(pass, &tot, red);
        const int incs = block_incl_scan<BLOCK>(pass ? u : 0, &tots, red);
        if (k < nsub) { A.V4[k] = pass ? carry_slot + inc - 1 : -1; A.V5[k] = carry_sup + incs - (pass ? u : 0); }
        carry_slot += tot; carry_sup += tots;
    }
    const int nslots = carry_slot;
    const int tbase = s, aux0 = B.aux[s];
    int ncalls = 0, nsup = 0;
    __syncthreads();                                                        // V4 / V5 of sub k were written by thread k
    for (int k = threadIdx.x >> 6; k < nsub; k += BLOCK / 64) {
        const int slot = A.V4[k];
        if (slot < 0) continue;
        const int r0 = A.V2[k], r1 = A.V2[k + 1], n = r1 - r0, u = A.V3[k];
        i64 bp1, bp2;
        if (type == CSV_DUP) {
            const int lo = (int)((double)n * 0.4), hi = (int)((double)n * 0.6);         // DUP:99-100
            if (lo == hi) { bp1 = PA[r0 + lo]; bp2 = PB[r0 + lo]; }
            else {
                i64 s1 = 0, s2 = 0;
                for (int r = r0 + lo + lane_id(); r < r0 + hi; r += 64) { s1 += PA[r]; s2 += PB[r]; }
                s1 = wave_sum_i64(s1); s2 = wave_sum_i64(s2);
                bp1 = (i64)((double)s1 / (double)(hi - lo));                            // DUP:108-109
                bp2 = (i64)((double)s2 / (double)(hi - lo));
            }
        } else {
            i64 s1 = 0, s2 = 0;
            for (int r = r0 + lane_id(); r < r1; r += 64) { s1 += PA[r]; s2 += PB[r]; }
            s1 = wave_sum_i64(s1); s2 = wave_sum_i64(s2);
            bp1 = (i64)rint((double)s1 / (double)n);                                    // INV:129-130 (round half even)
            bp2 = (i64)rint((double)s2 / (double)n);
        }
        const i64 d = bp2 - bp1;
        const int valid = (d >= sg.sv_size) && (d <= sg.max_size || sg.max_size == -1); // DUP:112; INV:132-134
        write_first_seen<LDS>(B, A, s, r0, r1, s + A.V5[k]);
        if (lane_id() == 0) {
            tmp_write(tmp_slot(B, it.j, tbase, slot), bp1, bp2, 0, -1, u, 0, 0, A.V5[k], valid, nslots, aux0);
            if (valid) { ncalls++; nsup += u; }
        }
    }
    ncalls = (int)block_sum_i64<BLOCK>(ncalls, red);
    nsup = (int)block_sum_i64<BLOCK>(nsup, red);
    if (threadIdx.x == 0) item_done(B, it.j, ncalls, nsup);
}

// The arrays are reached through FLAT pointers (they may also live in global scratch).  hipcc folds
// `K[q - 1]`-style neighbour reads into `flat_load ... offset:8` off a pointer to the element BEFORE
// the one read; the hardware picks the aperture from that base, so an array at LDS offset 0 would send
// the q = 0 iteration's base below the LDS aperture and fault.  LDS_LEAD keeps every base inside it.
constexpr int LDS_LEAD = 64;
template <int CAP> constexpr int refine_lds_bytes() { return LDS_LEAD + (CAP + ARR_PAD) * (8 + 8 + 4 * 5) + 64; }

// `big` selects the work list; items whose size is outside (m_lo, m_hi] are left to another launch
// (small list: DEL/INS go to k_refine_indel_wave; big list: a one-wavefront mid tier and the workgroup tier).
#ifndef CSV_RF_WAVES
#define CSV_RF_WAVES 4
#endif
// BIG (which list) is a template parameter: the small-list instantiation handles DUP / INV / TRA only (DEL / INS of up to 64
// signatures belong to k_refine_indel_wave) and so carries neither the code nor the registers of refine_indel
template <int BLOCK, int CAP, bool BIG> __global__ __launch_bounds__(BLOCK, (BLOCK == 64 ? CSV_RF_WAVES : 1)) void k_refine(DevBatch B, int m_lo, int m_hi)
{
    constexpr int big = BIG ? 1 : 0;
    // 64-key registers of the one-wavefront sorts: the tier's capacity.  (r05, rejected with a measurement: a second capacity - a
    // <64,128> launch at five wavefronts per SIMD for 65 .. 128 signatures next to <64,256> for the rest - made the tier 75 us
    // instead of 60 on the 90x genome and 36 instead of 19 on the simulation beds: two kernels in a row each last as long as
    // their slowest cluster.)
    constexpr int ME_ = (BLOCK == 64 && CAP <= 256) ? (CAP + 63) / 64 : 4;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    CSV_LDS char* smem = (CSV_LDS char*)smem_raw + LDS_LEAD;
    constexpr int N = CAP + ARR_PAD;
    ArraysT<true> L;
    L.K = (CSV_LDS u64*)smem; L.X = (CSV_LDS i64*)(smem + 8 * N);
    L.V1 = (CSV_LDS int*)(smem + 16 * N); L.V2 = L.V1 + N; L.V3 = L.V2 + N; L.V4 = L.V3 + N; L.V5 = L.V4 + N;
    i64* red = (i64*)(smem_raw + LDS_LEAD + 36 * N);
    int* ired = (int*)(red + 6);
    const int n = big ? B.cnt->n_items_big : (B.cnt->n_items - B.cnt->n_items_big - B.cnt->n_items_tiny);
    if constexpr (BIG && BLOCK > 64) { if (B.cnt->n_items_huge == 0) return; }        // (nothing above MID_CAP signatures in this batch)
    // (r04, rejected: "tier by occupancy" - when a batch has fewer clusters of 65 .. 256 signatures than two per CU, hand each
    // to a whole 256-thread workgroup of the tier below instead of one wavefront: 29 us against 23 for the ~100 such clusters of
    // the simulation beds.  The LDS network's 2 x 36 barriers and the serial np.std replay cost more than three more wavefronts
    // save.  What pays is not waiting for the tiers one after the other: see the second phase below.)
    const int lo = m_lo;
    // (one-wavefront tiers: the list entry and the item record of the NEXT cluster are fetched while this one is refined - two
    // dependent round trips off the head of every cluster.  Not in the workgroup tier: 10.3 -> 12.3 us on the 90x workload.)
    constexpr bool AHEAD = BLOCK == 64;
    int j_nx = 0; int4 rec_nx = make_int4(0, 0, 0, 0);
    if (AHEAD && (int)blockIdx.x < n) { j_nx = big ? B.list_big[blockIdx.x] : B.list_small[blockIdx.x].x; rec_nx = B.item_rec[j_nx]; }
    for (int q = blockIdx.x; q < n; q += gridDim.x) {
        ItemCtx it;
        int4 rec;
        if constexpr (AHEAD) {
            it.j = j_nx; rec = rec_nx;
            if (q + (int)gridDim.x < n) { j_nx = big ? B.list_big[q + gridDim.x] : B.list_small[q + gridDim.x].x; rec_nx = B.item_rec[j_nx]; }
        } else {
            it.j = big ? B.list_big[q] : B.list_small[q].x;
            rec = B.item_rec[it.j];
        }
        it.cid = rec.x; it.k = rec.y; it.s = rec.z; it.m = rec.w;
        if (it.m <= lo || it.m > m_hi) continue;
        if (!big) { const int ty = B.seg[it.k].svtype; if (ty == CSV_DEL || ty == CSV_INS) continue; }
        it.gsig0 = B.seg[it.k].sig_begin + ((i64)it.s - B.woff[it.k]);
        int P = 1;
        while (P < it.m) P <<= 1;
        it.P = P;
        it.ib = P <= (1 << IDX_BITS) ? IDX_BITS : IDX_BITS_HUGE;
        __syncthreads();                                  // LDS arrays of the previous item are dead
        if (it.m > MAX_CLUSTER) {
            if (threadIdx.x == 0) atomicOr(&B.cnt->error, ERR_CLUSTER_TOO_BIG);
            item_none(B, it.j);
            continue;
        }
        const int t = B.seg[it.k].svtype;
        const bool indel = t == CSV_DEL || t == CSV_INS;
        if (P <= CAP) {
            if constexpr (BIG) { if (indel) { refine_indel<BLOCK, true, (CAP <= 256), (BLOCK == 64 && CAP <= 256 && !CSV_ABL(24)), ME_>(B, it, L, red, ired); continue; } }
            refine_pair<BLOCK, true, ME_>(B, it, L, red, ired);
        } else if constexpr (CAP <= 256) {
            // the one-wavefront tiers are only launched for m <= CAP; keeping the global-scratch path (and its
            // serial np.std routine with a private stack) out of them keeps these kernels free of scratch
            if (threadIdx.x == 0) atomicOr(&B.cnt->error, ERR_CLUSTER_TOO_BIG);
            item_none(B, it.j);
        } else {
            // cluster larger than the LDS tier: its own [2s, 2s + P) slice of the global scratch
            ArraysT<false> G;
            const i64 o = 2ll * it.s;
            G.K = B.sc_k + o; G.X = B.sc_x + o; G.V1 = B.sc_v1 + o; G.V2 = B.sc_v2 + o; G.V3 = B.sc_v3 + o;
            G.V4 = B.sc_v4 + o; G.V5 = B.sc_v5 + o;
            if (indel) refine_indel<BLOCK, false, false>(B, it, G, red, ired);
            else refine_pair<BLOCK, false>(B, it, G, red, ired);
        }
    }
    // Second phase of the one-wavefront tier for 65 .. 256 signatures (pair_too: the launch that also stands in for
    // k_refine<64,64>): the DUP / INV / TRA clusters of at most 64 signatures, dealt round-robin over the workgroups where the
    // first phase stopped.  The two kernels used to run one after the other, each lasting as long as its slowest cluster (five
    // simulation beds: 11.5 + 23 us with ~100 and ~1500 busy wavefronts); as ONE grid the long clusters start first and the
    // short ones fill the rest of the chip beside them.
    if constexpr (BIG && BLOCK == 64) {
        if (m_hi < 0) return;                             // (never: keeps the parameter list of the instantiations alike)
        if (!B.pair_in_mid) return;
        const int ns = B.cnt->n_items - B.cnt->n_items_big - B.cnt->n_items_tiny;
        int q0 = (int)blockIdx.x - n % (int)gridDim.x;
        if (q0 < 0) q0 += gridDim.x;
        for (int q = q0; q < ns; q += gridDim.x) {
            ItemCtx it;
            const int4 e = B.list_small[q];               // {item, segment | svtype << 24, first w, size}
            const int ty = e.y >> 24;
            if (ty == CSV_DEL || ty == CSV_INS || e.w > 64) continue;
            it.j = e.x; it.k = e.y & 0xffffff; it.s = e.z; it.m = e.w;
            it.cid = B.item_rec[it.j].x;
            it.gsig0 = B.seg[it.k].sig_begin + ((i64)it.s - B.woff[it.k]);
            int P = 1;
            while (P < it.m) P <<= 1;
            it.P = P; it.ib = IDX_BITS;
            __syncthreads();
            refine_pair<BLOCK, true, ME_>(B, it, L, red, ired);
        }
    }
}

// ------------------------------------------------------------------------------------ refine: INDEL wavefront tiers
// DEL/INS clusters of m <= 64 signatures, everything in registers: one signature per lane, cross-lane traffic
// is v_readlane (uniform index), ds_permute / ds_bpermute and DPP; no LDS arrays, no barriers.  Same semantics
// as refine_indel above (INDEL:110-219, 319-432).  The kernel is written for a "sub-wave" of SW lanes:
//   SW = 32: TWO clusters of m <= 32 per wavefront (lanes 0-31 and 32-63) — the common case, halves the
//            instruction count per cluster;  SW = 64: one cluster of 32 < m <= 64 per wavefront.
// Scalars of a cluster (segment flags, sizes) are per-lane values that are uniform inside a sub-wave; every
// cross-lane operation is executed by all 64 lanes in wave-uniform control flow.
__device__ __forceinline__ i64 readlane_i64(i64 v, int l)
{
    const int lo = __builtin_amdgcn_readlane((int)(v & 0xffffffffll), l);
    const int hi = __builtin_amdgcn_readlane((int)(v >> 32), l);
    return ((i64)hi << 32) | (unsigned)lo;
}
__device__ __forceinline__ double shfl_f64(double v, int src) { return __longlong_as_double(shfl_i64(__double_as_longlong(v), src)); }
__device__ __forceinline__ i64 permute_i64(int dest_lane, i64 v)      // lane i sends v to lane dest_lane (a bijection)
{
    const int lo = __builtin_amdgcn_ds_permute(dest_lane << 2, (int)(v & 0xffffffffll));
    const int hi = __builtin_amdgcn_ds_permute(dest_lane << 2, (int)(v >> 32));
    return ((i64)hi << 32) | (unsigned)lo;
}
// value of sub-lane t of the caller's own sub-wave (t wave-uniform; g = lane / SW, the caller's sub-wave)
template <int SW> __device__ __forceinline__ int sub_rl(int x, int t, int g)
{
    if (SW == 64) return __builtin_amdgcn_readlane(x, t);
    if (SW == 32) {
        const int lo = __builtin_amdgcn_readlane(x, t), up = __builtin_amdgcn_readlane(x, 32 + t);
        return g ? up : lo;
    }
    const int v0 = __builtin_amdgcn_readlane(x, t), v1 = __builtin_amdgcn_readlane(x, 16 + t);
    const int v2 = __builtin_amdgcn_readlane(x, 32 + t), v3 = __builtin_amdgcn_readlane(x, 48 + t);
    return (g & 2) ? ((g & 1) ? v3 : v2) : ((g & 1) ? v1 : v0);
}
template <int SW> __device__ __forceinline__ i64 sub_rl64(i64 x, int t, int g)
{
    const int lo = sub_rl<SW>((int)(x & 0xffffffffll), t, g), up = sub_rl<SW>((int)(x >> 32), t, g);
    return ((i64)up << 32) | (unsigned)lo;
}
// ballot restricted to the caller's sub-wave, in sub-lane bit positions
template <int SW> __device__ __forceinline__ u64 sub_ballot(bool p, int g)
{
    const u64 m = __ballot(p);
    if (SW == 64) return m;
    return (m >> (g * SW)) & ((1ull << (SW & 63)) - 1ull);
}
// inclusive scans inside a sub-wave: the DPP network simply stops early (16 lanes = one DPP row: row_shr only;
// 32 lanes: before row_bcast:31)
template <int SW> __device__ __forceinline__ i64 sub_scan_i64(i64 v)
{
    v = dpp_add_i64<0x111, 0xf>(v);
    v = dpp_add_i64<0x112, 0xf>(v);
    v = dpp_add_i64<0x114, 0xf>(v);
    v = dpp_add_i64<0x118, 0xf>(v);
    if (SW >= 32) v = dpp_add_i64<0x142, 0xa>(v);
    if (SW == 64) v = dpp_add_i64<0x143, 0xc>(v);
    return v;
}
template <int SW> __device__ __forceinline__ int sub_scan_i32(int v)
{
    v += dpp_i32<0x111, 0xf>(0, v);
    v += dpp_i32<0x112, 0xf>(0, v);
    v += dpp_i32<0x114, 0xf>(0, v);
    v += dpp_i32<0x118, 0xf>(0, v);
    if (SW >= 32) v += dpp_i32<0x142, 0xa>(0, v);
    if (SW == 64) v += dpp_i32<0x143, 0xc>(0, v);
    return v;
}

// ds_bpermute with an immediate byte offset on top of the address register (lane addr4 / 4 + OFF / 4, mod 64).  The builtin has
// no offset operand and the compiler does not fold an added constant into the field; the wait is in the block because the
// compiler's counter bookkeeping does not look inside it.
template <int OFF> __device__ __forceinline__ int bperm_off(int addr4, int v)
{
    int r;
    asm volatile("ds_bpermute_b32 %0, %1, %2 offset:%3\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr4), "v"(v), "n"(OFF));
    return r;
}
template <int OFF> __device__ __forceinline__ double bperm_off_f64(int addr4, double v)
{
    const i64 b = __double_as_longlong(v);
    const int lo = bperm_off<OFF>(addr4, (int)(b & 0xffffffffll)), hi = bperm_off<OFF>(addr4, (int)(b >> 32));
    return __longlong_as_double(((i64)hi << 32) | (unsigned)lo);
}
// ds_bpermute with a ready byte address (lane << 2): no per-call index arithmetic
__device__ __forceinline__ double bperm_f64(int addr4, double v)
{
    const i64 b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_ds_bpermute(addr4, (int)(b & 0xffffffffll)), hi = __builtin_amdgcn_ds_bpermute(addr4, (int)(b >> 32));
    return __longlong_as_double(((i64)hi << 32) | (unsigned)lo);
}
// value of the lane to the right (lane 63 gets 0): DPP wave_shl:1
__device__ __forceinline__ double wave_shl1_f64(double v) { return __longlong_as_double(dpp_i64<0x130, 0xf>(0, __double_as_longlong(v))); }

// numpy's pairwise sum of two per-lane series (sq1, sq2) over the allele [r0, r0 + n) (absolute lanes), n <= 64,
// valid on the lane with i == 0 (i = lane - r0).  8 strided accumulators on lanes i < 8, the fixed combine
// tree, then the sequential tail (the tail values are fetched once and then walk down to the head lane with
// wave_shl:1).  rows / tail are wave-uniform upper bounds of n / 8 and of the tail length.  Both series share
// the cross-lane addresses and predicates.
__device__ __forceinline__ void np_sum_allele2(double sq1, double sq2, int n, int i, int rows, int tail, double& out1, double& out2)
{
    const int lane = lane_id();
    const int nfull = n - (n & 7);
    double acc1 = sq1, acc2 = sq2;
    // (byte addresses of other lanes: the lane's own address plus a constant - the immediate offset of ds_bpermute - or plus a
    // wave-uniform step; the hardware ignores the bits above the lane.  Spelled ((lane + k) & 63) << 2 they were loop invariants
    // that the compiler computed in front of the kernel's unit loop and kept in VGPRs for its whole life - for a path that one
    // unit in a thousand takes.)
    const int l4 = lane << 2;
    for (int t = 1; t < rows; t++) {
        const int addr = l4 + 32 * t;
        const double v1 = bperm_f64(addr, sq1), v2 = bperm_f64(addr, sq2);
        if (i < 8 && i + 8 * t < nfull) { acc1 += v1; acc2 += v2; }
    }
    const double p1 = acc1 + bperm_off_f64<4>(l4, acc1), q1 = acc2 + bperm_off_f64<4>(l4, acc2);
    const double p2 = p1 + bperm_off_f64<8>(l4, p1), q2 = q1 + bperm_off_f64<8>(l4, q1);
    const double p3 = p2 + bperm_off_f64<16>(l4, p2), q3 = q2 + bperm_off_f64<16>(l4, q2);
    double res1 = (n >= 8) ? p3 : 0.0, res2 = (n >= 8) ? q3 : 0.0;
    const int start = (n >= 8) ? nfull : 0;
    const int adt = l4 + 4 * start;                            // lane L now holds element start + (L - r0) of its allele
    double t1 = bperm_f64(adt, sq1), t2 = bperm_f64(adt, sq2);
    for (int e = 0; e < tail; e++) {
        if (start + e < n) { res1 += t1; res2 += t2; }
        t1 = wave_shl1_f64(t1); t2 = wave_shl1_f64(t2);
    }
    out1 = res1; out2 = res2;
}

// Sorting network inside sub-waves of SW = 16 / 32 lanes (every sub-wave of the wavefront at once, all ascending): the
// bitonic network's 10 / 15 compare-exchanges on one-word keys, three vector instructions each.  Which side keeps the
// smaller key is a compile-time lane mask (inverse_ballot hands it to the select as an SGPR pair).  Partners 1, 2 and 8 lanes
// away sit in the same DPP row: v_min_u32_dpp, v_max_u32_dpp, v_cndmask.  Partners 4 and 16 lanes away come through a
// ds_swizzle with an xor mask: swizzle, compare, select, and "keep mine" = the compare's mask XNOR the constant (scalar).
// (r03 ranked by counting: a ds_swizzle broadcast, a compare and an add for each of the up to 32 sub-lanes = 96 vector
// instructions per pair of clusters; this is 45.)
constexpr u64 cx_keepmin_mask(int SW, int KK, int J)
{
    u64 m = 0;
    for (int l = 0; l < 64; l++) { const bool lower = (l & J) == 0, asc = ((l & (SW - 1)) & KK) == 0; if (lower == asc) m |= 1ull << l; }
    return m;
}
template <int SW, int KK, int J> __device__ __forceinline__ unsigned cx_step(unsigned k)
{
    constexpr u64 KEEPMIN = cx_keepmin_mask(SW, KK, J);
    if constexpr (J == 1 || J == 2 || J == 8) {
        // partner inside the DPP row: hipcc folds the move into v_min_u32_dpp / v_max_u32_dpp, the select takes the constant mask
        constexpr int CTRL = J == 1 ? 0xB1 : (J == 2 ? 0x4E : 0x128);                                      // quad_perm [1,0,3,2] / [2,3,0,1] / row_ror:8
        const unsigned o = (unsigned)__builtin_amdgcn_update_dpp((int)k, (int)k, CTRL, 0xf, 0xf, true);
        const unsigned mn = k < o ? k : o, mx = k < o ? o : k;
        return __builtin_amdgcn_inverse_ballot_w64(KEEPMIN) ? mn : mx;
    } else if constexpr (J == 32) {
        // the other half of the wavefront: ds_bpermute with an immediate offset of 32 lanes on the lane's own address (no address
        // arithmetic, nothing for the compiler to hoist into a long-lived register)
        const unsigned o = (unsigned)bperm_off<128>(lane_id() << 2, (int)k);
        const u64 lt = __ballot(k < o);
        return __builtin_amdgcn_inverse_ballot_w64(~(lt ^ KEEPMIN)) ? k : o;
    } else {
        const unsigned o = (unsigned)__builtin_amdgcn_ds_swizzle((int)k, 0x1f | (J << 10));               // bit mode: lane ^ J (J = 4, 16)
        const u64 lt = __ballot(k < o);
        return __builtin_amdgcn_inverse_ballot_w64(~(lt ^ KEEPMIN)) ? k : o;                                // (keys are distinct)
    }
}
template <int SW> __device__ __forceinline__ unsigned sort_sub(unsigned k)
{
    static_assert(SW == 16 || SW == 32 || SW == 64, "sub-wave width");
    k = cx_step<SW, 2, 1>(k);
    k = cx_step<SW, 4, 2>(k); k = cx_step<SW, 4, 1>(k);
    k = cx_step<SW, 8, 4>(k); k = cx_step<SW, 8, 2>(k); k = cx_step<SW, 8, 1>(k);
    k = cx_step<SW, 16, 8>(k); k = cx_step<SW, 16, 4>(k); k = cx_step<SW, 16, 2>(k); k = cx_step<SW, 16, 1>(k);
    if (SW >= 32) { k = cx_step<SW, 32, 16>(k); k = cx_step<SW, 32, 8>(k); k = cx_step<SW, 32, 4>(k); k = cx_step<SW, 32, 2>(k); k = cx_step<SW, 32, 1>(k); }
    // (one cluster per wavefront: the select masks are compile-time constants here too.  The generic 64-lane network -
    // bitonic_wave32 - derives them from the lane id at run time, and the compiler kept a dozen of them in SGPR pairs across
    // the whole kernel: most of its 22 SGPR spills)
    if (SW == 64) { k = cx_step<SW, 64, 32>(k); k = cx_step<SW, 64, 16>(k); k = cx_step<SW, 64, 8>(k); k = cx_step<SW, 64, 4>(k); k = cx_step<SW, 64, 2>(k); k = cx_step<SW, 64, 1>(k); }
    return k;
}

// What a unit reads from memory, in two rounds: the list entries of its clusters (one per sub-wave), then - addresses
// known from the entry alone - the rows and the segment scalars together.  (The first version walked list -> item
// record -> segment -> rows: four dependent round trips per unit, and with ~10 k units on ~5 k resident wavefronts the
// kernel's time is a small multiple of one unit's latency.)
// Positions and lengths stay in the width of the caller's columns (crd_t): a batch of int32 columns (CSV_IN_SIG_I32, what
// SigStore.pinned() and bench.py send) is refined in 32-bit registers - every cross-lane move of a coordinate is one
// ds_permute / ds_bpermute / DPP instead of two, every select one v_cndmask (r03 widened at the load: 80 VGPRs + spills).
template <bool NARROW> using crd_t = typename std::conditional<NARROW, int, i64>::type;
template <bool NARROW> struct UnitIn {
    int4 e;                       // list entry {item, segment | svtype << 24, first w, size}; svtype -1: no item (per lane, uniform inside a sub-wave)
    crd_t<NARROW> a, b;
    int  rid, aux;
};
// entry of the sub-wave's cluster: unit p of a list whose units hold 64 / sw clusters (sw = 1 << sw_log2, wave-uniform)
__device__ __forceinline__ int4 unit_entry(const int4* list, int p, int nlist, int sw_log2)
{
    const int q = (p << (6 - sw_log2)) + (lane_id() >> sw_log2);
    return (p >= 0 && q < nlist) ? list[q] : make_int4(0, (int)0xff000000u, 0, 0);
}
// rows of the entry's cluster, one signature per lane of the sub-wave; m_lo < size <= sw or the lanes stay empty
template <bool NARROW> __device__ __forceinline__ void unit_rows(const DevBatch& B, const int4 e, int sw_log2, int m_lo, UnitIn<NARROW>& U)
{
    const int sl = lane_id() & ((1 << sw_log2) - 1), type = e.y >> 24, s = e.z, m = e.w;
    const bool in = (type == CSV_DEL || type == CSV_INS) && m > m_lo && m <= (1 << sw_log2) && sl < m;
    U.e = e;
    if constexpr (NARROW) { U.a = in ? B.a.p32[s + sl] : 0; U.b = in ? B.b.p32[s + sl] : 0; }
    else { U.a = in ? B.a.p64[s + sl] : 0; U.b = in ? B.b.p64[s + sl] : 0; }
    U.rid = in ? B.rid[s + sl] : -1 - lane_id();
    U.aux = (in && type == CSV_INS) ? B.aux[s + sl] : 0;
}

// cross-lane moves of a coordinate in either width; the ds_(b)permute forms take a READY byte address (lane << 2): the
// handful of source lanes a unit keeps going back to (rank 0 of its sub-wave, the first and last rank of its allele) are
// shifted once, not at every use (HIP's __shfl recomputes (src & 63) + (self & ~63) << 2 each time)
__device__ __forceinline__ int bperm(int addr4, int v) { return __builtin_amdgcn_ds_bpermute(addr4, v); }
__device__ __forceinline__ i64 bperm(int addr4, i64 v)
{
    const int lo = __builtin_amdgcn_ds_bpermute(addr4, (int)(v & 0xffffffffll)), hi = __builtin_amdgcn_ds_bpermute(addr4, (int)(v >> 32));
    return ((i64)hi << 32) | (unsigned)lo;
}
__device__ __forceinline__ int fperm(int addr4, int v) { return __builtin_amdgcn_ds_permute(addr4, v); }      // lane i SENDS v to lane addr4 >> 2 (a bijection)
__device__ __forceinline__ i64 fperm(int addr4, i64 v)
{
    const int lo = __builtin_amdgcn_ds_permute(addr4, (int)(v & 0xffffffffll)), hi = __builtin_amdgcn_ds_permute(addr4, (int)(v >> 32));
    return ((i64)hi << 32) | (unsigned)lo;
}
__device__ __forceinline__ int rdlane(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ i64 rdlane(i64 v, int l) { return readlane_i64(v, l); }
// value of the lane to the left (DPP wave_shr:1).  The move must stay an instruction of its own: left alone, hipcc folds it
// into the consumer - `v_subrev_u32_dpp v11, v14, v14 wave_shr:1 bound_ctrl:1` for len - lprev, the same register as both
// operands - and that instruction returned 0 in every lane on gfx950 (no allele was ever split; r04, found by
// test_narrow_input_columns).  The empty asm makes the moved value opaque to the DPP combiner.
__device__ __forceinline__ int shr1(int v) { int r = dpp_i32<0x138, 0xf>(0, v); asm volatile("" : "+v"(r)); return r; }
__device__ __forceinline__ i64 shr1(i64 v) { return wave_shr1_i64(v); }
template <int SW> __device__ __forceinline__ i64 sub_rlc(i64 x, int t, int g) { return sub_rl64<SW>(x, t, g); }
template <int SW> __device__ __forceinline__ int sub_rlc(int x, int t, int g) { return sub_rl<SW>(x, t, g); }
template <class T> struct crd_min;
template <> struct crd_min<int> { static constexpr int v = INT32_MIN; };
template <> struct crd_min<i64> { static constexpr i64 v = INT64_MIN; };

#ifndef CSV_IW_WAVES
#define CSV_IW_WAVES 6
#endif
// one unit of work: SW = 16 -> four clusters of m <= 16; SW = 32 -> the pair of items (2p, 2p + 1), each handled if m <= 32;
//                   SW = 64 -> the single item p, handled if 32 < m <= 64.
// LDS of one wavefront of k_refine_indel_wave: the tag table of the per-read de-duplication (below) and one mark byte per lane
constexpr int IW_TAG_BITS = 12;                     // 4096 one-byte slots per wavefront
struct IwLds { unsigned char tag[1 << IW_TAG_BITS]; unsigned char mark[64]; };
template <int SW, bool NARROW> __device__ __forceinline__ void indel_unit(const DevBatch& B, const UnitIn<NARROW>& U, CSV_LDS IwLds* L)
{
    typedef crd_t<NARROW> C;
    constexpr int NSUB = 64 / SW;                      // clusters per wavefront
    constexpr int MLO = SW == 64 ? 32 : 0;
    constexpr u64 SUBMASK = SW == 64 ? ~0ull : (1ull << (SW & 63)) - 1ull;
    const int lane = lane_id(), sl = lane & (SW - 1), hb = lane & ~(SW - 1), g = lane / SW;
    const int hb4 = hb << 2, last4 = (hb | (SW - 1)) << 2;                       // byte addresses of the sub-wave's first / last lane
    do {
        const int j = U.e.x, k = U.e.y & 0xffffff, type = U.e.y >> 24, s = U.e.z, m = U.e.w;
        // other types go to k_refine<64,64>, the other size class to the other instantiation
        const bool indel = type == CSV_DEL || type == CSV_INS;
        const bool act = indel && m > MLO && m <= SW;
        if (!__ballot(act)) break;
        if (CSV_ABL(6)) { if (act && sl == 0) item_done(B, j, 0, (int)(U.a + U.b + U.rid + U.aux)); break; }      // loads only
        // segment scalars: issued here, first needed after the de-duplication (the table is a few KB and cache resident)
        int rc = 0x7fffffff, msr = 0;
        double ratio = 0.0, rr0 = 1.0;
        if (act) {
            const csv_segment& sg = B.seg[k];
            rc = sg.read_count; msr = sg.min_support_reads; ratio = sg.diff_ratio; rr0 = sg.remain_reads_ratio;
        }
        const bool in = act && sl < m;
        const C a = U.a, b = U.b;
        const int rid = U.rid, aux = U.aux;
        // out-of-range length (negative, or beyond the sort keys of the other tiers): the cluster emits nothing
        bool bad;
        if constexpr (NARROW) bad = in && b < 0; else bad = in && (((u64)b) >> (63 - IDX_BITS)) != 0;
        const bool badk = sub_ballot<SW>(bad, g) != 0;
        if (badk && sl == 0) atomicOr(&B.seg_err[k], CSV_SEG_KEY_RANGE);
        const int mact = act ? m : 0;
        int mmax = __builtin_amdgcn_readlane(mact, 0);
        if (NSUB >= 2) mmax = max(mmax, __builtin_amdgcn_readlane(mact, SW));
        if (NSUB == 4) mmax = max(mmax, max(__builtin_amdgcn_readlane(mact, 32), __builtin_amdgcn_readlane(mact, 48)));

        // ---- per-read de-duplication (INDEL:125-131): first appearance F, kept signature = strictly longest
        int F = sl, ch = sl;
        C bl = b;
        // Which lanes hold a read id that some other lane of their cluster holds too?  A last-writer-wins table in LDS: every lane
        // writes its lane number at hash(read id) inside its cluster's part of the wavefront's 4096 one-byte slots and reads the
        // slot back; whoever does not find itself there lost to a lane with the same id (or, one time in a hundred, to a hash
        // collision) and marks the winner.  Lost or marked = "has a partner": a superset of the truth, and the exact loop below
        // compares the ids themselves.  ~12 vector instructions and four LDS operations, whatever the ids look like (r04: twelve
        // ballots over the id bits, 6 vector instructions each - 72 of a unit's ~680).
        bool dup_any = false;
        {
            constexpr int HB = IW_TAG_BITS - (NSUB == 4 ? 2 : NSUB == 2 ? 1 : 0);     // hash bits inside a cluster's part
            const unsigned h = (unsigned)rid ^ ((unsigned)rid >> HB) ^ ((unsigned)rid >> (2 * HB));
            const int slot = (int)(h & ((1u << HB) - 1u)) | (g << HB);
            L->mark[lane] = 0;
            if (in) L->tag[slot] = (unsigned char)lane;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int w = in ? (int)L->tag[slot] : lane;
            const bool lost = w != lane;
            if (lost) L->mark[w] = 1;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            dup_any = lost || (in && L->mark[lane] != 0);
            if (CSV_ABL(0)) dup_any = false;
        }
        // Reads with more than one signature in a cluster are common (a 30x cluster has one in ~40 % of the cases), so this is
        // not a rare path: walk only the lanes that have a partner (two to four per wavefront, in lane order = order of
        // appearance), each broadcast with plain v_readlane, instead of all mmax sub-lanes through the 4-way select.
        if (const u64 dm0 = __ballot(dup_any)) {
            if (dup_any) { F = -1; ch = -1; bl = crd_min<C>::v; }
            for (u64 dm = dm0; dm; dm &= dm - 1) {
                const int t = __ffsll((long long)dm) - 1;
                const int rt = __builtin_amdgcn_readlane(rid, t);
                const C bt = rdlane(b, t);
                if (dup_any && (t & ~(SW - 1)) == hb && rt == rid) {          // same cluster, same read (a folded-id false match fails here)
                    if (F < 0) F = t & (SW - 1);
                    if (bt > bl) { bl = bt; ch = t & (SW - 1); }
                }
            }
        }
        // remain_reads_ratio (INDEL:46-47, 169) is 1 unless the caller said otherwise: then every member of an allele is kept and
        // the ratio itself is never needed again - one flag instead of a double held (or a dependent load issued) mid-unit
        const bool rr_full = !(rr0 < 1.0);
        const bool rep = in && (F == sl);
        const u64 rm = sub_ballot<SW>(rep, g);
        const int U_ = __popcll(rm);
        const bool ok = act && !badk && U_ >= rc;                                // INDEL:133-134
        if (act && !ok && sl == 0) item_done(B, j, 0, 0);
        if (!__ballot(ok)) break;
        const int ch4 = (hb | (ch & (SW - 1))) << 2;
        const C pa = bperm(ch4, a);
        const int pax = bperm(ch4, aux);
        const int aux0 = bperm(hb4, aux);                                         // aux word of the cluster's first signature (k_emit: call_aux)

        // ---- stable sort of the kept signatures by length (INDEL:136): rank by (len, first appearance)
        int rank = 0, src_lane = -1;
        if (CSV_ABL(1)) rank = sl;
        else if constexpr (SW < 64) {
            if (!__ballot(rep && (bl >> 26) != 0)) {
                // every kept length fits 26 bits: (length, first appearance) packs into ONE word; the kept ones by that key, the
                // others behind them in lane order, through the sub-wave's sorting network; sorted position p then PULLS its row
                // from the lane its key came from
                const unsigned ks = sort_sub<SW>(rep ? (((unsigned)bl << 5) | (unsigned)sl) : (0x80000000u | (unsigned)sl));
                src_lane = hb | (int)(ks & (SW - 1));
            } else {
                for (int t = 0; t < mmax; t++) {
                    const C lt = sub_rlc<SW>(bl, t, g);
                    rank += ((rm >> t) & 1) && ((lt < bl) || (lt == bl && t < sl));
                }
            }
        } else if (!__ballot(rep && (bl >> 25) != 0)) {                  // same one-word key, 6 bits of first appearance
            // one cluster per wavefront: the keys (kept ones by (length, lane), the others behind them in lane order) go through
            // the 64-lane sorting network (21 compare-exchanges in DPP / bpermute, no scalar work), and sorted position p then
            // PULLS its row from the lane the key came from.  (A rank by counting - one v_readlane, compare and add per kept
            // signature - was 3 vector + 6 scalar instructions times up to 64.)
            const unsigned ks = sort_sub<64>(rep ? (((unsigned)bl << 6) | (unsigned)sl) : (0x80000000u | (unsigned)sl));
            src_lane = (int)(ks & 63u);
        } else {
            for (u64 mk = __ballot(rep); mk; mk &= mk - 1) {
                const int t = __ffsll((long long)mk) - 1;
                const C lt = rdlane(bl, t);
                rank += (lt < bl) || (lt == bl && t < sl);
            }
        }
        C pos, len; int chp, axp;
        if (src_lane >= 0) {                                                 // (wave-uniform: all lanes or none)
            const int s4 = src_lane << 2;
            pos = bperm(s4, pa); len = bperm(s4, bl);
            chp = bperm(s4, ch); axp = bperm(s4, pax);
        } else {
            const u64 sl_lt = (1ull << sl) - 1ull;
            const int dest4 = (hb | (rep ? rank : U_ + __popcll(~rm & sl_lt & SUBMASK))) << 2;
            pos = fperm(dest4, pa);
            len = fperm(dest4, bl);
            chp = fperm(dest4, ch);
            axp = fperm(dest4, pax);
        }
        const int r = sl;
        const bool live = ok && r < U_;

        // ---- sums.  A cluster whose positions and lengths all lie within 2^18 of its first member's (every ordinary one)
        // takes the FAST form: both deltas, biased to be non-negative, share ONE 64-bit scan (sums < 2^25 per half), and what
        // follows - allele sums, the member closest to the mean, the variances - works on these small exact integers.
        constexpr int DB = 18;
        const C bpos = bperm(hb4, pos), blen = bperm(hb4, len);                   // rank 0 of the sub-wave: the origin of the deltas
        int dp32, dl32;                                                           // the deltas (meaningful when `fast`)
        bool small;
        if constexpr (NARROW) {
            // (coordinates of an int32 column; a live length is >= 0 - `bad` - and the difference of two non-negative ints cannot wrap)
            const int dpi = (int)((unsigned)pos - (unsigned)bpos), dli = (int)((unsigned)len - (unsigned)blen);
            small = !live || ((pos | bpos) >= 0 && dpi > -(1 << DB) && dpi < (1 << DB) && dli > -(1 << DB) && dli < (1 << DB));
            dp32 = live ? dpi : 0; dl32 = live ? dli : 0;
        } else {
            const i64 dpi = pos - bpos, dli = len - blen;
            small = !live || ((u64)pos < (1ull << 31) && (u64)len < (1ull << 31) && dpi > -(1 << DB) && dpi < (1 << DB) && dli > -(1 << DB) && dli < (1 << DB));
            dp32 = live ? (int)dpi : 0; dl32 = live ? (int)dli : 0;
        }
        const bool fast = !__ballot(!small) && !CSV_ABL(15);
        // (the sums are needed as doubles only.  On the fast path they are n x origin + a small exact integer: with int32 columns
        // both factors and the product are exact in float64 - n <= 64, |origin| < 2^31 - so the double is built from two 32-bit
        // conversions, a multiply and an add instead of a 64-bit multiply-add and a 64-bit integer -> float64 conversion, which has
        // no instruction of its own; the result is the same number either way)
        i64 PK = 0;
        double d_lsum;
        if (fast) {
            const unsigned dpb = live ? (unsigned)(dp32 + (1 << DB)) : 0u, dlb = live ? (unsigned)(dl32 + (1 << DB)) : 0u;
            PK = sub_scan_i64<SW>((i64)(((u64)dlb << 32) | dpb));
            const int hi = (int)(unsigned)((u64)bperm(last4, PK) >> 32);                // sum of the biased deltas: < 2^25
            if constexpr (NARROW) d_lsum = (double)U_ * ((double)(int)blen - (double)(1 << DB)) + (double)hi;
            else d_lsum = (double)((i64)U_ * ((i64)blen - (1 << DB)) + (i64)hi);
        } else d_lsum = (double)bperm(last4, sub_scan_i64<SW>(live ? (i64)len : 0));
        // ---- allele split on consecutive length gaps (INDEL:138, 153-162)
        const double thr = ratio * div_by(d_lsum, (double)U_, B.rcp_tab[U_ & (SQRT_TAB - 1)]);
        const C lprev = shr1(len);
        bool f;
        if constexpr (NARROW) f = live && r > 0 && ((double)(int)((unsigned)len - (unsigned)lprev) > thr);
        else f = live && r > 0 && ((double)(len - lprev) > thr);
        const u64 fmask = __ballot(f);
        const u64 S = ((fmask >> hb) & SUBMASK) | 1ull;                           // allele start ranks (sub-lane positions)
        const u64 sl_le = (2ull << sl) - 1ull;
        const u64 below = S & sl_le, above = S & ~sl_le & SUBMASK;
        const int r0 = 63 - __clzll((long long)below);
        int r1 = above ? (__ffsll((long long)above) - 1) : U_;
        if (r1 > U_) r1 = U_;
        const int n = live ? r1 - r0 : 1, i = r - r0;

        const int e14 = (hb | ((r1 - 1) & (SW - 1))) << 2, e04 = (hb | ((r0 - 1) & (SW - 1))) << 2;     // last rank of the allele / of the one before
        // NB: every cross-lane op sits in wave-uniform control flow; only the selects are per lane
        double d_sp, d_sln;                                                       // the allele's position / length sums, as doubles (exact)
        int s1p = 0, s1l = 0;                                                     // fast: sums of the deltas over the allele
        if (fast) {
            const i64 k1 = bperm(e14, PK), k0 = bperm(e04, PK);
            const u64 seg = (u64)(k1 - (r0 > 0 ? k0 : 0));                        // (both halves ascend: no borrow across them)
            s1p = (int)(unsigned)(seg & 0xffffffffull) - n * (1 << DB);
            s1l = (int)(unsigned)(seg >> 32) - n * (1 << DB);
            if constexpr (NARROW) { d_sp = (double)n * (double)(int)bpos + (double)s1p; d_sln = (double)n * (double)(int)blen + (double)s1l; }
            else { d_sp = (double)((i64)n * (i64)bpos + s1p); d_sln = (double)((i64)n * (i64)blen + s1l); }
        } else {
            const i64 Ppos = sub_scan_i64<SW>(live ? (i64)pos : 0), Plen = sub_scan_i64<SW>(live ? (i64)len : 0);
            const i64 pp0 = bperm(e04, Ppos), pl0 = bperm(e04, Plen);
            d_sp = (double)(bperm(e14, Ppos) - (r0 > 0 ? pp0 : 0));
            d_sln = (double)(bperm(e14, Plen) - (r0 > 0 ? pl0 : 0));
        }

        // ---- emission order: stable ascending by support among alleles with n >= minimum_support_reads (INDEL:163-166)
        const bool pass = live && n >= msr;
        int erank = 0, soff = 0, npass = 0;
        const u64 starts = fmask | (NSUB == 4 ? 0x0001000100010001ull : NSUB == 2 ? 0x0000000100000001ull : 1ull);   // allele starts of every sub-wave, absolute lanes
        // (no allele split anywhere in the wavefront - three units in four on a 30x genome -: every cluster's one allele is
        // first in line, nothing in front of it: the walk over the allele starts has nothing to find)
        if (fmask == 0) npass = 1;
        else for (u64 mk = CSV_ABL(5) ? 0ull : starts; mk; mk &= mk - 1) {
            const int t = __ffsll((long long)mk) - 1;
            const int nt = __builtin_amdgcn_readlane(n, t);
            const int tl = t & (SW - 1);
            if ((t & ~(SW - 1)) == hb && nt >= msr) {
                npass++;
                if (nt < n || (nt == n && tl < r0)) { erank++; soff += nt; }
            }
        }

        // ---- statistics, all alleles at once
        int keep = n;
        if (__ballot(act && !rr_full)) {                                          // (a --remain_reads_ratio below 1: the ratio, now)
            double rr = 1.0;
            if (act) { rr = B.seg[k].remain_reads_ratio; if (rr > 1) rr = 1; }    // INDEL:46-47
            keep = (int)(rr * (double)n); if (keep < 1) keep = 1;                 // INDEL:169
        }
        const double rcp_n = B.rcp_tab[n & (SQRT_TAB - 1)];
        const double pmean = div_by(d_sp, (double)n, rcp_n), lmean = div_by(d_sln, (double)n, rcp_n);
        double bp = pmean, siglen = lmean;
        C search;
        if (CSV_ABL(3)) search = pos;
        else if (!__ballot(pass && keep < n) && fast) {
            // every member kept: search_threshold = first member with the smallest |pos - mean| (INDEL:171-177), in integers:
            // n pos - sum = n d - s1 exactly, and the reference's doubles |pos - fl(sum / n)| are exact differences too (both
            // operands are multiples of 2^-22 below 2^31, less than 2^19 apart), so they order like |n d - s1| except when two
            // members on OPPOSITE sides of the mean are exactly equidistant: then the side towards which the quotient was
            // rounded is the nearer one - the sign of the division's residual, which one fma gives exactly.
            const int nd = n * dp32 - s1p;                                      // |nd| < 2^25
            const unsigned an = (unsigned)(nd < 0 ? -nd : nd);
            const double resid = fma(-(double)n, pmean, d_sp);                    // sum - n * fl(sum / n): < 0 <=> the mean was rounded up
            const unsigned t = resid < 0.0 ? (nd < 0) : (resid > 0.0 ? (nd > 0) : 0);
            unsigned bk = ((2u * an + t) << 6) | (unsigned)r;                    // (distance, side, rank): one 32-bit minimum
            // segmented prefix minimum over the allele (the minimum of the whole allele arrives at its last rank): inside a DPP row
            // of 16 lanes with row_shr moves, then the row before through row_bcast:15 / row_bcast:31 - a lane whose allele began
            // before its row takes the (complete) value of that row's last lane, which belongs to the same allele.  All DPP: no
            // LDS round trip in a chain of five or six dependent steps (r04: ds_bpermute per step for the 32- and 64-lane units).
            {
                const int q = lane & 15;
#define CSV_MINSTEP(CTRL, D) { const unsigned o = (unsigned)dpp_i32<CTRL, 0xf>((int)bk, (int)bk); if (i >= D && q >= D && o < bk) bk = o; }
                CSV_MINSTEP(0x111, 1) CSV_MINSTEP(0x112, 2) CSV_MINSTEP(0x114, 4) CSV_MINSTEP(0x118, 8)
#undef CSV_MINSTEP
                if (SW >= 32) { const unsigned o = (unsigned)dpp_i32<0x142, 0xa>((int)bk, (int)bk); if ((lane & 16) && i > q && o < bk) bk = o; }
                if (SW == 64) { const unsigned o = (unsigned)dpp_i32<0x143, 0xc>((int)bk, (int)bk); if ((lane & 32) && i > (lane & 31) && o < bk) bk = o; }
            }
            search = bperm((hb | (bperm(e14, (int)bk) & (SW - 1))) << 2, pos);
        } else if (!__ballot(pass && keep < n)) {
            // the same on the doubles themselves (clusters that span more than 2^18 bases)
            double bd = fabs((double)pos - pmean); int bi = r;
            const int l4 = lane << 2;                        // (lane - d through ds_bpermute's immediate offset: see the integer form above)
#define CSV_MIND(D) if (D < SW) { const double od = bperm_off_f64<256 - 4 * D>(l4, bd); const int oi = bperm_off<256 - 4 * D>(l4, bi); \
                                  if (r - D >= r0 && (od < bd || (od == bd && oi < bi))) { bd = od; bi = oi; } }
            CSV_MIND(1) CSV_MIND(2) CSV_MIND(4) CSV_MIND(8) CSV_MIND(16) CSV_MIND(32)
#undef CSV_MIND
            search = bperm((hb | (bperm(e14, bi) & (SW - 1))) << 2, pos);
        } else {
            // keep the `keep` members closest to the mean, ties in allele order (INDEL:171-176, 182-187)
            const double dp = fabs((double)pos - pmean), dl = fabs((double)len - lmean);
            int rp = 0, rl = 0;
            for (int t = 0; t < mmax; t++) {
                const int r0t = sub_rl<SW>(r0, t, g);
                const double tp = __longlong_as_double(sub_rl64<SW>(__double_as_longlong(dp), t, g));
                const double tl = __longlong_as_double(sub_rl64<SW>(__double_as_longlong(dl), t, g));
                if (t < U_ && r0t == r0) {
                    rp += (tp < dp) || (tp == dp && t < r);
                    rl += (tl < dl) || (tl == dl && t < r);
                }
            }
            const i64 Kp = sub_scan_i64<SW>((live && rp < keep) ? (i64)pos : 0), Kl = sub_scan_i64<SW>((live && rl < keep) ? (i64)len : 0);
            const i64 Ks = sub_scan_i64<SW>((live && rp == 0) ? (i64)pos : 0);
            const i64 kp0 = bperm(e04, Kp), kl0 = bperm(e04, Kl), ks0 = bperm(e04, Ks);
            const i64 ks = bperm(e14, Kp) - (r0 > 0 ? kp0 : 0);
            const i64 kl = bperm(e14, Kl) - (r0 > 0 ? kl0 : 0);
            search = (C)(bperm(e14, Ks) - (r0 > 0 ? ks0 : 0));
            const double rcp_k = B.rcp_tab[keep & (SQRT_TAB - 1)];
            bp = div_by((double)ks, (double)keep, rcp_k); siglen = div_by((double)kl, (double)keep, rcp_k);   // INDEL:176-177, 187
        }
        // ---- cal_CIPOS of np.std over positions and lengths (INDEL:191-194, GT:58-60).  The reference's float64 result is
        // replayed exactly further down (numpy's pairwise summation order, IEEE division and square root); but only the INTEGER
        // int(1.96 * std / n ** 0.5) leaves the stage.  numpy's sum of (x - mean)^2 differs from the exact variance
        // (n * sum(d^2) - sum(d)^2) / n^2 (integers, d = x - x0) by a relative error below 4e-12 for coordinates under 2^31 (the
        // mean's rounding error e enters only as n e^2: the first-order term 2 e sum(x - mean) vanishes).  So: the exact integer
        // variance through one or two more scans, the value 1.96 * sqrt(N) / (n * n ** 0.5) in float32 (one v_sqrt_f32, a table
        // factor; relative error < 5e-7), and the replay only for a wavefront in which some allele's value lies within 2e-6
        // (relative) of an integer: about one unit in a thousand.  (Replay, four float64 divisions and two square roots per unit
        // were 28 % of this kernel's instructions.)
        int cip = 0, cil = 0;
        if (!CSV_ABL(2)) {
            bool exact_ok = fast;
            if (exact_ok) {
                i64 s2p, s2l;
                const int adp = dp32 < 0 ? -dp32 : dp32, adl = dl32 < 0 ? -dl32 : dl32;
                if (!__ballot((adp | adl) >> 12)) {
                    // every delta below 2^12 (all but clusters with two far-apart alleles): the squares are below 2^24, their sums
                    // over a sub-wave below 2^30, so both series share one scan like the deltas themselves
                    const i64 Q = sub_scan_i64<SW>((i64)(((u64)(unsigned)(adl * adl) << 32) | (unsigned)(adp * adp)));
                    const i64 q1 = bperm(e14, Q), q0 = bperm(e04, Q);
                    const u64 seg = (u64)(q1 - (r0 > 0 ? q0 : 0));
                    s2p = (i64)(seg & 0xffffffffull); s2l = (i64)(seg >> 32);
                } else {
                    const i64 Q2p = sub_scan_i64<SW>((i64)dp32 * dp32), Q2l = sub_scan_i64<SW>((i64)dl32 * dl32);      // squares < 2^36, sums < 2^42
                    const i64 qp0 = bperm(e04, Q2p), ql0 = bperm(e04, Q2l);
                    s2p = bperm(e14, Q2p) - (r0 > 0 ? qp0 : 0); s2l = bperm(e14, Q2l) - (r0 > 0 ? ql0 : 0);
                }
                const i64 np_ = (i64)n * s2p - (i64)s1p * s1p, nl_ = (i64)n * s2l - (i64)s1l * s1l;          // n^2 * variance, exact, < 2^48
                const float ck = B.cipk_tab[n & (SQRT_TAB - 1)];                     // 1.96 / (n * n ** 0.5)
                const float vp = __builtin_amdgcn_sqrtf((float)(double)np_) * ck, vl = __builtin_amdgcn_sqrtf((float)(double)nl_) * ck;
                cip = (int)vp; cil = (int)vl;
                const bool near_p = vp > 0.5f && fabsf(vp - rintf(vp)) <= 2e-6f * vp, near_l = vl > 0.5f && fabsf(vl - rintf(vl)) <= 2e-6f * vl;
                if (__ballot(live && (near_p || near_l || np_ < 0 || nl_ < 0))) exact_ok = false;        // too close to call: the replay decides
            }
            if (!exact_ok) {
                const int rows_u = (mmax >> 3) + 1, tail_u = mmax < 7 ? mmax : 7;     // wave-uniform bounds (allele size <= m)
                double vsp = 0.0, vsl = 0.0;
                np_sum_allele2(((double)pos - pmean) * ((double)pos - pmean), ((double)len - lmean) * ((double)len - lmean), n, i, rows_u, tail_u, vsp, vsl);
                const double rt = B.sqrt_tab[n & (SQRT_TAB - 1)];
                cip = (int)(1.96 * sqrt(vsp / (double)n) / rt);                       // INDEL:191, GT:58-60
                cil = (int)(1.96 * sqrt(vsl / (double)n) / rt);                       // INDEL:194
            }
        }

        // ---- INS: first member (allele order) whose sequence is long enough gives POS and ALT (INDEL:398-405)
        // (int32 columns: a mean of values inside the int32 range is inside it too - one v_cvt_i32_f64 instead of the dozen
        // instructions of a float64 -> int64 conversion; both truncate toward zero)
        i64 want, bp_t;
        if constexpr (NARROW) { want = (i64)(int)siglen; bp_t = (i64)(int)bp; } else { want = (i64)siglen; bp_t = (i64)bp; }
        const u64 okm = sub_ballot<SW>(live && type == CSV_INS && (i64)axp >= want, g);
        const u64 range = ((n >= 64) ? ~0ull : ((1ull << n) - 1ull)) << r0;
        const u64 mm = okm & range;
        const int pr = mm ? (__ffsll((long long)mm) - 1) : r0;
        const int pr4 = (hb | pr) << 2;
        const int pick_ch = bperm(pr4, chp);
        const C pick_pos = bperm(pr4, pos);
        i64 pick = -1; bool valid = true;
        i64 bp_i = bp_t, search_i = (i64)search;
        if (type == CSV_INS) {
            valid = mm != 0;
            bp_i = (i64)pick_pos;                                                 // (the reference's float(pos) -> int() round trip is exact below 2^53)
            search_i = bp_i;                                                      // INDEL:415
        }
        if (pass && !CSV_ABL(4)) B.sup_tmp[s + soff + i] = s + chp;                     // INDEL:205, 416
        const bool head = pass && i == 0;
        if (head && !CSV_ABL(4)) {
            // the picked signature as its row in w space; k_emit, which has the segment's offsets at hand for the support lists anyway,
            // turns it into the caller's global index (bit 1 of `valid`).  (Here that was two dependent loads - the segment's
            // sig_begin, its woff - and 64-bit sums at the very end of a unit, with every lane of the wavefront waiting.)
            int vflag = valid ? 1 : 0;
            if (type == CSV_INS && valid) { pick = (i64)(s + pick_ch); vflag |= 2; }
            tmp_write(tmp_slot(B, j, s, erank), bp_i, want, search_i, pick, n, cip, cil, soff, vflag, npass, aux0);
        }
        const int ncalls = __popcll(sub_ballot<SW>(head && valid, g));
        const int nsup = bperm(last4, sub_scan_i32<SW>((head && valid) ? n : 0));
        if (ok && sl == 0) item_done(B, j, ncalls, nsup);
    } while (0);
}

// (int64 columns keep their coordinates in register pairs: one wavefront per SIMD fewer, and no spills either)
template <bool NARROW> __global__ __launch_bounds__(256, NARROW ? CSV_IW_WAVES : CSV_IW_WAVES - 1) void k_refine_indel_wave(DevBatch B)
{
    __shared__ IwLds s_iw[4];
    CSV_LDS IwLds* L = (CSV_LDS IwLds*)&s_iw[__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))];       // (a scalar: not a VGPR held across the kernel)
    const int ntiny = B.cnt->n_items_tiny, nsmall = B.cnt->n_items - B.cnt->n_items_big - ntiny;
    const int wave = __builtin_amdgcn_readfirstlane((blockIdx.x * 256 + threadIdx.x) >> 6), nwaves = (gridDim.x * 256) >> 6;
    // Units: the wide items (one cluster of 33 .. 64 signatures per wavefront), the pairs of the small list, the quads of the
    // tiny list (clusters of at most 16 signatures: four per wavefront, a sub-wave is one DPP row); unit u goes to wavefront
    // u mod nwaves.  A unit costs two dependent round trips: its list entries, then its rows.
    const int n_pair = (nsmall + 1) / 2, n_quad = (ntiny + 3) / 4, n_wide = B.cnt->n_items_wide;
    // (r05, rejected with a measurement: handing the units out on demand - every wavefront's first unit its own index, further
    // ones from ticket counters drawn one unit ahead, 32 counters by residue class - made this kernel 63 us instead of 12.5 on
    // the 30x genome and 260 instead of 41 on the 90x one: ~10 k device-scope atomics on 32 addresses are served one after the
    // other at 150-200 ns each.  The static deal below is balanced in COUNT per SIMD already - workgroups go round-robin over
    // XCDs and CUs, so the waves that get a second unit are spread evenly - and the remaining spread is the units' own cost.)
    // Longest first: the wide items (32 < m <= 64, the one-cluster-per-wavefront form, about twice a pair's latency) go to
    // the wavefronts that are dispatched first; the pairs and then the quads are dealt round-robin over the OTHER
    // wavefronts, so that a wavefront that already has a wide item is not also the one that gets a second unit when there
    // are more units than wavefronts.  (More wide items than wavefronts - deep coverage - : everything over all of them.)
    for (int p = wave; p < n_wide; p += nwaves) {
        UnitIn<NARROW> U;
        unit_rows<NARROW>(B, unit_entry(B.list_wide, p, n_wide, 6), 6, 32, U);
        indel_unit<64, NARROW>(B, U, L);
    }
    const int skip = n_wide < nwaves ? n_wide : 0, M = nwaves - skip;
    if (wave < skip) return;
    const int slot = wave - skip;
    for (int p = slot; p < n_pair; p += M) {
        UnitIn<NARROW> U;
        unit_rows<NARROW>(B, unit_entry(B.list_small, p, nsmall, 5), 5, 0, U);
        indel_unit<32, NARROW>(B, U, L);                      // (members with 32 < m <= 64 are skipped here: they are units of their own)
    }
    int q0 = slot - n_pair % M;                            // the quads continue the round-robin where the pairs stopped
    if (q0 < 0) q0 += M;
    for (int p = q0; p < n_quad; p += M) {
        UnitIn<NARROW> U;
        unit_rows<NARROW>(B, unit_entry(B.list_tiny, p, ntiny, 4), 4, 0, U);
        indel_unit<16, NARROW>(B, U, L);
    }
    // (r04, rejected: loading the next unit's entry two units ahead and its rows one unit ahead - 12 more VGPRs, so five
    // wavefronts per SIMD or spills at six - was 17.4 / 19.1 us against 15.0 on the 30x genome: occupancy hides more latency
    // than the prefetch saves, because only two wavefronts in three have a second unit at all)
}

// ------------------------------------------------------------------------------------ order
// exclusive prefix of the packed (calls, supports) counts, in two levels: k_items_scan leaves, per tile of
// EM_TILE = 8 items, the prefix INSIDE its chunk of IS_CHUNK items plus one total per chunk; k_emit adds the totals
// of the earlier chunks (a few dozen L2-resident words, one wave reduction) and finishes the prefix inside its
// wavefront.  One workgroup per chunk; a wavefront takes IS_CH pieces of 512 items: coalesced row loads, an LDS
// transpose so that lane t owns tile t of the piece, one wave scan per piece.  (One workgroup sweeping the whole
// list - the first version - was a 6 us latency chain through a single CU.)
constexpr int IS_CH = 2;                             // pieces of 512 items per wavefront
constexpr int IS_NW = 4;                             // wavefronts per workgroup
constexpr int IS_CHUNK = IS_NW * IS_CH * 512;        // items per workgroup (4096)
__global__ __launch_bounds__(64 * IS_NW) void k_items_scan(DevBatch B)
{
    const int n = B.cnt->n_items;
    const int wv = threadIdx.x >> 6, lane = lane_id();
    const int base = blockIdx.x * (IS_CHUNK / EM_TILE);     // first tile of the chunk
    __shared__ i64 buf[IS_NW][64 * 9];                      // [tile][8 items], rows padded to 9
    __shared__ i64 wsum[IS_NW];
    // (the counts are loaded before the item count `n` is looked at - indices clamped to the table, stale entries masked -:
    // one round trip instead of two)
    const int imax = B.cap_items - 1;
    i64 raw[IS_CH][8];
#pragma unroll
    for (int c = 0; c < IS_CH; c++)
#pragma unroll
        for (int r = 0; r < 8; r++) { const int i = (base + (wv * IS_CH + c) * 64) * EM_TILE + r * 64 + lane; raw[c][r] = B.item_cnt[i < imax ? i : imax]; }
    const int ntiles = (n + EM_TILE - 1) / EM_TILE;
    if (base >= ntiles) return;
    i64 ts[IS_CH];
#pragma unroll
    for (int c = 0; c < IS_CH; c++) {
        const int tile0 = base + (wv * IS_CH + c) * 64;
        i64 v[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int i = tile0 * EM_TILE + r * 64 + lane;
            v[r] = i < n ? raw[c][r] : 0;
        }
#pragma unroll
        for (int r = 0; r < 8; r++) buf[wv][(r * 8 + (lane >> 3)) * 9 + (lane & 7)] = v[r];
        // buf[wv] belongs to this wavefront alone: its LDS operations execute in order, so a wave-level fence
        // (no instruction, just no reordering by the compiler) is all the transpose needs
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        i64 t = 0;
#pragma unroll
        for (int e = 0; e < 8; e++) t += buf[wv][lane * 9 + e];
        ts[c] = t;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    i64 inc[IS_CH]; i64 tot = 0;
#pragma unroll
    for (int c = 0; c < IS_CH; c++) { inc[c] = wave_incl_scan_i64(ts[c]); tot += lane63_i64(inc[c]); }
    if (lane == 0) wsum[wv] = tot;
    __syncthreads();
    i64 run = 0;
    for (int k = 0; k < wv; k++) run += wsum[k];
#pragma unroll
    for (int c = 0; c < IS_CH; c++) {
        const int tile = base + (wv * IS_CH + c) * 64 + lane;
        if (tile < ntiles) B.item_base[tile] = run + inc[c] - ts[c];
        run += lane63_i64(inc[c]);
    }
    if (threadIdx.x == 64 * IS_NW - 1) B.item_chunk[blockIdx.x] = run;
}
// packed (calls, supports) of the chunks before chunk c (every lane gets the sum)
__device__ __forceinline__ i64 chunks_before(const DevBatch& B, int c)
{
    i64 v = 0;
    for (int i = lane_id(); i < c; i += 64) v += B.item_chunk[i];
    return wave_sum_i64(v);
}

// one call record: six 16-byte stores to consecutive addresses
__device__ __forceinline__ void write_call(const DevBatch& B, int c, i64 bp1, i64 bp2, i64 search, i64 pick, i64 supoff, int support,
                                           int cipos, int cilen, int seg, int cluster, int aux, int4 ghdr)
{
    int4* r = (int4*)&B.o_rec[c];
    r[0] = make_int4((int)(bp1 & 0xffffffffll), (int)(bp1 >> 32), (int)(bp2 & 0xffffffffll), (int)(bp2 >> 32));
    r[1] = make_int4((int)(search & 0xffffffffll), (int)(search >> 32), (int)(pick & 0xffffffffll), (int)(pick >> 32));
    r[2] = make_int4((int)(supoff & 0xffffffffll), (int)(supoff >> 32), support, cipos);
    r[3] = make_int4(cilen, seg, cluster, aux);
    r[4] = make_int4(-1, -1, -1, 0);                     // dr, dv, gl_idx: "not genotyped" until a genotype kernel says otherwise
    r[5] = ghdr;
}

// compact one item's valid temp calls into the final arrays (one wavefront, any slot count)
__device__ __forceinline__ void emit_item_serial(const DevBatch& B, int j, i64 base)
{
    const int lane = lane_id();
    const int nslots = B.t_rec0[j].nslots, aux0 = B.t_rec0[j].aux0;
    const int4 rec = B.item_rec[j];
    const int cid = rec.x, k = rec.y, s = rec.z;
    const csv_segment& sgk = B.seg[k];
    const i64 gs = sgk.sig_begin + ((i64)s - B.woff[k]) - s;           // w -> global signature index
    const int4 ghdr = make_int4(sgk.chrom, sgk.svtype | (sgk.genotype ? 0x100 : 0), (int)(sgk.gt_bias & 0xffffffffll), (int)(sgk.gt_bias >> 32));
    int cb = (int)(base >> 32); i64 sb = base & 0xffffffffll;
    for (int c0 = 0; c0 < nslots; c0 += 64) {
        const int in = (c0 + lane) < nslots;
        TmpRec tr;
        tr.valid = 0;
        if (in) tr = *tmp_slot(B, j, s, c0 + lane);
        const int valid = in ? (tr.valid & 1) : 0;
        if (in && (tr.valid & 2)) tr.pick += gs;                            // (a w-space pick of the register tier -> the caller's row)
        const int nsup = valid ? tr.support : 0;
        const int tso = valid ? tr.supoff : 0;
        const u64 mk = __ballot(valid);
        const int c = cb + __popcll(mk & lanemask_lt());
        const int sinc = wave_incl_scan_i32(nsup);
        const i64 so = sb + sinc - nsup;
        if (valid) write_call(B, c, tr.bp1, tr.bp2, tr.search, tr.pick, so, nsup, tr.cipos, tr.cilen, k, cid, aux0, ghdr);
        u64 rest = mk;
        while (rest) {
            const int l = __ffsll((long long)rest) - 1;
            rest &= rest - 1;
            const int cc = __shfl(c, l), nn = __shfl(nsup, l);
            const i64 dst = shfl_i64(so, l);
            const int src = s + __shfl(tso, l);
            for (int i = lane; i < nn; i += 64) {
                const int w = B.sup_tmp[src + i];
                B.o_supsig[dst + i] = (int)(gs + w);
                if (sgk.genotype) { const int r = B.rid[w]; B.o_suprid[dst + i] = r; if (r < 0) atomicOr(&B.cnt->error, ERR_KEY_RANGE); }
                if (B.per_sig) B.allele_id[w] = cc;
            }
        }
        cb += __popcll(mk);
        sb += __builtin_amdgcn_readlane(sinc, 63);
    }
}

// One wavefront per tile of EM_TILE = 8 consecutive items; 8 lanes per item, one lane per temp slot,
// so the dependent loads of all 8 items are in flight together.  Items with more than 8 slots (rare)
// send the whole tile down the serial path.
__global__ __launch_bounds__(256) void k_emit(DevBatch B)
{
    const int wave = __builtin_amdgcn_readfirstlane((blockIdx.x * 256 + threadIdx.x) >> 6), nwaves = (gridDim.x * 256) >> 6;
    const int lane = lane_id(), g = lane >> 3, l8 = lane & 7;
    const int jmax = B.cap_items - 1;
    int n = -1;
    for (int tile = wave;; tile += nwaves) {
        if (n >= 0 && tile >= (n + EM_TILE - 1) / EM_TILE) break;        // (after the first tile the count is known: no loads for nothing)
        const int j = tile * EM_TILE + g;
        // round 1: everything that only depends on the item index is loaded together - and together with the item count
        // itself: the loads of a tile do not wait for `n` (indices are clamped to the tables' capacity; what lies beyond n is
        // stale and masked below), so a wavefront's first tile costs one round trip less.  (n through a vector load issued
        // AFTER the others: a scalar load of it is hoisted to the top of the kernel and waited for there.)
        const int jj = j < jmax ? j : jmax, tt = tile < jmax / EM_TILE ? tile : jmax / EM_TILE;
        const i64 cnt_raw = B.item_cnt[jj];
        const int4 rec_raw = B.item_rec[jj];
        // slot 0 of the item: dense and item-indexed, so it travels with round 1 (the 8 records of a tile are 512 contiguous
        // bytes; the group's first lane takes its item's four 16-byte words)
        int4 q0 = make_int4(0, 0, 0, 0), q1 = q0, q2 = q0, q3 = q0;
        if (l8 == 0) { const int4* r = (const int4*)&B.t_rec0[jj]; q0 = r[0]; q1 = r[1]; q2 = r[2]; q3 = r[3]; }
        const int nchunk = tt / (IS_CHUNK / EM_TILE);
        i64 cb_raw = B.item_chunk[lane < nchunk ? lane : 0];
        i64 tile_base = B.item_base[tt];
        if (n < 0) n = __hip_atomic_load(&B.cnt->n_items, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int ntiles = (n + EM_TILE - 1) / EM_TILE;
        if (tile >= ntiles) break;
        if (lane >= nchunk) cb_raw = 0;
        for (int i = 64 + lane; i < nchunk; i += 64) cb_raw += B.item_chunk[i];      // (more than 64 chunks: a quarter of a million items)
        tile_base += wave_sum_i64(cb_raw);
        const bool act = j < n;
        const i64 cnt = act ? cnt_raw : 0;
        const int4 rec = act ? rec_raw : make_int4(0, 0, 0, 0);
        const i64 ginc = wave_incl_scan_i64(l8 == 0 ? cnt : 0);      // prefix over the tile's 8 items (one lane per group contributes)
        const i64 base = tile_base + ginc - cnt;
        if (tile == ntiles - 1 && lane == 63) {             // the last tile closes the prefix: totals of the batch
            const i64 t = tile_base + ginc;
            B.cnt->n_calls = (int)(t >> 32); B.cnt->n_support = t & 0xffffffffll;
        }
        const int ns0 = __shfl(q3.y, g * 8);                // (cross-lane moves stay outside per-lane selects)
        const int nslots = cnt ? ns0 : 0;
        if (__ballot(nslots > 8)) {                         // wave-uniform
            for (int q = 0; q < EM_TILE; q++) {
                const i64 cq = shfl_i64(cnt, q * 8), bq = shfl_i64(base, q * 8);
                if (cq) emit_item_serial(B, tile * EM_TILE + q, bq);
            }
            continue;
        }
        const int cid = rec.x, k = rec.y, s = rec.z;
        const bool mine = l8 < nslots;
        // ... round 2: the records of the slots >= 1 (one lane per temp slot; one item in a hundred has any), valid or not, and
        // the segment scalars
        if (mine && l8 > 0) { const int4* r = (const int4*)&B.t_rec[s + l8]; q0 = r[0]; q1 = r[1]; q2 = r[2]; q3 = r[3]; }
        int valid = 0, nsup = 0, tso = 0, ci = 0, cl = 0;
        bool pick_w = false;
        i64 bp1 = 0, bp2 = 0, srch = 0, pick = 0;
        if (mine) {
            valid = q3.x & 1; pick_w = (q3.x & 2) != 0; nsup = q2.x; tso = q2.w; ci = q2.y; cl = q2.z;
            bp1 = ((i64)q0.y << 32) | (unsigned)q0.x; bp2 = ((i64)q0.w << 32) | (unsigned)q0.z;
            srch = ((i64)q1.y << 32) | (unsigned)q1.x; pick = ((i64)q1.w << 32) | (unsigned)q1.z;
        }
        const int aux0 = __shfl(q3.z, g * 8);               // (slot 0 carries the cluster's first aux word: no gather of B.aux[s])
        // (... and, before the slot records say how long the lists are, the first 32 supports of the item's first slot, whose list
        // always begins at the cluster's own first row: most items have one call with fewer supports than that, and their third
        // round trip disappears)
        int pre[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const i64 x = (i64)s + l8 + 8 * u; pre[u] = B.sup_tmp[x < B.W ? x : B.W]; }
        i64 gs = 0;
        int4 ghdr = make_int4(0, 0, 0, 0);
        if (nslots) {
            const csv_segment& sgk = B.seg[k];
            gs = sgk.sig_begin + ((i64)s - B.woff[k]) - s;
            ghdr = make_int4(sgk.chrom, sgk.svtype | (sgk.genotype ? 0x100 : 0), (int)(sgk.gt_bias & 0xffffffffll), (int)(sgk.gt_bias >> 32));
        }
        const bool gtseg = (ghdr.y & 0x100) != 0;
        if (pick_w) pick += gs;                               // (the register tier's picks are rows in w space: gs + w = the caller's row)
        if (!valid) { nsup = 0; tso = 0; }
        const u64 mk = __ballot(valid);
        const u64 gmask = 0xffull << (g * 8);
        const int c = (int)(base >> 32) + __popcll(mk & gmask & lanemask_lt());
        const int sinc = wave_incl_scan_i32(nsup);
        const int gprev = __shfl(sinc, (g * 8 - 1) & 63);
        const i64 so = (base & 0xffffffffll) + (sinc - nsup) - (g ? gprev : 0);
        if (valid) write_call(B, c, bp1, bp2, srch, pick, so, nsup, ci, cl, k, cid, aux0, ghdr);
        // supports: group g copies the lists of its own slots, 8 lanes at a time
#pragma unroll
        for (int sl = 0; sl < 8; sl++) {
            const int src_lane = g * 8 + sl;
            const int nn = __shfl(nsup, src_lane), cc = __shfl(c, src_lane), ts = __shfl(tso, src_lane);
            const i64 dst = shfl_i64(so, src_lane);
            if (!__ballot(nn > 0)) continue;                // wave-uniform
            // 4 strides per step, loads issued together (lists are 15-90 long)
            for (int i = l8; i < nn; i += 32) {
                int sr[4], rd[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (sl == 0 && i == l8 && ts == 0) sr[u] = (i + 8 * u < nn) ? pre[u] : -1;
                    else sr[u] = (i + 8 * u < nn) ? B.sup_tmp[s + ts + i + 8 * u] : -1;
                }
                // the supports' read ids seed the cover sets of the genotype kernels: gathered here, for genotyped segments only
                // (the rows were read by the refine kernel a moment ago)
#pragma unroll
                for (int u = 0; u < 4; u++) rd[u] = (gtseg && sr[u] >= 0) ? B.rid[sr[u]] : 0;
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (sr[u] >= 0) {
                        B.o_supsig[dst + i + 8 * u] = (int)(gs + sr[u]);
                        if (gtseg) {
                            B.o_suprid[dst + i + 8 * u] = rd[u];
                            if (rd[u] < 0) atomicOr(&B.cnt->error, ERR_KEY_RANGE);      // (a negative id would pass for an empty hash slot)
                        }
                        if (B.per_sig) B.allele_id[sr[u]] = cc;
                    }
            }
        }
    }
}

// ------------------------------------------------------------------------------------ publish
// The results straight into the caller's arrays, when those live in page-locked host memory (csv_host_alloc /
// csv_host_register): the device knows the counts, so one kernel writes the calls in their final structure-of-arrays
// layout, the support list, the segment status words and the counters across the link, and the host's whole download
// is one stream synchronisation - no counter round trip before the copies can be sized, no staging copy, no unpack
// loop.  Nothing is written when the caller's capacities are too small (the counters still are: CSV_E_CAPACITY).
struct PublishArgs {
    i64 cap_calls, cap_support;
    DevCounters* h_cnt;      // page-locked landing zone of the counters
    int* h_seg_err;          // n_seg words (staging block)
    int n_seg;
    int *call_seg, *call_cluster, *call_aux, *support, *cipos, *cilen, *dr, *dv, *gl_idx;      // (NULL: the caller does not want the field)
    void *bp1, *bp2, *search_pos, *seq_pick;    // i64, or int when coord32 (CSV_OUT_COORD_I32)
    i64 *support_off, *support_sig;
    int *support_sig32;      // the caller's int32 support list (then support_sig is NULL)
    int coord32, no_support; // CSV_OUT_COORD_I32 / CSV_OUT_NO_SUPPORT_LIST
};
__device__ __forceinline__ void publish_coord(void* p, int c32, i64 i, int lo, int hi)
{
    if (!p) return;
    if (c32) ((int*)p)[i] = lo; else ((i64*)p)[i] = ((i64)hi << 32) | (unsigned)lo;
}
__global__ __launch_bounds__(256) void k_publish(DevBatch B, PublishArgs P)
{
    const i64 tid = (i64)blockIdx.x * 256 + threadIdx.x, nth = (i64)gridDim.x * 256;
    const i64 nc = B.cnt->n_calls, ns = B.cnt->n_support;
    if (tid < (i64)(sizeof(DevCounters) / 4)) {            // (the reads-order state of the upload travels in the counters' slots)
        int v = ((const int*)B.cnt)[tid];
        if (B.rs && tid == (i64)(offsetof(DevCounters, n_runs) / 4)) v = B.rs->n_runs;
        if (B.rs && tid == (i64)(offsetof(DevCounters, ro_state) / 4)) v = B.rs->ro_state;
        if (B.rs && tid == (i64)(offsetof(DevCounters, error) / 4)) v |= B.rs->error;
        ((int*)P.h_cnt)[tid] = v;
    }
    for (i64 k = tid; k < P.n_seg; k += nth) P.h_seg_err[k] = B.seg_err[k];
    if (nc > P.cap_calls || (!P.no_support && ns > P.cap_support)) return;
    for (i64 i = tid; i < nc; i += nth) {
        const int4* r = (const int4*)&B.o_rec[i];
        const int4 r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3], r4 = r[4];
        // (which fields travel is wave-uniform: kernel arguments)
        publish_coord(P.bp1, P.coord32, i, r0.x, r0.y); publish_coord(P.bp2, P.coord32, i, r0.z, r0.w);
        publish_coord(P.search_pos, P.coord32, i, r1.x, r1.y); publish_coord(P.seq_pick, P.coord32, i, r1.z, r1.w);
        if (!P.no_support) P.support_off[i] = ((i64)r2.y << 32) | (unsigned)r2.x;
        P.support[i] = r2.z; P.call_seg[i] = r3.y;
        if (P.cipos) P.cipos[i] = r2.w;
        if (P.cilen) P.cilen[i] = r3.x;
        if (P.call_cluster) P.call_cluster[i] = r3.z;
        if (P.call_aux) P.call_aux[i] = r3.w;
        if (P.dr) P.dr[i] = r4.x;
        if (P.dv) P.dv[i] = r4.y;
        if (P.gl_idx) P.gl_idx[i] = r4.z;
    }
    if (P.no_support) return;
    if (tid == 0) P.support_off[nc] = ns;
    if (P.support_sig32) { for (i64 i = tid; i < ns; i += nth) P.support_sig32[i] = B.o_supsig[i]; }
    else for (i64 i = tid; i < ns; i += nth) P.support_sig[i] = (i64)B.o_supsig[i];
}

// ------------------------------------------------------------------------------------ reads: order + pack
// The reads block of a chromosome arrives in the order cuteSV's rebuild step leaves it (main script :810): the
// concatenation of per-worker extraction batches, each batch being the reads that START inside one task region in BAM
// order (:697-735), i.e. a permutation of DISJOINT, start-sorted runs.  overlap_cover sorts its sweep events inside
// the stage (cuteSV_genotype.py:101-109); here the stage brings every block into stable start order: k_reads_runs lists
// the descents (run starts), k_reads_plan orders the runs (one workgroup; a genome has a few hundred) and checks that
// they do not interleave, k_reads_gather moves whole runs.  A table that is not a permutation of disjoint runs (or has
// more than ro_cap of them) is reported back (RO_NEED_GENERAL) and the host re-runs the batch through the general stable
// radix sort (sort.hip.h) - correct for any input, just slower.
//
// What the genotype kernels read is the PACKED, start-ordered form k_reads_gather writes (in every mode - a table that was
// promised or found sorted is packed in place order): s_start / s_end in the width the caller sent (int32 when the
// coordinates fit: CSV_IN_READS_I32), s_idp = read id | primary << 31, and per chunk of 64 reads the largest end (cmax) and
// the chromosome's longest read (maxlen).  A stabbing query for the window [L, R]
// looks at the chunks between the first read that could still reach R (start >= R - longest read of the chromosome) and the
// last read with start <= L, and only at those whose cmax reaches R: 8 bytes per read actually scanned.  (The first form
// kept a per-read prefix maximum of the ends - 8 more bytes per read written and read - and scanned backwards while it
// stayed >= R: a prefix maximum is monotone, so ONE ultra-long read kept every later call scanning ~900 reads of which ~90
// covered: 6.7x the algorithmic traffic on the 90x ultra-long workload.)
__device__ __forceinline__ int chrom_of_read(const DevBatch& B, i64 i, int hint)
{
    if (B.reads_off[hint] <= i && i < B.reads_off[hint + 1]) return hint;
    int lo = 0, hi = B.n_chrom;
    while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (B.reads_off[mid] <= i) lo = mid; else hi = mid; }
    return lo;
}
constexpr i64 READ_END_MAX = 1ll << 40;             // ends are doubled in window arithmetic: anything beyond is a broken table

constexpr int RO_TILE = 256 * 8;
constexpr int RO_TCAP = 32;                         // run starts a tile of RO_TILE rows may hold (more: the general sort)
// A row starts a run when its start is smaller than its predecessor's (a descent) or jumps ahead by more than ro_gap.
// The second rule cuts the runs the extraction step glued together: a worker that processed the task regions 3 and then 7
// of a chromosome leaves them back to back without a descent, although the regions 4-6 (in other workers' files)
// belong in between; such a seam is at least one task region wide (--batches, 10 Mbp by default), ro_gap is 1 Mbp.
// The rule is only a heuristic for WHERE to cut - k_reads_plan verifies that the pieces do not interleave once
// ordered, and anything else goes to the general sort.  (Block starts are added by k_reads_plan.)
// Every tile keeps its own short list {row, its start, the start of the row before, out of range} in row order + a count: no
// device atomics, and k_reads_plan needs no look-up of the starts - its critical path is two round trips.
template <bool RN> __global__ __launch_bounds__(256) void k_reads_runs(DevBatch B)
{
    __shared__ int s_w[4];
    __shared__ u64 s_blk[RO_TILE / 64];                      // rows of the tile at which a chromosome block begins
    const int c0 = B.ro_tblk[blockIdx.x], c1 = B.ro_tblk[blockIdx.x + 1];
    if (threadIdx.x < RO_TILE / 64) s_blk[threadIdx.x] = 0;
    // a lane owns 8 consecutive rows (16-byte loads); the start before its first row comes from the lane to the left
    const i64 base = (i64)blockIdx.x * RO_TILE + (threadIdx.x >> 6) * 512, i0 = base + lane_id() * 8;
    i64 v[8];
    i64 prev0 = INT64_MIN;
    unsigned flags = 0;
    if (base < B.n_reads) {
        if (i0 + 8 <= B.n_reads) {
            if constexpr (RN) {
                const int4 x0 = *(const int4*)(B.r_start.p32 + i0), x1 = *(const int4*)(B.r_start.p32 + i0 + 4);
                v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w; v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
            } else {
#pragma unroll
                for (int r = 0; r < 8; r += 2) { const longlong2 x = *(const longlong2*)(B.r_start.p64 + i0 + r); v[r] = x.x; v[r + 1] = x.y; }
            }
        } else {
#pragma unroll
            for (int r = 0; r < 8; r++) v[r] = i0 + r < B.n_reads ? col_at<RN>(B.r_start, i0 + r) : INT64_MAX;
        }
        prev0 = (lane_id() == 0 && base > 0) ? col_at<RN>(B.r_start, base - 1) : INT64_MIN;
        const i64 left = wave_shr1_i64(v[7]);
        if (lane_id() != 0) prev0 = left;
        i64 prev = prev0;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const i64 i = i0 + r;
            if (i < B.n_reads && i > 0 && prev != INT64_MIN && (v[r] < prev || v[r] - prev > B.ro_gap)) flags |= 1u << r;
            prev = v[r];
        }
    }
    if (c1 > c0) {                                          // (wave-uniform, and rare in a genome of a few long chromosomes)
        __syncthreads();
        for (int c = c0 + threadIdx.x; c < c1; c += 256) {
            const i64 o = B.reads_off[c] - (i64)blockIdx.x * RO_TILE;
            if (o >= 0 && o < RO_TILE) atomicOr((unsigned long long*)&s_blk[o >> 6], 1ull << (o & 63));
        }
        __syncthreads();
        flags &= ~(unsigned)((const uint8_t*)s_blk)[((threadIdx.x >> 6) * 512 + lane_id() * 8) >> 3];       // a descent AT a block start is the plan's business
    }
    // the tile's records in row order: exclusive prefix of the per-thread counts (a tile has a handful)
    const int cnt = __popc(flags);
    const int inc = wave_incl_scan_i32(cnt);
    if (lane_id() == 63) s_w[threadIdx.x >> 6] = inc;
    __syncthreads();
    int slot = inc - cnt, tot = 0;
    for (int q = 0; q < 4; q++) { if (q < (int)(threadIdx.x >> 6)) slot += s_w[q]; tot += s_w[q]; }
    if (flags) {
        i64 prev = prev0;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            if ((flags >> r) & 1) {
                const int bad = (v[r] < 0 || v[r] >= (1ll << 32) || prev < 0 || prev >= (1ll << 32)) ? 1 : 0;
                if (slot < RO_TCAP) B.ro_ent[(i64)blockIdx.x * RO_TCAP + slot] = make_int4((int)(i0 + r), (int)(unsigned)v[r], (int)(unsigned)prev, bad);
                slot++;
            }
            prev = v[r];
        }
    }
    if (threadIdx.x == 0) B.ro_tcnt[blockIdx.x] = tot;
}

// one workgroup: the tiles' run starts + one start per chromosome block -> runs by (chromosome, first start, position);
// disjointness / stability check; destination offsets.  Two global round trips (counts + offsets, then the records); everything else is binary searches and
// short scans over LDS: the tiles' lists arrive in position order and are MERGED with the block starts by rank, and a run is
// ranked only against the runs of its own chromosome.  (Two 64-bit rank sorts of all runs against all runs were 20 of
// this kernel's 34 us for the 330 runs of a 30x genome.)  LDS: 36 bytes per run.
constexpr int RP_THREADS = 1024;
// exclusive block scan of one int per thread (RP_THREADS threads); returns the thread's offset, *total = the sum
__device__ __forceinline__ int rp_block_excl(int v, int* s_w, int* total)
{
    const int inc = wave_incl_scan_i32(v);
    __syncthreads();
    if (lane_id() == 63) s_w[threadIdx.x >> 6] = inc;
    __syncthreads();
    int off = 0, tot = 0;
    for (int q = 0; q < RP_THREADS / 64; q++) { const int x = s_w[q]; if (q < (int)(threadIdx.x >> 6)) off += x; tot += x; }
    *total = tot;
    return off + inc - v;
}
// first index in [0, n) with a[i] >= x / > x (a sorted ascending)
__device__ __forceinline__ int rp_lower(const int* a, int n, int x) { int lo = 0, hi = n; while (lo < hi) { const int mid = (lo + hi) >> 1; if (a[mid] < x) lo = mid + 1; else hi = mid; } return lo; }
__device__ __forceinline__ int rp_upper(const int* a, int n, int x) { int lo = 0, hi = n; while (lo < hi) { const int mid = (lo + hi) >> 1; if (a[mid] <= x) lo = mid + 1; else hi = mid; } return lo; }
constexpr int rp_lds_bytes(int cap) { return 9 * (cap + 2) * 4 + 64; }
template <bool RN> __global__ __launch_bounds__(RP_THREADS) void k_reads_plan(DevBatch B)
{
    extern __shared__ __attribute__((aligned(16))) char rp_smem[];
    const int cap = B.ro_cap, nc = B.n_chrom, tid = threadIdx.x, A = cap + 2;
    int* fpos = (int*)rp_smem;                 // found run starts: row (ascending), ...
    unsigned* ffv = (unsigned*)(fpos + A);     // ... its start, the start of the row before it
    unsigned* fpv = ffv + A;
    int* off = (int*)(fpv + A);                // reads_off
    int* kb = off + A;                         // blocks listed on their own (non-empty, not already a found start) before block c
    int* pos = kb + A;                         // merged, by position (+ sentinel): row, first start, preceding start, chromosome
    unsigned* fv = (unsigned*)(pos + A);
    unsigned* pv = fv + A;
    int* chv = (int*)(pv + A);
    int* ord = fpos;                           // (after the merge) start order -> merged index
    int* len_s = (int*)ffv;                    //                   lengths in start order
    __shared__ int s_bad, s_moved;
    __shared__ int s_w[RP_THREADS / 64];
    __shared__ unsigned s_last;
    if (tid == 0) { s_bad = 0; s_moved = 0; }
    if (nc + 1 > cap) { if (tid == 0) B.rs->ro_state = RO_NEED_GENERAL; return; }
    // round trip 1: the tiles' counts (eight per thread at a time, issued together), the block offsets, the last start
    const int n_tiles = (int)((B.n_reads + RO_TILE - 1) / RO_TILE);
    const int per = (n_tiles + RP_THREADS - 1) / RP_THREADS;
    const int t0 = tid * per < n_tiles ? tid * per : n_tiles, t1 = t0 + per < n_tiles ? t0 + per : n_tiles;
    int mine = 0, over = 0;
    int cnt8[8];                                                // the first eight counts stay in registers for the second pass
#pragma unroll
    for (int r = 0; r < 8; r++) cnt8[r] = t0 + r < t1 ? B.ro_tcnt[t0 + r] : 0;
    for (int c = tid; c <= nc; c += RP_THREADS) off[c] = (int)B.reads_off[c];
    if (tid == 0) {
        const i64 l = col_at<RN>(B.r_start, B.n_reads - 1);
        if (l < 0 || l >= (1ll << 32)) over = 1;
        s_last = (unsigned)l;
    }
#pragma unroll
    for (int r = 0; r < 8; r++) { mine += cnt8[r]; over |= cnt8[r] > RO_TCAP; }
    for (int t = t0 + 8; t < t1; t++) { const int k = B.ro_tcnt[t]; mine += k; over |= k > RO_TCAP; }
    int n_found;
    int at = rp_block_excl(mine, s_w, &n_found);               // (its barriers also publish off[], s_bad, s_last)
    if (over) atomicOr(&s_bad, 1);
    if (n_found + nc > cap) { if (tid == 0) B.rs->ro_state = RO_NEED_GENERAL; return; }
    if (CSV_ABL(20)) { if (tid == 0) B.rs->ro_state = RO_IDENTITY; return; }
    // round trip 2: the run records (a tile's few records in row order; the first record of each of the eight tiles is
    // fetched up front), and the first / preceding start of this thread's block
    i64 bf0 = 0, bp0 = 0;
    if (tid < nc && off[tid] < off[tid + 1]) { bf0 = col_at<RN>(B.r_start, off[tid]); bp0 = off[tid] > 0 ? col_at<RN>(B.r_start, off[tid] - 1) : 0; }
    int4 e8[8];
#pragma unroll
    for (int r = 0; r < 8; r++) e8[r] = (cnt8[r] > 0 && !over) ? B.ro_ent[(i64)(t0 + r) * RO_TCAP] : make_int4(0, 0, 0, 0);
    auto tile_records = [&](int t, int k) {                     // (a tile with several records: they are in row order)
        const int4* E = B.ro_ent + (i64)t * RO_TCAP;
        for (int j = 0; j < k; j++) {
            const int4 e = E[j];
            fpos[at + j] = e.x; ffv[at + j] = (unsigned)e.y; fpv[at + j] = (unsigned)e.z;
            if (e.w) atomicOr(&s_bad, 1);
        }
    };
    if (!over) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int k = cnt8[r];
            if (k == 1) { fpos[at] = e8[r].x; ffv[at] = (unsigned)e8[r].y; fpv[at] = (unsigned)e8[r].z; if (e8[r].w) atomicOr(&s_bad, 1); }
            else if (k > 1) tile_records(t0 + r, k);
            at += k;
        }
        for (int t = t0 + 8; t < t1; t++) { const int k = B.ro_tcnt[t]; if (k) tile_records(t, k); at += k; }
    }
    __syncthreads();
    if (s_bad) { if (tid == 0) B.rs->ro_state = RO_NEED_GENERAL; return; }
    // blocks that are not already a found start get an entry of their own
    if (CSV_ABL(21)) { if (tid == 0) B.rs->ro_state = RO_IDENTITY; return; }
    int n_kept = 0;
    for (int c0 = 0; c0 < nc; c0 += RP_THREADS) {
        const int c = c0 + tid;
        int keep = 0;
        if (c < nc && off[c] < off[c + 1]) { const int f = rp_lower(fpos, n_found, off[c]); keep = !(f < n_found && fpos[f] == off[c]); }
        int tot;
        const int x = n_kept + rp_block_excl(keep, s_w, &tot);
        if (c < nc) kb[c] = x;
        n_kept += tot;
    }
    if (tid == 0) kb[nc] = n_kept;
    const int n = n_found + n_kept;
    __syncthreads();
    // merge by rank
    for (int i = tid; i < n_found; i += RP_THREADS) {
        const int p = fpos[i], ch = rp_upper(off, nc, p) - 1;  // last block with off <= p: the non-empty one among equals
        const int m = i + kb[ch + 1];
        pos[m] = p; fv[m] = ffv[i]; pv[m] = fpv[i]; chv[m] = ch;
    }
    for (int c = tid; c < nc; c += RP_THREADS) {
        const int o = off[c];
        if (o < off[c + 1] && kb[c + 1] > kb[c]) {
            i64 f = bf0, pr = bp0;
            if (c >= RP_THREADS) { f = col_at<RN>(B.r_start, o); pr = o > 0 ? col_at<RN>(B.r_start, o - 1) : 0; }
            if (f < 0 || f >= (1ll << 32) || pr < 0 || pr >= (1ll << 32)) atomicOr(&s_bad, 1);
            const int m = kb[c] + rp_lower(fpos, n_found, o);
            pos[m] = o; fv[m] = (unsigned)f; pv[m] = (unsigned)pr; chv[m] = c;
        }
    }
    if (tid == 0) { pos[n] = (int)B.n_reads; pv[n] = s_last; B.rs->n_runs = n; }
    __syncthreads();
    if (CSV_ABL(22)) { if (tid == 0) B.rs->ro_state = RO_IDENTITY; return; }
    // start order inside every chromosome: a run's rank among the runs of its chromosome by (first start, position)
    for (int m = tid; m < n; m += RP_THREADS) {
        const int ch = chv[m], lo = rp_lower(chv, n, ch), hi = rp_upper(chv, n, ch);
        const unsigned f = fv[m];
        int r = lo;
        for (int j = lo; j < hi; j++) { const unsigned o = fv[j]; r += (o < f) || (o == f && j < m); }
        ord[r] = m;
    }
    __syncthreads();
    if (CSV_ABL(23)) { if (tid == 0) B.rs->ro_state = RO_IDENTITY; return; }
    // consecutive runs of one chromosome must not interleave: last start of the earlier <= first start of the later,
    // and on equality the earlier one must also come first by position (that is what a stable sort would do)
    for (int q = tid; q < n; q += RP_THREADS) {
        const int i = ord[q];
        if (i != q) atomicOr(&s_moved, 1);
        len_s[q] = pos[i + 1] - pos[i];
        if (q > 0) {
            const int ip = ord[q - 1];
            if (chv[ip] == chv[i]) {
                const unsigned last_prev = pv[ip + 1], first = fv[i];      // (the row before the run after ip ends run ip)
                if (last_prev > first || (last_prev == first && ip > i)) atomicOr(&s_bad, 1);
            }
        }
    }
    __syncthreads();
    if (s_bad) { if (tid == 0) B.rs->ro_state = RO_NEED_GENERAL; return; }
    if (!s_moved) { if (tid == 0) B.rs->ro_state = RO_IDENTITY; return; }
    // exclusive scan of the lengths in start order -> destination offsets
    int carry = 0;
    for (int b0 = 0; b0 < n; b0 += RP_THREADS) {
        const int q = b0 + tid;
        const int v = q < n ? len_s[q] : 0;
        int tot;
        const int d = carry + rp_block_excl(v, s_w, &tot);
        if (q < n) {
            const int i = ord[q];
            B.ro_table[q] = make_int4(pos[i], v, d, chv[i]);
        }
        carry += tot;
    }
    if (tid == 0) B.rs->ro_state = RO_REORDER;
}

// (a batch whose reads table turned out to need the general sort is run again by the host: nothing downstream of the
// reads_order stage does any work in the first attempt)
__device__ __forceinline__ bool reads_pending(const DevBatch& B) { return B.ro_mode == 1 && B.rs->ro_state == RO_NEED_GENERAL; }

// stores of one packed row (RN: the caller's columns are int32)
template <bool RN> __device__ __forceinline__ void sread_store(const DevBatch& B, i64 x, i64 st, i64 en, int idp)
{
    if constexpr (RN) { B.s_start32[x] = (int)st; B.s_end32[x] = (int)en; } else { B.s_start64[x] = st; B.s_end64[x] = en; }
    B.s_idp[x] = idp;
}
template <bool RN> __device__ __forceinline__ i64 sread_start(const DevBatch& B, i64 x) { if constexpr (RN) return B.s_start32[x]; else return B.s_start64[x]; }
template <bool RN> __device__ __forceinline__ i64 sread_end(const DevBatch& B, i64 x) { if constexpr (RN) return B.s_end32[x]; else return B.s_end64[x]; }

// One wavefront per 512 destination rows (8 chunks of 64): source row of every destination row - the row itself (table in
// order), the run table (whole sorted runs move; a span usually lies inside one run; k_reads_plan left the span's first run)
// or the permutation of the general sort - then copy + pack, and the chunk's largest end / longest read.  The 32 loads of
// a span are issued before the first is consumed (one chunk at a time, the kernel was a chain of eight dependent round
// trips per wavefront: 3.2 TB/s).  Validates ends and ids, and for a table that was promised sorted the order of the starts.
constexpr int GA_TAB = 512;                         // runs of the table k_reads_gather keeps in LDS (more: it searches global memory)
template <bool RN> __global__ __launch_bounds__(256) void k_reads_gather(DevBatch B)
{
    // the run table of a genome (a few hundred runs) is fetched into LDS by every workgroup before it knows whether it is
    // needed (the buffer exists whenever the mode is 1): one round trip for state + table, then the rows
    __shared__ int4 s_tab[GA_TAB];
    int state = RO_IDENTITY, n = 0;
    if (B.ro_mode == 1) {
        for (int i = threadIdx.x; i < GA_TAB; i += 256) s_tab[i] = B.ro_table[i];
        state = B.rs->ro_state; n = B.rs->n_runs;
        __syncthreads();
    }
    const int wv = threadIdx.x >> 6, lane = lane_id();
    const i64 span = (i64)blockIdx.x * 4 + wv, d0 = span * 512;
    if (d0 >= B.n_reads) return;
    const i64 d1 = d0 + 512 < B.n_reads ? d0 + 512 : B.n_reads;
    if (state == RO_NEED_GENERAL) return;                   // (the host runs the batch again through the general sort)
    const bool by_perm = B.ro_mode == 2, by_runs = state == RO_REORDER;
    i64 p[8];
#pragma unroll
    for (int r = 0; r < 8; r++) p[r] = d0 + r * 64 + lane;
    if (by_perm) {
#pragma unroll
        for (int r = 0; r < 8; r++) { const i64 x = d0 + r * 64 + lane; p[r] = x < d1 ? B.ro_perm[x] : 0; }
    } else if (by_runs) {
        // last run with destination begin <= d0 (64 probes per step), then the runs that reach into this span
        auto walk = [&](auto tab) {
            int lo = 0, hi = n;
            while (hi - lo > 1) {
                const int step = (hi - lo + 63) / 64;
                const int idx = lo + lane * step;
                const int t = __popcll(__ballot(idx < hi && (i64)tab[idx < hi ? idx : lo].z <= d0));
                const int nlo = lo + (t - 1) * step;
                int nhi = lo + t * step;
                if (nhi > hi) nhi = hi;
                lo = nlo; hi = nhi;
            }
            for (int qq = lo; qq < n; qq++) {
                const int4 rn = tab[qq];
                if ((i64)rn.z >= d1) break;
#pragma unroll
                for (int r = 0; r < 8; r++) { const i64 x = d0 + r * 64 + lane; if (x >= rn.z && x < (i64)rn.z + rn.y) p[r] = x + ((i64)rn.x - rn.z); }
            }
        };
        if (n <= GA_TAB) walk((const int4*)s_tab); else walk((const int4*)B.ro_table);
    }
    i64 st[8], en[8]; int id[8], pr[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const i64 x = d0 + r * 64 + lane;
        const i64 pp = x < d1 ? p[r] : p[0];                // (lane 0 of chunk 0 is always a row of the table)
        st[r] = col_at<RN>(B.r_start, pp); en[r] = col_at<RN>(B.r_end, pp); id[r] = B.r_id[pp]; pr[r] = B.r_primary[pp];
    }
    int hint = 0;
    i64 span_len = 0;
    bool bad = false, unsorted = false;
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const i64 c0 = d0 + r * 64, x = c0 + lane;
        if (c0 >= d1) break;
        const bool in = x < d1;
        i64 vmax = INT64_MIN, vlen = 0;
        if (in) {
            bad |= en[r] < 0 || en[r] >= READ_END_MAX || st[r] < 0 || id[r] < 0;
            sread_store<RN>(B, x, st[r], en[r], id[r] | (pr[r] == 1 ? (int)0x80000000 : 0));
            vmax = en[r]; vlen = en[r] > st[r] ? en[r] - st[r] : 0;
            if (B.ro_mode == 0) {                           // the caller's promise: every block sorted by start
                hint = chrom_of_read(B, x, hint);
                if (x > B.reads_off[hint] && st[r] < (i64)B.r_start[x - 1]) unsorted = true;
            }
        }
        i64 cm, cl;
        if constexpr (RN) {                                 // (coordinates of an int32 table: 32-bit reductions)
            int a = (int)(vmax < 0 ? -1 : vmax), l = (int)vlen, t;
            t = dpp_i32<0x111, 0xf>(-1, a); a = t > a ? t : a;  t = dpp_i32<0x111, 0xf>(0, l); l = t > l ? t : l;
            t = dpp_i32<0x112, 0xf>(-1, a); a = t > a ? t : a;  t = dpp_i32<0x112, 0xf>(0, l); l = t > l ? t : l;
            t = dpp_i32<0x114, 0xf>(-1, a); a = t > a ? t : a;  t = dpp_i32<0x114, 0xf>(0, l); l = t > l ? t : l;
            t = dpp_i32<0x118, 0xf>(-1, a); a = t > a ? t : a;  t = dpp_i32<0x118, 0xf>(0, l); l = t > l ? t : l;
            t = dpp_i32<0x142, 0xa>(-1, a); a = t > a ? t : a;  t = dpp_i32<0x142, 0xa>(0, l); l = t > l ? t : l;
            t = dpp_i32<0x143, 0xc>(-1, a); a = t > a ? t : a;  t = dpp_i32<0x143, 0xc>(0, l); l = t > l ? t : l;
            cm = __builtin_amdgcn_readlane(a, 63); cl = __builtin_amdgcn_readlane(l, 63);
            if (cm < 0) cm = INT64_MIN;
        } else {
            cm = lane63_i64(wave_incl_max_i64(vmax)); cl = lane63_i64(wave_incl_max_i64(vlen));
        }
        const i64 f0 = readlane_i64x(st[r], 0);                                // (lane 0 of a chunk is always a row of the table)
        if (lane == 0) { ((rd_t<RN>*)B.cmax)[c0 >> 6] = (rd_t<RN>)cm; ((rd_t<RN>*)B.cfirst)[c0 >> 6] = (rd_t<RN>)f0; if (((c0 >> 6) & 63) == 0) ((rd_t<RN>*)B.bfirst)[c0 >> 12] = (rd_t<RN>)f0; }
        span_len = cl > span_len ? cl : span_len;
    }
    // (the verdict on the table belongs to the upload, like the table: the stage may run before the counters of a run are zeroed)
    if (bad) atomicOr(&B.rs->error, ERR_KEY_RANGE);
    if (unsorted) atomicOr(&B.rs->error, ERR_READS_UNSORTED);
    if (lane == 0) B.span_len[span] = span_len;             // longest read of the span (k_reads_maxlen reduces them per chromosome)
}

// longest read per chromosome from the spans' maxima (a span of 512 rows that straddles two blocks counts for both: the bound
// may only be too generous).  (One atomic maximum per span from the gather itself was tried: 12 000 atomics on two dozen
// addresses took the gather from 48 to 137 us; one chromosome per 128-byte line and an atomic only when a value read first is
// smaller - a few hundred per line, at the start of the kernel - still cost it 12 us, six times this kernel and its boundary.)
__global__ __launch_bounds__(256) void k_reads_maxlen(DevBatch B)
{
    if (reads_pending(B)) return;
    __shared__ i64 sh[4];
    for (int c = blockIdx.x; c < B.n_chrom; c += gridDim.x) {
        const i64 r0 = B.reads_off[c], r1 = B.reads_off[c + 1];
        i64 v = 0;
        if (r1 > r0) for (i64 k = (r0 >> 9) + threadIdx.x; k <= ((r1 - 1) >> 9); k += 256) { const i64 x = B.span_len[k]; v = x > v ? x : v; }
        v = lane63_i64(wave_incl_max_i64(v));
        __syncthreads();
        if (lane_id() == 0) sh[threadIdx.x >> 6] = v;
        __syncthreads();
        if (threadIdx.x == 0) { i64 m = sh[0]; for (int k = 1; k < 4; k++) m = sh[k] > m ? sh[k] : m; B.maxlen[c] = m; }
    }
}

// chromosome of every row (key column of the general sort)
__global__ __launch_bounds__(256) void k_reads_chromcol(DevBatch B, int* out)
{
    const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    if (i < B.n_reads) out[i] = chrom_of_read(B, i, 0);
}

// ------------------------------------------------------------------------------------ genotype
// One wavefront per call.  cover(window) = primary reads with 2*start <= L2 and 2*end >= R2 (doubled
// coordinates keep the x.5 windows of DUP/INV exact; GT:95-159, semantics as in GT.duipai :206-212).
// DR = distinct cover names that are not support names (GT:167-170): a hash set is seeded with the
// support read ids, then every covering read id is inserted; each fresh insert is one DR.
// Three tiers, no size limit (the reference has none, GT:95-159): k_genotype<1024, 4> (4 KB of LDS per wavefront,
// full occupancy) handles every call whose support + cover fits ~700 reads and appends the others to an overflow
// list; k_genotype<8192, 1> (one wavefront per workgroup, 32 KB) finishes those up to ~6000 reads; what is deeper
// still (chrM, rDNA and centromeric pile-ups) gets a table in GLOBAL memory sized from the scan extent: a slice of
// the pool per workgroup, and the whole pool - which holds any call of the batch - for the last workgroup standing.
// Insert `id` into the open-addressing set for the lanes that `want` it; 1: the id was new.  ONE wave-uniform loop whose per-lane
// state lives in vector registers (integers, not lane masks): a per-lane probe loop entered under a divergent branch costs ~19
// SCALAR instructions per probe round for
// its exec-mask bookkeeping, and the genotype kernel issued 1.75 scalar instructions per vector one: 27.5 M per 90x launch, which at
// one scalar instruction per SIMD every four cycles is 45 of its 60 us.  (cfg5: 60 -> 53 us; cfg4: 21 -> 19 us.)
__device__ __forceinline__ unsigned umin32(unsigned a, unsigned b) { return a < b ? a : b; }
// (`id` is never negative, also in the lanes that do not `want`)
template <int HASH> __device__ __forceinline__ int hash_insert_v(int* tab, int id, int want)
{
    unsigned h = ((unsigned)id * 2654435761u) >> (32 - __builtin_ctz(HASH));
    int ins = 0, pend = want;
    // (first round unconditionally: a loop that tests before it probes pays its scalar bookkeeping twice for the common single
    // round; a lane that sits a round out sees "the id is there", so that neither `ins` nor `pend` needs its mask AND-ed in)
    do {
        int old = id;
        if (pend) old = atomicCAS(&tab[h], -1, id);
        ins |= (old == -1) ? 1 : 0;
        pend = (int)umin32((unsigned)old + 1u, (unsigned)(old ^ id));       // 0: inserted (old == -1) or present (old == id)
        h = (h + (pend ? 1u : 0u)) & (HASH - 1);
    } while (__ballot(pend != 0));
    return ins;
}
__device__ __forceinline__ int hash_insert_n(int* tab, int bits, int id)      // runtime size 2^bits (global-memory tables)
{
    const unsigned mask = (1u << bits) - 1u;
    unsigned h = ((unsigned)id * 2654435761u) >> (32 - bits);
    for (;;) {
        const int old = atomicCAS(&tab[h], -1, id);
        if (old == -1) return 1;
        if (old == id) return 0;
        h = (h + 1) & mask;
    }
}

// The scan range of a window in the block [r0, r1) of start-ordered reads: top = last read with 2*start <= L2 (r0 - 1: none),
// bot = first read with 2*start >= R2 - 2*maxlen (no earlier read is long enough to reach R).  Both 64-ary searches advance
// together, one probe of each per round trip.
template <bool RN> __device__ __forceinline__ void window_range(const DevBatch& B, i64 r0, i64 r1, i64 L2, i64 R2, i64 maxlen, i64& bot, i64& top)
{
    const i64 F2 = R2 - 2 * maxlen;                       // a covering read starts at or after F2 / 2
    i64 lo_u = r0, hi_u = r1, lo_l = r0, hi_l = r1;        // upper bound of start <= L; lower bound of start >= F
    while (hi_u - lo_u > 64 || hi_l - lo_l > 64) {
        const i64 su = (hi_u - lo_u + 63) / 64, sl = (hi_l - lo_l + 63) / 64;
        const i64 iu = lo_u + (i64)lane_id() * su, il = lo_l + (i64)lane_id() * sl;
        const i64 vu = sread_start<RN>(B, iu < hi_u ? iu : lo_u), vl = sread_start<RN>(B, il < hi_l ? il : lo_l);
        if (hi_u - lo_u > 64) {
            const int t = __popcll(__ballot(iu < hi_u && 2 * vu <= L2));
            if (t == 0) hi_u = lo_u;
            else { const i64 nlo = lo_u + (i64)(t - 1) * su + 1; i64 nhi = lo_u + (i64)t * su; if (nhi > hi_u) nhi = hi_u; lo_u = nlo; hi_u = nhi; }
        }
        if (hi_l - lo_l > 64) {
            const int t = __popcll(__ballot(il < hi_l && 2 * vl < F2));
            if (t == 0) hi_l = lo_l;
            else { const i64 nlo = lo_l + (i64)(t - 1) * sl + 1; i64 nhi = lo_l + (i64)t * sl; if (nhi > hi_l) nhi = hi_l; lo_l = nlo; hi_l = nhi; }
        }
    }
    const i64 iu = lo_u + lane_id(), il = lo_l + lane_id();
    const i64 safe = B.n_reads - 1;                        // (an exhausted range may sit at the end of the table)
    const i64 vu = sread_start<RN>(B, iu < hi_u ? iu : (lo_u < safe ? lo_u : safe)), vl = sread_start<RN>(B, il < hi_l ? il : (lo_l < safe ? lo_l : safe));
    top = lo_u + __popcll(__ballot(iu < hi_u && 2 * vu <= L2)) - 1;
    bot = lo_l + __popcll(__ballot(il < hi_l && 2 * vl < F2));
}

// Cover of one window into the LDS set, in three round trips.  The packed table carries two small indices written by
// k_reads_gather: cfirst[c] = start of the first read of chunk c (64 reads) and bfirst[k] = cfirst[64 k] (a block of 4096
// reads).  (1) every lane probes bfirst: the block that holds the last read with start <= L; (2) the 128 chunks of that
// block and its predecessor, cfirst and cmax together: the chunks between the first read that could still reach R
// (start >= R - longest read) and the last with start <= L whose largest end reaches R; (3) the reads of those chunks,
// four chunks per step, tested exactly (primary, 2 start <= L2, 2 end >= R2).  The first form searched the exact row
// bounds with two 64-ary searches over the start column (four dependent probes of 64 scattered cache lines each) before it
// looked at any chunk: six round trips per window, and with 3-8 calls per wavefront the kernel is a chain of such trips.
constexpr int GT_UNROLL = 4;                         // (2: 5 % faster on 30x HiFi, 2 % slower on 90x ONT; 6 and 8: slower on both)
constexpr int GT_NC = 256;                          // chromosomes whose block offsets / longest reads k_genotype keeps in LDS
// the first step of (1) below, issued by the caller together with the call's other loads: first starts of the first 128 blocks
template <bool RN> __device__ __forceinline__ void bfirst_probe(const DevBatch& B, int r0, int r1, rd_t<RN>& fa, rd_t<RN>& fc)
{
    // (indices past the block are masked by the caller, not clamped: the index arrays are padded by two steps, and a uniform base
    // plus the lane number needs no address arithmetic)
    const rd_t<RN>* bf = (const rd_t<RN>*)B.bfirst + (r0 >> 12);
    fa = bf[lane_id()]; fc = bf[64 + lane_id()];
}
// Rows, chunks and blocks are 32-bit numbers (a table has fewer than 2^31 reads), and the window tests run on HALVED bounds
// in the table's own width: 2 start <= L2 <=> start <= L2 >> 1, 2 end >= R2 <=> end >= (R2 + 1) >> 1 - the kernel was
// bound by the issue of 64-bit compares and address arithmetic, not by memory (0.4 k vector instructions per call).
// -DCSV_GT_PROF (a measurement build, never the product): per-phase shader-clock sums of k_genotype's first pass, one add per
// wavefront and phase into the first words of the (then idle) global pool; k_gt_prof_print reports and clears them.
#ifdef CSV_GT_PROF
#define GT_TICK(k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long t__ = __builtin_readcyclecounter(); gt_prof[k] += t__ - gt_t; gt_t = t__; } while (0)
#define GT_PROF_ARGS , unsigned long long* gt_prof, unsigned long long& gt_t
#define GT_PROF_PASS , gt_prof, gt_t
__global__ void k_gt_prof_print(DevBatch B)
{
    unsigned long long* g = (unsigned long long*)B.gt_pool;     // (one 128-byte slot per workgroup modulo 1024: same-address atomics
    unsigned long long p[16];                                    // from 16 k wavefronts queue up at one L2 channel and slow every load)
    for (int k = 0; k < 16; k++) { p[k] = 0; for (int b = 0; b < 1024; b++) { p[k] += g[16 * b + k]; g[16 * b + k] = 0; } }
    printf("gt_prof waves %llu calls %llu chunks %llu steps2 %llu windows %llu | ticks: lifetime %llu = loop %llu next_head %llu chrom %llu probes+clear %llu supports %llu step1 %llu step2_wait %llu step2_alu %llu rows %llu result %llu\n",
           p[15], p[8], p[9], p[10], p[11], p[14], p[7], p[12], p[13], p[0], p[1], p[2], p[3], p[4], p[5], p[6]);

}
#else
#define GT_TICK(k) do { } while (0)
#define GT_PROF_ARGS
#define GT_PROF_PASS
#endif
// lowest set bit of a wave-uniform mask, cleared in place; -1 when the mask is empty (`m &= m - 1` is three scalar instructions
// and a compare on top of the find)
__device__ __forceinline__ int pick_bit(u64& m)
{
    int j;
    asm("s_ff1_i32_b64 %0, %1\n\ts_bitset0_b64 %1, %0" : "=&s"(j), "+s"(m));
    return j;
}
// a negative number unless a <= b (the sign of a saturating difference in the table's 32-bit form)
template <bool RN> __device__ __forceinline__ int neg_unless_le(rd_t<RN> a, rd_t<RN> b)
{
    if constexpr (RN) return __builtin_elementwise_sub_sat(b, a); else return a <= b ? 0 : -1;
}
template <int HASH, bool RN> __device__ __forceinline__ int cover_window(const DevBatch& B, int* tab, int r0, int r1, i64 L2, i64 R2, i64 maxlen, int& filled, bool& overflow,
                                                                         rd_t<RN> fa0, rd_t<RN> fc0 GT_PROF_ARGS)
{
    using CT = rd_t<RN>;
    const CT* bf = (const CT*)B.bfirst; const CT* cf = (const CT*)B.cfirst; const CT* cx = (const CT*)B.cmax;
    const int filled0 = filled;                             // (every insert below is one DR)
    const int lane = lane_id();
    const i64 Lh64 = L2 >> 1, Rh64 = (R2 + 1) >> 1, Fh64 = (R2 - 2 * maxlen + 1) >> 1;      // a covering read starts at or after Fh
    CT Lh, Rh, Fh;
    if constexpr (RN) {
        if (Rh64 > INT32_MAX) return 0;                     // (no int32 end reaches it)
        Lh = (int)(Lh64 > INT32_MAX ? INT32_MAX : Lh64);
        Rh = (int)(Rh64 < INT32_MIN ? INT32_MIN : Rh64);
        Fh = (int)(Fh64 < INT32_MIN ? INT32_MIN : Fh64);   // (Fh <= Rh)
    } else { Lh = Lh64; Rh = Rh64; Fh = Fh64; }
    const int c0 = r0 >> 6, c1 = (r1 - 1) >> 6, k0 = c0 >> 6, k1 = c1 >> 6;
    // The range and order tests of (1) and (2) are sign bits: a term is negative where its condition fails, terms are OR-ed (all
    // must hold) or AND-ed (one must hold) in the vector unit, and one compare per ballot reaches the scalar unit - written as
    // boolean expressions each ballot cost three to five scalar mask operations.
    // (1) last block whose first start is <= L (the block that holds the chromosome's first read counts as such: its own
    // first read may belong to the previous chromosome)
    int ktop = k0;
    for (int kb = k0; kb <= k1; kb += 128) {
        const int ka = kb + lane, kc = ka + 64;
        CT fa = fa0, fc = fc0;
        if (kb != k0) { fa = (bf + kb)[lane]; fc = (bf + kb)[64 + lane]; }
        // k <= k1 and (k == k0 or first <= L); k >= k0 here, so k0 - k is zero at k0 and negative after it
        const int ta = (k1 - ka) | ((k0 - ka) & neg_unless_le<RN>(fa, Lh)), tc = (k1 - kc) | ((k0 - kc) & neg_unless_le<RN>(fc, Lh));
        const int nn = __popcll(__ballot(ta >= 0)) + __popcll(__ballot(tc >= 0));
        if (nn == 0) break;                                 // (the predicate is true on a prefix: starts ascend inside a chromosome)
        ktop = kb + nn - 1;
        if (nn < 128) break;
    }
    GT_TICK(2);
#ifdef CSV_GT_PROF
    gt_prof[11]++;
#endif
    // (2) chunks, two blocks per step, walking towards the chromosome's first chunk until the scan range is closed
    int top_chunk = -1;
    for (int kk = ktop; kk >= k0; kk -= 2) {
        const int two = kk > k0 ? 1 : 0;                    // blocks kk - 1 and kk, or the chromosome's first block alone
        const int cb = (kk - two) << 6;                     // first chunk of the step
        const int lo_c = cb > c0 ? cb : c0;
        int hi_c = cb + 63 + (two << 6); if (hi_c > c1) hi_c = c1;
        const int ca = cb + lane, cc = ca + 64;
        // (a chunk number is below 2^25: its byte offset fits 32 bits, and the column's base stays one scalar pair)
        const unsigned oa = (unsigned)ca * (unsigned)sizeof(CT), oc = oa + 64u * (unsigned)sizeof(CT);
        const CT fa = *(const CT*)((const char*)cf + oa), fc = *(const CT*)((const char*)cf + oc);
        const CT ma = *(const CT*)((const char*)cx + oa), mc = *(const CT*)((const char*)cx + oc);
        GT_TICK(3);
#ifdef CSV_GT_PROF
        gt_prof[10]++;
#endif
        if (kk == ktop) {                                   // the chunk of the last read with start <= L lies in this step
            const int ta = (ca - lo_c) | (hi_c - ca) | ((c0 - ca) & neg_unless_le<RN>(fa, Lh));
            const int tc = (cc - lo_c) | (hi_c - cc) | ((c0 - cc) & neg_unless_le<RN>(fc, Lh));
            const int nn = __popcll(__ballot(ta >= 0)) + __popcll(__ballot(tc >= 0));
            if (nn == 0) return 0;                          // no read of the chromosome starts at or before L
            top_chunk = lo_c + nn - 1;
        }
        // The scan's first chunk is the LAST chunk that begins before F (it may still hold later reads; the chromosome's first
        // chunk may begin in its predecessor): chunks of the step other than c0 whose first start is below F ...
        const int ba = (ca - lo_c) | (hi_c - ca) | (ca - c0 - 1) | ~neg_unless_le<RN>(Fh, fa);
        const int bc = (cc - lo_c) | (hi_c - cc) | (cc - c0 - 1) | ~neg_unless_le<RN>(Fh, fc);
        const u64 before_a = __ballot(ba >= 0), before_c = __ballot(bc >= 0);
        const bool any_before = (before_a | before_c) != 0;
        const bool closed = any_before || lo_c == c0;
        const int bot_chunk = before_c ? cb + 127 - __clzll((long long)before_c) : (before_a ? cb + 63 - __clzll((long long)before_a) : lo_c);
        // ... and every chunk from there to top_chunk that holds a read reaching R
        const int lo_t = bot_chunk, hi_t = top_chunk < hi_c ? top_chunk : hi_c;            // (bot_chunk >= lo_c)
        const int qa = (ca - lo_t) | (hi_t - ca) | neg_unless_le<RN>(Rh, ma), qc = (cc - lo_t) | (hi_t - cc) | neg_unless_le<RN>(Rh, mc);
        const u64 todo_a = CSV_ABL(26) ? 0ull : __ballot(qa >= 0), todo_c = CSV_ABL(26) ? (u64)(__popcll(__ballot(qc >= 0) | __ballot(qa >= 0)) == 64) : __ballot(qc >= 0);
        // (3) the reads of the flagged chunks, four chunks in flight.  The scalar unit was this kernel's busiest (1.1 scalar
        // instructions per vector one, ~43 per chunk here): a chunk is now picked by two scalar instructions, a column is read at
        // ONE base per step of (2) plus a 32-bit byte offset, the row tests are sign bits of differences OR-ed together in the
        // vector unit instead of five lane masks AND-ed in the scalar unit.
        GT_TICK(4);
#ifdef CSV_GT_PROF
        gt_prof[9] += __popcll(todo_a) + __popcll(todo_c);
#endif
        const char* st_b; const char* en_b;
        if constexpr (RN) { st_b = (const char*)(B.s_start32 + ((i64)cb << 6)); en_b = (const char*)(B.s_end32 + ((i64)cb << 6)); }
        else { st_b = (const char*)(B.s_start64 + ((i64)cb << 6)); en_b = (const char*)(B.s_end64 + ((i64)cb << 6)); }
        const char* id_b = (const char*)(B.s_idp + ((i64)cb << 6));
        const int nrow1 = r1 - r0 - 1;
#pragma unroll 1
        for (int half = 0; half < 2; half++) {
            u64 m = half ? todo_c : todo_a;
            if (!m) continue;
            const int rel_h = lane + (cb << 6) - r0 + (half << 12);        // row - r0 of this lane in chunk 0 of the half
            const unsigned off_h = ((unsigned)lane << 2) + ((unsigned)half << 14);
            while (m) {
                CT st[GT_UNROLL], en[GT_UNROLL]; int idp[GT_UNROLL], jj[GT_UNROLL];
                const int ng = __popcll(m);                  // chunks left in this half: the group's slots past them are skipped
#pragma unroll
                for (int u = 0; u < GT_UNROLL; u++) {
                    if (u > 0 && u >= ng) break;
                    jj[u] = pick_bit(m);
                    const unsigned o4 = ((unsigned)jj[u] << 8) + off_h;
                    st[u] = *(const CT*)(st_b + (RN ? o4 : 2u * o4)); en[u] = *(const CT*)(en_b + (RN ? o4 : 2u * o4));
                    idp[u] = *(const int*)(id_b + o4);
                }
#pragma unroll
                for (int u = 0; u < GT_UNROLL; u++) {
                    if (u > 0 && u >= ng) break;
                    if (filled + 64 > HASH * 3 / 4) { overflow = true; return 0; }
                    const int rowrel = (jj[u] << 6) + rel_h;
                    // negative unless: the row is one of the chromosome's, primary (bit 31 of idp), starts at or before L, reaches R
                    const int t = rowrel | __builtin_elementwise_sub_sat(nrow1, rowrel) | ~idp[u] | neg_unless_le<RN>(st[u], Lh) | neg_unless_le<RN>(Rh, en[u]);
                    const int cov = t >= 0 ? 1 : 0;
                    const int ins = CSV_ABL(17) ? cov : hash_insert_v<HASH>(tab, idp[u] & 0x7fffffff, cov);
                    filled += __popcll(__ballot(ins));
                }
            }
        }
        GT_TICK(5);
        if (closed) break;
    }
    return filled - filled0;
}

__device__ __forceinline__ int gl_index_dev(i64 c0, i64 c1)
{
    if (c0 == 3 && c1 == 1) return 101 * 101;
    if (c0 == 6 && c1 == 2) return 101 * 101 + 1;
    const i64 total = c0 + c1;
    if (total > 100) {
        const double frac = (double)c0 / (double)total;
        c0 = (i64)(100.0 * frac);
        c1 = 100 - c0;
    }
    return (int)(c0 * 101 + c1);
}

// what one call needs before it touches the reads table; loaded one call ahead (the per-call work is a chain of
// dependent round trips, this takes the first ones off it)
struct GtHead { int c; int4 h; i64 s0, s1, search, b1, b2; };
// The record is read through the CONSTANT address space: the fields read here were written by earlier kernels and do not change
// while this one runs (its own result goes to bytes 64-79 of the record, which nothing reads this way), `c` is wave-uniform, and
// the kernel is bound by the vector unit's issue rate - as ordinary loads these come back in vector registers (the kernel's own
// stores could alias them, so the compiler may not use the scalar cache) and everything derived from them - windows, halved
// bounds, block and chunk ranges: ~50 instructions of 64-bit arithmetic per call - runs on the vector unit with 64 equal lanes.
typedef int gt_v4i __attribute__((ext_vector_type(4)));
typedef const gt_v4i __attribute__((address_space(4))) * gt_rec_cptr;
__device__ __forceinline__ void gt_load_call(const DevBatch& B, int c, GtHead& H)
{
    gt_rec_cptr r = (gt_rec_cptr)(uintptr_t)&B.o_rec[c];
    const gt_v4i r0 = r[0], r1 = r[1], r2 = r[2], r5 = r[5];
    H.c = c; H.h = make_int4(r5.x, r5.y, r5.z, r5.w);
    H.b1 = ((i64)r0.y << 32) | (unsigned)r0.x; H.b2 = ((i64)r0.w << 32) | (unsigned)r0.z;
    H.search = ((i64)r1.y << 32) | (unsigned)r1.x;
    H.s0 = ((i64)r2.y << 32) | (unsigned)r2.x; H.s1 = H.s0 + r2.z;
}
__device__ __forceinline__ void gt_load_head(const DevBatch& B, int second, int q, GtHead& H)
{
    gt_load_call(B, second ? B.gt_over[q] : q, H);
}

// the one or two windows of a call in doubled coordinates (INDEL:450-451; DUP:146-151; INV:218-221)
struct GtWin { int n; i64 La, Ra, Lb, Rb; };              // (scalars, not arrays: a dynamically indexed private array is scratch memory)
__device__ __forceinline__ GtWin gt_windows(const GtHead& H)
{
    GtWin W;
    const int svtype = H.h.y & 0xff;
    const i64 gt_bias = ((i64)H.h.w << 32) | (unsigned)H.h.z;
    if (svtype == CSV_DEL || svtype == CSV_INS) {
        const i64 p = H.search, g = gt_bias;
        i64 L = p - g; if (L < 0) L = 0;
        W.n = 1; W.La = 2 * L; W.Ra = 2 * (p + g); W.Lb = 0; W.Rb = 0;
        return W;
    }
    i64 nb = gt_bias;
    if (svtype == CSV_DUP && H.b2 - H.b1 < nb) nb = H.b2 - H.b1;            // DUP:147
    W.n = 2;
    W.La = 2 * H.b1 - nb; if (W.La < 0) W.La = 0; W.Ra = 2 * H.b1 + nb;
    W.Lb = 2 * H.b2 - nb; if (W.Lb < 0) W.Lb = 0; W.Rb = 2 * H.b2 + nb;     // union of both: DUP:155-157
    return W;
}

// A call of any depth with its set in global memory; executed by `nthreads` threads of one workgroup (64: one
// wavefront with its slice of the pool; 64 * WPB = the whole workgroup with the whole pool).  Returns false when the
// table the call needs does not fit `cap` ints (nothing has been written then).
template <bool RN> __device__ bool genotype_global(const DevBatch& B, const GtHead& H, int* tab, i64 cap, int tid, int nthreads, bool whole_block, int* s_red)
{
    const int chrom = H.h.x;
    const i64 r0 = B.reads_off[chrom], r1 = B.reads_off[chrom + 1], maxlen = B.maxlen[chrom];
    const i64 ns = H.s1 - H.s0;
    const GtWin W = gt_windows(H);
    i64 need = ns;
    i64 bota, topa, botb = r0, topb = r0 - 1;
    window_range<RN>(B, r0, r1, W.La, W.Ra, maxlen, bota, topa);
    if (topa >= bota) need += topa - bota + 1;
    if (W.n == 2) {
        window_range<RN>(B, r0, r1, W.Lb, W.Rb, maxlen, botb, topb);
        if (topb >= botb) need += topb - botb + 1;
    }
    int bits = 10;
    while ((1ll << bits) < 2 * need) bits++;
    if ((1ll << bits) > cap || bits > 31) return false;
    const i64 T = 1ll << bits;
    for (i64 i = tid; i < T; i += nthreads) tab[i] = -1;
    if (whole_block) __syncthreads(); else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    for (i64 i = tid; i < ns; i += nthreads) hash_insert_n(tab, bits, B.o_suprid[H.s0 + i]);
    if (whole_block) __syncthreads(); else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    int dr = 0;
    for (i64 i = bota + tid; i <= topa; i += nthreads) { const int idp = B.s_idp[i]; if (idp < 0 && 2 * sread_end<RN>(B, i) >= W.Ra) dr += hash_insert_n(tab, bits, idp & 0x7fffffff); }
    for (i64 i = botb + tid; i <= topb; i += nthreads) { const int idp = B.s_idp[i]; if (idp < 0 && 2 * sread_end<RN>(B, i) >= W.Rb) dr += hash_insert_n(tab, bits, idp & 0x7fffffff); }
    dr = wave_sum_i32(dr);
    if (whole_block) {
        if (lane_id() == 0) s_red[tid >> 6] = dr;
        __syncthreads();
        dr = 0;
        for (int k = 0; k < nthreads / 64; k++) dr += s_red[k];
        __syncthreads();
    }
    if (tid == 0) ((int4*)&B.o_rec[H.c])[4] = make_int4(dr, (int)ns, gl_index_dev(dr, ns), 0);
    return true;
}

// SECOND is a template parameter so that the first pass - the one every call goes through - does not carry the code and the
// registers of the global-pool path (as a run-time argument: 28 scalar spills and 16 bytes of scratch per lane in the hot
// kernel; cfg-4 56.7 -> 40.6 us, cfg-5 136 -> 116 us)
template <int HASH, int WPB, bool SECOND, bool RN> __global__ __launch_bounds__(64 * WPB) __attribute__((amdgpu_num_sgpr(96))) void k_genotype(DevBatch B)
{
    constexpr int second = SECOND ? 1 : 0;
    __shared__ int tabs[WPB][HASH];
    __shared__ int s_red[WPB], s_last;
    // the chromosomes' blocks of the reads table and their longest reads, once per workgroup (a dependent look-up per call otherwise)
    __shared__ int s_off[GT_NC + 1];
    __shared__ int s_ml[GT_NC][2];                   // (as two words: read as an i64 the compiler merges this read and the global one of the
                                                     // other branch into ONE flat load through a selected pointer)
    int* tab = tabs[threadIdx.x >> 6];
    const bool in_lds = B.n_chrom <= GT_NC;
    if (in_lds) {
        for (int i = threadIdx.x; i <= B.n_chrom; i += 64 * WPB) s_off[i] = (int)B.reads_off[i];
        for (int i = threadIdx.x; i < B.n_chrom; i += 64 * WPB) { const i64 ml = B.maxlen[i]; s_ml[i][0] = (int)ml; s_ml[i][1] = (int)(ml >> 32); }
    }
    // (calls dealt to the XCDs in bands of 256 neighbours instead of launch order - one L2 per XCD - changed neither the time nor,
    // with it, the story: 50.3 vs 50.5 us)
    const int wave = __builtin_amdgcn_readfirstlane((blockIdx.x * (64 * WPB) + threadIdx.x) >> 6), nwaves = (gridDim.x * (64 * WPB)) >> 6;
    GtHead cur, nxt;
    // (the first call's record is fetched before the number of calls is known - the index is clamped to the table -: one round
    // trip less at the head of every wavefront)
    if constexpr (!SECOND) gt_load_call(B, wave < B.cap_tmp ? wave : B.cap_tmp, cur);
    const bool pending = reads_pending(B);
    const int n = second ? B.cnt->n_gt_over : B.cnt->n_calls;
    __syncthreads();
    if (pending) return;
    if constexpr (SECOND) {
        // tell the host how many calls overflowed the first pass: a later run of the same upload (same calls, same table) skips
        // this launch when the answer is none (second page-locked word, as k_chain_apply's for the refine tiers)
        if (B.host_flag && blockIdx.x == 0 && threadIdx.x == 0)
            __hip_atomic_store((unsigned long long*)B.host_flag + 1, ((unsigned long long)(unsigned)B.run_seq << 32) | (unsigned)n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (second && n == 0) return;                                   // (nothing overflowed the first pass: no list, no hand-over)
    if constexpr (SECOND) { if (wave < n) gt_load_head(B, second, wave, cur); }
#ifdef CSV_GT_PROF
    unsigned long long gt_prof[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long gt_t0 = __builtin_readcyclecounter();
    unsigned long long gt_t = gt_t0;
#endif
    for (int q = wave; q < n; q += nwaves, cur = nxt) {
        GT_TICK(7);
        nxt = cur;
        if (q + nwaves < n) gt_load_head(B, second, q + nwaves, nxt);
        const int c = cur.c, svtype = cur.h.y & 0xff, chrom = cur.h.x;
        if (!(cur.h.y & 0x100) || svtype == CSV_TRA) continue;    // TRA: k_genotype_tra
#ifdef CSV_GT_PROF
        gt_prof[8]++;
#endif
        GT_TICK(12);
        int r0, r1; i64 maxlen;
        if (in_lds) { r0 = s_off[chrom]; r1 = s_off[chrom + 1]; maxlen = ((i64)s_ml[chrom][1] << 32) | (unsigned)s_ml[chrom][0]; }
        else { r0 = (int)B.reads_off[chrom]; r1 = (int)B.reads_off[chrom + 1]; maxlen = B.maxlen[chrom]; }
        // one round trip: the first 128 supports and the block probes of the window(s); the table is cleared while they fly.
        // (Issuing this trip one call ahead, under the previous call's chunk loads, was measured twice - before and after the
        // instruction diet - and changed nothing: 50.0 vs 50.5 us on the 90x workload; see DESIGN.md on what bounds the kernel.)
        const GtWin W = gt_windows(cur);
        const i64 s0 = cur.s0; const int ns = (int)(cur.s1 - s0);       // (a call's support count is an int: gt_load_call)
        rd_t<RN> fa = 0, fc = 0;
        GT_TICK(13);
        if (r1 > r0) bfirst_probe<RN>(B, r0, r1, fa, fc);
        int sup0 = lane_id() < ns ? B.o_suprid[s0 + lane_id()] : 0;
        int sup1 = lane_id() + 64 < ns ? B.o_suprid[s0 + 64 + lane_id()] : 0;       // (90x: most calls have more than 64 supports)
        {
            int4* t4 = (int4*)tab;
#pragma unroll
            for (int k = 0; k < HASH / 256; k++) t4[k * 64 + lane_id()] = make_int4(-1, -1, -1, -1);
        }
        // (the loaded values are first looked at here: without this the compiler waits for each load right where it is issued)
        asm volatile("" : "+v"(fa), "+v"(fc), "+v"(sup0), "+v"(sup1));
        int filled = 0;
        bool overflow = false;
        GT_TICK(0);
        for (int base = 0; base < ns; base += 64) {
            if (filled + 64 > HASH * 3 / 4) { overflow = true; break; }
            const int i = base + lane_id();
            int ins = 0;
            if (!CSV_ABL(18)) ins = hash_insert_v<HASH>(tab, base == 0 ? sup0 : (base == 64 ? sup1 : (i < ns ? B.o_suprid[s0 + i] : 0)), i < ns ? 1 : 0);
            filled += __popcll(__ballot(ins));
        }
        int dr = 0;
        GT_TICK(1);
        if (!overflow && !CSV_ABL(16) && r1 > r0) {
            dr = cover_window<HASH, RN>(B, tab, r0, r1, W.La, W.Ra, maxlen, filled, overflow, fa, fc GT_PROF_PASS);
            if (W.n == 2 && !overflow) dr += cover_window<HASH, RN>(B, tab, r0, r1, W.Lb, W.Rb, maxlen, filled, overflow, fa, fc GT_PROF_PASS);      // (same chromosome: same probes)
        }
        if (overflow) {                                                       // wave-uniform
            if constexpr (!SECOND) { if (lane_id() == 0) B.gt_over[atomicAdd(&B.cnt->n_gt_over, 1)] = c; continue; }
            else {
                // deeper than the 32 KB tables: this wavefront's slice of the global pool ...
                const i64 slice = B.gt_pool_n / nwaves;
                if (!genotype_global<RN>(B, cur, B.gt_pool + (i64)wave * slice, slice, lane_id(), 64, false, s_red) && lane_id() == 0)
                    B.gt_huge[atomicAdd(&B.cnt->n_gt_huge, 1)] = c;           // ... or, later, the whole pool
                continue;
            }
        }
        if (lane_id() == 0) ((int4*)&B.o_rec[c])[4] = make_int4(dr, (int)ns, gl_index_dev(dr, ns), 0);
        GT_TICK(6);
    }
#ifdef CSV_GT_PROF
    if constexpr (!SECOND) { if (lane_id() == 0) { gt_prof[14] = __builtin_readcyclecounter() - gt_t0; gt_prof[15] = 1; for (int k = 0; k < 16; k++) atomicAdd((unsigned long long*)B.gt_pool + 16 * (blockIdx.x & 1023) + k, gt_prof[k]); } }
#endif
    if constexpr (!SECOND) return;
    // the last workgroup to get here owns the whole pool (every other one is done with its slice) and finishes the
    // calls that needed more than a slice, one at a time
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        s_last = atomicAdd(&B.cnt->gt_ticket, 1) == (int)gridDim.x - 1;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (!s_last) return;
    const int nh = __hip_atomic_load(&B.cnt->n_gt_huge, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int q = 0; q < nh; q++) {
        GtHead H;
        gt_load_call(B, __hip_atomic_load(&B.gt_huge[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), H);
        if (!genotype_global<RN>(B, H, B.gt_pool, B.gt_pool_n, threadIdx.x, 64 * WPB, WPB > 1, s_red) && threadIdx.x == 0)
            atomicOr(&B.cnt->error, ERR_COVER_OVERFLOW);                      // (cannot happen: the pool is sized for any call of the batch)
    }
}

// ---------------------------------------------------------------------------------------- TRA genotyping
// call_gt / count_coverage of cuteSV_resolveTRA.py:258-309 and cuteSV_genotype.py:62-93 with the reads table
// as the alignment stream (include/cutesv_hip.h).  One wavefront per TRA call.  The reference's loop is
// sequential with two early exits (querydata reaches up_bound; iteration reaches gt_round at a primary
// read); here a chunk of 64 reads is classified at once, the running counters become ballot prefixes and the
// first lane at which either exit fires ends the scan; only lanes up to it are committed to the set.
//   table entry flags: 1 = read supports the call (read_id_list), 2 = read is in querydata
// The set lives in LDS (4096 slots) and, for calls with more than ~3000 supports + spanning reads, in the global pool
// (same slice / whole-pool scheme as k_genotype).
constexpr int TG_HASH = 4096;                        // slots per wavefront (ids + flags: 32 KB of LDS)

// first index in [lo, hi) at which pred turns false (pred is true on a prefix); 64-ary search
template <class P> __device__ __forceinline__ i64 partition_point_wave(i64 lo, i64 hi, P pred)
{
    while (hi - lo > 64) {
        const i64 step = (hi - lo + 63) / 64;
        const i64 idx = lo + (i64)lane_id() * step;
        const int t = __popcll(__ballot(idx < hi && pred(idx < hi ? idx : lo)));
        if (t == 0) return lo;
        i64 nhi = lo + (i64)t * step;
        if (nhi > hi) nhi = hi;
        lo = lo + (i64)(t - 1) * step + 1; hi = nhi;
    }
    const i64 idx = lo + lane_id();
    return lo + __popcll(__ballot(idx < hi && pred(idx < hi ? idx : lo)));
}

// slot of id in the table of 2^bits entries, inserting it (flags 0) when absent; `fresh` = 1 when this call created the entry
__device__ __forceinline__ int tg_find_or_insert(int* ids, int bits, int id, int& fresh)
{
    const unsigned mask = (1u << bits) - 1u;
    unsigned h = ((unsigned)id * 2654435761u) >> (32 - bits);
    for (;;) {
        const int old = atomicCAS(&ids[h], -1, id);
        if (old == -1) { fresh = 1; return (int)h; }
        if (old == id) { fresh = 0; return (int)h; }
        h = (h + 1) & mask;
    }
}

__device__ __forceinline__ i64 tra_up_bound(i64 num)     // threshold_ref_count, cuteSV_genotype.py:62-70
{
    if (num <= 2) return 20 * num;
    if (num <= 5) return 9 * num;
    if (num <= 15) return 7 * num;
    return 5 * num;
}

// one count_coverage() call; returns the status (0 / 1 / -1).  nq / dr / filled are wave-uniform running totals.
template <bool RN> __device__ __forceinline__ int tra_window(const DevBatch& B, int* ids, int* fl, int bits, int chrom, i64 s, i64 e, i64 up_bound, i64 itround,
                                                             i64& nq, int& dr, int& filled, bool& overflow)
{
    if (s >= e) return 0;
    const i64 limit = (3ll << bits) / 4;
    const i64 r0 = B.reads_off[chrom], r1 = B.reads_off[chrom + 1], maxlen = B.maxlen[chrom];
    if (r1 <= r0) return 0;
    const i64 hi = partition_point_wave(r0, r1, [&](i64 i) { return sread_start<RN>(B, i) < e; });          // fetch(): start < e ...
    const i64 lo = partition_point_wave(r0, hi, [&](i64 i) { return sread_start<RN>(B, i) + maxlen <= s; }); // ... and end > s (no earlier read is long enough)
    i64 iteration = 0, primary = 0;
    const u64 le = lanemask_lt() | (1ull << lane_id());
    for (i64 base = lo; base < hi; base += 64) {
        if (filled + 64 > limit) { overflow = true; return 0; }
        const i64 i = base + lane_id();
        const bool in = i < hi;
        const i64 ii = in ? i : lo;
        const i64 rs = sread_start<RN>(B, ii), re = sread_end<RN>(B, ii);
        const int idp = B.s_idp[ii], id = idp & 0x7fffffff;
        const bool ov = in && re > s;                                          // GT:76-77
        const bool prim = ov && idp < 0;                                       // GT:78-80
        const bool span = prim && rs < s && re > e;                            // GT:81
        // a name that occurs twice among the chunk's spanning reads counts at its first occurrence
        bool dup = false;
        for (u64 m = __ballot(span); m;) {
            const int j = __builtin_amdgcn_readfirstlane(__builtin_ctzll(m));
            m &= m - 1;
            const int idj = __builtin_amdgcn_readlane(id, j);
            if (span && lane_id() > j && id == idj) dup = true;
        }
        int fresh = 0, slot = 0;
        if (span && !dup) slot = tg_find_or_insert(ids, bits, id, fresh);
        filled += __popcll(__ballot(fresh));
        const int flags = (span && !dup && !fresh) ? fl[slot] : 0;
        const bool isnew = span && !dup && !(flags & 2);
        const u64 m_ov = __ballot(ov), m_pr = __ballot(prim), m_new = __ballot(isnew);
        const i64 it_i = iteration + __popcll(m_ov & le), pn_i = primary + __popcll(m_pr & le), nq_i = nq + __popcll(m_new & le);
        const bool exitA = span && nq_i >= up_bound;                           // GT:83-85
        const bool exitB = prim && it_i >= itround;                            // GT:86-91
        const u64 stop = __ballot(exitA || exitB);
        const int t = stop ? __builtin_ctzll(stop) : 63;
        const bool commit = isnew && lane_id() <= t;
        if (commit) fl[slot] = flags | 2;
        dr += __popcll(__ballot(commit && !(flags & 1)));
        if (stop) {
            nq = __builtin_amdgcn_readlane((int)nq_i, t);
            const int a_t = __builtin_amdgcn_readlane((int)exitA, t);
            const i64 it_t = __builtin_amdgcn_readlane((int)it_i, t), pn_t = __builtin_amdgcn_readlane((int)pn_i, t);
            if (a_t) return 1;
            return (5 * pn_t <= it_t) ? 1 : -1;                                // float(primary_num / iteration) <= 0.2
        }
        nq += __popcll(m_new); iteration += __popcll(m_ov); primary += __popcll(m_pr);
    }
    return 0;
}

// one TRA call with its set in ids / fl (2^bits slots each); returns false when the set overflowed (nothing written)
template <bool RN> __device__ bool tra_call(const DevBatch& B, int c, int* ids, int* fl, int bits)
{
    const CallRec rec = B.o_rec[c];
    const csv_segment& sg = B.seg[rec.seg];
    const int chr1 = sg.chrom, chr2 = rec.aux >> 3;
    if (chr2 < 0 || chr2 >= B.n_chrom) { if (lane_id() == 0) atomicOr(&B.cnt->error, ERR_TRA_CHROM); return true; }
    const i64 T = 1ll << bits, limit = (3ll << bits) / 4;
    for (i64 i = lane_id(); i < T; i += 64) { ids[i] = -1; fl[i] = 0; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    const i64 s0 = rec.supoff, ns = rec.support;
    int filled = 0;
    bool overflow = false;
    for (i64 base = 0; base < ns && !overflow; base += 64) {           // read_id_list: flag 1
        if (filled + 64 > limit) { overflow = true; break; }
        const i64 i = base + lane_id();
        int fresh = 0;
        if (i < ns) { const int slot = tg_find_or_insert(ids, bits, B.o_suprid[s0 + i], fresh); fl[slot] = 1; }
        filled += __popcll(__ballot(fresh));
    }
    const i64 up_bound = tra_up_bound(ns);                             // TRA:266
    const i64 bias = sg.gt_bias;
    i64 nq = 0;
    int dr = 0, status = 0;
    if (!overflow) {
        i64 s = rec.bp1 - bias, e = rec.bp1 + bias;                   // TRA:263-264
        if (s < 0) s = 0;
        if (e > B.contig_len[chr1]) e = B.contig_len[chr1];
        status = tra_window<RN>(B, ids, fl, bits, chr1, s, e, up_bound, sg.gt_round, nq, dr, filled, overflow);
        if (status == 0 && !overflow) {                               // TRA:289-299 (status_2 is not looked at)
            s = rec.bp2 - bias; e = rec.bp2 + bias;
            if (s < 0) s = 0;
            if (e > B.contig_len[chr2]) e = B.contig_len[chr2];
            tra_window<RN>(B, ids, fl, bits, chr2, s, e, up_bound, sg.gt_round, nq, dr, filled, overflow);
        }
    }
    if (overflow) return false;
    if (lane_id() == 0)                                                 // status -1: TRA:276-281
        ((int4*)&B.o_rec[c])[4] = status == -1 ? make_int4(-1, (int)ns, -1, 0) : make_int4(dr, (int)ns, gl_index_dev(dr, ns), 0);
    return true;
}

// table bits that surely hold a TRA call: supports + at most up_bound querydata names per window + one chunk
__device__ __forceinline__ int tra_bits_for(i64 ns)
{
    const i64 need = ns + 2 * (tra_up_bound(ns) + 64) + 128;
    int bits = 10;
    while ((3ll << bits) / 4 < need + 64) bits++;
    return bits;
}

template <bool RN> __global__ __launch_bounds__(64) void k_genotype_tra(DevBatch B)
{
    __shared__ int ids[TG_HASH];
    __shared__ int fl[TG_HASH];
    __shared__ int s_last;
    if (reads_pending(B)) return;
    const int n = B.cnt->n_calls;
    const i64 slice = B.gt_pool_n / gridDim.x;          // (the pool is free: k_genotype ran before this kernel)
    for (int c0 = blockIdx.x * 64; c0 < n; c0 += gridDim.x * 64) {
        // 64 calls per round: the lanes look for genotyped TRA calls, the wavefront then takes them one by one
        const int cl = c0 + lane_id();
        bool mine = false;
        if (cl < n) mine = B.o_rec[cl].type_gt == (CSV_TRA | 0x100);
        for (u64 todo = __ballot(mine); todo;) {
            const int c = c0 + __builtin_amdgcn_readfirstlane(__builtin_ctzll(todo));
            todo &= todo - 1;
            if (tra_call<RN>(B, c, ids, fl, 12)) continue;
            // more names than the LDS set holds: this workgroup's slice of the global pool, or the whole pool later
            const int bits = tra_bits_for(B.o_rec[c].support);
            if ((2ll << bits) <= slice && bits <= 30) {
                int* g = B.gt_pool + (i64)blockIdx.x * slice;
                if (!tra_call<RN>(B, c, g, g + (1ll << bits), bits) && lane_id() == 0) atomicOr(&B.cnt->error, ERR_COVER_OVERFLOW);
            } else if (lane_id() == 0) B.gt_huge[atomicAdd(&B.cnt->n_tra_huge, 1)] = c;
        }
    }
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        s_last = atomicAdd(&B.cnt->tra_ticket, 1) == (int)gridDim.x - 1;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (!s_last) return;
    const int nh = __hip_atomic_load(&B.cnt->n_tra_huge, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int q = 0; q < nh; q++) {
        const int c = __hip_atomic_load(&B.gt_huge[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int bits = tra_bits_for(B.o_rec[c].support);
        if ((2ll << bits) > B.gt_pool_n || bits > 30 || !tra_call<RN>(B, c, B.gt_pool, B.gt_pool + (1ll << bits), bits))
            if (lane_id() == 0) atomicOr(&B.cnt->error, ERR_COVER_OVERFLOW);   // (cannot happen: the pool is sized for any call of the batch)
    }
}

}  // namespace csv
