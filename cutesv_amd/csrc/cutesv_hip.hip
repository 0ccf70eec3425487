// cutesv_hip.hip — host side of libcutesv_hip.so: context, device arena, the C ABI of
// include/cutesv_hip.h and the launch sequence of one batch.  gfx950 only.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <map>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "kernels.hip.h"
#include "sort.hip.h"
#include "cigar.hip.h"
#include "split.hip.h"

using namespace csv;

namespace {

// Page-locked host ranges handed out (or registered) through this library, process wide: {base -> bytes, device address}.
// csv_batch_download writes results straight into such memory from a kernel; a range leaves the table before it is freed,
// so a table hit is always live memory (addresses pinned by other means are asked about through the HIP runtime each time).
struct PinnedRange { size_t bytes; char* dev; bool owned; };
std::mutex g_pinned_mu;
std::map<uintptr_t, PinnedRange> g_pinned;
void pinned_note(void* p, size_t bytes, bool owned)
{
    void* d = nullptr;
    if (hipHostGetDevicePointer(&d, p, 0) != hipSuccess) { (void)hipGetLastError(); return; }
    std::lock_guard<std::mutex> lk(g_pinned_mu);
    g_pinned[(uintptr_t)p] = PinnedRange{bytes, (char*)d, owned};
}
void pinned_forget(void* p)
{
    std::lock_guard<std::mutex> lk(g_pinned_mu);
    g_pinned.erase((uintptr_t)p);
}
// device address of host pointer p (with `bytes` behind it) if it is page-locked, else nullptr.  Memory from csv_host_alloc
// is owned by this library (it leaves the table in csv_host_free, before it is released): a hit there is live memory.  A
// range the CALLER registered (csv_host_register) may have been freed or re-used without csv_host_unregister - a kernel
// writing through such a stale mapping would fault the GPU - so those hits, like unknown addresses, are confirmed with
// the HIP runtime on every use.
void* pinned_device_address(const void* p, size_t bytes)
{
    {
        std::lock_guard<std::mutex> lk(g_pinned_mu);
        auto it = g_pinned.upper_bound((uintptr_t)p);
        if (it != g_pinned.begin()) {
            --it;
            const uintptr_t off = (uintptr_t)p - it->first;
            if (off < it->second.bytes) {
                if (off + bytes > it->second.bytes) return nullptr;
                if (it->second.owned) return it->second.dev + off;
            }
        }
    }
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (at.type != hipMemoryTypeHost || !at.devicePointer) return nullptr;
    if (bytes > 1) {                                            // the last byte must belong to the same page-locked range
        hipPointerAttribute_t a2;
        if (hipPointerGetAttributes(&a2, (const char*)p + bytes - 1) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        if (a2.type != hipMemoryTypeHost || (char*)a2.devicePointer != (char*)at.devicePointer + bytes - 1) return nullptr;
    }
    return at.devicePointer;
}

struct Buf {                      // a slice of an arena (or, for the few stand-alone buffers, its own allocation)
    void*  p = nullptr;
    size_t cap = 0;
};

// One device allocation for everything a batch needs: the buffers are planned (sizes -> offsets), the arena grows
// only when the plan does not fit, and every Buf becomes a pointer into it.  (The first version reserved ~90 buffers
// with one hipMalloc each: 2.7 ms on the first upload, and a larger batch re-allocated them one by one.)
struct Arena {
    char*  base = nullptr;
    size_t cap = 0;
};
struct Plan {
    std::vector<std::pair<Buf*, size_t>> items;
    size_t total = 0;
    void add(Buf& b, size_t bytes)
    {
        items.emplace_back(&b, total);
        b.cap = bytes;
        total += (bytes + 255) & ~(size_t)255;
    }
};

// one timing slot per launch (group), in launch order
const char* kStageName[CSV_N_STAGES] = {"init", "k_chain_count", "k_chain_apply", "k_refine_indel_wave", "k_refine_wave", "k_refine_mid",
                                        "k_refine_block", "k_items_scan", "k_emit", "k_reads_order", "k_reads_gather", "k_reads_maxlen",
                                        "k_genotype", "k_genotype_tra", "event_floor", "", "", "", "", "", "", "", "", ""};
constexpr int N_COPY_STREAMS = 2;
// workgroups of k_genotype<1024,4>: about two resident sets - calls differ a lot in cost, and workgroups that start as others
// finish even the tail out (measured on the 90x workload: 1536..2048 -> 82-86 us, 4096..8192 -> 76 us, 16384 -> 80 us)
constexpr int GT_GRID = 4096;
constexpr int RO_CAP = 4096;                 // sorted runs the reads_order stage plans (k_reads_plan packs the rank in 12 bits)

}  // namespace

struct csv_ctx {
    int         device = 0;
    hipStream_t stream = nullptr;
    hipStream_t side[3] = {};         // side streams: [0] mid + workgroup tier, [1] DUP/INV/TRA wavefront tier, [2] reads order + prefix max
    hipStream_t copy[N_COPY_STREAMS] = {};    // host -> device column copies (one DMA engine each)
    hipEvent_t  ev_init = nullptr, ev_sel = nullptr, ev_aux[3] = {}, ev_copy[N_COPY_STREAMS] = {}, ev_reads = nullptr, ev_anc = nullptr, ev_rd[5] = {};
    std::string err;
    hipEvent_t  ev[CSV_N_STAGES + 2] = {};
    Arena       arena, arena_rb;
    // batch buffers (slices of `arena`)
    Buf seg, woff, seg_drop, a, b, rid, aux, a32, b32;
    Buf tile_lead, tabs;
    Buf ad16, anc;                             // CSV_IN_SIG_DELTA16: the gaps in w space; the anchor tables {per-tile offsets, w, value}
    Buf rd16, ranc, rl16, rlesc;               // CSV_IN_READS_DELTA16: start gaps + their anchors, lengths + their escape rows / values
    int  reads_delta = 0;                      // bit 0: the last upload's reads starts crossed as gaps, bit 1: its ends as lengths (csv_batch_info 3)
    bool delta16 = false;                      // the last upload rebuilt its position column from gaps (csv_batch_info 2)
    bool rstate_dirty = true;                  // the reads-order state may hold an earlier upload's verdict
    bool reads_early = false;                  // the last upload decoded the reads table's start column on side[1] (upload_impl)
    bool unpack_pending = false; UnpackArgs unpack_args{}; int unpack_tiles = 0;      // ... and k_unpack_a16 is still to be queued (one-shot calls: by the run)
    Buf cluster_id, partial, tile_cnt, item_rec, list_small, list_big, list_tiny, list_wide, seg_gate, tile_info, ch_masks, tile_items, seg_err;
    Buf item_cnt, item_base, item_chunk, sup_tmp;
    Buf t_rec, t_rec0;
    Buf sc_k, sc_x, sc_v1, sc_v2, sc_v3, sc_v4, sc_v5;
    Buf o_rec, o_supsig, o_suprid, allele_id;
    Buf o_rec2, o_supsig2;                     // the second result arena (runs alternate: a publish may still read the other one)
    Buf reads_off, r_start, r_end, r_primary, r_id, s_start, s_end, s_idp, cmax, cfirst, bfirst, span_len, maxlen, gt_over, gt_huge, gt_pool, contig_len;
    Buf ro_tcnt, ro_ent, ro_table, ro_tblk;
    std::vector<int> h_tblk;                   // per tile of the reads table: the first chromosome block that begins at or after it
    // stand-alone
    Buf sqrt_tab, rcp_tab, cipk_tab, cnt, rstate;
    Buf gs_chrom, gs_perm0, gs_perm1, gs_hist, gs_tot;          // general reads sort (fallback), allocated on first use
    Buf flush;                                                   // csv_cache_flush scratch
    // rebuild step (slices of `arena_rb`)
    Buf rb_seg, rb_a, rb_b, rb_rid, rb_aux, rb_auxk, rb_major, rb_nodedup, rb_perm0, rb_perm1, rb_hist, rb_tot, rb_partial;
    Buf rb_oseg, rb_oa, rb_ob, rb_orid, rb_oaux, rb_osrc, rb_segcnt, rb_rank, rb_mx, rb_drop, rb_el0, rb_el1;
    // the device-resident signature pool (stand-alone allocations: it outlives the per-call arenas)
    Buf pool_seg, pool_a, pool_b, pool_read, pool_aux, sp_qlen;
    i64 pool_n = 0, pool_cap = 0;
    // CIGAR scan (slices of `arena_rb` as well: the two steps never overlap)
    Buf sp_off, sp_len, sp_c0, sp_c1, sp_f0, sp_f1, sp_chr, sp_mapq, sp_strand, sp_primary, sp_seg, sp_cnt, sp_tiles, sp_tot,
        sp_kind, sp_read, sp_ochr, sp_aux, sp_a, sp_b, sp_c, sp_d;
    Buf cg_qlen, cg_off, cg_ops, cg_start, cg_use, cg_cnt, cg_tiles, cg_tot, cg_iread, cg_ipos, cg_ilen, cg_ip0, cg_inp, cg_pq, cg_pl, cg_dread, cg_dpos, cg_dlen;
    // page-locked host staging: small tables on the way in, counters + call records + support lists on the way out
    char*  h_pin = nullptr;
    size_t h_pin_cap = 0;
    // two page-locked 64-bit words the device writes {run sequence, count}: items above 64 signatures (k_chain_apply), calls that
    // overflowed the first genotype pass (k_genotype<8192>); read when a LATER run of the same upload is planned
    // CSV_* environment switches of the run path (timing / debugging aids), read once per upload: csv_batch_run - a 36 us
    // step - looks nothing up in the environment
    struct RunOpts {
        bool debug = false, debug_counters = false, no_fork = false, fork_always = false, no_swap = false, no_peek = false;
        bool no_pair_in_mid = false, no_publish = false;
        int  iw_grid = 0, gt_grid = 0, tier_fork_min = 1 << 30, mid_grid = 0, big_grid = 0;
        bool pub_inplace = false, no_reads_overlap = false;
    } opt;
    volatile int* h_flag = nullptr;
    int*          d_flag = nullptr;
    int           run_seq = 0;
    int           upload_seq0 = 0;          // run_seq when the resident batch was uploaded: later sequence numbers are runs of it
    int           n_cu = 256;              // compute units of the device
    // host copies
    std::vector<csv_segment> h_seg;
    std::vector<i64>         h_woff;
    bool     uploaded = false, ran = false, any_genotype = false, any_pair = false, any_tra_gt = false, lds_set = false;
    bool     reads_ready = false;              // the packed start-ordered reads table of this upload exists (a completed reads stage)
    bool     reuse_reads = true;               // ... and resident re-runs keep it (csv_batch_option CSV_OPT_REUSE_READS_ORDER)
    bool     have_tab = false;                 // this upload issued copies of the reads table frame (reads_off, contig_len, columns) on side[2]
    bool     reads_general = false;            // this batch's reads table needs the general sort (found out by a first run)
    i64      sqrt_n = 0;                       // entries of sqrt_tab (grown to the longest segment seen: an allele is never larger)
    bool     copies_pending = false;           // csv_cluster_batch: the column copies are still in flight behind ev_copy[0] / [1]
    // pipelined delivery (csv_batch_publish_async): runs alternate between two result arenas {call records, support list,
    // counters}; the k_publish of run k reads arena k & 1 on its own stream while run k + 1 fills the other one
    hipStream_t pub = nullptr;
    hipEvent_t  ev_run[2] = {}, ev_pub[2] = {};
    int         parity = 0;                    // arena of the last run
    struct Pend { csv_batch_out* out = nullptr; bool live = false; } pend[2];
    int         pend_order[2] = {0, 0}, n_pend = 0;      // arenas with a publish in flight, oldest first
    bool        settled = false;               // a run of this upload has been downloaded synchronously (reads mode final, capacities known)
    char*       h_pub = nullptr;               // page-locked landing zones of the asynchronous publishes: 2 x {counters 256 B, status words}
    size_t      h_pub_cap = 0;
    // block delivery: when the caller's result arrays sit back to back in page-locked memory (at most PUB_MAX_SPANS runs of
    // adjacent arrays), k_publish writes them into a device image of those runs and the copy engine moves each run in one piece
    void*       pub_stage[2] = {nullptr, nullptr};
    size_t      pub_stage_cap[2] = {0, 0};
    bool     lazy_pending = false;             // gate-first call: this upload's first run still has to fetch the gated rows from the caller's columns
    bool     partial_cols = false;             // ... and its device columns hold only the rows the kernels read (csv_batch_validate refuses)
    i64      lazy_bytes = 0;                   // bytes the bulk copy of this upload did NOT send (measurement aid: csv_batch_lazy_info)
    i64      n_sig_host = 0, n_reads = 0;
    DevBatch B;
    DevCounters h_cnt;
};

namespace {

int fail(csv_ctx* c, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    return code;
}

#define HIP_TRY(c, call)                                                                                   \
    do {                                                                                                   \
        hipError_t e_ = (call);                                                                            \
        if (e_ != hipSuccess) return fail((c), CSV_E_HIP, "%s failed: %s", #call, hipGetErrorString(e_));   \
    } while (0)

int reserve(csv_ctx* c, Buf& b, size_t bytes)            // stand-alone grow-only buffer
{
    if (bytes <= b.cap) return CSV_OK;
    if (b.p) { HIP_TRY(c, hipFree(b.p)); b.p = nullptr; b.cap = 0; }
    size_t want = bytes + bytes / 4 + 256;
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) { b.p = nullptr; return fail(c, CSV_E_NOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e)); }
    b.cap = want;
    return CSV_OK;
}

int commit(csv_ctx* c, Arena& A, const Plan& P)
{
    if (P.total > A.cap) {
        if (A.base) { HIP_TRY(c, hipFree(A.base)); A.base = nullptr; A.cap = 0; }
        const size_t want = P.total + P.total / 8 + 4096;
        void* p = nullptr;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) return fail(c, CSV_E_NOMEM, "hipMalloc(%zu) for the batch arena failed: %s", want, hipGetErrorString(e));
        A.base = (char*)p; A.cap = want;
    }
    for (const auto& it : P.items) it.first->p = A.base + it.second;
    return CSV_OK;
}

int pin_reserve(csv_ctx* c, size_t bytes)
{
    if (bytes <= c->h_pin_cap) return CSV_OK;
    if (c->h_pin) { HIP_TRY(c, hipHostFree(c->h_pin)); c->h_pin = nullptr; c->h_pin_cap = 0; }
    const size_t want = bytes + bytes / 4 + 4096;
    void* p = nullptr;
    hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
    if (e != hipSuccess) return fail(c, CSV_E_NOMEM, "hipHostMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    c->h_pin = (char*)p; c->h_pin_cap = want;
    return CSV_OK;
}

template <class T> T* dp(const Buf& b) { return (T*)b.p; }

int div_up(i64 a, i64 b) { return (int)((a + b - 1) / b); }

int env_int(const char* name, int dflt)
{
    const char* v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}

void load_run_opts(csv_ctx* c)
{
    auto& o = c->opt;
    o.debug = getenv("CSV_DEBUG") != nullptr;
    o.debug_counters = o.debug || getenv("CSV_DEBUG_COUNTERS") != nullptr;
    o.no_fork = getenv("CSV_NO_FORK") != nullptr;
    o.fork_always = getenv("CSV_FORK_ALWAYS") != nullptr;
    o.no_swap = getenv("CSV_NO_SWAP") != nullptr;
    o.no_peek = getenv("CSV_NO_PEEK") != nullptr;
    o.no_pair_in_mid = getenv("CSV_NO_PAIR_IN_MID") != nullptr;
    o.no_publish = getenv("CSV_NO_PUBLISH") != nullptr;
    o.iw_grid = env_int("CSV_IW_GRID", 0);
    o.gt_grid = env_int("CSV_GT_GRID", 0);
    o.tier_fork_min = env_int("CSV_TIER_FORK_MIN", 1 << 30);
    o.mid_grid = env_int("CSV_MID_GRID", 0);
    o.big_grid = env_int("CSV_BIG_GRID", 0);
    o.pub_inplace = getenv("CSV_PUB_INPLACE") != nullptr;
    o.no_reads_overlap = getenv("CSV_NO_READS_OVERLAP") != nullptr;
}

int upload_impl(csv_ctx* c, const csv_batch_in* in, bool per_sig_forced, bool sync, bool lazy_ok);
int read_counters(csv_ctx* c);

// `num ** 0.5` of cal_CIPOS is libm pow(), not sqrt() (GT:59; the two differ for 271 integers below 300 000): the device reads
// a table built with the host libm that covers every n an allele of the batch can have (n <= its segment's length).  Grown,
// never shrunk; a batch whose longest segment fits the table costs nothing here.
int sqrt_table(csv_ctx* c, i64 n)
{
    if (n <= c->sqrt_n) return CSV_OK;
    const i64 want = n + n / 4;
    std::vector<double> tab((size_t)want), rcp((size_t)want);
    std::vector<float> cipk((size_t)want);
    for (i64 i = 0; i < want; i++) {
        tab[(size_t)i] = pow((double)i, 0.5);
        rcp[(size_t)i] = i ? 1.0 / (double)i : 0.0;                                   // IEEE division: correctly rounded
        cipk[(size_t)i] = i ? (float)(1.96 / ((double)i * tab[(size_t)i])) : 0.0f;
    }
    HIP_TRY(c, hipDeviceSynchronize());                       // (a kernel of an earlier batch may still read the old tables)
    Buf* tb[3] = {&c->sqrt_tab, &c->rcp_tab, &c->cipk_tab};
    const void* src[3] = {tab.data(), rcp.data(), cipk.data()};
    const size_t esz[3] = {sizeof(double), sizeof(double), sizeof(float)};
    for (int q = 0; q < 3; q++) {
        if (tb[q]->p) { HIP_TRY(c, hipFree(tb[q]->p)); tb[q]->p = nullptr; tb[q]->cap = 0; }
        const int rc = reserve(c, *tb[q], (size_t)want * esz[q]);
        if (rc) return rc;
        HIP_TRY(c, hipMemcpy(tb[q]->p, src[q], (size_t)want * esz[q], hipMemcpyHostToDevice));
    }
    c->sqrt_n = want;
    return CSV_OK;
}

}  // namespace

extern "C" {

int csv_abi_version(void) { return CSV_ABI_VERSION; }

int csv_struct_size(int which)
{
    switch (which) {
    case 0: return (int)sizeof(csv_segment);
    case 1: return (int)sizeof(csv_batch_in);
    case 2: return (int)sizeof(csv_batch_out);
    case 3: return (int)sizeof(csv_run_stats);
    case 4: return (int)sizeof(csv_rebuild_in);
    case 5: return (int)sizeof(csv_rebuild_out);
    case 6: return (int)sizeof(csv_vcf_in);
    case 7: return (int)sizeof(csv_rows_in);
    case 8: return (int)sizeof(csv_cigar_in);
    case 9: return (int)sizeof(csv_cigar_out);
    case 10: return (int)sizeof(csv_split_in);
    case 11: return (int)sizeof(csv_split_out);
    default: return -1;
    }
}

const char* csv_stage_name(int s) { return (s >= 0 && s < CSV_N_STAGES) ? kStageName[s] : ""; }

int csv_device_count(int* n)
{
    int k = 0;
    hipError_t e = hipGetDeviceCount(&k);
    if (n) *n = (e == hipSuccess) ? k : 0;
    return e == hipSuccess ? CSV_OK : CSV_E_HIP;
}

int csv_device_info(int device_id, char* pci_bus_id, int cap, int* n_cu)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device_id < 0 || device_id >= n) return CSV_E_HIP;
    if (pci_bus_id && cap > 0) { pci_bus_id[0] = 0; if (hipDeviceGetPCIBusId(pci_bus_id, cap, device_id) != hipSuccess) return CSV_E_HIP; }
    if (n_cu && hipDeviceGetAttribute(n_cu, hipDeviceAttributeMultiprocessorCount, device_id) != hipSuccess) return CSV_E_HIP;
    return CSV_OK;
}

int32_t csv_gl_index(int64_t c0, int64_t c1)
{
    if (c0 == 3 && c1 == 1) return 101 * 101;
    if (c0 == 6 && c1 == 2) return 101 * 101 + 1;
    const int64_t total = c0 + c1;
    if (total > 100) {
        const double frac = (double)c0 / (double)total;
        c0 = (int64_t)(100.0 * frac);
        c1 = 100 - c0;
    }
    return (int32_t)(c0 * 101 + c1);
}

int csv_host_alloc(int64_t bytes, void** out)
{
    if (!out || bytes < 0) return CSV_E_INVALID;
    *out = nullptr;
    void* p = nullptr;
    const size_t n = (size_t)(bytes > 0 ? bytes : 1);
    if (hipHostMalloc(&p, n, hipHostMallocPortable) != hipSuccess) return CSV_E_NOMEM;
    pinned_note(p, n, true);
    *out = p;
    return CSV_OK;
}
void csv_host_free(void* p) { if (p) { pinned_forget(p); (void)hipHostFree(p); } }
int csv_host_register(void* p, int64_t bytes)
{
    if (!p || bytes <= 0) return CSV_E_INVALID;
    if (hipHostRegister(p, (size_t)bytes, hipHostRegisterPortable) != hipSuccess) return CSV_E_HIP;
    pinned_note(p, (size_t)bytes, false);
    return CSV_OK;
}
int csv_host_unregister(void* p) { if (p) pinned_forget(p); return (p && hipHostUnregister(p) == hipSuccess) ? CSV_OK : CSV_E_HIP; }

int csv_ctx_create(int device_id, csv_ctx** out)
{
    if (!out) return CSV_E_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device_id < 0 || device_id >= n) return CSV_E_HIP;
    csv_ctx* c = new csv_ctx();
    c->device = device_id;
    if (hipSetDevice(device_id) != hipSuccess || hipStreamCreate(&c->stream) != hipSuccess) { delete c; return CSV_E_HIP; }
    if (hipDeviceGetAttribute(&c->n_cu, hipDeviceAttributeMultiprocessorCount, device_id) != hipSuccess || c->n_cu < 1) c->n_cu = 256;
    for (auto& e : c->ev) if (hipEventCreate(&e) != hipSuccess) { delete c; return CSV_E_HIP; }
    {
        // side[2] carries the reads stage, whose critical path runs through two single-workgroup kernels (k_reads_plan,
        // k_pmax_scan): at default priority they wait for a free CU behind the clustering kernels of the main stream
        // (27 and 15 us in the trace for a few microseconds of work), so that stream gets the highest priority
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        for (int q = 0; q < 3; q++) {
            const hipError_t e = (q == 2 && !getenv("CSV_NO_PRIORITY")) ? hipStreamCreateWithPriority(&c->side[q], hipStreamNonBlocking, greatest)
                                                                          : hipStreamCreateWithFlags(&c->side[q], hipStreamNonBlocking);
            if (e != hipSuccess) { delete c; return CSV_E_HIP; }
        }
    }
    for (auto& s2 : c->copy) if (hipStreamCreateWithFlags(&s2, hipStreamNonBlocking) != hipSuccess) { delete c; return CSV_E_HIP; }
    if (hipStreamCreateWithFlags(&c->pub, hipStreamNonBlocking) != hipSuccess) { delete c; return CSV_E_HIP; }
    for (int q = 0; q < 2; q++)
        if (hipEventCreateWithFlags(&c->ev_run[q], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->ev_pub[q], hipEventDisableTiming) != hipSuccess) { delete c; return CSV_E_HIP; }
    if (hipEventCreateWithFlags(&c->ev_init, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_sel, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_reads, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_anc, hipEventDisableTiming) != hipSuccess) { delete c; return CSV_E_HIP; }
    for (auto& e : c->ev_aux) if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { delete c; return CSV_E_HIP; }
    for (auto& e : c->ev_rd) if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { delete c; return CSV_E_HIP; }
    for (auto& e : c->ev_copy) if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { delete c; return CSV_E_HIP; }
    {
        void* hf = nullptr;
        if (hipHostMalloc(&hf, 64, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess) {
            void* df = nullptr;
            if (hipHostGetDevicePointer(&df, hf, 0) == hipSuccess) { c->h_flag = (volatile int*)hf; c->d_flag = (int*)df; memset(hf, 0, 64); }
            else (void)hipHostFree(hf);
        }
    }
    if (reserve(c, c->cnt, 1024) || reserve(c, c->rstate, sizeof(ReadsState)) || pin_reserve(c, 1 << 20) || sqrt_table(c, SQRT_TAB)) {
        delete c;
        return CSV_E_HIP;
    }
    *out = c;
    return CSV_OK;
}

void csv_ctx_destroy(csv_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    Buf* own[] = {&c->pool_seg, &c->pool_a, &c->pool_b, &c->pool_read, &c->pool_aux, &c->sp_qlen, &c->sqrt_tab, &c->rcp_tab, &c->cipk_tab, &c->cnt, &c->rstate, &c->gs_chrom, &c->gs_perm0, &c->gs_perm1, &c->gs_hist, &c->gs_tot, &c->flush};
    for (Buf* b : own) if (b->p) (void)hipFree(b->p);
    if (c->arena.base) (void)hipFree(c->arena.base);
    if (c->arena_rb.base) (void)hipFree(c->arena_rb.base);
    if (c->h_pin) (void)hipHostFree(c->h_pin);
    if (c->h_flag) (void)hipHostFree((void*)c->h_flag);
    if (c->h_pub) (void)hipHostFree(c->h_pub);
    for (auto& ps : c->pub_stage) if (ps) (void)hipFree(ps);
    for (int q = 0; q < 2; q++) { if (c->ev_run[q]) (void)hipEventDestroy(c->ev_run[q]); if (c->ev_pub[q]) (void)hipEventDestroy(c->ev_pub[q]); }
    if (c->pub) (void)hipStreamDestroy(c->pub);
    for (auto& e : c->ev) if (e) (void)hipEventDestroy(e);
    for (auto& e : c->ev_aux) if (e) (void)hipEventDestroy(e);
    for (auto& e : c->ev_copy) if (e) (void)hipEventDestroy(e);
    if (c->ev_init) (void)hipEventDestroy(c->ev_init);
    if (c->ev_sel) (void)hipEventDestroy(c->ev_sel);
    if (c->ev_reads) (void)hipEventDestroy(c->ev_reads);
    if (c->ev_anc) (void)hipEventDestroy(c->ev_anc);
    for (auto& e : c->ev_rd) if (e) (void)hipEventDestroy(e);
    for (auto& s2 : c->side) if (s2) (void)hipStreamDestroy(s2);
    for (auto& s2 : c->copy) if (s2) (void)hipStreamDestroy(s2);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char* csv_last_error(const csv_ctx* c) { return c ? c->err.c_str() : "null context"; }

int csv_ctx_sync(csv_ctx* c)
{
    if (!c) return CSV_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return CSV_OK;
}

int csv_batch_upload(csv_ctx* c, const csv_batch_in* in) { return upload_impl(c, in, false, true, false); }

int csv_batch_reads_mode(const csv_ctx* c) { return (c && c->uploaded && c->n_reads > 0) ? c->B.ro_mode : -1; }

int csv_batch_info(const csv_ctx* c, int which, int64_t* value)
{
    if (!c || !value) return CSV_E_INVALID;
    if (which == 0) *value = c->partial_cols ? 1 : 0;
    else if (which == 1) *value = c->lazy_bytes;
    else if (which == 2) *value = c->delta16 ? 1 : 0;
    else if (which == 3) *value = c->reads_delta;
    else return CSV_E_INVALID;
    return CSV_OK;
}

int csv_batch_option(csv_ctx* c, int option, int value)
{
    if (!c) return CSV_E_INVALID;
    if (option == CSV_OPT_REUSE_READS_ORDER) { c->reuse_reads = value != 0; return CSV_OK; }
    return fail(c, CSV_E_INVALID, "unknown option %d", option);
}

}  // extern "C"

namespace {

// Host -> device.  The small tables travel as ONE copy out of the page-locked staging block; the columns go out on up
// to four copy streams (one DMA engine each), the reads table on its own so that the clustering kernels never wait
// for it.  `sync` = false (csv_cluster_batch): nothing waits here, the kernels are ordered behind the copies by events
// and the final download synchronises before the call returns.
int upload_impl(csv_ctx* c, const csv_batch_in* in, bool per_sig_forced, bool sync, bool lazy_ok)
{
    if (!c || !in) return CSV_E_INVALID;
    if (c->n_pend) { (void)hipStreamSynchronize(c->pub); c->n_pend = 0; c->pend[0].live = c->pend[1].live = false; }      // (results nobody waited for)
    c->uploaded = c->ran = false; c->settled = false; c->parity = 0;
    c->lazy_pending = c->partial_cols = false; c->lazy_bytes = 0;
    c->reads_general = false;
    c->reads_ready = false;
    c->upload_seq0 = c->run_seq;
    load_run_opts(c);
    HIP_TRY(c, hipSetDevice(c->device));
    if (in->n_seg < 0 || in->n_sig < 0 || (in->n_seg > 0 && !in->seg)) return fail(c, CSV_E_INVALID, "bad batch header");
    // (support lists and seq_pick name signatures by their global index in 32 bits on the device)
    if (in->n_sig >= (1ll << 31)) return fail(c, CSV_E_INVALID, "n_sig = %lld: a batch indexes at most 2^31 - 1 signature rows (split the store)", (long long)in->n_sig);
    const int S = in->n_seg;
    c->h_seg.assign(in->seg, in->seg + S);
    c->h_woff.assign(S + 1, 0);
    c->any_genotype = false;
    c->any_pair = false;
    c->any_tra_gt = false;
    const bool have_reads_off = in->reads_off != nullptr;
    if (have_reads_off) {                                   // the reads table is trusted by the kernels: check its frame here
        if (in->n_chrom < 0 || in->n_reads < 0 || in->n_reads >= (1ll << 31) - 4096) return fail(c, CSV_E_INVALID, "bad reads table header");
        if (in->reads_off[0] < 0) return fail(c, CSV_E_INVALID, "reads_off[0] is negative");
        for (int k = 0; k < in->n_chrom; k++)
            if (in->reads_off[k + 1] < in->reads_off[k]) return fail(c, CSV_E_INVALID, "reads_off decreases at chromosome %d", k);
        if (in->reads_off[in->n_chrom] > in->n_reads) return fail(c, CSV_E_INVALID, "reads_off[n_chrom] exceeds n_reads");
    }
    std::vector<uint8_t> drop(S + 1, 0);
    i64 cap_items = 16, cap_tmp = 16, maxseg_gt = 0, tra_gt_len = 0;
    for (int k = 0; k < S; k++) {
        const csv_segment& g = c->h_seg[k];
        if (g.svtype < CSV_DEL || g.svtype > CSV_TRA) return fail(c, CSV_E_INVALID, "segment %d: unknown svtype %d", k, g.svtype);
        if (g.sig_begin < 0 || g.sig_begin > g.sig_end || g.sig_end > in->n_sig) return fail(c, CSV_E_INVALID, "segment %d: bad signature range", k);
        if (g.genotype) {
            c->any_genotype = true;
            if (g.chrom < 0 || g.chrom >= in->n_chrom) return fail(c, CSV_E_INVALID, "segment %d: chrom %d outside the reads table", k, g.chrom);
            if (g.svtype == CSV_TRA) {
                // call_gt of cuteSV_resolveTRA.py:258-309 over the reads table; no "no reads block" gate there
                if (!in->reads_off || !in->contig_len) return fail(c, CSV_E_INVALID, "segment %d: TRA genotyping needs reads_off and contig_len", k);
                c->any_tra_gt = true;
            } else {
                drop[k] = (!in->reads_off || in->reads_off[g.chrom + 1] == in->reads_off[g.chrom]) ? 1 : 0;
            }
        }
        const i64 len = g.sig_end - g.sig_begin;
        if (g.genotype && g.svtype == CSV_TRA && len > tra_gt_len) tra_gt_len = len;
        if (g.genotype && g.svtype != CSV_TRA && len > maxseg_gt) maxseg_gt = len;
        if (len > 0 && g.svtype != CSV_DEL && g.svtype != CSV_INS) c->any_pair = true;
        c->h_woff[k + 1] = c->h_woff[k] + len;
        const i64 rc = g.read_count > 1 ? g.read_count : 1;
        const i64 msr = g.min_support_reads > 1 ? g.min_support_reads : 1;
        cap_items += len / rc + 1;
        if (g.svtype == CSV_DEL || g.svtype == CSV_INS) cap_tmp += len / msr + 1;
        else if (g.svtype == CSV_TRA) cap_tmp += 2 * (len / rc) + 2;
        else cap_tmp += len / rc + 1;
    }
    const i64 W = c->h_woff[S];
    {
        i64 longest = 0;
        for (int k = 0; k < S; k++) if (c->h_woff[k + 1] - c->h_woff[k] > longest) longest = c->h_woff[k + 1] - c->h_woff[k];
        const int rc = sqrt_table(c, longest + 2);
        if (rc) return rc;
    }
    if (W >= (1ll << 31) - 4096 || cap_tmp >= (1ll << 31) - 1) return fail(c, CSV_E_INVALID, "batch too large for 32-bit work indices (%lld signatures)", (long long)W);
    if (c->any_genotype && in->reads_off && (!in->r_start || !in->r_end || !in->r_primary || !in->r_id) && in->n_reads > 0)
        return fail(c, CSV_E_INVALID, "reads columns missing");
    const bool per_sig = per_sig_forced || (in->flags & CSV_IN_PER_SIG);
    const bool dev_cols = (in->flags & CSV_IN_DEVICE_COLUMNS) != 0;       // a / b / read_id / aux are device pointers: device-to-device copies
    const hipMemcpyKind col_kind = dev_cols ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    // Gate-first (one-shot calls only: the caller's columns are valid until the call returns): when b / read_id / aux live in
    // page-locked memory the device can read, only the position column travels in bulk and k_lazy_fetch pulls the rows of the
    // clusters that pass the size gate (kernels.hip.h).  Small batches are cheaper in one piece (CSV_LAZY_MIN signatures,
    // default 64 Ki: below that the extra kernel and its PCIe round trips cost more than the bytes they save).
    const bool sig32_ = (in->flags & CSV_IN_SIG_I32) != 0;
    const void *lz_b = nullptr, *lz_rid = nullptr, *lz_aux = nullptr, *lz_rows8 = nullptr;
    bool lazy = lazy_ok && !dev_cols && W > 0 && !getenv("CSV_NO_LAZY") && W >= (i64)env_int("CSV_LAZY_MIN", 64 << 10) && in->b && in->read_id && in->aux;
    if (lazy) {
        const size_t nb = (size_t)in->n_sig;
        // (ABI v8) {b, read_id} interleaved, page-locked: the fetch reads one array instead of two
        if (sig32_ && in->rows8 && !getenv("CSV_NO_ROWS8")) lz_rows8 = pinned_device_address(in->rows8, nb * 8);
        if (!lz_rows8) {
            lz_b = pinned_device_address(in->b, nb * (sig32_ ? 4 : 8));
            lz_rid = lz_b ? pinned_device_address(in->read_id, nb * 4) : nullptr;
        }
        lz_aux = (lz_rows8 || lz_rid) ? pinned_device_address(in->aux, nb * 4) : nullptr;
        lazy = lz_aux != nullptr;
    }

    // CSV_IN_SIG_DELTA16: the position column as 16-bit gaps + anchors (kernels.hip.h k_unpack_a16).  Needs disjoint segments (an
    // escape row belongs to one w): anything else takes the column itself.
    bool delta16 = sig32_ && !dev_cols && W > 0 && (in->flags & CSV_IN_SIG_DELTA16) && in->a_delta && in->a && in->n_esc >= 0 &&
                   (in->n_esc == 0 || (in->a_esc_row && in->a_esc_val)) && !getenv("CSV_NO_DELTA16") && W >= (i64)env_int("CSV_DELTA16_MIN", 32 << 10) &&
                   // a sparse column (sites far apart: a HiFi call set, the simulation beds) is mostly escapes: each costs the host an
                   // anchor (search + sort: 50 k of them 1.1 ms, measured on cfg4) and saves nothing - the column itself then
                   in->n_esc * (i64)env_int("CSV_DELTA16_ESC", 64) <= in->n_sig;
    std::vector<std::pair<i64, int>> by_begin;            // non-empty segments by their first source row
    if (delta16) {
        for (int k = 0; k < S; k++) if (c->h_woff[k + 1] > c->h_woff[k]) by_begin.emplace_back(c->h_seg[k].sig_begin, k);
        std::sort(by_begin.begin(), by_begin.end());
        for (size_t i = 0; i + 1 < by_begin.size(); i++)
            if (c->h_seg[by_begin[i].second].sig_end > by_begin[i + 1].first) { delta16 = false; break; }
    }
    const i64 n_anc_cap = delta16 ? (div_up(W, CH_TILE) + S + in->n_esc + 8) : 0;
    c->delta16 = delta16;

    // ---- device memory: one plan, one arena
    const i64 R = (c->any_genotype && in->reads_off) ? in->n_reads : 0;
    const bool reorder = R > 0 && !(in->flags & CSV_IN_READS_SORTED);
    const bool have_tab = c->any_genotype && in->reads_off;
    const i64 nt = div_up(W, CH_TILE) + 2;                 // chain tiles
    const i64 SC = 2 * W + 16 + 2 * ARR_PAD;
    i64 rc_max = 0;                                        // largest reads block: bounds the cover set of one call
    if (R > 0) for (int k = 0; k < in->n_chrom; k++) { const i64 d = in->reads_off[k + 1] - in->reads_off[k]; if (d > rc_max) rc_max = d; }
    // global hash pool (ints, a power of two): holds the set of ANY call of the batch - supports <= its segment, cover <= two
    // scans of a reads block (genotype_global: table < 4 * need), TRA: < 59 * supports + 1710 ints (tra_bits_for)
    i64 pool_n = 1 << 20;
    if (R > 0) while (pool_n < 2 * (2 * rc_max + maxseg_gt) + 4096 || pool_n < 64 * tra_gt_len + 8192) { pool_n <<= 1; if (pool_n >= (1ll << 32)) break; }
    Plan P;
#define PL(buf, bytes) P.add(c->buf, (size_t)(bytes))
    const bool sig32 = (in->flags & CSV_IN_SIG_I32) != 0, rd32 = (in->flags & CSV_IN_READS_I32) != 0;
    // CSV_IN_READS_DELTA16: starts as gaps / ends as lengths, each only where escapes are rare (host work per escape, nothing saved)
    const bool rdz = rd32 && R > (i64)env_int("CSV_DELTA16_MIN", 32 << 10) && (in->flags & CSV_IN_READS_DELTA16) && !getenv("CSV_NO_DELTA16");
    const i64 esc_per = (i64)env_int("CSV_DELTA16_ESC", 64);
    const bool r_gaps = rdz && in->r_delta && in->r_start && in->n_r_esc >= 0 && (in->n_r_esc == 0 || (in->r_esc_row && in->r_esc_val)) && in->n_r_esc * esc_per <= R;
    const bool r_lens = rdz && in->r_len16 && in->n_l_esc >= 0 && (in->n_l_esc == 0 || (in->l_esc_row && in->l_esc_val)) && in->n_l_esc * esc_per <= R;
    const i64 r_ntile = div_up(R, CH_TILE), r_anc_cap = r_gaps ? (r_ntile + in->n_chrom + in->n_r_esc + 8) : 0;
    c->reads_delta = (r_gaps ? 1 : 0) | (r_lens ? 2 : 0);
    c->reads_early = false;
    // the small tables (segments, prefix, drop marks, gate records, status words, chain tile records) are ONE block laid out like
    // their page-locked staging copy: one DMA copy brings them all (r04: five blit kernels of ~5 us each in front of the columns)
    const size_t o_seg = 0, o_woff = o_seg + (size_t)(S + 1) * sizeof(csv_segment), o_drop = o_woff + (size_t)(S + 2) * 8,
                 o_gate = (o_drop + (size_t)S + 1 + 15) & ~(size_t)15, o_serr = o_gate + (size_t)(S + 1) * 16,
                 o_tiles = (o_serr + (size_t)(S + 1) * 4 + 15) & ~(size_t)15, o_end = o_tiles + (size_t)nt * TILE_REC * 16,
                 o_ones = (o_end + 255) & ~(size_t)255, ones_bytes = (size_t)(CH_TILE + 64) * 8, o_anc = o_ones + ones_bytes,
                 anc_bytes = delta16 ? (size_t)(div_up(W, CH_TILE) + 2 + 2 * n_anc_cap) * 4 : 0, o_ranc = o_anc + ((anc_bytes + 255) & ~(size_t)255),
                 ranc_bytes = r_gaps ? (size_t)(r_ntile + 2 + 2 * r_anc_cap) * 4 : 0, o_lesc = o_ranc + ((ranc_bytes + 255) & ~(size_t)255),
                 lesc_bytes = r_lens ? (size_t)(in->n_l_esc + 1) * 12 : 0, o_stage_end = o_lesc + ((lesc_bytes + 255) & ~(size_t)255);
    PL(tabs, o_end);
    // positions and lengths stay in the width they arrive in: the kernels read int32 columns as they are (kernels.hip.h Col)
    // (the position column is followed by a tile of padding, so that the chain kernel can read any span that begins inside the batch)
    if (sig32) { PL(a32, (W + CH_TILE + 64) * 4); PL(b32, (W + 1) * 4); } else { PL(a, (W + CH_TILE + 64) * 8); PL(b, (W + 1) * 8); }
    PL(rid, (W + 1) * 4); PL(aux, (W + 1) * 4);
    if (delta16) { PL(ad16, (W + CH_TILE + 64) * 2); PL(anc, (size_t)(div_up(W, CH_TILE) + 2 + 2 * n_anc_cap) * 4); }
    PL(sup_tmp, (W + 1) * 4);
    if (per_sig) { PL(cluster_id, (W + 1) * 4); PL(allele_id, (W + 1) * 4); }
    PL(partial, nt * 4); PL(tile_cnt, nt * 16);
    if (lazy) PL(tile_lead, nt * 4);
    if (per_sig) PL(ch_masks, nt * CT_WORDS * 8);
    PL(tile_items, nt * (size_t)TI_STRIDE * 16);
    PL(item_rec, cap_items * 16); PL(list_small, cap_items * 16); PL(list_big, cap_items * 4); PL(list_tiny, cap_items * 16); PL(list_wide, cap_items * 16);
    PL(item_cnt, cap_items * 8); PL(item_base, (cap_items + 8) * 8); PL(item_chunk, (cap_items / IS_CHUNK + 2) * 8);
    // temp call records are indexed by w (a cluster's slots live in its own signature range)
    PL(t_rec, (W + 1) * sizeof(TmpRec)); PL(t_rec0, (cap_items + 1) * sizeof(TmpRec));
    PL(sc_k, SC * 8); PL(sc_x, SC * 8); PL(sc_v1, SC * 4); PL(sc_v2, SC * 4); PL(sc_v3, SC * 4); PL(sc_v4, SC * 4); PL(sc_v5, SC * 4);
    PL(o_rec, (cap_tmp + 1) * sizeof(CallRec)); PL(o_supsig, (W + 1) * 4); PL(o_suprid, (W + 1) * 4);
    PL(o_rec2, (cap_tmp + 1) * sizeof(CallRec)); PL(o_supsig2, (W + 1) * 4);
    if (have_tab) { PL(reads_off, (in->n_chrom + 1) * 8); PL(contig_len, (in->n_chrom + 1) * 8); }
    if (R > 0) {
        PL(gt_over, (cap_tmp + 2) * 4); PL(gt_huge, (cap_tmp + 2) * 4); PL(gt_pool, pool_n * 4);
        // the table as uploaded and its packed start-ordered form, both in the caller's width (int32: 13 + 12 bytes per read)
        const size_t cw = rd32 ? 4 : 8;
        PL(r_start, R * cw); PL(r_end, R * cw); PL(r_primary, R); PL(r_id, R * 4);
        if (r_gaps) { PL(rd16, (R + CH_TILE + 64) * 2); PL(ranc, (size_t)(r_ntile + 2 + 2 * r_anc_cap) * 4); }
        if (r_lens) { PL(rl16, (R + 64) * 2); PL(rlesc, (size_t)(in->n_l_esc + 1) * 12); }
        PL(s_start, (R + 64) * cw); PL(s_end, (R + 64) * cw); PL(s_idp, (R + 64) * 4);      // (whole chunks of 64 rows are read)
        PL(cmax, (div_up(R, 64) + 136) * 8); PL(span_len, (div_up(R, 512) + 8) * 8); PL(cfirst, (div_up(R, 64) + 136) * 8); PL(bfirst, (div_up(R, 4096) + 136) * 8);      // (+ two steps of padding: k_genotype reads 128 entries from any valid one)
        PL(maxlen, (in->n_chrom + 1) * 8);
        if (reorder) { PL(ro_tcnt, (div_up(R, RO_TILE) + 1) * 4); PL(ro_ent, (div_up(R, RO_TILE) + 1) * (size_t)RO_TCAP * 16); PL(ro_tblk, (div_up(R, RO_TILE) + 2) * 4); PL(ro_table, RO_CAP * 16); }
    }
#undef PL
    {
        // the arena may move: nothing may still be running out of the old one
        if (P.total > c->arena.cap) HIP_TRY(c, hipDeviceSynchronize());
        const int rc = commit(c, c->arena, P);
        if (rc) return rc;
    }

    // ---- small tables: staged in page-locked memory, one copy
    {
        char* tb = (char*)c->tabs.p;
        c->seg.p = tb + o_seg; c->woff.p = tb + o_woff; c->seg_drop.p = tb + o_drop; c->seg_gate.p = tb + o_gate; c->seg_err.p = tb + o_serr; c->tile_info.p = tb + o_tiles;
    }
    // (the staging block is also the landing zone of the results: never smaller than one counters struct)
    { const int rc = pin_reserve(c, o_stage_end + sizeof(DevCounters) + 256); if (rc) return rc; }
    hipStream_t st = c->stream;
    // The column copies go out on the kernels' OWN stream: the kernels that wait for them then wait on a barrier packet in
    // their queue.  On a copy stream of their own (r01-r04, CSV_COPY_STREAM=1) the dependency was an event across queues, which
    // this runtime resolves late: ~90 us between the end of the copy and the first kernel in a one-shot call (cfg3 gate-first
    // 0.64 -> 0.55 ms) - more than the chain kernels (~15 us) ever overlapped with the second copy group.
    hipStream_t cs = env_int("CSV_COPY_STREAM", 0) ? c->copy[0] : st;
    HIP_TRY(c, hipStreamSynchronize(st));                   // the staging block may still be the source of an earlier copy
    // whatever ran before (a resident caller's kernels still reading the columns this upload rewrites) is over before the
    // column copies start - and nothing else: they do not wait for the tables or the zero fills below
    HIP_TRY(c, hipEventRecord(c->ev_init, st));
    HIP_TRY(c, hipStreamWaitEvent(cs, c->ev_init, 0));
    memset(c->h_pin + o_serr, 0, (size_t)(S + 1) * 4);
    memcpy(c->h_pin + o_seg, c->h_seg.data(), (size_t)S * sizeof(csv_segment));
    memcpy(c->h_pin + o_woff, c->h_woff.data(), (size_t)(S + 1) * 8);
    memcpy(c->h_pin + o_drop, drop.data(), (size_t)S + 1);
    int* gate = (int*)(c->h_pin + o_gate);                  // {read_count, dropped, svtype, -} per segment
    for (int k = 0; k < S; k++) { gate[4 * k] = c->h_seg[k].read_count; gate[4 * k + 1] = drop[k]; gate[4 * k + 2] = c->h_seg[k].svtype; gate[4 * k + 3] = 0; }
    {
        // per chain tile: first / last segment and the chain / gate scalars of up to three non-empty segments inline
        // (kernels.hip.h TILE_REC).  Empty segments own no row and are skipped.
        int* ti = (int*)(c->h_pin + o_tiles);
        memset(ti, 0, (size_t)nt * TILE_REC * 16);
        int k = 0;
        for (i64 t = 0; t < nt; t++) {
            int* r = ti + 4 * TILE_REC * t;
            const i64 w0 = t * (i64)CH_TILE, w1 = (w0 + CH_TILE < W ? w0 + CH_TILE : W) - 1;
            if (w0 >= W) { r[1] = -1; continue; }
            while (k + 1 < S && c->h_woff[k + 1] <= w0) k++;
            int k1 = k;
            while (k1 + 1 < S && c->h_woff[k1 + 1] <= w1) k1++;
            r[0] = k; r[1] = k1;
            int nin = 0;
            bool wide_bias = false;
            for (int q = k; q <= k1; q++) {
                if (c->h_woff[q + 1] == c->h_woff[q]) continue;
                if (nin < 3) {
                    const csv_segment& g = c->h_seg[q];
                    int* a = r + 4 + 4 * nin;
                    a[0] = (int)c->h_woff[q]; a[1] = g.read_count; a[2] = q | (g.svtype << 24) | (drop[q] ? (1 << 28) : 0); a[3] = (int)g.max_cluster_bias;
                    if (g.max_cluster_bias != (i64)(int)g.max_cluster_bias || q >= (1 << 24)) wide_bias = true;
                }
                nin++;
            }
            r[2] = (nin <= 3 && !wide_bias) ? nin : 0;
        }
    }
    // (the upload's reads-order state is cleared here, in FRONT of every copy - behind the column copies the fill kernel was one
    // more switch between the copy engine and the compute queue on the one-shot call's critical path - and only when there is a
    // reads table to order)
    if (have_tab || c->rstate_dirty) { HIP_TRY(c, hipMemsetAsync(c->rstate.p, 0, sizeof(ReadsState), st)); c->rstate_dirty = have_tab; }
    HIP_TRY(c, hipMemcpyAsync(c->tabs.p, c->h_pin, o_end, hipMemcpyHostToDevice, st));

    // ---- columns, on the copy stream, in two groups: what the chain kernels read (positions, lengths / pos2, the strand
    // and chr2 words of INV / TRA segments), then what only the refine kernels read (read ids, INS sequence lengths).  In a
    // one-shot call the chain kernels start when the first group has landed and run under the second.  Segments whose
    // source ranges are adjacent travel as one copy; aux is not read for DEL / DUP segments (include/cutesv_hip.h) and is
    // zero-filled on the device instead of crossing PCIe.  (One stream: a second DMA engine adds nothing on this link -
    // scripts/micro/h2d_bw.hip measures 57 GB/s with 1, 2, 4 or 8 streams, from page-locked and pageable memory alike.)
    // The padding behind the position column (positive values) is a DMA copy of a block of ones out of the staging area, in
    // FRONT of the column: a fill kernel behind the DMA copy cost the stream an engine switch (~40 us in the trace) right
    // where the chain kernels wait.
    {
        int* ones = (int*)(c->h_pin + o_ones);
        if (!delta16) for (size_t i = 0; i < ones_bytes / 4; i++) ones[i] = 1;
        if (delta16) {}                                   // (k_unpack_a16 writes the padding together with the column)
        else if (sig32) HIP_TRY(c, hipMemcpyAsync(dp<int>(c->a32) + W, ones, (size_t)(CH_TILE + 64) * 4, hipMemcpyHostToDevice, cs));
        else HIP_TRY(c, hipMemcpyAsync(dp<i64>(c->a) + W, ones, (size_t)(CH_TILE + 64) * 8, hipMemcpyHostToDevice, cs));
    }
    auto aux_kind = [&](int q) { const int t = c->h_seg[q].svtype; return t == CSV_INS ? 2 : (t == CSV_INV || t == CSV_TRA) ? 1 : 0; };
    for (int group = 1; group <= 2; group++) {
        for (int k = 0; k < S;) {
            int e = k;
            while (e + 1 < S && c->h_seg[e + 1].sig_begin == c->h_seg[e].sig_end) e++;
            const i64 src = c->h_seg[k].sig_begin, n = c->h_woff[e + 1] - c->h_woff[k], dst = c->h_woff[k];
            if (n > 0) {
                if (group == 1 && sig32) {
                    if (delta16) { HIP_TRY(c, hipMemcpyAsync(dp<uint16_t>(c->ad16) + dst, in->a_delta + src, n * 2, col_kind, cs)); c->lazy_bytes += n * 2; }
                    else HIP_TRY(c, hipMemcpyAsync(dp<int>(c->a32) + dst, (const int32_t*)in->a + src, n * 4, col_kind, cs));
                    if (!lazy) HIP_TRY(c, hipMemcpyAsync(dp<int>(c->b32) + dst, (const int32_t*)in->b + src, n * 4, col_kind, cs));
                } else if (group == 1) {
                    HIP_TRY(c, hipMemcpyAsync(dp<i64>(c->a) + dst, in->a + src, n * 8, col_kind, cs));
                    if (!lazy) HIP_TRY(c, hipMemcpyAsync(dp<i64>(c->b) + dst, in->b + src, n * 8, col_kind, cs));
                } else if (!lazy) HIP_TRY(c, hipMemcpyAsync(dp<int>(c->rid) + dst, in->read_id + src, n * 4, col_kind, cs));
                if (lazy && group == 2) c->lazy_bytes += n * (sig32 ? 4 : 8) + n * 4;      // b and read_id of the range stay behind
                for (int q = k; q <= e;) {                  // aux: runs of segments of this group's kind
                    int q2 = q;
                    while (q2 + 1 <= e && aux_kind(q2 + 1) == aux_kind(q)) q2++;
                    const i64 na = c->h_woff[q2 + 1] - c->h_woff[q];
                    if (aux_kind(q) == group && na > 0) {
                        if (!(lazy && group == 2)) HIP_TRY(c, hipMemcpyAsync(dp<int>(c->aux) + c->h_woff[q], in->aux + c->h_seg[q].sig_begin, na * 4, col_kind, cs));
                        else c->lazy_bytes += na * 4;
                    }
                    // aux of DEL / DUP segments is not the caller's to define (include/cutesv_hip.h): zero on the device.  Their own
                    // ranges only, on the main stream - nothing a copy writes, so the copies wait for no fill (gate-first: k_lazy_fetch
                    // writes the zeros of the rows it fetches)
                    if (group == 1 && aux_kind(q) == 0 && na > 0 && !lazy) HIP_TRY(c, hipMemsetAsync(dp<int>(c->aux) + c->h_woff[q], 0, (size_t)na * 4, st));
                    // gate-first: the chain predicates of INV / TRA segments read b (kernels.hip.h sig_flag): those ranges travel whole
                    if (lazy && group == 1 && aux_kind(q) == 1 && na > 0) {
                        const i64 sb = c->h_seg[q].sig_begin;
                        if (sig32) HIP_TRY(c, hipMemcpyAsync(dp<int>(c->b32) + c->h_woff[q], (const int32_t*)in->b + sb, na * 4, col_kind, cs));
                        else HIP_TRY(c, hipMemcpyAsync(dp<i64>(c->b) + c->h_woff[q], in->b + sb, na * 8, col_kind, cs));
                        c->lazy_bytes -= na * (sig32 ? 4 : 8);
                    }
                    q = q2 + 1;
                }
            }
            k = e + 1;
        }
        if (group == 1 && delta16) {
            // the anchors, built while the gaps are on the link: the first row of every chain tile, the first row of every
            // segment, the caller's escape rows that lie in a segment - {w, a[source row of w]} by ascending w, one slice per tile
            const int32_t* ha = (const int32_t*)in->a;
            const i64 ntile = div_up(W, CH_TILE);
            std::vector<std::pair<i64, int>> anc;
            anc.reserve((size_t)n_anc_cap);
            {
                int k = 0;
                for (i64 t = 0; t < ntile; t++) {
                    const i64 w = t * (i64)CH_TILE;
                    while (k + 1 < S && c->h_woff[k + 1] <= w) k++;
                    anc.emplace_back(w, ha[c->h_seg[k].sig_begin + (w - c->h_woff[k])]);
                }
            }
            for (auto& bk : by_begin) anc.emplace_back(c->h_woff[bk.second], ha[bk.first]);
            for (i64 e = 0; e < in->n_esc; e++) {
                const i64 g = in->a_esc_row[e];
                auto it = std::upper_bound(by_begin.begin(), by_begin.end(), std::make_pair(g, INT32_MAX));
                if (it == by_begin.begin()) continue;
                --it;
                const csv_segment& sg = c->h_seg[it->second];
                if (g >= sg.sig_end) continue;                      // (a row no segment of this batch holds)
                anc.emplace_back(c->h_woff[it->second] + (g - sg.sig_begin), in->a_esc_val[e]);
            }
            std::sort(anc.begin(), anc.end());
            anc.erase(std::unique(anc.begin(), anc.end(), [](const std::pair<i64, int>& x, const std::pair<i64, int>& y) { return x.first == y.first; }), anc.end());
            int* h_off = (int*)(c->h_pin + o_anc);
            int* h_w = h_off + ntile + 2;
            int* h_v = h_w + n_anc_cap;
            size_t q = 0;
            for (i64 t = 0; t <= ntile; t++) {
                while (q < anc.size() && anc[q].first < t * (i64)CH_TILE) q++;
                h_off[t] = (int)q;
            }
            h_off[ntile + 1] = (int)anc.size();
            for (size_t i = 0; i < anc.size(); i++) { h_w[i] = (int)anc[i].first; h_v[i] = anc[i].second; }
            // (on a copy stream of its own: behind the gaps on the kernels' stream it was 7 us of copy + 9 us of hand-over between
            // two copies on the call's critical path; here it lands while the gaps are still on the link, and the event has long
            // fired when k_unpack_a16 - queued behind the gaps - gets to wait for it)
            HIP_TRY(c, hipMemcpyAsync(c->anc.p, c->h_pin + o_anc, anc_bytes, hipMemcpyHostToDevice, c->copy[0]));
            HIP_TRY(c, hipEventRecord(c->ev_anc, c->copy[0]));
        }
        HIP_TRY(c, hipEventRecord(c->ev_copy[group - 1], cs));
    }
    c->copies_pending = true;                               // run_impl orders the kernels behind the two events
    // reads table: its own stream (side[2] runs the reads_order / prefix-max kernels behind it)
    hipStream_t sr = c->side[2];
    if (have_tab) {
        HIP_TRY(c, hipStreamWaitEvent(sr, c->ev_init, 0));
        HIP_TRY(c, hipMemcpyAsync(c->reads_off.p, in->reads_off, (size_t)(in->n_chrom + 1) * 8, hipMemcpyHostToDevice, sr));
        if (c->any_tra_gt && in->n_chrom > 0) HIP_TRY(c, hipMemcpyAsync(c->contig_len.p, in->contig_len, (size_t)in->n_chrom * 8, hipMemcpyHostToDevice, sr));
    }
    if (R > 0 && reorder) {
        // which chromosome blocks begin inside every tile of RO_TILE rows: k_reads_runs leaves a descent AT a block start out of
        // its lists (k_reads_plan adds every block start anyway; a reference with hundreds of small contigs has dozens of them
        // per tile)
        const int ntl = div_up(R, RO_TILE);
        c->h_tblk.assign((size_t)ntl + 1, 0);
        int k = 0;
        for (int t = 0; t <= ntl; t++) {
            while (k < in->n_chrom && in->reads_off[k] < (i64)t * RO_TILE) k++;
            c->h_tblk[(size_t)t] = k;
        }
        HIP_TRY(c, hipMemcpyAsync(c->ro_tblk.p, c->h_tblk.data(), ((size_t)ntl + 1) * 4, hipMemcpyHostToDevice, sr));
    }
    const bool r_packed = R > 0 && in->r_idp != nullptr && rd32 && !getenv("CSV_NO_DELTA16");
    // all three 16-bit / packed forms on offer: the decode of one column runs (on a side stream, behind an event) while the next
    // column is on the link - id | primary first (the largest), the start gaps + their anchors, the lengths last, so that only
    // k_reads_end16 is left when the last byte has landed.  In one stream, in the order copies - copies - kernels, the three
    // decode kernels and the anchors' copy (95 us with their hand-overs) all followed the last byte.
    const bool r_overlap = r_packed && r_gaps && r_lens && !c->opt.no_reads_overlap;
    auto reads_anchors = [&]() {
        // anchors of the start column, built while the table is on the link: the first row of every tile of 2048 rows, the
        // first row of every chromosome block, the caller's escape rows (the first row of every sorted run is one) - rows of
        // the table itself: no w space here.  They follow the table on its own stream (page-locked staging: h_ranc).
        const int32_t* hs = (const int32_t*)in->r_start;
        std::vector<std::pair<i64, int>> anc;
        anc.reserve((size_t)r_anc_cap);
        for (i64 t = 0; t < r_ntile; t++) anc.emplace_back(t * (i64)CH_TILE, hs[t * (i64)CH_TILE]);
        for (int k = 0; k < in->n_chrom; k++) { const i64 r0 = in->reads_off[k]; if (r0 >= 0 && r0 < R) anc.emplace_back(r0, hs[r0]); }
        for (i64 e = 0; e < in->n_r_esc; e++) { const i64 r0 = in->r_esc_row[e]; if (r0 >= 0 && r0 < R) anc.emplace_back(r0, in->r_esc_val[e]); }
        std::sort(anc.begin(), anc.end());
        anc.erase(std::unique(anc.begin(), anc.end(), [](const std::pair<i64, int>& x, const std::pair<i64, int>& y) { return x.first == y.first; }), anc.end());
        int* h_off = (int*)(c->h_pin + o_ranc);          // (page-locked staging: a copy out of pageable memory would block the host
                                                         // until the table in front of it on this stream has crossed the link)
        int* h_w = h_off + r_ntile + 2;
        int* h_v = h_w + r_anc_cap;
        size_t q = 0;
        for (i64 t = 0; t <= r_ntile; t++) { while (q < anc.size() && anc[q].first < t * (i64)CH_TILE) q++; h_off[t] = (int)q; }
        h_off[r_ntile + 1] = (int)anc.size();
        for (size_t i = 0; i < anc.size(); i++) { h_w[i] = (int)anc[i].first; h_v[i] = anc[i].second; }
        return h_off;
    };
    auto reads_unpack = [&](hipStream_t s) {
        UnpackArgs UA{dp<uint16_t>(c->rd16), dp<int>(c->r_start), R, dp<int>(c->ranc), dp<int>(c->ranc) + r_ntile + 2, dp<int>(c->ranc) + r_ntile + 2 + r_anc_cap, (int)r_ntile, 0, 0};
        DevBatch none;
        memset(&none, 0, sizeof none);
        hipLaunchKernelGGL(k_unpack_a16, dim3((unsigned)r_ntile), dim3(256), 0, s, UA, none);
    };
    auto reads_ends = [&](hipStream_t s) -> int {
        hipLaunchKernelGGL(k_reads_end16, dim3(div_up(R, 256)), dim3(256), 0, s, (const int*)dp<int>(c->r_start), (const uint16_t*)dp<uint16_t>(c->rl16), dp<int>(c->r_end), R);
        if (in->n_l_esc > 0) {
            char* base = (char*)c->rlesc.p;
            char* hst = c->h_pin + o_lesc;                // (through the page-locked staging, as above)
            memcpy(hst, in->l_esc_row, (size_t)in->n_l_esc * 8);
            memcpy(hst + (size_t)in->n_l_esc * 8, in->l_esc_val, (size_t)in->n_l_esc * 4);
            HIP_TRY(c, hipMemcpyAsync(base, hst, (size_t)in->n_l_esc * 12, hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(k_scatter_rows_i32, dim3(div_up(in->n_l_esc, 256)), dim3(256), 0, s, (const i64*)base, (const int*)(base + (size_t)in->n_l_esc * 8), dp<int>(c->r_end), in->n_l_esc, R);
        }
        return CSV_OK;
    };
    if (R > 0 && r_overlap) {
        hipStream_t sk = c->side[1];
        // (the packed word lands where the packed start-ordered table will be built - s_idp is not written before k_reads_gather)
        HIP_TRY(c, hipMemcpyAsync(c->s_idp.p, in->r_idp, (size_t)R * 4, hipMemcpyHostToDevice, sr));
        HIP_TRY(c, hipEventRecord(c->ev_rd[0], sr));
        HIP_TRY(c, hipMemcpyAsync(c->rd16.p, in->r_delta, (size_t)R * 2, hipMemcpyHostToDevice, sr));
        HIP_TRY(c, hipEventRecord(c->ev_rd[1], sr));
        HIP_TRY(c, hipMemcpyAsync(c->rl16.p, in->r_len16, (size_t)R * 2, hipMemcpyHostToDevice, sr));
        // (the anchors travel on another copy stream: between two columns in this one they cost the link 34 us of turn-arounds)
        int* h_off = reads_anchors();
        HIP_TRY(c, hipStreamWaitEvent(c->copy[1], c->ev_init, 0));
        HIP_TRY(c, hipMemcpyAsync(c->ranc.p, h_off, ranc_bytes, hipMemcpyHostToDevice, c->copy[1]));
        HIP_TRY(c, hipEventRecord(c->ev_rd[3], c->copy[1]));
        HIP_TRY(c, hipStreamWaitEvent(sk, c->ev_rd[0], 0));
        hipLaunchKernelGGL(k_reads_split_idp, dim3(div_up(R, 256)), dim3(256), 0, sk, (const unsigned*)c->s_idp.p, dp<int>(c->r_id), dp<uint8_t>(c->r_primary), R);
        HIP_TRY(c, hipStreamWaitEvent(sk, c->ev_rd[1], 0));
        HIP_TRY(c, hipStreamWaitEvent(sk, c->ev_rd[3], 0));
        reads_unpack(sk);
        HIP_TRY(c, hipEventRecord(c->ev_rd[2], sk));
        HIP_TRY(c, hipStreamWaitEvent(sr, c->ev_rd[2], 0));          // (long fired when the lengths have crossed)
        { const int rc = reads_ends(sr); if (rc) return rc; }
        c->reads_delta |= 4;
        c->reads_early = true;                              // (the first run may order the table on the decode stream)
    } else if (R > 0) {
        const size_t cw = rd32 ? 4 : 8;
        if (r_gaps) HIP_TRY(c, hipMemcpyAsync(c->rd16.p, in->r_delta, (size_t)R * 2, hipMemcpyHostToDevice, sr));
        else HIP_TRY(c, hipMemcpyAsync(c->r_start.p, in->r_start, R * cw, hipMemcpyHostToDevice, sr));
        if (r_lens) HIP_TRY(c, hipMemcpyAsync(c->rl16.p, in->r_len16, (size_t)R * 2, hipMemcpyHostToDevice, sr));
        else HIP_TRY(c, hipMemcpyAsync(c->r_end.p, in->r_end, R * cw, hipMemcpyHostToDevice, sr));
        if (r_packed) {
            // (the packed word lands where the packed start-ordered table will be built - s_idp is not written before k_reads_gather -
            // and is split into the two columns the reads stage reads)
            HIP_TRY(c, hipMemcpyAsync(c->s_idp.p, in->r_idp, (size_t)R * 4, hipMemcpyHostToDevice, sr));
            hipLaunchKernelGGL(k_reads_split_idp, dim3(div_up(R, 256)), dim3(256), 0, sr, (const unsigned*)c->s_idp.p, dp<int>(c->r_id), dp<uint8_t>(c->r_primary), R);
            c->reads_delta |= 4;
        } else {
            HIP_TRY(c, hipMemcpyAsync(c->r_primary.p, in->r_primary, R, hipMemcpyHostToDevice, sr));
            HIP_TRY(c, hipMemcpyAsync(c->r_id.p, in->r_id, R * 4, hipMemcpyHostToDevice, sr));
        }
        if (r_gaps) {
            int* h_off = reads_anchors();
            HIP_TRY(c, hipMemcpyAsync(c->ranc.p, h_off, ranc_bytes, hipMemcpyHostToDevice, sr));
            reads_unpack(sr);
        }
        if (r_lens) { const int rc = reads_ends(sr); if (rc) return rc; }
    }
    // (the main stream does not wait for the reads table: the kernels that read it are ordered behind this event)
    if (have_tab) HIP_TRY(c, hipEventRecord(c->ev_reads, sr));
    c->unpack_pending = false;
    if (delta16) {
        // The position column out of its gaps (k_unpack_a16) is queued behind EVERY copy of the upload, as the first kernel of
        // the run (a resident upload: right here).  Measured on cfg4 (80 MB of reads table on its own stream): the kernel queued
        // between the column copies delayed the copies behind it until the reads table had left the copy engine (1.77 -> 2.82 ms);
        // queued from here in a one-shot call, after the last copy, the launch itself blocked the host for 1.1 ms.
        const i64 ntile = div_up(W, CH_TILE);
        c->unpack_args = UnpackArgs{dp<uint16_t>(c->ad16), dp<int>(c->a32), W, dp<int>(c->anc), dp<int>(c->anc) + ntile + 2, dp<int>(c->anc) + ntile + 2 + n_anc_cap, (int)ntile, 0, 1};
        c->unpack_tiles = (int)ntile;
        c->unpack_pending = true;
        if (sync) {
            HIP_TRY(c, hipStreamWaitEvent(st, c->ev_copy[0], 0));
            HIP_TRY(c, hipStreamWaitEvent(st, c->ev_anc, 0));
            DevBatch none;                                // (a resident upload is never gate-first: nothing of the batch is read)
            memset(&none, 0, sizeof none);
            hipLaunchKernelGGL(k_unpack_a16, dim3((unsigned)ntile + 1), dim3(256), 0, st, c->unpack_args, none);
            c->unpack_pending = false;
        }
    }
    c->have_tab = have_tab;
    if (sync) {
        HIP_TRY(c, hipStreamSynchronize(st)); HIP_TRY(c, hipStreamSynchronize(cs));
        if (have_tab) HIP_TRY(c, hipStreamSynchronize(sr));
        c->copies_pending = false;
    }

    DevBatch& B = c->B;
    memset(&B, 0, sizeof B);
    B.n_seg = S; B.n_chrom = in->n_chrom; B.W = W;
    B.seg = dp<csv_segment>(c->seg); B.woff = dp<i64>(c->woff); B.seg_drop = dp<uint8_t>(c->seg_drop);
    if (sig32) { B.a = Col{nullptr, dp<int>(c->a32)}; B.b = Col{nullptr, dp<int>(c->b32)}; }
    else { B.a = Col{dp<i64>(c->a), nullptr}; B.b = Col{dp<i64>(c->b), nullptr}; }
    B.rid = dp<int>(c->rid); B.aux = dp<int>(c->aux);
    B.per_sig = per_sig ? 1 : 0;
    B.end_z = 0;                                            // is the batch's last signature a (0,0) element?  (the reference's sentinel rule, INDEL:62-64)
    for (int k = S - 1; k >= 0; k--)
        if (c->h_seg[k].sig_end > c->h_seg[k].sig_begin) {
            const i64 last = c->h_seg[k].sig_end - 1;
            if (dev_cols) {                                 // (columns in device memory: the two values are fetched)
                i64 va = 0, vb = 0; int32_t wa = 0, wb = 0;
                if (sig32) { HIP_TRY(c, hipMemcpy(&wa, (const int32_t*)in->a + last, 4, hipMemcpyDeviceToHost)); HIP_TRY(c, hipMemcpy(&wb, (const int32_t*)in->b + last, 4, hipMemcpyDeviceToHost)); va = wa; vb = wb; }
                else { HIP_TRY(c, hipMemcpy(&va, in->a + last, 8, hipMemcpyDeviceToHost)); HIP_TRY(c, hipMemcpy(&vb, in->b + last, 8, hipMemcpyDeviceToHost)); }
                B.end_z = va == 0 && vb == 0;
            } else
                B.end_z = sig32 ? (((const int32_t*)in->a)[last] == 0 && ((const int32_t*)in->b)[last] == 0) : (in->a[last] == 0 && in->b[last] == 0);
            break;
        }
    B.cluster_id = per_sig ? dp<int>(c->cluster_id) : nullptr; B.allele_id = per_sig ? dp<int>(c->allele_id) : nullptr;
    B.partial = dp<int>(c->partial); B.tile_cnt = dp<int4>(c->tile_cnt);
    B.item_rec = dp<int4>(c->item_rec); B.list_small = dp<int4>(c->list_small); B.list_big = dp<int>(c->list_big); B.list_tiny = dp<int4>(c->list_tiny); B.list_wide = dp<int4>(c->list_wide);
    B.seg_gate = dp<int4>(c->seg_gate); B.tile_info = dp<int4>(c->tile_info);
    if (lazy) { B.h_b = lz_b; B.h_rid = (const int*)lz_rid; B.h_aux = (const int*)lz_aux; B.h_rows8 = (const int2*)lz_rows8; B.tile_lead = dp<int>(c->tile_lead); c->lazy_pending = c->partial_cols = true; }
    B.ch_masks = per_sig ? dp<u64>(c->ch_masks) : nullptr; B.tile_items = dp<int4>(c->tile_items);
    B.seg_err = dp<int>(c->seg_err);
    B.tiny_max = getenv("CSV_NO_TINY") ? 0 : 16;               // (timing aid: 0 sends every DEL/INS cluster of m <= 32 through the paired path)
    B.item_cnt = dp<i64>(c->item_cnt); B.item_base = dp<i64>(c->item_base); B.item_chunk = dp<i64>(c->item_chunk);
    B.sup_tmp = dp<int>(c->sup_tmp);
    B.t_rec = dp<TmpRec>(c->t_rec); B.t_rec0 = dp<TmpRec>(c->t_rec0);
    B.cap_tmp = (int)cap_tmp; B.cap_items = (int)cap_items;
    B.sc_k = dp<u64>(c->sc_k); B.sc_x = dp<i64>(c->sc_x); B.sc_v1 = dp<int>(c->sc_v1); B.sc_v2 = dp<int>(c->sc_v2); B.sc_v3 = dp<int>(c->sc_v3); B.sc_v4 = dp<int>(c->sc_v4); B.sc_v5 = dp<int>(c->sc_v5);
    B.o_rec = dp<CallRec>(c->o_rec); B.o_supsig = dp<int>(c->o_supsig); B.o_suprid = dp<int>(c->o_suprid);
    B.n_reads = R;
    if (have_tab) { B.reads_off = dp<i64>(c->reads_off); B.contig_len = dp<i64>(c->contig_len); }
    if (R > 0) {
        if (rd32) { B.r_start = Col{nullptr, dp<int>(c->r_start)}; B.r_end = Col{nullptr, dp<int>(c->r_end)}; B.s_start32 = dp<int>(c->s_start); B.s_end32 = dp<int>(c->s_end); }
        else { B.r_start = Col{dp<i64>(c->r_start), nullptr}; B.r_end = Col{dp<i64>(c->r_end), nullptr}; B.s_start64 = dp<i64>(c->s_start); B.s_end64 = dp<i64>(c->s_end); }
        B.r_primary = dp<uint8_t>(c->r_primary); B.r_id = dp<int>(c->r_id);
        B.s_idp = dp<int>(c->s_idp); B.cmax = dp<void>(c->cmax); B.span_len = dp<i64>(c->span_len); B.cfirst = dp<void>(c->cfirst); B.bfirst = dp<void>(c->bfirst); B.maxlen = dp<i64>(c->maxlen);
        B.gt_over = dp<int>(c->gt_over); B.gt_huge = dp<int>(c->gt_huge);
        B.gt_pool = dp<int>(c->gt_pool); B.gt_pool_n = pool_n;
        B.ro_mode = reorder ? 1 : 0;
        if (reorder) {
            B.ro_tcnt = dp<int>(c->ro_tcnt); B.ro_ent = dp<int4>(c->ro_ent); B.ro_tblk = dp<int>(c->ro_tblk); B.ro_table = dp<int4>(c->ro_table); B.ro_cap = RO_CAP;
            B.ro_gap = env_int("CSV_READS_GAP", 1000000);      // (tests shrink it together with their task regions)
        }
    }
    B.sqrt_tab = dp<double>(c->sqrt_tab); B.rcp_tab = dp<double>(c->rcp_tab); B.cipk_tab = dp<float>(c->cipk_tab); B.cnt = dp<DevCounters>(c->cnt); B.rs = dp<ReadsState>(c->rstate);
    c->n_sig_host = in->n_sig;
    c->n_reads = R;
    c->uploaded = true;
    return CSV_OK;
}

// general stable sort of the reads table by (chromosome, start): the fallback of the reads_order stage for tables that
// are not a permutation of disjoint sorted runs.  LSD radix passes of sort.hip.h over the 5 start bytes and the
// chromosome bytes; the result is a row permutation that k_reads_gather applies.
int general_reads_sort(csv_ctx* c, hipStream_t st)
{
    const i64 R = c->n_reads;
    const int nunits = div_up(R, SORT_WTILE), nblk = div_up(nunits, 4);
    int rc;
    if ((rc = reserve(c, c->gs_chrom, R * 4)) || (rc = reserve(c, c->gs_perm0, R * 4)) || (rc = reserve(c, c->gs_perm1, R * 4)) ||
        (rc = reserve(c, c->gs_hist, (size_t)256 * nunits * 4)) || (rc = reserve(c, c->gs_tot, 256 * 4))) return rc;
    hipLaunchKernelGGL(k_reads_chromcol, dim3(div_up(R, 256)), dim3(256), 0, st, c->B, dp<int>(c->gs_chrom));
    int cbytes = 0;
    for (u64 v = (u64)(c->B.n_chrom > 0 ? c->B.n_chrom - 1 : 0); v; v >>= 8) cbytes++;
    const int* pin = nullptr;
    int* pout = dp<int>(c->gs_perm0);
    for (int f = 0; f < 2; f++) {
        const int nb = f == 0 ? 5 : cbytes;                // starts < 2^40 (checked with the ends by k_reads_gather)
        for (int byte = 0; byte < nb; byte++) {
            const bool rn = c->B.r_start.p32 != nullptr;
            if (f == 0 && rn && byte >= 4) continue;         // (int32 starts have four key bytes)
            SortPass P{f == 0 ? (rn ? (const void*)c->B.r_start.p32 : (const void*)c->B.r_start.p64) : (const void*)c->gs_chrom.p, (f == 0 && !rn) ? 1 : 0, byte * 8, R, nunits, pin, pout, dp<int>(c->gs_hist)};
            hipLaunchKernelGGL(k_sort_hist, dim3(nblk), dim3(256), 0, st, P);
            hipLaunchKernelGGL(k_sort_rowsum, dim3(256), dim3(256), 0, st, dp<int>(c->gs_hist), nunits, dp<int>(c->gs_tot));
            hipLaunchKernelGGL(k_sort_rowscan, dim3(256), dim3(256), 0, st, dp<int>(c->gs_hist), nunits, dp<int>(c->gs_tot));
            hipLaunchKernelGGL(k_sort_scatter, dim3(nblk), dim3(256), 0, st, P);
            pin = pout;
            pout = (pout == dp<int>(c->gs_perm0)) ? dp<int>(c->gs_perm1) : dp<int>(c->gs_perm0);
        }
    }
    c->B.ro_perm = pin;
    HIP_TRY(c, hipGetLastError());
    return CSV_OK;
}

int run_impl(csv_ctx* c, csv_run_stats* stats)
{
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    DevBatch& B = c->B;
    const i64 W = B.W;
    constexpr int LDS_SMALL = refine_lds_bytes<64>();
    constexpr int LDS_MID = refine_lds_bytes<256>();
    constexpr int LDS_BIG = refine_lds_bytes<2048>();
    constexpr int LDS_PLAN = rp_lds_bytes(RO_CAP);
    if (!c->lds_set) {
        HIP_TRY(c, hipFuncSetAttribute((const void*)k_refine<256, 2048, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BIG));
        HIP_TRY(c, hipFuncSetAttribute((const void*)k_reads_plan<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_PLAN));
        HIP_TRY(c, hipFuncSetAttribute((const void*)k_reads_plan<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_PLAN));
        c->lds_set = true;
    }
    int ev = 0;
    auto mark = [&]() -> hipError_t { return stats ? hipEventRecord(c->ev[ev++], st) : hipSuccess; };
    const auto& O = c->opt;
    const bool dbg = O.debug;
#define DBG(name)                                                                                          \
    do {                                                                                                   \
        if (dbg) {                                                                                         \
            fprintf(stderr, "[csv] %s ...", name); fflush(stderr);                                         \
            hipError_t e_ = hipDeviceSynchronize();                                                        \
            fprintf(stderr, " %s\n", hipGetErrorString(e_)); fflush(stderr);                               \
        }                                                                                                  \
    } while (0)
    // this run's result arena: the other one than the last run's (whose publish may still be reading it on the publish stream)
    {
        const int p = c->parity ^ 1;
        if (c->pend[p].live) HIP_TRY(c, hipStreamWaitEvent(st, c->ev_pub[p], 0));       // (launched two runs ago)
        c->parity = p;
        B.cnt = (DevCounters*)((char*)c->cnt.p + 256 * p);
        B.o_rec = p ? dp<CallRec>(c->o_rec2) : dp<CallRec>(c->o_rec);
        B.o_supsig = p ? dp<int>(c->o_supsig2) : dp<int>(c->o_supsig);
    }
    HIP_TRY(c, mark());
    if (W == 0) HIP_TRY(c, hipMemsetAsync(B.cnt, 0, sizeof(DevCounters), st));      // otherwise k_chain_count zeroes them
    HIP_TRY(c, mark());                                                              // slot 0: init (empty batch only)
    // Plain runs fork the independent kernels onto side streams (joined again before k_items_scan /
    // k_genotype); instrumented runs (stats != NULL) and CSV_DEBUG keep everything on the main stream so
    // that every kernel is timed alone.
    // (forking costs a few event waits: only worth it when the batch has pair types or genotyping)
    const bool do_gt = c->any_genotype && B.n_reads > 0;
    // the packed table of an upload does not change between runs: a resident re-run keeps it (csv_batch_option) and has no reads stage
    const bool keep_reads = c->reads_ready && c->reuse_reads && !stats;
    const bool fork = !stats && !dbg && !O.no_fork && (c->any_pair || (do_gt && !keep_reads) || O.fork_always);
    // A genotyping batch has two producer chains - clustering (k_chain_count .. k_emit) and the reads stage - that meet in
    // k_genotype.  A wait across queues costs 6-11 us when the event fires late and next to nothing when it fired long ago,
    // so the LONGER chain stays on the main stream together with the genotype kernels and the shorter one is forked off:
    // its completion event has long fired when the main stream gets there.  (Reads dominate a 30x HiFi genome, clustering a
    // 90x all-types one.)  `st` is the stream of the clustering chain from here on, `sM` the main stream.
    hipStream_t sM = c->stream;
    const bool swap = fork && do_gt && !keep_reads && !c->copies_pending && B.n_reads > 4 * W && W > 0 && !O.no_swap;
    if (swap) st = c->side[2];
    hipStream_t sB = fork ? c->side[0] : st, sC = fork ? c->side[1] : st, sD = swap ? sM : (fork ? c->side[2] : st);
#define LAUNCH_ON(strm, name, kern, grid, block, lds, ...)                             \
    do {                                                                               \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, strm, __VA_ARGS__);      \
        DBG(name);                                                                     \
        HIP_TRY(c, mark());                                                            \
    } while (0)
#define LAUNCH(name, kern, grid, block, lds, ...) LAUNCH_ON(st, name, kern, grid, block, lds, __VA_ARGS__)
    // s_early: the stream the upload decoded the start column on (a one-shot call whose reads table came in its 16-bit forms):
    // k_reads_runs / k_reads_plan read the starts and nothing else of the table, so they run there, behind the decode, while the
    // end column is still on the link; s2 waits for their event before it packs the table
    auto reads_stage = [&](hipStream_t s2, hipStream_t s_early = nullptr) -> int {       // reads order + pack + longest read per chromosome on stream s2
        const int nr = div_up(B.n_reads, 2048);
        const bool rn = B.r_start.p32 != nullptr;
        const bool keep = keep_reads;
        if (!keep) {                                      // (k_reads_plan leaves a state on every path; nothing to reset)
            if (B.ro_mode == 2) {
                const int rc = general_reads_sort(c, s2);
                if (rc) return rc;
            } else if (B.ro_mode == 1) {
                const hipStream_t so = s_early ? s_early : s2;
                if (rn) {
                    hipLaunchKernelGGL(k_reads_runs<true>, dim3(div_up(B.n_reads, RO_TILE)), dim3(256), 0, so, B);
                    hipLaunchKernelGGL(k_reads_plan<true>, dim3(1), dim3(RP_THREADS), LDS_PLAN, so, B);
                } else {
                    hipLaunchKernelGGL(k_reads_runs<false>, dim3(div_up(B.n_reads, RO_TILE)), dim3(256), 0, so, B);
                    hipLaunchKernelGGL(k_reads_plan<false>, dim3(1), dim3(RP_THREADS), LDS_PLAN, so, B);
                }
                if (s_early) { HIP_TRY(c, hipEventRecord(c->ev_rd[4], so)); HIP_TRY(c, hipStreamWaitEvent(s2, c->ev_rd[4], 0)); }
            }
        }
        DBG("reads_order");
        if (s2 == st || stats) HIP_TRY(c, mark());
        if (!keep) {
            if (rn) hipLaunchKernelGGL(k_reads_gather<true>, dim3(nr), dim3(256), 0, s2, B);
            else hipLaunchKernelGGL(k_reads_gather<false>, dim3(nr), dim3(256), 0, s2, B);
        }
        DBG("reads_gather");
        if (s2 == st || stats) HIP_TRY(c, mark());
        if (!keep) hipLaunchKernelGGL(k_reads_maxlen, dim3(B.n_chrom < 1024 ? (B.n_chrom > 0 ? B.n_chrom : 1) : 1024), dim3(256), 0, s2, B);
        DBG("reads_maxlen");
        if (s2 == st || stats) HIP_TRY(c, mark());
        c->reads_ready = true;
        return CSV_OK;
    };
    if (c->copies_pending) HIP_TRY(c, hipStreamWaitEvent(st, c->ev_copy[0], 0));       // positions, lengths, INV / TRA words
    bool zero_done = false;
    if (c->unpack_pending) {                               // CSV_IN_SIG_DELTA16: the position column out of its gaps, first kernel of the call
        // (a gate-first call: the same kernel fetches `b` of the rows at position 0 - k_lazy_zero's whole job - as it writes them)
        HIP_TRY(c, hipStreamWaitEvent(st, c->ev_anc, 0));
        c->unpack_args.zero_b = (c->lazy_pending && W > 0) ? 1 : 0;
        zero_done = c->unpack_args.zero_b != 0;
        hipLaunchKernelGGL(k_unpack_a16, dim3((unsigned)c->unpack_tiles + 1), dim3(256), 0, st, c->unpack_args, B);
        c->unpack_pending = false;
    }
    // A run decides what it launches from what IT knows - nothing is carried over from earlier runs of the upload (r05 skipped
    // the tiers above 64 signatures when an earlier, identical run had found them empty: state only a benchmark loop has).  In
    // a one-shot call the column copies are still on the link when the chain kernels are queued, so the host can wait for
    // k_chain_apply's word {this run, work items above 64 signatures} - it arrives long before the copies end - and queue
    // only the tiers that have work; a resident run queues them all (an empty tier costs its launch, ~4.5 us).
    const bool peek = c->copies_pending && c->h_flag && !O.no_peek && !stats && !dbg;
    B.host_flag = peek ? c->d_flag : nullptr;
    B.run_seq = ++c->run_seq;
    if (W > 0) {
        const int nb = div_up(W, CH_TILE);
        if (fork && do_gt && !keep_reads) {
            // reads order + pack: independent of the clustering kernels.  Either chain waits only for whatever ran before on
            // the main stream (the previous run's genotype / publish kernels read what this run rewrites).  The stage's
            // verdict on the table goes to the upload's state, not to the run's counters (which k_chain_count zeroes).
            HIP_TRY(c, hipEventRecord(c->ev_init, sM));
            HIP_TRY(c, hipStreamWaitEvent(swap ? st : sD, c->ev_init, 0));
            const bool early = c->reads_early && c->copies_pending && !swap && !O.no_reads_overlap;
            if (early) HIP_TRY(c, hipStreamWaitEvent(c->side[1], c->ev_init, 0));
            const int rc = reads_stage(sD, early ? c->side[1] : nullptr);
            c->reads_early = false;
            if (rc) return rc;
            if (!swap) HIP_TRY(c, hipEventRecord(c->ev_aux[2], sD));
        }
        const bool lazy = c->lazy_pending;                 // (the first run of a gate-first upload; stats are never taken on one)
        if (lazy && !zero_done) {
            const int gz = nb < 2048 ? nb : 2048;
            if (B.a.p32) hipLaunchKernelGGL(k_lazy_zero<true>, dim3(gz), dim3(256), 0, st, B);
            else hipLaunchKernelGGL(k_lazy_zero<false>, dim3(gz), dim3(256), 0, st, B);
        }
        if (B.a.p32) LAUNCH("chain_count", k_chain_count<true>, nb, 256, 0, B);
        else LAUNCH("chain_count", k_chain_count<false>, nb, 256, 0, B);
        LAUNCH("chain_apply", k_chain_apply, div_up(nb, 4), 256, 0, B);
        if (B.per_sig) hipLaunchKernelGGL(k_chain_ids, dim3(nb), dim3(256), 0, st, B);      // (optional outputs; timed with whatever follows)
        if (lazy) {
            if (B.a.p32) hipLaunchKernelGGL(k_lazy_fetch<true>, dim3(nb), dim3(256), 0, st, B);
            else hipLaunchKernelGGL(k_lazy_fetch<false>, dim3(nb), dim3(256), 0, st, B);
            DBG("lazy_fetch");
            // the device columns now hold every row a kernel reads: later runs of this upload (the general-sort re-run of a reads
            // table, a caller's csv_batch_run) take them as they are - the caller's host columns are not touched again
            c->lazy_pending = false;
            B.h_b = nullptr; B.h_rid = nullptr; B.h_aux = nullptr; B.h_rows8 = nullptr; B.tile_lead = nullptr;
        }
        if (c->copies_pending) HIP_TRY(c, hipStreamWaitEvent(st, c->ev_copy[1], 0));   // read ids, INS sequence lengths
        int g_small = B.cap_items < 8192 ? B.cap_items : 8192;
        if (g_small < 1) g_small = 1;
        // (the resident set: CSV_IW_WAVES wavefronts per SIMD on every CU.  The units are dealt longest first, so a grid
        // that is resident at once finishes sooner than one whose last workgroups wait for a slot: 23.1 vs 23.5 us on cfg3
        // with 1536 vs 3072 workgroups)
        const int g_res = c->n_cu * CSV_IW_WAVES;
        int g_iw = div_up(B.cap_items, 4) < g_res ? div_up(B.cap_items, 4) : g_res;
        if (O.iw_grid > 0) g_iw = O.iw_grid;              // tuning aid
        if (g_iw < 1) g_iw = 1;
        // The tiers above 64 signatures usually have nothing to do (a 30x genome has no such cluster).  With the answer of THIS
        // run's k_chain_apply in hand (one-shot calls, see `peek` above) only the tiers with work are queued; the wait ends when
        // the position column has crossed the link and the two chain kernels have run, while the stream goes on to fetch / wait
        // for the other columns - it never runs dry because of it.  No answer within 20 ms: everything is queued.
        bool need_big = true;
        if (B.host_flag) {
            const auto t_end = std::chrono::steady_clock::now() + std::chrono::milliseconds(20);
            for (;;) {
                const unsigned long long w = *(volatile unsigned long long*)c->h_flag;
                if ((unsigned)(w >> 32) == (unsigned)B.run_seq) { need_big = (unsigned)w > 0; break; }
                if (std::chrono::steady_clock::now() > t_end) break;
                __builtin_ia32_pause();
            }
        }
        // The tiers run one after the other.  Side by side (CSV_TIER_FORK_MIN=<signatures>; the default for >= 4 Mi until r05) the
        // register tier and the one-wavefront tier - both bound by vector issue - share the CUs and finish when their sum would
        // have (r06 timeline of the 90x genome: 45 + 90 us overlapped = 99 us, against 37 + 61 in a row), and the fork and the
        // join add 8 + 14 us of event waits: 256.5 -> 246 us for the step without it.
        const bool tier_fork = fork && W >= (i64)O.tier_fork_min;
        // with clusters above 64 signatures in the batch, the one-wavefront tier for 65 .. 256 also takes the DUP / INV / TRA clusters
        // of at most 64 (its second phase): one grid, the long clusters first, instead of two kernels in a row
        B.pair_in_mid = (need_big && c->any_pair && !O.no_pair_in_mid) ? 1 : 0;
        const bool side_b = tier_fork && need_big, side_c = tier_fork && c->any_pair && !B.pair_in_mid;
        if (!tier_fork) { sB = st; sC = st; }
        if (side_b || side_c) {
            HIP_TRY(c, hipEventRecord(c->ev_sel, st));
            if (side_b) HIP_TRY(c, hipStreamWaitEvent(sB, c->ev_sel, 0));
            if (side_c) HIP_TRY(c, hipStreamWaitEvent(sC, c->ev_sel, 0));
        }
        if (B.a.p32) LAUNCH("refine_indel_wave", k_refine_indel_wave<true>, g_iw, 256, 0, B);
        else LAUNCH("refine_indel_wave", k_refine_indel_wave<false>, g_iw, 256, 0, B);
        if (c->any_pair && !B.pair_in_mid) LAUNCH_ON(sC, "refine_wave", (k_refine<64, 64, false>), g_small, 64, LDS_SMALL, B, 0, 64);
        else HIP_TRY(c, mark());
        const int mid_cap = O.mid_grid > 0 ? O.mid_grid : 8192, big_cap = O.big_grid > 0 ? O.big_grid : 512;
        int g_mid = B.cap_items < mid_cap ? B.cap_items : mid_cap;
        if (g_mid < 1) g_mid = 1;
        int g_big = B.cap_items < big_cap ? B.cap_items : big_cap;
        if (g_big < 1) g_big = 1;
        if (need_big) {
            LAUNCH_ON(sB, "refine_mid", (k_refine<64, 256, true>), g_mid, 64, LDS_MID, B, 64, MID_CAP);
            LAUNCH_ON(sB, "refine_block", (k_refine<256, 2048, true>), g_big, 256, LDS_BIG, B, MID_CAP, 0x7fffffff);
        } else { HIP_TRY(c, mark()); HIP_TRY(c, mark()); }
        if (side_b) { HIP_TRY(c, hipEventRecord(c->ev_aux[0], sB)); HIP_TRY(c, hipStreamWaitEvent(st, c->ev_aux[0], 0)); }
        if (side_c) { HIP_TRY(c, hipEventRecord(c->ev_aux[1], sC)); HIP_TRY(c, hipStreamWaitEvent(st, c->ev_aux[1], 0)); }
        LAUNCH("items_scan", k_items_scan, B.cap_items / IS_CHUNK + 1, 64 * IS_NW, 0, B);
        LAUNCH("emit", k_emit, 2048, 256, 0, B);
        if (swap) {                                       // the clustering chain joins the main stream
            HIP_TRY(c, hipEventRecord(c->ev_aux[2], st));
            HIP_TRY(c, hipStreamWaitEvent(sM, c->ev_aux[2], 0));
            st = sM;
        }
        if (do_gt) {
            if (swap) {}
            else if (fork && !keep_reads) HIP_TRY(c, hipStreamWaitEvent(st, c->ev_aux[2], 0));
            else { HIP_TRY(c, hipStreamWaitEvent(st, c->ev_reads, 0)); const int rc = reads_stage(st); if (rc) return rc; }
            // the second pass (overflow list of the first; global tables beyond) has usually nothing to do and is queued all the
            // same: whether it has is known when the first pass ends, and nothing is carried over from earlier runs
            const int gt_grid = O.gt_grid > 0 ? O.gt_grid : GT_GRID;
            if (B.r_start.p32) {
                hipLaunchKernelGGL((k_genotype<1024, 4, false, true>), dim3(gt_grid), dim3(256), 0, st, B);
                hipLaunchKernelGGL((k_genotype<8192, 4, true, true>), dim3(256), dim3(256), 0, st, B);
            } else {
                hipLaunchKernelGGL((k_genotype<1024, 4, false, false>), dim3(gt_grid), dim3(256), 0, st, B);
                hipLaunchKernelGGL((k_genotype<8192, 4, true, false>), dim3(256), dim3(256), 0, st, B);
            }
#ifdef CSV_GT_PROF
            hipLaunchKernelGGL(k_gt_prof_print, dim3(1), dim3(1), 0, st, B);
#endif
            DBG("genotype");
            HIP_TRY(c, mark());
        } else if (stats) { for (int q = 0; q < 4; q++) HIP_TRY(c, mark()); }
        if (c->any_tra_gt) {
            // (reads_off / contig_len / the reads columns travel on side[2]: a batch whose only genotyped segments are TRA
            // segments, or one without reads, has not waited for them yet)
            if (c->copies_pending && c->have_tab) HIP_TRY(c, hipStreamWaitEvent(st, c->ev_reads, 0));
            if (B.r_start.p32) LAUNCH("genotype_tra", k_genotype_tra<true>, 256, 64, 0, B);
            else LAUNCH("genotype_tra", k_genotype_tra<false>, 256, 64, 0, B);
        } else HIP_TRY(c, mark());
        // one more record with nothing in front of it: what a slot reads when it holds no kernel (the event records occupy the
        // stream themselves).  bench.py subtracts THIS from the kernel slots instead of guessing an empty one.
        HIP_TRY(c, mark());
    }
#undef LAUNCH
#undef LAUNCH_ON
    if (c->copies_pending) {
        // whatever this run did not consume is still waited for before the call returns (an empty batch; a reads table next
        // to zero signatures): the caller's page-locked columns must not be the source of a copy in flight after the call,
        // and the next upload re-plans the arena
        HIP_TRY(c, hipStreamWaitEvent(st, c->ev_copy[1], 0));
        if (c->have_tab) HIP_TRY(c, hipStreamWaitEvent(st, c->ev_reads, 0));
        c->copies_pending = false;
    }
    HIP_TRY(c, hipGetLastError());
    c->ran = true;
    if (stats) {
        memset(stats, 0, sizeof *stats);
        const int rc = read_counters(c);
        if (rc) return rc;
        for (int i = 0; i + 1 < ev && i < CSV_N_STAGES; i++) HIP_TRY(c, hipEventElapsedTime(&stats->ms_stage[i], c->ev[i], c->ev[i + 1]));
        HIP_TRY(c, hipEventElapsedTime(&stats->ms_total, c->ev[0], c->ev[ev - 1]));
        stats->n_clusters = c->h_cnt.n_clusters;
        stats->n_work_block = c->h_cnt.n_items_big;
        stats->n_work_wave = c->h_cnt.n_items - c->h_cnt.n_items_big;
        stats->n_calls = c->h_cnt.n_calls;
        stats->n_support = c->h_cnt.n_support;
    }
    return CSV_OK;
}

// device counters -> c->h_cnt (through the page-locked block).  A reads table that the run-level reorder could not
// handle switches the batch to the general sort and runs it again, once.
int read_counters(csv_ctx* c)
{
    hipStream_t st = c->stream;
    for (int attempt = 0; attempt < 2; attempt++) {
        HIP_TRY(c, hipMemcpyAsync(c->h_pin, c->B.cnt, sizeof(DevCounters), hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipStreamSynchronize(st));
        memcpy(&c->h_cnt, c->h_pin, sizeof(DevCounters));
        {   // the reads-order state of the upload lives outside the per-run counters
            ReadsState rs{};
            HIP_TRY(c, hipMemcpy(&rs, c->rstate.p, sizeof rs, hipMemcpyDeviceToHost));
            c->h_cnt.n_runs = rs.n_runs; c->h_cnt.ro_state = rs.ro_state; c->h_cnt.error |= rs.error;
        }
        if (c->opt.debug_counters)
            fprintf(stderr, "[csv] counters: clusters %d items %d calls %d error %d | reads: mode %d runs %d state %d | gt_over %d gt_huge %d tra_huge %d\n",
                    c->h_cnt.n_clusters, c->h_cnt.n_items, c->h_cnt.n_calls, c->h_cnt.error, c->B.ro_mode, c->h_cnt.n_runs, c->h_cnt.ro_state,
                    c->h_cnt.n_gt_over, c->h_cnt.n_gt_huge, c->h_cnt.n_tra_huge);
        if (c->B.ro_mode == 1 && c->B.n_reads > 0 && c->any_genotype && c->h_cnt.ro_state == RO_NEED_GENERAL && attempt == 0) {
            c->reads_general = true;
            c->B.ro_mode = 2;
            c->reads_ready = false;
            const int rc = run_impl(c, nullptr);
            if (rc) return rc;
            continue;
        }
        break;
    }
    return CSV_OK;
}

}  // namespace

extern "C" {

int csv_batch_run(csv_ctx* c, csv_run_stats* stats)
{
    if (!c) return CSV_E_INVALID;
    if (!c->uploaded) return fail(c, CSV_E_STATE, "csv_batch_run before csv_batch_upload");
    return run_impl(c, stats);
}

int csv_measure_copy_bandwidth(csv_ctx* c, int64_t bytes, int reps, double* gb_per_s)
{
    if (!c || !gb_per_s || bytes <= 0 || reps <= 0) return CSV_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    void *src = nullptr, *dst = nullptr;
    if (hipMalloc(&src, (size_t)bytes) != hipSuccess || hipMalloc(&dst, (size_t)bytes) != hipSuccess) {
        if (src) (void)hipFree(src);
        return fail(c, CSV_E_NOMEM, "copy-bandwidth buffers (%lld bytes each)", (long long)bytes);
    }
    (void)hipMemsetAsync(src, 1, (size_t)bytes, c->stream);
    float best = 1e30f;
    hipError_t e = hipSuccess;
    for (int r = 0; r < reps && e == hipSuccess; r++) {
        e = hipEventRecord(c->ev[0], c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToDevice, c->stream);
        if (e == hipSuccess) e = hipEventRecord(c->ev[1], c->stream);
        if (e == hipSuccess) e = hipEventSynchronize(c->ev[1]);
        float ms = 0;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, c->ev[0], c->ev[1]);
        if (e == hipSuccess && ms < best) best = ms;
    }
    (void)hipFree(src); (void)hipFree(dst);
    if (e != hipSuccess) return fail(c, CSV_E_HIP, "copy-bandwidth measurement: %s", hipGetErrorString(e));
    *gb_per_s = 2.0 * (double)bytes / ((double)best * 1e-3) / 1e9;
    return CSV_OK;
}

int csv_cache_flush(csv_ctx* c, int64_t bytes)
{
    if (!c || bytes <= 0) return CSV_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    const int rc = reserve(c, c->flush, (size_t)bytes);
    if (rc) return rc;
    HIP_TRY(c, hipMemsetAsync(c->flush.p, 0x5a, (size_t)bytes, c->stream));
    return CSV_OK;
}

int csv_batch_validate(csv_ctx* c)
{
    if (!c) return CSV_E_INVALID;
    if (!c->uploaded) return fail(c, CSV_E_STATE, "csv_batch_validate before csv_batch_upload");
    if (c->partial_cols) return fail(c, CSV_E_STATE, "csv_batch_validate needs a csv_batch_upload: a csv_cluster_batch call from page-locked columns keeps only the rows its kernels read");
    if (c->n_pend) return fail(c, CSV_E_STATE, "csv_batch_validate while %d asynchronous publish(es) are in flight: csv_batch_publish_wait first", c->n_pend);
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    // k_validate_order reports through the batch's counter pointer.  That pointer alternates between the two result arenas
    // from run to run (advisor, r05: clearing and reading arena 0 while the kernel wrote arena 1 returned CSV_OK for an
    // unsorted batch after an odd number of runs): the check gets a counter block of its own, which no run and no publish uses.
    DevBatch V = c->B;
    V.cnt = (DevCounters*)((char*)c->cnt.p + 512);
    HIP_TRY(c, hipMemsetAsync(V.cnt, 0, sizeof(DevCounters), st));
    if (V.W > 0) hipLaunchKernelGGL(k_validate_order, dim3(div_up(V.W, 256)), dim3(256), 0, st, V);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(c->h_pin, V.cnt, sizeof(DevCounters), hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    memcpy(&c->h_cnt, c->h_pin, sizeof(DevCounters));
    c->ran = false;
    if (c->h_cnt.error & ERR_SIG_ORDER)
        return fail(c, CSV_E_UNSORTED, "a segment is not in the rebuild order of cuteSV (main script :764-802) or holds adjacent duplicates");
    return CSV_OK;
}

// Device -> host: the counters, then ONE copy of the call records and one of the support lists into the page-locked
// block, unpacked into the caller's arrays on the host (a few MB; the per-signature outputs only when they were asked
// for at upload).
// Are all of the caller's call arrays page-locked (device addressable)?  Then fill `P` with their device addresses.
constexpr int PUB_MAX_SPANS = 3;
struct PubLayout {                                   // the result arrays as the caller laid them out: runs of exactly adjacent arrays
    const char* host[15]; size_t bytes[15]; size_t stage_off[15];
    struct Span { const char* host; size_t bytes, stage_off; } span[15];
    int n_span = 0; size_t stage_bytes = 0;
};
static void publish_spans(PubLayout& L)
{
    int idx[15], n = 0;
    for (int i = 0; i < 15; i++) if (L.host[i] && L.bytes[i]) idx[n++] = i;
    std::sort(idx, idx + n, [&](int x, int y) { return L.host[x] < L.host[y]; });
    L.n_span = 0; L.stage_bytes = 0;
    for (int q = 0; q < n; q++) {
        const int i = idx[q];
        if (L.n_span && L.span[L.n_span - 1].host + L.span[L.n_span - 1].bytes == L.host[i]) {
            auto& sp = L.span[L.n_span - 1];
            L.stage_off[i] = sp.stage_off + sp.bytes; sp.bytes += L.bytes[i];
        } else {
            // (a run starts at the same offset modulo 256 as on the host: every array keeps its alignment in the image)
            const size_t off = ((L.stage_bytes + 255) & ~(size_t)255) + ((uintptr_t)L.host[i] & 255);
            L.span[L.n_span++] = {L.host[i], L.bytes[i], off};
            L.stage_off[i] = off;
        }
        L.stage_bytes = L.span[L.n_span - 1].stage_off + L.span[L.n_span - 1].bytes;
    }
}
bool publish_targets(csv_ctx* c, const csv_batch_out* out, PublishArgs& P, PubLayout* L = nullptr, char* stage = nullptr)
{
    (void)c;
    if (c->opt.no_publish || out->cap_calls < 0 || out->cap_support < 0) return false;
    const size_t nc = (size_t)out->cap_calls, ns = (size_t)out->cap_support;
    const bool sup32 = out->support_sig32 != nullptr, nosup = (out->flags & CSV_OUT_NO_SUPPORT_LIST) != 0;
    const size_t cw = (out->flags & CSV_OUT_COORD_I32) ? 4 : 8;
    const void* host[15] = {out->call_seg, out->call_cluster, out->call_aux, out->support, out->cipos, out->cilen, out->dr, out->dv, out->gl_idx,
                            out->bp1, out->bp2, out->search_pos, out->seq_pick, out->support_off, sup32 ? (const void*)out->support_sig32 : (const void*)out->support_sig};
    const size_t bytes[15] = {nc * 4, nc * 4, nc * 4, nc * 4, nc * 4, nc * 4, nc * 4, nc * 4, nc * 4, nc * cw, nc * cw, nc * cw, nc * cw, (nc + 1) * 8, ns * (sup32 ? 4 : 8)};
    // (ABI v7) optional fields may be NULL: not written; every array that IS given must be page-locked
    const bool required[15] = {true, false, false, true, false, false, false, false, false, true, true, false, false, !nosup, !nosup};
    void* dev[15];
    for (int i = 0; i < 15; i++) {
        dev[i] = nullptr;
        if (nosup && i >= 13) continue;
        if (!host[i]) { if (required[i]) return false; continue; }
        if (stage) { dev[i] = stage + L->stage_off[i]; continue; }          // (second pass: the device image of a laid-out result)
        dev[i] = pinned_device_address(host[i], bytes[i] ? bytes[i] : 1);
        if (!dev[i]) return false;
    }
    if (L && !stage) {
        for (int i = 0; i < 15; i++) { L->host[i] = dev[i] ? (const char*)host[i] : nullptr; L->bytes[i] = dev[i] ? bytes[i] : 0; L->stage_off[i] = 0; }
        publish_spans(*L);
    }
    P.call_seg = (int*)dev[0]; P.call_cluster = (int*)dev[1]; P.call_aux = (int*)dev[2]; P.support = (int*)dev[3]; P.cipos = (int*)dev[4];
    P.cilen = (int*)dev[5]; P.dr = (int*)dev[6]; P.dv = (int*)dev[7]; P.gl_idx = (int*)dev[8];
    P.bp1 = dev[9]; P.bp2 = dev[10]; P.search_pos = dev[11]; P.seq_pick = dev[12]; P.support_off = (i64*)dev[13];
    P.support_sig = sup32 ? nullptr : (i64*)dev[14]; P.support_sig32 = sup32 ? (int*)dev[14] : nullptr;
    P.coord32 = cw == 4; P.no_support = nosup;
    return true;
}

int csv_batch_download(csv_ctx* c, csv_batch_out* out)
{
    if (!c || !out) return CSV_E_INVALID;
    if (!c->ran) return fail(c, CSV_E_STATE, "csv_batch_download before csv_batch_run");
    if (c->n_pend) return fail(c, CSV_E_STATE, "csv_batch_download while %d asynchronous publish(es) are in flight: csv_batch_publish_wait first", c->n_pend);
    const bool nosup = (out->flags & CSV_OUT_NO_SUPPORT_LIST) != 0, coord32 = (out->flags & CSV_OUT_COORD_I32) != 0;
    if (!nosup && ((out->support_sig != nullptr) == (out->support_sig32 != nullptr) || !out->support_off))
        return fail(c, CSV_E_INVALID, "csv_batch_out: support_off and exactly one of support_sig / support_sig32 must be given (or CSV_OUT_NO_SUPPORT_LIST)");
    if (!out->call_seg || !out->bp1 || !out->bp2 || !out->support) return fail(c, CSV_E_INVALID, "csv_batch_out: call_seg, bp1, bp2 and support are required");
    if (coord32 && !c->B.a.p32) return fail(c, CSV_E_INVALID, "CSV_OUT_COORD_I32 needs a batch of CSV_IN_SIG_I32 columns");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const int S0 = (int)c->h_seg.size();
    PublishArgs P{};
    bool published = false;
    if (publish_targets(c, out, P)) {
        // page-locked result arrays: k_publish writes everything in place; the download is one synchronisation
        const size_t o_cnt = 0, o_err = 256, need = o_err + (size_t)(S0 + 1) * 4;
        static_assert(sizeof(DevCounters) <= 256, "counters landing zone");
        if (need > c->h_pin_cap) { HIP_TRY(c, hipStreamSynchronize(st)); const int rc = pin_reserve(c, need); if (rc) return rc; }
        void* dpin = nullptr;
        HIP_TRY(c, hipHostGetDevicePointer(&dpin, c->h_pin, 0));
        P.cap_calls = out->cap_calls; P.cap_support = out->cap_support; P.n_seg = S0;
        P.h_cnt = (DevCounters*)((char*)dpin + o_cnt); P.h_seg_err = (int*)((char*)dpin + o_err);
        for (int attempt = 0; attempt < 2; attempt++) {
            hipLaunchKernelGGL(k_publish, dim3(512), dim3(256), 0, st, c->B, P);
            HIP_TRY(c, hipStreamSynchronize(st));
            memcpy(&c->h_cnt, c->h_pin + o_cnt, sizeof(DevCounters));
            if (c->opt.debug_counters)
                fprintf(stderr, "[csv] counters (published): clusters %d items %d calls %d error %d | reads: mode %d runs %d state %d\n",
                        c->h_cnt.n_clusters, c->h_cnt.n_items, c->h_cnt.n_calls, c->h_cnt.error, c->B.ro_mode, c->h_cnt.n_runs, c->h_cnt.ro_state);
            if (c->B.ro_mode == 1 && c->B.n_reads > 0 && c->any_genotype && c->h_cnt.ro_state == RO_NEED_GENERAL && attempt == 0) {
                c->reads_general = true;             // (as read_counters does: the batch again, through the general sort)
                c->B.ro_mode = 2;
                c->reads_ready = false;
                const int rc = run_impl(c, nullptr);
                if (rc) return rc;
                continue;
            }
            break;
        }
        published = true;
    } else {
        const int rc = read_counters(c);
        if (rc) return rc;
    }
    c->settled = true;
    const DevCounters& k = c->h_cnt;
    out->n_calls = k.n_calls; out->n_support = k.n_support; out->n_clusters = k.n_clusters;
    if (k.error & ERR_READS_UNSORTED) return fail(c, CSV_E_UNSORTED, "a reads block is not sorted by start although CSV_IN_READS_SORTED was set");
    if (k.error & ERR_CLUSTER_TOO_BIG) return fail(c, CSV_E_INVALID, "a chained cluster has more than %lld signatures", (long long)MAX_CLUSTER);
    if (k.error & ERR_KEY_RANGE) return fail(c, CSV_E_INVALID, "a read id is negative, or a read end is negative or >= 2^40");
    if (k.error & ERR_COVER_OVERFLOW) return fail(c, CSV_E_INVALID, "internal: the genotype hash pool was too small");
    if (k.error & ERR_TRA_CHROM) return fail(c, CSV_E_INVALID, "a TRA call names a mate chromosome outside the reads table");
    if (k.error & ERR_TMP_OVERFLOW) return fail(c, CSV_E_INVALID, "internal: temp call capacity exceeded");
    if ((out->cluster_id || out->allele_id) && !c->B.per_sig)
        return fail(c, CSV_E_STATE, "cluster_id / allele_id requested but the batch was uploaded without CSV_IN_PER_SIG");
    if (k.n_calls > out->cap_calls || (!nosup && k.n_support > out->cap_support))
        return fail(c, CSV_E_CAPACITY, "need %d calls / %lld supports", k.n_calls, (long long)k.n_support);
    const size_t nc = (size_t)k.n_calls; size_t ns = (size_t)k.n_support;
    const DevBatch& B = c->B;
    const int S = (int)c->h_seg.size();
    const size_t o_rec = 256, o_err = published ? 256 : o_rec + ((nc * sizeof(CallRec) + 255) & ~(size_t)255), o_end = o_err + (size_t)(S + 1) * 4;
    int* sup_stage = nullptr;
    if (!published) {
    if (nosup) ns = 0;                                       // (the staging path below moves no support list then)
    if (o_end + (out->support_sig32 ? 0 : ns * 4) + 64 > c->h_pin_cap) { const int rc = pin_reserve(c, o_end + (out->support_sig32 ? 0 : ns * 4) + 64); if (rc) return rc; }
    // the call records first: they are unpacked on the host while the (larger) support list is still on its way
    if (nc) HIP_TRY(c, hipMemcpyAsync(c->h_pin + o_rec, B.o_rec, nc * sizeof(CallRec), hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipEventRecord(c->ev_sel, st));
    // the support list is int32 on the device: straight into the caller's int32 array, or widened on the host behind the copy
    if (ns && out->support_sig32) HIP_TRY(c, hipMemcpyAsync(out->support_sig32, B.o_supsig, ns * 4, hipMemcpyDeviceToHost, st));
    else if (ns) { sup_stage = (int*)(c->h_pin + o_end); HIP_TRY(c, hipMemcpyAsync(sup_stage, B.o_supsig, ns * 4, hipMemcpyDeviceToHost, st)); }
    if (S) HIP_TRY(c, hipMemcpyAsync(c->h_pin + o_err, B.seg_err, (size_t)S * 4, hipMemcpyDeviceToHost, st));
    }
    if (out->cluster_id) memset(out->cluster_id, 0xff, (size_t)c->n_sig_host * 4);
    if (out->allele_id) memset(out->allele_id, 0xff, (size_t)c->n_sig_host * 4);
    if (out->cluster_id || out->allele_id) {
        for (int s = 0; s < S;) {
            int e = s;
            while (e + 1 < S && c->h_seg[e + 1].sig_begin == c->h_seg[e].sig_end) e++;
            const i64 dst = c->h_seg[s].sig_begin, n = c->h_woff[e + 1] - c->h_woff[s], src = c->h_woff[s];
            if (n > 0 && out->cluster_id) HIP_TRY(c, hipMemcpyAsync(out->cluster_id + dst, B.cluster_id + src, n * 4, hipMemcpyDeviceToHost, st));
            if (n > 0 && out->allele_id) HIP_TRY(c, hipMemcpyAsync(out->allele_id + dst, B.allele_id + src, n * 4, hipMemcpyDeviceToHost, st));
            s = e + 1;
        }
    }
    if (!published) {
        HIP_TRY(c, hipEventSynchronize(c->ev_sel));
        const CallRec* r = (const CallRec*)(c->h_pin + o_rec);
        for (size_t i = 0; i < nc; i++) {
            const CallRec& x = r[i];
            out->call_seg[i] = x.seg; out->support[i] = x.support;
            if (out->call_cluster) out->call_cluster[i] = x.cluster;
            if (out->call_aux) out->call_aux[i] = x.aux;
            if (out->cipos) out->cipos[i] = x.cipos;
            if (out->cilen) out->cilen[i] = x.cilen;
            if (coord32) {
                ((int32_t*)out->bp1)[i] = (int32_t)x.bp1; ((int32_t*)out->bp2)[i] = (int32_t)x.bp2;
                if (out->search_pos) ((int32_t*)out->search_pos)[i] = (int32_t)x.search;
                if (out->seq_pick) ((int32_t*)out->seq_pick)[i] = (int32_t)x.pick;
            } else {
                out->bp1[i] = x.bp1; out->bp2[i] = x.bp2;
                if (out->search_pos) out->search_pos[i] = x.search;
                if (out->seq_pick) out->seq_pick[i] = x.pick;
            }
            if (out->dr) out->dr[i] = x.dr;
            if (out->dv) out->dv[i] = x.dv;
            if (out->gl_idx) out->gl_idx[i] = x.gl;
            if (!nosup) out->support_off[i] = x.supoff;
        }
        if (!nosup) out->support_off[nc] = (int64_t)ns;
    }
    if (!published || out->cluster_id || out->allele_id) HIP_TRY(c, hipStreamSynchronize(st));
    if (sup_stage) for (size_t i = 0; i < ns; i++) out->support_sig[i] = sup_stage[i];
    if (out->seg_status && S) memcpy(out->seg_status, c->h_pin + o_err, (size_t)S * 4);
    return CSV_OK;
}

// ---- pipelined delivery: the result of run k crosses PCIe while run k + 1 computes
static int result_status(csv_ctx* c, const DevCounters& k, csv_batch_out* out, bool nosup)
{
    out->n_calls = k.n_calls; out->n_support = k.n_support; out->n_clusters = k.n_clusters;
    if (k.error & ERR_READS_UNSORTED) return fail(c, CSV_E_UNSORTED, "a reads block is not sorted by start although CSV_IN_READS_SORTED was set");
    if (k.error & ERR_CLUSTER_TOO_BIG) return fail(c, CSV_E_INVALID, "a chained cluster has more than %lld signatures", (long long)MAX_CLUSTER);
    if (k.error & ERR_KEY_RANGE) return fail(c, CSV_E_INVALID, "a read id is negative, or a read end is negative or >= 2^40");
    if (k.error & ERR_COVER_OVERFLOW) return fail(c, CSV_E_INVALID, "internal: the genotype hash pool was too small");
    if (k.error & ERR_TRA_CHROM) return fail(c, CSV_E_INVALID, "a TRA call names a mate chromosome outside the reads table");
    if (k.error & ERR_TMP_OVERFLOW) return fail(c, CSV_E_INVALID, "internal: temp call capacity exceeded");
    if (k.n_calls > out->cap_calls || (!nosup && k.n_support > out->cap_support))
        return fail(c, CSV_E_CAPACITY, "need %d calls / %lld supports", k.n_calls, (long long)k.n_support);
    return CSV_OK;
}

int csv_batch_publish_async(csv_ctx* c, csv_batch_out* out)
{
    if (!c || !out) return CSV_E_INVALID;
    if (!c->ran) return fail(c, CSV_E_STATE, "csv_batch_publish_async before csv_batch_run");
    if (!c->settled) return fail(c, CSV_E_STATE, "csv_batch_publish_async needs one csv_batch_download of this upload first (it settles how the reads table is ordered and what the result needs)");
    if (c->B.per_sig || out->cluster_id || out->allele_id) return fail(c, CSV_E_INVALID, "csv_batch_publish_async delivers no per-signature outputs");
    if (c->pend[c->parity].live) return fail(c, CSV_E_STATE, "this run's result is already being published");
    if (c->n_pend >= 2) return fail(c, CSV_E_STATE, "two publishes in flight: csv_batch_publish_wait first");
    const bool nosup = (out->flags & CSV_OUT_NO_SUPPORT_LIST) != 0, coord32 = (out->flags & CSV_OUT_COORD_I32) != 0;
    if (!nosup && ((out->support_sig != nullptr) == (out->support_sig32 != nullptr) || !out->support_off))
        return fail(c, CSV_E_INVALID, "csv_batch_out: support_off and exactly one of support_sig / support_sig32 must be given (or CSV_OUT_NO_SUPPORT_LIST)");
    if (!out->call_seg || !out->bp1 || !out->bp2 || !out->support) return fail(c, CSV_E_INVALID, "csv_batch_out: call_seg, bp1, bp2 and support are required");
    if (coord32 && !c->B.a.p32) return fail(c, CSV_E_INVALID, "CSV_OUT_COORD_I32 needs a batch of CSV_IN_SIG_I32 columns");
    HIP_TRY(c, hipSetDevice(c->device));
    PublishArgs P{};
    PubLayout L;
    if (!publish_targets(c, out, P, &L)) return fail(c, CSV_E_INVALID, "csv_batch_publish_async writes the result in place: every array of csv_batch_out must be page-locked (csv_host_alloc / csv_host_register)");
    const int S0 = (int)c->h_seg.size(), p = c->parity;
    const size_t zone = (256 + (size_t)(S0 + 1) * 4 + 255) & ~(size_t)255;
    if (2 * zone > c->h_pub_cap) {
        if (c->n_pend) return fail(c, CSV_E_STATE, "internal: landing zones in use");
        if (c->h_pub) { HIP_TRY(c, hipHostFree(c->h_pub)); c->h_pub = nullptr; c->h_pub_cap = 0; }
        void* hp = nullptr;
        if (hipHostMalloc(&hp, 2 * zone + 4096, hipHostMallocDefault) != hipSuccess) return fail(c, CSV_E_NOMEM, "hipHostMalloc for the publish landing zones failed");
        c->h_pub = (char*)hp; c->h_pub_cap = 2 * zone + 4096;
    }
    void* dpub = nullptr;
    HIP_TRY(c, hipHostGetDevicePointer(&dpub, c->h_pub, 0));
    const size_t half = c->h_pub_cap / 2 & ~(size_t)255;
    // Block delivery (arrays back to back in page-locked memory: engine.result_buffers lays them out so).  A kernel that stores
    // across PCIe holds up every kernel BOUNDARY of the next run beside it - the end-of-kernel cache write-back waits for the
    // posted writes in flight, whoever issued them (measured: k_chain_apply 6 -> 62 us next to a k_publish of 3.4 MB, a step 110 us
    // = run + delivery, nothing overlapped) - while a copy-engine transfer leaves the kernels alone.  So: the image is written into
    // device memory behind the run's own kernels (main stream, ~4 us) and the copy engine moves it under the next run.
    const bool block = L.n_span > 0 && L.n_span <= PUB_MAX_SPANS && !c->opt.pub_inplace;
    if (block) {
        if (L.stage_bytes + 256 > c->pub_stage_cap[p]) {
            if (c->pub_stage[p]) { HIP_TRY(c, hipStreamSynchronize(c->pub)); HIP_TRY(c, hipFree(c->pub_stage[p])); c->pub_stage[p] = nullptr; c->pub_stage_cap[p] = 0; }
            const size_t cap = L.stage_bytes + L.stage_bytes / 8 + 4096;
            if (hipMalloc(&c->pub_stage[p], cap) != hipSuccess) return fail(c, CSV_E_NOMEM, "hipMalloc(%zu) for the result image failed", cap);
            c->pub_stage_cap[p] = cap;
        }
        if (!publish_targets(c, out, P, &L, (char*)c->pub_stage[p])) return fail(c, CSV_E_INVALID, "internal: result image");
    }
    P.cap_calls = out->cap_calls; P.cap_support = out->cap_support; P.n_seg = S0;
    P.h_cnt = (DevCounters*)((char*)dpub + half * p); P.h_seg_err = (int*)((char*)dpub + half * p + 256);
    const hipStream_t ps = c->pub;
    if (block) {
        hipLaunchKernelGGL(k_publish, dim3(512), dim3(256), 0, c->stream, c->B, P);       // (c->B points at arena p: the last run's)
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipEventRecord(c->ev_run[p], c->stream));
        HIP_TRY(c, hipStreamWaitEvent(ps, c->ev_run[p], 0));
        for (int q = 0; q < L.n_span; q++)
            HIP_TRY(c, hipMemcpyAsync((void*)L.span[q].host, (char*)c->pub_stage[p] + L.span[q].stage_off, L.span[q].bytes, hipMemcpyDeviceToHost, ps));
    } else {
        // (the run's kernels are all in the main stream's queue: an event recorded now marks their end - a plain run pays nothing for it)
        HIP_TRY(c, hipEventRecord(c->ev_run[p], c->stream));
        HIP_TRY(c, hipStreamWaitEvent(c->pub, c->ev_run[p], 0));
        hipLaunchKernelGGL(k_publish, dim3(512), dim3(256), 0, c->pub, c->B, P);
        HIP_TRY(c, hipGetLastError());
    }
    HIP_TRY(c, hipEventRecord(c->ev_pub[p], ps));
    c->pend[p].out = out; c->pend[p].live = true;
    c->pend_order[c->n_pend++] = p;
    return CSV_OK;
}

int csv_batch_publish_wait(csv_ctx* c, csv_batch_out** done)
{
    if (!c) return CSV_E_INVALID;
    if (done) *done = nullptr;
    if (!c->n_pend) return fail(c, CSV_E_STATE, "csv_batch_publish_wait: nothing in flight");
    HIP_TRY(c, hipSetDevice(c->device));
    const int p = c->pend_order[0];
    HIP_TRY(c, hipEventSynchronize(c->ev_pub[p]));
    c->pend_order[0] = c->pend_order[1]; c->n_pend--;
    csv_batch_out* out = c->pend[p].out;
    c->pend[p].live = false; c->pend[p].out = nullptr;
    if (done) *done = out;
    const size_t half = c->h_pub_cap / 2 & ~(size_t)255;
    DevCounters k;
    memcpy(&k, c->h_pub + half * p, sizeof k);
    const int S = (int)c->h_seg.size();
    if (out->seg_status && S) memcpy(out->seg_status, c->h_pub + half * p + 256, (size_t)S * 4);
    return result_status(c, k, out, (out->flags & CSV_OUT_NO_SUPPORT_LIST) != 0);
}

// room for `extra` more rows in the pool (grows by copying: the pool is not in an arena)
static int pool_reserve(csv_ctx* c, i64 extra)
{
    const i64 need = c->pool_n + extra;
    if (need <= c->pool_cap) return CSV_OK;
    if (need >= (1ll << 31) - 4096) return fail(c, CSV_E_INVALID, "signature pool too large (%lld rows)", (long long)need);
    const i64 cap = need + need / 2 + 4096;
    Buf* cols[5] = {&c->pool_seg, &c->pool_a, &c->pool_b, &c->pool_read, &c->pool_aux};
    const size_t w[5] = {4, 8, 8, 4, 4};
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (int k = 0; k < 5; k++) {
        void* p = nullptr;
        hipError_t e = hipMalloc(&p, (size_t)cap * w[k]);
        if (e != hipSuccess) return fail(c, CSV_E_NOMEM, "hipMalloc(%zu) for the signature pool failed: %s", (size_t)cap * w[k], hipGetErrorString(e));
        if (c->pool_n > 0) HIP_TRY(c, hipMemcpy(p, cols[k]->p, (size_t)c->pool_n * w[k], hipMemcpyDeviceToDevice));
        if (cols[k]->p) HIP_TRY(c, hipFree(cols[k]->p));
        cols[k]->p = p; cols[k]->cap = (size_t)cap * w[k];
    }
    c->pool_cap = cap;
    return CSV_OK;
}

int csv_pool_reset(csv_ctx* c)
{
    if (!c) return CSV_E_INVALID;
    c->pool_n = 0;
    return CSV_OK;
}

int csv_pool_rows(const csv_ctx* c, int64_t* n_rows)
{
    if (!c || !n_rows) return CSV_E_INVALID;
    *n_rows = c->pool_n;
    return CSV_OK;
}

int csv_pool_append(csv_ctx* c, int64_t n, const int32_t* seg_id, const int64_t* a, const int64_t* b, const int32_t* read, const int32_t* aux)
{
    if (!c || n < 0 || (n > 0 && (!seg_id || !a || !b || !read || !aux))) return CSV_E_INVALID;
    if (n == 0) return CSV_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    { const int rc = pool_reserve(c, n); if (rc) return rc; }
    hipStream_t st = c->stream;
    const i64 o = c->pool_n;
    HIP_TRY(c, hipMemcpyAsync(dp<int>(c->pool_seg) + o, seg_id, (size_t)n * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(dp<i64>(c->pool_a) + o, a, (size_t)n * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(dp<i64>(c->pool_b) + o, b, (size_t)n * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(dp<int>(c->pool_read) + o, read, (size_t)n * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(dp<int>(c->pool_aux) + o, aux, (size_t)n * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipStreamSynchronize(st));                   // (the caller's arrays may be pageable and are free again on return)
    c->pool_n += n;
    return CSV_OK;
}

int csv_rebuild_signatures(csv_ctx* c, const csv_rebuild_in* in, csv_rebuild_out* out)
{
    if (!c || !in || !out) return CSV_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    const bool from_pool = (in->flags & CSV_RB_FROM_POOL) != 0;
    const i64 n = from_pool ? c->pool_n : in->n;
    out->n_out = 0; out->ms_device = 0; out->n_passes = 0;
    if (n < 0 || n >= (1ll << 31) - 4096 || in->n_seg <= 0) return fail(c, CSV_E_INVALID, "bad rebuild input");
    if (from_pool && (!in->read_rank || in->n_rank <= 0 || !in->seg_aux_major)) return fail(c, CSV_E_INVALID, "CSV_RB_FROM_POOL needs read_rank");
    if (n == 0) return CSV_OK;
    // key widths (bits that are non-zero somewhere) and the validity of every row: found on the device, behind the upload (r04
    // walked the columns on the host first: ~3 ms for a 30x genome's 2.85 M rows, in front of a 0.5 ms sort)
    i64 mx_a = 0, mx_b = 0; int mx_rid = 0, mx_aux = 0, mx_seg = 0, mx_aux_all = 0;
    auto nbytes = [](u64 v) { int k = 0; while (v) { k++; v >>= 8; } return k; };
    auto nbits = [](u64 v) { int k = 0; while (v) { k++; v >>= 1; } return k; };
    const int nunits = div_up(n, SORT_WTILE), nblk = div_up(nunits, 4), ntile = div_up(n, 2048);
    Plan P;
#define PL(buf, bytes) P.add(c->buf, (size_t)(bytes))
    PL(rb_seg, n * 4); PL(rb_a, n * 8); PL(rb_b, n * 8); PL(rb_rid, n * 4); PL(rb_aux, n * 4); PL(rb_auxk, n * 4);
    PL(rb_major, in->n_seg); PL(rb_nodedup, in->n_seg); PL(rb_perm0, n * 4); PL(rb_perm1, n * 4); PL(rb_hist, (size_t)RS_RADIX * nunits * 4);
    PL(rb_tot, RS_RADIX * 4); PL(rb_partial, (ntile + 2) * 4);
    PL(rb_el0, (size_t)n * 32); PL(rb_el1, (size_t)n * 32);          // composite-key elements (16 or 32 bytes each; sized for either)
    PL(rb_oseg, n * 4); PL(rb_oa, n * 8); PL(rb_ob, n * 8); PL(rb_orid, n * 4); PL(rb_oaux, n * 4); PL(rb_osrc, n * 4); PL(rb_segcnt, ((size_t)in->n_seg + 2) * 8);
    if (from_pool) PL(rb_rank, (size_t)in->n_rank * 4);
    PL(rb_mx, 64);
    if (in->tie_order && in->seg_nodedup) PL(rb_drop, n + 64);
#undef PL
    {
        if (P.total > c->arena_rb.cap) HIP_TRY(c, hipDeviceSynchronize());
        const int rc = commit(c, c->arena_rb, P);
        if (rc) return rc;
    }
    hipStream_t st = c->stream;
    HIP_TRY(c, hipMemcpyAsync(c->rb_major.p, in->seg_aux_major, in->n_seg, hipMemcpyHostToDevice, st));
    if (in->seg_nodedup) HIP_TRY(c, hipMemcpyAsync(c->rb_nodedup.p, in->seg_nodedup, in->n_seg, hipMemcpyHostToDevice, st));
    if (!from_pool) {
        HIP_TRY(c, hipMemcpyAsync(c->rb_seg.p, in->seg_id, n * 4, hipMemcpyHostToDevice, st));
        HIP_TRY(c, hipMemcpyAsync(c->rb_a.p, in->a, n * 8, hipMemcpyHostToDevice, st));
        HIP_TRY(c, hipMemcpyAsync(c->rb_b.p, in->b, n * 8, hipMemcpyHostToDevice, st));
        HIP_TRY(c, hipMemcpyAsync(c->rb_rid.p, in->read_id, n * 4, hipMemcpyHostToDevice, st));
        HIP_TRY(c, hipMemcpyAsync(c->rb_aux.p, in->aux, n * 4, hipMemcpyHostToDevice, st));
    }
    {
        // the rows -> the input columns (pool rows: read index -> name rank; host rows: in place), the key widths from the device
        if (from_pool) HIP_TRY(c, hipMemcpyAsync(c->rb_rank.p, in->read_rank, (size_t)in->n_rank * 4, hipMemcpyHostToDevice, st));
        HIP_TRY(c, hipMemsetAsync(c->rb_mx.p, 0, 64, st));
        if (from_pool)
            hipLaunchKernelGGL(k_pool_to_rows, dim3(div_up(n, 2048)), dim3(256), 0, st, dp<int>(c->pool_seg), dp<i64>(c->pool_a), dp<i64>(c->pool_b),
                               dp<int>(c->pool_read), dp<int>(c->pool_aux), n, dp<int>(c->rb_rank), (i64)in->n_rank, in->n_seg, dp<uint8_t>(c->rb_major),
                               dp<int>(c->rb_seg), dp<i64>(c->rb_a), dp<i64>(c->rb_b), dp<int>(c->rb_rid), dp<int>(c->rb_aux), dp<unsigned long long>(c->rb_mx));
        else
            hipLaunchKernelGGL(k_pool_to_rows, dim3(div_up(n, 2048)), dim3(256), 0, st, dp<int>(c->rb_seg), dp<i64>(c->rb_a), dp<i64>(c->rb_b),
                               dp<int>(c->rb_rid), dp<int>(c->rb_aux), n, (const int*)nullptr, (i64)0, in->n_seg, dp<uint8_t>(c->rb_major),
                               dp<int>(c->rb_seg), dp<i64>(c->rb_a), dp<i64>(c->rb_b), dp<int>(c->rb_rid), dp<int>(c->rb_aux), dp<unsigned long long>(c->rb_mx));
        unsigned long long mx[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        HIP_TRY(c, hipMemcpyAsync(mx, c->rb_mx.p, 56, hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipStreamSynchronize(st));
        if (mx[5]) return fail(c, CSV_E_INVALID, from_pool ? "a pool row has a negative key, a segment out of range or a read without a rank"
                                                          : "a row has a negative key or a segment out of range");
        mx_a = (i64)mx[0]; mx_b = (i64)mx[1]; mx_rid = (int)mx[2]; mx_aux = (int)mx[3]; mx_seg = (int)mx[4]; mx_aux_all = (int)mx[6];
    }
    HIP_TRY(c, hipEventRecord(c->ev[0], st));
    // the composite-key sort (sort.hip.h) whenever the key fits 128 bits; CSV_RB_PERM_SORT=1 forces the permutation sort
    KeyLayout KL{};
    KL.ib = nbits((u64)(n - 1)) > 0 ? nbits((u64)(n - 1)) : 1;
    KL.rb = nbits((u64)mx_rid) > 0 ? nbits((u64)mx_rid) : 1; KL.bb = nbits((u64)mx_b) > 0 ? nbits((u64)mx_b) : 1;
    KL.ab = nbits((u64)mx_a) > 0 ? nbits((u64)mx_a) : 1; KL.xb = nbits((u64)mx_aux); KL.sb = nbits((u64)mx_seg) > 0 ? nbits((u64)mx_seg) : 1;
    KL.pb = nbits((u64)mx_aux_all);
    KL.T = KL.rb + KL.bb + KL.ab + KL.xb + KL.sb;
    // (the compact element holds aux | key | index in 128 bits; with no aux bits and ib + T == 128 its shifts would be by 128: wide then)
    KL.wide = (KL.ib + KL.T + KL.pb <= 128 && KL.ib + KL.T < 128) ? 0 : 1;
    const bool composite = KL.T <= 128 && !getenv("CSV_RB_PERM_SORT");
    RebuildArgs R{};
    R.n = n;
    R.seg = dp<int>(c->rb_seg); R.a = dp<i64>(c->rb_a); R.b = dp<i64>(c->rb_b); R.rid = dp<int>(c->rb_rid); R.aux = dp<int>(c->rb_aux);
    R.auxk = dp<int>(c->rb_auxk); R.keep = nullptr; R.partial = dp<int>(c->rb_partial);
    R.nodedup = in->seg_nodedup ? dp<uint8_t>(c->rb_nodedup) : nullptr;
    R.o_seg = dp<int>(c->rb_oseg); R.o_a = dp<i64>(c->rb_oa); R.o_b = dp<i64>(c->rb_ob); R.o_rid = dp<int>(c->rb_orid);
    R.o_aux = dp<int>(c->rb_oaux); R.o_src = dp<int>(c->rb_osrc); R.n_out = (int*)((char*)c->cnt.p + 768);      // (a word of its own: the run arenas at +0 / +256 may have a publish in flight)
    R.drop = nullptr;
    out->n_tie_rows = 0; out->n_tie_dropped = 0;
    bool ties_settled = false;
    int npass = 0;
    // the tie groups' round trip to the caller (csv_tie_order_fn): `lst` = {position | continues << 31, source row}, any order
    std::vector<int> tie_pos, tie_src; std::vector<uint8_t> tie_flag;
    auto ask_caller = [&](std::vector<int2>& lst) -> int {
        const int n_list = (int)lst.size();
        std::sort(lst.begin(), lst.end(), [](const int2& x, const int2& y) { return (x.x & 0x7fffffff) < (y.x & 0x7fffffff); });
        std::vector<int64_t> goff;
        std::vector<int> src((size_t)n_list), order((size_t)n_list, -1);
        std::vector<uint8_t> drop((size_t)n_list, 0);
        tie_pos.assign((size_t)n_list, 0); tie_src.assign((size_t)n_list, 0); tie_flag.assign((size_t)n_list, 0);
        for (int k = 0; k < n_list; k++) {
            if (!(lst[k].x & (int)0x80000000)) goff.push_back(k);               // a group's head
            tie_pos[k] = lst[k].x & 0x7fffffff; src[k] = lst[k].y;
        }
        goff.push_back(n_list);
        const int rc = in->tie_order(in->tie_user, (int64_t)goff.size() - 1, goff.data(), src.data(), order.data(), drop.data());
        if (rc != 0) return fail(c, CSV_E_INVALID, "tie_order returned %d", rc);
        std::vector<uint8_t> seen((size_t)n_list, 0);
        int64_t dropped = 0;
        for (size_t g = 0; g + 1 < goff.size(); g++) {                            // order[] must be a permutation inside every group
            const int64_t g0 = goff[g], g1 = goff[g + 1];
            for (int64_t k = g0; k < g1; k++) {
                const int64_t o = order[k];
                if (o < 0 || o >= g1 - g0 || seen[g0 + o]) return fail(c, CSV_E_INVALID, "tie_order: order[] is not a permutation inside group %zu", g);
                seen[g0 + o] = 1;
                tie_src[g0 + o] = src[k]; tie_flag[g0 + o] = drop[k] ? 1 : 0;
                dropped += drop[k] ? 1 : 0;
            }
        }
        out->n_tie_rows = n_list; out->n_tie_dropped = dropped;
        return CSV_OK;
    };
    if (composite) {
        RsCols C{dp<int>(c->rb_seg), dp<i64>(c->rb_a), dp<i64>(c->rb_b), dp<int>(c->rb_rid), dp<int>(c->rb_aux), dp<uint8_t>(c->rb_major)};
        void* e_in = c->rb_el0.p; void* e_out = c->rb_el1.p;
        auto run_sort = [&](auto wide_tag) -> int {
            constexpr bool W = decltype(wide_tag)::value;
            typedef RsElem<W> E;
            hipLaunchKernelGGL(k_rs_pack<W>, dim3(div_up(n, 256)), dim3(256), 0, st, C, n, KL, (E*)e_in);
            for (int shift = 0; shift < KL.T; shift += RS_BITS) {
                const int dbits = KL.T - shift < RS_BITS ? KL.T - shift : RS_BITS;
                hipLaunchKernelGGL(k_rs_hist<W>, dim3(nblk), dim3(256), 0, st, (const E*)e_in, n, nunits, KL, shift, dbits, dp<int>(c->rb_hist));
                hipLaunchKernelGGL(k_sort_rowsum, dim3(RS_RADIX), dim3(256), 0, st, dp<int>(c->rb_hist), nunits, dp<int>(c->rb_tot));
                hipLaunchKernelGGL(k_sort_rowscan, dim3(RS_RADIX), dim3(256), 0, st, dp<int>(c->rb_hist), nunits, dp<int>(c->rb_tot));
                hipLaunchKernelGGL(k_rs_scatter<W>, dim3(nblk), dim3(256), 0, st, (const E*)e_in, (E*)e_out, n, nunits, KL, shift, dbits, dp<int>(c->rb_hist));
                std::swap(e_in, e_out);
                npass++;
            }
            RsTail T{};
            T.n = n; T.L = KL; T.nodedup = R.nodedup; T.drop = nullptr; T.partial = R.partial;
            T.o_seg = R.o_seg; T.o_a = R.o_a; T.o_b = R.o_b; T.o_rid = R.o_rid; T.o_aux = R.o_aux; T.o_src = R.o_src; T.n_out = R.n_out;
            if (in->tie_order && R.nodedup) {
                int* d_n = (int*)((char*)c->cnt.p + 768);
                HIP_TRY(c, hipMemsetAsync(d_n, 0, 4, st));
                HIP_TRY(c, hipMemsetAsync(c->rb_drop.p, 0, (size_t)n, st));
                hipLaunchKernelGGL(k_rs_ties<W>, dim3(div_up(n, 256)), dim3(256), 0, st, T, (const E*)e_in, (int2*)c->rb_oa.p, d_n);
                int n_list = 0;
                HIP_TRY(c, hipMemcpyAsync(&n_list, d_n, 4, hipMemcpyDeviceToHost, st));
                HIP_TRY(c, hipStreamSynchronize(st));
                if (n_list > 0) {
                    std::vector<int2> lst((size_t)n_list);
                    HIP_TRY(c, hipMemcpy(lst.data(), c->rb_oa.p, (size_t)n_list * 8, hipMemcpyDeviceToHost));
                    const int rc = ask_caller(lst);
                    if (rc) return rc;
                    HIP_TRY(c, hipMemcpyAsync(c->rb_oseg.p, tie_pos.data(), (size_t)n_list * 4, hipMemcpyHostToDevice, st));
                    HIP_TRY(c, hipMemcpyAsync(c->rb_orid.p, tie_src.data(), (size_t)n_list * 4, hipMemcpyHostToDevice, st));
                    HIP_TRY(c, hipMemcpyAsync(c->rb_ob.p, tie_flag.data(), (size_t)n_list, hipMemcpyHostToDevice, st));
                    hipLaunchKernelGGL(k_rs_tie_apply<W>, dim3(div_up(n_list, 256)), dim3(256), 0, st, n_list, dp<int>(c->rb_oseg), dp<int>(c->rb_orid),
                                       dp<uint8_t>(c->rb_ob), dp<int>(c->rb_aux), KL, (E*)e_in, dp<uint8_t>(c->rb_drop));
                    HIP_TRY(c, hipStreamSynchronize(st));                                   // (the host vectors are the copies' sources)
                }
                T.drop = dp<uint8_t>(c->rb_drop);
                ties_settled = true;
            }
            hipLaunchKernelGGL(k_rs_count<W>, dim3(ntile), dim3(256), 0, st, T, (const E*)e_in);
            hipLaunchKernelGGL(k_rs_apply<W>, dim3(ntile), dim3(256), 0, st, T, (const E*)e_in);
            return CSV_OK;
        };
        const int rc = KL.wide ? run_sort(std::true_type{}) : run_sort(std::false_type{});
        if (rc) return rc;
    } else {
        hipLaunchKernelGGL(k_rebuild_auxkey, dim3(div_up(n, 256)), dim3(256), 0, st, n, dp<int>(c->rb_seg), dp<int>(c->rb_aux),
                           dp<uint8_t>(c->rb_major), dp<int>(c->rb_auxk));
        // least significant key first: read_id, b, a, [aux], segment
        struct Field { const void* col; int elem64; int bytes; };
        const Field fields[5] = {{c->rb_rid.p, 0, nbytes((u64)mx_rid)}, {c->rb_b.p, 1, nbytes((u64)mx_b)}, {c->rb_a.p, 1, nbytes((u64)mx_a)},
                                 {c->rb_auxk.p, 0, nbytes((u64)mx_aux)}, {c->rb_seg.p, 0, nbytes((u64)mx_seg) > 0 ? nbytes((u64)mx_seg) : 1}};
        const int* pin = nullptr;
        int* pout = dp<int>(c->rb_perm0);
        for (const Field& f : fields)
            for (int byte = 0; byte < f.bytes; byte++) {
                SortPass SP{f.col, f.elem64, byte * 8, n, nunits, pin, pout, dp<int>(c->rb_hist)};
                hipLaunchKernelGGL(k_sort_hist, dim3(nblk), dim3(256), 0, st, SP);
                hipLaunchKernelGGL(k_sort_rowsum, dim3(256), dim3(256), 0, st, dp<int>(c->rb_hist), nunits, dp<int>(c->rb_tot));
                hipLaunchKernelGGL(k_sort_rowscan, dim3(256), dim3(256), 0, st, dp<int>(c->rb_hist), nunits, dp<int>(c->rb_tot));
                hipLaunchKernelGGL(k_sort_scatter, dim3(nblk), dim3(256), 0, st, SP);
                pin = pout;
                pout = (pout == dp<int>(c->rb_perm0)) ? dp<int>(c->rb_perm1) : dp<int>(c->rb_perm0);
                npass++;
            }
        R.perm = pin;
        if (in->tie_order && R.nodedup) {
            // INS rows that tie on their integer keys: the caller orders them (by sequence) and names the duplicates; the answer is
            // written into the permutation / a drop map on the device, and the gather below never knows.  The output buffers are
            // free until the gather: rb_oa holds the list, rb_oseg / rb_orid / rb_ob the answer on its way back.
            int* d_n = (int*)((char*)c->cnt.p + 768);
            HIP_TRY(c, hipMemsetAsync(d_n, 0, 4, st));
            HIP_TRY(c, hipMemsetAsync(c->rb_drop.p, 0, (size_t)n, st));
            hipLaunchKernelGGL(k_rebuild_ties, dim3(div_up(n, 256)), dim3(256), 0, st, R, (int2*)c->rb_oa.p, d_n);
            int n_list = 0;
            HIP_TRY(c, hipMemcpyAsync(&n_list, d_n, 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(c, hipStreamSynchronize(st));
            if (n_list > 0) {
                std::vector<int2> lst((size_t)n_list);
                HIP_TRY(c, hipMemcpy(lst.data(), c->rb_oa.p, (size_t)n_list * 8, hipMemcpyDeviceToHost));
                const int rc = ask_caller(lst);
                if (rc) return rc;
                HIP_TRY(c, hipMemcpyAsync(c->rb_oseg.p, tie_pos.data(), (size_t)n_list * 4, hipMemcpyHostToDevice, st));
                HIP_TRY(c, hipMemcpyAsync(c->rb_orid.p, tie_src.data(), (size_t)n_list * 4, hipMemcpyHostToDevice, st));
                HIP_TRY(c, hipMemcpyAsync(c->rb_ob.p, tie_flag.data(), (size_t)n_list, hipMemcpyHostToDevice, st));
                hipLaunchKernelGGL(k_rebuild_tie_apply, dim3(div_up(n_list, 256)), dim3(256), 0, st, n_list, dp<int>(c->rb_oseg), dp<int>(c->rb_orid),
                                   dp<uint8_t>(c->rb_ob), const_cast<int*>(R.perm), dp<uint8_t>(c->rb_drop));
                HIP_TRY(c, hipStreamSynchronize(st));                                       // (the host vectors are the copies' sources)
            }
            R.drop = dp<uint8_t>(c->rb_drop);
            ties_settled = true;
        }
        hipLaunchKernelGGL(k_rebuild_count, dim3(ntile), dim3(256), 0, st, R);
        hipLaunchKernelGGL(k_rebuild_apply, dim3(ntile), dim3(256), 0, st, R);
    }
    // rows per segment and the INS tie count ([n_seg] = ties), from the sorted output
    HIP_TRY(c, hipMemsetAsync(dp<i64>(c->rb_segcnt) + in->n_seg, 0, 8, st));
    // (the tie count is a grid-stride loop over the sorted rows: enough workgroups for ~4 rows per thread)
    const int g_sc = std::max(div_up(in->n_seg, 256), std::min(div_up(n, 1024), 8192));
    if (ties_settled) R.nodedup = nullptr;                  // (nothing left to count)
    hipLaunchKernelGGL(k_rebuild_segcount, dim3(g_sc), dim3(256), 0, st, R, in->n_seg,
                       dp<i64>(c->rb_segcnt), dp<i64>(c->rb_segcnt) + in->n_seg);
    HIP_TRY(c, hipEventRecord(c->ev[1], st));
    HIP_TRY(c, hipGetLastError());
    int n_out = 0;
    std::vector<i64> segcnt((size_t)in->n_seg + 1);
    HIP_TRY(c, hipMemcpyAsync(&n_out, (char*)c->cnt.p + 768, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipMemcpyAsync(segcnt.data(), c->rb_segcnt.p, ((size_t)in->n_seg + 1) * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    HIP_TRY(c, hipEventElapsedTime(&out->ms_device, c->ev[0], c->ev[1]));
    out->n_out = n_out; out->n_passes = npass;
    out->n_ins_ties = ties_settled ? 0 : segcnt[(size_t)in->n_seg];
    if (out->seg_count) memcpy(out->seg_count, segcnt.data(), (size_t)in->n_seg * 8);
    const bool keep_dev = (in->flags & CSV_RB_KEEP_ON_DEVICE) != 0;
    out->dev_seg_id = out->dev_a = out->dev_b = out->dev_read_id = out->dev_aux = out->dev_src_row = nullptr;
    if (keep_dev) {
        out->dev_seg_id = c->rb_oseg.p; out->dev_a = c->rb_oa.p; out->dev_b = c->rb_ob.p; out->dev_read_id = c->rb_orid.p;
        out->dev_aux = c->rb_oaux.p; out->dev_src_row = c->rb_osrc.p;
    }
    // (with CSV_RB_KEEP_ON_DEVICE a NULL host array is simply not filled; without the flag all six are required, as before)
#define RB_D2H(dst, buf, bytes) do { if ((dst) || !keep_dev) HIP_TRY(c, hipMemcpyAsync((dst), c->buf.p, (bytes), hipMemcpyDeviceToHost, st)); } while (0)
    RB_D2H(out->seg_id, rb_oseg, (size_t)n_out * 4); RB_D2H(out->a, rb_oa, (size_t)n_out * 8); RB_D2H(out->b, rb_ob, (size_t)n_out * 8);
    RB_D2H(out->read_id, rb_orid, (size_t)n_out * 4); RB_D2H(out->aux, rb_oaux, (size_t)n_out * 4); RB_D2H(out->src_row, rb_osrc, (size_t)n_out * 4);
#undef RB_D2H
    HIP_TRY(c, hipStreamSynchronize(st));
    c->uploaded = c->ran = false;          // cnt was used as scratch
    return CSV_OK;
}

int csv_cigar_signatures(csv_ctx* c, const csv_cigar_in* in, csv_cigar_out* out)
{
    if (!c || !in || !out) return CSV_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    out->n_sig_ins = out->n_piece_ins = out->n_sig_del = 0; out->ms_device = 0;
    const i64 n = in->n_reads;
    if (n < 0 || (n > 0 && (!in->cig_off || !in->ref_start))) return fail(c, CSV_E_INVALID, "bad CIGAR batch header");
    if (n == 0) return CSV_OK;
    const i64 nops = in->cig_off[n] - in->cig_off[0];
    if (in->cig_off[0] != 0 || nops < 0 || (nops > 0 && !in->cigar)) return fail(c, CSV_E_INVALID, "cig_off must start at 0 and not decrease");
    if (nops >= (1ll << 31) - 4096) return fail(c, CSV_E_INVALID, "CIGAR batch too large (%lld operations): split it", (long long)nops);
    for (i64 r = 0; r < n; r++)                                 // the kernels index `cigar` with these: every offset is checked here
        if (in->cig_off[r] < 0 || in->cig_off[r + 1] < in->cig_off[r] || in->cig_off[r + 1] > nops)
            return fail(c, CSV_E_INVALID, "cig_off decreases or leaves the CIGAR array at read %lld", (long long)r);
    const int ntile = div_up(n, CG_TILE);
    // every op can be a piece and a signature of its own: size the outputs for the worst case the caller allows, but never
    // more than the operations there are
    const i64 cap_i = out->cap_sig_ins < nops ? out->cap_sig_ins : nops, cap_p = out->cap_piece_ins < nops ? out->cap_piece_ins : nops,
              cap_d = out->cap_sig_del < nops ? out->cap_sig_del : nops;
    Plan P;
#define PL(buf, bytes) P.add(c->buf, (size_t)(bytes))
    const bool to_pool = (in->flags & CSV_CG_TO_POOL) != 0;
    PL(cg_off, (n + 1) * 8); PL(cg_ops, (nops + 1) * 4); PL(cg_start, n * 8); PL(cg_use, n); PL(cg_cnt, n * 16);
    if (to_pool && in->query_len) PL(cg_qlen, n * 4);
    PL(cg_tiles, (size_t)ntile * 24); PL(cg_tot, 32);
    PL(cg_iread, (cap_i + 1) * 4); PL(cg_ipos, (cap_i + 1) * 8); PL(cg_ilen, (cap_i + 1) * 8); PL(cg_ip0, (cap_i + 1) * 8); PL(cg_inp, (cap_i + 1) * 4);
    PL(cg_pq, (cap_p + 1) * 4); PL(cg_pl, (cap_p + 1) * 4);
    PL(cg_dread, (cap_d + 1) * 4); PL(cg_dpos, (cap_d + 1) * 8); PL(cg_dlen, (cap_d + 1) * 8);
#undef PL
    {
        if (P.total > c->arena_rb.cap) HIP_TRY(c, hipDeviceSynchronize());
        const int rc = commit(c, c->arena_rb, P);
        if (rc) return rc;
    }
    hipStream_t st = c->stream;
    HIP_TRY(c, hipMemcpyAsync(c->cg_off.p, in->cig_off, (size_t)(n + 1) * 8, hipMemcpyHostToDevice, st));
    if (nops) HIP_TRY(c, hipMemcpyAsync(c->cg_ops.p, in->cigar, (size_t)nops * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(c->cg_start.p, in->ref_start, (size_t)n * 8, hipMemcpyHostToDevice, st));
    if (in->use) HIP_TRY(c, hipMemcpyAsync(c->cg_use.p, in->use, (size_t)n, hipMemcpyHostToDevice, st));
    CigarArgs A{};
    A.n_reads = n; A.cig_off = dp<i64>(c->cg_off); A.cigar = dp<unsigned>(c->cg_ops); A.ref_start = dp<i64>(c->cg_start);
    A.use = in->use ? dp<uint8_t>(c->cg_use) : nullptr;
    A.min_siglength = in->min_siglength; A.merge_ins = in->merge_ins_threshold; A.merge_del = in->merge_del_threshold;
    A.cnt = dp<int4>(c->cg_cnt); A.tile_sum = dp<i64>(c->cg_tiles); A.totals = dp<i64>(c->cg_tot);
    A.ins_read = dp<int>(c->cg_iread); A.ins_pos = dp<i64>(c->cg_ipos); A.ins_len = dp<i64>(c->cg_ilen); A.ins_piece0 = dp<i64>(c->cg_ip0);
    A.ins_npiece = dp<int>(c->cg_inp); A.piece_qoff = dp<int>(c->cg_pq); A.piece_len = dp<int>(c->cg_pl);
    A.del_read = dp<int>(c->cg_dread); A.del_pos = dp<i64>(c->cg_dpos); A.del_len = dp<i64>(c->cg_dlen);
    const int grid = div_up(n, 4) < 4096 ? div_up(n, 4) : 4096;
    HIP_TRY(c, hipEventRecord(c->ev[0], st));
    hipLaunchKernelGGL(k_cigar_count, dim3(grid), dim3(256), 0, st, A);
    hipLaunchKernelGGL(k_cigar_tiles, dim3(ntile), dim3(256), 0, st, A);
    hipLaunchKernelGGL(k_cigar_offsets, dim3(ntile), dim3(256), 0, st, A);
    HIP_TRY(c, hipEventRecord(c->ev[1], st));
    HIP_TRY(c, hipGetLastError());
    i64 tot[3] = {0, 0, 0};
    HIP_TRY(c, hipMemcpyAsync(tot, c->cg_tot.p, 24, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    out->n_sig_ins = tot[0]; out->n_piece_ins = tot[1]; out->n_sig_del = tot[2];
    float ms1 = 0, ms2 = 0;
    HIP_TRY(c, hipEventElapsedTime(&ms1, c->ev[0], c->ev[1]));
    if (tot[0] > out->cap_sig_ins || tot[1] > out->cap_piece_ins || tot[2] > out->cap_sig_del)
        return fail(c, CSV_E_CAPACITY, "need %lld INS signatures / %lld INS pieces / %lld DEL signatures", (long long)tot[0], (long long)tot[1], (long long)tot[2]);
    HIP_TRY(c, hipEventRecord(c->ev[2], st));
    hipLaunchKernelGGL(k_cigar_emit, dim3(grid), dim3(256), 0, st, A);
    HIP_TRY(c, hipEventRecord(c->ev[3], st));
    HIP_TRY(c, hipGetLastError());
    if (to_pool && tot[0] + tot[2] > 0) {
        if (in->read_base < 0 || in->read_base + n >= (1ll << 31)) return fail(c, CSV_E_INVALID, "read_base out of range");
        { const int rc = pool_reserve(c, tot[0] + tot[2]); if (rc) return rc; }
        if (in->query_len) HIP_TRY(c, hipMemcpyAsync(c->cg_qlen.p, in->query_len, (size_t)n * 4, hipMemcpyHostToDevice, st));
        PoolCols PC{dp<int>(c->pool_seg), dp<i64>(c->pool_a), dp<i64>(c->pool_b), dp<int>(c->pool_read), dp<int>(c->pool_aux)};
        hipLaunchKernelGGL(k_pool_from_cigar, dim3(div_up(tot[0] + tot[2], 256)), dim3(256), 0, st, PC, c->pool_n, A, tot[0], tot[2], in->seg_ins, in->seg_del,
                           in->read_base, in->query_len ? dp<int>(c->cg_qlen) : nullptr);
        HIP_TRY(c, hipGetLastError());
        c->pool_n += tot[0] + tot[2];
    }
    // (with CSV_CG_TO_POOL an output array that is NULL is not written)
#define D2H(dst, buf, bytes) do { if ((bytes) > 0 && ((dst) || !to_pool)) HIP_TRY(c, hipMemcpyAsync((dst), c->buf.p, (size_t)(bytes), hipMemcpyDeviceToHost, st)); } while (0)
    D2H(out->ins_read, cg_iread, tot[0] * 4); D2H(out->ins_pos, cg_ipos, tot[0] * 8); D2H(out->ins_len, cg_ilen, tot[0] * 8);
    D2H(out->ins_piece0, cg_ip0, tot[0] * 8); D2H(out->ins_npiece, cg_inp, tot[0] * 4);
    D2H(out->piece_qoff, cg_pq, tot[1] * 4); D2H(out->piece_len, cg_pl, tot[1] * 4);
    D2H(out->del_read, cg_dread, tot[2] * 4); D2H(out->del_pos, cg_dpos, tot[2] * 8); D2H(out->del_len, cg_dlen, tot[2] * 8);
#undef D2H
    HIP_TRY(c, hipStreamSynchronize(st));
    HIP_TRY(c, hipEventElapsedTime(&ms2, c->ev[2], c->ev[3]));
    out->ms_device = ms1 + ms2;
    return CSV_OK;
}

int csv_split_signatures(csv_ctx* c, const csv_split_in* in, csv_split_out* out)
{
    if (!c || !in || !out) return CSV_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    out->n = 0; out->ms_device = 0;
    const i64 n = in->n_reads;
    if (n < 0 || (n > 0 && (!in->ent_off || !in->read_len))) return fail(c, CSV_E_INVALID, "bad split-read batch header");
    if (n == 0) return CSV_OK;
    const i64 ne = in->ent_off[n] - in->ent_off[0];
    if (in->ent_off[0] != 0 || ne < 0) return fail(c, CSV_E_INVALID, "ent_off must start at 0 and not decrease");
    if (ne > 0 && (!in->c0 || !in->c1 || !in->f0 || !in->f1 || !in->chr || !in->mapq || !in->strand || !in->primary))
        return fail(c, CSV_E_INVALID, "split-read entry columns missing");
    if (ne >= (1ll << 31) - 4096 || n >= (1ll << 31) - 4096) return fail(c, CSV_E_INVALID, "split-read batch too large: split it");
    for (i64 r = 0; r < n; r++)
        if (in->ent_off[r + 1] < in->ent_off[r]) return fail(c, CSV_E_INVALID, "ent_off decreases at read %lld", (long long)r);
    const int ntile = div_up(n, CG_TILE);
    const i64 cap = out->cap < 0 ? 0 : out->cap;
    Plan P;
#define PL(buf, bytes) P.add(c->buf, (size_t)(bytes))
    PL(sp_off, (n + 1) * 8); PL(sp_len, n * 8); PL(sp_c0, (ne + 1) * 8); PL(sp_c1, (ne + 1) * 8); PL(sp_f0, (ne + 1) * 8); PL(sp_f1, (ne + 1) * 8);
    PL(sp_chr, (ne + 1) * 4); PL(sp_mapq, (ne + 1) * 4); PL(sp_strand, ne + 1); PL(sp_primary, ne + 1); PL(sp_seg, (ne + 1) * sizeof(SpSeg));
    PL(sp_cnt, n * 16); PL(sp_tiles, (size_t)ntile * 24); PL(sp_tot, 32);
    PL(sp_kind, cap + 1); PL(sp_read, (cap + 1) * 4); PL(sp_ochr, (cap + 1) * 4); PL(sp_aux, (cap + 1) * 4);
    PL(sp_a, (cap + 1) * 8); PL(sp_b, (cap + 1) * 8); PL(sp_c, (cap + 1) * 8); PL(sp_d, (cap + 1) * 8);
#undef PL
    {
        if (P.total > c->arena_rb.cap) HIP_TRY(c, hipDeviceSynchronize());
        const int rc = commit(c, c->arena_rb, P);
        if (rc) return rc;
    }
    hipStream_t st = c->stream;
#define H2D(buf, src, bytes) do { if ((bytes) > 0) HIP_TRY(c, hipMemcpyAsync(c->buf.p, (src), (size_t)(bytes), hipMemcpyHostToDevice, st)); } while (0)
    H2D(sp_off, in->ent_off, (n + 1) * 8); H2D(sp_len, in->read_len, n * 8);
    H2D(sp_c0, in->c0, ne * 8); H2D(sp_c1, in->c1, ne * 8); H2D(sp_f0, in->f0, ne * 8); H2D(sp_f1, in->f1, ne * 8);
    H2D(sp_chr, in->chr, ne * 4); H2D(sp_mapq, in->mapq, ne * 4); H2D(sp_strand, in->strand, ne); H2D(sp_primary, in->primary, ne);
#undef H2D
    SplitArgs A{};
    A.n_reads = n; A.ent_off = dp<i64>(c->sp_off); A.read_len = dp<i64>(c->sp_len);
    A.c0 = dp<i64>(c->sp_c0); A.c1 = dp<i64>(c->sp_c1); A.f0 = dp<i64>(c->sp_f0); A.f1 = dp<i64>(c->sp_f1);
    A.chr = dp<int>(c->sp_chr); A.mapq = dp<int>(c->sp_mapq); A.strand = dp<uint8_t>(c->sp_strand); A.primary = dp<uint8_t>(c->sp_primary);
    A.sv = in->sv_size; A.max_size = in->max_size; A.min_mapq = in->min_mapq; A.parts = in->max_split_parts;
    A.seg = dp<SpSeg>(c->sp_seg); A.cnt = dp<int4>(c->sp_cnt); A.cap = cap;
    A.kind = dp<uint8_t>(c->sp_kind); A.read = dp<int>(c->sp_read); A.o_chr = dp<int>(c->sp_ochr); A.aux = dp<int>(c->sp_aux);
    A.a = dp<i64>(c->sp_a); A.b = dp<i64>(c->sp_b); A.c = dp<i64>(c->sp_c); A.d = dp<i64>(c->sp_d);
    CigarArgs SC{};                                         // the per-read prefix is the CIGAR scan's (k_cigar_tiles / k_cigar_offsets)
    SC.n_reads = n; SC.cnt = A.cnt; SC.tile_sum = dp<i64>(c->sp_tiles); SC.totals = dp<i64>(c->sp_tot);
    const int grid = div_up(n, 256);
    HIP_TRY(c, hipEventRecord(c->ev[0], st));
    hipLaunchKernelGGL(k_split_count, dim3(grid), dim3(256), 0, st, A);
    hipLaunchKernelGGL(k_cigar_tiles, dim3(ntile), dim3(256), 0, st, SC);
    hipLaunchKernelGGL(k_cigar_offsets, dim3(ntile), dim3(256), 0, st, SC);
    HIP_TRY(c, hipEventRecord(c->ev[1], st));
    HIP_TRY(c, hipGetLastError());
    i64 tot[3] = {0, 0, 0};
    HIP_TRY(c, hipMemcpyAsync(tot, c->sp_tot.p, 24, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    out->n = tot[0];
    if (tot[0] > out->cap) return fail(c, CSV_E_CAPACITY, "need %lld candidates", (long long)tot[0]);
    HIP_TRY(c, hipEventRecord(c->ev[2], st));
    hipLaunchKernelGGL(k_split_emit, dim3(grid), dim3(256), 0, st, A);
    HIP_TRY(c, hipEventRecord(c->ev[3], st));
    HIP_TRY(c, hipGetLastError());
    const bool to_pool = (in->flags & CSV_CG_TO_POOL) != 0;
    if (to_pool && tot[0] > 0) {
        if (in->read_base < 0 || in->read_base + n >= (1ll << 31)) return fail(c, CSV_E_INVALID, "read_base out of range");
        { const int rc = pool_reserve(c, tot[0]); if (rc) return rc; }
        if (in->query_len) { const int rc = reserve(c, c->sp_qlen, (size_t)n * 4); if (rc) return rc; HIP_TRY(c, hipMemcpyAsync(c->sp_qlen.p, in->query_len, (size_t)n * 4, hipMemcpyHostToDevice, st)); }
        PoolCols PC{dp<int>(c->pool_seg), dp<i64>(c->pool_a), dp<i64>(c->pool_b), dp<int>(c->pool_read), dp<int>(c->pool_aux)};
        PoolSegBase SB{};
        for (int k = 0; k < 5; k++) SB.b[k] = in->pool_seg_base[k];
        hipLaunchKernelGGL(k_pool_from_split, dim3(div_up(tot[0], 256)), dim3(256), 0, st, PC, c->pool_n, A, tot[0], SB, in->read_base,
                           in->query_len ? dp<int>(c->sp_qlen) : nullptr);
        HIP_TRY(c, hipGetLastError());
        c->pool_n += tot[0];
    }
#define D2H(dst, buf, bytes) do { if ((bytes) > 0 && ((dst) || !to_pool)) HIP_TRY(c, hipMemcpyAsync((dst), c->buf.p, (size_t)(bytes), hipMemcpyDeviceToHost, st)); } while (0)
    D2H(out->kind, sp_kind, tot[0]); D2H(out->read, sp_read, tot[0] * 4); D2H(out->chr, sp_ochr, tot[0] * 4); D2H(out->aux, sp_aux, tot[0] * 4);
    D2H(out->a, sp_a, tot[0] * 8); D2H(out->b, sp_b, tot[0] * 8); D2H(out->c, sp_c, tot[0] * 8); D2H(out->d, sp_d, tot[0] * 8);
#undef D2H
    HIP_TRY(c, hipStreamSynchronize(st));
    float ms1 = 0, ms2 = 0;
    HIP_TRY(c, hipEventElapsedTime(&ms1, c->ev[0], c->ev[1]));
    HIP_TRY(c, hipEventElapsedTime(&ms2, c->ev[2], c->ev[3]));
    out->ms_device = ms1 + ms2;
    return CSV_OK;
}

int csv_cluster_batch(csv_ctx* c, const csv_batch_in* in, csv_batch_out* out)
{
    if (!c || !in || !out) return CSV_E_INVALID;
    const bool tm = getenv("CSV_DEBUG_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = tm ? now() : 0;
    int rc = upload_impl(c, in, out->cluster_id != nullptr || out->allele_id != nullptr, false, true);
    const double t1 = tm ? now() : 0;
    if (rc == CSV_OK) rc = run_impl(c, nullptr);
    const double t2 = tm ? now() : 0;
    if (tm && rc == CSV_OK) { (void)hipStreamSynchronize(c->stream); }
    const double t3 = tm ? now() : 0;
    if (rc == CSV_OK) rc = csv_batch_download(c, out);
    if (tm) fprintf(stderr, "[csv] one shot: upload issue %.3f ms, run issue %.3f ms, wait for the kernels %.3f ms, download %.3f ms, total %.3f ms\n",
                    t1 - t0, t2 - t1, t3 - t2, now() - t3, now() - t0);
    // the caller's columns may still be the source of a copy in flight when something failed on the way
    if (rc != CSV_OK && rc != CSV_E_CAPACITY) (void)hipDeviceSynchronize();
    return rc;
}

}  // extern "C"
