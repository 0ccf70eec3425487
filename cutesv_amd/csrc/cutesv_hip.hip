// cutesv_hip.hip — host side of libcutesv_hip.so: context, device arena, the C ABI of
// include/cutesv_hip.h and the launch sequence of one batch.  gfx950 only.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "kernels.hip.h"
#include "sort.hip.h"

using namespace csv;

namespace {

struct Buf {                      // grow-only device buffer
    void*  p = nullptr;
    size_t cap = 0;
};

// one timing slot per launch, in launch order
const char* kStageName[CSV_N_STAGES] = {"init", "k_chain_count", "k_chain_apply",
                                        "k_refine_indel_wave", "k_refine_wave", "k_refine_mid", "k_refine_block", "k_items_scan",
                                        "k_emit", "k_pmax_count", "k_pmax_apply", "k_genotype", "k_genotype_tra", "", "", ""};

}  // namespace

struct csv_ctx {
    int         device = 0;
    hipStream_t stream = nullptr;
    hipStream_t side[3] = {};         // side streams: [0] mid + workgroup tier, [1] DUP/INV/TRA wavefront tier, [2] reads prefix max
    hipEvent_t  ev_init = nullptr, ev_sel = nullptr, ev_aux[3] = {};
    std::string err;
    hipEvent_t  ev[CSV_N_STAGES + 2] = {};
    // device buffers
    Buf seg, woff, seg_drop, a, b, rid, aux;
    Buf cluster_id, partial, partial64, item_rec, list_small, list_big, list_tiny, tile_prev, partial_t, seg_gate;
    Buf item_nslots, item_cnt, item_base, sup_tmp;
    Buf t_bp1, t_bp2, t_search, t_pick, t_support, t_cipos, t_cilen, t_supoff, t_valid;
    Buf sc_k, sc_x, sc_v1, sc_v2, sc_v3, sc_v4, sc_v5;
    Buf o_seg, o_cluster, o_aux, o_bp1, o_bp2, o_support, o_cipos, o_cilen, o_search, o_pick, o_dr, o_dv, o_gl, o_ghdr;
    Buf o_supoff, o_supsig, o_suprid, allele_id;
    Buf reads_off, r_start, r_end, r_primary, r_id, r_pmax, pm_partial, gt_over, contig_len;
    Buf sqrt_tab, cnt;
    Buf rb_seg, rb_a, rb_b, rb_rid, rb_aux, rb_auxk, rb_major, rb_perm0, rb_perm1, rb_hist, rb_tot, rb_partial;
    Buf rb_oseg, rb_oa, rb_ob, rb_orid, rb_oaux, rb_osrc;
    // host copies
    std::vector<csv_segment> h_seg;
    std::vector<i64>         h_woff;
    std::vector<int>         h_seg_gate;
    bool     uploaded = false, ran = false, any_genotype = false, any_pair = false, any_tra_gt = false, big_lds_set = false;
    i64      n_sig_host = 0;
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

int reserve(csv_ctx* c, Buf& b, size_t bytes)
{
    if (bytes <= b.cap) return CSV_OK;
    if (b.p) { HIP_TRY(c, hipFree(b.p)); b.p = nullptr; b.cap = 0; }
    size_t want = bytes + bytes / 4 + 256;
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) { b.p = nullptr; return fail(c, CSV_E_NOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e)); }
    b.cap = want;
    return CSV_OK;
}

#define RES(buf, bytes) do { int rc_ = reserve(c, c->buf, (size_t)(bytes)); if (rc_) return rc_; } while (0)

template <class T> T* dp(const Buf& b) { return (T*)b.p; }

int div_up(i64 a, i64 b) { return (int)((a + b - 1) / b); }

}  // namespace

extern "C" {

int csv_abi_version(void) { return CSV_ABI_VERSION; }

const char* csv_stage_name(int s) { return (s >= 0 && s < CSV_N_STAGES) ? kStageName[s] : ""; }

int csv_device_count(int* n)
{
    int k = 0;
    hipError_t e = hipGetDeviceCount(&k);
    if (n) *n = (e == hipSuccess) ? k : 0;
    return e == hipSuccess ? CSV_OK : CSV_E_HIP;
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

int csv_ctx_create(int device_id, csv_ctx** out)
{
    if (!out) return CSV_E_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device_id < 0 || device_id >= n) return CSV_E_HIP;
    csv_ctx* c = new csv_ctx();
    c->device = device_id;
    if (hipSetDevice(device_id) != hipSuccess || hipStreamCreate(&c->stream) != hipSuccess) { delete c; return CSV_E_HIP; }
    for (auto& e : c->ev) if (hipEventCreate(&e) != hipSuccess) { delete c; return CSV_E_HIP; }
    for (auto& s2 : c->side) if (hipStreamCreateWithFlags(&s2, hipStreamNonBlocking) != hipSuccess) { delete c; return CSV_E_HIP; }
    if (hipEventCreateWithFlags(&c->ev_init, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_sel, hipEventDisableTiming) != hipSuccess) { delete c; return CSV_E_HIP; }
    for (auto& e : c->ev_aux) if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { delete c; return CSV_E_HIP; }
    // `num ** 0.5` of cal_CIPOS is libm pow(), not sqrt(): tabulate it with the host libm (GT:59)
    std::vector<double> tab(SQRT_TAB);
    for (int i = 0; i < SQRT_TAB; i++) tab[i] = pow((double)i, 0.5);
    if (reserve(c, c->sqrt_tab, SQRT_TAB * sizeof(double)) || reserve(c, c->cnt, sizeof(DevCounters)) ||
        hipMemcpy(c->sqrt_tab.p, tab.data(), SQRT_TAB * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) {
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
    (void)hipStreamSynchronize(c->stream);
    Buf* all[] = {&c->seg, &c->woff, &c->seg_drop, &c->a, &c->b, &c->rid, &c->aux, &c->cluster_id, &c->partial,
                  &c->partial64, &c->item_rec, &c->list_small, &c->list_big, &c->list_tiny, &c->tile_prev, &c->partial_t, &c->seg_gate, &c->item_nslots,
                  &c->item_cnt, &c->item_base, &c->sup_tmp, &c->t_bp1, &c->t_bp2, &c->t_search, &c->t_pick,
                  &c->t_support, &c->t_cipos, &c->t_cilen, &c->t_supoff, &c->t_valid, &c->sc_k, &c->sc_x, &c->sc_v1, &c->sc_v2,
                  &c->sc_v3, &c->sc_v4, &c->sc_v5, &c->o_seg, &c->o_cluster, &c->o_aux, &c->o_bp1, &c->o_bp2, &c->o_support,
                  &c->o_cipos, &c->o_cilen, &c->o_search, &c->o_pick, &c->o_dr, &c->o_dv, &c->o_gl, &c->o_supoff, &c->o_supsig,
                  &c->o_suprid, &c->allele_id, &c->reads_off, &c->r_start, &c->r_end, &c->r_primary, &c->r_id, &c->r_pmax, &c->pm_partial, &c->gt_over, &c->contig_len,
                  &c->sqrt_tab, &c->cnt, &c->rb_seg, &c->rb_a, &c->rb_b, &c->rb_rid, &c->rb_aux, &c->rb_auxk, &c->rb_major,
                  &c->rb_perm0, &c->rb_perm1, &c->rb_hist, &c->rb_tot, &c->rb_partial, &c->rb_oseg, &c->rb_oa, &c->rb_ob,
                  &c->rb_orid, &c->rb_oaux, &c->rb_osrc};
    for (Buf* b : all) if (b->p) (void)hipFree(b->p);
    for (auto& e : c->ev) if (e) (void)hipEventDestroy(e);
    for (auto& e : c->ev_aux) if (e) (void)hipEventDestroy(e);
    if (c->ev_init) (void)hipEventDestroy(c->ev_init);
    if (c->ev_sel) (void)hipEventDestroy(c->ev_sel);
    for (auto& s2 : c->side) if (s2) { (void)hipStreamSynchronize(s2); (void)hipStreamDestroy(s2); }
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

int csv_batch_upload(csv_ctx* c, const csv_batch_in* in)
{
    if (!c || !in) return CSV_E_INVALID;
    c->uploaded = c->ran = false;
    HIP_TRY(c, hipSetDevice(c->device));
    if (in->n_seg < 0 || in->n_sig < 0 || (in->n_seg > 0 && !in->seg)) return fail(c, CSV_E_INVALID, "bad batch header");
    const int S = in->n_seg;
    c->h_seg.assign(in->seg, in->seg + S);
    c->h_woff.assign(S + 1, 0);
    std::vector<uint8_t> drop(S + 1, 0);
    c->any_genotype = false;
    c->any_pair = false;
    c->any_tra_gt = false;
    i64 cap_items = 16, cap_tmp = 16;
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
    if (W >= (1ll << 31) - 4096 || cap_tmp >= (1ll << 31) - 1) return fail(c, CSV_E_INVALID, "batch too large for 32-bit work indices (%lld signatures)", (long long)W);
    if (c->any_genotype && in->reads_off && (!in->r_start || !in->r_end || !in->r_primary || !in->r_id) && in->n_reads > 0)
        return fail(c, CSV_E_INVALID, "reads columns missing");

    // ---- device memory
    RES(seg, (S + 1) * sizeof(csv_segment)); RES(woff, (S + 2) * sizeof(i64)); RES(seg_drop, S + 1);
    RES(a, (W + 1) * 8); RES(b, (W + 1) * 8); RES(rid, (W + 1) * 4); RES(aux, (W + 1) * 4);
    RES(cluster_id, (W + 1) * 4); RES(sup_tmp, (W + 1) * 4); RES(allele_id, (W + 1) * 4);
    const i64 R = (c->any_genotype && in->reads_off) ? in->n_reads : 0;
    const i64 np32 = div_up(W, CH_TILE) + 2;              // per chain tile
    const i64 np64 = np32;
    RES(partial, np32 * 4); RES(partial64, np64 * 8); RES(tile_prev, np32 * 8); RES(partial_t, np32 * 4); RES(seg_gate, (S + 1) * 16);
    RES(item_rec, cap_items * 16); RES(list_small, cap_items * 4); RES(list_big, cap_items * 4); RES(list_tiny, cap_items * 4);
    RES(item_nslots, cap_items * 4); RES(item_cnt, cap_items * 8);
    RES(item_base, (cap_items + 8) * 8);
    // temp call records are indexed by w (a cluster's slots live in its own signature range)
    RES(t_bp1, (W + 1) * 8); RES(t_bp2, (W + 1) * 8); RES(t_search, (W + 1) * 8); RES(t_pick, (W + 1) * 8);
    RES(t_support, (W + 1) * 4); RES(t_cipos, (W + 1) * 4); RES(t_cilen, (W + 1) * 4); RES(t_supoff, (W + 1) * 4); RES(t_valid, (W + 1) * 4);
    const i64 SC = 2 * W + 16 + 2 * ARR_PAD;
    RES(sc_k, SC * 8); RES(sc_x, SC * 8); RES(sc_v1, SC * 4); RES(sc_v2, SC * 4); RES(sc_v3, SC * 4); RES(sc_v4, SC * 4); RES(sc_v5, SC * 4);
    RES(o_seg, cap_tmp * 4); RES(o_cluster, cap_tmp * 4); RES(o_aux, cap_tmp * 4); RES(o_bp1, cap_tmp * 8); RES(o_bp2, cap_tmp * 8);
    RES(o_support, cap_tmp * 4); RES(o_cipos, cap_tmp * 4); RES(o_cilen, cap_tmp * 4); RES(o_search, cap_tmp * 8); RES(o_pick, cap_tmp * 8);
    RES(o_dr, cap_tmp * 4); RES(o_dv, cap_tmp * 4); RES(o_gl, cap_tmp * 4); RES(o_ghdr, cap_tmp * 16); RES(o_supoff, (cap_tmp + 1) * 8);
    RES(o_supsig, (W + 1) * 8); RES(o_suprid, (W + 1) * 4);
    const bool have_tab = c->any_genotype && in->reads_off;
    if (have_tab) { RES(reads_off, (in->n_chrom + 1) * 8); RES(contig_len, (in->n_chrom + 1) * 8); }
    if (R > 0) {
        RES(pm_partial, (div_up(R, PM_TILE) + 2) * 8); RES(gt_over, (cap_tmp + 2) * 4);
        RES(r_start, R * 8); RES(r_end, R * 8); RES(r_primary, R); RES(r_id, R * 4); RES(r_pmax, R * 8);
    }

    // ---- host -> device.  Segments whose source ranges are adjacent travel as one copy.
    hipStream_t st = c->stream;
    HIP_TRY(c, hipMemcpyAsync(c->seg.p, c->h_seg.data(), S * sizeof(csv_segment), hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(c->woff.p, c->h_woff.data(), (S + 1) * sizeof(i64), hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(c->seg_drop.p, drop.data(), S + 1, hipMemcpyHostToDevice, st));
    c->h_seg_gate.assign((size_t)(S + 1) * 4, 0);           // {read_count, dropped, svtype, -} per segment
    for (int k = 0; k < S; k++) { c->h_seg_gate[4 * k] = c->h_seg[k].read_count; c->h_seg_gate[4 * k + 1] = drop[k]; c->h_seg_gate[4 * k + 2] = c->h_seg[k].svtype; }
    HIP_TRY(c, hipMemcpyAsync(c->seg_gate.p, c->h_seg_gate.data(), (size_t)(S + 1) * 16, hipMemcpyHostToDevice, st));
    for (int k = 0; k < S;) {
        int e = k;
        while (e + 1 < S && c->h_seg[e + 1].sig_begin == c->h_seg[e].sig_end) e++;
        const i64 src = c->h_seg[k].sig_begin, n = c->h_woff[e + 1] - c->h_woff[k], dst = c->h_woff[k];
        if (n > 0) {
            HIP_TRY(c, hipMemcpyAsync(dp<i64>(c->a) + dst, in->a + src, n * 8, hipMemcpyHostToDevice, st));
            HIP_TRY(c, hipMemcpyAsync(dp<i64>(c->b) + dst, in->b + src, n * 8, hipMemcpyHostToDevice, st));
            HIP_TRY(c, hipMemcpyAsync(dp<int>(c->rid) + dst, in->read_id + src, n * 4, hipMemcpyHostToDevice, st));
            HIP_TRY(c, hipMemcpyAsync(dp<int>(c->aux) + dst, in->aux + src, n * 4, hipMemcpyHostToDevice, st));
        }
        k = e + 1;
    }
    if (have_tab) HIP_TRY(c, hipMemcpyAsync(c->reads_off.p, in->reads_off, (in->n_chrom + 1) * 8, hipMemcpyHostToDevice, st));
    if (c->any_tra_gt && in->n_chrom > 0) HIP_TRY(c, hipMemcpyAsync(c->contig_len.p, in->contig_len, in->n_chrom * 8, hipMemcpyHostToDevice, st));
    if (R > 0) {
        HIP_TRY(c, hipMemcpyAsync(c->r_start.p, in->r_start, R * 8, hipMemcpyHostToDevice, st));
        HIP_TRY(c, hipMemcpyAsync(c->r_end.p, in->r_end, R * 8, hipMemcpyHostToDevice, st));
        HIP_TRY(c, hipMemcpyAsync(c->r_primary.p, in->r_primary, R, hipMemcpyHostToDevice, st));
        HIP_TRY(c, hipMemcpyAsync(c->r_id.p, in->r_id, R * 4, hipMemcpyHostToDevice, st));
    }
    HIP_TRY(c, hipStreamSynchronize(st));

    DevBatch& B = c->B;
    memset(&B, 0, sizeof B);
    B.n_seg = S; B.n_chrom = in->n_chrom; B.W = W;
    B.seg = dp<csv_segment>(c->seg); B.woff = dp<i64>(c->woff); B.seg_drop = dp<uint8_t>(c->seg_drop);
    B.a = dp<i64>(c->a); B.b = dp<i64>(c->b); B.rid = dp<int>(c->rid); B.aux = dp<int>(c->aux);
    B.cluster_id = dp<int>(c->cluster_id); B.partial = dp<int>(c->partial); B.partial64 = dp<i64>(c->partial64);
    B.item_rec = dp<int4>(c->item_rec); B.list_small = dp<int>(c->list_small); B.list_big = dp<int>(c->list_big); B.list_tiny = dp<int>(c->list_tiny); B.tile_prev = dp<int2>(c->tile_prev); B.partial_t = dp<int>(c->partial_t); B.seg_gate = dp<int4>(c->seg_gate);
    B.tiny_max = getenv("CSV_NO_TINY") ? 0 : 16;               // (timing aid: 0 sends every DEL/INS cluster of m <= 32 through the paired path)
    B.item_nslots = dp<int>(c->item_nslots); B.item_cnt = dp<i64>(c->item_cnt); B.item_base = dp<i64>(c->item_base);
    B.sup_tmp = dp<int>(c->sup_tmp);
    B.t_bp1 = dp<i64>(c->t_bp1); B.t_bp2 = dp<i64>(c->t_bp2); B.t_search = dp<i64>(c->t_search); B.t_pick = dp<i64>(c->t_pick);
    B.t_support = dp<int>(c->t_support); B.t_cipos = dp<int>(c->t_cipos); B.t_cilen = dp<int>(c->t_cilen); B.t_supoff = dp<int>(c->t_supoff); B.t_valid = dp<int>(c->t_valid);
    B.cap_tmp = (int)cap_tmp; B.cap_items = (int)cap_items;
    B.sc_k = dp<u64>(c->sc_k); B.sc_x = dp<i64>(c->sc_x); B.sc_v1 = dp<int>(c->sc_v1); B.sc_v2 = dp<int>(c->sc_v2); B.sc_v3 = dp<int>(c->sc_v3); B.sc_v4 = dp<int>(c->sc_v4); B.sc_v5 = dp<int>(c->sc_v5);
    B.o_seg = dp<int>(c->o_seg); B.o_cluster = dp<int>(c->o_cluster); B.o_aux = dp<int>(c->o_aux);
    B.o_bp1 = dp<i64>(c->o_bp1); B.o_bp2 = dp<i64>(c->o_bp2); B.o_support = dp<int>(c->o_support); B.o_cipos = dp<int>(c->o_cipos); B.o_cilen = dp<int>(c->o_cilen);
    B.o_search = dp<i64>(c->o_search); B.o_pick = dp<i64>(c->o_pick); B.o_dr = dp<int>(c->o_dr); B.o_dv = dp<int>(c->o_dv); B.o_gl = dp<int>(c->o_gl); B.o_ghdr = dp<int4>(c->o_ghdr);
    B.o_supoff = dp<i64>(c->o_supoff); B.o_supsig = dp<i64>(c->o_supsig); B.o_suprid = dp<int>(c->o_suprid); B.allele_id = dp<int>(c->allele_id);
    B.reads_off = dp<i64>(c->reads_off); B.n_reads = R;
    B.r_start = dp<i64>(c->r_start); B.r_end = dp<i64>(c->r_end); B.r_primary = dp<uint8_t>(c->r_primary); B.r_id = dp<int>(c->r_id); B.r_pmax = dp<i64>(c->r_pmax); B.pm_partial = dp<i64>(c->pm_partial); B.gt_over = dp<int>(c->gt_over); B.contig_len = dp<i64>(c->contig_len);
    B.sqrt_tab = dp<double>(c->sqrt_tab); B.cnt = dp<DevCounters>(c->cnt);
    c->n_sig_host = in->n_sig;
    c->uploaded = true;
    return CSV_OK;
}

int csv_batch_run(csv_ctx* c, csv_run_stats* stats)
{
    if (!c) return CSV_E_INVALID;
    if (!c->uploaded) return fail(c, CSV_E_STATE, "csv_batch_run before csv_batch_upload");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const DevBatch& B = c->B;
    const i64 W = B.W;
    constexpr int LDS_SMALL = refine_lds_bytes<64>();
    constexpr int LDS_MID = refine_lds_bytes<256>();
    constexpr int LDS_BIG = refine_lds_bytes<2048>();
    if (!c->big_lds_set) {
        HIP_TRY(c, hipFuncSetAttribute((const void*)k_refine<256, 2048>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BIG));
        c->big_lds_set = true;
    }
    int ev = 0;
    auto mark = [&]() -> hipError_t { return stats ? hipEventRecord(c->ev[ev++], st) : hipSuccess; };
    const bool dbg = getenv("CSV_DEBUG") != nullptr;
#define DBG(name)                                                                                          \
    do {                                                                                                   \
        if (dbg) {                                                                                         \
            fprintf(stderr, "[csv] %s ...", name); fflush(stderr);                                         \
            hipError_t e_ = hipStreamSynchronize(st);                                                      \
            fprintf(stderr, " %s\n", hipGetErrorString(e_)); fflush(stderr);                               \
        }                                                                                                  \
    } while (0)
    HIP_TRY(c, mark());
    if (W == 0) HIP_TRY(c, hipMemsetAsync(c->cnt.p, 0, sizeof(DevCounters), st));      // otherwise k_chain_count zeroes them
    HIP_TRY(c, mark());                                                              // slot 0: init (empty batch only)
    // Plain runs fork the independent kernels onto side streams (joined again before k_items_scan /
    // k_genotype); instrumented runs (stats != NULL) and CSV_DEBUG keep everything on the main stream so
    // that every kernel is timed alone.
    // (forking costs a few event waits: only worth it when the batch has pair types or genotyping)
    const bool fork = !stats && !dbg && !getenv("CSV_NO_FORK") && (c->any_pair || (c->any_genotype && B.n_reads > 0));
    hipStream_t sB = fork ? c->side[0] : st, sC = fork ? c->side[1] : st, sD = fork ? c->side[2] : st;
#define LAUNCH_ON(strm, name, kern, grid, block, lds, ...)                             \
    do {                                                                               \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, strm, __VA_ARGS__);      \
        DBG(name);                                                                     \
        HIP_TRY(c, mark());                                                            \
    } while (0)
#define LAUNCH(name, kern, grid, block, lds, ...) LAUNCH_ON(st, name, kern, grid, block, lds, __VA_ARGS__)
    if (W > 0) {
        const int nb = div_up(W, CH_TILE);
        const bool do_gt = c->any_genotype && B.n_reads > 0;
        const int nr = do_gt ? div_up(B.n_reads, PM_TILE) : 0;
        LAUNCH("chain_count", k_chain_count, nb, 320, 0, B);
        if (fork && do_gt) {                              // reads prefix max: independent of the clustering kernels
            HIP_TRY(c, hipEventRecord(c->ev_init, st));   // (after the counters were zeroed)
            HIP_TRY(c, hipStreamWaitEvent(sD, c->ev_init, 0));
            LAUNCH_ON(sD, "pmax_count", k_pmax_count, nr, 256, 0, B);
            LAUNCH_ON(sD, "pmax_apply", k_pmax_apply, nr, 256, 0, B);
            HIP_TRY(c, hipEventRecord(c->ev_aux[2], sD));
        }
        LAUNCH("chain_apply", k_chain_apply, nb, 256, 0, B);
        int g_small = B.cap_items < 8192 ? B.cap_items : 8192;
        if (g_small < 1) g_small = 1;
        int g_iw = div_up(B.cap_items, 4) < 2048 ? div_up(B.cap_items, 4) : 2048;
        if (getenv("CSV_IW_GRID")) g_iw = atoi(getenv("CSV_IW_GRID"));       // tuning aid
        if (g_iw < 1) g_iw = 1;
        if (fork) {
            HIP_TRY(c, hipEventRecord(c->ev_sel, st));
            HIP_TRY(c, hipStreamWaitEvent(sB, c->ev_sel, 0));
            if (c->any_pair) HIP_TRY(c, hipStreamWaitEvent(sC, c->ev_sel, 0));
        }
        LAUNCH("refine_indel_wave", k_refine_indel_wave, g_iw, 256, 0, B);
        if (c->any_pair) LAUNCH_ON(sC, "refine_wave", (k_refine<64, 64>), g_small, 64, LDS_SMALL, B, 0, 0, 64);
        else HIP_TRY(c, mark());
        int g_mid = B.cap_items < 8192 ? B.cap_items : 8192;
        if (g_mid < 1) g_mid = 1;
        LAUNCH_ON(sB, "refine_mid", (k_refine<64, 256>), g_mid, 64, LDS_MID, B, 1, 64, 256);
        int g_big = B.cap_items < 512 ? B.cap_items : 512;
        if (g_big < 1) g_big = 1;
        LAUNCH_ON(sB, "refine_block", (k_refine<256, 2048>), g_big, 256, LDS_BIG, B, 1, 256, 0x7fffffff);
        if (fork) {                                       // join
            HIP_TRY(c, hipEventRecord(c->ev_aux[0], sB));
            HIP_TRY(c, hipStreamWaitEvent(st, c->ev_aux[0], 0));
            if (c->any_pair) { HIP_TRY(c, hipEventRecord(c->ev_aux[1], sC)); HIP_TRY(c, hipStreamWaitEvent(st, c->ev_aux[1], 0)); }
        }
        LAUNCH("items_scan", k_items_scan, 1, 64 * IS_NW, 0, B);
        LAUNCH("emit", k_emit, 2048, 256, 0, B);
        if (do_gt) {
            if (fork) HIP_TRY(c, hipStreamWaitEvent(st, c->ev_aux[2], 0));
            else {
                LAUNCH("pmax_count", k_pmax_count, nr, 256, 0, B);
                LAUNCH("pmax_apply", k_pmax_apply, nr, 256, 0, B);
            }
            hipLaunchKernelGGL((k_genotype<1024, 4>), dim3(2048), dim3(256), 0, st, B, 0);
            hipLaunchKernelGGL((k_genotype<8192, 1>), dim3(256), dim3(64), 0, st, B, 1);      // overflow list of the first pass
            DBG("genotype");
            HIP_TRY(c, mark());
        } else if (stats) { HIP_TRY(c, mark()); HIP_TRY(c, mark()); HIP_TRY(c, mark()); }
        if (c->any_tra_gt) {
            LAUNCH("genotype_tra", k_genotype_tra, 256, 64, 0, B);
        }
    }
#undef LAUNCH
#undef LAUNCH_ON
    HIP_TRY(c, hipGetLastError());
    c->ran = true;
    if (stats) {
        memset(stats, 0, sizeof *stats);
        HIP_TRY(c, hipMemcpyAsync(&c->h_cnt, c->cnt.p, sizeof(DevCounters), hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipStreamSynchronize(st));
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

int csv_batch_validate(csv_ctx* c)
{
    if (!c) return CSV_E_INVALID;
    if (!c->uploaded) return fail(c, CSV_E_STATE, "csv_batch_validate before csv_batch_upload");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    DevCounters zero;
    memset(&zero, 0, sizeof zero);
    HIP_TRY(c, hipMemcpyAsync(c->cnt.p, &zero, sizeof zero, hipMemcpyHostToDevice, st));
    if (c->B.W > 0) hipLaunchKernelGGL(k_validate_order, dim3(div_up(c->B.W, 256)), dim3(256), 0, st, c->B);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(&c->h_cnt, c->cnt.p, sizeof(DevCounters), hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    c->ran = false;
    if (c->h_cnt.error & ERR_SIG_ORDER)
        return fail(c, CSV_E_UNSORTED, "a segment is not in the rebuild order of cuteSV (main script :764-802) or holds adjacent duplicates");
    return CSV_OK;
}

int csv_batch_download(csv_ctx* c, csv_batch_out* out)
{
    if (!c || !out) return CSV_E_INVALID;
    if (!c->ran) return fail(c, CSV_E_STATE, "csv_batch_download before csv_batch_run");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    HIP_TRY(c, hipMemcpyAsync(&c->h_cnt, c->cnt.p, sizeof(DevCounters), hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    const DevCounters& k = c->h_cnt;
    out->n_calls = k.n_calls; out->n_support = k.n_support; out->n_clusters = k.n_clusters;
    if (k.error & ERR_READS_UNSORTED) return fail(c, CSV_E_UNSORTED, "a reads block is not sorted by start");
    if (k.error & ERR_CLUSTER_TOO_BIG) return fail(c, CSV_E_INVALID, "a chained cluster has more than %lld signatures", (long long)MAX_CLUSTER);
    if (k.error & ERR_KEY_RANGE) return fail(c, CSV_E_INVALID, "a length / pos2 value is negative or >= 2^42");
    if (k.error & ERR_COVER_OVERFLOW) return fail(c, CSV_E_INVALID, "support + cover set of a call exceeds ~6000 reads");
    if (k.error & ERR_TRA_CHROM) return fail(c, CSV_E_INVALID, "a TRA call names a mate chromosome outside the reads table");
    if (k.error & ERR_TMP_OVERFLOW) return fail(c, CSV_E_INVALID, "internal: temp call capacity exceeded");
    if (k.n_calls > out->cap_calls || k.n_support > out->cap_support)
        return fail(c, CSV_E_CAPACITY, "need %d calls / %lld supports", k.n_calls, (long long)k.n_support);
    const size_t nc = k.n_calls, ns = (size_t)k.n_support;
    const DevBatch& B = c->B;
#define D2H(dst, src, bytes) do { if ((bytes) > 0) HIP_TRY(c, hipMemcpyAsync((dst), (src), (bytes), hipMemcpyDeviceToHost, st)); } while (0)
    D2H(out->call_seg, B.o_seg, nc * 4); D2H(out->call_cluster, B.o_cluster, nc * 4); D2H(out->call_aux, B.o_aux, nc * 4);
    D2H(out->bp1, B.o_bp1, nc * 8); D2H(out->bp2, B.o_bp2, nc * 8); D2H(out->support, B.o_support, nc * 4);
    D2H(out->cipos, B.o_cipos, nc * 4); D2H(out->cilen, B.o_cilen, nc * 4); D2H(out->search_pos, B.o_search, nc * 8);
    D2H(out->seq_pick, B.o_pick, nc * 8); D2H(out->dr, B.o_dr, nc * 4); D2H(out->dv, B.o_dv, nc * 4); D2H(out->gl_idx, B.o_gl, nc * 4);
    D2H(out->support_off, B.o_supoff, (nc + 1) * 8); D2H(out->support_sig, B.o_supsig, ns * 8);
    if (out->cluster_id) memset(out->cluster_id, 0xff, (size_t)c->n_sig_host * 4);
    if (out->allele_id) memset(out->allele_id, 0xff, (size_t)c->n_sig_host * 4);
    if (out->cluster_id || out->allele_id) {
        const int S = (int)c->h_seg.size();
        for (int s = 0; s < S;) {
            int e = s;
            while (e + 1 < S && c->h_seg[e + 1].sig_begin == c->h_seg[e].sig_end) e++;
            const i64 dst = c->h_seg[s].sig_begin, n = c->h_woff[e + 1] - c->h_woff[s], src = c->h_woff[s];
            if (out->cluster_id) D2H(out->cluster_id + dst, B.cluster_id + src, n * 4);
            if (out->allele_id) D2H(out->allele_id + dst, B.allele_id + src, n * 4);
            s = e + 1;
        }
    }
#undef D2H
    HIP_TRY(c, hipStreamSynchronize(st));
    if (nc == 0 && out->cap_calls >= 0 && out->support_off) out->support_off[0] = 0;
    return CSV_OK;
}

int csv_rebuild_signatures(csv_ctx* c, const csv_rebuild_in* in, csv_rebuild_out* out)
{
    if (!c || !in || !out) return CSV_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    const i64 n = in->n;
    out->n_out = 0; out->ms_device = 0; out->n_passes = 0;
    if (n < 0 || n >= (1ll << 31) - 4096 || in->n_seg <= 0) return fail(c, CSV_E_INVALID, "bad rebuild input");
    if (n == 0) return CSV_OK;
    // key widths (bytes that are non-zero somewhere) from one host pass over the columns
    i64 mx_a = 0, mx_b = 0; int mx_rid = 0, mx_aux = 0, mx_seg = 0;
    for (i64 i = 0; i < n; i++) {
        const int sg = in->seg_id[i];
        if (sg < 0 || sg >= in->n_seg || in->a[i] < 0 || in->b[i] < 0 || in->read_id[i] < 0 || in->aux[i] < 0)
            return fail(c, CSV_E_INVALID, "row %lld: negative key or segment out of range", (long long)i);
        if (in->a[i] > mx_a) mx_a = in->a[i];
        if (in->b[i] > mx_b) mx_b = in->b[i];
        if (in->read_id[i] > mx_rid) mx_rid = in->read_id[i];
        if (in->seg_aux_major[sg] && in->aux[i] > mx_aux) mx_aux = in->aux[i];
        if (sg > mx_seg) mx_seg = sg;
    }
    auto nbytes = [](u64 v) { int k = 0; while (v) { k++; v >>= 8; } return k; };
    const int nunits = div_up(n, SORT_WTILE), nblk = div_up(nunits, 4), ntile = div_up(n, 2048);
    RES(rb_seg, n * 4); RES(rb_a, n * 8); RES(rb_b, n * 8); RES(rb_rid, n * 4); RES(rb_aux, n * 4); RES(rb_auxk, n * 4);
    RES(rb_major, in->n_seg); RES(rb_perm0, n * 4); RES(rb_perm1, n * 4); RES(rb_hist, (size_t)256 * nunits * 4);
    RES(rb_tot, 256 * 4); RES(rb_partial, (ntile + 2) * 4);
    RES(rb_oseg, n * 4); RES(rb_oa, n * 8); RES(rb_ob, n * 8); RES(rb_orid, n * 4); RES(rb_oaux, n * 4); RES(rb_osrc, n * 4);
    hipStream_t st = c->stream;
    HIP_TRY(c, hipMemcpyAsync(c->rb_seg.p, in->seg_id, n * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(c->rb_a.p, in->a, n * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(c->rb_b.p, in->b, n * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(c->rb_rid.p, in->read_id, n * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(c->rb_aux.p, in->aux, n * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(c->rb_major.p, in->seg_aux_major, in->n_seg, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipEventRecord(c->ev[0], st));
    hipLaunchKernelGGL(k_rebuild_auxkey, dim3(div_up(n, 256)), dim3(256), 0, st, n, dp<int>(c->rb_seg), dp<int>(c->rb_aux),
                       dp<uint8_t>(c->rb_major), dp<int>(c->rb_auxk));
    // least significant key first: read_id, b, a, [aux], segment
    struct Field { const void* col; int elem64; int bytes; };
    const Field fields[5] = {{c->rb_rid.p, 0, nbytes((u64)mx_rid)}, {c->rb_b.p, 1, nbytes((u64)mx_b)}, {c->rb_a.p, 1, nbytes((u64)mx_a)},
                             {c->rb_auxk.p, 0, nbytes((u64)mx_aux)}, {c->rb_seg.p, 0, nbytes((u64)mx_seg) > 0 ? nbytes((u64)mx_seg) : 1}};
    const int* pin = nullptr;
    int* pout = dp<int>(c->rb_perm0);
    int npass = 0;
    for (const Field& f : fields)
        for (int byte = 0; byte < f.bytes; byte++) {
            SortPass P{f.col, f.elem64, byte * 8, n, nunits, pin, pout, dp<int>(c->rb_hist)};
            hipLaunchKernelGGL(k_sort_hist, dim3(nblk), dim3(256), 0, st, P);
            hipLaunchKernelGGL(k_sort_rowsum, dim3(256), dim3(256), 0, st, dp<int>(c->rb_hist), nunits, dp<int>(c->rb_tot));
            hipLaunchKernelGGL(k_sort_rowscan, dim3(256), dim3(256), 0, st, dp<int>(c->rb_hist), nunits, dp<int>(c->rb_tot));
            hipLaunchKernelGGL(k_sort_scatter, dim3(nblk), dim3(256), 0, st, P);
            pin = pout;
            pout = (pout == dp<int>(c->rb_perm0)) ? dp<int>(c->rb_perm1) : dp<int>(c->rb_perm0);
            npass++;
        }
    RebuildArgs R{};
    R.n = n; R.perm = pin;
    R.seg = dp<int>(c->rb_seg); R.a = dp<i64>(c->rb_a); R.b = dp<i64>(c->rb_b); R.rid = dp<int>(c->rb_rid); R.aux = dp<int>(c->rb_aux);
    R.auxk = dp<int>(c->rb_auxk); R.keep = nullptr; R.partial = dp<int>(c->rb_partial);
    R.o_seg = dp<int>(c->rb_oseg); R.o_a = dp<i64>(c->rb_oa); R.o_b = dp<i64>(c->rb_ob); R.o_rid = dp<int>(c->rb_orid);
    R.o_aux = dp<int>(c->rb_oaux); R.o_src = dp<int>(c->rb_osrc); R.n_out = (int*)c->cnt.p;
    hipLaunchKernelGGL(k_rebuild_count, dim3(ntile), dim3(256), 0, st, R);
    hipLaunchKernelGGL(k_rebuild_apply, dim3(ntile), dim3(256), 0, st, R);
    HIP_TRY(c, hipEventRecord(c->ev[1], st));
    HIP_TRY(c, hipGetLastError());
    int n_out = 0;
    HIP_TRY(c, hipMemcpyAsync(&n_out, c->cnt.p, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    HIP_TRY(c, hipEventElapsedTime(&out->ms_device, c->ev[0], c->ev[1]));
    out->n_out = n_out; out->n_passes = npass;
    HIP_TRY(c, hipMemcpyAsync(out->seg_id, c->rb_oseg.p, (size_t)n_out * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipMemcpyAsync(out->a, c->rb_oa.p, (size_t)n_out * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipMemcpyAsync(out->b, c->rb_ob.p, (size_t)n_out * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipMemcpyAsync(out->read_id, c->rb_orid.p, (size_t)n_out * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipMemcpyAsync(out->aux, c->rb_oaux.p, (size_t)n_out * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipMemcpyAsync(out->src_row, c->rb_osrc.p, (size_t)n_out * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    c->uploaded = c->ran = false;          // cnt was used as scratch
    return CSV_OK;
}

int csv_cluster_batch(csv_ctx* c, const csv_batch_in* in, csv_batch_out* out)
{
    int rc = csv_batch_upload(c, in);
    if (rc) return rc;
    rc = csv_batch_run(c, nullptr);
    if (rc) return rc;
    return csv_batch_download(c, out);
}

}  // extern "C"
