/*
 * cutesv_oracle.c — CPU restatement of cuteSV's clustering-and-refinement hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load liboracle.so.  The product path (cutesv_amd/) never
 * imports, links or executes anything in oracle/.
 *
 * Parity pin: the reference ships no tests for this path ("parity unpinned by the
 * reference").  This restatement is pinned instead by golden vectors made in
 * the build container by importing the reference itself (tests/golden/make_golden.py ->
 * tests/golden/ (json + npz files); checked by tests/test_oracle_golden.py).
 *
 * It follows the reference function by function (citations into /root/reference/src/cuteSV/):
 *   chain()           resolution_DEL/INS/DUP/INV/TRA outer loops
 *                     cuteSV_resolveINDEL.py:55-100, 261-310; cuteSV_resolveDUP.py:28-70;
 *                     cuteSV_resolveINV.py:45-92; cuteSV_resolveTRA.py:39-102
 *   refine_indel()    generate_del_cluster / generate_ins_cluster  cuteSV_resolveINDEL.py:110-219, 319-432
 *   refine_dup()      generate_dup_cluster        cuteSV_resolveDUP.py:79-131
 *   refine_inv()      generate_semi_inv_cluster   cuteSV_resolveINV.py:101-203
 *   refine_tra()      generate_semi_tra_cluster   cuteSV_resolveTRA.py:106-254
 *   genotype()        call_gt -> overlap_cover -> assign_gt
 *                     cuteSV_resolveINDEL.py:441-458, cuteSV_resolveDUP.py:137-165,
 *                     cuteSV_resolveINV.py:208-235, cuteSV_genotype.py:95-173
 *   csvo_gl_index()   cal_GL special cases + rescale_read_counts  cuteSV_genotype.py:25-39
 * numpy arithmetic on the path (np.mean, np.std) is restated bit for bit: np_sum_f64() is
 * numpy 2.2's pairwise summation (8 strided accumulators per <=128 block, recursive halving,
 * 8192-element buffer chunks), verified against numpy in tests/test_oracle_golden.py.
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off: no FMA contraction, IEEE double)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/cutesv_hip.h"

/* ------------------------------------------------------------------ numpy arithmetic */

static double pairwise_f64(const double* a, int64_t n)
{
    if (n < 8) {
        double r = 0.0;
        for (int64_t i = 0; i < n; i++) r += a[i];
        return r;
    }
    if (n <= 128) {
        double r[8];
        int64_t i;
        for (int j = 0; j < 8; j++) r[j] = a[j];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    }
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    return pairwise_f64(a, n2) + pairwise_f64(a + n2, n - n2);
}

/* np.add.reduce over a contiguous float64 vector (numpy 2.2, default bufsize 8192). */
double csvo_np_sum_f64(const double* a, int64_t n)
{
    double acc = 0.0;
    for (int64_t i = 0; i < n; i += 8192) {
        int64_t c = n - i < 8192 ? n - i : 8192;
        acc += pairwise_f64(a + i, c);
    }
    return acc;
}

/* np.std(list_of_ints): mean = sum/n (exact integer sum), sqrt(sum((x-mean)^2)/n). */
double csvo_np_std_i64(const int64_t* v, int64_t n, double* scratch)
{
    int64_t s = 0;
    for (int64_t i = 0; i < n; i++) s += v[i];
    double mean = (double)s / (double)n;
    for (int64_t i = 0; i < n; i++) {
        double x = (double)v[i] - mean;
        scratch[i] = x * x;
    }
    return sqrt(csvo_np_sum_f64(scratch, n) / (double)n);
}

/* cal_CIPOS (cuteSV_genotype.py:58-60): int(1.96 * std / num ** 0.5).  `num ** 0.5` is libm
 * pow(), which is NOT always equal to sqrt() (first difference at num = 2921). */
int32_t csvo_cipos(double std, int64_t num)
{
    return (int32_t)(1.96 * std / pow((double)num, 0.5));
}

/* cal_GL's input normalisation (cuteSV_genotype.py:25-39) -> table index. */
int32_t csvo_gl_index(int64_t c0, int64_t c1)
{
    if (c0 == 3 && c1 == 1) return 101 * 101;
    if (c0 == 6 && c1 == 2) return 101 * 101 + 1;
    int64_t total = c0 + c1;
    if (total > 100) {
        double frac = (double)c0 / (double)total;
        c0 = (int64_t)(100.0 * frac);
        c1 = 100 - c0;
    }
    return (int32_t)(c0 * 101 + c1);
}

/* ------------------------------------------------------------------ small helpers */

typedef struct {
    const csv_batch_in* in;
    csv_batch_out*      out;
    int64_t             n_calls;
    int64_t             n_support;
    /* per-call pending genotype info is taken from the out arrays themselves */
    /* grow-only scratch */
    int64_t  cap;
    int64_t* idx;       /* working permutation */
    int64_t* idx2;
    int64_t* tmp;
    int64_t* vals;
    int64_t* vals2;
    double*  dev;
    double*  sq;
    int32_t* ids;
    int32_t* ids2;
    /* reads: prefix max of r_end per chromosome block */
    int64_t* pmax;
    /* start-ordered copy of the reads table (when the caller's blocks are not sorted) */
    csv_batch_in in_sorted;
    int64_t* s_start; int64_t* s_end; uint8_t* s_primary; int32_t* s_id;
} work_t;

static int ensure(work_t* w, int64_t m)
{
    if (m <= w->cap) return 0;
    int64_t c = w->cap ? w->cap : 1024;
    while (c < m) c *= 2;
#define GROW(p, T) do { void* q = realloc(w->p, (size_t)c * sizeof(T)); if (!q) return -1; w->p = (T*)q; } while (0)
    GROW(idx, int64_t); GROW(idx2, int64_t); GROW(tmp, int64_t); GROW(vals, int64_t); GROW(vals2, int64_t);
    GROW(dev, double); GROW(sq, double); GROW(ids, int32_t); GROW(ids2, int32_t);
#undef GROW
    w->cap = c;
    return 0;
}

/* stable merge sort of an index array by an int64 key column */
static void msort_i64(int64_t* idx, int64_t* tmp, int64_t n, const int64_t* key)
{
    if (n < 2) return;
    if (n <= 16) {
        for (int64_t i = 1; i < n; i++) {
            int64_t x = idx[i], k = key[x], j = i;
            while (j > 0 && key[idx[j - 1]] > k) { idx[j] = idx[j - 1]; j--; }
            idx[j] = x;
        }
        return;
    }
    int64_t h = n / 2;
    msort_i64(idx, tmp, h, key);
    msort_i64(idx + h, tmp, n - h, key);
    int64_t i = 0, j = h, o = 0;
    while (i < h && j < n) tmp[o++] = (key[idx[j]] < key[idx[i]]) ? idx[j++] : idx[i++];
    while (i < h) tmp[o++] = idx[i++];
    while (j < n) tmp[o++] = idx[j++];
    memcpy(idx, tmp, (size_t)n * sizeof(int64_t));
}

/* stable sort of local indices 0..n-1 by a double key (deviation sort, INDEL:171-173) */
static void msort_f64(int64_t* idx, int64_t* tmp, int64_t n, const double* key)
{
    if (n < 2) return;
    if (n <= 16) {
        for (int64_t i = 1; i < n; i++) {
            int64_t x = idx[i], j = i;
            double  k = key[x];
            while (j > 0 && key[idx[j - 1]] > k) { idx[j] = idx[j - 1]; j--; }
            idx[j] = x;
        }
        return;
    }
    int64_t h = n / 2;
    msort_f64(idx, tmp, h, key);
    msort_f64(idx + h, tmp, n - h, key);
    int64_t i = 0, j = h, o = 0;
    while (i < h && j < n) tmp[o++] = (key[idx[j]] < key[idx[i]]) ? idx[j++] : idx[i++];
    while (i < h) tmp[o++] = idx[i++];
    while (j < n) tmp[o++] = idx[j++];
    memcpy(idx, tmp, (size_t)n * sizeof(int64_t));
}

static int cmp_i32(const void* x, const void* y)
{
    int32_t a = *(const int32_t*)x, b = *(const int32_t*)y;
    return (a > b) - (a < b);
}

/* number of distinct ids among ids[0..n) (destroys order) */
static int64_t count_unique(int32_t* ids, int64_t n)
{
    if (n == 0) return 0;
    qsort(ids, (size_t)n, sizeof(int32_t), cmp_i32);
    int64_t u = 1;
    for (int64_t i = 1; i < n; i++) u += ids[i] != ids[i - 1];
    return u;
}

/* begin a call; returns its slot (may be beyond capacity: then fields are not written) */
static int64_t call_begin(work_t* w, int32_t seg, int32_t cluster, int32_t aux)
{
    int64_t c = w->n_calls++;
    csv_batch_out* o = w->out;
    if (c < o->cap_calls) {
        o->call_seg[c] = seg;
        o->call_cluster[c] = cluster;
        /* aux is not part of a DEL / DUP signature ("ignored", include/cutesv_hip.h): reported as 0 */
        o->call_aux[c] = (w->in->seg[seg].svtype == CSV_DEL || w->in->seg[seg].svtype == CSV_DUP) ? 0 : aux;
        o->bp1[c] = 0; o->bp2[c] = 0; o->support[c] = 0; o->cipos[c] = 0; o->cilen[c] = 0;
        o->search_pos[c] = 0; o->seq_pick[c] = -1; o->dr[c] = -1; o->dv[c] = -1; o->gl_idx[c] = -1;
        o->support_off[c] = w->n_support;
    }
    return c;
}

static void call_add_support(work_t* w, int64_t c, int64_t sig)
{
    csv_batch_out* o = w->out;
    int64_t s = w->n_support++;
    if (s < o->cap_support) o->support_sig[s] = sig;
    if (o->allele_id && c < o->cap_calls) o->allele_id[sig] = (int32_t)c;
}

static void call_end(work_t* w, int64_t c)
{
    csv_batch_out* o = w->out;
    if (c < o->cap_calls) o->support_off[c + 1] = w->n_support;
}

/* ------------------------------------------------------------------ DEL / INS */

/* generate_del_cluster / generate_ins_cluster (INDEL:110-219, 319-432) on the chained
 * cluster [s, e) of segment sg. */
static int refine_indel(work_t* w, const csv_segment* sg, int32_t seg_i, int32_t cid, int64_t s, int64_t e)
{
    const csv_batch_in* in = w->in;
    const int64_t m = e - s;
    if (ensure(w, m + 2)) return CSV_E_NOMEM;
    const int is_ins = sg->svtype == CSV_INS;
    double rr = sg->remain_reads_ratio;
    if (rr > 1) rr = 1;                                    /* INDEL:46-47 */

    /* Remove duplicates (INDEL:125-131): dict keyed by read; first appearance keeps the slot,
     * a strictly longer later signature of the same read replaces the value. */
    int64_t* chosen = w->idx2;  /* chosen[u] = global index kept for the u-th distinct read */
    int64_t  U = 0;
    for (int64_t j = s; j < e; j++) {
        int64_t u;
        for (u = 0; u < U; u++)
            if (in->read_id[chosen[u]] == in->read_id[j]) break;
        if (u == U) chosen[U++] = j;
        else if (in->b[j] > in->b[chosen[u]]) chosen[u] = j;
    }
    if (U < sg->read_count) return CSV_OK;                 /* INDEL:133-134 */

    /* sorted(..., key=len) is stable (INDEL:136) */
    int64_t* ord = w->idx;
    for (int64_t u = 0; u < U; u++) { ord[u] = u; w->vals[u] = in->b[chosen[u]]; }
    msort_i64(ord, w->tmp, U, w->vals);
    int64_t sum_len = 0;
    for (int64_t u = 0; u < U; u++) sum_len += w->vals[u];
    const double thr = sg->diff_ratio * ((double)sum_len / (double)U);   /* INDEL:138 */

    /* allele split on consecutive length gaps (INDEL:153-162); alleles are ranges of ord[] */
    int64_t* astart = w->tmp;            /* reuse: A+1 entries, A <= U */
    int64_t  A = 0;
    astart[A++] = 0;
    for (int64_t r = 1; r < U; r++) {
        int64_t gap = w->vals[ord[r]] - w->vals[ord[r - 1]];
        if ((double)gap > thr) astart[A++] = r;
    }
    astart[A] = U;

    /* allele_sort = sorted(allele_collect, key=count) — stable ascending (INDEL:163) */
    int64_t* aord = (int64_t*)malloc((size_t)(A + 1) * 3 * sizeof(int64_t));
    if (!aord) return CSV_E_NOMEM;
    int64_t* acnt = aord + A + 1;
    int64_t* atmp = acnt + A + 1;
    for (int64_t k = 0; k < A; k++) { aord[k] = k; acnt[k] = astart[k + 1] - astart[k]; }
    msort_i64(aord, atmp, A, acnt);
    /* astart lives in w->tmp which msort above did not touch (it used atmp) */

    int64_t* P = w->vals2;   /* positions of the current allele, allele order */
    int64_t* L = (int64_t*)malloc((size_t)(U + 1) * 3 * sizeof(int64_t));
    if (!L) { free(aord); return CSV_E_NOMEM; }
    int64_t* loc = L + U + 1;
    int64_t* ltmp = loc + U + 1;

    for (int64_t q = 0; q < A; q++) {
        const int64_t k = aord[q];
        const int64_t n = acnt[k];
        if (n < sg->min_support_reads) continue;           /* INDEL:166 */
        const int64_t r0 = astart[k];
        int64_t sp = 0, sl = 0;
        for (int64_t i = 0; i < n; i++) {
            int64_t g = chosen[ord[r0 + i]];
            P[i] = in->a[g]; L[i] = in->b[g];
            sp += P[i]; sl += L[i];
        }
        int64_t keep = (int64_t)(rr * (double)n);          /* INDEL:169 */
        if (keep < 1) keep = 1;

        const double pos_mean = (double)sp / (double)n;
        for (int64_t i = 0; i < n; i++) { w->dev[i] = fabs((double)P[i] - pos_mean); loc[i] = i; }
        msort_f64(loc, ltmp, n, w->dev);                   /* INDEL:171-173 */
        int64_t ks = 0;
        for (int64_t i = 0; i < keep; i++) ks += P[loc[i]];
        double  bp = (double)ks / (double)keep;            /* INDEL:176 */
        int64_t search = P[loc[0]];                        /* INDEL:177 */

        const double len_mean = (double)sl / (double)n;
        for (int64_t i = 0; i < n; i++) { w->dev[i] = fabs((double)L[i] - len_mean); loc[i] = i; }
        msort_f64(loc, ltmp, n, w->dev);
        ks = 0;
        for (int64_t i = 0; i < keep; i++) ks += L[loc[i]];
        const double sig_len = (double)ks / (double)keep;  /* INDEL:187 */

        const int32_t cipos = csvo_cipos(csvo_np_std_i64(P, n, w->sq), n);   /* INDEL:191 */
        const int32_t cilen = csvo_cipos(csvo_np_std_i64(L, n, w->sq), n);   /* INDEL:194 */

        int64_t pick = -1;
        if (is_ins) {                                      /* INDEL:398-405 */
            const int64_t want = (int64_t)sig_len;
            for (int64_t i = 0; i < n; i++) {
                int64_t g = chosen[ord[r0 + i]];
                if ((int64_t)in->aux[g] >= want) { pick = g; bp = (double)P[i]; break; }
            }
            if (pick < 0) continue;
            search = (int64_t)bp;                          /* INDEL:415 */
        }

        int64_t c = call_begin(w, seg_i, cid, in->aux[s]);
        if (c < w->out->cap_calls) {
            csv_batch_out* o = w->out;
            o->bp1[c] = (int64_t)bp;                       /* INDEL:199 / 410 */
            o->bp2[c] = (int64_t)sig_len;                  /* INDEL:200 (sign applied by the host) / 411 */
            o->support[c] = (int32_t)n;
            o->cipos[c] = cipos;
            o->cilen[c] = cilen;
            o->search_pos[c] = search;
            o->seq_pick[c] = pick;
        }
        for (int64_t i = 0; i < n; i++) call_add_support(w, c, chosen[ord[r0 + i]]);
        call_end(w, c);
    }
    free(L);
    free(aord);
    return CSV_OK;
}

/* first-seen distinct reads of the signatures ord[r0..r1) -> support list; returns count */
static int64_t emit_unique_support(work_t* w, int64_t c, const int64_t* ord, int64_t r0, int64_t r1)
{
    const csv_batch_in* in = w->in;
    int64_t u = 0;
    for (int64_t r = r0; r < r1; r++) {
        int32_t id = in->read_id[ord[r]];
        int64_t k;
        for (k = 0; k < u; k++) if (w->ids2[k] == id) break;
        if (k == u) { w->ids2[u++] = id; call_add_support(w, c, ord[r]); }
    }
    return u;
}

static int64_t unique_reads(work_t* w, const int64_t* ord, int64_t r0, int64_t r1)
{
    for (int64_t r = r0; r < r1; r++) w->ids[r - r0] = w->in->read_id[ord[r]];
    return count_unique(w->ids, r1 - r0);
}

/* ------------------------------------------------------------------ DUP */

static int refine_dup(work_t* w, const csv_segment* sg, int32_t seg_i, int32_t cid, int64_t s, int64_t e)
{
    const csv_batch_in* in = w->in;
    const int64_t m = e - s;
    if (ensure(w, m)) return CSV_E_NOMEM;
    int64_t* ord = w->idx;
    for (int64_t i = 0; i < m; i++) ord[i] = s + i;
    if (unique_reads(w, ord, 0, m) < sg->read_count) return CSV_OK;      /* DUP:82-84 */
    msort_i64(ord, w->tmp, m, in->b);                                     /* DUP:86 */
    int64_t r0 = 0;
    for (int64_t r = 1; r <= m; r++) {
        if (r < m && !(in->b[ord[r]] - in->b[ord[r - 1]] > sg->max_cluster_bias)) continue;  /* DUP:91 */
        const int64_t n = r - r0;
        const int64_t uniq = unique_reads(w, ord, r0, r);
        if (uniq >= sg->read_count) {                                     /* DUP:96-98 */
            const int64_t lo = (int64_t)((double)n * 0.4);                /* DUP:99-100 */
            const int64_t hi = (int64_t)((double)n * 0.6);
            int64_t bp1, bp2;
            if (lo == hi) {
                bp1 = in->a[ord[r0 + lo]];
                bp2 = in->b[ord[r0 + lo]];
            } else {
                int64_t s1 = 0, s2 = 0;
                for (int64_t i = lo; i < hi; i++) { s1 += in->a[ord[r0 + i]]; s2 += in->b[ord[r0 + i]]; }
                bp1 = (int64_t)((double)s1 / (double)(hi - lo));          /* DUP:108-109 */
                bp2 = (int64_t)((double)s2 / (double)(hi - lo));
            }
            const int64_t d = bp2 - bp1;
            if ((sg->sv_size <= d && d <= sg->max_size) || (sg->sv_size <= d && sg->max_size == -1)) {  /* DUP:112 */
                int64_t c = call_begin(w, seg_i, cid, in->aux[s]);
                int64_t u = emit_unique_support(w, c, ord, r0, r);
                if (c < w->out->cap_calls) {
                    w->out->bp1[c] = bp1; w->out->bp2[c] = bp2; w->out->support[c] = (int32_t)u;
                }
                call_end(w, c);
            }
        }
        r0 = r;
    }
    return CSV_OK;
}

/* ------------------------------------------------------------------ INV */

static int refine_inv(work_t* w, const csv_segment* sg, int32_t seg_i, int32_t cid, int64_t s, int64_t e)
{
    const csv_batch_in* in = w->in;
    const int64_t m = e - s;
    if (ensure(w, m)) return CSV_E_NOMEM;
    int64_t* ord = w->idx;
    for (int64_t i = 0; i < m; i++) ord[i] = s + i;
    if (unique_reads(w, ord, 0, m) < sg->read_count) return CSV_OK;      /* INV:106-109 */
    msort_i64(ord, w->tmp, m, in->b);                                     /* INV:111 */
    int64_t r0 = 0;
    for (int64_t r = 1; r <= m; r++) {
        if (r < m && !(in->b[ord[r]] - in->b[ord[r - 1]] > sg->max_cluster_bias)) continue;  /* INV:125 */
        const int64_t n = r - r0;
        if (n >= sg->read_count) {                                        /* INV:126 / 173 */
            int64_t s1 = 0, s2 = 0;
            for (int64_t i = r0; i < r; i++) { s1 += in->a[ord[i]]; s2 += in->b[ord[i]]; }
            const int64_t uniq = unique_reads(w, ord, r0, r);
            /* Python round(): float64 quotient rounded half-to-even (INV:129-130) */
            const int64_t bp1 = (int64_t)rint((double)s1 / (double)n);
            const int64_t bp2 = (int64_t)rint((double)s2 / (double)n);
            const int64_t len = bp2 - bp1;
            if (len >= sg->sv_size && uniq >= sg->read_count &&
                (len <= sg->max_size || sg->max_size == -1)) {            /* INV:132-134 */
                int64_t c = call_begin(w, seg_i, cid, in->aux[s]);
                int64_t u = emit_unique_support(w, c, ord, r0, r);
                if (c < w->out->cap_calls) {
                    w->out->bp1[c] = bp1; w->out->bp2[c] = bp2; w->out->support[c] = (int32_t)u;
                }
                call_end(w, c);
            }
        }
        r0 = r;
    }
    return CSV_OK;
}

/* ------------------------------------------------------------------ TRA */

static int refine_tra(work_t* w, const csv_segment* sg, int32_t seg_i, int32_t cid, int64_t s, int64_t e)
{
    const csv_batch_in* in = w->in;
    const int64_t m = e - s;
    if (ensure(w, m + 1)) return CSV_E_NOMEM;
    int64_t* ord = w->idx;
    for (int64_t i = 0; i < m; i++) ord[i] = s + i;
    const int32_t aux0 = in->aux[s];
    msort_i64(ord, w->tmp, m, in->b);                                     /* TRA:109 */
    /* sub-clusters over pos2 gaps; the loop at TRA:116-124 revisits element 0, so the first
     * sub-cluster's sums and its read list contain element 0 twice. */
    int64_t nsub = 0;
    int64_t* sub0 = w->vals;      /* start rank of each sub-cluster */
    sub0[nsub++] = 0;
    for (int64_t r = 1; r < m; r++)
        if (in->b[ord[r]] - in->b[ord[r - 1]] > sg->max_cluster_bias) sub0[nsub++] = r;
    sub0[nsub] = m;
    if (unique_reads(w, ord, 0, m) < sg->read_count) return CSV_OK;      /* TRA:128-129 */

    /* temp = sorted(temp, key=-len(set(reads))) — stable; only the first two matter */
    int64_t best = -1, second = -1, ubest = -1, usecond = -1;
    for (int64_t k = 0; k < nsub; k++) {
        int64_t u = unique_reads(w, ord, sub0[k], sub0[k + 1]);
        if (u > ubest) { second = best; usecond = ubest; best = k; ubest = u; }
        else if (u > usecond) { second = k; usecond = u; }
    }
    const int type = aux0 & 7;
    if (type > 3) return CSV_OK;                                          /* TRA:154-155, 226-227 */

    int64_t emit[2]; int n_emit = 0;
    if (nsub > 1 && (double)usecond >= 0.5 * (double)sg->read_count) {    /* TRA:133 */
        if ((double)(ubest + usecond) >= (double)m * sg->diff_ratio) { emit[0] = best; emit[1] = second; n_emit = 2; }  /* TRA:134 */
    } else {
        if ((double)ubest >= (double)m * sg->diff_ratio) { emit[0] = best; n_emit = 1; }   /* TRA:211 */
    }
    for (int q = 0; q < n_emit; q++) {
        const int64_t k = emit[q];
        int64_t s1 = 0, s2 = 0, cnt = sub0[k + 1] - sub0[k];
        for (int64_t r = sub0[k]; r < sub0[k + 1]; r++) { s1 += in->a[ord[r]]; s2 += in->b[ord[r]]; }
        if (k == 0) { s1 += in->a[ord[0]]; s2 += in->b[ord[0]]; cnt += 1; }   /* the double count */
        int64_t c = call_begin(w, seg_i, cid, aux0);
        int64_t u = emit_unique_support(w, c, ord, sub0[k], sub0[k + 1]);
        if (c < w->out->cap_calls) {
            w->out->bp1[c] = (int64_t)((double)s1 / (double)cnt);         /* TRA:173 */
            w->out->bp2[c] = (int64_t)((double)s2 / (double)cnt);         /* TRA:175 */
            w->out->support[c] = (int32_t)u;
        }
        call_end(w, c);
    }
    return CSV_OK;
}

/* ------------------------------------------------------------------ chaining */

static int seg_break(const csv_batch_in* in, const csv_segment* sg, int64_t i)
{
    /* predicate between file-order neighbours i-1 and i */
    const int64_t da = in->a[i] - in->a[i - 1];
    switch (sg->svtype) {
    case CSV_DEL: case CSV_INS: case CSV_DUP:
        return da > sg->max_cluster_bias;                                  /* INDEL:61,271; DUP:35 */
    case CSV_INV:
        return da > sg->max_cluster_bias || (in->b[i] - in->b[i - 1]) > sg->max_cluster_bias ||
               in->aux[i] != in->aux[i - 1];                               /* INV:56 */
    default:
        return da > sg->max_cluster_bias || in->aux[i] != in->aux[i - 1];  /* TRA:41,65 */
    }
}

static int refine(work_t* w, const csv_segment* sg, int32_t seg_i, int32_t cid, int64_t s, int64_t e)
{
    const csv_batch_in* in = w->in;
    /* the size gate counts signatures (INDEL:62,86) and a trailing (0,0) element behaves like
     * the reference's [0,0,''] sentinel: the cluster is skipped (INDEL:63-64) */
    if (e - s < sg->read_count) return CSV_OK;
    if (in->a[e - 1] == 0 && in->b[e - 1] == 0) return CSV_OK;
    /* the build's only per-segment condition (csv_batch_out.seg_status): a length / pos2 value that no genome produces,
     * outside [0, 2^42) ([0, 2^32) inside a chained cluster of more than 2^21 signatures).  The cluster emits nothing. */
    {
        const int bits = (e - s) > (1 << 21) ? 32 : 42;
        for (int64_t i = s; i < e; i++)
            if (((uint64_t)in->b[i]) >> bits) { if (w->out->seg_status) w->out->seg_status[seg_i] |= CSV_SEG_KEY_RANGE; return CSV_OK; }
    }
    switch (sg->svtype) {
    case CSV_DEL: case CSV_INS: return refine_indel(w, sg, seg_i, cid, s, e);
    case CSV_DUP: return refine_dup(w, sg, seg_i, cid, s, e);
    case CSV_INV: return refine_inv(w, sg, seg_i, cid, s, e);
    default:      return refine_tra(w, sg, seg_i, cid, s, e);
    }
}

/* ------------------------------------------------------------------ genotype */

/* cover set of one window in doubled coordinates: primary reads with 2*start <= L2 and
 * 2*end >= R2 (the net effect of GT:95-159; semantics spelled out by duipai, GT:206-212). */
static int64_t collect_cover(work_t* w, int64_t r0, int64_t r1, int64_t L2, int64_t R2, int32_t* dst, int64_t n)
{
    const csv_batch_in* in = w->in;
    /* upper bound of start <= L */
    int64_t lo = r0, hi = r1;
    while (lo < hi) {
        int64_t mid = lo + (hi - lo) / 2;
        if (2 * in->r_start[mid] <= L2) lo = mid + 1; else hi = mid;
    }
    for (int64_t i = lo - 1; i >= r0; i--) {
        if (2 * w->pmax[i] < R2) break;           /* nothing at or before i reaches R */
        if (in->r_primary[i] == 1 && 2 * in->r_end[i] >= R2) dst[n++] = in->r_id[i];
    }
    return n;
}

/* threshold_ref_count (GT:62-70) */
static int64_t tra_up_bound(int64_t num)
{
    if (num <= 2) return 20 * num;
    if (num <= 5) return 9 * num;
    if (num <= 15) return 7 * num;
    return 5 * num;
}

/* count_coverage (GT:72-93) with `f.fetch(chr, s, e)` restated over the reads block of chromosome ch: the
 * reads with start < e and end > s, in block (start-sorted) order; `flag in [0, 16]` is r_primary.
 * q / nq: the querydata set shared by the two windows (TRA:262). */
static int tra_count_coverage(work_t* w, int32_t ch, int64_t s, int64_t e, int32_t* q, int64_t* nq,
                              int64_t up_bound, int64_t itround)
{
    const csv_batch_in* in = w->in;
    int status = 0;
    int64_t iteration = 0, primary_num = 0;
    if (s >= e) return 0;
    for (int64_t i = in->reads_off[ch]; i < in->reads_off[ch + 1]; i++) {
        if (in->r_start[i] >= e) break;
        if (in->r_end[i] <= s) continue;
        iteration++;                                                     /* GT:77 */
        if (in->r_primary[i] != 1) continue;                             /* GT:78-79 */
        primary_num++;
        if (in->r_start[i] < s && in->r_end[i] > e) {                    /* GT:81-85 */
            int64_t j = 0;
            while (j < *nq && q[j] != in->r_id[i]) j++;
            if (j == *nq) q[(*nq)++] = in->r_id[i];
            if (*nq >= up_bound) { status = 1; break; }
        }
        if (iteration >= itround) {                                      /* GT:86-91 */
            status = ((double)primary_num / (double)iteration <= 0.2) ? 1 : -1;
            break;
        }
    }
    return status;
}

/* call_gt of cuteSV_resolveTRA.py:258-309 for call c */
static int tra_genotype(work_t* w, int64_t c, const csv_segment* sg, int32_t** qbuf, int64_t* qcap)
{
    const csv_batch_in* in = w->in;
    csv_batch_out* o = w->out;
    const int64_t ns = o->support_off[c + 1] - o->support_off[c];      /* len(read_id_list): a set of names */
    const int64_t up_bound = tra_up_bound(ns);                           /* TRA:266 */
    const int32_t chr1 = sg->chrom, chr2 = o->call_aux[c] >> 3;
    if (chr2 < 0 || chr2 >= in->n_chrom) return CSV_E_INVALID;
    if (*qcap < up_bound + 2) {
        int32_t* nq_ = (int32_t*)realloc(*qbuf, (size_t)(up_bound + 2) * sizeof(int32_t));
        if (!nq_) return CSV_E_NOMEM;
        *qbuf = nq_; *qcap = up_bound + 2;
    }
    int32_t* q = *qbuf;
    int64_t nq = 0;
    const int64_t bias = sg->gt_bias;
    int64_t s = o->bp1[c] - bias, e = o->bp1[c] + bias;                 /* TRA:263-264 */
    if (s < 0) s = 0;
    if (e > in->contig_len[chr1]) e = in->contig_len[chr1];
    int status = tra_count_coverage(w, chr1, s, e, q, &nq, up_bound, sg->gt_round);
    o->dv[c] = (int32_t)ns;
    if (status == -1) { o->dr[c] = -1; o->gl_idx[c] = -1; return CSV_OK; }   /* TRA:276-281 */
    if (status == 0) {                                                   /* TRA:290-299: status_2 is not looked at */
        s = o->bp2[c] - bias; e = o->bp2[c] + bias;
        if (s < 0) s = 0;
        if (e > in->contig_len[chr2]) e = in->contig_len[chr2];
        tra_count_coverage(w, chr2, s, e, q, &nq, up_bound, sg->gt_round);
    }
    int64_t dr = 0;                                                      /* TRA:284-287, 301-304 */
    for (int64_t j = 0; j < nq; j++) {
        int found = 0;
        for (int64_t i = 0; i < ns && !found; i++) found = in->read_id[o->support_sig[o->support_off[c] + i]] == q[j];
        if (!found) dr++;
    }
    o->dr[c] = (int32_t)dr;
    o->gl_idx[c] = csvo_gl_index(dr, ns);
    return CSV_OK;
}

static int genotype_all(work_t* w)
{
    const csv_batch_in* in = w->in;
    csv_batch_out* o = w->out;
    const int64_t nc = w->n_calls < o->cap_calls ? w->n_calls : o->cap_calls;
    int any = 0;
    for (int32_t k = 0; k < in->n_seg; k++) any |= in->seg[k].genotype;
    if (!any || !in->reads_off) return CSV_OK;
    /* overlap_cover sorts its events itself (GT:101-109) and count_coverage walks the BAM in start order: bring every
     * block into STABLE start order first (the order of equal starts is the order of the block) */
    {
        int sorted = 1;
        for (int32_t ch = 0; ch < in->n_chrom && sorted; ch++)
            for (int64_t i = in->reads_off[ch] + 1; i < in->reads_off[ch + 1]; i++)
                if (in->r_start[i] < in->r_start[i - 1]) { sorted = 0; break; }
        if (!sorted) {
            if (in->flags & CSV_IN_READS_SORTED) return CSV_E_UNSORTED;
            const int64_t R = in->n_reads;
            int64_t* perm = (int64_t*)malloc((size_t)(R + 1) * sizeof(int64_t));
            int64_t* tmp = (int64_t*)malloc((size_t)(R + 1) * sizeof(int64_t));
            w->s_start = (int64_t*)malloc((size_t)(R + 1) * sizeof(int64_t));
            w->s_end = (int64_t*)malloc((size_t)(R + 1) * sizeof(int64_t));
            w->s_primary = (uint8_t*)malloc((size_t)(R + 1));
            w->s_id = (int32_t*)malloc((size_t)(R + 1) * sizeof(int32_t));
            if (!perm || !tmp || !w->s_start || !w->s_end || !w->s_primary || !w->s_id) { free(perm); free(tmp); return CSV_E_NOMEM; }
            for (int64_t i = 0; i < R; i++) perm[i] = i;
            for (int32_t ch = 0; ch < in->n_chrom; ch++)
                msort_i64(perm + in->reads_off[ch], tmp, in->reads_off[ch + 1] - in->reads_off[ch], in->r_start);
            for (int64_t i = 0; i < R; i++) {
                const int64_t p = (i >= in->reads_off[0] && i < in->reads_off[in->n_chrom]) ? perm[i] : i;
                w->s_start[i] = in->r_start[p]; w->s_end[i] = in->r_end[p]; w->s_primary[i] = in->r_primary[p]; w->s_id[i] = in->r_id[p];
            }
            free(perm); free(tmp);
            w->in_sorted = *in;
            w->in_sorted.r_start = w->s_start; w->in_sorted.r_end = w->s_end; w->in_sorted.r_primary = w->s_primary; w->in_sorted.r_id = w->s_id;
            w->in = in = &w->in_sorted;
        }
    }
    w->pmax = (int64_t*)malloc((size_t)(in->n_reads + 1) * sizeof(int64_t));
    if (!w->pmax) return CSV_E_NOMEM;
    for (int32_t ch = 0; ch < in->n_chrom; ch++) {
        int64_t mx = INT64_MIN;
        for (int64_t i = in->reads_off[ch]; i < in->reads_off[ch + 1]; i++) {
            if (i > in->reads_off[ch] && in->r_start[i] < in->r_start[i - 1]) return CSV_E_UNSORTED;
            if (in->r_end[i] > mx) mx = in->r_end[i];
            w->pmax[i] = mx;
        }
    }
    int64_t cap = 1024;
    int32_t* cov = (int32_t*)malloc((size_t)cap * sizeof(int32_t));
    int32_t* sup = NULL; int64_t sup_cap = 0;
    int32_t* qbuf = NULL; int64_t qcap = 0;
    if (!cov) return CSV_E_NOMEM;
    for (int64_t c = 0; c < nc; c++) {
        const csv_segment* sg = &in->seg[o->call_seg[c]];
        if (!sg->genotype) continue;
        if (sg->svtype == CSV_TRA) {
            const int rc = tra_genotype(w, c, sg, &qbuf, &qcap);
            if (rc) { free(cov); free(sup); free(qbuf); return rc; }
            continue;
        }
        const int64_t r0 = in->reads_off[sg->chrom], r1 = in->reads_off[sg->chrom + 1];
        if (cap < 2 * (r1 - r0) + 16) {
            cap = 2 * (r1 - r0) + 16;
            int32_t* q = (int32_t*)realloc(cov, (size_t)cap * sizeof(int32_t));
            if (!q) { free(cov); free(sup); free(qbuf); return CSV_E_NOMEM; }
            cov = q;
        }
        int64_t n = 0;
        if (sg->svtype == CSV_DEL || sg->svtype == CSV_INS) {
            const int64_t p = o->search_pos[c], g = sg->gt_bias;         /* INDEL:450-451 */
            int64_t L = p - g; if (L < 0) L = 0;
            n = collect_cover(w, r0, r1, 2 * L, 2 * (p + g), cov, n);
        } else {
            int64_t nb = sg->gt_bias;
            if (sg->svtype == CSV_DUP) { int64_t d = o->bp2[c] - o->bp1[c]; if (d < nb) nb = d; }   /* DUP:147 */
            /* windows (max(bp - nb/2, 0), bp + nb/2) as doubled integers (DUP:148-151, INV:219-221) */
            int64_t L2 = 2 * o->bp1[c] - nb; if (L2 < 0) L2 = 0;
            n = collect_cover(w, r0, r1, L2, 2 * o->bp1[c] + nb, cov, n);
            L2 = 2 * o->bp2[c] - nb; if (L2 < 0) L2 = 0;
            n = collect_cover(w, r0, r1, L2, 2 * o->bp2[c] + nb, cov, n);   /* union: DUP:155-157 */
        }
        /* distinct names (cover sets hold names, GT:149-152) */
        int64_t u = 0;
        if (n) { qsort(cov, (size_t)n, sizeof(int32_t), cmp_i32); u = 1; for (int64_t i = 1; i < n; i++) if (cov[i] != cov[u - 1]) cov[u++] = cov[i]; }
        const int64_t ns = o->support_off[c + 1] - o->support_off[c];
        if (ns > sup_cap) { sup_cap = ns * 2; int32_t* q = (int32_t*)realloc(sup, (size_t)sup_cap * sizeof(int32_t)); if (!q) { free(cov); free(sup); free(qbuf); return CSV_E_NOMEM; } sup = q; }
        for (int64_t i = 0; i < ns; i++) sup[i] = in->read_id[o->support_sig[o->support_off[c] + i]];
        qsort(sup, (size_t)ns, sizeof(int32_t), cmp_i32);
        int64_t dr = 0;                                                    /* GT:167-170 */
        for (int64_t i = 0; i < u; i++)
            if (!bsearch(&cov[i], sup, (size_t)ns, sizeof(int32_t), cmp_i32)) dr++;
        o->dr[c] = (int32_t)dr;
        o->dv[c] = (int32_t)ns;                                            /* GT:171-172: len(read_id_dict[idx]) */
        o->gl_idx[c] = csvo_gl_index(dr, ns);
    }
    free(cov); free(sup); free(qbuf);
    return CSV_OK;
}

/* ------------------------------------------------------------------ entry point */

int csvo_cluster_batch(const csv_batch_in* in, csv_batch_out* out)
{
    work_t w;
    memset(&w, 0, sizeof w);
    w.in = in; w.out = out;
    int rc = CSV_OK;
    int32_t cid = -1;
    if (out->seg_status) for (int32_t k = 0; k < in->n_seg; k++) out->seg_status[k] = 0;
    if (out->allele_id) for (int64_t i = 0; i < in->n_sig; i++) out->allele_id[i] = -1;
    if (out->cluster_id) for (int64_t i = 0; i < in->n_sig; i++) out->cluster_id[i] = -1;
    for (int32_t k = 0; k < in->n_seg && rc == CSV_OK; k++) {
        const csv_segment* sg = &in->seg[k];
        if (sg->svtype < CSV_DEL || sg->svtype > CSV_TRA || sg->sig_begin > sg->sig_end ||
            sg->sig_end > in->n_sig || (sg->svtype == CSV_TRA && sg->genotype && (!in->contig_len || !in->reads_off))) { rc = CSV_E_INVALID; break; }
        /* a genotyped segment whose chromosome has no reads block yields nothing (INDEL:443-444, DUP:139-140,
         * INV:210-211); TRA genotyping has no such gate (TRA:258-309) */
        const int drop = sg->genotype && sg->svtype != CSV_TRA && (!in->reads_off || in->reads_off[sg->chrom + 1] == in->reads_off[sg->chrom]);
        int64_t start = sg->sig_begin;
        for (int64_t i = sg->sig_begin; i < sg->sig_end; i++) {
            int brk = (i == sg->sig_begin);
            if (!brk) {
                /* a (0,0) element is indistinguishable from the reference's sentinel: whatever
                 * follows it restarts the cluster and the old one is never processed (INDEL:80-82) */
                brk = seg_break(in, sg, i) || (in->a[i - 1] == 0 && in->b[i - 1] == 0);
            }
            if (brk) {
                if (i > sg->sig_begin && !drop) { rc = refine(&w, sg, k, cid, start, i); if (rc) break; }
                cid++;
                start = i;
            }
            if (out->cluster_id) out->cluster_id[i] = cid;
        }
        if (rc == CSV_OK && sg->sig_end > sg->sig_begin && !drop) rc = refine(&w, sg, k, cid, start, sg->sig_end);
    }
    out->n_clusters = cid + 1;
    if (rc == CSV_OK) {
        if (w.n_calls > out->cap_calls || w.n_support > out->cap_support) rc = CSV_E_CAPACITY;
        else rc = genotype_all(&w);
    }
    out->n_calls = w.n_calls;
    out->n_support = w.n_support;
    free(w.idx); free(w.idx2); free(w.tmp); free(w.vals); free(w.vals2); free(w.dev); free(w.sq);
    free(w.ids); free(w.ids2); free(w.pmax);
    free(w.s_start); free(w.s_end); free(w.s_primary); free(w.s_id);
    return rc;
}

/* Stand-alone overlap_cover restatement for the golden tests: for each window (L2, R2 in
 * doubled coordinates) the number of distinct primary read names covering it. */
int csvo_cover_count(const int64_t* r_start, const int64_t* r_end, const uint8_t* r_primary, const int32_t* r_id,
                     int64_t n_reads, const int64_t* L2, const int64_t* R2, int64_t n_win, int32_t* out_count)
{
    for (int64_t k = 0; k < n_win; k++) {
        int32_t* ids = (int32_t*)malloc((size_t)(n_reads + 1) * sizeof(int32_t));
        if (!ids) return CSV_E_NOMEM;
        int64_t n = 0;
        for (int64_t i = 0; i < n_reads; i++)
            if (r_primary[i] == 1 && 2 * r_start[i] <= L2[k] && 2 * r_end[i] >= R2[k]) ids[n++] = r_id[i];
        out_count[k] = (int32_t)count_unique(ids, n);
        free(ids);
    }
    return CSV_OK;
}

/* ------------------------------------------------------------------ CIGAR scan */

/* The CIGAR part of parse_read (main script :606-655) + generate_combine_sigs (:515-575), one read after the other.
 * Same structs as csv_cigar_signatures (include/cutesv_hip.h). */
int csvo_cigar_signatures(const csv_cigar_in* in, csv_cigar_out* out)
{
    int64_t n_i = 0, n_p = 0, n_d = 0;
    for (int64_t r = 0; r < in->n_reads; r++) {
        const int64_t c0 = in->cig_off[r], c1 = in->cig_off[r + 1];
        if (c1 <= c0 || (in->use && !in->use[r])) continue;
        int64_t sig_start = in->ref_start[r];                                  /* :616 */
        int64_t shift = ((in->cigar[c0] & 15u) == 5u) ? -(int64_t)(in->cigar[c0] >> 4) : 0;   /* :621-627 */
        int i_open = 0, d_open = 0;
        int64_t i_last = 0, i_k = 0, d_last = 0, d_k = 0;
        for (int64_t k = c0; k < c1; k++) {
            const int op = (int)(in->cigar[k] & 15u);
            const int64_t oplen = (int64_t)(in->cigar[k] >> 4);
            if (op != 2) shift += oplen;                                       /* :630-631 */
            if (oplen >= in->min_siglength && (op == 1 || op == 2)) {          /* :632 */
                if (op == 2) {
                    /* generate_combine_sigs, DEL branch (:553-575) */
                    if (d_open && sig_start - d_last <= in->merge_del_threshold) {
                        if (d_k < out->cap_sig_del) out->del_len[d_k] += oplen;
                        d_last = sig_start + oplen;
                    } else {
                        const int was_open = d_open;
                        d_k = n_d++;
                        if (d_k < out->cap_sig_del) { out->del_read[d_k] = (int32_t)r; out->del_pos[d_k] = sig_start; out->del_len[d_k] = oplen; }
                        d_open = 1;
                        d_last = was_open ? sig_start : sig_start + oplen;     /* :555 sum(sigs[0]) / :569 append(i[0]) */
                    }
                    sig_start += oplen;                                        /* :635 */
                } else {
                    /* INS branch (:530-552) */
                    const int64_t p = n_p++;
                    if (p < out->cap_piece_ins) { out->piece_qoff[p] = (int32_t)(shift - oplen); out->piece_len[p] = (int32_t)oplen; }
                    if (i_open && sig_start - i_last <= in->merge_ins_threshold) {
                        if (i_k < out->cap_sig_ins) { out->ins_len[i_k] += oplen; out->ins_npiece[i_k] += 1; }
                    } else {
                        i_k = n_i++;
                        if (i_k < out->cap_sig_ins) { out->ins_read[i_k] = (int32_t)r; out->ins_pos[i_k] = sig_start; out->ins_len[i_k] = oplen; out->ins_piece0[i_k] = p; out->ins_npiece[i_k] = 1; }
                        i_open = 1;
                    }
                    i_last = sig_start;
                }
            } else if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) {  /* REFCHANGEOP (:589-602) */
                sig_start += oplen;
            }
        }
    }
    out->n_sig_ins = n_i; out->n_piece_ins = n_p; out->n_sig_del = n_d;
    out->ms_device = 0;
    return (n_i > out->cap_sig_ins || n_p > out->cap_piece_ins || n_d > out->cap_sig_del) ? CSV_E_CAPACITY : CSV_OK;
}

/* ------------------------------------------------------------------ split-read analysis */

/* organize_split_signal (main script :483-513) + analysis_split_read / analysis_inv / analysis_bnd (:50-464), one read
 * after the other, written as the reference is: element [0..5] = read_start, read_end, ref_start, ref_end, chr, strand.
 * Same structs as csv_split_signatures (include/cutesv_hip.h). */
typedef struct { int64_t rs, re, fs, fe; int32_t chr; int32_t st; } sp_seg;     /* st: 0 '+', 1 '-' */
typedef struct { const csv_split_in* in; csv_split_out* out; int64_t n; int32_t read; } sp_sink;

static void sp_put(sp_sink* S, int kind, int32_t chr, int32_t aux, int64_t a, int64_t b, int64_t c, int64_t d)
{
    const int64_t k = S->n++;
    if (k >= S->out->cap) return;
    S->out->kind[k] = (uint8_t)kind; S->out->read[k] = S->read; S->out->chr[k] = chr; S->out->aux[k] = aux;
    S->out->a[k] = a; S->out->b[k] = b; S->out->c[k] = c; S->out->d[k] = d;
}
static sp_seg sp_flip(sp_seg x, int64_t L) { sp_seg y = x; y.rs = L - x.re; y.re = L - x.rs; return y; }    /* [RLength-x[1], RLength-x[0]] + x[2:] */
static double sp_max(int64_t a, int64_t delta) { const double q = (double)delta / 5.0; return (double)a > q ? (double)a : ((double)a == q ? (double)a : q); }

static void sp_inv(sp_sink* S, sp_seg e1, sp_seg e2, int64_t SV)                 /* analysis_inv :50-95 */
{
    if (e1.st == 0) {
        if (e1.fe - e2.fe >= SV && 2 * e2.rs + (e1.fe - e2.fe) >= 2 * e1.re) sp_put(S, 3, e1.chr, 0, e2.fe, e1.fe, 0, 0);
        if (e2.fe - e1.fe >= SV && 2 * e2.rs + (e2.fe - e1.fe) >= 2 * e1.re) sp_put(S, 3, e1.chr, 0, e1.fe, e2.fe, 0, 0);
    } else {
        if (e2.fs - e1.fs >= SV && 2 * e2.rs + (e2.fs - e1.fs) >= 2 * e1.re) sp_put(S, 3, e1.chr, 1, e1.fs, e2.fs, 0, 0);
        if (e1.fs - e2.fs >= SV && 2 * e2.rs + (e1.fs - e2.fs) >= 2 * e1.re) sp_put(S, 3, e1.chr, 1, e2.fs, e1.fs, 0, 0);
    }
}
static void sp_bnd(sp_sink* S, sp_seg e1, sp_seg e2)                            /* analysis_bnd :97-188 */
{
    if (e2.rs - e1.re > 100) return;
    const int lt = e1.chr < e2.chr;
    if (e1.st == 0 && e2.st == 0) { if (lt) sp_put(S, 4, e1.chr, 0, e1.fe, e2.fs, e2.chr, 0); else sp_put(S, 4, e2.chr, 3, e2.fs, e1.fe, e1.chr, 0); }
    else if (e1.st == 0)          { if (lt) sp_put(S, 4, e1.chr, 1, e1.fe, e2.fe, e2.chr, 0); else sp_put(S, 4, e2.chr, 1, e2.fe, e1.fe, e1.chr, 0); }
    else if (e2.st == 0)          { if (lt) sp_put(S, 4, e1.chr, 2, e1.fs, e2.fs, e2.chr, 0); else sp_put(S, 4, e2.chr, 2, e2.fs, e1.fs, e1.chr, 0); }
    else                          { if (lt) sp_put(S, 4, e1.chr, 3, e1.fs, e2.fe, e2.chr, 0); else sp_put(S, 4, e2.chr, 0, e2.fe, e1.fs, e1.chr, 0); }
}
/* the INS / DEL pair of rules on two consecutive same-strand segments (:241-259, :358-376, :382-399, :411-428);
 * need3: the extra `ele_3[2] >= ele_2[3]` test of :361 / :371 */
static void sp_indel(sp_sink* S, sp_seg e1, sp_seg e2, int64_t SV, int64_t Max, int rc, int need3, int ok3)
{
    int64_t delta = e2.rs + e1.fe - e2.fs - e1.re;
    if ((double)(e1.fe - e2.fs) < sp_max(SV, delta) && delta >= SV)
        if ((double)(e2.fs - e1.fe) <= sp_max(100, delta) && (delta <= Max || Max == -1))
            if (!need3 || ok3) sp_put(S, 1, e2.chr, 2 | rc, e2.fs + e1.fe, delta, e1.re + (e2.fs - e1.fe) / 2, e2.rs - (e2.fs - e1.fe) / 2);
    delta = e2.fs - e2.rs + e1.re - e1.fe;
    if ((double)(e1.fe - e2.fs) < sp_max(SV, delta) && delta >= SV)
        if ((double)(e2.rs - e1.re) <= sp_max(100, delta) && (delta <= Max || Max == -1))
            if (!need3 || ok3) sp_put(S, 0, e2.chr, 0, e1.fe, delta, 0, 0);
}

int csvo_split_signatures(const csv_split_in* in, csv_split_out* out)
{
    sp_sink S = {in, out, 0, 0};
    const int64_t SV = in->sv_size, Max = in->max_size;
    int64_t cap_seg = 16;
    sp_seg* SP = (sp_seg*)malloc((size_t)cap_seg * sizeof(sp_seg));
    for (int64_t r = 0; r < in->n_reads; r++) {
        const int64_t e0 = in->ent_off[r], e1o = in->ent_off[r + 1], L = in->read_len[r];
        if (e1o - e0 > cap_seg) { cap_seg = e1o - e0; SP = (sp_seg*)realloc(SP, (size_t)cap_seg * sizeof(sp_seg)); }
        S.read = (int32_t)r;
        /* organize_split_signal */
        int n = 0;
        int min_mapq = in->min_mapq;
        for (int64_t k = e0; k < e1o; k++) {
            sp_seg x; x.chr = in->chr[k]; x.st = in->strand[k];
            if (in->primary[k]) { x.rs = in->c0[k]; x.re = in->c1[k]; x.fs = in->f0[k]; x.fe = in->f1[k]; min_mapq = 0; }      /* :486-488 */
            else {
                if (in->mapq[k] < min_mapq) continue;                                                                             /* :501 */
                if (x.st == 0) { x.rs = in->c0[k]; x.re = L - in->c1[k]; } else { x.rs = in->c1[k]; x.re = L - in->c0[k]; }       /* :503-510 */
                x.fs = in->f0[k]; x.fe = in->f0[k] + in->f1[k];
            }
            /* stable insertion by read_start (sorted(..., key = x[0]), :195) */
            int p = n++;
            while (p > 0 && SP[p - 1].rs > x.rs) { SP[p] = SP[p - 1]; p--; }
            SP[p] = x;
        }
        if (!(n <= in->max_split_parts || in->max_split_parts == -1)) continue;                                                    /* :512 */
        /* analysis_split_read */
        int trigger = 0;
        if (n == 2) {
            sp_seg a1 = SP[0], a2 = SP[1];
            if (a1.chr == a2.chr) {
                if (a1.st != a2.st) sp_inv(&S, a1, a2, SV);
                else {
                    int rc = 0;
                    if (a1.st == 1) { a1 = sp_flip(SP[1], L); a2 = sp_flip(SP[0], L); rc = 1; }
                    if (a1.fe - a2.fs >= SV) {                                                                                     /* :225 */
                        if (a2.rs - a1.re >= a1.fe - a2.fs)
                            sp_put(&S, 1, a2.chr, 2 | rc, a1.fe + a2.fs, a2.rs + a1.fe - a2.fs - a1.re, a1.re + (a2.fs - a1.fe) / 2, a2.rs - (a2.fs - a1.fe) / 2);
                        else sp_put(&S, 2, a2.chr, 0, a2.fs, a1.fe, 0, 0);
                    }
                    sp_indel(&S, a1, a2, SV, Max, rc, 0, 0);
                }
            } else sp_bnd(&S, a1, a2);
        } else {
            for (int a = 0; a + 2 < n; a++) {
                sp_seg a1 = SP[a], a2 = SP[a + 1], a3 = SP[a + 2];
                int a3_none = 0;
                const int last = (n - 3 == a);
                if (a1.chr == a2.chr) {
                    if (a2.chr == a3.chr) {
                        if (a1.st == a3.st && a1.st != a2.st) {                                                                    /* :270 */
                            if (a2.st == 1) {
                                const int64_t d = a3.fs - a1.fe;
                                if (2 * a2.rs + d >= 2 * a1.re && 2 * a3.rs + d >= 2 * a2.re)
                                    if (a2.fs >= a1.fe && a3.fs >= a2.fe) { sp_put(&S, 3, a1.chr, 0, a1.fe, a2.fe, 0, 0); sp_put(&S, 3, a1.chr, 1, a2.fs, a3.fs, 0, 0); }
                            } else {
                                const int64_t d = a1.fs - a3.fe;
                                if (2 * a1.re <= 2 * a2.rs + d && 2 * a3.rs + d >= 2 * a2.re)
                                    if (a2.fs - a3.fe >= -50 && a1.fs - a2.fe >= -50) { sp_put(&S, 3, a1.chr, 0, a3.fe, a2.fe, 0, 0); sp_put(&S, 3, a1.chr, 1, a2.fs, a1.fs, 0, 0); }
                            }
                        }
                        if (last && a1.st != a3.st) {                                                                              /* :316 */
                            if (a2.st == a1.st) sp_inv(&S, a2, a3, SV); else sp_inv(&S, a1, a2, SV);
                        }
                        if (a1.st == a3.st && a1.st == a2.st) {                                                                    /* :333 */
                            int rc = 0;
                            if (a1.st == 1) { a1 = sp_flip(SP[a + 2], L); a2 = sp_flip(SP[a + 1], L); a3 = sp_flip(SP[a], L); rc = 1; }
                            if (a2.fe - a3.fs >= SV && a2.fs < a3.fe) sp_put(&S, 2, a2.chr, 0, a3.fs, a2.fe, 0, 0);
                            if (a == 0 && a1.fe - a2.fs >= SV) sp_put(&S, 2, a2.chr, 0, a2.fs, a1.fe, 0, 0);
                            sp_indel(&S, a1, a2, SV, Max, rc, 1, a3.fs >= a2.fe);
                            if (last) { a1 = a2; a2 = a3; sp_indel(&S, a1, a2, SV, Max, rc, 0, 0); }
                        }
                        if (last && a1.st != a2.st && a2.st == a3.st) { a1 = a2; a2 = a3; a3_none = 1; }                          /* :401 */
                        if (a3_none || (a1.st == a2.st && a2.st != a3.st)) {                                                       /* :405 */
                            int rc = 0;
                            if (a1.st == 1) { a1 = sp_flip(SP[a + 1], L); a2 = sp_flip(SP[a], L); rc = 1; }
                            sp_indel(&S, a1, a2, SV, Max, rc, 0, 0);
                        }
                    }
                } else {
                    trigger = 1;
                    sp_bnd(&S, a1, a2);
                    if (last && a2.chr != a3.chr) sp_bnd(&S, a2, a3);
                }
            }
        }
        if (n >= 3 && trigger && SP[0].chr == SP[n - 1].chr && SP[0].st == SP[n - 1].st) {                                         /* :439 */
            sp_seg a1, a2; int rc = 0;
            if (SP[0].st == 0) { a1 = SP[0]; a2 = SP[n - 1]; } else { a1 = sp_flip(SP[n - 1], L); a2 = sp_flip(SP[0], L); rc = 1; }
            const int64_t dis_ref = a2.fs - a1.fe, dis_read = a2.rs - a1.re, dl = dis_read - dis_ref;
            const int64_t ad = dis_ref < 0 ? -dis_ref : dis_ref;
            if ((double)ad < sp_max(SV, dl) && dl >= SV && (dl <= Max || Max == -1))
                sp_put(&S, 1, a2.chr, rc, a2.fs < a1.fe ? a2.fs : a1.fe, dl, a1.re + dis_ref / 2, a2.rs - dis_ref / 2);
            if (dis_ref <= -SV) sp_put(&S, 2, a2.chr, 0, a2.fs, a1.fe, 0, 0);
        }
    }
    free(SP);
    out->n = S.n; out->ms_device = 0;
    return S.n > out->cap ? CSV_E_CAPACITY : CSV_OK;
}
