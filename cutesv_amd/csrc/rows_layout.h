// rows_layout.h — THE statement of the reference's row layouts on the csv_batch_out structure of arrays, written once
// and compiled against two sinks:
//   * the text sink of csv_rows_emit (rows_emit.cpp, C ABI: one blob, fields '\t', rows '\n'), and
//   * the CPython sink of cutesv_amd/_rows_native (rows_py.cpp), which creates the list-of-str objects directly.
// Layouts:
//   DEL  cuteSV_resolveINDEL.py:207-219 (no genotype) / :464-478 (genotype)      13 fields
//   INS  cuteSV_resolveINDEL.py:419-432 / :464-478                               14 fields
//   DUP  cuteSV_resolveDUP.py:121-131 / :170-180                                 11 fields
//   INV  cuteSV_resolveINV.py:145-156 / :240-251                                 12 fields
//   TRA  cuteSV_resolveTRA.py:171-182 (genotype fields from call_gt, :258-309)   12 fields
// All numeric fields are decimal text; read names are joined by ','.
//
// Sink interface: row_begin(n_fields) / row_end(); field_begin(exact_len_or_0) ... field_end(); inside a field raw(),
// ch(), num(), acgt(), or reserve(n) -> pointer for n bytes (null: count only) followed by commit(n).  A field announced with exact_len > 0 holds exactly that many bytes (the CPython
// sink writes it straight into the str object); short fields pass 0.
#pragma once
#include <stdint.h>
#include <string.h>

#include "../../include/cutesv_hip.h"
#include "soa_access.h"

namespace csv_rows {

static const char kDigits2[201] =
    "0001020304050607080910111213141516171819202122232425262728293031323334353637383940414243444546474849"
    "5051525354555657585960616263646566676869707172737475767778798081828384858687888990919293949596979899";

// decimal text of v into the END of b[24]; returns the first used index
inline int fmt_i64(int64_t v, char (&b)[24])
{
    int k = 24;
    const bool neg = v < 0;
    uint64_t u = neg ? (uint64_t)0 - (uint64_t)v : (uint64_t)v;
    while (u >= 100) { const unsigned r = (unsigned)(u % 100); u /= 100; k -= 2; memcpy(b + k, kDigits2 + 2 * r, 2); }
    if (u >= 10) { k -= 2; memcpy(b + k, kDigits2 + 2 * u, 2); } else b[--k] = (char)('0' + u);
    if (neg) b[--k] = '-';
    return k;
}
inline int dec_len(uint32_t u) { int n = 1; while (u >= 10) { u /= 10; n++; } return n; }
// "%0<width>d" of a non-negative 32-bit value at d; returns the end
inline char* put_padded_u32(char* d, uint32_t u, int width)
{
    char b[12];
    int k = 12;
    while (u >= 100) { const unsigned r = u % 100; u /= 100; k -= 2; memcpy(b + k, kDigits2 + 2 * r, 2); }
    if (u >= 10) { k -= 2; memcpy(b + k, kDigits2 + 2 * u, 2); } else b[--k] = (char)('0' + u);
    const int n = 12 - k;
    if (n < width) { memset(d, '0', (size_t)(width - n)); d += width - n; }
    memcpy(d, b + k, (size_t)n);
    return d + n;
}

// rows of the calls [c_begin, c_end) (c_end < 0: all)
template <class Sink> int layout(const csv_rows_in* in, Sink& S, int64_t c_begin = 0, int64_t c_end = -1)
{
    if (!in || !in->res || (in->n_seg > 0 && !in->seg) || !in->chrom_name) return CSV_E_INVALID;
    const csv_batch_out& R = *in->res;
    if (!csv_soa::has_support_list(R)) return CSV_E_INVALID;      // (a row's read list: not a CSV_OUT_NO_SUPPORT_LIST result)
    const int64_t nc = (c_end < 0 || c_end > R.n_calls) ? R.n_calls : c_end;
    static const char* kTraAlt[4][2] = {{"N[", "["}, {"N]", "]"}, {"[", "[N"}, {"]", "]N"}};     // cuteSV_resolveTRA.py:142-153
    static const char* kType[4] = {"DEL", "INS", "DUP", "INV"};
    const int name_prefix_len = in->name_prefix ? (int)strlen(in->name_prefix) : 0;
    for (int64_t c = c_begin < 0 ? 0 : c_begin; c < nc; c++) {
        const int k = R.call_seg[c];
        if (k < 0 || k >= in->n_seg) return CSV_E_INVALID;
        const csv_segment& sg = in->seg[k];
        if (sg.chrom < 0 || sg.chrom >= in->n_chrom) return CSV_E_INVALID;
        const int t = sg.svtype;
        const char* chrom = in->chrom_name[sg.chrom];
        const int64_t bp1 = csv_soa::bp1(R, c), bp2 = csv_soa::bp2(R, c);
        // genotype fields: str(DR), GT, PL, GQ, QUAL (cuteSV_resolveINDEL.py:471-475); '.' fields when the task was not
        // genotyped or count_coverage gave up (cuteSV_resolveTRA.py:276-281)
        const bool gt_on = sg.genotype != 0 && csv_soa::gl_idx(R, c) >= 0;
        const char* g[4] = {"./.", ".,.,.", ".", "."};
        int64_t gn[4] = {3, 5, 1, 1};
        if (gt_on) {                                         // table row "GT \t PL \t GQ \t QUAL"
            if (!in->gl_blob || !in->gl_off || csv_soa::gl_idx(R, c) >= CSV_GL_TABLE_SIZE) return CSV_E_INVALID;
            const char* p = in->gl_blob + in->gl_off[csv_soa::gl_idx(R, c)];
            const char* e = in->gl_blob + in->gl_off[csv_soa::gl_idx(R, c) + 1];
            for (int q = 0; q < 4; q++) {
                const char* tb = q < 3 ? (const char*)memchr(p, '\t', (size_t)(e - p)) : e;
                if (!tb) return CSV_E_INVALID;
                g[q] = p; gn[q] = tb - p; p = tb + 1;
            }
        }
        auto f_str = [&](const char* s, int64_t n) { S.field_begin(0); S.raw(s, n); S.field_end(); };
        auto f_num = [&](int64_t v) { S.field_begin(0); S.num(v); S.field_end(); };
        auto f_ci = [&](int32_t v) { S.field_begin(0); S.ch('-'); S.num(v); S.ch(','); S.num(v); S.field_end(); };   // cal_CIPOS, GT:60
        auto f_dr = [&]() { S.field_begin(0); if (gt_on) S.num(csv_soa::dr(R, c)); else S.ch('.'); S.field_end(); };
        auto f_gl = [&](int q) { f_str(g[q], gn[q]); };
        auto f_reads = [&]() -> bool {
            const int64_t s0 = R.support_off[c], s1 = R.support_off[c + 1];
            int64_t total = s1 > s0 ? s1 - s0 - 1 : 0;      // commas
            if (in->name_blob) {
                for (int64_t i = s0; i < s1; i++) {
                    const int32_t id = in->read_id[R.support_sig ? R.support_sig[i] : (int64_t)R.support_sig32[i]];
                    if (id < 0 || id >= in->n_names) return false;
                    total += in->name_off[id + 1] - in->name_off[id];
                }
            } else {
                for (int64_t i = s0; i < s1; i++) {
                    const int32_t id = in->read_id[R.support_sig ? R.support_sig[i] : (int64_t)R.support_sig32[i]];
                    if (id < 0) return false;
                    const int d = id < 1000000000 && in->name_width >= 9 ? 0 : dec_len((uint32_t)id);
                    total += name_prefix_len + (d > in->name_width ? d : in->name_width);
                }
            }
            S.field_begin(total > 0 ? total : 0);
            char* d = S.reserve(total);                      // the whole list is written through one pointer (null: the sink only counts)
            if (d) {
                for (int64_t i = s0; i < s1; i++) {
                    if (i > s0) *d++ = ',';
                    const int32_t id = in->read_id[R.support_sig ? R.support_sig[i] : (int64_t)R.support_sig32[i]];
                    if (in->name_blob) { const int64_t n = in->name_off[id + 1] - in->name_off[id]; memcpy(d, in->name_blob + in->name_off[id], (size_t)n); d += n; }
                    else { if (name_prefix_len) { memcpy(d, in->name_prefix, (size_t)name_prefix_len); d += name_prefix_len; } d = put_padded_u32(d, (uint32_t)id, in->name_width); }
                }
            }
            S.commit(total);
            S.field_end();
            return true;
        };
        const int nf = t == CSV_DEL ? 13 : t == CSV_INS ? 14 : t == CSV_DUP ? 11 : 12;
        S.row_begin(nf);
        f_str(chrom, (int64_t)strlen(chrom));
        if (t == CSV_DEL || t == CSV_INS) {
            f_str(kType[t], 3); f_num(bp1); f_num(t == CSV_DEL ? -bp2 : bp2); f_num(R.support[c]); f_ci(csv_soa::cipos(R, c)); f_ci(csv_soa::cilen(R, c));
            f_dr(); f_gl(0); f_gl(1); f_gl(2); f_gl(3);
            if (!f_reads()) return CSV_E_INVALID;
            if (t == CSV_INS) {                              // the inserted sequence sliced to SVLEN (INDEL:402)
                const int64_t sig = csv_soa::seq_pick(R, c);
                if (sig < 0) return CSV_E_INVALID;
                int64_t n = in->ins_blob ? in->ins_off[sig + 1] - in->ins_off[sig] : (in->aux ? in->aux[sig] : 0);
                if (bp2 < n) n = bp2 < 0 ? 0 : bp2;
                S.field_begin(n);
                if (in->ins_blob) S.raw(in->ins_blob + in->ins_off[sig], n); else S.acgt(n);
                S.field_end();
            }
        } else if (t == CSV_DUP) {
            f_str("DUP", 3); f_num(bp1); f_num(bp2 - bp1); f_num(R.support[c]);
            f_dr(); f_gl(0); f_gl(1); f_gl(2); f_gl(3);
            if (!f_reads()) return CSV_E_INVALID;
        } else if (t == CSV_INV) {
            if (csv_soa::call_aux(R, c) < 0 || csv_soa::call_aux(R, c) >= in->n_strand) return CSV_E_INVALID;
            const char* sn = in->strand_name[csv_soa::call_aux(R, c)];
            f_str("INV", 3); f_num(bp1); f_num(bp2 - bp1); f_num(R.support[c]);
            f_dr(); f_gl(0); f_str(sn, (int64_t)strlen(sn)); f_gl(1); f_gl(2); f_gl(3);
            if (!f_reads()) return CSV_E_INVALID;
        } else {
            const int code = csv_soa::call_aux(R, c) & 7, c2 = csv_soa::call_aux(R, c) >> 3;
            if (code > 3 || c2 < 0 || c2 >= in->n_chrom) return CSV_E_INVALID;
            const char* chr2 = in->chrom_name[c2];
            const int64_t n2 = (int64_t)strlen(chr2);
            const int64_t mate = bp2 + ((code == 0 || code == 2) ? 1 : 0);          // types A / C, cuteSV_resolveTRA.py:140
            S.field_begin(0);
            S.raw(kTraAlt[code][0], (int64_t)strlen(kTraAlt[code][0])); S.raw(chr2, n2); S.ch(':'); S.num(mate);
            S.raw(kTraAlt[code][1], (int64_t)strlen(kTraAlt[code][1]));
            S.field_end();
            f_num(bp1); f_str(chr2, n2); f_num(bp2); f_num(R.support[c]);
            f_dr(); f_gl(0); f_gl(1); f_gl(2); f_gl(3);
            if (!f_reads()) return CSV_E_INVALID;
        }
        if (!S.row_end()) return CSV_E_NOMEM;
    }
    return CSV_OK;
}

}  // namespace csv_rows
