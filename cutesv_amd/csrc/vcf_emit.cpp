// vcf_emit.cpp — host side of the stage's OUTPUT boundary: calls (structure of arrays) -> VCF body text.
// Restates generate_output (cuteSV_genotype.py:242-467) and the SVID numbering of main_ctrl
// (cuteSV main script :1208-1237) on the csv_batch_out layout; see include/cutesv_hip.h.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/cutesv_hip.h"

namespace {

struct Sink {
    char*   out;
    int64_t cap, n;
    void put(const char* s, size_t len)
    {
        if (n + (int64_t)len <= cap) memcpy(out + n, s, len);
        n += (int64_t)len;
    }
    void put(const char* s) { put(s, strlen(s)); }
    void put(const std::string& s) { put(s.data(), s.size()); }
    void put(char c) { put(&c, 1); }
    void num(long long v)
    {
        char b[32];
        const int k = snprintf(b, sizeof b, "%lld", v);
        put(b, (size_t)k);
    }
};

// trans_table = str.maketrans('RYSWKMBDHV', 'ACCAGACAAA')   (cuteSV_genotype.py:262)
inline char iupac(char c)
{
    switch (c) {
    case 'R': return 'A'; case 'Y': return 'C'; case 'S': return 'C'; case 'W': return 'A'; case 'K': return 'G';
    case 'M': return 'A'; case 'B': return 'C'; case 'D': return 'A'; case 'H': return 'A'; case 'V': return 'A';
    default: return c;
    }
}

// str(round(dv / (dv + dr), 4))  (cuteSV_genotype.py:284): correctly rounded to 4 decimals, then Python's
// shortest repr, which for such a value is the 4-decimal string without trailing zeros (one decimal kept).
void put_af(Sink& o, long long dv, long long dr)
{
    char b[32];
    int k = snprintf(b, sizeof b, "%.4f", (double)dv / (double)(dv + dr));
    while (k > 0 && b[k - 1] == '0' && b[k - 2] != '.') k--;
    o.put(b, (size_t)k);
}

struct GlRow { const char* gt; size_t gt_n; const char* pl; size_t pl_n; const char* gq; size_t gq_n; const char* qual; size_t qual_n; };

bool split_gl(const char* s, GlRow& r)
{
    const char* t1 = strchr(s, '\t'); if (!t1) return false;
    const char* t2 = strchr(t1 + 1, '\t'); if (!t2) return false;
    const char* t3 = strchr(t2 + 1, '\t'); if (!t3) return false;
    r.gt = s; r.gt_n = (size_t)(t1 - s);
    r.pl = t1 + 1; r.pl_n = (size_t)(t2 - t1 - 1);
    r.gq = t2 + 1; r.gq_n = (size_t)(t3 - t2 - 1);
    r.qual = t3 + 1; r.qual_n = strlen(t3 + 1);
    return true;
}

}  // namespace

extern "C" int csv_vcf_emit(const csv_vcf_in* in, char* out, int64_t cap, int64_t* n_written, int64_t* svid)
{
    if (!in || !in->res || !n_written || !svid) return CSV_E_INVALID;
    const csv_batch_out& R = *in->res;
    const int64_t nc = R.n_calls;
    Sink o{out, out ? cap : 0, 0};
    enum { ID_INS = 0, ID_DEL = 1, ID_BND = 2, ID_DUP = 3, ID_INV = 4 };

    // calls per chromosome in the order main_ctrl concatenates task results (the call order of the batch),
    // then generate_output's stable sort by int(row[2])  (cuteSV_genotype.py:252)
    std::vector<std::vector<int64_t>> per(in->n_chrom);
    for (int64_t c = 0; c < nc; c++) {
        const int k = R.call_seg[c];
        if (k < 0 || k >= in->n_seg) return CSV_E_INVALID;
        const int ch = in->seg[k].chrom;
        if (ch < 0 || ch >= in->n_chrom) return CSV_E_INVALID;
        per[ch].push_back(c);
    }
    std::vector<int> order(in->n_chrom);
    for (int i = 0; i < in->n_chrom; i++) order[i] = i;
    std::sort(order.begin(), order.end(), [&](int x, int y) { return in->chrom_rank[x] < in->chrom_rank[y]; });

    static const char* kFormat = "GT:DR:DV:PL:GQ";
    for (int ch : order) {
        std::vector<int64_t>& v = per[ch];
        if (v.empty()) continue;
        std::stable_sort(v.begin(), v.end(), [&](int64_t x, int64_t y) { return R.bp1[x] < R.bp1[y]; });
        const char* seq = in->chrom_seq ? in->chrom_seq[ch] : nullptr;
        const int64_t slen = in->chrom_len ? in->chrom_len[ch] : 0;
        const char* cname = in->chrom_name[ch];
        auto base = [&](int64_t i, char& c) -> bool { if (!seq || i < 0 || i >= slen) return false; c = seq[i]; return true; };
        for (int64_t c : v) {
            const csv_segment& sg = in->seg[R.call_seg[c]];
            const int type = sg.svtype;
            // TRA calls whose count_coverage gave up carry the '.' fields too (cuteSV_resolveTRA.py:276-281)
            const bool gt_on = sg.genotype != 0 && R.gl_idx[c] >= 0;
            // genotype strings of the row ('.' fields when the task was not genotyped)
            GlRow g{"./.", 3, ".,.,.", 5, ".", 1, ".", 1};
            if (gt_on) {
                const int32_t key = R.gl_idx[c];
                const int32_t* it = std::lower_bound(in->gl_key, in->gl_key + in->n_gl, key);
                if (it == in->gl_key + in->n_gl || *it != key || !split_gl(in->gl_str[it - in->gl_key], g)) return CSV_E_INVALID;
            }
            const bool imprecise = g.gt_n == 3 && memcmp(g.gt, "0/0", 3) == 0;
            // filter label (cuteSV_genotype.py:289-292)
            const char* filt = "PASS";
            if (!(g.qual_n == 1 && g.qual[0] == '.')) filt = strtod(std::string(g.qual, g.qual_n).c_str(), nullptr) >= 5.0 ? "PASS" : "q5";
            const long long pos = R.bp1[c], re = R.support[c];
            auto put_rnames = [&]() {
                if (in->report_readid) { o.put(";RNAMES="); if (in->rnames) o.put(in->rnames + in->rnames_off[c], (size_t)(in->rnames_off[c + 1] - in->rnames_off[c])); }
            };
            auto put_af_field = [&](bool numeric) {
                if (!in->genotype) return;
                o.put(";AF=");
                if (numeric) put_af(o, re, R.dr[c]); else o.put('.');
            };
            auto put_tail = [&]() {           // QUAL FILTER INFO are written by the caller; this is FORMAT + sample
                o.put('\t'); o.put(kFormat); o.put('\t');
                o.put(g.gt, g.gt_n); o.put(':');
                if (gt_on) o.num(R.dr[c]); else o.put('.');
                o.put(':'); o.num(re); o.put(':'); o.put(g.pl, g.pl_n); o.put(':'); o.put(g.gq, g.gq_n); o.put('\n');
            };
            if (type == CSV_DEL || type == CSV_INS) {
                const long long len = R.bp2[c];                                         // |SVLEN|
                if (len > in->max_size && in->max_size != -1) continue;                   // :265-266
                if (len < in->min_size) continue;                                       // :267-268
                const long long end = type == CSV_INS ? pos : pos + len;
                const long long r0 = pos - 1 > 0 ? pos - 1 : 0;
                o.put(cname); o.put('\t'); o.num(pos); o.put('\t');
                o.put(type == CSV_INS ? "cuteSV.INS." : "cuteSV.DEL."); o.num(svid[type == CSV_INS ? ID_INS : ID_DEL]++); o.put('\t');
                if (in->ignore_sequence) {
                    o.put("N\t"); o.put(type == CSV_INS ? "<INS>" : "<DEL>");
                } else if (type == CSV_INS) {
                    char b0;
                    if (!base(r0, b0)) return CSV_E_INVALID;                             // ref_chrom[max(pos-1, 0)] (:297)
                    o.put(iupac(b0)); o.put('\t'); o.put(b0);
                    if (in->ins_alt) o.put(in->ins_alt + in->ins_alt_off[c], (size_t)(in->ins_alt_off[c + 1] - in->ins_alt_off[c]));
                } else {
                    long long r1 = pos + len;                                           // slice end, clamped like Python
                    if (r1 > slen) r1 = slen;
                    for (long long i = r0; i < r1; i++) o.put(iupac(seq[i]));
                    o.put('\t');
                    char b0;
                    if (!base(r0, b0)) return CSV_E_INVALID;
                    o.put(b0);
                }
                o.put('\t'); o.put(g.qual, g.qual_n); o.put('\t'); o.put(filt); o.put('\t');
                o.put(imprecise ? "IMPRECISE" : "PRECISE");
                o.put(type == CSV_INS ? ";SVTYPE=INS;SVLEN=" : ";SVTYPE=DEL;SVLEN=");
                o.num(type == CSV_INS ? len : -len);
                o.put(";END="); o.num(end);
                o.put(";CIPOS=-"); o.num(R.cipos[c]); o.put(','); o.num(R.cipos[c]);
                o.put(";CILEN=-"); o.num(R.cilen[c]); o.put(','); o.num(R.cilen[c]);
                o.put(";RE="); o.num(re);
                put_rnames();
                put_af_field(gt_on);
                if (type == CSV_DEL) o.put(";STRAND=+-");
                put_tail();
            } else if (type == CSV_DUP) {
                const long long len = R.bp2[c] - R.bp1[c];
                if (llabs(len) > in->max_size && in->max_size != -1) continue;            // :315-316
                char b0;
                if (!base(pos, b0)) return CSV_E_INVALID;                                // ref_chrom[int(i[2])] (:334)
                o.put(cname); o.put('\t'); o.num(pos + 1); o.put('\t');
                o.put("cuteSV.DUP."); o.num(svid[ID_DUP]++); o.put('\t');
                o.put(iupac(b0)); o.put("\t<DUP>\t");
                o.put(g.qual, g.qual_n); o.put('\t'); o.put(filt); o.put('\t');
                o.put(imprecise ? "IMPRECISE" : "PRECISE");
                o.put(";SVTYPE=DUP;SVLEN="); o.num(len);
                o.put(";END="); o.num(pos + 1 + llabs(len));
                o.put(";RE="); o.num(re); o.put(";STRAND=-+");
                put_rnames();
                put_af_field(gt_on);
                put_tail();
            } else if (type == CSV_INV) {
                const long long len = R.bp2[c] - R.bp1[c];
                if (llabs(len) > in->max_size && in->max_size != -1) continue;            // :351-352
                const char* strand = in->strand_name[R.call_aux[c]];
                const bool pp = strcmp(strand, "++") == 0;                              // :360-365
                const long long pinv = pp ? pos : pos + 1;
                const long long ridx = pp ? (pos - 1 > 0 ? pos - 1 : 0) : pos;
                char b0;
                if (!base(ridx, b0)) return CSV_E_INVALID;
                o.put(cname); o.put('\t'); o.num(pinv); o.put('\t');
                o.put("cuteSV.INV."); o.num(svid[ID_INV]++); o.put('\t');
                o.put(iupac(b0)); o.put("\t<INV>\t");
                o.put(g.qual, g.qual_n); o.put('\t'); o.put(filt); o.put('\t');
                o.put(imprecise ? "IMPRECISE" : "PRECISE");
                o.put(";SVTYPE=INV;SVLEN="); o.num(len);
                o.put(";END="); o.num(pinv + llabs(len));
                o.put(";RE="); o.num(re); o.put(";STRAND="); o.put(strand);
                put_rnames();
                put_af_field(gt_on);
                put_tail();
            } else {                                                                    // BND (:400-458)
                const int code = R.call_aux[c] & 7;
                const char* chr2 = in->chrom_name[R.call_aux[c] >> 3];
                const long long mate = R.bp2[c] + ((code == 0 || code == 2) ? 1 : 0);    // TRA:140
                const bool nfirst = code < 2;                                           // ALT starts with 'N' (types A/B)
                const long long pbnd = nfirst ? pos : pos + 1;
                char b0 = 'N';
                if (nfirst) { char t; if (base(pos - 1 > 0 ? pos - 1 : 0, t)) b0 = t; }  // try/except -> 'N' (:431-434)
                else { char t; if (base(pos, t)) b0 = t; }
                o.put(cname); o.put('\t'); o.num(pbnd); o.put('\t');
                o.put("cuteSV.BND."); o.num(svid[ID_BND]++); o.put('\t');
                o.put(iupac(b0)); o.put('\t');
                char mt[64];
                const int mk = snprintf(mt, sizeof mt, ":%lld", mate);
                if (code == 0) { o.put(b0); o.put('['); o.put(chr2); o.put(mt, (size_t)mk); o.put('['); }
                else if (code == 1) { o.put(b0); o.put(']'); o.put(chr2); o.put(mt, (size_t)mk); o.put(']'); }
                else if (code == 2) { o.put('['); o.put(chr2); o.put(mt, (size_t)mk); o.put('['); o.put(b0); }
                else { o.put(']'); o.put(chr2); o.put(mt, (size_t)mk); o.put(']'); o.put(b0); }
                o.put('\t'); o.put(g.qual, g.qual_n); o.put('\t'); o.put(filt); o.put('\t');
                o.put(imprecise ? "IMPRECISE" : "PRECISE");
                o.put(";SVTYPE=BND;RE="); o.num(re);
                put_rnames();
                put_af_field(gt_on);                                                    // DR '.' -> AF=. (:412-414)
                put_tail();
            }
        }
    }
    *n_written = o.n;
    return (out && o.n <= cap) ? CSV_OK : CSV_E_CAPACITY;
}
