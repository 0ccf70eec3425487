// vcf_emit.cpp — host side of the stage's OUTPUT boundary: calls (structure of arrays) -> VCF body text.
// Restates generate_output (cuteSV_genotype.py:242-467) and the SVID numbering of main_ctrl
// (cuteSV main script :1208-1237) on the csv_batch_out layout; see include/cutesv_hip.h.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <unistd.h>
#include <functional>
#include <string>
#include <thread>
#include <vector>

#include "../../include/cutesv_hip.h"
#include "soa_access.h"

namespace {

// text of the records a worker thread formats: a growable buffer that belongs to the THREAD and keeps its capacity from call to
// call (a fresh std::string per slice meant ~30 MB of new pages per emission once the records carry their REF sequences: page
// faults under the process's memory-map lock, on 16 threads at once, were most of the emitter's time)
struct Sink {
    std::string& buf;
    void put(const char* s, size_t len) { buf.append(s, len); }
    void put(const char* s) { buf.append(s); }
    void put(const std::string& s) { buf.append(s); }
    void put(char c) { buf.push_back(c); }
    void num(long long v)                                  // decimal text (snprintf was a third of the emitter's time)
    {
        char b[24];
        int k = 24;
        unsigned long long u = v < 0 ? 0ull - (unsigned long long)v : (unsigned long long)v;
        do { b[--k] = (char)('0' + u % 10); u /= 10; } while (u);
        if (v < 0) b[--k] = '-';
        buf.append(b + k, (size_t)(24 - k));
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

struct IupacTable { char t[256]; IupacTable() { for (int c = 0; c < 256; c++) t[c] = iupac((char)c); } };
const IupacTable kIupac;

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

// Records are formatted by worker threads, one slice of a chromosome's sorted calls at a time; the SVID counters a slice
// starts with are the numbers of records of each type EMITTED before it (generate_output's size filters decide that), found
// by a first pass over the slices.  Order of the text: chromosomes by name rank, records by position, as the reference writes.
namespace {
constexpr int64_t VCF_SLICE = 1024;                    // calls per work item
enum { ID_INS = 0, ID_DEL = 1, ID_BND = 2, ID_DUP = 3, ID_INV = 4 };

// size filters of generate_output (cuteSV_genotype.py:265-268, 315-316, 351-352): is the call written, and under which counter
inline int emitted_id(const csv_vcf_in* in, const csv_batch_out& R, int64_t c)
{
    const int type = in->seg[R.call_seg[c]].svtype;
    if (type == CSV_DEL || type == CSV_INS) {
        const long long len = csv_soa::bp2(R, c);
        if ((len > in->max_size && in->max_size != -1) || len < in->min_size) return -1;
        return type == CSV_INS ? ID_INS : ID_DEL;
    }
    if (type == CSV_DUP || type == CSV_INV) {
        const long long len = csv_soa::bp2(R, c) - csv_soa::bp1(R, c);
        if (llabs(len) > in->max_size && in->max_size != -1) return -1;
        return type == CSV_DUP ? ID_DUP : ID_INV;
    }
    return ID_BND;
}

int emit_slice(const csv_vcf_in* in, const int64_t* v, int64_t nv, int ch, int64_t* svid, Sink& o);

// A team of worker threads that lives as long as the process (thread creation is ~30 us: 15 fresh threads per call were a third
// of a 1.4 ms emission).  Between calls the workers sleep on a condition variable; inside a call (wake() .. park()) they spin
// with yield() between the three parallel loops.  run(n, f): f(i) for i in [0, n) dealt dynamically; returns when all are done.
class Team {
public:
    explicit Team(int nthreads) : n_(nthreads < 1 ? 1 : nthreads)
    {
        for (int t = 1; t < n_; t++) th_.emplace_back([this] { loop(); });
    }
    ~Team()
    {
        { std::lock_guard<std::mutex> g(m_); quit_.store(true); active_.store(true); gen_.fetch_add(1); }
        cv_.notify_all();
        for (auto& t : th_) t.join();
    }
    int size() const { return n_; }
    void wake() { { std::lock_guard<std::mutex> g(m_); active_.store(true); } cv_.notify_all(); }
    void park() { active_.store(false); }
    template <class F> void run(int64_t n, F&& f)
    {
        if (n <= 0) return;
        std::function<void(int64_t)> fn = f;
        fn_ = &fn; total_ = n; next_.store(0); done_.store(0);
        gen_.fetch_add(1);                                  // release the workers
        work();
        while (done_.load() < n_) std::this_thread::yield();
    }
private:
    void work() { for (int64_t i; (i = next_.fetch_add(1)) < total_;) (*fn_)(i); done_.fetch_add(1); }
    void loop()
    {
        for (int seen = 0;;) {
            int g;
            while ((g = gen_.load()) == seen) {
                if (!active_.load()) {
                    std::unique_lock<std::mutex> lk(m_);
                    cv_.wait(lk, [&] { return active_.load() || quit_.load(); });
                } else std::this_thread::yield();
            }
            seen = g;
            if (quit_.load()) return;
            work();
        }
    }
    int n_;
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_;
    std::atomic<int> gen_{0}, done_{0};
    std::atomic<bool> quit_{false}, active_{false};
    std::atomic<int64_t> next_{0};
    int64_t total_ = 0;
    std::function<void(int64_t)>* fn_ = nullptr;
};

// the process's team (a forked child starts its own: threads do not survive fork); one emission at a time uses it
std::mutex g_team_mu;
Team* g_team = nullptr;
int g_team_pid = 0, g_team_n = 0;
struct TeamLease {
    Team* t = nullptr; bool own = false; std::unique_lock<std::mutex> lk;
    explicit TeamLease(int nthreads)
    {
        if (nthreads <= 1) { t = new Team(1); own = true; return; }
        lk = std::unique_lock<std::mutex>(g_team_mu, std::try_to_lock);
        if (!lk.owns_lock()) { t = new Team(nthreads); own = true; t->wake(); return; }     // (another thread is emitting: a team of its own)
        const int pid = (int)getpid();
        if (!g_team || g_team_pid != pid || g_team_n != nthreads) {
            if (g_team && g_team_pid == pid) delete g_team;      // (after a fork the parent's object is just forgotten: its threads are not ours)
            g_team = new Team(nthreads); g_team_pid = pid; g_team_n = nthreads;
        }
        t = g_team; t->wake();
    }
    ~TeamLease() { if (own) delete t; else if (t) t->park(); }
};
}  // namespace

extern "C" int csv_vcf_emit(const csv_vcf_in* in, char* out, int64_t cap, int64_t* n_written, int64_t* svid)
{
    if (!in || !in->res || !n_written || !svid) return CSV_E_INVALID;
    const csv_batch_out& R = *in->res;
    const int64_t nc = R.n_calls;
    int nthreads = (int)std::thread::hardware_concurrency();
    if (const char* e = getenv("CSV_VCF_THREADS")) nthreads = atoi(e);
    if (nthreads > 16) nthreads = 16;
    if (nthreads < 1 || nc < 2 * VCF_SLICE) nthreads = 1;
    TeamLease lease(nthreads);
    Team& team = *lease.t;

    // calls per chromosome in the order main_ctrl concatenates task results (the call order of the batch): counting sort
    std::vector<int64_t> coff((size_t)in->n_chrom + 1, 0), idx((size_t)nc);
    for (int64_t c = 0; c < nc; c++) {
        const int k = R.call_seg[c];
        if (k < 0 || k >= in->n_seg) return CSV_E_INVALID;
        const int ch = in->seg[k].chrom;
        if (ch < 0 || ch >= in->n_chrom) return CSV_E_INVALID;
        coff[(size_t)ch + 1]++;
    }
    for (int i = 0; i < in->n_chrom; i++) coff[(size_t)i + 1] += coff[(size_t)i];
    {
        std::vector<int64_t> fill(coff.begin(), coff.end() - 1);
        for (int64_t c = 0; c < nc; c++) idx[(size_t)fill[(size_t)in->seg[R.call_seg[c]].chrom]++] = c;
    }
    std::vector<int> order(in->n_chrom);
    for (int i = 0; i < in->n_chrom; i++) order[i] = i;
    std::sort(order.begin(), order.end(), [&](int x, int y) { return in->chrom_rank[x] < in->chrom_rank[y]; });
    // generate_output's stable sort by int(row[2]) per chromosome (cuteSV_genotype.py:252)
    team.run(in->n_chrom, [&](int64_t ch) {
        std::stable_sort(idx.begin() + coff[(size_t)ch], idx.begin() + coff[(size_t)ch + 1], [&](int64_t x, int64_t y) { return csv_soa::bp1(R, x) < csv_soa::bp1(R, y); });
    });
    // slices in emission order
    struct Slice { int ch; int64_t lo, hi; int64_t cnt[5]; int64_t start[5]; int rc; const std::string* buf; size_t off, len; };
    static std::atomic<uint64_t> g_call{0};
    const uint64_t call_id = g_call.fetch_add(1) + 1;
    std::vector<Slice> sl;
    for (int ch : order)
        for (int64_t lo = coff[(size_t)ch]; lo < coff[(size_t)ch + 1]; lo += VCF_SLICE) {
            Slice x{};
            x.ch = ch; x.lo = lo; x.hi = lo + VCF_SLICE < coff[(size_t)ch + 1] ? lo + VCF_SLICE : coff[(size_t)ch + 1];
            sl.push_back(std::move(x));
        }
    // records per type every slice emits -> the counters it starts with (main script :1208-1237: one counter per type over the
    // whole file)
    team.run((int64_t)sl.size(), [&](int64_t i) {
        Slice& x = sl[(size_t)i];
        for (int64_t q = x.lo; q < x.hi; q++) { const int id = emitted_id(in, R, idx[(size_t)q]); if (id >= 0) x.cnt[id]++; }
    });
    int64_t run[5] = {svid[0], svid[1], svid[2], svid[3], svid[4]};
    for (Slice& x : sl) for (int t = 0; t < 5; t++) { x.start[t] = run[t]; run[t] += x.cnt[t]; }
    team.run((int64_t)sl.size(), [&](int64_t i) {
        Slice& x = sl[(size_t)i];
        static thread_local std::string tl_buf;
        static thread_local uint64_t tl_call = 0;
        if (tl_call != call_id) { tl_buf.clear(); tl_call = call_id; }         // (capacity kept)
        Sink o{tl_buf};
        x.buf = &tl_buf; x.off = tl_buf.size();
        int64_t sv[5] = {x.start[0], x.start[1], x.start[2], x.start[3], x.start[4]};
        x.rc = emit_slice(in, idx.data() + x.lo, x.hi - x.lo, x.ch, sv, o);
        x.len = tl_buf.size() - x.off;
    });
    int64_t total = 0;
    for (Slice& x : sl) { if (x.rc != CSV_OK) return x.rc; total += (int64_t)x.len; }
    *n_written = total;
    if (!out || total > cap) return CSV_E_CAPACITY;
    std::vector<int64_t> toff(sl.size() + 1, 0);
    for (size_t i = 0; i < sl.size(); i++) toff[i + 1] = toff[i] + (int64_t)sl[i].len;
    team.run((int64_t)sl.size(), [&](int64_t i) { const Slice& x = sl[(size_t)i]; memcpy(out + toff[(size_t)i], x.buf->data() + x.off, x.len); });
    for (int t = 0; t < 5; t++) svid[t] = run[t];
    return CSV_OK;
}

namespace {
// the records of the calls v[0 .. nv) of chromosome ch, in that order; svid: the counters to start with
int emit_slice(const csv_vcf_in* in, const int64_t* v, int64_t nv, int ch, int64_t* svid, Sink& o)
{
    const csv_batch_out& R = *in->res;
    static const char* kFormat = "GT:DR:DV:PL:GQ";
    {
        const char* seq = in->chrom_seq ? in->chrom_seq[ch] : nullptr;
        const int64_t slen = in->chrom_len ? in->chrom_len[ch] : 0;
        const char* cname = in->chrom_name[ch];
        // a contig read straight from a FASTA file (mmap): line_bases bases, then the line break, per line of line_width bytes
        const int64_t lb = (in->chrom_line_bases && in->chrom_line_width) ? in->chrom_line_bases[ch] : 0;
        const int64_t lw = lb > 0 ? in->chrom_line_width[ch] : 0;
        auto at = [&](int64_t i) -> char { return lb > 0 ? seq[(i / lb) * lw + i % lb] : seq[i]; };
        auto base = [&](int64_t i, char& c) -> bool { if (!seq || i < 0 || i >= slen) return false; c = at(i); return true; };
        for (int64_t qi = 0; qi < nv; qi++) {
            const int64_t c = v[qi];
            const csv_segment& sg = in->seg[R.call_seg[c]];
            const int type = sg.svtype;
            // TRA calls whose count_coverage gave up carry the '.' fields too (cuteSV_resolveTRA.py:276-281)
            const bool gt_on = sg.genotype != 0 && csv_soa::gl_idx(R, c) >= 0;
            // genotype strings of the row ('.' fields when the task was not genotyped)
            GlRow g{"./.", 3, ".,.,.", 5, ".", 1, ".", 1};
            if (gt_on) {
                const int32_t key = csv_soa::gl_idx(R, c);
                const int32_t* it = std::lower_bound(in->gl_key, in->gl_key + in->n_gl, key);
                if (it == in->gl_key + in->n_gl || *it != key || !split_gl(in->gl_str[it - in->gl_key], g)) return CSV_E_INVALID;
            }
            const bool imprecise = g.gt_n == 3 && memcmp(g.gt, "0/0", 3) == 0;
            // filter label (cuteSV_genotype.py:289-292)
            const char* filt = "PASS";
            if (!(g.qual_n == 1 && g.qual[0] == '.')) filt = strtod(std::string(g.qual, g.qual_n).c_str(), nullptr) >= 5.0 ? "PASS" : "q5";
            const long long pos = csv_soa::bp1(R, c), re = R.support[c];
            auto put_rnames = [&]() {
                if (in->report_readid) { o.put(";RNAMES="); if (in->rnames) o.put(in->rnames + in->rnames_off[c], (size_t)(in->rnames_off[c + 1] - in->rnames_off[c])); }
            };
            auto put_af_field = [&](bool numeric) {
                if (!in->genotype) return;
                o.put(";AF=");
                if (numeric) put_af(o, re, csv_soa::dr(R, c)); else o.put('.');
            };
            auto put_tail = [&]() {           // QUAL FILTER INFO are written by the caller; this is FORMAT + sample
                o.put('\t'); o.put(kFormat); o.put('\t');
                o.put(g.gt, g.gt_n); o.put(':');
                if (gt_on) o.num(csv_soa::dr(R, c)); else o.put('.');
                o.put(':'); o.num(re); o.put(':'); o.put(g.pl, g.pl_n); o.put(':'); o.put(g.gq, g.gq_n); o.put('\n');
            };
            if (type == CSV_DEL || type == CSV_INS) {
                const long long len = csv_soa::bp2(R, c);                                         // |SVLEN|
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
                    if (!seq && r1 > r0) return CSV_E_INVALID;
                    // the deleted bases: whole runs of a FASTA line at a time (memcpy), then the IUPAC table over the copy
                    // (a call, a division and a push_back per base made this slice 95 % of the emitter's time on a 30x genome)
                    if (r1 > r0) {
                        const size_t w0 = o.buf.size();
                        o.buf.resize(w0 + (size_t)(r1 - r0));
                        char* dst = &o.buf[w0];
                        if (lb > 0) {
                            long long i = r0;
                            while (i < r1) {
                                const long long in_line = i % lb, run = std::min<long long>(lb - in_line, r1 - i);
                                memcpy(dst, seq + (i / lb) * lw + in_line, (size_t)run);
                                dst += run; i += run;
                            }
                        } else memcpy(dst, seq + r0, (size_t)(r1 - r0));
                        char* p = &o.buf[w0];
                        for (long long k = 0; k < r1 - r0; k++) p[k] = kIupac.t[(unsigned char)p[k]];
                    }
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
                o.put(";CIPOS=-"); o.num(csv_soa::cipos(R, c)); o.put(','); o.num(csv_soa::cipos(R, c));
                o.put(";CILEN=-"); o.num(csv_soa::cilen(R, c)); o.put(','); o.num(csv_soa::cilen(R, c));
                o.put(";RE="); o.num(re);
                put_rnames();
                put_af_field(gt_on);
                if (type == CSV_DEL) o.put(";STRAND=+-");
                put_tail();
            } else if (type == CSV_DUP) {
                const long long len = csv_soa::bp2(R, c) - csv_soa::bp1(R, c);
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
                const long long len = csv_soa::bp2(R, c) - csv_soa::bp1(R, c);
                if (llabs(len) > in->max_size && in->max_size != -1) continue;            // :351-352
                const char* strand = in->strand_name[csv_soa::call_aux(R, c)];
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
                const int code = csv_soa::call_aux(R, c) & 7;
                const char* chr2 = in->chrom_name[csv_soa::call_aux(R, c) >> 3];
                const long long mate = csv_soa::bp2(R, c) + ((code == 0 || code == 2) ? 1 : 0);    // TRA:140
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
    return CSV_OK;
}
}  // namespace

// ---------------------------------------------------------------------------------------- FASTA index
// The `.fai` of a FASTA file held in memory (mmap): per contig its name (up to the first white space of the header), length,
// byte offset of the first base, bases per line and bytes per line - what `samtools faidx` writes and pysam.FastaFile reads
// (generate_output opens the reference with it, cuteSV_genotype.py:254-259).  One memchr-driven pass; lines of a contig must
// have one length (the last may be shorter), like faidx demands.  Returns the number of contigs (may exceed max_contigs: call
// again with larger arrays), or -CSV_E_INVALID for a file that cannot be indexed.  name_off / name_len address the header text inside `data`.
extern "C" int64_t csv_fasta_index(const char* data, int64_t size, int64_t max_contigs, int64_t* name_off, int32_t* name_len,
                                   int64_t* length, int64_t* offset, int32_t* line_bases, int32_t* line_width)
{
    if (!data || size < 0) return -(int64_t)CSV_E_INVALID;
    int64_t n = 0, p = 0;
    while (p < size) {
        if (data[p] != '>') {                                       // (blank lines between records are tolerated)
            const char* nl = (const char*)memchr(data + p, '\n', (size_t)(size - p));
            if (!nl) break;
            if (nl != data + p && !(nl == data + p + 1 && data[p] == '\r')) return -(int64_t)CSV_E_INVALID;      // sequence before any header
            p = nl - data + 1;
            continue;
        }
        const char* nl = (const char*)memchr(data + p, '\n', (size_t)(size - p));
        const int64_t hdr_end = nl ? nl - data : size;
        int64_t e = p + 1;
        while (e < hdr_end && data[e] != ' ' && data[e] != '\t' && data[e] != '\r') e++;
        const int64_t seq0 = nl ? hdr_end + 1 : size;
        int64_t q = seq0, len = 0;
        int32_t lb = 0, lw = 0;
        bool last_short = false;
        while (q < size && data[q] != '>') {
            const char* l2 = (const char*)memchr(data + q, '\n', (size_t)(size - q));
            const int64_t le = l2 ? l2 - data : size;                  // end of the line's text
            int64_t bases = le - q;
            if (bases > 0 && data[le - 1] == '\r') bases--;
            const int64_t width = (l2 ? le + 1 : le) - q;
            if (bases == 0) { q = l2 ? le + 1 : size; last_short = true; continue; }      // an empty line ends the contig's regular part
            if (last_short) return -(int64_t)CSV_E_INVALID;                     // a longer line after a short one: not indexable
            if (lb == 0) { if (bases > INT32_MAX || width > INT32_MAX) return -(int64_t)CSV_E_INVALID; lb = (int32_t)bases; lw = (int32_t)width; }
            else if (bases > lb) return -(int64_t)CSV_E_INVALID;
            if (bases < lb || (l2 && width != lw)) last_short = true;
            len += bases;
            q = l2 ? le + 1 : size;
        }
        if (n < max_contigs) {
            if (name_off) name_off[n] = p + 1;
            if (name_len) name_len[n] = (int32_t)(e - (p + 1));
            if (length) length[n] = len;
            if (offset) offset[n] = seq0;
            if (line_bases) line_bases[n] = lb;
            if (line_width) line_width[n] = lw;
        }
        n++;
        p = q;
    }
    return n;
}
