// rows_emit.cpp — csv_rows_emit (include/cutesv_hip.h): the calls of a batch as one text blob in the reference's row
// layouts (rows_layout.h), fields '\t', rows '\n'.  No GPU work: plain C++ on the caller's thread.
#include "rows_layout.h"

namespace {

struct TextSink {
    char*   p;
    int64_t cap, n;
    int     fld;
    inline void raw(const char* s, int64_t len)
    {
        if (n + len <= cap) memcpy(p + n, s, (size_t)len);
        n += len;
    }
    inline void ch(char c)
    {
        if (n < cap) p[n] = c;
        n++;
    }
    inline void num(int64_t v)
    {
        char b[24];
        const int k = csv_rows::fmt_i64(v, b);
        raw(b + k, 24 - k);
    }
    inline char* reserve(int64_t len) { return n + len <= cap ? p + n : nullptr; }
    inline void commit(int64_t len) { n += len; }
    inline void acgt(int64_t len)                         // "ACGT" repeated to len (inserted sequence of the synthetic workloads)
    {
        if (len <= 0) return;
        if (n + len <= cap) {
            char* d = p + n;
            const int64_t first = len < 4 ? len : 4;
            memcpy(d, "ACGT", (size_t)first);
            for (int64_t have = first; have < len;) {     // doubling copy
                const int64_t k = have < len - have ? have : len - have;
                memcpy(d + have, d, (size_t)k);
                have += k;
            }
        }
        n += len;
    }
    inline void row_begin(int) { fld = 0; }
    inline bool row_end() { ch('\n'); return true; }
    inline void field_begin(int64_t) { if (fld++) ch('\t'); }
    inline void field_end() {}
};

}  // namespace

extern "C" int csv_rows_emit(const csv_rows_in* in, char* out, int64_t cap, int64_t* n_written)
{
    if (!n_written) return CSV_E_INVALID;
    TextSink S{out, out ? cap : 0, 0, 0};
    const int rc = csv_rows::layout(in, S);
    if (rc) return rc;
    *n_written = S.n;
    return (out && S.n <= cap) ? CSV_OK : CSV_E_CAPACITY;
}
