// rows_py.cpp — CPython face of the row builder: cutesv_amd/_rows_native.
//
// The reference's resolvers return Python lists of strings (one list per candidate SV); main_ctrl and generate_output
// consume exactly that (cuteSV main script :1191-1197, cuteSV_genotype.py:242-467).  This module compiles the ONE
// statement of the row layouts (rows_layout.h, shared with the C ABI's csv_rows_emit) against a sink that creates the
// str objects directly: the ~350 k strings of a 30x genome are built in one pass over the structure of arrays, the
// short repeated fields ("DEL", "19", "-3,3", "./.", chromosome names ...) shared through a small direct-mapped cache.
// (Round 1 built them in a per-call Python loop: 92 ms; blob + str.split: ~35 ms; one pass of this sink: 19 ms; with the
// text formatted by worker threads beside the object creation - build_parallel below -: 8-10 ms for a 30x genome.)
//   build(addr_of_csv_rows_in) -> list[list[str]]       split(blob, n_rows) -> the same from csv_rows_emit's text
#define PY_SSIZE_T_CLEAN
#include <Python.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

#include "rows_layout.h"

namespace {

constexpr int CACHE_BITS = 12;
struct Slot { uint64_t key; uint32_t len; PyObject* obj; };

struct Cache {
    Slot* s;
    Cache() : s((Slot*)PyMem_Calloc((size_t)1 << CACHE_BITS, sizeof(Slot))) {}
    ~Cache() { if (s) { for (size_t i = 0; i < ((size_t)1 << CACHE_BITS); i++) Py_XDECREF(s[i].obj); PyMem_Free(s); } }
};

inline bool is_ascii(const char* s, Py_ssize_t n)
{
    uint64_t acc = 0;
    Py_ssize_t i = 0;
    for (; i + 8 <= n; i += 8) { uint64_t w; memcpy(&w, s + i, 8); acc |= w; }
    for (; i < n; i++) acc |= (uint64_t)(unsigned char)s[i];
    return (acc & 0x8080808080808080ull) == 0;
}

inline PyObject* make_str(const char* s, Py_ssize_t n)
{
    if (!is_ascii(s, n)) return PyUnicode_DecodeUTF8(s, n, "replace");
    PyObject* u = PyUnicode_New(n, 127);
    if (u && n) memcpy(PyUnicode_1BYTE_DATA(u), s, (size_t)n);
    return u;
}

// new reference to the str of the n <= 8 bytes at s, shared through the cache
inline PyObject* short_str(Cache& C, const char* s, Py_ssize_t n)
{
    uint64_t key = 0;
    memcpy(&key, s, (size_t)n);
    const uint64_t h = (key * 0x9E3779B97F4A7C15ull) ^ (uint64_t)n;
    Slot& e = C.s[h >> (64 - CACHE_BITS)];
    if (e.obj && e.key == key && e.len == (uint32_t)n) { Py_INCREF(e.obj); return e.obj; }
    PyObject* u = make_str(s, n);
    if (u) { Py_XDECREF(e.obj); e.obj = u; e.key = key; e.len = (uint32_t)n; Py_INCREF(u); }
    return u;
}

struct PySink {
    Cache     C;
    PyObject* rows;          // list of rows
    PyObject* row = nullptr;
    int64_t   r = 0;
    int       f = 0;
    bool      ok = true;
    // the field under construction: short ones in tmp, announced long ones straight in their str object
    char      tmp[256];
    char*     w = tmp;
    char*     w_end = tmp + sizeof tmp;
    PyObject* big = nullptr;
    char*     heap = nullptr; // overflow of an unannounced field beyond tmp (never on the reference's layouts)
    size_t    heap_cap = 0;

    explicit PySink(int64_t n) : rows(PyList_New(n)) { ok = rows && C.s; }
    ~PySink() { Py_XDECREF(big); if (heap) PyMem_Free(heap); }

    inline void grow(int64_t need)
    {
        if (big) { ok = false; return; }                   // an announced length was wrong: refuse rather than overrun
        const size_t used = (size_t)(w - (heap ? heap : tmp));
        size_t cap = heap_cap ? heap_cap * 2 : 1024;
        while (cap < used + (size_t)need) cap *= 2;
        char* nb = (char*)PyMem_Malloc(cap);
        if (!nb) { ok = false; return; }
        memcpy(nb, heap ? heap : tmp, used);
        if (heap) PyMem_Free(heap);
        heap = nb; heap_cap = cap; w = nb + used; w_end = nb + cap;
    }
    inline void raw(const char* s, int64_t len)
    {
        if (w + len > w_end) { grow(len); if (!ok) return; }
        memcpy(w, s, (size_t)len);
        w += len;
    }
    inline void ch(char c)
    {
        if (w + 1 > w_end) { grow(1); if (!ok) return; }
        *w++ = c;
    }
    inline void num(int64_t v)
    {
        char b[24];
        const int k = csv_rows::fmt_i64(v, b);
        raw(b + k, 24 - k);
    }
    inline char* reserve(int64_t len)
    {
        if (w + len > w_end) { grow(len); if (!ok) return nullptr; }
        return w;
    }
    inline void commit(int64_t len) { if (ok) w += len; }
    inline void acgt(int64_t len)
    {
        if (len <= 0) return;
        if (w + len > w_end) { grow(len); if (!ok) return; }
        const int64_t first = len < 4 ? len : 4;
        memcpy(w, "ACGT", (size_t)first);
        for (int64_t have = first; have < len;) {
            const int64_t k = have < len - have ? have : len - have;
            memcpy(w + have, w, (size_t)k);
            have += k;
        }
        w += len;
    }
    inline void row_begin(int nf)
    {
        row = ok ? PyList_New(nf) : nullptr;
        if (!row) { ok = false; return; }
        PyList_SET_ITEM(rows, r, row);
        f = 0;
    }
    inline bool row_end() { r++; return ok; }
    inline void field_begin(int64_t exact)
    {
        if (!ok) return;
        if (exact > (int64_t)sizeof tmp) {                 // long field: its bytes go straight into the object
            big = PyUnicode_New(exact, 127);
            if (!big) { ok = false; return; }
            w = (char*)PyUnicode_1BYTE_DATA(big); w_end = w + exact;
        } else { w = tmp; w_end = tmp + sizeof tmp; }
    }
    inline void field_end()
    {
        if (!ok || !row) return;
        PyObject* u;
        if (big) {
            if (w != w_end) { ok = false; return; }        // (announced length must be exact)
            u = big; big = nullptr;
            if (!is_ascii((const char*)PyUnicode_1BYTE_DATA(u), PyUnicode_GET_LENGTH(u))) {   // non-ASCII names: decode properly
                PyObject* d = PyUnicode_DecodeUTF8((const char*)PyUnicode_1BYTE_DATA(u), PyUnicode_GET_LENGTH(u), "replace");
                Py_DECREF(u); u = d;
            }
        } else {
            const char* base = heap ? heap : tmp;
            const Py_ssize_t n = w - base;
            u = n <= 8 ? short_str(C, base, n) : make_str(base, n);
            if (heap) { PyMem_Free(heap); heap = nullptr; heap_cap = 0; }
        }
        if (!u) { ok = false; return; }
        if (f < PyList_GET_SIZE(row)) PyList_SET_ITEM(row, f, u); else { Py_DECREF(u); ok = false; }
        f++;
        w = tmp; w_end = tmp + sizeof tmp;
    }
};

// The text of a slice of the rows, produced WITHOUT the interpreter (so that slices can be made by several threads while
// the GIL is released): the bytes of all fields back to back, the length of every field, the field count of every row.
inline double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct SpanSink {
    char*                 p = nullptr;       // raw bytes (malloc: no value initialisation of 20 MB of text)
    size_t                n = 0, cap = 0;
    std::vector<uint32_t> flen;
    std::vector<uint8_t>  nf;
    size_t                fstart = 0;
    bool                  ok = true;
    ~SpanSink() { free(p); }
    inline bool need(size_t len)
    {
        if (n + len <= cap) return true;
        size_t c = cap ? cap * 2 : (1u << 20);
        while (c < n + len) c *= 2;
        char* q = (char*)realloc(p, c);
        if (!q) { ok = false; return false; }
        p = q; cap = c;
        return true;
    }
    inline void raw(const char* s, int64_t len) { if (len > 0 && need((size_t)len)) { memcpy(p + n, s, (size_t)len); n += (size_t)len; } }
    inline void ch(char c) { if (need(1)) p[n++] = c; }
    inline void num(int64_t v) { char b[24]; const int k = csv_rows::fmt_i64(v, b); raw(b + k, 24 - k); }
    inline char* reserve(int64_t len) { return need((size_t)len) ? p + n : nullptr; }
    inline void commit(int64_t len) { if (ok) n += (size_t)len; }
    inline void acgt(int64_t len)
    {
        if (len <= 0 || !need((size_t)len)) return;
        char* d = p + n;
        const int64_t first = len < 4 ? len : 4;
        memcpy(d, "ACGT", (size_t)first);
        for (int64_t have = first; have < len;) { const int64_t k = have < len - have ? have : len - have; memcpy(d + have, d, (size_t)k); have += k; }
        n += (size_t)len;
    }
    inline void row_begin(int k) { nf.push_back((uint8_t)k); }
    inline bool row_end() { return ok; }
    inline void field_begin(int64_t) { fstart = n; }
    inline void field_end() { flen.push_back((uint32_t)(n - fstart)); }
};

// Large batches, two kinds of work side by side: worker threads (no interpreter: pure C++) format the text of one slice
// of the calls after the other - gathers of read ids, name joins, number formatting, the inserted sequences: 70 % of the
// single-pass time - while this thread, which holds the GIL, turns every finished slice into list and str objects, in
// order.  The object creation (one thread, ~20 ns per object) is what remains on the critical path.
PyObject* build_parallel(const csv_rows_in* in, int n_threads)
{
    const csv_batch_out& R = *in->res;
    const int64_t nc = R.n_calls;
    const bool dbg = getenv("CSV_ROWS_DEBUG") != nullptr;
    const double t_begin = dbg ? now_ms() : 0;
    // slices of about equal support counts (the read lists are the bulk of the text), several per worker so that the first
    // ones are ready early
    const int n_slices = n_threads * 4;
    std::vector<int64_t> cut(n_slices + 1, 0);
    const int64_t ns = R.support_off[nc];
    for (int t = 1; t < n_slices; t++) {
        const int64_t want = ns * t / n_slices;
        cut[t] = std::lower_bound(R.support_off, R.support_off + nc, want) - R.support_off;
        if (cut[t] < cut[t - 1]) cut[t] = cut[t - 1];
    }
    cut[n_slices] = nc;
    std::vector<SpanSink> part(n_slices);
    std::vector<int> rc(n_slices, CSV_OK);
    std::vector<std::atomic<int>> ready(n_slices);
    for (auto& x : ready) x.store(0, std::memory_order_relaxed);
    std::atomic<int> next{0};
    std::vector<std::thread> th;
    auto work = [&] {
        for (;;) {
            const int t = next.fetch_add(1, std::memory_order_relaxed);
            if (t >= n_slices) return;
            try {
                part[t].need((size_t)((cut[t + 1] - cut[t]) * 900 + 4096)); part[t].flen.reserve((size_t)(cut[t + 1] - cut[t]) * 14);
                rc[t] = csv_rows::layout(in, part[t], cut[t], cut[t + 1]);
                if (!part[t].ok) rc[t] = CSV_E_NOMEM;
            } catch (...) { rc[t] = CSV_E_NOMEM; }
            ready[t].store(1, std::memory_order_release);
        }
    };
    try { for (int t = 0; t < n_threads; t++) th.emplace_back(work); }
    catch (...) { }                                        // (fewer threads than asked for: the ones that started drain the queue)
    if (th.empty()) work();
    PyObject* rows = PyList_New(nc);
    Cache C;
    bool ok = rows && C.s;
    int err = CSV_OK;
    int64_t r = 0;
    double t_wait = 0;
    for (int t = 0; t < n_slices; t++) {
        const double w0 = dbg ? now_ms() : 0;
        while (!ready[t].load(std::memory_order_acquire)) std::this_thread::yield();
        if (dbg) t_wait += now_ms() - w0;
        if (rc[t] != CSV_OK) { err = rc[t]; ok = false; }
        if (!ok) continue;                                 // (keep draining: the workers hold references to this frame)
        const SpanSink& P = part[t];
        const char* p = P.p;
        size_t fi = 0;
        for (size_t q = 0; q < P.nf.size() && ok; q++, r++) {
            const int nf = P.nf[q];
            PyObject* row = PyList_New(nf);
            if (!row) { ok = false; break; }
            PyList_SET_ITEM(rows, r, row);
            for (int f = 0; f < nf; f++) {
                const Py_ssize_t n = P.flen[fi++];
                PyObject* u = n <= 8 ? short_str(C, p, n) : make_str(p, n);
                if (!u) { ok = false; break; }
                PyList_SET_ITEM(row, f, u);
                p += n;
            }
        }
        free(part[t].p); part[t].p = nullptr;              // (give the text back as soon as it is objects)
    }
    for (auto& x : th) x.join();
    if (ok && r != nc) { ok = false; err = CSV_E_INVALID; }
    if (!ok) {
        Py_XDECREF(rows);
        if (!PyErr_Occurred()) { if (err) PyErr_Format(PyExc_RuntimeError, "row builder failed (code %d)", err); else PyErr_NoMemory(); }
        return nullptr;
    }
    if (dbg) fprintf(stderr, "[rows] %d threads, %d slices: %.2f ms, of which %.2f waiting for text\n", n_threads, n_slices, now_ms() - t_begin, t_wait);
    return rows;
}

PyObject* build(PyObject*, PyObject* arg)
{
    const csv_rows_in* in = (const csv_rows_in*)PyLong_AsVoidPtr(arg);
    if (!in) { if (!PyErr_Occurred()) PyErr_SetString(PyExc_ValueError, "null csv_rows_in"); return nullptr; }
    if (!in->res) { PyErr_SetString(PyExc_ValueError, "csv_rows_in.res is null"); return nullptr; }
    // Rows are lists of str: they cannot be part of a reference cycle, yet every PyList_New counts towards the collector's
    // thresholds and 25 k of them trigger dozens of young-generation passes (~15 % of the call).  Collection is paused
    // for the duration of the build.
    const int gc_was_on = PyGC_Disable();
    // large batches: text in parallel, objects afterwards (CSV_ROWS_THREADS overrides the thread count; 1 = single pass)
    const int64_t nc = in->res->n_calls;
    int n_threads = 1;
    if (nc >= 4096 && in->res->support_off) {
        const unsigned hw = std::thread::hardware_concurrency();
        n_threads = (int)(hw ? (hw < 16 ? hw : 16) : 4);
        if (const char* e = getenv("CSV_ROWS_THREADS")) n_threads = atoi(e) > 0 ? atoi(e) : 1;
        if (n_threads > 64) n_threads = 64;
        if ((int64_t)n_threads > nc / 1024) n_threads = (int)(nc / 1024 > 0 ? nc / 1024 : 1);
    }
    if (n_threads > 1) {
        PyObject* rows = build_parallel(in, n_threads);
        if (gc_was_on) PyGC_Enable();
        return rows;
    }
    PySink S(in->res->n_calls);
    const int rc = S.ok ? csv_rows::layout(in, S) : CSV_E_NOMEM;
    if (gc_was_on) PyGC_Enable();
    if (rc != CSV_OK || !S.ok) {
        Py_XDECREF(S.rows);
        if (!PyErr_Occurred()) PyErr_Format(PyExc_RuntimeError, "row builder failed (code %d)", rc ? rc : CSV_E_NOMEM);
        return nullptr;
    }
    return S.rows;
}

// build_some(addr_of_csv_rows_in, int64 call indices) -> the rows of exactly those calls, in the order given: what a lazy row
// sequence (cutesv_amd/rows.py LazyRows) materialises when a consumer indexes or iterates it
PyObject* build_some(PyObject*, PyObject* args)
{
    PyObject* addr;
    Py_buffer view;
    if (!PyArg_ParseTuple(args, "Oy*", &addr, &view)) return nullptr;
    const csv_rows_in* in = (const csv_rows_in*)PyLong_AsVoidPtr(addr);
    if (!in || !in->res) { PyBuffer_Release(&view); if (!PyErr_Occurred()) PyErr_SetString(PyExc_ValueError, "null csv_rows_in"); return nullptr; }
    const int64_t n = (int64_t)(view.len / 8), nc = in->res->n_calls;
    const int64_t* idx = (const int64_t*)view.buf;
    const int gc_was_on = PyGC_Disable();
    PySink S(n);
    int rc = S.ok ? CSV_OK : CSV_E_NOMEM;
    for (int64_t i = 0; i < n && rc == CSV_OK && S.ok; i++) {
        if (idx[i] < 0 || idx[i] >= nc) { rc = CSV_E_INVALID; break; }
        rc = csv_rows::layout(in, S, idx[i], idx[i] + 1);
    }
    if (gc_was_on) PyGC_Enable();
    PyBuffer_Release(&view);
    if (rc != CSV_OK || !S.ok || S.r != n) {
        Py_XDECREF(S.rows);
        if (!PyErr_Occurred()) PyErr_Format(PyExc_RuntimeError, "row builder failed (code %d)", rc ? rc : CSV_E_NOMEM);
        return nullptr;
    }
    return S.rows;
}

// split(buffer, n_rows) -> list of n_rows lists of str from csv_rows_emit's text (rows end with '\n', fields are
// separated by '\t'): the generic way in for hosts that only have the blob
PyObject* split(PyObject*, PyObject* args)
{
    Py_buffer view;
    Py_ssize_t n_rows;
    if (!PyArg_ParseTuple(args, "y*n", &view, &n_rows)) return nullptr;
    const char* p = (const char*)view.buf;
    const char* end = p + view.len;
    PyObject* rows = PyList_New(n_rows);
    Cache C;
    bool ok = rows && C.s;
    for (Py_ssize_t r = 0; ok && r < n_rows; r++) {
        const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
        if (!nl) { PyErr_SetString(PyExc_ValueError, "fewer rows in the blob than announced"); ok = false; break; }
        Py_ssize_t nf = 1;
        for (const char* q = p; (q = (const char*)memchr(q, '\t', (size_t)(nl - q))) != nullptr; q++) nf++;
        PyObject* row = PyList_New(nf);
        if (!row) { ok = false; break; }
        PyList_SET_ITEM(rows, r, row);
        const char* q = p;
        for (Py_ssize_t f = 0; f < nf; f++) {
            const char* t = (const char*)memchr(q, '\t', (size_t)(nl - q));
            if (!t) t = nl;
            PyObject* u = t - q <= 8 ? short_str(C, q, t - q) : make_str(q, t - q);
            if (!u) { ok = false; break; }
            PyList_SET_ITEM(row, f, u);
            q = t + 1;
        }
        p = nl + 1;
    }
    PyBuffer_Release(&view);
    if (!ok) { Py_XDECREF(rows); if (!PyErr_Occurred()) PyErr_NoMemory(); return nullptr; }
    return rows;
}

PyMethodDef methods[] = {{"build", build, METH_O, "build(address of a csv_rows_in) -> list[list[str]]"},
                         {"build_some", build_some, METH_VARARGS, "build_some(address of a csv_rows_in, int64 call indices) -> list[list[str]]"},
                         {"split", split, METH_VARARGS, "split(buffer, n_rows) -> list[list[str]]"},
                         {nullptr, nullptr, 0, nullptr}};
PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_rows_native", "row builder of cutesv_amd", -1, methods, nullptr, nullptr, nullptr, nullptr};

}  // namespace

PyMODINIT_FUNC PyInit__rows_native(void) { return PyModule_Create(&moddef); }
