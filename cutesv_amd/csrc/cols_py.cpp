// _cols_native: the reference's task lists -> flat columns, without a Python-level loop per tuple.
//
// A pool worker of the reference unpickles one list of tuples per task (cuteSV_resolveINDEL.py:52-58, cuteSV_resolveDUP.py:25-27,
// cuteSV_resolveINV.py:42-44, cuteSV_resolveTRA.py:36-38: `pickle.load` at sigs_index[type][chr]) and walks it tuple by tuple.
// The drop-in (cutesv_amd/resolve.py run_*) has to turn the same list into the columns of include/cutesv_hip.h first; done
// with list comprehensions and numpy.fromiter that was ~50 ms for the 110 862 signatures of INS chr2 - 50x the GPU call it
// prepares.  walk() reads every tuple once in C: integer fields (int() of an int or an x.5 float, main script :228) into
// int64 / int32 / uint8 buffers, string fields interned by FIRST APPEARANCE into int32 ids through a dict the caller keeps
// (read names: the list is already in the rebuild's order, ids only have to tell reads apart), len() of a field (the
// inserted sequence of an INS signature) into an int32 buffer.
//
//   walk(seq, ints, interns, lens) -> None
//     seq      list (or tuple) of tuples
//     ints     tuple of (field, buffer)              buffer: writable, C-contiguous, itemsize 8, 4 or 1, len(seq) items
//     interns  tuple of (field, buffer, dict)        buffer: int32; dict: value -> id, extended in place with len(dict) for new values
//     lens     tuple of (field, buffer)              buffer: int32
// Errors (wrong shapes, a field that is neither int nor float, ids beyond int32) raise; nothing is partially trusted.
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>
#include <string.h>
#include <vector>

namespace {

struct Buf {
    Py_buffer v{};
    bool held = false;
    ~Buf() { if (held) PyBuffer_Release(&v); }
    bool get(PyObject* o, Py_ssize_t n, const char* what)
    {
        if (PyObject_GetBuffer(o, &v, PyBUF_WRITABLE | PyBUF_C_CONTIGUOUS) != 0) return false;
        held = true;
        if (v.itemsize <= 0 || v.len / v.itemsize < n) { PyErr_Format(PyExc_ValueError, "%s buffer holds %zd items, the list %zd", what, v.len / (v.itemsize > 0 ? v.itemsize : 1), n); return false; }
        return true;
    }
};
struct IntSpec { Py_ssize_t field; Buf buf; };
struct InternSpec { Py_ssize_t field; Buf buf; PyObject* dict; };

bool to_i64(PyObject* x, int64_t& out)
{
    if (PyLong_Check(x)) {
        int overflow = 0;
        const long long v = PyLong_AsLongLongAndOverflow(x, &overflow);
        if (overflow || (v == -1 && PyErr_Occurred())) { if (!PyErr_Occurred()) PyErr_SetString(PyExc_OverflowError, "integer field beyond int64"); return false; }
        out = (int64_t)v;
        return true;
    }
    if (PyFloat_Check(x)) { out = (int64_t)PyFloat_AS_DOUBLE(x); return true; }      // int(x): truncation toward zero
    PyObject* as_int = PyNumber_Long(x);                                              // numpy scalars and the like
    if (!as_int) return false;
    const bool ok = to_i64(as_int, out);
    Py_DECREF(as_int);
    return ok;
}

PyObject* walk(PyObject*, PyObject* args)
{
    PyObject *seq, *ints, *interns, *lens;
    if (!PyArg_ParseTuple(args, "OO!O!O!", &seq, &PyTuple_Type, &ints, &PyTuple_Type, &interns, &PyTuple_Type, &lens)) return nullptr;
    PyObject* fast = PySequence_Fast(seq, "walk: a list or tuple of tuples is expected");
    if (!fast) return nullptr;
    struct Guard { PyObject* o; ~Guard() { Py_DECREF(o); } } guard{fast};
    const Py_ssize_t n = PySequence_Fast_GET_SIZE(fast);
    PyObject** rows = PySequence_Fast_ITEMS(fast);

    const Py_ssize_t ni = PyTuple_GET_SIZE(ints), nn = PyTuple_GET_SIZE(interns), nl = PyTuple_GET_SIZE(lens);
    std::vector<IntSpec> is((size_t)ni), ls((size_t)nl);
    std::vector<InternSpec> ns((size_t)nn);
    for (Py_ssize_t k = 0; k < ni; k++) {
        PyObject* t = PyTuple_GET_ITEM(ints, k); PyObject* b;
        if (!PyArg_ParseTuple(t, "nO", &is[k].field, &b) || !is[k].buf.get(b, n, "int")) return nullptr;
        const Py_ssize_t w = is[k].buf.v.itemsize;
        if (w != 8 && w != 4 && w != 1) { PyErr_SetString(PyExc_ValueError, "int buffers must have itemsize 8, 4 or 1"); return nullptr; }
    }
    for (Py_ssize_t k = 0; k < nn; k++) {
        PyObject* t = PyTuple_GET_ITEM(interns, k); PyObject* b;
        if (!PyArg_ParseTuple(t, "nOO!", &ns[k].field, &b, &PyDict_Type, &ns[k].dict) || !ns[k].buf.get(b, n, "intern")) return nullptr;
        if (ns[k].buf.v.itemsize != 4) { PyErr_SetString(PyExc_ValueError, "intern buffers must be int32"); return nullptr; }
    }
    for (Py_ssize_t k = 0; k < nl; k++) {
        PyObject* t = PyTuple_GET_ITEM(lens, k); PyObject* b;
        if (!PyArg_ParseTuple(t, "nO", &ls[k].field, &b) || !ls[k].buf.get(b, n, "len")) return nullptr;
        if (ls[k].buf.v.itemsize != 4) { PyErr_SetString(PyExc_ValueError, "len buffers must be int32"); return nullptr; }
    }
    for (Py_ssize_t i = 0; i < n; i++) {
        PyObject* row = rows[i];
        if (!PyTuple_Check(row)) { PyErr_Format(PyExc_TypeError, "walk: element %zd is not a tuple", i); return nullptr; }
        const Py_ssize_t width = PyTuple_GET_SIZE(row);
        for (auto& s : is) {
            if (s.field < 0 || s.field >= width) { PyErr_Format(PyExc_IndexError, "walk: tuple %zd has no field %zd", i, s.field); return nullptr; }
            int64_t v;
            if (!to_i64(PyTuple_GET_ITEM(row, s.field), v)) return nullptr;
            if (s.buf.v.itemsize == 8) ((int64_t*)s.buf.v.buf)[i] = v;
            else if (s.buf.v.itemsize == 4) {
                if (v < INT32_MIN || v > INT32_MAX) { PyErr_Format(PyExc_OverflowError, "walk: field %zd of tuple %zd does not fit int32", s.field, i); return nullptr; }
                ((int32_t*)s.buf.v.buf)[i] = (int32_t)v;
            } else ((uint8_t*)s.buf.v.buf)[i] = (uint8_t)v;
        }
        for (auto& s : ns) {
            if (s.field < 0 || s.field >= width) { PyErr_Format(PyExc_IndexError, "walk: tuple %zd has no field %zd", i, s.field); return nullptr; }
            PyObject* key = PyTuple_GET_ITEM(row, s.field);
            PyObject* val = PyDict_GetItemWithError(s.dict, key);                    // borrowed
            long id;
            if (val) id = PyLong_AsLong(val);
            else {
                if (PyErr_Occurred()) return nullptr;
                const Py_ssize_t next = PyDict_GET_SIZE(s.dict);
                if (next > INT32_MAX) { PyErr_SetString(PyExc_OverflowError, "walk: more than 2^31 distinct values"); return nullptr; }
                PyObject* nv = PyLong_FromSsize_t(next);
                if (!nv || PyDict_SetItem(s.dict, key, nv) != 0) { Py_XDECREF(nv); return nullptr; }
                Py_DECREF(nv);
                id = (long)next;
            }
            if (id == -1 && PyErr_Occurred()) return nullptr;
            ((int32_t*)s.buf.v.buf)[i] = (int32_t)id;
        }
        for (auto& s : ls) {
            if (s.field < 0 || s.field >= width) { PyErr_Format(PyExc_IndexError, "walk: tuple %zd has no field %zd", i, s.field); return nullptr; }
            const Py_ssize_t len = PyObject_Length(PyTuple_GET_ITEM(row, s.field));
            if (len < 0) return nullptr;
            if (len > INT32_MAX) { PyErr_SetString(PyExc_OverflowError, "walk: a field is longer than 2^31"); return nullptr; }
            ((int32_t*)s.buf.v.buf)[i] = (int32_t)len;
        }
    }
    Py_RETURN_NONE;
}

// intern(specs) -> list of the distinct values in order of first appearance
//   specs   tuple of (seq, field, buffer): buffer[i] = id of seq[i][field]; ONE id space over all specs, in the order given.
// For the read names of a task (10^5 distinct str objects fresh from pickle.load): an open-addressing table of our own keyed by
// the objects' hashes - a Python dict with PyLong values spent 4x as long on resizes and on the id objects.
PyObject* intern(PyObject*, PyObject* args)
{
    PyObject* specs;
    if (!PyArg_ParseTuple(args, "O!", &PyTuple_Type, &specs)) return nullptr;
    const Py_ssize_t ns = PyTuple_GET_SIZE(specs);
    struct Spec { PyObject* fast = nullptr; Py_ssize_t field = 0, n = 0; Buf buf; ~Spec() { Py_XDECREF(fast); } };
    std::vector<Spec> sp((size_t)ns);
    Py_ssize_t total = 0;
    for (Py_ssize_t k = 0; k < ns; k++) {
        PyObject *seq, *b;
        if (!PyArg_ParseTuple(PyTuple_GET_ITEM(specs, k), "OnO", &seq, &sp[k].field, &b)) return nullptr;
        sp[k].fast = PySequence_Fast(seq, "intern: a list or tuple of tuples is expected");
        if (!sp[k].fast) return nullptr;
        sp[k].n = PySequence_Fast_GET_SIZE(sp[k].fast);
        if (!sp[k].buf.get(b, sp[k].n, "intern")) return nullptr;
        if (sp[k].buf.v.itemsize != 4) { PyErr_SetString(PyExc_ValueError, "intern buffers must be int32"); return nullptr; }
        total += sp[k].n;
    }
    size_t cap = 64;
    while (cap < (size_t)total * 2 + 8) cap <<= 1;
    std::vector<int32_t> slot(cap, -1);
    std::vector<Py_hash_t> hashes;
    hashes.reserve((size_t)total / 2 + 8);
    PyObject* uniq = PyList_New(0);
    if (!uniq) return nullptr;
    for (auto& s : sp) {
        PyObject** rows = PySequence_Fast_ITEMS(s.fast);
        int32_t* out = (int32_t*)s.buf.v.buf;
        for (Py_ssize_t i = 0; i < s.n; i++) {
            PyObject* row = rows[i];
            if (!PyTuple_Check(row) || s.field < 0 || s.field >= PyTuple_GET_SIZE(row)) {
                PyErr_Format(PyExc_IndexError, "intern: element %zd has no field %zd", i, s.field); Py_DECREF(uniq); return nullptr;
            }
            PyObject* key = PyTuple_GET_ITEM(row, s.field);
            const Py_hash_t h = PyObject_Hash(key);
            if (h == -1 && PyErr_Occurred()) { Py_DECREF(uniq); return nullptr; }
            size_t p = ((size_t)h * 0x9E3779B97F4A7C15ull) >> 7 & (cap - 1);
            int32_t id = -1;
            for (;; p = (p + 1) & (cap - 1)) {
                const int32_t q = slot[p];
                if (q < 0) break;
                if (hashes[(size_t)q] != h) continue;
                PyObject* other = PyList_GET_ITEM(uniq, q);
                if (other == key) { id = q; break; }
                const int eq = PyObject_RichCompareBool(other, key, Py_EQ);
                if (eq < 0) { Py_DECREF(uniq); return nullptr; }
                if (eq) { id = q; break; }
            }
            if (id < 0) {
                const Py_ssize_t next = PyList_GET_SIZE(uniq);
                if (next >= INT32_MAX) { PyErr_SetString(PyExc_OverflowError, "intern: more than 2^31 distinct values"); Py_DECREF(uniq); return nullptr; }
                if (PyList_Append(uniq, key) != 0) { Py_DECREF(uniq); return nullptr; }
                hashes.push_back(h);
                slot[p] = id = (int32_t)next;
            }
            out[i] = id;
        }
    }
    return uniq;
}

// column(seq, field) -> list of seq[i][field] (one C loop instead of a list comprehension; the objects are shared, not copied)
PyObject* column(PyObject*, PyObject* args)
{
    PyObject* seq; Py_ssize_t field;
    if (!PyArg_ParseTuple(args, "On", &seq, &field)) return nullptr;
    PyObject* fast = PySequence_Fast(seq, "column: a list or tuple of tuples is expected");
    if (!fast) return nullptr;
    const Py_ssize_t n = PySequence_Fast_GET_SIZE(fast);
    PyObject** rows = PySequence_Fast_ITEMS(fast);
    PyObject* out = PyList_New(n);
    if (!out) { Py_DECREF(fast); return nullptr; }
    for (Py_ssize_t i = 0; i < n; i++) {
        PyObject* row = rows[i];
        if (!PyTuple_Check(row) || field < 0 || field >= PyTuple_GET_SIZE(row)) {
            PyErr_Format(PyExc_IndexError, "column: element %zd has no field %zd", i, field);
            Py_DECREF(out); Py_DECREF(fast); return nullptr;
        }
        PyObject* x = PyTuple_GET_ITEM(row, field);
        Py_INCREF(x);
        PyList_SET_ITEM(out, i, x);
    }
    Py_DECREF(fast);
    return out;
}

// clip_join(table, picks, lens, out_len) -> bytes: b"".join(table[picks[i]][:lens[i]].encode() for i ...), the lengths actually
// taken written to out_len - the ALT strings of a batch's INS calls (cuteSV_genotype.py:297-309: the inserted sequence sliced
// to SVLEN) as csv_vcf_in.ins_alt takes them, in two C passes over the picked strings instead of four Python-level ones.
//   table    list / tuple by index, dict keyed by int, or None: the synthetic stores' "ACGT" repeated (lens is then the length)
//   picks    int64 buffer (indices / keys); lens: int64 buffer (Python slice ends: negative counts from the end); out_len: int64, writable
// A string that is not ASCII takes the Python slice itself (code points, not bytes).
PyObject* clip_join(PyObject*, PyObject* args)
{
    PyObject *table, *opicks, *olens, *oout;
    if (!PyArg_ParseTuple(args, "OOOO", &table, &opicks, &olens, &oout)) return nullptr;
    Py_buffer pk{}, ln{};
    if (PyObject_GetBuffer(opicks, &pk, PyBUF_C_CONTIGUOUS) != 0) return nullptr;
    if (PyObject_GetBuffer(olens, &ln, PyBUF_C_CONTIGUOUS) != 0) { PyBuffer_Release(&pk); return nullptr; }
    Buf out;
    PyObject* res = nullptr;
    PyObject* fast = nullptr;
    std::vector<PyObject*> owned;                 // slices of non-ASCII strings (kept until the copy is done)
    struct Piece { const char* p; Py_ssize_t n; };
    std::vector<Piece> pieces;
    do {
        if (pk.itemsize != 8 || ln.itemsize != 8 || pk.len != ln.len) { PyErr_SetString(PyExc_ValueError, "clip_join: picks and lens must be int64 buffers of one length"); break; }
        const Py_ssize_t n = pk.len / 8;
        if (!out.get(oout, n, "clip_join out_len") || out.v.itemsize != 8) { if (!PyErr_Occurred()) PyErr_SetString(PyExc_ValueError, "clip_join: out_len must be int64"); break; }
        const int64_t* picks = (const int64_t*)pk.buf;
        const int64_t* lens = (const int64_t*)ln.buf;
        int64_t* ol = (int64_t*)out.v.buf;
        const bool synthetic = table == Py_None, is_dict = PyDict_Check(table);
        PyObject** items = nullptr; Py_ssize_t n_items = 0;
        if (!synthetic && !is_dict) {
            fast = PySequence_Fast(table, "clip_join: table must be a list, a tuple, a dict or None");
            if (!fast) break;
            items = PySequence_Fast_ITEMS(fast); n_items = PySequence_Fast_GET_SIZE(fast);
        }
        pieces.resize((size_t)n);
        int64_t total = 0, longest = 0;
        bool ok = true;
        for (Py_ssize_t i = 0; i < n && ok; i++) {
            if (synthetic) {
                const int64_t k = lens[i] > 0 ? lens[i] : 0;
                pieces[(size_t)i] = {nullptr, (Py_ssize_t)k};
                ol[i] = k; total += k; if (k > longest) longest = k;
                continue;
            }
            PyObject* s;
            if (is_dict) {
                PyObject* key = PyLong_FromLongLong(picks[i]);
                if (!key) { ok = false; break; }
                s = PyDict_GetItemWithError(table, key);
                Py_DECREF(key);
                if (!s) { if (!PyErr_Occurred()) PyErr_Format(PyExc_KeyError, "%lld", (long long)picks[i]); ok = false; break; }
            } else {
                if (picks[i] < 0 || picks[i] >= n_items) { PyErr_Format(PyExc_IndexError, "clip_join: index %lld outside the table", (long long)picks[i]); ok = false; break; }
                s = items[picks[i]];
            }
            if (!PyUnicode_Check(s)) { PyErr_SetString(PyExc_TypeError, "clip_join: the table must hold str"); ok = false; break; }
            Py_ssize_t len = PyUnicode_GET_LENGTH(s);
            int64_t end = lens[i] < 0 ? lens[i] + len : lens[i];
            if (end < 0) end = 0;
            if (end > len) end = len;
            if (!PyUnicode_IS_ASCII(s)) {
                s = PyUnicode_Substring(s, 0, (Py_ssize_t)end);
                if (!s) { ok = false; break; }
                owned.push_back(s);
                Py_ssize_t nb;
                const char* p = PyUnicode_AsUTF8AndSize(s, &nb);
                if (!p) { ok = false; break; }
                pieces[(size_t)i] = {p, nb};
                ol[i] = nb; total += nb;
                continue;
            }
            pieces[(size_t)i] = {(const char*)PyUnicode_1BYTE_DATA(s), (Py_ssize_t)end};
            ol[i] = end; total += end;
        }
        if (!ok) break;
        res = PyBytes_FromStringAndSize(nullptr, (Py_ssize_t)total);
        if (!res) break;
        char* dst = PyBytes_AS_STRING(res);
        std::vector<char> pattern;
        if (synthetic) { pattern.resize((size_t)longest + 4); for (size_t k = 0; k < pattern.size(); k++) pattern[k] = "ACGT"[k & 3]; }
        for (const Piece& q : pieces) {
            memcpy(dst, synthetic ? pattern.data() : q.p, (size_t)q.n);
            dst += q.n;
        }
    } while (false);
    for (PyObject* o : owned) Py_DECREF(o);
    Py_XDECREF(fast);
    PyBuffer_Release(&pk); PyBuffer_Release(&ln);
    return res;
}

// ------------------------------------------------------------------------------------------ the task pickles without the objects
// A pool worker of the reference starts every task with pickle.load at sigs_index[type][chr] (cuteSV_resolveINDEL.py:52-58,
// cuteSV_resolveDUP.py:25-27, cuteSV_resolveINV.py:42-44, cuteSV_resolveTRA.py:36-38): a list of tuples of ints, x.5 floats and
// strings that the main script wrote with pickle.dumps (:817-857).  For the 110 862 signatures of INS chr2 that is ~450 k Python
// objects and 21-24 ms before the first of them is looked at - and the drop-in wants columns, not objects.  pickle_table() walks
// the opcode stream itself: integer fields go straight into int64 columns (int() of a float: truncation, main script :228),
// string fields become (offset, length) spans of their UTF-8 bytes inside the buffer (a 26 MB block of inserted sequences is
// never touched), memo references are followed.  It understands exactly what pickle emits for such lists - protocols 2 to 5:
// PROTO FRAME EMPTY_LIST MARK APPEND(S) TUPLE/1/2/3 EMPTY_TUPLE BININT/1/2 LONG1 BINFLOAT (SHORT_)BINUNICODE(8) MEMOIZE BINPUT
// LONG_BINPUT BINGET LONG_BINGET NEWTRUE NEWFALSE NONE STOP - and returns None for anything else (the caller then uses pickle).
//
//   pickle_table(buf, offset, width, int_fields, str_fields[, end_hint]) -> None | (n_rows, end_offset, [int64 bytes per int field],
//                                                                      [(offsets int64 bytes, lengths int32 bytes) per str field],
//                                                                      all_ascii: no string payload of the stream has a byte >= 0x80)
//   span_intern(((buf, offsets, lengths, ids int32 out), ...)) -> (blob bytes, offsets int64 bytes, lengths int32 bytes) of the
//                 distinct strings by first appearance, ONE id space over all specs
//   span_join(buf, offsets, lengths, picks int64, clips int64 | None, out_len int64) -> bytes     (clip_join for a span table)
//   span_cplen(buf, offsets, lengths, out int32)       len() of every string (code points; bytes when the text is ASCII)
namespace pk {
enum Kind : uint8_t { LIST, MARKER, ROW, INT, FLOAT, STR, OTHER };
struct Cell { Kind kind; int32_t len; int64_t a; };

// what PyUnicode_DecodeUTF8(.., "surrogatepass") - pickle's decoder of BINUNICODE payloads - accepts: well-formed UTF-8 (no
// overlong forms, nothing above U+10FFFF) plus three-byte encoded surrogates
inline bool utf8_ok(const unsigned char* s, int64_t n, bool* not_ascii = nullptr)
{
    {   // all ASCII - every read name and sequence of a real file - is one OR over the payload, 32 bytes a step (memory speed)
        uint64_t acc = 0;
        int64_t j = 0;
        for (; j + 32 <= n; j += 32) { uint64_t w[4]; memcpy(w, s + j, 32); acc |= (w[0] | w[1]) | (w[2] | w[3]); }
        for (; j + 8 <= n; j += 8) { uint64_t w; memcpy(&w, s + j, 8); acc |= w; }
        if (j < n) {                                      // (a read name: one or two words and the word that ends at its last byte)
            if (n >= 8) { uint64_t w; memcpy(&w, s + n - 8, 8); acc |= w; }
            else for (; j < n; j++) acc |= s[j];
        }
        if (!(acc & 0x8080808080808080ull)) return true;
    }
    if (not_ascii) *not_ascii = true;
    int64_t i = 0;
    while (i < n) {
        if (i + 8 <= n) { uint64_t w; memcpy(&w, s + i, 8); if (!(w & 0x8080808080808080ull)) { i += 8; continue; } }
        const unsigned c = s[i];
        if (c < 0x80) { i++; continue; }
        if (c < 0xC2) return false;
        if (c < 0xE0) { if (i + 1 >= n || (s[i + 1] & 0xC0) != 0x80) return false; i += 2; continue; }
        if (c < 0xF0) {
            if (i + 2 >= n || (s[i + 1] & 0xC0) != 0x80 || (s[i + 2] & 0xC0) != 0x80) return false;
            if (c == 0xE0 && s[i + 1] < 0xA0) return false;
            i += 3; continue;
        }
        if (c < 0xF5) {
            if (i + 3 >= n || (s[i + 1] & 0xC0) != 0x80 || (s[i + 2] & 0xC0) != 0x80 || (s[i + 3] & 0xC0) != 0x80) return false;
            if ((c == 0xF0 && s[i + 1] < 0x90) || (c == 0xF4 && s[i + 1] > 0x8F)) return false;
            i += 4; continue;
        }
        return false;
    }
    return true;
}
struct Reader {
    const unsigned char* p; int64_t n, i;
    bool need(int64_t k) const { return i + k <= n; }
    uint64_t le(int k) { uint64_t v = 0; for (int b = 0; b < k; b++) v |= (uint64_t)p[i + b] << (8 * b); i += k; return v; }
};
inline uint32_t le32(const unsigned char* q) { return (uint32_t)q[0] | (uint32_t)q[1] << 8 | (uint32_t)q[2] << 16 | (uint32_t)q[3] << 24; }
inline uint64_t le64(const unsigned char* q) { return (uint64_t)le32(q) | (uint64_t)le32(q + 4) << 32; }

// the memo (and nothing else): plain records in a block that grows by realloc - for the 12 M entries of a HiFi chromosome's reads
// block that is a remap of pages, not the copy a std::vector makes at every doubling
template <class T> struct Pod {
    T* p = nullptr; size_t n = 0, cap = 0;
    ~Pod() { free(p); }
    bool push(const T& x)
    {
        if (n == cap) {
            const size_t c = cap ? cap * 2 : 4096;
            T* q = (T*)realloc(p, c * sizeof(T));
            if (!q) return false;
            p = q; cap = c;
        }
        p[n++] = x;
        return true;
    }
};

// The table's columns ARE the bytes objects pickle_table returns: written in place, grown together (one capacity check per row)
// through _PyBytes_Resize - realloc underneath - in a few steps when the caller knows where the block ends, and cut to the row count.  (r05: a std::vector per column, copied at
// every doubling and once more into its bytes object - 0.3 GB of extra traffic for a 6 M-row reads block.)
struct Columns {
    std::vector<PyObject*> o; std::vector<int> item; std::vector<char*> base;
    size_t cap = 0;
    ~Columns() { for (PyObject* x : o) Py_XDECREF(x); }
    void add(int itemsize) { o.push_back(nullptr); item.push_back(itemsize); base.push_back(nullptr); }
    bool reserve(size_t rows)
    {
        for (size_t k = 0; k < o.size(); k++) {
            if (!o[k]) { o[k] = PyBytes_FromStringAndSize(nullptr, (Py_ssize_t)(rows * (size_t)item[k])); if (!o[k]) return false; }
            else if (_PyBytes_Resize(&o[k], (Py_ssize_t)(rows * (size_t)item[k])) != 0) return false;      // (o[k] is NULL then, the error set)
            base[k] = PyBytes_AS_STRING(o[k]);
        }
        cap = rows;
        return true;
    }
    PyObject* take(size_t k) { PyObject* x = o[k]; o[k] = nullptr; return x; }
};
}   // namespace pk

PyObject* pickle_table(PyObject*, PyObject* args)
{
    PyObject *obuf, *oints, *ostrs; Py_ssize_t offset, width, end_hint = -1;
    if (!PyArg_ParseTuple(args, "OnnO!O!|n", &obuf, &offset, &width, &PyTuple_Type, &oints, &PyTuple_Type, &ostrs, &end_hint)) return nullptr;
    Py_buffer view{};
    if (PyObject_GetBuffer(obuf, &view, PyBUF_SIMPLE) != 0) return nullptr;
    struct Rel { Py_buffer* v; ~Rel() { PyBuffer_Release(v); } } rel{&view};
    if (offset < 0 || offset > view.len) { PyErr_SetString(PyExc_ValueError, "pickle_table: offset outside the buffer"); return nullptr; }
    std::vector<Py_ssize_t> fi, fs;
    for (Py_ssize_t k = 0; k < PyTuple_GET_SIZE(oints); k++) { fi.push_back(PyLong_AsSsize_t(PyTuple_GET_ITEM(oints, k))); }
    for (Py_ssize_t k = 0; k < PyTuple_GET_SIZE(ostrs); k++) { fs.push_back(PyLong_AsSsize_t(PyTuple_GET_ITEM(ostrs, k))); }
    if (PyErr_Occurred()) return nullptr;
    for (Py_ssize_t f : fi) if (f < 0) { PyErr_SetString(PyExc_ValueError, "pickle_table: negative field"); return nullptr; }
    for (Py_ssize_t f : fs) if (f < 0) { PyErr_SetString(PyExc_ValueError, "pickle_table: negative field"); return nullptr; }

    using namespace pk;
    Reader R{(const unsigned char*)view.buf, (int64_t)view.len, (int64_t)offset};
    std::vector<Cell> st;
    Pod<Cell> memo;
    Columns C;                                                           // int fields (int64), then (offset int64, length int32) per string field
    const size_t NI = fi.size(), NS = fs.size();
    for (size_t k = 0; k < NI; k++) C.add(8);
    for (size_t k = 0; k < NS; k++) { C.add(8); C.add(4); }
    if (!C.reserve(1024)) return nullptr;
    int64_t n_rows = 0;
    bool done = false, unsupported = false, oom = false, any_non_ascii = false;       // (all-ASCII text: len() of a string is its byte count)
    const char* corrupt = nullptr;
    auto memo_put = [&](uint64_t k) {
        if (st.empty()) { corrupt = "memo of an empty stack"; return; }
        if (k > memo.n) { unsupported = true; return; }                              // (pickle numbers its memo densely; anything else: not ours)
        if (k == memo.n) { if (!memo.push(st.back())) oom = true; } else memo.p[(size_t)k] = st.back();
    };
    // the fields x[0 .. c) of one tuple -> row n_rows of the columns; false: not a row of ours (nothing is written then)
    constexpr size_t MAXF = 16;
    auto put_row = [&](const Cell* x, size_t c) -> bool {
        if (NI > MAXF || NS > MAXF) return false;
        int64_t iv[MAXF];
        for (size_t k = 0; k < NI; k++) {
            if ((size_t)fi[k] >= c) return false;
            const Cell& y = x[(size_t)fi[k]];
            if (y.kind == INT) iv[k] = y.a;
            else if (y.kind == FLOAT) {
                double d; memcpy(&d, &y.a, 8);
                if (!(d > -9.2e18 && d < 9.2e18)) return false;                       // (nan / inf / beyond int64: let pickle + int() say it)
                iv[k] = (int64_t)d;                                                   // int(x): truncation toward zero
            } else return false;
        }
        for (size_t k = 0; k < NS; k++) if ((size_t)fs[k] >= c || x[(size_t)fs[k]].kind != STR) return false;
        if ((size_t)n_rows == C.cap) {
            // full: twice the rows - or, when the caller has said where the block ends (the next offset of the index), the rows the
            // rest of it will hold at the bytes per row seen so far (+ 3 %), in steps of at most eight times what is already there:
            // a few growths instead of one per doubling, and a hint that is wrong (the end of a 50 GB file for a block in its
            // middle) can ask for eight times too much address space, never for more than the machine has
            size_t want = C.cap * 2;
            if (end_hint > R.i && end_hint <= R.n && n_rows >= 1024 && R.i > (int64_t)offset) {
                const double per_row = (double)(R.i - (int64_t)offset) / (double)n_rows;
                const double est = (double)n_rows + (double)(end_hint - R.i) / per_row * 1.03 + 1024.0;
                if (est > (double)want) want = est < (double)(C.cap * 8) ? (size_t)est : C.cap * 8;
            }
            if (!C.reserve(want)) { oom = true; return false; }                       // (what the block needs is what doubling would reach, too)
        }
        for (size_t k = 0; k < NI; k++) ((int64_t*)C.base[k])[n_rows] = iv[k];
        for (size_t k = 0; k < NS; k++) {
            const Cell& y = x[(size_t)fs[k]];
            ((int64_t*)C.base[NI + 2 * k])[n_rows] = y.a;
            ((int32_t*)C.base[NI + 2 * k + 1])[n_rows] = y.len;
        }
        return true;
    };
    auto make_row = [&](size_t first) {              // the cells st[first ..] are the fields of one tuple
        const size_t c = st.size() - first;
        // a row must be an element of THE list: [LIST, row] (APPEND form) or [LIST, MARKER, row, row, ...] (APPENDS batches)
        const bool placed = (first == 1 && st[0].kind == LIST) ||
                            (first >= 2 && st[0].kind == LIST && st[1].kind == MARKER && (first == 2 || st[first - 1].kind == ROW));
        if (!placed || (width >= 0 && (Py_ssize_t)c != width) || !put_row(st.data() + first, c)) { unsupported = true; return; }
        st.resize(first);
        st.push_back(Cell{ROW, 0, n_rows});
        n_rows++;
    };
    auto top_marker = [&]() -> long {
        for (long q = (long)st.size() - 1; q >= 0; q--) if (st[(size_t)q].kind == MARKER) return q;
        return -1;
    };
    // One row in one go: MARK <fields> TUPLE as the pickler writes an element of the list - the whole of a task's stream but
    // for a few bytes per batch of 1000.  Same opcodes, same checks as the loop below, the fields in a local array instead of on
    // the stack, one bounds check per opcode (16 bytes of slack) instead of one per operand.  Anything else - an opcode the rows
    // of the main script's lists do not have, the last bytes of the buffer, a row the columns cannot take - and the row is handed
    // back untouched (false: R.i and the memo as they were) for the loop below to read, and to judge.
    auto fast_row = [&]() -> bool {
        const unsigned char* const p = R.p;
        const int64_t n = R.n;
        int64_t i = R.i;
        const size_t memo0 = memo.n;
        Cell f[MAXF];
        size_t nf = 0;
        for (;;) {
            if (n - i < 16) break;
            const unsigned char op = p[i++];
            Cell c;
            switch (op) {
            case 'K': c = Cell{INT, 0, (int64_t)p[i]}; i += 1; goto field;
            case 'M': c = Cell{INT, 0, (int64_t)(p[i] | p[i + 1] << 8)}; i += 2; goto field;
            case 'J': c = Cell{INT, 0, (int64_t)(int32_t)le32(p + i)}; i += 4; goto field;
            case 0x8a: {                                                                                                                     // LONG1
                const int k = p[i++];
                if (k > 8) goto bail;
                uint64_t v = 0;
                for (int b = 0; b < k; b++) v |= (uint64_t)p[i + b] << (8 * b);
                i += k;
                if (k && k < 8 && (v >> (8 * k - 1)) & 1) v |= ~0ull << (8 * k);
                c = Cell{INT, 0, (int64_t)v};
                goto field;
            }
            case 'G': c = Cell{FLOAT, 0, (int64_t)__builtin_bswap64(le64(p + i))}; i += 8; goto field;                                       // BINFLOAT (big endian)
            case 0x88: c = Cell{INT, 0, 1}; goto field;
            case 0x89: c = Cell{INT, 0, 0}; goto field;
            case 'N': c = Cell{OTHER, 0, 0}; goto field;
            case 0x8c: case 'X': case 0x8d: {
                uint64_t len;
                if (op == 0x8c) { len = p[i]; i += 1; } else if (op == 'X') { len = le32(p + i); i += 4; } else { len = le64(p + i); i += 8; }
                if (len > (uint64_t)INT32_MAX || (uint64_t)(n - i) < len) goto bail;
                if (!utf8_ok(p + i, (int64_t)len, &any_non_ascii)) goto bail;
                c = Cell{STR, (int32_t)len, i};
                i += (int64_t)len;
                goto field;
            }
            case 0x94: if (!nf) goto bail; if (!memo.push(f[nf - 1])) { oom = true; goto bail; } continue;                                  // MEMOIZE
            case 'q': case 'r': {                                                                                                            // (LONG_)BINPUT
                uint64_t k;
                if (op == 'q') { k = p[i]; i += 1; } else { k = le32(p + i); i += 4; }
                if (!nf || k != memo.n) goto bail;
                if (!memo.push(f[nf - 1])) { oom = true; goto bail; }
                continue;
            }
            case 'h': case 'j': {                                                                                                            // (LONG_)BINGET
                uint64_t id;
                if (op == 'h') { id = p[i]; i += 1; } else { id = le32(p + i); i += 4; }
                if (id >= memo.n) goto bail;
                c = memo.p[(size_t)id];
                if (c.kind == ROW || c.kind == LIST || c.kind == MARKER) goto bail;
                goto field;
            }
            case 0x95: {                                                                                                                     // FRAME
                const uint64_t flen = le64(p + i); i += 8;
                if (flen > (uint64_t)INT64_MAX || (uint64_t)(n - i) < flen) goto bail;
                continue;
            }
            case 't': {
                if ((width >= 0 && (Py_ssize_t)nf != width) || !put_row(f, nf)) goto bail;
                st.push_back(Cell{ROW, 0, n_rows});
                n_rows++;
                R.i = i;
                return true;
            }
            default: goto bail;
            }
        field:
            if (nf == MAXF) goto bail;
            f[nf++] = c;
        }
    bail:
        memo.n = memo0;
        return false;
    };
    while (!done && !unsupported && !corrupt && !oom) {
        if (!R.need(1)) { corrupt = "truncated"; break; }
        const unsigned char op = R.p[R.i++];
        switch (op) {
        case 0x80: if (!R.need(1)) { corrupt = "truncated"; break; } if (R.p[R.i] < 2 || R.p[R.i] > 5) unsupported = true; R.i += 1; break;     // PROTO
        case 0x95: {                                                                                                                         // FRAME
            // pickle reads the whole frame before it goes on (load_frame: "pickle data was truncated" when the announced bytes are
            // not there): a frame that runs past the buffer is not ours to accept
            if (!R.need(8)) { corrupt = "truncated"; break; }
            const uint64_t flen = R.le(8);
            if (flen > (uint64_t)INT64_MAX || !R.need((int64_t)flen)) unsupported = true;
            break;
        }
        case ']': st.push_back(Cell{LIST, 0, 0}); break;
        case '(':
            // the MARK of a row - of the APPEND form, or inside an APPENDS batch - opens the fast path; the MARK of a batch does not
            if (((st.size() == 1 && st[0].kind == LIST) || (st.size() >= 2 && st[0].kind == LIST && st[1].kind == MARKER && (st.size() == 2 || st.back().kind == ROW)))
                && fast_row()) break;
            st.push_back(Cell{MARKER, 0, 0});
            break;
        case 'K': if (!R.need(1)) { corrupt = "truncated"; break; } st.push_back(Cell{INT, 0, (int64_t)R.le(1)}); break;
        case 'M': if (!R.need(2)) { corrupt = "truncated"; break; } st.push_back(Cell{INT, 0, (int64_t)R.le(2)}); break;
        case 'J': if (!R.need(4)) { corrupt = "truncated"; break; } st.push_back(Cell{INT, 0, (int64_t)(int32_t)(uint32_t)R.le(4)}); break;
        case 0x8a: {                                                                                                                         // LONG1
            if (!R.need(1)) { corrupt = "truncated"; break; }
            const int k = R.p[R.i++];
            if (k > 8) { unsupported = true; break; }
            if (!R.need(k)) { corrupt = "truncated"; break; }
            uint64_t v = k ? R.le(k) : 0;
            if (k && k < 8 && (v >> (8 * k - 1)) & 1) v |= ~0ull << (8 * k);             // sign extension
            st.push_back(Cell{INT, 0, (int64_t)v});
            break;
        }
        case 'G': {                                                                                                                          // BINFLOAT (big endian)
            if (!R.need(8)) { corrupt = "truncated"; break; }
            uint64_t v = 0; for (int b = 0; b < 8; b++) v = (v << 8) | R.p[R.i + b];
            R.i += 8;
            st.push_back(Cell{FLOAT, 0, (int64_t)v});
            break;
        }
        case 0x8c: case 'X': case 0x8d: {                                                                                                    // (SHORT_)BINUNICODE(8)
            const int k = op == 0x8c ? 1 : (op == 'X' ? 4 : 8);
            if (!R.need(k)) { corrupt = "truncated"; break; }
            const uint64_t len = R.le(k);
            if (len > (uint64_t)INT32_MAX) { unsupported = true; break; }
            if (!R.need((int64_t)len)) { corrupt = "truncated"; break; }
            // (pickle decodes the payload as it reads it - UTF-8, surrogatepass - and raises on anything else; the spans handed out
            // here are decoded later, or never: the check is made now, 8 bytes at a time for ASCII)
            if (!utf8_ok(R.p + R.i, (int64_t)len, &any_non_ascii)) { unsupported = true; break; }
            st.push_back(Cell{STR, (int32_t)len, R.i});
            R.i += (int64_t)len;
            break;
        }
        case 0x94: memo_put(memo.n); break;                                                                                             // MEMOIZE
        case 'q': if (!R.need(1)) { corrupt = "truncated"; break; } memo_put(R.le(1)); break;                                               // BINPUT
        case 'r': if (!R.need(4)) { corrupt = "truncated"; break; } memo_put(R.le(4)); break;                                               // LONG_BINPUT
        case 'h': case 'j': {                                                                                                                // BINGET / LONG_BINGET
            const int k = op == 'h' ? 1 : 4;
            if (!R.need(k)) { corrupt = "truncated"; break; }
            const uint64_t id = R.le(k);
            if (id >= memo.n) { corrupt = "memo reference before its definition"; break; }
            const Cell c = memo.p[(size_t)id];
            if (c.kind == ROW || c.kind == LIST || c.kind == MARKER) { unsupported = true; break; }     // a shared tuple / list: objects matter
            st.push_back(c);
            break;
        }
        case 0x88: st.push_back(Cell{INT, 0, 1}); break;                                                                                     // NEWTRUE (int(True))
        case 0x89: st.push_back(Cell{INT, 0, 0}); break;
        case 'N': st.push_back(Cell{OTHER, 0, 0}); break;
        case ')': make_row(st.size()); break;                                                                                                // EMPTY_TUPLE
        case 0x85: case 0x86: case 0x87: {
            const size_t c = (size_t)(op - 0x84);
            if (st.size() < c) { corrupt = "tuple of a short stack"; break; }
            make_row(st.size() - c);
            break;
        }
        case 't': {
            const long m = top_marker();
            if (m < 0) { corrupt = "TUPLE without MARK"; break; }
            st.erase(st.begin() + m);                                                    // the fields slide down over their marker
            make_row((size_t)m);
            break;
        }
        case 'e': {                                                                                                                          // APPENDS
            const long m = top_marker();
            if (m != 1 || st[0].kind != LIST) { if (m < 0) corrupt = "APPENDS without MARK"; else unsupported = true; break; }
            for (size_t q = 2; q < st.size(); q++) if (st[q].kind != ROW) unsupported = true;
            st.resize(1);
            break;
        }
        case 'a':                                                                                                                            // APPEND
            if (st.size() != 2 || st[0].kind != LIST || st[1].kind != ROW) { unsupported = true; break; }
            st.resize(1);
            break;
        case '.': if (st.size() != 1 || st[0].kind != LIST) unsupported = true; done = true; break;                                          // STOP
        default: unsupported = true; break;
        }
    }
    if (oom) { if (!PyErr_Occurred()) PyErr_NoMemory(); return nullptr; }
    if (corrupt) { PyErr_Format(PyExc_ValueError, "pickle_table: %s at byte %lld", corrupt, (long long)R.i); return nullptr; }
    if (unsupported) Py_RETURN_NONE;
    if (!C.reserve((size_t)n_rows)) return nullptr;                       // cut to the rows there are
    PyObject* ints = PyList_New((Py_ssize_t)NI);
    PyObject* strs = PyList_New((Py_ssize_t)NS);
    if (!ints || !strs) { Py_XDECREF(ints); Py_XDECREF(strs); return nullptr; }
    for (size_t k = 0; k < NI; k++) PyList_SET_ITEM(ints, (Py_ssize_t)k, C.take(k));
    for (size_t k = 0; k < NS; k++) {
        PyObject* o = C.take(NI + 2 * k);
        PyObject* l = C.take(NI + 2 * k + 1);
        PyObject* t = PyTuple_Pack(2, o, l);
        Py_DECREF(o); Py_DECREF(l);
        if (!t) { Py_DECREF(ints); Py_DECREF(strs); return nullptr; }
        PyList_SET_ITEM(strs, (Py_ssize_t)k, t);
    }
    return Py_BuildValue("(LLNNO)", (long long)n_rows, (long long)R.i, ints, strs, any_non_ascii ? Py_False : Py_True);
}

// read-only views of (buffer, int64 offsets, int32 lengths)
struct SpanArgs {
    Py_buffer buf{}, off{}, len{};
    bool hb = false, ho = false, hl = false;
    ~SpanArgs() { if (hb) PyBuffer_Release(&buf); if (ho) PyBuffer_Release(&off); if (hl) PyBuffer_Release(&len); }
    bool get(PyObject* b, PyObject* o, PyObject* l)
    {
        if (PyObject_GetBuffer(b, &buf, PyBUF_SIMPLE) != 0) return false;
        hb = true;
        if (PyObject_GetBuffer(o, &off, PyBUF_C_CONTIGUOUS) != 0) return false;
        ho = true;
        if (PyObject_GetBuffer(l, &len, PyBUF_C_CONTIGUOUS) != 0) return false;
        hl = true;
        if (off.itemsize != 8 || len.itemsize != 4 || off.len / 8 != len.len / 4) { PyErr_SetString(PyExc_ValueError, "spans: int64 offsets and int32 lengths of one length are expected"); return false; }
        return true;
    }
    Py_ssize_t n() const { return off.len / 8; }
    bool span(Py_ssize_t i, const char*& p, int32_t& k) const
    {
        const int64_t o = ((const int64_t*)off.buf)[i]; k = ((const int32_t*)len.buf)[i];
        if (o < 0 || k < 0 || o + k > buf.len) { PyErr_Format(PyExc_ValueError, "span %zd lies outside the buffer", i); return false; }
        p = (const char*)buf.buf + o;
        return true;
    }
};

inline uint64_t hash_bytes(const char* p, int32_t n)
{
    uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t)n;
    while (n >= 8) { uint64_t w; memcpy(&w, p, 8); h = (h ^ w) * 0xFF51AFD7ED558CCDull; h ^= h >> 32; p += 8; n -= 8; }
    uint64_t w = 0;
    if (n) memcpy(&w, p, (size_t)n);
    h = (h ^ w) * 0xC4CEB9FE1A85EC53ull;
    return h ^ (h >> 29);
}

PyObject* span_intern(PyObject*, PyObject* args)
{
    PyObject* specs;
    if (!PyArg_ParseTuple(args, "O!", &PyTuple_Type, &specs)) return nullptr;
    const Py_ssize_t ns = PyTuple_GET_SIZE(specs);
    std::vector<SpanArgs> sp((size_t)ns);
    std::vector<Buf> ids((size_t)ns);
    Py_ssize_t total = 0;
    for (Py_ssize_t k = 0; k < ns; k++) {
        PyObject *b, *o, *l, *out;
        if (!PyArg_ParseTuple(PyTuple_GET_ITEM(specs, k), "OOOO", &b, &o, &l, &out) || !sp[(size_t)k].get(b, o, l)) return nullptr;
        if (!ids[(size_t)k].get(out, sp[(size_t)k].n(), "span_intern ids") || ids[(size_t)k].v.itemsize != 4) { if (!PyErr_Occurred()) PyErr_SetString(PyExc_ValueError, "span_intern: ids must be int32"); return nullptr; }
        total += sp[(size_t)k].n();
    }
    size_t cap = 64;
    while (cap < (size_t)total * 2 + 8) cap <<= 1;
    // a slot keeps 32 bits of the hash beside the id: a probe that passes somebody else's entry never leaves the table
    struct Slot { uint32_t tag; int32_t id; };
    struct U { const char* p; int32_t n; };
    // the scratch outlives the call (one call at a time, under the interpreter lock): a worker interns task after task of about
    // the same size, and fresh pages cost more than the probes (1.4 us a page on the pool's virtual machines; ~5 MB a task).
    // What has grown beyond 64 MB is given back at the end.
    static std::vector<Slot> slot;
    static std::vector<U> uniq;
    static std::vector<uint64_t> hs;
    struct Trim {
        ~Trim()
        {
            if (slot.capacity() * sizeof(Slot) > (64u << 20)) std::vector<Slot>().swap(slot);
            if (uniq.capacity() * sizeof(U) > (64u << 20)) std::vector<U>().swap(uniq);
            if (hs.capacity() * 8 > (64u << 20)) std::vector<uint64_t>().swap(hs);
        }
    } trim;
    slot.assign(cap, Slot{0, -1});
    uniq.clear();
    int64_t blob_bytes = 0;
    for (Py_ssize_t k = 0; k < ns; k++) {
        const SpanArgs& S = sp[(size_t)k];
        int32_t* out = (int32_t*)ids[(size_t)k].v.buf;
        const Py_ssize_t m = S.n();
        // Two passes, each with its misses asked for ahead of time: the hashes (the payloads - the kept reads of a task lie
        // scattered over a 0.2 GB file - fetched 12 names ahead), then the table (a name's first slot fetched 8 names ahead).
        // One name at a time, miss after miss, this was 60-75 ns a name; a task of a 30x genome has 10^5 of them.
        hs.resize((size_t)m);
        for (Py_ssize_t i = 0; i < m; i++) {
            if (i + 12 < m) {
                const int64_t o = ((const int64_t*)S.off.buf)[i + 12];
                if (o >= 0 && o < S.buf.len) __builtin_prefetch((const char*)S.buf.buf + o);
            }
            const char* p; int32_t n;
            if (!S.span(i, p, n)) return nullptr;
            hs[(size_t)i] = hash_bytes(p, n);
        }
        const char* prev_p = nullptr; int32_t prev_n = -1, prev_id = -1;
        for (Py_ssize_t i = 0; i < m; i++) {
            if (i + 8 < m) __builtin_prefetch(&slot[(size_t)(hs[(size_t)i + 8] >> 7) & (cap - 1)]);
            const char* p; int32_t n;
            if (!S.span(i, p, n)) return nullptr;
            // the SAME bytes as the row before - a memo reference of the pickle: the chromosome of every row of a reads block, the
            // "DEL" of every signature - need no probe
            if (p == prev_p && n == prev_n) { out[i] = prev_id; continue; }
            const uint64_t h = hs[(size_t)i];
            const uint32_t tag = (uint32_t)(h >> 32);
            size_t q = (size_t)(h >> 7) & (cap - 1);
            int32_t id = -1;
            for (;; q = (q + 1) & (cap - 1)) {
                const Slot e = slot[q];
                if (e.id < 0) break;
                if (e.tag != tag) continue;
                const U& x = uniq[(size_t)e.id];
                if (x.n == n && memcmp(x.p, p, (size_t)n) == 0) { id = e.id; break; }
            }
            if (id < 0) {
                if (uniq.size() >= (size_t)INT32_MAX) { PyErr_SetString(PyExc_OverflowError, "span_intern: more than 2^31 distinct values"); return nullptr; }
                id = (int32_t)uniq.size();
                slot[q] = Slot{tag, id};
                uniq.push_back(U{p, n});
                blob_bytes += n;
            }
            out[i] = id;
            prev_p = p; prev_n = n; prev_id = id;
        }
    }
    PyObject* blob = PyBytes_FromStringAndSize(nullptr, (Py_ssize_t)blob_bytes);
    PyObject* off = PyBytes_FromStringAndSize(nullptr, (Py_ssize_t)(uniq.size() * 8));
    PyObject* len = PyBytes_FromStringAndSize(nullptr, (Py_ssize_t)(uniq.size() * 4));
    if (!blob || !off || !len) { Py_XDECREF(blob); Py_XDECREF(off); Py_XDECREF(len); return nullptr; }
    char* d = PyBytes_AS_STRING(blob);
    int64_t* po = (int64_t*)PyBytes_AS_STRING(off);
    int32_t* pl = (int32_t*)PyBytes_AS_STRING(len);
    int64_t at = 0;
    for (size_t u = 0; u < uniq.size(); u++) {
        memcpy(d + at, uniq[u].p, (size_t)uniq[u].n);
        po[u] = at; pl[u] = uniq[u].n; at += uniq[u].n;
    }
    return Py_BuildValue("(NNN)", blob, off, len);
}

// reads_near(pos1 int64, pos2 int64 | None, r_start int64, r_end int64, margin, shift) -> None | bytes (one 0 / 1 byte per read)
// Which reads of a chromosome's block can cover a genotyping window of ONE task (columns.SigStore.from_task_pickles, where the
// argument is made: every window lies in the union of [x - margin, x + margin] over the task's signature coordinates x, and a read
// that covers a window intersects that union).  The union as a flag per 2^shift-bp bin, a running count of flagged bins, and one
// subtraction per read - a pass over the signatures and a pass over the block instead of the dozen numpy passes (and their
// temporaries: 240 ms over the 48 tasks of a HiFi genome) this replaces.  None: nothing to decide on (no signatures, no reads),
// or coordinates outside [0, 2^40) - the caller keeps every read then.
struct I64View {
    Py_buffer v{}; bool held = false;
    ~I64View() { if (held) PyBuffer_Release(&v); }
    bool get(PyObject* o, const char* what)
    {
        if (PyObject_GetBuffer(o, &v, PyBUF_C_CONTIGUOUS | PyBUF_FORMAT) != 0) return false;
        held = true;
        if (v.itemsize != 8 || !v.format || (strcmp(v.format, "l") != 0 && strcmp(v.format, "q") != 0)) { PyErr_Format(PyExc_ValueError, "reads_near: %s must be a contiguous int64 array", what); return false; }
        return true;
    }
    const int64_t* p() const { return (const int64_t*)v.buf; }
    Py_ssize_t n() const { return v.len / 8; }
};

PyObject* reads_near(PyObject*, PyObject* args)
{
    PyObject *o1, *o2, *os, *oe; long long margin; int shift;
    if (!PyArg_ParseTuple(args, "OOOOLi", &o1, &o2, &os, &oe, &margin, &shift)) return nullptr;
    I64View x1, x2, rs, re;
    if (!x1.get(o1, "pos1") || (o2 != Py_None && !x2.get(o2, "pos2")) || !rs.get(os, "r_start") || !re.get(oe, "r_end")) return nullptr;
    if (rs.n() != re.n()) { PyErr_SetString(PyExc_ValueError, "reads_near: r_start and r_end differ in length"); return nullptr; }
    if (margin < 0 || shift < 0 || shift > 30) { PyErr_SetString(PyExc_ValueError, "reads_near: margin >= 0 and 0 <= shift <= 30"); return nullptr; }
    const Py_ssize_t nx = x1.n(), nr = rs.n();
    if (nx == 0 || nr == 0) Py_RETURN_NONE;
    int64_t xmin = INT64_MAX, xmax = INT64_MIN, smin = INT64_MAX, emax = INT64_MIN;
    for (const I64View* X : {&x1, &x2}) for (Py_ssize_t i = 0; i < (X->held ? X->n() : 0); i++) { const int64_t v = X->p()[i]; xmin = v < xmin ? v : xmin; xmax = v > xmax ? v : xmax; }
    for (Py_ssize_t i = 0; i < nr; i++) { const int64_t s = rs.p()[i], e = re.p()[i]; smin = s < smin ? s : smin; emax = e > emax ? e : emax; }
    const int64_t lim = (int64_t)1 << 40;
    if (xmin < 0 || smin < 0 || xmax >= lim || emax >= lim || margin >= lim) Py_RETURN_NONE;
    const int64_t hi = (xmax > emax ? xmax : emax) + margin + ((int64_t)2 << shift);
    if (hi >= lim) Py_RETURN_NONE;
    const int64_t nb = (hi >> shift) + 2;
    std::vector<int32_t> d;
    try { d.assign((size_t)nb + 1, 0); } catch (...) { return PyErr_NoMemory(); }
    // +1 at the first bin of a flagged range, -1 behind its last (a task has fewer than 2^31 signatures: csv_segment's int32 counts)
    for (const I64View* X : {&x1, &x2}) for (Py_ssize_t i = 0; i < (X->held ? X->n() : 0); i++) {
        const int64_t v = X->p()[i], lo = v - margin;
        d[(size_t)((lo > 0 ? lo : 0) >> shift)] += 1;
        d[(size_t)(((v + margin) >> shift) + 1)] -= 1;
    }
    // d[k] becomes the number of flagged bins in front of bin k
    int32_t open = 0, count = 0;
    for (int64_t k = 0; k < nb; k++) { open += d[(size_t)k]; d[(size_t)k] = count; count += open > 0; }
    d[(size_t)nb] = count;
    PyObject* out = PyBytes_FromStringAndSize(nullptr, nr);
    if (!out) return nullptr;
    unsigned char* m = (unsigned char*)PyBytes_AS_STRING(out);
    for (Py_ssize_t i = 0; i < nr; i++) {
        const int64_t s = rs.p()[i], e = re.p()[i] > s ? re.p()[i] : s;
        int64_t b0 = s >> shift, b1 = e >> shift;
        b0 = b0 < nb - 1 ? b0 : nb - 1; b1 = b1 < nb - 1 ? b1 : nb - 1;
        m[i] = d[(size_t)b1 + 1] - d[(size_t)b0] > 0;
    }
    return out;
}

// bytes of the first `ncp` code points of the UTF-8 text p[0 .. n)
inline int32_t utf8_prefix(const char* p, int32_t n, int64_t ncp)
{
    int32_t i = 0;
    while (i < n && ncp > 0) { i++; while (i < n && ((unsigned char)p[i] & 0xC0) == 0x80) i++; ncp--; }
    return i;
}
inline bool is_ascii(const char* p, int32_t n)
{
    uint64_t acc = 0;
    while (n >= 8) { uint64_t w; memcpy(&w, p, 8); acc |= w; p += 8; n -= 8; }
    while (n > 0) { acc |= (unsigned char)*p++; n--; }
    return (acc & 0x8080808080808080ull) == 0;
}
inline int64_t utf8_cplen(const char* p, int32_t n)
{
    if (is_ascii(p, n)) return n;
    int64_t c = 0;
    for (int32_t i = 0; i < n; i++) c += ((unsigned char)p[i] & 0xC0) != 0x80;
    return c;
}

PyObject* span_join(PyObject*, PyObject* args)
{
    PyObject *b, *o, *l, *opicks, *oclips, *oout;
    if (!PyArg_ParseTuple(args, "OOOOOO", &b, &o, &l, &opicks, &oclips, &oout)) return nullptr;
    SpanArgs S;
    if (!S.get(b, o, l)) return nullptr;
    Py_buffer pk{}, cl{};
    if (PyObject_GetBuffer(opicks, &pk, PyBUF_C_CONTIGUOUS) != 0) return nullptr;
    struct R1 { Py_buffer* v; ~R1() { PyBuffer_Release(v); } } r1{&pk};
    const bool clipped = oclips != Py_None;
    if (clipped && PyObject_GetBuffer(oclips, &cl, PyBUF_C_CONTIGUOUS) != 0) return nullptr;
    struct R2 { Py_buffer* v; bool on; ~R2() { if (on) PyBuffer_Release(v); } } r2{&cl, clipped};
    if (pk.itemsize != 8 || (clipped && (cl.itemsize != 8 || cl.len != pk.len))) { PyErr_SetString(PyExc_ValueError, "span_join: picks / clips must be int64 buffers of one length"); return nullptr; }
    const Py_ssize_t n = pk.len / 8;
    Buf out;
    if (!out.get(oout, n, "span_join out_len") || out.v.itemsize != 8) { if (!PyErr_Occurred()) PyErr_SetString(PyExc_ValueError, "span_join: out_len must be int64"); return nullptr; }
    const int64_t* picks = (const int64_t*)pk.buf;
    const int64_t* clips = clipped ? (const int64_t*)cl.buf : nullptr;
    int64_t* ol = (int64_t*)out.v.buf;
    struct Piece { const char* p; int32_t n; };
    std::vector<Piece> pieces((size_t)n);
    int64_t total = 0;
    for (Py_ssize_t i = 0; i < n; i++) {
        if (picks[i] < 0 || picks[i] >= S.n()) { PyErr_Format(PyExc_IndexError, "span_join: index %lld outside the table", (long long)picks[i]); return nullptr; }
        const char* p; int32_t k;
        if (!S.span((Py_ssize_t)picks[i], p, k)) return nullptr;
        if (clipped) {                                                                   // the Python slice [:clip] (code points)
            int64_t end = clips[i];
            if (end < 0 || !is_ascii(p, k)) {
                const int64_t cps = utf8_cplen(p, k);
                if (end < 0) end += cps;
                if (end < 0) end = 0;
                k = utf8_prefix(p, k, end);
            } else if (end < k) k = (int32_t)end;
        }
        pieces[(size_t)i] = Piece{p, k};
        ol[i] = k; total += k;
    }
    PyObject* res = PyBytes_FromStringAndSize(nullptr, (Py_ssize_t)total);
    if (!res) return nullptr;
    char* d = PyBytes_AS_STRING(res);
    for (const Piece& q : pieces) { memcpy(d, q.p, (size_t)q.n); d += q.n; }
    return res;
}

PyObject* span_cplen(PyObject*, PyObject* args)
{
    PyObject *b, *o, *l, *oout;
    if (!PyArg_ParseTuple(args, "OOOO", &b, &o, &l, &oout)) return nullptr;
    SpanArgs S;
    if (!S.get(b, o, l)) return nullptr;
    Buf out;
    if (!out.get(oout, S.n(), "span_cplen out") || out.v.itemsize != 4) { if (!PyErr_Occurred()) PyErr_SetString(PyExc_ValueError, "span_cplen: out must be int32"); return nullptr; }
    int32_t* d = (int32_t*)out.v.buf;
    for (Py_ssize_t i = 0; i < S.n(); i++) {
        const char* p; int32_t k;
        if (!S.span(i, p, k)) return nullptr;
        d[i] = (int32_t)utf8_cplen(p, k);
    }
    Py_RETURN_NONE;
}

PyMethodDef kMethods[] = {
    {"walk", walk, METH_VARARGS, "walk(seq, ints, interns, lens): fill column buffers from a list of tuples"},
    {"intern", intern, METH_VARARGS, "intern(((seq, field, int32 buffer), ...)) -> distinct values by first appearance; ids into the buffers"},
    {"column", column, METH_VARARGS, "column(seq, field) -> [x[field] for x in seq]"},
    {"pickle_table", pickle_table, METH_VARARGS, "pickle_table(buf, offset, width, int_fields, str_fields) -> None | (n, end, [int64 bytes], [(off bytes, len bytes)], all_ascii)"},
    {"span_intern", span_intern, METH_VARARGS, "span_intern(((buf, off, len, ids), ...)) -> (blob, off, len) of the distinct strings by first appearance"},
    {"reads_near", reads_near, METH_VARARGS, "reads_near(pos1, pos2 | None, r_start, r_end, margin, shift) -> None | bytes: which reads can cover a window near a signature"},
    {"span_join", span_join, METH_VARARGS, "span_join(buf, off, len, picks, clips | None, out_len) -> bytes"},
    {"span_cplen", span_cplen, METH_VARARGS, "span_cplen(buf, off, len, out int32): len() of every string"},
    {"clip_join", clip_join, METH_VARARGS, "clip_join(table, picks, lens, out_len) -> bytes of table[picks[i]][:lens[i]] joined"},
    {nullptr, nullptr, 0, nullptr}};
PyModuleDef kModule = {PyModuleDef_HEAD_INIT, "_cols_native", "task lists -> flat columns (cutesv_amd/columns.py)", -1, kMethods, nullptr, nullptr, nullptr, nullptr};

}   // namespace

PyMODINIT_FUNC PyInit__cols_native(void) { return PyModule_Create(&kModule); }
