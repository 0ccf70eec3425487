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

PyMethodDef kMethods[] = {
    {"walk", walk, METH_VARARGS, "walk(seq, ints, interns, lens): fill column buffers from a list of tuples"},
    {"intern", intern, METH_VARARGS, "intern(((seq, field, int32 buffer), ...)) -> distinct values by first appearance; ids into the buffers"},
    {"column", column, METH_VARARGS, "column(seq, field) -> [x[field] for x in seq]"},
    {"clip_join", clip_join, METH_VARARGS, "clip_join(table, picks, lens, out_len) -> bytes of table[picks[i]][:lens[i]] joined"},
    {nullptr, nullptr, 0, nullptr}};
PyModuleDef kModule = {PyModuleDef_HEAD_INIT, "_cols_native", "task lists -> flat columns (cutesv_amd/columns.py)", -1, kMethods, nullptr, nullptr, nullptr, nullptr};

}   // namespace

PyMODINIT_FUNC PyInit__cols_native(void) { return PyModule_Create(&kModule); }
