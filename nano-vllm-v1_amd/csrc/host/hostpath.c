/* Host-side batch flattening in C: the per-step tensors of the reference's ModelRunner.prepare_model_input
 * (engine/model_runner.py:438-495) built straight from the Sequence objects, without the per-token and
 * per-sequence Python loops.  Same outputs as nanovllm.engine.model_runner._flatten_batch_py (the tests
 * compare the two); used when this module has been built, the Python version otherwise.
 *
 * flatten_batch(seqs, block_size) -> dict with
 *   input_ids, positions  int64 [T]        cu_seqlens_q, cu_seqlens_k  int32 [B+1]
 *   slot_mapping          int32 [S]        context_lens int32 [B]      need int32 [B']
 *   block_tables          int32 [B, W] (-1 padded) or None             max_seqlen_q, max_seqlen_k  int
 */
#define PY_SSIZE_T_CLEAN
#define NPY_NO_DEPRECATED_API NPY_1_7_API_VERSION
#include <Python.h>
#include <numpy/arrayobject.h>
#include <stdint.h>
#include <stdlib.h>

static PyObject *s_cached, *s_new, *s_table, *s_tokens;

static int get_long_attr(PyObject* o, PyObject* name, long* out) {
    PyObject* v = PyObject_GetAttr(o, name);
    if (!v) return -1;
    *out = PyLong_AsLong(v);
    Py_DECREF(v);
    return (*out == -1 && PyErr_Occurred()) ? -1 : 0;
}

static PyObject* new_array(int nd, npy_intp* dims, int type) { return PyArray_SimpleNew(nd, dims, type); }

static PyObject* flatten_batch(PyObject* self, PyObject* args) {
    PyObject* seqs_in;
    long bs = 256;
    if (!PyArg_ParseTuple(args, "O|l", &seqs_in, &bs)) return NULL;
    PyObject* seqs = PySequence_Fast(seqs_in, "seqs must be a sequence");
    if (!seqs) return NULL;
    const Py_ssize_t n = PySequence_Fast_GET_SIZE(seqs);
    PyObject** items = PySequence_Fast_ITEMS(seqs);

    long* meta = (long*)malloc(sizeof(long) * 4 * (size_t)(n > 0 ? n : 1));       /* cached, new, total, ntab */
    PyObject** tabs = (PyObject**)calloc((size_t)(n > 0 ? n : 1), sizeof(PyObject*));
    PyObject *ids = NULL, *pos = NULL, *cuq = NULL, *cuk = NULL, *slots = NULL, *ctxl = NULL, *need = NULL, *tables = NULL, *out = NULL;
    if (!meta || !tabs) { PyErr_NoMemory(); goto fail; }

    long T = 0, S = 0, width = 0, max_q = 0, max_k = 0, total_k = 0, n_need = 0;
    for (Py_ssize_t i = 0; i < n; ++i) {
        long *m = meta + 4 * i;
        if (get_long_attr(items[i], s_cached, &m[0]) || get_long_attr(items[i], s_new, &m[1])) goto fail;
        Py_ssize_t len = PyObject_Length(items[i]);
        if (len < 0) goto fail;
        m[2] = (long)len;
        tabs[i] = PyObject_GetAttr(items[i], s_table);
        if (!tabs[i]) goto fail;
        Py_ssize_t nt = PyObject_Length(tabs[i]);
        if (nt < 0) goto fail;
        m[3] = (long)nt;
        if (m[0] < 0 || m[1] < 0 || m[0] + m[1] > m[2]) { PyErr_SetString(PyExc_ValueError, "scheduled window outside the sequence"); goto fail; }
        T += m[1];
        if (nt > 0) {
            S += m[1];
            if ((m[0] + m[1] + bs - 1) / bs > nt) { PyErr_SetString(PyExc_IndexError, "block table too short for the scheduled tokens"); goto fail; }
        }
        if (nt > width) width = nt;
        if (m[1] > max_q) max_q = m[1];
        if (m[0] + m[1] > max_k) max_k = m[0] + m[1];
        total_k += m[0] + m[1];
        if (m[2] == m[0] + m[1] && nt > 0) ++n_need;
    }
    {
        npy_intp d1[1], d2[2];
        d1[0] = T; ids = new_array(1, d1, NPY_INT64); pos = new_array(1, d1, NPY_INT64);
        d1[0] = n + 1; cuq = new_array(1, d1, NPY_INT32); cuk = new_array(1, d1, NPY_INT32);
        d1[0] = S; slots = new_array(1, d1, NPY_INT32);
        d1[0] = n; ctxl = new_array(1, d1, NPY_INT32);
        d1[0] = n_need; need = new_array(1, d1, NPY_INT32);
        if (!ids || !pos || !cuq || !cuk || !slots || !ctxl || !need) goto fail;
        if (width > 0) {
            d2[0] = n; d2[1] = width;
            tables = new_array(2, d2, NPY_INT32);
            if (!tables) goto fail;
        }
    }
    {
        int64_t* p_ids = (int64_t*)PyArray_DATA((PyArrayObject*)ids);
        int64_t* p_pos = (int64_t*)PyArray_DATA((PyArrayObject*)pos);
        int32_t* p_cuq = (int32_t*)PyArray_DATA((PyArrayObject*)cuq);
        int32_t* p_cuk = (int32_t*)PyArray_DATA((PyArrayObject*)cuk);
        int32_t* p_slot = (int32_t*)PyArray_DATA((PyArrayObject*)slots);
        int32_t* p_ctx = (int32_t*)PyArray_DATA((PyArrayObject*)ctxl);
        int32_t* p_need = (int32_t*)PyArray_DATA((PyArrayObject*)need);
        int32_t* p_tab = tables ? (int32_t*)PyArray_DATA((PyArrayObject*)tables) : NULL;
        p_cuq[0] = p_cuk[0] = 0;
        long t = 0, s = 0, k = 0;
        for (Py_ssize_t i = 0; i < n; ++i) {
            const long *m = meta + 4 * i;
            const long cached = m[0], nnew = m[1], ctx = m[0] + m[1], nt = m[3];
            p_cuq[i + 1] = (int32_t)(p_cuq[i] + nnew);
            p_cuk[i + 1] = (int32_t)(p_cuk[i] + ctx);
            p_ctx[i] = (int32_t)ctx;
            if (m[2] == ctx && nt > 0) p_need[k++] = (int32_t)i;
            /* block table row */
            int32_t* row = p_tab ? p_tab + i * width : NULL;
            if (row) {
                PyObject* tf = PySequence_Fast(tabs[i], "block_table must be a sequence");
                if (!tf) goto fail;
                PyObject** ti = PySequence_Fast_ITEMS(tf);
                for (long j = 0; j < nt; ++j) {
                    long v = PyLong_AsLong(ti[j]);
                    if (v == -1 && PyErr_Occurred()) { Py_DECREF(tf); goto fail; }
                    row[j] = (int32_t)v;
                }
                Py_DECREF(tf);
                for (long j = nt; j < width; ++j) row[j] = -1;
            }
            if (nnew == 0) continue;
            /* token ids of the scheduled window */
            PyObject* toks = PyObject_GetAttr(items[i], s_tokens);
            if (!toks) goto fail;
            if (PyList_CheckExact(toks)) {
                if (PyList_GET_SIZE(toks) < ctx) { Py_DECREF(toks); PyErr_SetString(PyExc_IndexError, "token_ids shorter than the scheduled window"); goto fail; }
                for (long j = 0; j < nnew; ++j) {
                    long v = PyLong_AsLong(PyList_GET_ITEM(toks, cached + j));
                    if (v == -1 && PyErr_Occurred()) { Py_DECREF(toks); goto fail; }
                    p_ids[t + j] = v;
                }
            } else {                                            /* windowed view on tensor-parallel workers */
                PyObject *lo = PyLong_FromLong(cached), *hi = PyLong_FromLong(ctx);
                PyObject* sl = (lo && hi) ? PySlice_New(lo, hi, NULL) : NULL;
                Py_XDECREF(lo); Py_XDECREF(hi);
                PyObject* part = sl ? PyObject_GetItem(toks, sl) : NULL;
                Py_XDECREF(sl);
                PyObject* pf = part ? PySequence_Fast(part, "token window must be a sequence") : NULL;
                Py_XDECREF(part);
                if (!pf || PySequence_Fast_GET_SIZE(pf) != nnew) {
                    if (pf && !PyErr_Occurred()) PyErr_SetString(PyExc_IndexError, "token window has the wrong length");
                    Py_XDECREF(pf); Py_DECREF(toks); goto fail;
                }
                for (long j = 0; j < nnew; ++j) {
                    long v = PyLong_AsLong(PySequence_Fast_GET_ITEM(pf, j));
                    if (v == -1 && PyErr_Occurred()) { Py_DECREF(pf); Py_DECREF(toks); goto fail; }
                    p_ids[t + j] = v;
                }
                Py_DECREF(pf);
            }
            Py_DECREF(toks);
            for (long j = 0; j < nnew; ++j) p_pos[t + j] = cached + j;
            if (nt > 0)
                for (long j = 0; j < nnew; ++j) {
                    const long p = cached + j;
                    p_slot[s++] = (int32_t)((long)row[p / bs] * bs + p % bs);
                }
            t += nnew;
        }
    }
    out = Py_BuildValue("{s:O,s:O,s:O,s:O,s:l,s:l,s:O,s:O,s:O,s:O}", "input_ids", ids, "positions", pos, "cu_seqlens_q", cuq,
                        "cu_seqlens_k", cuk, "max_seqlen_q", max_q, "max_seqlen_k", max_k, "slot_mapping", slots, "context_lens", ctxl,
                        "need", need, "block_tables", (tables && total_k > T) ? tables : Py_None);
fail:
    if (tabs) for (Py_ssize_t i = 0; i < n; ++i) Py_XDECREF(tabs[i]);
    free(tabs); free(meta);
    Py_XDECREF(ids); Py_XDECREF(pos); Py_XDECREF(cuq); Py_XDECREF(cuk); Py_XDECREF(slots); Py_XDECREF(ctxl); Py_XDECREF(need); Py_XDECREF(tables);
    Py_DECREF(seqs);
    return out;
}

static PyMethodDef methods[] = {{"flatten_batch", flatten_batch, METH_VARARGS, "Per-step input tensors from scheduled sequences."},
                                {NULL, NULL, 0, NULL}};
static struct PyModuleDef module = {PyModuleDef_HEAD_INIT, "_hostpath", "C helpers of the runner's host path", -1, methods};

PyMODINIT_FUNC PyInit__hostpath(void) {
    import_array();
    s_cached = PyUnicode_InternFromString("num_cached_tokens");
    s_new = PyUnicode_InternFromString("num_new_tokens");
    s_table = PyUnicode_InternFromString("block_table");
    s_tokens = PyUnicode_InternFromString("token_ids");
    return PyModule_Create(&module);
}
