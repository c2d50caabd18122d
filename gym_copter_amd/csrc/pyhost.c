/* _cs_call: the per-step call of the Python host without ctypes' per-call marshalling.
 *
 * CopterVecEnv.step() is host-bound when launched eagerly (the kernel takes 4 us); building seven
 * ctypes argument objects and going through libffi costs about 1.5 us of that.  This module calls the
 * SAME C-ABI entry point, cs_step (include/copterstep.h), through its address: every argument arrives
 * as a Python int (context handle, device pointers, hipStream_t) and is passed on unchanged.  It links
 * neither libcopterstep nor HIP; it computes nothing. */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>

typedef int (*cs_step_fn)(void* ctx, const float* actions, float* obs, float* reward, uint8_t* terminated,
                          uint8_t* truncated, void* stream);

static int as_ptr(PyObject* o, void** out) {
  if (o == Py_None) {
    *out = NULL;
    return 0;
  }
  unsigned long long v = PyLong_AsUnsignedLongLong(o);
  if (v == (unsigned long long)-1 && PyErr_Occurred()) return -1;
  *out = (void*)(uintptr_t)v;
  return 0;
}

/* step(fn, ctx, actions, obs, reward, terminated, truncated, stream) -> status code of cs_step */
static PyObject* call_step(PyObject* self, PyObject* const* args, Py_ssize_t nargs) {
  void* p[8];
  (void)self;
  if (nargs != 8) {
    PyErr_SetString(PyExc_TypeError, "step() takes 8 integer arguments");
    return NULL;
  }
  for (int i = 0; i < 8; ++i)
    if (as_ptr(args[i], &p[i]) != 0) return NULL;
  if (p[0] == NULL || p[1] == NULL) {
    PyErr_SetString(PyExc_ValueError, "step(): null entry point or context");
    return NULL;
  }
  /* without the GIL, as ctypes does: hosts that drive distinct contexts from distinct threads (allowed by
     include/copterstep.h) then enqueue their launches concurrently */
  int rc;
  Py_BEGIN_ALLOW_THREADS
  rc = ((cs_step_fn)(uintptr_t)p[0])(p[1], (const float*)p[2], (float*)p[3], (float*)p[4], (uint8_t*)p[5],
                                     (uint8_t*)p[6], p[7]);
  Py_END_ALLOW_THREADS
  return PyLong_FromLong(rc);
}

static PyMethodDef methods[] = {
    {"step", (PyCFunction)(void (*)(void))call_step, METH_FASTCALL,
     "step(fn, ctx, actions, obs, reward, terminated, truncated, stream): call cs_step at address fn"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef module = {PyModuleDef_HEAD_INIT, "_cs_call",
                                    "address-level call of libcopterstep's cs_step", -1, methods,
                                    NULL, NULL, NULL, NULL};

PyMODINIT_FUNC PyInit__cs_call(void) { return PyModule_Create(&module); }
