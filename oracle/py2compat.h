/*
 * oracle/py2compat.h -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Forced-include header (gcc -include) that lets the UNMODIFIED reference
 * translation unit /root/reference/csrc/workhorse.c (a CPython-2 extension)
 * compile against the CPython 3.10 headers of this image.  It contains no
 * algorithmic code: it only renames the handful of CPython-2 spellings the
 * reference uses to their CPython-3 equivalents and provides the module-init
 * glue (recipe: SURVEY.md Appendix A).  Under CPython 3 every string crosses
 * the boundary as bytes.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#define PyString_FromStringAndSize PyBytes_FromStringAndSize
#define PyString_Check PyBytes_Check
#define PyString_AsString PyBytes_AsString
#define PyString_Size PyBytes_Size
#define PyString_FromString PyUnicode_FromString
#define PyInt_FromLong PyLong_FromLong
#define PyInt_AS_LONG PyLong_AsLong
static PyObject *kv_compat_module = NULL;
static struct PyModuleDef kv_compat_def = { PyModuleDef_HEAD_INIT, "kvarq.engine", NULL, -1, NULL };
static PyObject *kv_compat_initmodule(const char *name, PyMethodDef *methods) {
    (void) name;
    kv_compat_def.m_methods = methods;
    kv_compat_module = PyModule_Create(&kv_compat_def);
    PyDict_SetItemString(PyImport_GetModuleDict(), "kvarq.engine", kv_compat_module);
    return kv_compat_module;
}
#define Py_InitModule(name, methods) kv_compat_initmodule(name, methods)
PyMODINIT_FUNC initengine(void);
PyMODINIT_FUNC PyInit_engine(void) { (void) initengine(); return kv_compat_module; }
