/* bossx_py.h - PRIVATE helper of the ctypes binding (boss-runs_amd/_lib.py).  Not part of the C-ABI
 * (that is include/bossx.h): a reference-side FFI in another language would not bind this. */
#ifndef BOSSX_PY_H
#define BOSSX_PY_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* Pointer + length of the UTF-8 buffer of each str in the Python list `list` (a PyObject*), obtained
 * through the addresses of the interpreter's PyList_GetItem and PyUnicode_AsUTF8AndSize (so this
 * library does not link against libpython); lets the ctypes layer hand a dict of reads to
 * bossx_stage_batch_ptrs without one ctypes call per read.  Call with the GIL held (ctypes.PyDLL). */
int bossx_py_str_pointers(void *list, int64_t n, void *list_get_item, void *as_utf8_and_size,
                          const char **ptrs, int64_t *lens);
/* The same for a dict of str -> str in ONE pass (PyDict_Next): pointers / lengths of the keys' and of the
 * values' UTF-8 buffers, in the dict's iteration order.  Returns the number of items, or a negative
 * BOSSX_E_* code (a key or value that is not a str; more than `cap` items). */
int64_t bossx_py_dict_pointers(void *dict, int64_t cap, void *dict_next, void *as_utf8_and_size,
                               const char **key_ptrs, int64_t *key_lens, const char **val_ptrs, int64_t *val_lens);
#ifdef __cplusplus
}
#endif
#endif
