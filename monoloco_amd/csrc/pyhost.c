/* pyhost.c -- host-side marshalling helper of the Python surface (NOT part of the C ABI in include/monoloco_hip.h: it takes a
 * PyObject, so it is loaded through ctypes.PyDLL and called with the GIL held).
 *
 * The reference's API hands keypoints over as nested Python lists ([m][3][17] floats, preprocess_pifpaf's output,
 * monoloco/network/process.py:155-207; consumed by Loco.forward, net.py:92-93 `torch.tensor(keypoints)`).  Turning them into one
 * float32 array is the largest host-side item of a frame (np.asarray 20 us, np.fromiter 13 us for 16 persons); walking the lists
 * with the C API straight into the pinned staging buffer the kernel reads takes ~2 us.
 *
 * ml_py_fill_kps(list, dst, m): returns 0 when `list` is exactly m lists of 3 lists of 17 Python floats / ints and dst (m * 51
 * floats) has been filled; 1 otherwise (nothing is promised about dst; the caller takes its numpy route, which also raises the
 * proper error for ragged input). */
#define PY_SSIZE_T_CLEAN
#include <Python.h>

/* PY_VERSION_HEX of the headers this file was compiled against: the loader refuses the library when the running interpreter is
 * another one (the list / float accessors below are struct-layout macros). */
long ml_py_version_hex(void) { return (long)PY_VERSION_HEX; }

int ml_py_fill_kps(PyObject* list, float* dst, long m) {
    if (!list || !dst || !PyList_CheckExact(list) || PyList_GET_SIZE(list) != m) return 1;
    for (long i = 0; i < m; ++i) {
        PyObject* person = PyList_GET_ITEM(list, i);
        if (!PyList_CheckExact(person) || PyList_GET_SIZE(person) != 3) return 1;
        for (int c = 0; c < 3; ++c) {
            PyObject* row = PyList_GET_ITEM(person, c);
            if (!PyList_CheckExact(row) || PyList_GET_SIZE(row) != 17) return 1;
            float* out = dst + (i * 3 + c) * 17;
            for (int j = 0; j < 17; ++j) {
                PyObject* v = PyList_GET_ITEM(row, j);
                if (PyFloat_CheckExact(v)) {
                    out[j] = (float)PyFloat_AS_DOUBLE(v);
                } else if (PyLong_CheckExact(v)) {
                    const double d = PyLong_AsDouble(v);
                    if (d == -1.0 && PyErr_Occurred()) {
                        PyErr_Clear();
                        return 1;
                    }
                    out[j] = (float)d;
                } else {
                    return 1;
                }
            }
        }
    }
    return 0;
}
