"""Builds and binds tests/emul/libgm_emul.so: the DEVICE source of the Gaussian fit compiled for the
CPU with one thread per workgroup.  Test infrastructure only (checks the kernel's control flow
against the oracle without a GPU); the shipped library has no CPU path."""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "emul", "gm_emul.cpp")
_SO = os.path.join(_HERE, "emul", "libgm_emul.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        deps = [_SRC, os.path.join(_HERE, "emul", "gm_host.h"), os.path.join(_HERE, "emul", "bm_host.h")] + \
               [os.path.join(_HERE, "..", "pareben_amd", "csrc", f) for f in ("gm_fit.h", "bm_fit.h", "blk.h", "types.h")]
        if not os.path.exists(_SO) or any(os.path.getmtime(d) > os.path.getmtime(_SO) for d in deps):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", _SO, _SRC])
        _lib = C.CDLL(_SO)
    return _lib


def cv_grid(BASIS, y, fold_id, n_folds, alpha, lam, prior="gaussian", epis=False):
    dp, ip, lp = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_int64)
    X = np.asfortranarray(BASIS, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    fid = np.ascontiguousarray(fold_id, dtype=np.int32)
    a = np.ascontiguousarray(alpha, dtype=np.float64); l = np.ascontiguousarray(lam, dtype=np.float64)
    out = np.zeros((len(a), n_folds)); st = np.zeros((len(a), n_folds), dtype=np.int32)
    cnt = np.zeros((len(a), n_folds, 14), dtype=np.int64)
    fn = (lib().emul_gf_cv_grid if epis else lib().emul_gm_cv_grid) if prior == "gaussian" else (lib().emul_bf_cv_grid if epis else lib().emul_bm_cv_grid)
    rc = fn(X.ctypes.data_as(dp), X.shape[0], X.shape[1], y.ctypes.data_as(dp), fid.ctypes.data_as(ip),
                               n_folds, a.ctypes.data_as(dp), l.ctypes.data_as(dp), len(a), out.ctypes.data_as(dp),
                               st.ctypes.data_as(ip), cnt.ctypes.data_as(lp))
    assert rc == 0
    return out, st, cnt
