"""ctypes binding of oracle/libksw_oracle.so (CPU ORACLE — test infrastructure only).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_DIR = os.path.dirname(_HERE)
_LIB_PATH = os.path.join(_ORACLE_DIR, "libksw_oracle.so")

I32P = C.POINTER(C.c_int32)


def build(force=False):
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-C", _ORACLE_DIR, "libksw_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        ext_args = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                    C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                    I32P, I32P, I32P, I32P, I32P, C.c_int, C.POINTER(C.c_uint64)]
        for name in ("ksw_extend2_ref", "ksw_extend2_rowsync_model"):
            f = getattr(L, name)
            f.argtypes = ext_args
            f.restype = C.c_int
        L.bsw_pair_ref.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.bsw_pair_ref.restype = None
        L.bsw_pair_batch_ref.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]
        L.bsw_pair_batch_ref.restype = None
        L.bsw_pair_batch_avx2.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]
        L.bsw_pair_batch_avx2.restype = None
        L.bsw_ext_batch_ref.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]
        L.bsw_ext_batch_ref.restype = None
        L.ksw_global2_ref.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_int, I32P, C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.c_uint64)]
        L.ksw_global2_ref.restype = C.c_int
        L.ksw_align2_ref.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_int, C.c_void_p, C.POINTER(C.c_uint64)]
        L.ksw_align2_ref.restype = None
        L.ksw_align2_batch_ref.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]
        L.ksw_align2_batch_ref.restype = C.c_uint64
        _lib = L
    return _lib


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a, a.ctypes.data


def extend2(query, target, mat, o_del, e_del, o_ins, e_ins, w, end_bonus, zdrop, h0,
            variant=0, model=False, m=5):
    """One ksw_extend2 call on the oracle (or on the row-synchronous model)."""
    q, qp = _u8(query)
    t, tp = _u8(target)
    mt = np.ascontiguousarray(mat, dtype=np.int8)
    outs = [C.c_int32(0) for _ in range(5)]
    cells = C.c_uint64(0)
    f = lib().ksw_extend2_rowsync_model if model else lib().ksw_extend2_ref
    score = f(len(q), qp, len(t), tp, m, mt.ctypes.data, o_del, e_del, o_ins, e_ins, w, end_bonus,
              zdrop, h0, *[C.byref(o) for o in outs], variant, C.byref(cells))
    return dict(score=score, qle=outs[0].value, tle=outs[1].value, gtle=outs[2].value,
                gscore=outs[3].value, max_off=outs[4].value, cells=cells.value)


def pair_batch(params, tasks, nthreads=1):
    """params: 1-element PARAMS array; tasks: TASK array -> RESULT array."""
    from_dtype = tasks.dtype
    assert from_dtype.itemsize == 72
    import importlib
    host = importlib.import_module("bwa_mem_sw_amd.host")
    out = np.zeros(len(tasks), dtype=host.RESULT)
    lib().bsw_pair_batch_ref(params.ctypes.data, tasks.ctypes.data, len(tasks), out.ctypes.data, nthreads)
    return out


def pair_batch_avx2(params, tasks, nthreads=1):
    """The strong CPU baseline (oracle/ksw_extend_avx2.c: 16 seeds per AVX2 register); same bytes as pair_batch."""
    assert tasks.dtype.itemsize == 72
    import importlib
    host = importlib.import_module("bwa_mem_sw_amd.host")
    out = np.zeros(len(tasks), dtype=host.RESULT)
    lib().bsw_pair_batch_avx2(params.ctypes.data, tasks.ctypes.data, len(tasks), out.ctypes.data, nthreads)
    return out


def ext_batch(params, etasks, nthreads=1):
    import importlib
    host = importlib.import_module("bwa_mem_sw_amd.host")
    out = np.zeros(len(etasks), dtype=host.EXT)
    lib().bsw_ext_batch_ref(params.ctypes.data, etasks.ctypes.data, len(etasks), out.ctypes.data, nthreads)
    return out


_libc = C.CDLL(None)
_libc.free.argtypes = [C.c_void_p]


def global2(query, target, mat, o_del, e_del, o_ins, e_ins, w, want_cigar=True, m=5):
    """bwa ksw_global2 on the oracle.  Returns dict(score, cigar=[(op, len), ...], cells); ops 0=M 1=I 2=D."""
    q, qp = _u8(query)
    t, tp = _u8(target)
    mt = np.ascontiguousarray(mat, dtype=np.int8)
    n = C.c_int32(0)
    cig = C.POINTER(C.c_uint32)()
    cells = C.c_uint64(0)
    score = lib().ksw_global2_ref(len(q), qp, len(t), tp, m, mt.ctypes.data, o_del, e_del, o_ins, e_ins, w,
                                  C.byref(n) if want_cigar else None, C.byref(cig) if want_cigar else None, C.byref(cells))
    ops = [(int(cig[i]) & 0xf, int(cig[i]) >> 4) for i in range(n.value)]
    if want_cigar and cig:
        _libc.free(cig)
    return dict(score=score, cigar=ops, cells=cells.value)


KSW_XBYTE, KSW_XSTOP, KSW_XSUBO, KSW_XSTART = 0x10000, 0x20000, 0x40000, 0x80000
ALIGN_FIELDS = ("score", "te", "qe", "score2", "te2", "tb", "qb")


def align2(query, target, mat, o_del, e_del, o_ins, e_ins, xtra, m=5):
    """bwa ksw_align2 on the oracle (literal emulation of the striped SSE2 code).  Returns dict(score, te, qe, score2,
    te2, tb, qb, cells)."""
    q, qp = _u8(query)
    t, tp = _u8(target)
    mt = np.ascontiguousarray(mat, dtype=np.int8)
    out = np.zeros(7, dtype=np.int32)
    cells = C.c_uint64(0)
    lib().ksw_align2_ref(len(q), qp, len(t), tp, m, mt.ctypes.data, o_del, e_del, o_ins, e_ins, xtra, out.ctypes.data, C.byref(cells))
    d = {k: int(v) for k, v in zip(ALIGN_FIELDS, out)}
    d["cells"] = cells.value
    return d


def align2_batch(mat, o_del, e_del, o_ins, e_ins, atasks, nthreads=1):
    """atasks: structured array with query/target pointers, qlen, tlen, xtra (host.ATASK).  Returns (int32[n][7], cells)."""
    mt = np.ascontiguousarray(mat, dtype=np.int8)
    n = len(atasks)
    out = np.zeros((n, 7), dtype=np.int32)
    qptr = np.ascontiguousarray(atasks["query"], dtype=np.uint64)
    tptr = np.ascontiguousarray(atasks["target"], dtype=np.uint64)
    ql = np.ascontiguousarray(atasks["qlen"], dtype=np.int32)
    tl = np.ascontiguousarray(atasks["tlen"], dtype=np.int32)
    xt = np.ascontiguousarray(atasks["xtra"], dtype=np.int32)
    cells = lib().ksw_align2_batch_ref(mt.ctypes.data, o_del, e_del, o_ins, e_ins, qptr.ctypes.data, tptr.ctypes.data, ql.ctypes.data,
                                       tl.ctypes.data, xt.ctypes.data, n, out.ctypes.data, nthreads)
    return out, int(cells)
