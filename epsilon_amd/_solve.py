"""ctypes binding of libepsilon_hip.so with the call signatures of `epopt._solve`.

Reference module: python/epopt/solvemodule.cc:265-271

    solve(problem_bytes, parameters, solver_params_bytes, data) -> (status_bytes, {var_id: bytes})
    eval_prox(f_expr_bytes, lam, data, v) -> {var_id: bytes}

`data` / `v` are {str: bytes}; values come back as float64 column-major bytes.  A failed
call raises `_solve.error`, as the reference does through its longjmp handler
(solvemodule.cc:245-248,286-289).  There is no fallback: if the library is not built or no
HIP device is present the call fails.
"""

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libepsilon_hip.so")


class error(Exception):
    """`_solve.error` (reference solvemodule.cc:286-289)."""


class _Blob(ctypes.Structure):
    _fields_ = [("key", ctypes.c_char_p), ("ptr", ctypes.c_void_p),
                ("len", ctypes.c_size_t), ("kind", ctypes.c_int)]


class _Param(ctypes.Structure):
    _fields_ = [("id", ctypes.c_char_p), ("constant_proto", ctypes.c_void_p),
                ("len", ctypes.c_size_t)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise error("libepsilon_hip.so is not built (run `python -m epsilon_amd.build`)")
        L = ctypes.CDLL(LIB_PATH)
        L.eps_last_error.restype = ctypes.c_char_p
        L.eps_version.restype = ctypes.c_char_p
        L.eps_result_num_vars.restype = ctypes.c_size_t
        L.eps_result_num_vars.argtypes = [ctypes.c_void_p]
        L.eps_result_free.argtypes = [ctypes.c_void_p]
        L.eps_result_copy_var.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
        L.eps_result_copy_var.restype = ctypes.c_int
        L.eps_result_free.restype = None
        L.eps_solver_destroy.argtypes = [ctypes.c_void_p]
        L.eps_solver_destroy.restype = None
        L.eps_set_option.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise error(lib().eps_last_error().decode("utf-8", "replace") or "CHECK failed")


def set_option(key, value):
    _check(lib().eps_set_option(key.encode(), str(value).encode()))


def device_count():
    return lib().eps_device_count()


def _blobs(d, keep):
    """{key: bytes | ("device", ptr, count, "f32"|"f64")} -> (array of eps_blob, n)."""
    items = list(d.items())
    arr = (_Blob * max(len(items), 1))()
    for i, (k, v) in enumerate(items):
        kb = k.encode("utf-8")
        keep.append(kb)
        arr[i].key = kb
        if isinstance(v, tuple) and v and v[0] == "device":
            arr[i].ptr = ctypes.c_void_p(v[1])
            arr[i].len = v[2]
            arr[i].kind = 1 if v[3] == "f32" else 2
        else:
            if isinstance(v, np.ndarray):
                v = np.ascontiguousarray(v)
                keep.append(v)
                arr[i].ptr = v.ctypes.data_as(ctypes.c_void_p)
                arr[i].len = v.nbytes
            elif isinstance(v, bytes):
                # the bytes object's own buffer (no copy: a data matrix can be gigabytes; the
                # library reads it during the call or copies what it keeps)
                keep.append(v)
                ref = ctypes.c_char_p(v) if len(v) else None
                keep.append(ref)
                arr[i].ptr = ctypes.cast(ref, ctypes.c_void_p) if ref is not None else None
                arr[i].len = len(v)
            else:
                buf = (ctypes.c_char * len(v)).from_buffer_copy(v) if len(v) else None
                keep.append(buf)
                arr[i].ptr = ctypes.cast(buf, ctypes.c_void_p) if buf is not None else None
                arr[i].len = len(v)
            arr[i].kind = 0
    return arr, len(items)


def _params(parameters, keep):
    items = list(parameters or [])
    arr = (_Param * max(len(items), 1))()
    for i, (pid, cbytes) in enumerate(items):
        pb = pid.encode("utf-8")
        buf = ctypes.create_string_buffer(cbytes, len(cbytes))
        keep.extend([pb, buf])
        arr[i].id = pb
        arr[i].constant_proto = ctypes.cast(buf, ctypes.c_void_p)
        arr[i].len = len(cbytes)
    return arr, len(items)


_BIG_RESULT = 64 << 20
_new_bytes = ctypes.pythonapi.PyBytes_FromStringAndSize
_new_bytes.restype = ctypes.py_object
_new_bytes.argtypes = [ctypes.c_void_p, ctypes.c_ssize_t]
_bytes_ptr = ctypes.pythonapi.PyBytes_AsString
_bytes_ptr.restype = ctypes.c_void_p
_bytes_ptr.argtypes = [ctypes.py_object]


def _take_result(res):
    L = lib()
    try:
        sp = ctypes.c_void_p()
        sl = ctypes.c_size_t()
        L.eps_result_status(res, ctypes.byref(sp), ctypes.byref(sl))
        status = ctypes.string_at(sp, sl.value) if sl.value else b""
        out = {}
        for i in range(L.eps_result_num_vars(res)):
            cid = ctypes.c_char_p()
            vals = ctypes.POINTER(ctypes.c_double)()
            cnt = ctypes.c_size_t()
            L.eps_result_var(res, ctypes.c_size_t(i), ctypes.byref(cid), ctypes.byref(vals),
                             ctypes.byref(cnt))
            nbytes = cnt.value * 8
            if nbytes >= _BIG_RESULT:
                # an uninitialised bytes object filled by the library's host threads before anyone
                # else sees it (the C-API contract of PyBytes_FromStringAndSize(NULL, n)); one
                # thread's string_at of a 0.8 GB iterate costs 0.15-0.2 s
                b = _new_bytes(None, nbytes)
                if L.eps_result_copy_var(res, ctypes.c_size_t(i), _bytes_ptr(b), ctypes.c_size_t(nbytes)) != 0:
                    raise error("eps_result_copy_var failed")
                out[cid.value.decode("utf-8")] = b
            else:
                out[cid.value.decode("utf-8")] = ctypes.string_at(vals, nbytes)
        return status, out
    finally:
        L.eps_result_free(res)


def solve(problem_bytes, parameters, solver_params_bytes, data):
    """reference `_solve.solve` (solvemodule.cc:110-187)."""
    L = lib()
    keep = []
    blobs, nb = _blobs(data, keep)
    params, np_ = _params(parameters, keep)
    res = ctypes.c_void_p()
    _check(L.eps_solve(problem_bytes, ctypes.c_size_t(len(problem_bytes)),
                       solver_params_bytes, ctypes.c_size_t(len(solver_params_bytes)),
                       blobs, ctypes.c_size_t(nb), params, ctypes.c_size_t(np_),
                       ctypes.byref(res)))
    return _take_result(res)


def eval_prox(f_expr_bytes, lam, data, v):
    """reference `_solve.eval_prox` (solvemodule.cc:189-242)."""
    L = lib()
    keep = []
    blobs, nb = _blobs(data, keep)
    vb, nv = _blobs(v, keep)
    res = ctypes.c_void_p()
    _check(L.eps_eval_prox(f_expr_bytes, ctypes.c_size_t(len(f_expr_bytes)),
                           ctypes.c_double(lam), blobs, ctypes.c_size_t(nb), vb,
                           ctypes.c_size_t(nv), ctypes.byref(res)))
    return _take_result(res)[1]


class Solver(object):
    """Live solver handle (include/epsilon_hip.h eps_solver_*): keeps the data matrix, the
    cached factorisation and the iterates in HBM between calls (warm start; staged timing)."""

    def __init__(self, problem_bytes, solver_params_bytes, data):
        L = lib()
        keep = []  # host blobs are copied by the library: nothing to keep alive after the call
        blobs, nb = _blobs(data, keep)
        self._h = ctypes.c_void_p()
        _check(L.eps_solver_create(problem_bytes, ctypes.c_size_t(len(problem_bytes)),
                                   solver_params_bytes,
                                   ctypes.c_size_t(len(solver_params_bytes)), blobs,
                                   ctypes.c_size_t(nb), ctypes.byref(self._h)))

    def set_parameter(self, parameter_id, constant_bytes, data=None):
        keep = []
        blobs, nb = _blobs(data or {}, keep)
        _check(lib().eps_solver_set_parameter(self._h, parameter_id.encode(), constant_bytes,
                                              ctypes.c_size_t(len(constant_bytes)), blobs,
                                              ctypes.c_size_t(nb)))

    def init(self):
        _check(lib().eps_solver_init(self._h))

    def run(self, max_sweeps=-1):
        done = ctypes.c_int()
        _check(lib().eps_solver_run(self._h, ctypes.c_int(max_sweeps), ctypes.byref(done)))
        return done.value

    def result(self):
        res = ctypes.c_void_p()
        _check(lib().eps_solver_result(self._h, ctypes.byref(res)))
        return _take_result(res)

    def timing(self):
        a, b = ctypes.c_double(), ctypes.c_double()
        lib().eps_solver_timing(self._h, ctypes.byref(a), ctypes.byref(b))
        return a.value, b.value

    def close(self):
        if self._h:
            lib().eps_solver_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- sharded solves ---------------------------------------------------------------------------------

ALLREDUCE_FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int,
                                ctypes.c_void_p)
_comm_keep = []


def comm_unique_id():
    buf = ctypes.create_string_buffer(128)
    _check(lib().eps_comm_unique_id(buf))
    return buf.raw


def comm_init_rccl(rank, world, unique_id):
    _check(lib().eps_comm_init_rccl(ctypes.c_int(rank), ctypes.c_int(world), unique_id))


def comm_init_callback(rank, world, fn):
    """fn(numpy_array) must sum the array in place across ranks."""
    def trampoline(ptr, count, dtype, ctx):
        ctype = ctypes.c_float if dtype == 0 else ctypes.c_double
        arr = np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctype)), shape=(count,))
        fn(arr)
    cb = ALLREDUCE_FN(trampoline)
    _comm_keep.append(cb)
    _check(lib().eps_comm_init_callback(ctypes.c_int(rank), ctypes.c_int(world), cb, None))


def comm_warmup(count=1 << 20):
    """Collective: one checked all-reduce + all-gather of `count` floats (barrier + RCCL first-use
    setup)."""
    _check(lib().eps_comm_warmup(ctypes.c_size_t(count)))


def comm_enable_peer(slot_floats=0, rehearse_ranks=0):
    """Collective: set up the one-shot peer-write window (include/epsilon_hip.h).  Returns
    (enabled, reason-if-not)."""
    on = ctypes.c_int(0)
    _check(lib().eps_comm_enable_peer(ctypes.c_size_t(slot_floats), ctypes.c_int(rehearse_ranks),
                                      ctypes.byref(on)))
    return bool(on.value), ("" if on.value else lib().eps_last_error().decode("utf-8", "replace"))


def comm_disable_peer():
    _check(lib().eps_comm_disable_peer())


def comm_shutdown():
    _check(lib().eps_comm_shutdown())
    del _comm_keep[:]


def shard_keys(keys):
    keys = [k.encode("utf-8") for k in keys]
    arr = (ctypes.c_char_p * max(len(keys), 1))(*keys)
    _check(lib().eps_shard_keys(arr, ctypes.c_size_t(len(keys))))


def shard_consensus_terms(on=True):
    _check(lib().eps_shard_consensus_terms(ctypes.c_int(1 if on else 0)))


def block_solve_stats(reset=False):
    """(largest kappa_1 estimate of a pivot block, refinement steps) since the last reset (fp32)."""
    c, r = ctypes.c_double(0), ctypes.c_int(0)
    _check(lib().eps_block_solve_stats(ctypes.byref(c), ctypes.byref(r), ctypes.c_int(1 if reset else 0)))
    return c.value, r.value


def graph_stats(reset=False):
    """(sweeps of the generic operator path replayed from a hipGraph, captures) since the last reset."""
    a, b = ctypes.c_longlong(0), ctypes.c_longlong(0)
    _check(lib().eps_graph_stats(ctypes.byref(a), ctypes.byref(b), ctypes.c_int(1 if reset else 0)))
    return a.value, b.value


def profile_enable(on=True):
    _check(lib().eps_profile_enable(ctypes.c_int(1 if on else 0)))


def profile_reset():
    _check(lib().eps_profile_reset())


def profile_dump():
    """{tag: (count, total_ms)} of the live HIP-event kernel timers."""
    buf = ctypes.create_string_buffer(1 << 16)
    _check(lib().eps_profile_dump(buf, ctypes.c_size_t(len(buf))))
    out = {}
    for line in buf.value.decode().splitlines():
        tag, count, ms = line.rsplit(" ", 2)
        out[tag] = (int(count), float(ms))
    return out


# ---- per-operator entry points ------------------------------------------------------------------


def linear_map_apply(lmap, x, transpose=False, inverse=False):
    """y = A x (or A^T x, or A^-1 x through the explicit inverse map) for an `ir.LMap`."""
    L = lib()
    keep = []
    blobs, nb = _blobs(lmap.data, keep)
    pbytes = lmap.proto.SerializeToString()
    x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1)
    ny = lmap.n if transpose else lmap.m
    y = np.empty(ny, dtype=np.float64)
    _check(L.eps_linear_map_apply(pbytes, ctypes.c_size_t(len(pbytes)), blobs,
                                  ctypes.c_size_t(nb),
                                  ctypes.c_int((1 if transpose else 0) | (2 if inverse else 0)),
                                  x.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(x.size),
                                  y.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(ny)))
    return y


def linear_map_binary(op, A, B, ta=False, tb=False):
    """(result ImplType, dense m x n values) of A op B through the dispatch tables."""
    L = lib()
    keep = []
    data = dict(A.data)
    data.update(B.data)
    blobs, nb = _blobs(data, keep)
    ab, bb = A.proto.SerializeToString(), B.proto.SerializeToString()
    am, an = (A.n, A.m) if ta else (A.m, A.n)
    bm, bn = (B.n, B.m) if tb else (B.m, B.n)
    m, n = (am, an) if op == "+" else (am, bn)
    dense = np.empty(m * n, dtype=np.float64)
    rt = ctypes.c_int()
    mm, nn = ctypes.c_int64(), ctypes.c_int64()
    _check(L.eps_linear_map_binary(ctypes.c_char(op.encode()), ab, ctypes.c_size_t(len(ab)),
                                   ctypes.c_int(int(ta)), bb, ctypes.c_size_t(len(bb)),
                                   ctypes.c_int(int(tb)), blobs, ctypes.c_size_t(nb),
                                   ctypes.byref(rt), ctypes.byref(mm), ctypes.byref(nn),
                                   dense.ctypes.data_as(ctypes.c_void_p),
                                   ctypes.c_size_t(dense.size)))
    return rt.value, dense.reshape((mm.value, nn.value), order="F")


def linear_map_inverse(A):
    L = lib()
    keep = []
    blobs, nb = _blobs(A.data, keep)
    ab = A.proto.SerializeToString()
    dense = np.empty(A.m * A.n, dtype=np.float64)
    _check(L.eps_linear_map_inverse(ab, ctypes.c_size_t(len(ab)), blobs, ctypes.c_size_t(nb),
                                    dense.ctypes.data_as(ctypes.c_void_p),
                                    ctypes.c_size_t(dense.size)))
    return dense.reshape((A.n, A.m), order="F")


def tv1d(v, lam):
    v = np.ascontiguousarray(v, dtype=np.float64).reshape(-1)
    x = np.empty_like(v)
    _check(lib().eps_tv1d(v.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(v.size),
                          ctypes.c_double(lam), x.ctypes.data_as(ctypes.c_void_p)))
    return x
