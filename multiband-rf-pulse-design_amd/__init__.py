"""mbfir -- MI355X-native convex FIR / SLR beta-polynomial designer (host side).

Python mirror of the four convex designers of shanghong/Multiband-RF-pulse-Design
(same names, argument order, defaults, return convention and error behaviour as the
MATLAB functions they replace):

    h, status = fir_ap_cvx(n, f, a, d, obj, Peak, dbg)      # reference fir_ap_cvx.m:1
    h, status = fir_qp_cvx(n, f, a, d, k, obj, dbg)         # reference fir_qp_cvx.m:1
    h, status = fir_linprog(n, f, a, d, h0, dbg)            # reference ss/fir_linprog.m:2
    h, status = fir_qprog_phs(n, f, ac, dc, x0, dbg)        # reference ss/fir_qprog_phs.m:1

Everything is computed by the hand-written HIP solver behind the C ABI of
include/mbfir.h (libmbfir.so, built in-tree by __graft_entry__.build()); this module is
a ctypes binding plus argument checking.  There is NO CPU fallback: if the library is
missing or no GPU is present the calls raise.

The directory name contains '-', so import the package through the repo-root shim:
    import mbfir
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmbfir.so")

SOLVED, INFEASIBLE, NUMERICAL, EARLY_FAIL = 0, 1, 2, 3
E_ARG, E_HIP, E_NODEVICE = -1, -2, -3


class MbfirError(RuntimeError):
    pass


class Opts(C.Structure):
    """struct mbfir_opts (include/mbfir.h)."""
    _fields_ = [("grid_m", C.c_int), ("max_iter", C.c_int), ("feastol", C.c_double),
                ("abstol", C.c_double), ("reltol", C.c_double), ("refine", C.c_int),
                ("verbose", C.c_int), ("shard_rank", C.c_int), ("shard_size", C.c_int),
                ("dense_trig", C.c_int), ("ddkkt", C.c_int), ("lanes", C.c_int)]


class Info(C.Structure):
    """struct mbfir_info (include/mbfir.h)."""
    _fields_ = [("status", C.c_int), ("iters", C.c_int), ("n_unknowns", C.c_int), ("n_rows", C.c_int),
                ("n_freq", C.c_int), ("n_lp", C.c_int), ("n_q3", C.c_int), ("n_big", C.c_int),
                ("pcost", C.c_double), ("dcost", C.c_double), ("gap", C.c_double), ("relgap", C.c_double),
                ("pres", C.c_double), ("dres", C.c_double),
                ("ms_assemble", C.c_double), ("ms_solve", C.c_double), ("ms_post", C.c_double),
                ("ms_total", C.c_double), ("ms_gram", C.c_double), ("ms_chol", C.c_double),
                ("gram_flop", C.c_double), ("gram_launches", C.c_int), ("lattice", C.c_int),
                ("chol_flop", C.c_double), ("chol_launches", C.c_int), ("builds", C.c_int),
                ("dd_iters", C.c_int), ("dd_kmax", C.c_int), ("collectives", C.c_int), ("lanes", C.c_int),
                ("dd_form", C.c_int), ("ms_cap", C.c_double), ("cap_flop", C.c_double),
                ("collective_bytes", C.c_double), ("correctors", C.c_int), ("correctors_taken", C.c_int),
                ("gv_passes", C.c_int), ("gtv_passes", C.c_int)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class Job(C.Structure):
    """struct mbfir_job (include/mbfir.h)."""
    _fields_ = [("which", C.c_int), ("n", C.c_int), ("nband", C.c_int), ("rc", C.c_int),
                ("f", C.POINTER(C.c_double)), ("a", C.POINTER(C.c_double)), ("d", C.POINTER(C.c_double)),
                ("params", C.c_double * 4), ("h_re", C.POINTER(C.c_double)), ("h_im", C.POINTER(C.c_double)),
                ("info", Info), ("z", C.POINTER(C.c_double)), ("z_cap", C.c_int), ("err", C.c_char * 128)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_long, C.c_int, C.c_void_p)

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_lib = None

# every symbol include/mbfir.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "mbfir_create": (C.c_void_p, [C.c_int]),
    "mbfir_destroy": (None, [C.c_void_p]),
    "mbfir_last_error": (C.c_char_p, [C.c_void_p]),
    "mbfir_default_opts": (None, [C.POINTER(Opts)]),
    "mbfir_set_allreduce": (None, [C.c_void_p, ALLREDUCE_FN, C.c_void_p]),
    "mbfir_version": (C.c_char_p, []),
    "mbfir_comm_unique_id": (C.c_int, [C.c_void_p, C.c_char_p]),
    "mbfir_comm_init": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_char_p]),
    "mbfir_comm_destroy": (None, [C.c_void_p]),
    "mbfir_test_comm_allreduce": (C.c_int, [C.c_void_p, _dp, C.c_long, C.c_int]),
    "mbfir_last_solution": (C.c_int, [C.c_void_p, _dp, C.c_int]),
    "mbfir_ap_solve": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _dp, _dp, _dp, C.c_double, C.c_double,
                                 C.POINTER(Opts), _dp, _dp, C.POINTER(Info)]),
    "mbfir_qp_solve": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _dp, _dp, _dp, C.c_double, _dp, C.c_int,
                                 C.POINTER(Opts), _dp, _dp, C.POINTER(Info)]),
    "mbfir_linprog_solve": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _dp, _dp, _dp,
                                      C.POINTER(Opts), _dp, _dp, C.POINTER(Info)]),
    "mbfir_qprog_phs_solve": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp,
                                        C.POINTER(Opts), _dp, _dp, C.POINTER(Info)]),
    "mbfir_solve_batch": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.POINTER(Job), C.c_int, C.POINTER(Opts)]),
    "mbfir_b2a": (C.c_int, [C.c_void_p, C.c_int, _dp, _dp, _dp, _dp]),
    "mbfir_ab2rf": (C.c_int, [C.c_void_p, C.c_int, _dp, _dp, _dp, _dp, _dp, _dp]),
    "mbfir_b2rf": (C.c_int, [C.c_void_p, C.c_int, _dp, _dp, _dp, _dp]),
    "mbfir_abr": (C.c_int, [C.c_void_p, C.c_int, _dp, _dp, _dp, C.c_int, _dp, C.c_int, _dp, _dp, _dp, _dp]),
    "mbfir_bloch": (C.c_int, [C.c_void_p, C.c_int, _dp, _dp, _dp, _dp, _dp, _dp, C.c_double, C.c_double, C.c_int, _dp, C.c_int,
                              _dp, _dp, _dp, C.c_int, C.c_double, _dp, _dp, _dp]),
    "mbfir_assemble": (C.c_int, [C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp, C.c_int,
                                 C.POINTER(C.c_void_p), C.c_char_p, C.c_int]),
    "mbfir_program_free": (None, [C.c_void_p]),
    "mbfir_program_shard": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "mbfir_program_dims": (None, [C.c_void_p, _ip]),
    "mbfir_program_trig": (None, [C.c_void_p, _dp, _ip, _dp, _dp, _ip, _dp, _dp]),
    "mbfir_program_rows": (None, [C.c_void_p, _ip, _ip, _dp, _dp, _dp, _dp]),
    "mbfir_program_replicated": (None, [C.c_void_p, _ip]),
    "mbfir_test_gram": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp]),
    "mbfir_test_chol": (C.c_int, [C.c_void_p, C.c_int, _dp, _dp, _dp]),
    "mbfir_test_chol_lanes": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), _dp, _dp, _dp]),
    "mbfir_test_specfact": (C.c_int, [C.c_void_p, C.c_int, _dp, _dp, _dp]),
    "mbfir_test_fold": (C.c_int, [_dp, C.c_int, C.c_int, C.POINTER(C.c_long)]),
    "mbfir_test_ddsolve": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _dp, _dp, _dp, C.c_int, _dp, _dp, _dp, _dp, _ip, _dp, _dp]),
    "mbfir_test_mfma_peak": (C.c_int, [C.c_void_p, _dp, _dp]),
    "mbfir_test_time_kernels": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp]),
}


def load_library():
    """dlopen libmbfir.so and bind every symbol of include/mbfir.h.  Raises when the
    library has not been built (no silent fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own HIP runtime (same SONAME as /opt/rocm's).  Whichever is loaded first serves
    # the whole process, and torch stops seeing the GPU when it is not its own -- so load torch first.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise MbfirError("libmbfir.so not built (%s); run `python -c 'import __graft_entry__ as g; g.build()'`"
                         % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _vec(x, dtype=np.float64):
    return np.ascontiguousarray(np.asarray(x, dtype=dtype).ravel())


def _ptr(a):
    return a.ctypes.data_as(_dp)


class Context:
    """Owns the device memory, stream and (optionally) the all-reduce hook of one GPU.
    Reusable across calls; not thread-safe (mirrors mbfir_ctx)."""

    def __init__(self, device=None):
        lib = load_library()
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0"))
        self.device = device
        self._h = lib.mbfir_create(device)
        if not self._h:
            raise MbfirError("mbfir_create(%d) failed: %s" % (device, lib.mbfir_last_error(None).decode()))
        self._cb = None

    def close(self):
        if getattr(self, "_h", None):
            load_library().mbfir_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def last_error(self):
        return load_library().mbfir_last_error(self._h).decode()

    def last_solution(self, n_unknowns):
        """Conic solution [x ; y] / tau of the last solve (mbfir_last_solution)."""
        z = np.zeros(int(n_unknowns))
        k = load_library().mbfir_last_solution(self._h, _ptr(z), len(z))
        if k < 0:
            raise MbfirError("mbfir_last_solution failed")
        return z[:k]

    def set_allreduce(self, fn):
        """fn(ptr:int, count:int, op:int) -> int ; op 0 = sum, 1 = max (device pointer).  An exception inside the
        hook is logged and reported as a failed collective (the solve then raises) instead of being swallowed by
        ctypes with an undefined return value."""
        def cb(buf, count, op, user):
            try:
                return int(fn(buf, count, op))
            except BaseException:                          # noqa: BLE001 -- must not propagate into C
                import traceback
                traceback.print_exc()
                return 1
        self._cb = ALLREDUCE_FN(cb)
        load_library().mbfir_set_allreduce(self._h, self._cb, None)

    def init_comm(self, rank=None, size=None, group=None):
        """Native RCCL communicator for row-sharded solves: rank 0 makes the unique id, torch.distributed (any
        backend) carries its 128 bytes to the other ranks, every rank joins (mbfir_comm_init).  From then on the
        solver issues its all-reduces itself, on its own stream."""
        import torch
        import torch.distributed as dist
        lib = load_library()
        # one RCCL per process: point the library at the copy torch has loaded
        cand = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        if os.path.exists(cand):
            os.environ.setdefault("MBFIR_RCCL_PATH", cand)
        rank = dist.get_rank(group) if rank is None else rank
        size = dist.get_world_size(group) if size is None else size
        buf = C.create_string_buffer(128)
        if rank == 0:
            _check(self, lib.mbfir_comm_unique_id(self._h, buf))
        box = [bytes(buf.raw)]
        if size > 1:
            dist.broadcast_object_list(box, src=0, group=group)
        _check(self, lib.mbfir_comm_init(self._h, int(size), int(rank), box[0]))

    def comm_allreduce(self, v, op=0):
        """test hook: all-reduce a host array through the context's RCCL communicator (in place)."""
        v = np.ascontiguousarray(v, dtype=np.float64)
        _check(self, load_library().mbfir_test_comm_allreduce(self._h, _ptr(v), v.size, int(op)))
        return v

    def destroy_comm(self):
        load_library().mbfir_comm_destroy(self._h)


class _DevArray:
    """__cuda_array_interface__ view of `count` doubles at a raw device pointer."""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


def device_tensor(ptr, count):
    """torch tensor aliasing device memory owned by the solver (zero copy)."""
    import torch
    return torch.as_tensor(_DevArray(ptr, count), device="cuda")


def make_torch_allreduce(group=None, wrap=None):
    """All-reduce hook for row-sharded solves: fn(ptr, count, op) over torch.distributed
    (backend nccl = RCCL over xGMI; op 0 = sum, 1 = max).  `wrap(ptr, count)` makes the tensor
    (default: a zero-copy view of the device pointer); the hook returns once the result is in place."""
    import torch
    import torch.distributed as dist
    wrap = wrap or device_tensor

    def hook(ptr, count, op):
        t = wrap(ptr, count)
        dist.all_reduce(t, op=dist.ReduceOp.MAX if op == 1 else dist.ReduceOp.SUM, group=group)
        if t.is_cuda:
            torch.cuda.current_stream().synchronize()
        return 0

    return hook


_default_ctx = {}


def get_context(device=None):
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0"))
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]


def make_opts(**kw):
    o = Opts()
    load_library().mbfir_default_opts(C.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise TypeError("unknown option %r" % k)
        setattr(o, k, v)
    return o


def opts_with(opts, **kw):
    """a copy of `opts` (None: the defaults) with the given fields replaced"""
    o = make_opts()
    if opts is not None:
        C.memmove(C.byref(o), C.byref(opts), C.sizeof(Opts))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise TypeError("unknown option %r" % k)
        setattr(o, k, v)
    return o


def _finish(ctx, rc, hre, him, info, want_info):
    if rc < 0:
        msg = ctx.last_error()
        if rc == E_ARG:
            raise ValueError(msg)            # the reference's error() calls
        raise MbfirError("mbfir solve failed (%d): %s" % (rc, msg))
    if rc == SOLVED:
        h, status = hre + 1j * him, "Solved"
    else:
        h, status = np.zeros(0, dtype=np.complex128), "Failed"      # h = [] on failure
    if want_info:
        d = info.as_dict()
        d["rc"] = rc
        return h, status, d
    return h, status


def fir_ap_cvx(n, f, a, d, obj=0.0, Peak=1e-3, dbg=0, *, opts=None, ctx=None, info=False):
    """Arbitrary-phase multiband magnitude design (reference fir_ap_cvx.m).
    Returns (h, status): h complex ndarray of n taps (the reference's 1 x n row),
    status 'Solved' or 'Failed' (h empty).  obj < 0 raises ValueError('invalid input of obj')."""
    if n is None or f is None or a is None or d is None:
        raise ValueError("not enough input")                       # fir_ap_cvx.m:32
    ctx = ctx or get_context()
    f, a, d = _vec(f), _vec(a), _vec(d)
    _check_spec(f, a, d)
    hre, him, inf = np.zeros(n), np.zeros(n), Info()
    o = opts if opts is not None else make_opts(verbose=1 if dbg else 0)
    rc = load_library().mbfir_ap_solve(ctx._h, int(n), len(d), _ptr(f), _ptr(a), _ptr(d), float(obj),
                                       float(Peak), C.byref(o), _ptr(hre), _ptr(him), C.byref(inf))
    return _finish(ctx, rc, hre, him, inf, info)


def fir_qp_cvx(n, f, a, d, k=100.0, obj=0.0, dbg=0, *, opts=None, ctx=None, info=False):
    """Quadratic-phase design (reference fir_qp_cvx.m).  obj scalar -> E_total + obj*Peak,
    two entries -> delta + obj(1)*E_total + obj(2)*Peak.  Returns (h, status), h n x 1."""
    if n is None or f is None or a is None or d is None:
        raise ValueError("not enough input")                       # fir_qp_cvx.m:28
    ctx = ctx or get_context()
    f, a, d = _vec(f), _vec(a), _vec(d)
    _check_spec(f, a, d)
    objv = _vec(obj)
    if len(objv) not in (1, 2):
        raise ValueError("invalid input of obj")                   # fir_qp_cvx.m:194-196
    hre, him, inf = np.zeros(n), np.zeros(n), Info()
    o = opts if opts is not None else make_opts(verbose=1 if dbg else 0)
    rc = load_library().mbfir_qp_solve(ctx._h, int(n), len(d), _ptr(f), _ptr(a), _ptr(d), float(k),
                                       _ptr(objv), len(objv), C.byref(o), _ptr(hre), _ptr(him), C.byref(inf))
    return _finish(ctx, rc, hre, him, inf, info)


def fir_linprog(n, f, a, d, h0=None, dbg=0, *, opts=None, ctx=None, info=False):
    """Linear-phase (Hermitian-symmetric) multiband LP (reference ss/fir_linprog.m).
    h0 is the reference's warm start for its active-set linprog; an interior-point
    method has no use for it and it is ignored.  Returns (h, status), h n x 1."""
    ctx = ctx or get_context()
    f, a, d = _vec(f), _vec(a), _vec(d)
    _check_spec(f, a, d)
    hre, him, inf = np.zeros(n), np.zeros(n), Info()
    o = opts if opts is not None else make_opts(verbose=1 if dbg else 0)
    rc = load_library().mbfir_linprog_solve(ctx._h, int(n), len(d), _ptr(f), _ptr(a), _ptr(d), C.byref(o),
                                            _ptr(hre), _ptr(him), C.byref(inf))
    return _finish(ctx, rc, hre, him, inf, info)


def fir_qprog_phs(n, f, ac, dc, x0=None, dbg=0, *, opts=None, ctx=None, info=False):
    """Minimum-energy design with per-band magnitude and phase bounds (reference
    ss/fir_qprog_phs.m).  ac (2 per band) and dc (1 per band) are complex.  x0 is
    overwritten by [] in the reference (:338) and ignored here.  Returns (h, status)."""
    ctx = ctx or get_context()
    f = _vec(f)
    ac = np.ascontiguousarray(np.asarray(ac, dtype=np.complex128).ravel())
    dc = np.ascontiguousarray(np.asarray(dc, dtype=np.complex128).ravel())
    if len(f) % 2 or len(ac) != len(f) or len(dc) != len(f) // 2:
        raise ValueError("f, ac, dc have inconsistent lengths")
    are, aim = _vec(ac.real), _vec(ac.imag)
    dre, dim = _vec(dc.real), _vec(dc.imag)
    hre, him, inf = np.zeros(n), np.zeros(n), Info()
    o = opts if opts is not None else make_opts(verbose=1 if dbg else 0)
    rc = load_library().mbfir_qprog_phs_solve(ctx._h, int(n), len(dc), _ptr(f), _ptr(are), _ptr(aim), _ptr(dre),
                                              _ptr(dim), C.byref(o), _ptr(hre), _ptr(him), C.byref(inf))
    return _finish(ctx, rc, hre, him, inf, info)


# ---- inverse SLR: beta polynomial -> alpha -> RF (dzrf_mb.m:239-244) -------------------------------
def _split(z):
    z = np.asarray(z, dtype=np.complex128).ravel()
    return np.ascontiguousarray(z.real), np.ascontiguousarray(z.imag)


def b2a(bc, *, ctx=None):
    """`aca = b2a(bc)` (b2a.m:15): the minimum-phase alpha polynomial consistent with beta."""
    ctx = ctx or get_context()
    bre, bim = _split(bc)
    are, aim = np.zeros(len(bre)), np.zeros(len(bre))
    _check(ctx, load_library().mbfir_b2a(ctx._h, len(bre), _ptr(bre), _ptr(bim), _ptr(are), _ptr(aim)))
    return are + 1j * aim


def ab2rf(ac, bc, *, ctx=None):
    """`rf = ab2rf(ac, bc)` (ab2rf.m:14): inverse SLR transform, rf in radians per sample."""
    ctx = ctx or get_context()
    are, aim = _split(ac)
    bre, bim = _split(bc)
    if len(are) != len(bre):
        raise ValueError("ab2rf: alpha and beta must have the same length")
    rre, rim = np.zeros(len(bre)), np.zeros(len(bre))
    _check(ctx, load_library().mbfir_ab2rf(ctx._h, len(bre), _ptr(are), _ptr(aim), _ptr(bre), _ptr(bim),
                                           _ptr(rre), _ptr(rim)))
    return rre + 1j * rim


def b2rf(bc, *, ctx=None):
    """`rf = b2rf(bc)` = ab2rf(b2a(bc), bc) with alpha kept on the device (rf_tools/mex5/b2rf.c)."""
    ctx = ctx or get_context()
    bre, bim = _split(bc)
    rre, rim = np.zeros(len(bre)), np.zeros(len(bre))
    _check(ctx, load_library().mbfir_b2rf(ctx._h, len(bre), _ptr(bre), _ptr(bim), _ptr(rre), _ptr(rim)))
    return rre + 1j * rim


def abrm(rf, g=None, x=None, *, hard_pulse=False, ctx=None):
    """`[a b] = abrm(rf, g, x)` (rf_tools/abrm.m): Cayley-Klein parameters of the pulse at positions x; with two
    arguments the second one is x.  hard_pulse=True simulates the model ab2rf inverts exactly instead."""
    ctx = ctx or get_context()
    if x is None:
        x, g = g, None
    rre, rim = _split(rf)
    xv = _vec(x)
    gv = _vec(g) if g is not None else None
    if gv is not None and len(gv) != len(rre):
        raise ValueError("abrm: g must have one entry per rf sample")
    out = [np.zeros(len(xv)) for _ in range(4)]
    _check(ctx, load_library().mbfir_abr(ctx._h, len(rre), _ptr(rre), _ptr(rim), _ptr(gv) if gv is not None else None,
                                         len(xv), _ptr(xv), 1 if hard_pulse else 0, *[_ptr(o) for o in out]))
    return out[0] + 1j * out[1], out[2] + 1j * out[3]


def abr(rf, g=None, x=None, *, ctx=None):
    """`[a b] = abr(rf, g, x)` (rf_tools/abr.m:19-34): abrm with Le Roux's convention on beta, b = -conj(b)."""
    a, b = abrm(rf, g, x, ctx=ctx)
    return a, -np.conj(b)


def rfscaleg(rf, t, gamma):
    """`rfs = rfscaleg(rf, t, gamma)` (rfscaleg.m:12-16): radians -> Gauss; t in ms, gamma in kHz/G."""
    rf = np.asarray(rf)
    return rf / (2 * np.pi * gamma * (t / len(rf)))


from . import spec          # noqa: E402  (physical multiband description -> (f, a, d); host only)
from . import io            # noqa: E402  (rfwrite / rfwrite_varian / signa)
from .io import rfwrite, rfwrite_varian, signa   # noqa: E402
from .flipzero import fir_flip_zero   # noqa: E402  (fir_flip_zero.m)
from .dzrf import dzrf_mb, fir_upsample, rf_mrange_desired   # noqa: E402  (dzrf_mb.m driver)
from .search import fir_ap, fir_qp, fir_min_order_linprog, fir_min_order_qprog_phs   # noqa: E402  (outer bisections)

_WHICH = {"fir_ap_cvx": 0, "fir_qp_cvx": 1, "fir_linprog": 2, "fir_qprog_phs": 3}
_pools = {}


def get_pool(streams=4, device=None):
    """`streams` contexts (one HIP stream each) on one device, created once and reused."""
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0"))
    key = (device, streams)
    if key not in _pools:
        _pools[key] = [Context(device) for _ in range(streams)]
    return _pools[key]


def solve_batch(jobs, *, opts=None, streams=4, ctxs=None, info=False, solutions=False):
    """Independent designs, `streams` in flight at a time on one GPU (mbfir_solve_batch): the shape of
    the reference's outer loops -- the probes of a min-order / min-duration bisection, parameter sweeps.
    jobs: sequence of (designer, args) with designer in {'fir_ap_cvx', 'fir_qp_cvx', 'fir_linprog',
    'fir_qprog_phs'} and args the positional arguments of that function (n, f, a, d, ...).
    Returns a list of (h, status) -- or (h, status, info) -- in job order, as the single calls return.
    solutions=True appends the conic solution z of every job (mbfir_last_solution) to its tuple."""
    ctxs = ctxs or get_pool(streams)
    o = opts if opts is not None else make_opts()
    arr = (Job * len(jobs))()
    keep = []
    for q, (name, args) in enumerate(jobs):
        which = _WHICH[name]
        n, f = int(args[0]), _vec(args[1])
        params = [0.0] * 4
        if which == 3:
            ac = np.asarray(args[2], dtype=np.complex128).ravel()
            dc = np.asarray(args[3], dtype=np.complex128).ravel()
            if len(f) % 2 or len(ac) != len(f) or len(dc) != len(f) // 2:
                raise ValueError("f, ac, dc have inconsistent lengths")
            a, d = _vec(np.stack([ac.real, ac.imag], 1)), _vec(np.stack([dc.real, dc.imag], 1))
            nband = len(dc)
        else:
            a, d = _vec(args[2]), _vec(args[3])
            _check_spec(f, a, d)
            nband = len(d)
            if which == 0:
                params[0] = float(args[4]) if len(args) > 4 else 0.0
                params[1] = float(args[5]) if len(args) > 5 else 1e-3
            elif which == 1:
                params[0] = float(args[4]) if len(args) > 4 else 100.0
                objv = _vec(args[5] if len(args) > 5 else 0.0)
                if len(objv) not in (1, 2):
                    raise ValueError("invalid input of obj")
                params[1:1 + len(objv)] = list(objv)
                params[3] = float(len(objv))
        hre, him = np.zeros(n), np.zeros(n)
        zbuf = np.zeros(4 * n + 16) if solutions else None       # every designer has at most 2n + 3 unknowns; checked below
        keep.append((f, a, d, hre, him, zbuf))
        J = arr[q]
        if solutions:
            J.z, J.z_cap = _ptr(zbuf), len(zbuf)
        J.which, J.n, J.nband = which, n, nband
        J.f, J.a, J.d, J.h_re, J.h_im = _ptr(f), _ptr(a), _ptr(d), _ptr(hre), _ptr(him)
        for t in range(4):
            J.params[t] = params[t]
    handles = (C.c_void_p * len(ctxs))(*[c._h for c in ctxs])
    load_library().mbfir_solve_batch(handles, len(ctxs), arr, len(jobs), C.byref(o))
    out = []
    for q in range(len(jobs)):
        hre, him = keep[q][3], keep[q][4]
        inf = Info.from_buffer_copy(arr[q].info)
        # errors are reported per job, like the single calls do (which context ran it is not recorded)
        if arr[q].rc < 0:
            if arr[q].rc == E_ARG:
                raise ValueError("job %d: invalid argument" % q)
            raise MbfirError("job %d failed (%d): %s" % (q, arr[q].rc, arr[q].err.decode(errors="replace")))
        res = _finish(ctxs[0], arr[q].rc, hre, him, inf, info)
        if solutions and inf.n_unknowns > len(keep[q][5]):
            raise MbfirError("job %d: the conic solution has %d unknowns, the buffer %d" % (q, inf.n_unknowns, len(keep[q][5])))
        out.append(res + (keep[q][5][: inf.n_unknowns],) if solutions else res)
    return out


def _check_spec(f, a, d):
    if len(f) % 2 or len(a) != len(f) or len(d) != len(f) // 2:
        raise ValueError("f, a, d have inconsistent lengths")


# ---- host-only introspection (no GPU): structured program -> dense (c, G, h) -------------------
def assemble_dense(which, n, f, a, d, params=(0.0,), grid_m=0, shard=None, rows=None):
    """Run the product's C++ problem assembly for designer `which` (0 ap, 1 qp, 2 linprog,
    3 qprog_phs; for 3 pass complex a, d) and expand the structured rows to dense arrays.
    shard=(rank, size) returns the rows that rank keeps in a row-sharded solve.
    rows: expand only these rows of G / h (a program too large to expand whole, e.g. n=2048, m=131072).
    Returns (rc, dict) -- used by the CPU tests to compare against the oracle."""
    lib = load_library()
    f = _vec(f)
    if which == 3:
        ac = np.asarray(a, dtype=np.complex128).ravel()
        dc = np.asarray(d, dtype=np.complex128).ravel()
        a = _vec(np.stack([ac.real, ac.imag], 1))
        d = _vec(np.stack([dc.real, dc.imag], 1))
    else:
        a, d = _vec(a), _vec(d)
    params = _vec(list(params) + [0.0] * 4)
    out = C.c_void_p()
    err = C.create_string_buffer(256)
    rc = lib.mbfir_assemble(which, int(n), len(f) // 2, _ptr(f), _ptr(a), _ptr(d), _ptr(params), int(grid_m),
                            C.byref(out), err, 256)
    if rc != 0:
        return rc, err.value.decode()
    if shard is not None:
        sub = C.c_void_p()
        rc2 = lib.mbfir_program_shard(out, int(shard[0]), int(shard[1]), C.byref(sub))
        lib.mbfir_program_free(out)
        if rc2 != 0:
            return rc2, "bad shard"
        out = sub
    try:
        dims = np.zeros(10, dtype=np.int32)
        lib.mbfir_program_dims(out, dims.ctypes.data_as(_ip))
        Nt, Ne, R, l, nq3, big, Mf, quad = [int(v) for v in dims[:8]]
        w = np.zeros(Mf)
        kind = np.zeros(Nt, dtype=np.int32)
        tau, scale, psign = np.zeros(Nt), np.zeros(Nt), np.zeros(Nt)
        pcol = np.zeros(Nt, dtype=np.int32)
        c = np.zeros(Nt + Ne)
        lib.mbfir_program_trig(out, _ptr(w), kind.ctypes.data_as(_ip), _ptr(tau), _ptr(scale),
                               pcol.ctypes.data_as(_ip), _ptr(psign), _ptr(c))
        freq = np.zeros(R, dtype=np.int32)
        col = np.zeros(R, dtype=np.int32)
        al, be, ey, h = np.zeros(R), np.zeros(R), np.zeros((R, 3)), np.zeros(R)
        lib.mbfir_program_rows(out, freq.ctypes.data_as(_ip), col.ctypes.data_as(_ip), _ptr(al), _ptr(be),
                               _ptr(ey), _ptr(h))
        rep = np.zeros(R, dtype=np.int32)
        lib.mbfir_program_replicated(out, rep.ctypes.data_as(_ip))
    finally:
        lib.mbfir_program_free(out)
    if rows is not None:
        rows = np.asarray(rows, dtype=np.int64)
        freq, col, al, be, ey, h, rep = freq[rows], col[rows], al[rows], be[rows], ey[rows], h[rows], rep[rows]
        used, inv = np.unique(freq[freq >= 0], return_inverse=True)
        wsub = w[used]
        fmap = np.full(len(freq), -1, dtype=np.int64)
        fmap[freq >= 0] = inv
    else:
        wsub, fmap = w, freq
    arg = np.outer(wsub, tau)
    A1 = scale * np.where(kind == 0, np.cos(arg), np.sin(arg))
    A2 = psign * A1[:, pcol] if quad else np.zeros_like(A1)
    G = np.zeros((len(freq), Nt + Ne))
    tr = fmap >= 0
    G[tr, :Nt] = al[tr, None] * A1[fmap[tr]] + be[tr, None] * A2[fmap[tr]]
    idr = np.nonzero(col >= 0)[0]
    G[idr, col[idr]] += al[idr]
    G[:, Nt:] = ey[:, :Ne]
    return 0, dict(c=c, G=G, h=h, l=l, nq3=nq3, big=big, w=w, Mf=Mf, Nt=Nt, Ne=Ne, quad=quad, freq=freq, R=R, rep=rep)


# ---- device kernel test hooks --------------------------------------------------------------------
def _check(ctx, rc):
    if rc != 0:
        raise MbfirError("mbfir test hook failed (%d): %s" % (rc, ctx.last_error()))


def test_gram(A, d, ctx=None):
    ctx = ctx or get_context()
    A = np.ascontiguousarray(A, dtype=np.float64)
    d = np.ascontiguousarray(np.atleast_2d(d), dtype=np.float64)
    m, nt = A.shape
    nw = d.shape[0]
    out = np.zeros((nw, nt, nt))
    _check(ctx, load_library().mbfir_test_gram(ctx._h, m, nt, nw, _ptr(A), _ptr(d), _ptr(out)))
    return out


def test_chol(H, ctx=None):
    ctx = ctx or get_context()
    H = np.ascontiguousarray(H, dtype=np.float64)
    n = H.shape[0]
    L, M = np.zeros((n, n)), np.zeros((n, n))
    _check(ctx, load_library().mbfir_test_chol(ctx._h, n, _ptr(H), _ptr(L), _ptr(M)))
    return L, M


def test_chol_lanes(Hs, form=-1, mask=None, ctx=None):
    """Hs: (nlanes, n, n) SPD matrices factorised together like a lock-step batch; returns (L, M) of shape (nlanes, n, n)."""
    ctx = ctx or get_context()
    Hs = np.ascontiguousarray(Hs, dtype=np.float64)
    nl, n = Hs.shape[0], Hs.shape[1]
    L, M = np.zeros((nl, n, n)), np.zeros((nl, n, n))
    mk = None
    if mask is not None:
        mk = (C.c_int * nl)(*[int(v) for v in mask])
    _check(ctx, load_library().mbfir_test_chol_lanes(ctx._h, n, nl, int(form), mk, _ptr(Hs), _ptr(L), _ptr(M)))
    return L, M


GAMMA_C13 = 6726.1          # rad/s/G, blochC.c:5
GAMMA_H1 = 26754.0          # blochH.c:6


def bloch(b1, gr, tp, t1, t2, df, dp, mode=0, mx=None, my=None, mz=None, nucleus="C-13", ctx=None):
    """[mx, my, mz] = bloch(b1, gr, tp, t1, t2, df, dp, mode, mx, my, mz) of bloch_simulation/bloch.m:1-44 on the device
    (blochC for nucleus 'C-13', blochH for 'H-1', as sim_rf_spectral.m:53-60 picks them).  b1 complex (Gauss), gr (ntime,)
    or (ntime, 1..3) G/cm, tp a scalar interval, ntime intervals, or ntime monotonically increasing end times
    (blochC.c:649-681), df Hz, dp (npos,) or (npos, 1..3) cm.  Returns arrays of shape (nfreq, npos) or, with mode & 2,
    (nfreq, npos, ntime)."""
    ctx = ctx or get_context()
    b1 = np.asarray(b1, dtype=np.complex128).ravel()
    nt = len(b1)
    gr = np.zeros((nt, 1)) if gr is None else np.asarray(gr, dtype=np.float64).reshape(nt, -1)
    g3 = [_vec(gr[:, i]) if i < gr.shape[1] else None for i in range(3)]
    tp = np.asarray(tp, dtype=np.float64).ravel()
    if tp.size == 1:
        ts = np.full(nt, tp[0])
    elif tp.size != nt:
        raise MbfirError("Time-point length differs from B1 length")
    else:
        iv = np.diff(np.concatenate([[0.0], tp]))
        ts = iv if np.all(iv > 0) else tp                       # increasing end times -> intervals (times2intervals)
    df = _vec(df)
    dp = np.asarray(dp, dtype=np.float64)
    dp = dp.reshape(-1, 1) if dp.ndim < 2 else dp
    p3 = [_vec(dp[:, i]) if i < dp.shape[1] else None for i in range(3)]
    nf, npos = len(df), dp.shape[0]
    ntout = nt if (int(mode) & 2) else 1
    out = []
    for init, dflt in ((mx, 0.0), (my, 0.0), (mz, 1.0)):
        o = np.zeros((nf * npos, ntout))
        o[:, 0] = dflt if init is None or np.size(init) != nf * npos else np.asarray(init, dtype=np.float64).ravel()
        out.append(np.ascontiguousarray(o))
    gamma = GAMMA_C13 if nucleus == "C-13" else GAMMA_H1 if nucleus == "H-1" else float(nucleus)
    nul = C.cast(None, _dp)
    _check(ctx, load_library().mbfir_bloch(ctx._h, nt, _ptr(_vec(b1.real)), _ptr(_vec(b1.imag)),
                                           *[_ptr(g) if g is not None else nul for g in g3], _ptr(_vec(ts)), float(t1), float(t2),
                                           nf, _ptr(df), npos, *[_ptr(p) if p is not None else nul for p in p3], int(mode),
                                           gamma, _ptr(out[0]), _ptr(out[1]), _ptr(out[2])))
    shape = (nf, npos, nt) if ntout > 1 else (nf, npos)
    return tuple(o.reshape(shape) for o in out)


def test_ddsolve(H, U, X, bh, bl, ctx=None, factor=False):
    """(H + U' diag(X) U)^-1 b through the device's double-double kernels; b, x: (nrhs, n) hi / lo parts.
    factor=True also returns the Cholesky factor (hi, lo)."""
    ctx = ctx or get_context()
    H = np.ascontiguousarray(H, dtype=np.float64)
    U = np.ascontiguousarray(U, dtype=np.float64).reshape(-1, H.shape[0])
    X = _vec(X)
    bh = np.ascontiguousarray(np.atleast_2d(bh), dtype=np.float64)
    bl = np.ascontiguousarray(np.atleast_2d(bl), dtype=np.float64)
    xh, xl = np.zeros_like(bh), np.zeros_like(bh)
    nfix = C.c_int(0)
    Lh, Ll = (np.zeros_like(H), np.zeros_like(H)) if factor else (None, None)
    _check(ctx, load_library().mbfir_test_ddsolve(ctx._h, H.shape[0], U.shape[0], _ptr(H), _ptr(U), _ptr(X), bh.shape[0],
                                                  _ptr(bh), _ptr(bl), _ptr(xh), _ptr(xl), C.byref(nfix),
                                                  _ptr(Lh) if factor else None, _ptr(Ll) if factor else None))
    if factor:
        return xh, xl, nfix.value, np.tril(Lh), np.tril(Ll)
    return xh, xl, nfix.value


def test_specfact(x, n, ctx=None):
    ctx = ctx or get_context()
    x = _vec(x)
    hre, him = np.zeros(n), np.zeros(n)
    _check(ctx, load_library().mbfir_test_specfact(ctx._h, int(n), _ptr(x), _ptr(hre), _ptr(him)))
    return hre + 1j * him


def time_kernels(n=1024, m=16394, nt=1023, reps=20, ctx=None):
    """ms per Cholesky+inverse phase (n x n) and per k_gram launch (m x nt) -- tuning aid."""
    ctx = ctx or get_context()
    a, b = C.c_double(), C.c_double()
    _check(ctx, load_library().mbfir_test_time_kernels(ctx._h, n, m, nt, reps, C.byref(a), C.byref(b)))
    return a.value, b.value


def mfma_peak(ctx=None):
    """Measured fp64 MFMA and VALU rates in TFLOP/s (roofline denominators)."""
    ctx = ctx or get_context()
    a, b = C.c_double(), C.c_double()
    _check(ctx, load_library().mbfir_test_mfma_peak(ctx._h, C.byref(a), C.byref(b)))
    return a.value, b.value


def test_fold(w, fold=True):
    """Host-side grid analysis of the lattice kernels (runs without a GPU): dict(ok, nfold, pairs, runs, longest, bad)."""
    w = np.ascontiguousarray(w, dtype=np.float64)
    out = (C.c_long * 6)()
    rc = load_library().mbfir_test_fold(_ptr(w), len(w), 1 if fold else 0, out)
    if rc != 0:
        raise RuntimeError("mbfir_test_fold failed (%d)" % rc)
    return dict(zip(("ok", "nfold", "pairs", "runs", "longest", "bad"), [int(v) for v in out]))
