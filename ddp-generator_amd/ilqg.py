"""ctypes view of the batched iLQG C-ABI (include/ilqg_batch.h).

This module holds no numerics: every method forwards to the shared library
built from ddp-generator_amd/csrc (C host + HIP kernels).  If that library is
missing the import of a solver FAILS LOUDLY — there is no Python or CPU
fallback for the hot path.

Naming follows the reference's MEX entry
`[success, x, u, cost] = iLQG<Problem>(x0, u_nom, params, opts)`
(iLQG_mex.c:19-52): parameters by name, options by name, same option keys and
error messages (iLQG.c:91-216).
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIBDIR = os.environ.get("ILQG_LIBDIR", os.path.join(HERE, "lib"))  # override: experiments with other builds

_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")

STATUS = {0: "active", 1: "converged_grad", 2: "converged_fun", 3: "max_iter", 4: "no_descent",
          5: "lambda_max", 6: "derivs_failed", 7: "init_failed"}
# the reference's iLQG() return value for each exit (1 = "success", iLQG.c:365-378 and SURVEY Appendix B-11).
# Exit 6 (calc_derivs failed) is not in the table: iLQG() leaves its loop with the back-pass flag of the PREVIOUS
# iteration (iLQG.c:247-249, 367), i.e. returns 1 unless it happened in the very first iteration — see success().
REFERENCE_SUCCESS = {1: 1, 2: 1, 5: 1, 3: 0, 4: 0, 7: 0}

MAX_ALPHA = 16


class IlqgError(RuntimeError):
    pass


_extra_libdirs = []


def add_library_dir(path):
    """another directory problem libraries are looked up in (a `make ... LIBDIR=<dir>` of the caller: a problem built
    from its own <problem>_gen_files/)"""
    path = os.path.abspath(path)
    if path not in _extra_libdirs:
        _extra_libdirs.append(path)


def library_path(problem="carparking", full_ddp=0, strict=False):
    """strict=True: the -ffp-contract=off build (bit-for-bit CPU parity of the backward pass; tests only);
    strict="wave": the build of a small problem forced into the one-wavefront-per-trajectory mapping;
    strict="elem": the n = 16 problem built with the one-output-element-per-lane backward step (FMA-free);
    strict="lean": the n = 16 problem's product build with the quad step laid out for two wavefronts per SIMD;
    strict="exp" / "exp_strict": the -DILQG_EXPERIMENTS builds (measured negative results kept tested: the backward pass
    on two wavefronts, the derivative record in parts, the box QP's pattern tables), product / FMA-free;
    strict="direct": the hint-free n = 16 pair with its callbacks on the record itself (-DILQG_DEV_ELEMENT=0: comparison)"""
    suffix = "_" + strict if strict in ("wave", "elem", "lean", "exp", "exp_strict", "direct") else ("_strict" if strict else "")
    name = "libilqg_%s_fd%d_hip%s.so" % (problem, int(full_ddp), suffix)
    for d in [LIBDIR] + _extra_libdirs:
        if os.path.exists(os.path.join(d, name)):
            return os.path.join(d, name)
    return os.path.join(LIBDIR, name)


_libs = {}


class _Named(C.Structure):
    """ilqg_named_t of include/ilqg_batch.h"""
    _fields_ = [("name", C.c_char_p), ("value", C.POINTER(C.c_double)), ("n", C.c_int)]


def _named_list(items):
    keep = []  # the arrays must outlive the call
    arr = (_Named * max(1, len(items)))()
    for i, (k, v) in enumerate(items.items()):
        a = np.ascontiguousarray(np.atleast_1d(v), dtype=np.float64)
        keep.append(a)
        arr[i] = _Named(k.encode(), a.ctypes.data_as(C.POINTER(C.c_double)), a.size)
    return arr, len(items), keep


def load_library(problem="carparking", full_ddp=0, strict=False):
    path = library_path(problem, full_ddp, strict)
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise IlqgError("HIP library %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(make -C ddp-generator_amd/csrc). There is no CPU fallback." % path)
    lib = C.CDLL(path)
    v = C.c_void_p
    lib.ilqg_problem_dims.argtypes = [_ip]
    lib.ilqg_problem_param_name.restype = C.c_char_p
    lib.ilqg_problem_param_name.argtypes = [C.c_int]
    lib.ilqg_problem_param_size.argtypes = [C.c_int]
    lib.ilqg_reference_success.argtypes = [C.c_int, C.c_int]
    lib.ilqg_batch_create.restype = v
    lib.ilqg_batch_create.argtypes = [C.c_int, C.c_int, C.c_int]
    lib.ilqg_batch_create_groups.restype = v
    lib.ilqg_batch_create_groups.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
    lib.ilqg_batch_groups.argtypes = [v]
    lib.ilqg_batch_scalar_to_device.argtypes = [v, C.c_char_p, v]
    lib.ilqg_batch_destroy.argtypes = [v]
    lib.ilqg_batch_error.restype = C.c_char_p
    lib.ilqg_batch_error.argtypes = [v]
    lib.ilqg_batch_set_option.argtypes = [v, C.c_char_p, _dp, C.c_int]
    lib.ilqg_batch_set_param.argtypes = [v, C.c_char_p, _dp, C.c_int]
    lib.ilqg_batch_set_x0.argtypes = [v, _dp]
    lib.ilqg_batch_set_u.argtypes = [v, _dp]
    lib.ilqg_batch_set_x.argtypes = [v, _dp]
    for f in ("init", "solve", "sync", "calc_derivs", "line_search", "update"):
        getattr(lib, "ilqg_batch_" + f).argtypes = [v]
    lib.ilqg_batch_iterate.argtypes = [v, C.c_int]
    lib.ilqg_batch_back_pass.argtypes = [v, C.c_int]
    lib.ilqg_batch_active.argtypes = [v, _ip]
    lib.ilqg_batch_get_x.argtypes = [v, _dp]
    lib.ilqg_batch_get_u.argtypes = [v, _dp]
    lib.ilqg_batch_get_gains.argtypes = [v, _dp, _dp]
    lib.ilqg_batch_set_gains.argtypes = [v, _dp, _dp]
    lib.ilqg_batch_get_derivs.argtypes = [v, _dp, _dp]
    lib.ilqg_problem_multiplier_dims.argtypes = [_ip]
    lib.ilqg_batch_get_multipliers.argtypes = [v, _dp, _dp]
    lib.ilqg_batch_set_multipliers.argtypes = [v, _dp, _dp]
    lib.ilqg_batch_set_derivs.argtypes = [v, _dp, _dp]
    lib.ilqg_batch_get_scalar.argtypes = [v, C.c_char_p, _dp]
    lib.ilqg_batch_set_scalar.argtypes = [v, C.c_char_p, _dp]
    lib.ilqg_batch_get_int.argtypes = [v, C.c_char_p, _ip]
    lib.ilqg_batch_set_int.argtypes = [v, C.c_char_p, _ip]
    lib.ilqg_batch_cost_device_ptr.restype = v
    lib.ilqg_batch_cost_device_ptr.argtypes = [v]
    lib.ilqg_batch_stream.restype = v
    lib.ilqg_batch_stream.argtypes = [v]
    lib.ilqg_batch_timing.argtypes = [v, C.c_int]
    lib.ilqg_batch_kernel_name.restype = C.c_char_p
    lib.ilqg_batch_kernel_name.argtypes = [C.c_int]
    lib.ilqg_batch_get_timing.argtypes = [v, C.c_int, _ip, _dp]
    lib.ilqg_batch_get_busy.argtypes = [v, C.c_int, _dp]
    lib.ilqg_boxqp_batch.argtypes = [C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp, _ip, _ip, _dp, _ip]
    lib.ilqg_boxqp_wave_batch.argtypes = lib.ilqg_boxqp_batch.argtypes
    lib.ilqg_boxqp_table_batch.argtypes = lib.ilqg_boxqp_batch.argtypes
    lib.ilqg_sincos_batch.argtypes = [C.c_int, C.c_int, _dp, _dp, _dp]
    lib.ilqg_multi_create.restype = v
    lib.ilqg_multi_create.argtypes = [C.c_int, _ip, C.c_int, C.c_int]
    lib.ilqg_multi_destroy.argtypes = [v]
    lib.ilqg_multi_error.restype = C.c_char_p
    lib.ilqg_multi_error.argtypes = [v]
    lib.ilqg_multi_devices.argtypes = [v]
    lib.ilqg_multi_set_option.argtypes = [v, C.c_char_p, _dp, C.c_int]
    lib.ilqg_multi_set_param.argtypes = [v, C.c_char_p, _dp, C.c_int]
    for f in ("set_x0", "set_u", "get_x", "get_u", "gather_costs"):
        getattr(lib, "ilqg_multi_" + f).argtypes = [v, _dp]
    for f in ("init", "solve", "sync"):
        getattr(lib, "ilqg_multi_" + f).argtypes = [v]
    lib.ilqg_multi_iterate.argtypes = [v, C.c_int]
    lib.ilqg_multi_active.argtypes = [v, _ip]
    lib.ilqg_multi_get_int.argtypes = [v, C.c_char_p, _ip]
    lib.ilqg_solve_single.argtypes = [C.c_int, _dp, _dp, C.POINTER(_Named), C.c_int, C.POINTER(_Named), C.c_int, _dp, _dp,
                                      _dp, _ip, _dp, C.c_char_p, C.c_int]
    _libs[path] = lib
    return lib


class Problem:
    """compile-time facts of one problem library"""

    def __init__(self, problem="carparking", full_ddp=0, strict=False):
        self.name, self.full_ddp = problem, int(full_ddp)
        self.lib = load_library(problem, full_ddp, strict)
        d = np.zeros(8, dtype=np.int32)
        self.lib.ilqg_problem_dims(d)
        self.nx, self.nu, _, self.rec_host, self.rec_dev, self.state_dep_limits, self.n_params = [int(x) for x in d[:7]]
        self.wave_mapping = bool(d[7])
        self.sxx = self.nx * (self.nx + 1) // 2
        self.suu = self.nu * (self.nu + 1) // 2
        self.params = [(self.lib.ilqg_problem_param_name(i).decode(), self.lib.ilqg_problem_param_size(i))
                       for i in range(self.n_params)]

    def device_count(self):
        return self.lib.ilqg_device_count()


class BatchSolver:
    """B trajectories of one problem advanced in lock step on one GPU."""

    def __init__(self, problem="carparking", full_ddp=0, batch=1, n_hor=500, device=0, params=None, opts=None,
                 strict=False, groups=0):
        """groups: the batch advances as that many independent sets of trajectories on separate HIP streams
        (0 = the library's choice, see ilqg_batch_create_groups)"""
        self.problem = Problem(problem, full_ddp, strict)
        self.lib = self.problem.lib
        self.B, self.N = int(batch), int(n_hor)
        self.h = self.lib.ilqg_batch_create_groups(int(device), self.B, self.N, int(groups))
        if not self.h:
            raise IlqgError(self.lib.ilqg_batch_error(None).decode())
        for k, val in (params or {}).items():
            self.set_param(k, val)
        for k, val in (opts or {}).items():
            self.set_option(k, val)

    # -- lifecycle ---------------------------------------------------------
    def close(self):
        if getattr(self, "h", None):
            self.lib.ilqg_batch_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc:
            raise IlqgError(self.lib.ilqg_batch_error(self.h).decode())

    # -- configuration -----------------------------------------------------
    def set_option(self, name, value):
        v = np.ascontiguousarray(np.atleast_1d(value), dtype=np.float64)
        self._ck(self.lib.ilqg_batch_set_option(self.h, name.encode(), v, v.size))

    def set_param(self, name, value):
        v = np.ascontiguousarray(np.atleast_1d(value), dtype=np.float64)
        self._ck(self.lib.ilqg_batch_set_param(self.h, name.encode(), v, v.size))

    def init(self, x0, u0):
        """x0 [B,nx], u0 [B,N,nu]: initial roll-out (clamps u) and solver entry state"""
        x0 = np.ascontiguousarray(x0, dtype=np.float64).reshape(self.B, self.problem.nx)
        u0 = np.ascontiguousarray(u0, dtype=np.float64).reshape(self.B, self.N, self.problem.nu)
        self._ck(self.lib.ilqg_batch_set_x0(self.h, x0))
        self._ck(self.lib.ilqg_batch_set_u(self.h, u0))
        self._ck(self.lib.ilqg_batch_init(self.h))

    # -- solver ------------------------------------------------------------
    def iterate(self, n=1):
        self._ck(self.lib.ilqg_batch_iterate(self.h, int(n)))

    def solve(self):
        self._ck(self.lib.ilqg_batch_solve(self.h))

    def solve_stream(self, x0, u0, with_trajectories=False):
        """a stream of len(x0) starts through this batch's slots (ilqg_batch_solve_stream): dict of cost, status, iterations
        per start, and x / u if asked for"""
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        u0 = np.ascontiguousarray(u0, dtype=np.float64)
        total = x0.shape[0]
        assert u0.shape == (total, self.N, self.problem.nu) and x0.shape == (total, self.problem.nx)
        cost = np.zeros(total)
        status, iters = np.zeros(total, dtype=np.int32), np.zeros(total, dtype=np.int32)
        x = np.zeros((total, self.N + 1, self.problem.nx)) if with_trajectories else None
        u = np.zeros((total, self.N, self.problem.nu)) if with_trajectories else None
        vp = C.c_void_p
        self.lib.ilqg_batch_solve_stream.argtypes = [vp, C.c_int, _dp, _dp, _dp, _ip, _ip, vp, vp]
        self._ck(self.lib.ilqg_batch_solve_stream(self.h, total, x0, u0, cost, status, iters,
                                                  x.ctypes.data_as(vp) if with_trajectories else None,
                                                  u.ctypes.data_as(vp) if with_trajectories else None))
        return dict(cost=cost, status=status, iterations=iters, x=x, u=u)

    def solve_trace(self):
        """the last solve(), poll by poll: (iterations done, trajectories active, slots iterated over) arrays and the number
        of times the active set was gathered into a smaller context (option "compact")"""
        cap = 4096
        it, act, slots = (np.zeros(cap, dtype=np.int32) for _ in range(3))
        comp = C.c_int(0)
        self.lib.ilqg_batch_solve_trace.argtypes = [C.c_void_p, _ip, _ip, _ip, C.c_int, C.POINTER(C.c_int)]
        n = min(cap, self.lib.ilqg_batch_solve_trace(self.h, it, act, slots, cap, C.byref(comp)))
        return it[:n].copy(), act[:n].copy(), slots[:n].copy(), comp.value

    def sync(self):
        self._ck(self.lib.ilqg_batch_sync(self.h))

    def active(self):
        n = np.zeros(1, dtype=np.int32)
        self._ck(self.lib.ilqg_batch_active(self.h, n))
        return int(n[0])

    def calc_derivs(self):
        self._ck(self.lib.ilqg_batch_calc_derivs(self.h))

    def back_pass(self, single_sweep=False, fused=False):
        """single_sweep: one sweep on stored records (the drop-in back_pass()); fused: derivatives on the fly"""
        self._ck(self.lib.ilqg_batch_back_pass(self.h, 1 if single_sweep else (2 if fused else 0)))

    def line_search(self):
        self._ck(self.lib.ilqg_batch_line_search(self.h))

    def update(self):
        self._ck(self.lib.ilqg_batch_update(self.h))

    # -- results -----------------------------------------------------------
    def x(self):
        out = np.zeros((self.B, self.N + 1, self.problem.nx))
        self._ck(self.lib.ilqg_batch_get_x(self.h, out))
        return out

    def u(self):
        out = np.zeros((self.B, self.N, self.problem.nu))
        self._ck(self.lib.ilqg_batch_get_u(self.h, out))
        return out

    def gains(self):
        l = np.zeros((self.B, self.N, self.problem.nu))
        L = np.zeros((self.B, self.N, self.problem.nu * self.problem.nx))
        self._ck(self.lib.ilqg_batch_get_gains(self.h, l, L))
        return l, L

    def set_x(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(self.B, self.N + 1, self.problem.nx)
        self._ck(self.lib.ilqg_batch_set_x(self.h, x))

    def set_u(self, u):
        u = np.ascontiguousarray(u, dtype=np.float64).reshape(self.B, self.N, self.problem.nu)
        self._ck(self.lib.ilqg_batch_set_u(self.h, u))

    def set_gains(self, l, L):
        self._ck(self.lib.ilqg_batch_set_gains(self.h, np.ascontiguousarray(l, dtype=np.float64),
                                               np.ascontiguousarray(L, dtype=np.float64)))

    def derivs(self):
        rec = np.zeros((self.B, self.N, self.problem.rec_host))
        fin = np.zeros((self.B, self.problem.nx + self.problem.sxx))
        self._ck(self.lib.ilqg_batch_get_derivs(self.h, rec, fin))
        return rec, fin

    def set_derivs(self, rec, fin):
        rec = np.ascontiguousarray(rec, dtype=np.float64).reshape(self.B, self.N, self.problem.rec_host)
        fin = np.ascontiguousarray(fin, dtype=np.float64).reshape(self.B, self.problem.nx + self.problem.sxx)
        self._ck(self.lib.ilqg_batch_set_derivs(self.h, rec, fin))

    def multiplier_dims(self):
        d = np.zeros(2, dtype=np.int32)
        self.lib.ilqg_problem_multiplier_dims(d)
        return int(d[0]), int(d[1])

    def multipliers(self):
        """(running [B, N, el], final [B, fin]): multipliersEl_t / multipliersFin_t member by member"""
        me, mf = self.multiplier_dims()
        run = np.zeros((self.B, self.N, max(me, 1)))
        fin = np.zeros((self.B, max(mf, 1)))
        self._ck(self.lib.ilqg_batch_get_multipliers(self.h, run, fin))
        return run[:, :, :me], fin[:, :mf]

    def set_multipliers(self, running, final):
        me, mf = self.multiplier_dims()
        # a part the problem does not have is not read by the library: any buffer will do
        run = (np.ascontiguousarray(running, dtype=np.float64).reshape(self.B, self.N, me) if me
               else np.zeros((self.B, self.N, 1)))
        fin = np.ascontiguousarray(final, dtype=np.float64).reshape(self.B, mf) if mf else np.zeros((self.B, 1))
        self._ck(self.lib.ilqg_batch_set_multipliers(self.h, run, fin))

    def scalar(self, name):
        w = MAX_ALPHA if name == "alpha_cost" else 1
        out = np.zeros((self.B, w))
        self._ck(self.lib.ilqg_batch_get_scalar(self.h, name.encode(), out))
        return out if w > 1 else out[:, 0]

    def set_scalar(self, name, value):
        v = np.ascontiguousarray(np.broadcast_to(np.asarray(value, dtype=np.float64), (self.B,)))
        self._ck(self.lib.ilqg_batch_set_scalar(self.h, name.encode(), v))

    def ints(self, name):
        w = MAX_ALPHA if name == "alpha_ok" else 1
        out = np.zeros((self.B, w), dtype=np.int32)
        self._ck(self.lib.ilqg_batch_get_int(self.h, name.encode(), out))
        return out if w > 1 else out[:, 0]

    def set_ints(self, name, value):
        v = np.ascontiguousarray(np.broadcast_to(np.asarray(value, dtype=np.int32), (self.B,)))
        self._ck(self.lib.ilqg_batch_set_int(self.h, name.encode(), v))

    def success(self):
        """the reference's iLQG() return value per trajectory (what the drop-in iLQG() of this library returns too)"""
        status, iters = self.ints("status"), self.ints("iterations")
        return np.array([self.lib.ilqg_reference_success(int(s), int(it)) for s, it in zip(status, iters)], dtype=np.int32)

    # -- plumbing for collectives / profiling ------------------------------
    def cost_device_ptr(self):
        return self.lib.ilqg_batch_cost_device_ptr(self.h)

    def groups(self):
        return int(self.lib.ilqg_batch_groups(self.h))

    def scalar_to_device(self, name, device_ptr):
        """per-trajectory scalar of the whole batch into caller-owned device memory (B doubles), no host copy"""
        self._ck(self.lib.ilqg_batch_scalar_to_device(self.h, name.encode(), C.c_void_p(int(device_ptr))))

    def stream(self):
        return self.lib.ilqg_batch_stream(self.h)

    def timing(self, enable=True):
        self._ck(self.lib.ilqg_batch_timing(self.h, 1 if enable else 0))

    def kernel_times(self):
        """{kernel name: (launches, total ms)} measured with HIP events on the solver's stream"""
        out = {}
        n = np.zeros(1, dtype=np.int32)
        ms = np.zeros(1)
        for k in range(self.lib.ilqg_batch_kernel_count()):
            self._ck(self.lib.ilqg_batch_get_timing(self.h, k, n, ms))
            out[self.lib.ilqg_batch_kernel_name(k).decode()] = (int(n[0]), float(ms[0]))
        return out


    def kernel_busy(self):
        """{kernel name: ms of wall clock its launches occupied} (union of the launch intervals: launches of one kernel
        on two streams overlap in the event clock while they take turns on the chip)"""
        out = {}
        ms = np.zeros(1)
        for k in range(self.lib.ilqg_batch_kernel_count()):
            self._ck(self.lib.ilqg_batch_get_busy(self.h, k, ms))
            if ms[0] > 0:
                out[self.lib.ilqg_batch_kernel_name(k).decode()] = float(ms[0])
        return out


class MultiSolver:
    """B trajectories sharded over several GPUs of one node in ONE process (ilqg_multi_*): contiguous blocks of
    ceil(B / G) trajectories per device, no exchange except costs() = one RCCL gather to the first device."""

    def __init__(self, problem="carparking", full_ddp=0, batch=2, n_hor=500, devices=(0,), params=None, opts=None):
        self.problem = Problem(problem, full_ddp)
        self.lib = self.problem.lib
        self.B, self.N = int(batch), int(n_hor)
        devs = np.ascontiguousarray(devices, dtype=np.int32)
        self.h = self.lib.ilqg_multi_create(devs.size, devs, self.B, self.N)
        if not self.h:
            raise IlqgError(self.lib.ilqg_multi_error(None).decode())
        for k, val in (params or {}).items():
            a = np.ascontiguousarray(np.atleast_1d(val), dtype=np.float64)
            self._ck(self.lib.ilqg_multi_set_param(self.h, k.encode(), a, a.size))
        for k, val in (opts or {}).items():
            a = np.ascontiguousarray(np.atleast_1d(val), dtype=np.float64)
            self._ck(self.lib.ilqg_multi_set_option(self.h, k.encode(), a, a.size))

    def _ck(self, rc):
        if rc:
            raise IlqgError(self.lib.ilqg_multi_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            self.lib.ilqg_multi_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def devices(self):
        return int(self.lib.ilqg_multi_devices(self.h))

    def init(self, x0, u0):
        x0 = np.ascontiguousarray(x0, dtype=np.float64).reshape(self.B, self.problem.nx)
        u0 = np.ascontiguousarray(u0, dtype=np.float64).reshape(self.B, self.N, self.problem.nu)
        self._ck(self.lib.ilqg_multi_set_x0(self.h, x0))
        self._ck(self.lib.ilqg_multi_set_u(self.h, u0))
        self._ck(self.lib.ilqg_multi_init(self.h))

    def iterate(self, n=1):
        self._ck(self.lib.ilqg_multi_iterate(self.h, int(n)))

    def solve(self):
        self._ck(self.lib.ilqg_multi_solve(self.h))

    def sync(self):
        self._ck(self.lib.ilqg_multi_sync(self.h))

    def active(self):
        n = np.zeros(1, dtype=np.int32)
        self._ck(self.lib.ilqg_multi_active(self.h, n))
        return int(n[0])

    def costs(self):
        """the single collective: per-trajectory costs of all devices, gathered over RCCL"""
        out = np.zeros(self.B)
        self._ck(self.lib.ilqg_multi_gather_costs(self.h, out))
        return out

    def x(self):
        out = np.zeros((self.B, self.N + 1, self.problem.nx))
        self._ck(self.lib.ilqg_multi_get_x(self.h, out))
        return out

    def u(self):
        out = np.zeros((self.B, self.N, self.problem.nu))
        self._ck(self.lib.ilqg_multi_get_u(self.h, out))
        return out

    def ints(self, name):
        out = np.zeros((self.B, MAX_ALPHA if name == "alpha_ok" else 1), dtype=np.int32)
        self._ck(self.lib.ilqg_multi_get_int(self.h, name.encode(), out))
        return out if name == "alpha_ok" else out[:, 0]


def solve_single(x0, u_nom, params, opts=None, problem="carparking", full_ddp=0, strict=False):
    """[success, x, u, cost] = iLQG<Problem>(x0, u_nom, params, opts) — the reference's MEX entry (iLQG_mex.c:19-144)
    through ilqg_solve_single: the drop-in iLQG() with back_pass() / line_search() on the GPU.
    Returns dict(success, x [N+1,nx], u [N,nu], cost, iterations, seconds)."""
    prob = Problem(problem, full_ddp, strict)
    u_nom = np.ascontiguousarray(u_nom, dtype=np.float64).reshape(-1, prob.nu)
    n_hor = u_nom.shape[0]
    x0 = np.ascontiguousarray(x0, dtype=np.float64).reshape(prob.nx)
    p_arr, p_n, keep1 = _named_list(params or {})
    o_arr, o_n, keep2 = _named_list(opts or {})
    x = np.zeros((n_hor + 1, prob.nx))
    u = np.zeros((n_hor, prob.nu))
    cost, secs, iters = np.zeros(1), np.zeros(1), np.zeros(1, dtype=np.int32)
    err = C.create_string_buffer(512)
    rc = prob.lib.ilqg_solve_single(n_hor, x0, u_nom, p_arr, p_n, o_arr, o_n, x, u, cost, iters, secs, err, 512)
    if rc < 0:
        raise IlqgError(err.value.decode())
    return dict(success=int(rc), x=x, u=u, cost=float(cost[0]), iterations=int(iters[0]), seconds=float(secs[0]))


def boxqp_batch(n, H, g, lower, upper, x0, problem="carparking", full_ddp=0, device=0, strict=False, cooperative=False):
    """device box-QP on `count` independent problems (arrays [count, ...]); unit-test entry.
    cooperative: the form of the one-wavefront-per-trajectory mapping (one lane per variable);
    cooperative="table": the per-lane form with the factorisations of all clamp patterns made up front"""
    lib = load_library(problem, full_ddp, strict)
    H = np.ascontiguousarray(H, dtype=np.float64)
    count = H.shape[0]
    t = n * (n + 1) // 2
    x = np.array(x0, dtype=np.float64).reshape(count, n).copy()
    clamp = np.zeros((count, n), dtype=np.int32)
    nfree = np.zeros(count, dtype=np.int32)
    invH = np.zeros((count, t))
    rc = np.zeros(count, dtype=np.int32)
    fn = lib.ilqg_boxqp_table_batch if cooperative == "table" else (lib.ilqg_boxqp_wave_batch if cooperative else lib.ilqg_boxqp_batch)
    r = fn(device, n, count, H.reshape(count, t), np.ascontiguousarray(g, dtype=np.float64),
           np.ascontiguousarray(lower, dtype=np.float64), np.ascontiguousarray(upper, dtype=np.float64),
           x, clamp, nfree, invH, rc)
    if r:
        raise IlqgError("ilqg_boxqp_batch failed")
    return dict(rc=rc, x=x, clamp=clamp, n_free=nfree, invH=invH)


def sincos_batch(x, problem="carparking", full_ddp=0, device=0, strict=False):
    """device sin/cos exactly as the generated callbacks get them; unit-test entry"""
    lib = load_library(problem, full_ddp, strict)
    x = np.ascontiguousarray(x, dtype=np.float64)
    s, c = np.zeros_like(x), np.zeros_like(x)
    if lib.ilqg_sincos_batch(device, x.size, x, s, c):
        raise IlqgError("ilqg_sincos_batch failed")
    return s, c


# CarParking demo parameters, reference examples/CarParking/testCar.m:2-11
CAR_PARAMS = dict(
    d=[2.0], h=[0.03],
    pf=[0.01, 0.01, 0.01, 1.0], cf=[0.1, 0.1, 1.0, 0.3],
    cu=[1e-2, 1e-4], cx=[1e-3, 1e-3], px=[0.1, 0.1],
    limW=[-0.5, 0.5], limA=[-2.0, 2.0],
)
