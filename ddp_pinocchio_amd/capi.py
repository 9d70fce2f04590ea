"""ctypes binding of libddp_hip.so (include/ddp_hip/ddp_hip.h).

Plumbing only: loads the in-tree shared library, mirrors the C structs and exposes thin numpy
helpers used by tests/ and bench.py.  There is no CPU fallback here or anywhere in the package:
if the library (or a HIP device) is missing, the calls fail loudly.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DDP_HIP_LIB", os.path.join(_HERE, "libddp_hip.so"))   # override: development A/B builds

MAX_JOINTS = 64

OK = 0
EV_LLT_RESTART = 1
EV_LINESEARCH_FLOOR = 2
E_ARG, E_HIP, E_NODEVICE, E_UNSUPPORTED, E_MAX_RESTARTS, E_COMM = -1, -2, -3, -4, -5, -6

MODEL_PENDULUM, MODEL_TREE = 0, 1
EQ_NONE, EQ_CONFIG, EQ_FRAME = 0, 1, 2
BUILTIN_PENDULUM, BUILTIN_CHAIN6, BUILTIN_TREE38, BUILTIN_CHAIN6_FF, BUILTIN_TREE38_FF = 0, 1, 2, 3, 4
JOINT_REVOLUTE, JOINT_PRISMATIC, JOINT_FREEFLYER = 0, 1, 2
FLAG_NO_TENSORS, FLAG_TRACE = 1, 2
LIN_COST, LIN_FIRST, LIN_SECOND, LIN_EQ = 1, 2, 4, 8

SEQ_NAMES = [
    "X", "U", "X_NEW", "U_NEW", "LFX", "LFXX", "LX", "LU", "LXX", "LUX", "LUU",
    "F_VAL", "FX", "FU", "FXX", "FUX", "FUU",
    "EQ_VAL", "EQ_X", "EQ_U", "EQ_XX", "EQ_UX", "EQ_UU",
    "MULT_ORIGIN", "MULT_VAL", "MULT_JAC", "FB_ORIGIN", "FB_VAL", "FB_JAC",
    "VX_TRACE", "VXX_TRACE", "COSTS_OLD", "COSTS_NEW",
]
SEQ = {name: i for i, name in enumerate(SEQ_NAMES)}

K_BWD_ASSEMBLE, K_BWD_GAINS, K_FWD_ROLLOUT, K_LIN_FIRST, K_LIN_SECOND = range(5)

# every symbol include/ddp_hip/ddp_hip.h declares
EXPORTS = [
    "ddp_hip_abi_version", "ddp_hip_strerror", "ddp_hip_device_count", "ddp_hip_create", "ddp_hip_destroy",
    "ddp_hip_stream", "ddp_hip_synchronize", "ddp_hip_set_async", "ddp_hip_seq_size", "ddp_hip_device_ptr", "ddp_hip_upload",
    "ddp_hip_download", "ddp_hip_fill", "ddp_hip_rollout", "ddp_hip_linearize", "ddp_hip_linearize_stages",
    "ddp_hip_backward",
    "ddp_hip_forward", "ddp_hip_cost_seq_aug", "ddp_hip_swap_traj",
    "ddp_hip_update_origin", "ddp_hip_optimality", "ddp_hip_update_multipliers", "ddp_hip_profile_enable",
    "ddp_hip_profile_reset", "ddp_hip_profile_get", "ddp_hip_bwd_algorithmic_bytes", "ddp_hip_bwd_stream_bytes", "ddp_hip_comm_unique_id",
    "ddp_hip_comm_init", "ddp_hip_comm_destroy", "ddp_hip_shard_best", "ddp_hip_shard_pick", "ddp_hip_shard_broadcast",
    "ddp_hip_builtin_model",
    "ddp_hip_batch", "ddp_hip_set_active", "ddp_hip_solve", "ddp_hip_ctx_info",
    "ddp_hip_model_create", "ddp_hip_model_destroy", "ddp_hip_model_aba", "ddp_hip_model_aba_derivatives", "ddp_hip_model_frame",
]

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_lp = C.POINTER(C.c_int64)


class Model(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("nv", C.c_int32), ("mass", C.c_double), ("length", C.c_double),
        ("parent", _ip), ("jtype", _ip), ("axis", _dp), ("Rp", _dp), ("pp", _dp),
        ("mass_j", _dp), ("com", _dp), ("Ic", _dp), ("gravity", C.c_double * 3),
    ]


class Problem(C.Structure):
    _fields_ = [
        ("model", Model), ("dt", C.c_double), ("c", C.c_double), ("T", C.c_int64), ("batch", C.c_int64),
        ("eq_kind", C.c_int32), ("eq_advance", C.c_int32), ("ne", _lp), ("eq_target", _dp),
        ("frame_joint", C.c_int32), ("frame_off", C.c_double * 3),
        ("first_order_fd", C.c_int32), ("fd_mode", C.c_int32),
    ]


class SolverParams(C.Structure):
    _fields_ = [("max_iterations", C.c_int64), ("optimality_stopping_threshold", C.c_double), ("mu", C.c_double),
                ("reg", C.c_double), ("w", C.c_double), ("n", C.c_double), ("n_alpha", C.c_int32), ("pad_", C.c_int32),
                ("max_restarts", C.c_int64)]


class SolveLog(C.Structure):
    _fields_ = [("iterations", C.c_int64), ("result", C.c_int32), ("pad_", C.c_int32), ("mu", C.c_double), ("reg", C.c_double),
                ("w", C.c_double), ("n", C.c_double), ("last_step", C.c_double), ("opt_obj", C.c_double),
                ("opt_constr", C.c_double)]


class Info(C.Structure):
    _fields_ = [("device", C.c_int32), ("lin_path", C.c_int32), ("first_order", C.c_int32), ("bwd_path", C.c_int32),
                ("fwd_path", C.c_int32), ("has_tensors", C.c_int32), ("hbm_bytes", C.c_int64)]


class ModelStorage(C.Structure):
    _fields_ = [
        ("parent", C.c_int32 * MAX_JOINTS), ("jtype", C.c_int32 * MAX_JOINTS),
        ("axis", C.c_double * (MAX_JOINTS * 3)), ("Rp", C.c_double * (MAX_JOINTS * 9)),
        ("pp", C.c_double * (MAX_JOINTS * 3)), ("mass_j", C.c_double * MAX_JOINTS),
        ("com", C.c_double * (MAX_JOINTS * 3)), ("Ic", C.c_double * (MAX_JOINTS * 9)),
    ]


_lib = None


def lib():
    """Loads libddp_hip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback)")
    L = C.CDLL(LIB_PATH)
    L.ddp_hip_abi_version.restype = C.c_int
    L.ddp_hip_strerror.restype = C.c_char_p
    L.ddp_hip_strerror.argtypes = [C.c_int]
    L.ddp_hip_device_count.restype = C.c_int
    L.ddp_hip_create.argtypes = [C.POINTER(Problem), C.c_int, C.c_uint32, C.POINTER(C.c_void_p)]
    L.ddp_hip_destroy.argtypes = [C.c_void_p]
    L.ddp_hip_stream.restype = C.c_void_p
    L.ddp_hip_stream.argtypes = [C.c_void_p]
    L.ddp_hip_synchronize.argtypes = [C.c_void_p]
    L.ddp_hip_set_async.argtypes = [C.c_void_p, C.c_int]
    L.ddp_hip_seq_size.restype = C.c_int64
    L.ddp_hip_seq_size.argtypes = [C.c_void_p, C.c_int]
    L.ddp_hip_device_ptr.restype = C.c_void_p
    L.ddp_hip_device_ptr.argtypes = [C.c_void_p, C.c_int]
    L.ddp_hip_upload.argtypes = [C.c_void_p, C.c_int, _dp, C.c_int64, C.c_int64]
    L.ddp_hip_download.argtypes = [C.c_void_p, C.c_int, _dp, C.c_int64, C.c_int64]
    L.ddp_hip_fill.argtypes = [C.c_void_p, C.c_int, C.c_double]
    L.ddp_hip_rollout.argtypes = [C.c_void_p]
    L.ddp_hip_linearize.argtypes = [C.c_void_p]
    L.ddp_hip_linearize_stages.argtypes = [C.c_void_p, C.c_uint32]
    L.ddp_hip_backward.argtypes = [C.c_void_p, _dp, _dp, _lp, C.c_int64]
    L.ddp_hip_forward.argtypes = [C.c_void_p, _dp, C.c_int32, _dp, _dp]
    L.ddp_hip_cost_seq_aug.argtypes = [C.c_void_p, C.c_int, _dp]
    L.ddp_hip_swap_traj.argtypes = [C.c_void_p]
    L.ddp_hip_update_origin.argtypes = [C.c_void_p, C.c_int]
    L.ddp_hip_optimality.argtypes = [C.c_void_p, _dp, _dp, _dp]
    L.ddp_hip_update_multipliers.argtypes = [C.c_void_p, _dp]
    L.ddp_hip_profile_enable.argtypes = [C.c_void_p, C.c_int]
    L.ddp_hip_profile_reset.argtypes = [C.c_void_p]
    L.ddp_hip_profile_get.argtypes = [C.c_void_p, C.c_int, _dp, _lp]
    L.ddp_hip_bwd_algorithmic_bytes.restype = C.c_int64
    L.ddp_hip_bwd_stream_bytes.restype = C.c_int64
    L.ddp_hip_bwd_stream_bytes.argtypes = [C.c_void_p]
    L.ddp_hip_bwd_algorithmic_bytes.argtypes = [C.c_void_p]
    L.ddp_hip_comm_unique_id.argtypes = [C.POINTER(C.c_ubyte)]
    L.ddp_hip_comm_init.argtypes = [C.POINTER(C.c_ubyte), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.ddp_hip_comm_destroy.argtypes = [C.c_void_p]
    L.ddp_hip_shard_best.argtypes = [C.c_void_p, C.c_double, C.c_int64, _dp, _lp]
    L.ddp_hip_shard_pick.argtypes = [C.c_void_p, C.c_void_p, _dp, _lp]
    L.ddp_hip_shard_broadcast.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64]
    L.ddp_hip_builtin_model.argtypes = [C.c_int, C.c_uint64, C.POINTER(ModelStorage), C.POINTER(Model)]
    L.ddp_hip_batch.restype = C.c_int64
    L.ddp_hip_batch.argtypes = [C.c_void_p]
    L.ddp_hip_set_active.argtypes = [C.c_void_p, _ip]
    L.ddp_hip_solve.argtypes = [C.c_void_p, C.POINTER(SolverParams), C.POINTER(SolveLog)]
    L.ddp_hip_ctx_info.argtypes = [C.c_void_p, C.POINTER(Info)]
    L.ddp_hip_model_create.argtypes = [C.POINTER(Model), C.c_int, C.POINTER(C.c_void_p)]
    L.ddp_hip_model_destroy.argtypes = [C.c_void_p]
    L.ddp_hip_model_aba.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp]
    L.ddp_hip_model_aba_derivatives.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp, _dp, _dp]
    L.ddp_hip_model_frame.argtypes = [C.c_void_p, C.c_int32, _dp, _dp, _dp, _dp]
    _lib = L
    return L


class DdpHipError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        super().__init__(f"{where}: {lib().ddp_hip_strerror(code).decode()} ({code})")


def _check(code, where):
    if code < 0:
        raise DdpHipError(code, where)
    return code


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a):
    return a.ctypes.data_as(_dp)


class BuiltinModel:
    """Seeded model table from the library (ddp_hip_builtin_model), as numpy arrays."""

    def __init__(self, which, seed=0):
        self.storage = ModelStorage()
        self.model = Model()
        _check(lib().ddp_hip_builtin_model(which, seed, C.byref(self.storage), C.byref(self.model)), "builtin_model")
        self.kind, self.nv = self.model.kind, self.model.nv
        self.mass, self.length = self.model.mass, self.model.length
        self.gravity = np.array(list(self.model.gravity))
        st = self.storage
        # a free-flyer root (jtype[0] == JOINT_FREEFLYER) owns six of the nv velocities: nv - 5 joints, nq = nv + 1
        self.ff = self.kind == MODEL_TREE and st.jtype[0] == JOINT_FREEFLYER
        self.nj = nv = self.nv - 5 if self.ff else self.nv
        self.nq = self.nv + 1 if self.ff else self.nv
        self.parent = np.array(st.parent[:nv], dtype=np.int32)
        self.jtype = np.array(st.jtype[:nv], dtype=np.int32)
        self.axis = np.array(st.axis[:3 * nv]).reshape(nv, 3)
        self.Rp = np.array(st.Rp[:9 * nv]).reshape(nv, 3, 3)
        self.pp = np.array(st.pp[:3 * nv]).reshape(nv, 3)
        self.mass_j = np.array(st.mass_j[:nv])
        self.com = np.array(st.com[:3 * nv]).reshape(nv, 3)
        self.Ic = np.array(st.Ic[:9 * nv]).reshape(nv, 3, 3)

    def neutral(self):
        """neutral configuration (pinocchio::neutral): zeros, unit quaternion for a free-flyer root"""
        q = np.zeros(self.nq)
        if self.ff:
            q[6] = 1.0
        return q


class TableModel(BuiltinModel):
    """A tree of 1-DoF joints given as arrays (what adapters/urdf_reader.hpp produces from a URDF, or any robot description
    the host has): parent[nv] (parent[i] < i, -1 = world), jtype[nv] (JOINT_REVOLUTE / JOINT_PRISMATIC), axis[nv][3] (unit,
    joint frame), Rp[nv][3][3] and pp[nv][3] (placement in the parent frame), mass_j[nv], com[nv][3], Ic[nv][3][3] (about the
    com).  Same attributes as BuiltinModel, so the product (ProblemSpec) and the oracle take it alike."""

    def __init__(self, parent, jtype, axis, Rp, pp, mass_j, com, Ic, gravity=(0.0, 0.0, -9.81)):
        nv = len(parent)
        assert 1 <= nv <= 64
        self.storage = ModelStorage()
        self.model = Model()
        st = self.storage
        self.parent = np.ascontiguousarray(parent, dtype=np.int32)
        self.jtype = np.ascontiguousarray(jtype, dtype=np.int32)
        self.axis, self.Rp, self.pp = _f64(axis).reshape(nv, 3), _f64(Rp).reshape(nv, 3, 3), _f64(pp).reshape(nv, 3)
        self.mass_j, self.com, self.Ic = _f64(mass_j).reshape(nv), _f64(com).reshape(nv, 3), _f64(Ic).reshape(nv, 3, 3)
        for name, arr in (("parent", self.parent), ("jtype", self.jtype), ("axis", self.axis), ("Rp", self.Rp), ("pp", self.pp),
                          ("mass_j", self.mass_j), ("com", self.com), ("Ic", self.Ic)):
            dst = getattr(st, name)
            flat = arr.reshape(-1)
            for i in range(flat.size):
                dst[i] = flat[i]
        m = self.model
        m.kind, m.nv, m.mass, m.length = MODEL_TREE, nv, 0.0, 0.0
        m.parent = C.cast(st.parent, _ip); m.jtype = C.cast(st.jtype, _ip)
        m.axis = C.cast(st.axis, _dp); m.Rp = C.cast(st.Rp, _dp); m.pp = C.cast(st.pp, _dp)
        m.mass_j = C.cast(st.mass_j, _dp); m.com = C.cast(st.com, _dp); m.Ic = C.cast(st.Ic, _dp)
        m.gravity = (C.c_double * 3)(*[float(g) for g in gravity])
        self.kind, self.nv, self.mass, self.length = MODEL_TREE, nv, 0.0, 0.0
        self.gravity = np.array([float(g) for g in gravity])
        self.ff, self.nj, self.nq = False, nv, nv


class ModelHandle:
    """Point evaluations of the Model concept on the device (ddp_hip_model_*, seam B2)"""

    def __init__(self, model, device=0):
        self.model, self.nv, self.nq = model, model.nv, getattr(model, "nq", model.nv)
        self._h = C.c_void_p()
        _check(lib().ddp_hip_model_create(C.byref(model.model), device, C.byref(self._h)), "model_create")

    def close(self):
        if self._h:
            lib().ddp_hip_model_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def aba(self, q, v, tau):
        q, v, tau, out = _f64(q), _f64(v), _f64(tau), np.zeros(self.nv)
        _check(lib().ddp_hip_model_aba(self._h, _ptr(q), _ptr(v), _ptr(tau), _ptr(out)), "model_aba")
        return out

    def aba_derivatives(self, q, v, tau):
        q, v, tau = _f64(q), _f64(v), _f64(tau)
        n = self.nv
        dq, dv, dt = np.zeros(n * n), np.zeros(n * n), np.zeros(n * n)
        _check(lib().ddp_hip_model_aba_derivatives(self._h, _ptr(q), _ptr(v), _ptr(tau), _ptr(dq), _ptr(dv), _ptr(dt)), "model_aba_derivatives")
        return dq.reshape(n, n).T.copy(), dv.reshape(n, n).T.copy(), dt.reshape(n, n).T.copy()

    def frame(self, joint, off, q, jac=True):
        off, q = _f64(off), _f64(q)
        p3, J = np.zeros(3), np.zeros(3 * self.nv)
        _check(lib().ddp_hip_model_frame(self._h, joint, _ptr(off), _ptr(q), _ptr(p3), _ptr(J) if jac else None), "model_frame")
        return p3, J.reshape(self.nv, 3).T.copy()


class ProblemSpec:
    """Host-side description of problem_t (problem.hpp:872-1150) for one context."""

    def __init__(self, model, T, dt=0.01, c=1.0, batch=1, eq_kind=EQ_NONE, eq_advance=2, ne=None, eq_target=None,
                 frame_joint=0, frame_off=(0.0, 0.0, 0.0), first_order_fd=None, fd_mode=0):
        self.model = model
        self.T, self.dt, self.c, self.batch = int(T), float(dt), float(c), int(batch)
        self.eq_kind, self.eq_advance = int(eq_kind), int(eq_advance)
        self.ne = np.zeros(self.T, dtype=np.int64) if ne is None else np.ascontiguousarray(ne, dtype=np.int64)
        self.eq_target = _f64(np.zeros(0) if eq_target is None else eq_target)
        self.frame_joint, self.frame_off = int(frame_joint), tuple(float(v) for v in frame_off)
        if first_order_fd is None:
            first_order_fd = 0 if model.kind == MODEL_PENDULUM else 1
        self.first_order_fd, self.fd_mode = int(first_order_fd), int(fd_mode)
        self.nv = model.nv
        self.n, self.m, self.nx = 2 * model.nv, model.nv, getattr(model, "nq", model.nv) + model.nv
        self.Etot = int(self.ne.sum())

    def c_struct(self):
        p = Problem()
        p.model = self.model.model
        p.dt, p.c, p.T, p.batch = self.dt, self.c, self.T, self.batch
        p.eq_kind, p.eq_advance = self.eq_kind, self.eq_advance
        p.ne = self.ne.ctypes.data_as(_lp)
        p.eq_target = _ptr(self.eq_target) if self.eq_target.size else None
        p.frame_joint = self.frame_joint
        p.frame_off = (C.c_double * 3)(*self.frame_off)
        p.first_order_fd, p.fd_mode = self.first_order_fd, self.fd_mode
        return p


class Context:
    """One ddp_hip context (= `batch` resident problem instances on one GPU)."""

    def __init__(self, spec, device=0, flags=0):
        self.spec = spec
        self._h = C.c_void_p()
        self._c_problem = spec.c_struct()
        _check(lib().ddp_hip_create(C.byref(self._c_problem), device, flags, C.byref(self._h)), "ddp_hip_create")
        self.batch = spec.batch

    def close(self):
        if self._h:
            lib().ddp_hip_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def seq_size(self, name):
        return int(lib().ddp_hip_seq_size(self._h, SEQ[name]))

    def device_ptr(self, name):
        return lib().ddp_hip_device_ptr(self._h, SEQ[name])

    def upload(self, name, arr, first=0, count=None):
        arr = _f64(arr)
        count = self.batch - first if count is None else count
        assert arr.size == self.seq_size(name) * count, (name, arr.size, self.seq_size(name), count)
        _check(lib().ddp_hip_upload(self._h, SEQ[name], _ptr(arr), first, count), f"upload {name}")

    def download(self, name, first=0, count=None):
        count = self.batch - first if count is None else count
        out = np.empty((count, self.seq_size(name)), dtype=np.float64)
        _check(lib().ddp_hip_download(self._h, SEQ[name], _ptr(out), first, count), f"download {name}")
        return out

    def fill(self, name, value):
        _check(lib().ddp_hip_fill(self._h, SEQ[name], float(value)), f"fill {name}")

    def synchronize(self):
        _check(lib().ddp_hip_synchronize(self._h), "synchronize")

    def rollout(self):
        return _check(lib().ddp_hip_rollout(self._h), "rollout")

    def linearize(self, stages=None):
        if stages is None:
            return _check(lib().ddp_hip_linearize(self._h), "linearize")
        return _check(lib().ddp_hip_linearize_stages(self._h, stages), "linearize_stages")

    def backward(self, reg, mu, max_restarts=64):
        reg = _f64(np.broadcast_to(reg, (self.batch,))).copy()
        mu = _f64(np.broadcast_to(mu, (self.batch,))).copy()
        restarts = np.zeros(self.batch, dtype=np.int64)
        rc = _check(lib().ddp_hip_backward(self._h, _ptr(reg), _ptr(mu), restarts.ctypes.data_as(_lp), max_restarts),
                    "backward")
        return rc, reg, mu, restarts

    def forward(self, mu, n_alpha=8):
        mu = _f64(np.broadcast_to(mu, (self.batch,))).copy()
        step = np.zeros(self.batch)
        dcost = np.zeros(self.batch)
        rc = _check(lib().ddp_hip_forward(self._h, _ptr(mu), n_alpha, _ptr(step), _ptr(dcost)), "forward")
        return rc, step, dcost

    def cost_seq_aug(self, which, mu):
        mu = _f64(np.broadcast_to(mu, (self.batch,))).copy()
        return _check(lib().ddp_hip_cost_seq_aug(self._h, which, _ptr(mu)), "cost_seq_aug")

    def bwd_stream_bytes(self):
        return int(lib().ddp_hip_bwd_stream_bytes(self._h))

    def set_async(self, on=True):
        _check(lib().ddp_hip_set_async(self._h, 1 if on else 0), "set_async")

    def swap_traj(self):
        _check(lib().ddp_hip_swap_traj(self._h), "swap_traj")

    def shard_pick(self, comm=None):
        """(best cost, global index) over every instance of every rank of `comm` (a Comm, or None: this context alone):
        the cost of the trajectory `forward` just produced; one 16-byte all-gather (ddp_hip_shard_pick)"""
        c, i = C.c_double(), C.c_int64()
        _check(lib().ddp_hip_shard_pick(comm._h if comm is not None else None, self._h, C.byref(c), C.byref(i)), "shard_pick")
        return c.value, i.value

    def shard_broadcast(self, best_global_index, dst_local=0, comm=None):
        """the winner's X, U, FB_* from its owner (rank = index mod G, local position index div G) into local instance
        dst_local of every rank (ddp_hip_shard_broadcast)"""
        _check(lib().ddp_hip_shard_broadcast(comm._h if comm is not None else None, self._h, int(best_global_index), int(dst_local)),
               "shard_broadcast")

    def update_origin(self, which):
        _check(lib().ddp_hip_update_origin(self._h, which), "update_origin")

    def optimality(self, mu):
        mu = _f64(np.broadcast_to(mu, (self.batch,))).copy()
        obj = np.zeros(self.batch)
        constr = np.zeros(self.batch)
        _check(lib().ddp_hip_optimality(self._h, _ptr(mu), _ptr(obj), _ptr(constr)), "optimality")
        return obj, constr

    def update_multipliers(self, mu):
        mu = _f64(np.broadcast_to(mu, (self.batch,))).copy()
        _check(lib().ddp_hip_update_multipliers(self._h, _ptr(mu)), "update_multipliers")

    def set_active(self, active=None):
        """active: [batch] of 0 / 1 (None = all): inactive instances are frozen (ddp_hip_set_active)"""
        if active is None:
            _check(lib().ddp_hip_set_active(self._h, None), "set_active")
            return
        a = np.ascontiguousarray(np.broadcast_to(active, (self.batch,)), dtype=np.int32)
        _check(lib().ddp_hip_set_active(self._h, a.ctypes.data_as(_ip)), "set_active")

    def solve(self, max_iterations, threshold, mu, reg, w, n, n_alpha=8, max_restarts=1000):
        """solve<M> (ddp.hpp:745-842) of every instance (ddp_hip_solve); returns (rc, dict of per-instance arrays)"""
        sp = SolverParams(int(max_iterations), float(threshold), float(mu), float(reg), float(w), float(n), int(n_alpha), 0,
                          int(max_restarts))
        logs = (SolveLog * self.batch)()
        rc = _check(lib().ddp_hip_solve(self._h, C.byref(sp), logs), "solve")
        out = {k: np.array([getattr(l, k) for l in logs]) for k, _ in SolveLog._fields_ if k != "pad_"}
        out["done"] = out["result"] == 1
        return rc, out

    def info(self):
        i = Info()
        _check(lib().ddp_hip_ctx_info(self._h, C.byref(i)), "ctx_info")
        return {k: getattr(i, k) for k, _ in Info._fields_}

    def profile_enable(self, on=True, kernels=None):
        """on: every kernel class; kernels: an iterable of K_* ids to bracket only those"""
        mask = (1 if on else 0) if kernels is None else sum(2 << k for k in kernels)
        _check(lib().ddp_hip_profile_enable(self._h, mask), "profile_enable")

    def profile_reset(self):
        _check(lib().ddp_hip_profile_reset(self._h), "profile_reset")

    def profile_get(self, kernel_id):
        ms = C.c_double()
        n = C.c_int64()
        _check(lib().ddp_hip_profile_get(self._h, kernel_id, C.byref(ms), C.byref(n)), "profile_get")
        return ms.value, n.value

    def bwd_algorithmic_bytes(self):
        return int(lib().ddp_hip_bwd_algorithmic_bytes(self._h))


class Comm:
    """The library's own RCCL communicator (csrc/comm.cpp).  `unique_id()` on rank 0, the 128 bytes travel to the other
    ranks by whatever means the host has (bench.py: torch.distributed), then Comm(uid, rank, nranks, device) on every rank."""

    @staticmethod
    def unique_id():
        uid = (C.c_ubyte * 128)()
        _check(lib().ddp_hip_comm_unique_id(uid), "comm_unique_id")
        return bytes(uid)

    def __init__(self, uid, rank, nranks, device):
        buf = (C.c_ubyte * 128)(*uid)
        self._h = C.c_void_p()
        _check(lib().ddp_hip_comm_init(buf, rank, nranks, device, C.byref(self._h)), "comm_init")
        self.rank, self.nranks = rank, nranks

    def best(self, cost, gidx):
        c, i = C.c_double(), C.c_int64()
        _check(lib().ddp_hip_shard_best(self._h, float(cost), int(gidx), C.byref(c), C.byref(i)), "shard_best")
        return c.value, i.value

    def close(self):
        if self._h:
            lib().ddp_hip_comm_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
