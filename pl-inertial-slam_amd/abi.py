"""ctypes description of the C ABI declared in include/plba.h.

The same signature table binds two different shared objects:
  * the product:  pl-inertial-slam_amd/libplba_hip.so   (prefix ``plba_``), HIP kernels for gfx950;
  * the checker:  oracle/_build/libplba_oracle.so        (prefix ``orc_``), loaded ONLY by tests/,
    __graft_entry__.smoke() and bench.py's cpu_baseline leg (see oracle/oracle.py).
Nothing in this file computes anything: it is the Python spelling of plba.h.
"""
import ctypes as C
import numpy as np

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)
c_uint8_p = C.POINTER(C.c_uint8)

EDGE_POINT, EDGE_LINE, EDGE_IMU_PVR, EDGE_IMU_BIAS, EDGE_PRIOR = range(5)

STATUS = {0: "PLBA_OK", -1: "PLBA_ERR_INVALID", -2: "PLBA_ERR_STATE", -3: "PLBA_ERR_DEVICE",
          -4: "PLBA_ERR_NUMERIC", -5: "PLBA_ERR_EXCHANGE"}


class Options(C.Structure):
    _fields_ = [("tau", C.c_double), ("good_step_lower", C.c_double), ("good_step_upper", C.c_double),
                ("max_trials", C.c_int), ("user_lambda_init", C.c_double), ("marg_eps", C.c_double),
                ("fix_line_position_jacobian", C.c_int),
                ("device", C.c_int), ("use_mfma", C.c_int), ("profile", C.c_int), ("factor_block", C.c_int), ("factor_flow", C.c_int), ("chain_elim", C.c_int), ("wide_steps", C.c_int), ("band_solve", C.c_int), ("marg_exact", C.c_int), ("lm_fused", C.c_int),
                ("lm_fused_min_obs", C.c_int), ("lm_group_steps", C.c_int), ("chain_seg", C.c_int), ("twin_max_tiles", C.c_int), ("diag", C.c_int)]


class Stats(C.Structure):
    _fields_ = [("iterations", C.c_int), ("trials", C.c_int), ("stop_reason", C.c_int),
                ("solver_failures", C.c_int), ("chi2_initial", C.c_double), ("chi2_final", C.c_double),
                ("lambda_final", C.c_double), ("ms_total", C.c_double), ("ms_phase", C.c_double * 8)]


class TraceRow(C.Structure):
    _fields_ = [("iteration", C.c_int), ("trial", C.c_int), ("accepted", C.c_int), ("solver_ok", C.c_int),
                ("lam", C.c_double), ("chi2_current", C.c_double), ("chi2_trial", C.c_double),
                ("scale", C.c_double), ("rho", C.c_double)]


class Prior(C.Structure):
    _fields_ = [("n", C.c_int), ("m", C.c_int), ("nv", C.c_int),
                ("vid", c_int32_p), ("size", c_int32_p), ("idx", c_int32_p),
                ("x0", c_double_p), ("J0", c_double_p), ("r0", c_double_p),
                ("Ar", c_double_p), ("br", c_double_p)]


class LbaOptions(C.Structure):
    _fields_ = [("lambda_lm", C.c_double), ("lambda_k", C.c_double), ("max_iters", C.c_int), ("homog_th", C.c_double),
                ("min_error", C.c_double), ("min_error_change", C.c_double), ("use_iterate_poses", C.c_int), ("variant", C.c_int)]


class LbaStats(C.Structure):
    _fields_ = [("iterations", C.c_int), ("updates", C.c_int), ("err_first", C.c_double), ("err_last", C.c_double),
                ("lam", C.c_double), ("solver_failed", C.c_int), ("reserved", C.c_int)]


class Slide(C.Structure):
    """plba_slide of include/plba.h"""
    _fields_ = [("n_drop", C.c_int), ("drop_point", c_uint8_p), ("drop_line", c_uint8_p), ("drop_point_obs", c_uint8_p), ("drop_line_obs", c_uint8_p),
                ("K_add", C.c_int), ("vid_pvr", c_int32_p), ("vid_bias", c_int32_p),
                ("P3", c_double_p), ("V3", c_double_p), ("q_xyzw4", c_double_p), ("bg3", c_double_p), ("ba3", c_double_p), ("dbg3", c_double_p), ("dba3", c_double_p),
                ("fixed_pvr", c_uint8_p), ("fixed_bias", c_uint8_p),
                ("M_add", C.c_int), ("imu_kf_i", c_int32_p), ("imu_kf_j", c_int32_p), ("preint142", c_double_p), ("info_pvr81", c_double_p), ("info_bias36", c_double_p),
                ("Np_add", C.c_int), ("xyz3", c_double_p), ("point_fixed", c_uint8_p),
                ("Nl_add", C.c_int), ("sPeP6", c_double_p), ("line_fixed", c_uint8_p),
                ("Ep_add", C.c_int), ("po_pt", c_int32_p), ("po_kf", c_int32_p), ("uv2", c_double_p), ("po_inv_sigma2", c_double_p),
                ("El_add", C.c_int), ("lo_ln", c_int32_p), ("lo_kf", c_int32_p), ("l3", c_double_p), ("lo_inv_sigma2", c_double_p)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)

_P = C.c_void_p  # plba_problem*

# entry points of the product that have no counterpart in the reference's algorithm (memory management of the device-resident window): the
# CPU oracle — a restatement of the reference — does not implement them
PRODUCT_ONLY = {"slide_window", "get_sizes"}

# name -> (restype, argtypes); every symbol plba.h declares
SIGNATURES = {
    "default_options": (None, [C.POINTER(Options)]),
    "create": (C.c_int, [C.POINTER(Options), C.POINTER(_P)]),
    "destroy": (None, [_P]),
    "last_error": (C.c_char_p, [_P]),
    "backend_name": (C.c_char_p, []),
    "set_camera": (C.c_int, [_P, C.c_double, C.c_double, C.c_double, C.c_double, c_double_p, c_double_p]),
    "set_gravity": (C.c_int, [_P, c_double_p]),
    "set_keyframes": (C.c_int, [_P, C.c_int, c_int32_p, c_int32_p] + [c_double_p] * 7 + [c_uint8_p, c_uint8_p]),
    "set_points": (C.c_int, [_P, C.c_int, c_double_p, c_uint8_p]),
    "set_lines": (C.c_int, [_P, C.c_int, c_double_p, c_uint8_p]),
    "set_point_obs": (C.c_int, [_P, C.c_int, c_int32_p, c_int32_p, c_double_p, c_double_p]),
    "set_line_obs": (C.c_int, [_P, C.c_int, c_int32_p, c_int32_p, c_double_p, c_double_p]),
    "set_imu_edges": (C.c_int, [_P, C.c_int, c_int32_p, c_int32_p, c_double_p, c_double_p, c_double_p]),
    "set_prior": (C.c_int, [_P, C.c_int, C.c_int, c_int32_p, c_int32_p, c_int32_p, c_double_p, c_double_p, c_double_p]),
    "set_robust": (C.c_int, [_P, C.c_int, C.c_int, C.c_double]),
    "set_levels": (C.c_int, [_P, C.c_int, c_uint8_p]),
    "get_levels": (C.c_int, [_P, C.c_int, c_uint8_p]),
    "slide_window": (C.c_int, [_P, C.POINTER(Slide), c_int32_p, c_int32_p]),
    "get_sizes": (C.c_int, [_P, c_int32_p]),
    "set_shard": (C.c_int, [_P, C.c_int, C.c_int, ALLREDUCE_FN, C.c_void_p]),
    "set_stream": (C.c_int, [_P, C.c_void_p]),
    "optimize": (C.c_int, [_P, C.c_int, c_uint8_p, C.POINTER(Stats)]),
    "gate_outliers": (C.c_int, [_P, C.c_double, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "recompute_errors": (C.c_int, [_P]),
    "cull_observations": (C.c_int, [_P, C.c_double, c_uint8_p, c_uint8_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "get_edge_chi2": (C.c_int, [_P, C.c_int, c_double_p, c_uint8_p]),
    "get_trace": (C.c_int, [_P, C.POINTER(TraceRow), C.c_int, C.POINTER(C.c_int)]),
    "get_keyframes": (C.c_int, [_P] + [c_double_p] * 5),
    "get_points": (C.c_int, [_P, c_double_p]),
    "get_lines": (C.c_int, [_P, c_double_p]),
    "save_state": (C.c_int, [_P]),
    "restore_state": (C.c_int, [_P]),
    "marginalize": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(Prior)]),
    "marginalize_factors": (C.c_int, [_P, C.c_int, c_int32_p, C.c_int, c_int32_p, C.c_int, c_int32_p, C.c_int, C.c_int, c_int32_p, C.POINTER(Prior)]),
    "preintegrate": (C.c_int, [_P, C.c_int, c_int32_p, C.POINTER(C.c_longdouble), c_double_p, c_double_p, C.POINTER(C.c_longdouble), C.POINTER(C.c_longdouble), c_double_p, c_double_p, C.c_double, C.c_double, c_double_p]),
    "prior_free": (None, [C.POINTER(Prior)]),
    "set_marg_eps": (C.c_int, [_P, C.c_double]),
    "lba_default_options": (None, [C.POINTER(LbaOptions)]),
    "lba_visual": (C.c_int, [_P, C.POINTER(LbaOptions), C.c_int, c_double_p, c_int32_p, C.c_int, c_double_p, C.c_int, c_double_p,
                             C.c_int, c_int32_p, c_int32_p, c_double_p, C.c_int, c_int32_p, c_int32_p, c_double_p,
                             C.c_double, C.c_double, C.c_double, C.c_double, c_double_p, c_uint8_p, c_uint8_p, C.POINTER(LbaStats)]),
    "debug_build": (C.c_int, [_P, C.c_double, C.c_int]),
    "debug_get": (C.c_int, [_P, C.c_char_p, c_double_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "dense_solve": (C.c_int, [_P, C.c_int, c_double_p, c_double_p, c_double_p, C.POINTER(C.c_int)]),
    "debug_dense_solve": (C.c_int, [_P, C.c_int, c_double_p, c_double_p, c_double_p, C.POINTER(C.c_int)]),
}


class PlbaError(RuntimeError):
    pass


def _dp(a):
    return None if a is None else a.ctypes.data_as(c_double_p)


def _ip(a):
    return None if a is None else a.ctypes.data_as(c_int32_p)


def _up(a):
    return None if a is None else a.ctypes.data_as(c_uint8_p)


def _f64(a, shape=None):
    if a is None:
        return None
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a


def _i32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.int32)


def _u8(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.uint8)


class Lib:
    """A loaded implementation of plba.h (``prefix`` selects plba_* or orc_*)."""

    def __init__(self, path, prefix, optional=False):
        self.path = str(path)
        self.prefix = prefix
        self.cdll = C.CDLL(self.path)
        self.fn = {}
        missing = []
        for name, (res, args) in SIGNATURES.items():
            sym = prefix + name
            try:
                f = getattr(self.cdll, sym)
            except AttributeError:
                missing.append(sym)
                continue
            f.restype = res
            f.argtypes = args
            self.fn[name] = f
        missing = [m for m in missing if not (prefix != "plba_" and m[len(prefix):] in PRODUCT_ONLY)]
        if missing and not optional:      # (optional: a library that implements a subset — the quad-precision oracle build, oracle/make_quad.py)
            raise PlbaError("%s does not export: %s" % (self.path, ", ".join(missing)))

    def backend_name(self):
        return self.fn["backend_name"]().decode()

    def default_options(self):
        o = Options()
        self.fn["default_options"](C.byref(o))
        return o


class Problem:
    """One BA problem = one g2o::SparseOptimizer of the reference call site
    (src/mapHandler.cpp:5787-5797), behind the C ABI."""

    def __init__(self, lib, **opts):
        self.lib = lib
        o = lib.default_options()
        for k, v in opts.items():
            if not hasattr(o, k):
                raise PlbaError("unknown option %r" % k)
            setattr(o, k, v)
        self._h = _P()
        rc = lib.fn["create"](C.byref(o), C.byref(self._h))
        if rc != 0:
            msg = lib.fn["last_error"](None)
            raise PlbaError("create failed: %s (%s)" % (STATUS.get(rc, rc), msg.decode() if msg else ""))
        self._cb = None
        self.dims = {}

    # -- plumbing -----------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self.lib.fn["destroy"](self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc, allow_positive=False):
        if rc < 0 or (rc > 0 and not allow_positive):
            msg = self.lib.fn["last_error"](self._h)
            raise PlbaError("%s: %s" % (STATUS.get(rc, rc), msg.decode() if msg else ""))
        return rc

    def call(self, name, *args, allow_positive=False):
        return self._ck(self.lib.fn[name](self._h, *args), allow_positive)

    # -- upload ------------------------------------------------------------------------------
    def set_camera(self, fx, fy, cx, cy, Rbc, Pbc):
        Rbc = _f64(Rbc, (9,)); Pbc = _f64(Pbc, (3,))
        self.call("set_camera", fx, fy, cx, cy, _dp(Rbc), _dp(Pbc))

    def set_gravity(self, gw):
        gw = _f64(gw, (3,))
        self.call("set_gravity", _dp(gw))

    def set_keyframes(self, vid_pvr, vid_bias, P, V, q, bg=None, ba=None, dbg=None, dba=None,
                      fixed_pvr=None, fixed_bias=None):
        K = len(vid_pvr)
        arrs = [_f64(a) for a in (P, V, q, bg, ba, dbg, dba)]
        vp, vb = _i32(vid_pvr), _i32(vid_bias)
        fp, fb = _u8(fixed_pvr), _u8(fixed_bias)
        self.call("set_keyframes", K, _ip(vp), _ip(vb), *[_dp(a) for a in arrs], _up(fp), _up(fb))
        self.dims["K"] = K

    def set_points(self, xyz, fixed=None):
        xyz = _f64(xyz, (-1, 3)); f = _u8(fixed)
        self.call("set_points", len(xyz), _dp(xyz), _up(f))
        self.dims["Np"] = len(xyz)

    def set_lines(self, sPeP, fixed=None):
        l = _f64(sPeP, (-1, 6)); f = _u8(fixed)
        self.call("set_lines", len(l), _dp(l), _up(f))
        self.dims["Nl"] = len(l)

    def set_point_obs(self, pt, kf, uv, inv_sigma2=None):
        pt, kf, uv, w = _i32(pt), _i32(kf), _f64(uv, (-1, 2)), _f64(inv_sigma2)
        self.call("set_point_obs", len(pt), _ip(pt), _ip(kf), _dp(uv), _dp(w))
        self.dims["Ep"] = len(pt)

    def set_line_obs(self, ln, kf, l3, inv_sigma2=None):
        ln, kf, l3, w = _i32(ln), _i32(kf), _f64(l3, (-1, 3)), _f64(inv_sigma2)
        self.call("set_line_obs", len(ln), _ip(ln), _ip(kf), _dp(l3), _dp(w))
        self.dims["El"] = len(ln)

    def set_imu_edges(self, kf_i, kf_j, preint142, info_pvr, info_bias):
        ki, kj = _i32(kf_i), _i32(kf_j)
        pre, ip, ib = _f64(preint142, (-1, 142)), _f64(info_pvr, (-1, 81)), _f64(info_bias, (-1, 36))
        self.call("set_imu_edges", len(ki), _ip(ki), _ip(kj), _dp(pre), _dp(ip), _dp(ib))
        self.dims["M"] = len(ki)

    def set_prior(self, prior):
        """prior: dict(n, vid, size, idx, x0, J0 (n x n, J0[r, c]), r0) or None to clear."""
        if prior is None or len(prior["vid"]) == 0:
            self.call("set_prior", 0, 0, None, None, None, None, None, None)
            self.dims["n_prior"] = 0
            return
        n = int(prior["n"])
        vid, size, idx = _i32(prior["vid"]), _i32(prior["size"]), _i32(prior["idx"])
        x0, r0 = _f64(prior["x0"]), _f64(prior["r0"])
        J0 = np.asfortranarray(np.asarray(prior["J0"], dtype=np.float64).reshape(n, n))  # column-major
        self.call("set_prior", n, len(vid), _ip(vid), _ip(size), _ip(idx), _dp(x0),
                  J0.ctypes.data_as(c_double_p), _dp(r0))
        self.dims["n_prior"] = n

    def set_robust(self, kind, enabled, delta=0.0):
        self.call("set_robust", kind, int(enabled), float(delta))

    def set_levels(self, kind, level):
        lv = _u8(level)
        self.call("set_levels", kind, _up(lv))

    def get_levels(self, kind):
        n = self.dims["Ep"] if kind == EDGE_POINT else self.dims["El"]
        lv = np.zeros(n, np.uint8)
        self.call("get_levels", kind, _up(lv))
        return lv

    def set_shard(self, rank, world, allreduce):
        """allreduce(dev_ptr:int, n:int, op:int, stream:int) -> None, raising on failure."""
        def _tramp(user, buf, n, op, stream):
            try:
                allreduce(buf, n, op, stream)
                return 0
            except Exception as e:  # surfaced as PLBA_ERR_EXCHANGE
                import traceback
                traceback.print_exc()
                return 1
        self._cb = ALLREDUCE_FN(_tramp)
        self.call("set_shard", rank, world, self._cb, None)

    def set_shard_native(self, rank, world, fn_addr, user_ptr):
        """plba_set_shard with a NATIVE plba_allreduce_fn (e.g. plba_rccl_allreduce of include/plba_rccl.h) and its user
        pointer: no Python frame runs inside the LM loop."""
        self._cb = C.cast(C.c_void_p(fn_addr), ALLREDUCE_FN)
        self.call("set_shard", rank, world, self._cb, C.c_void_p(user_ptr))

    def set_stream(self, stream_handle):
        self.call("set_stream", C.c_void_p(stream_handle))

    # -- solve -------------------------------------------------------------------------------
    def optimize(self, iters, abort=None):
        st = Stats()
        ab = _up(abort) if abort is not None else None
        self.call("optimize", int(iters), ab, C.byref(st))
        return st

    def gate_outliers(self, thresh=5.991):
        a, b = C.c_int(0), C.c_int(0)
        self.call("gate_outliers", float(thresh), C.byref(a), C.byref(b), allow_positive=True)
        return a.value, b.value

    def recompute_errors(self):
        self.call("recompute_errors")

    def cull_observations(self, thresh=5.991):
        """Culling decision of the call site after the final optimize (mapHandler.cpp:5541-5620): per-observation
        bad flags for points and lines (level-1 edges are re-evaluated on the final estimates first)."""
        Ep, El = self.dims.get("Ep", 0), self.dims.get("El", 0)
        bp, bl = np.zeros(max(Ep, 1), np.uint8), np.zeros(max(El, 1), np.uint8)
        a, b = C.c_int(0), C.c_int(0)
        self.call("cull_observations", float(thresh), _up(bp), _up(bl), C.byref(a), C.byref(b), allow_positive=True)
        return {"bad_points": bp[:Ep].astype(bool), "bad_lines": bl[:El].astype(bool), "n_points": a.value, "n_lines": b.value}

    def edge_chi2(self, kind):
        n = {EDGE_POINT: self.dims.get("Ep", 0), EDGE_LINE: self.dims.get("El", 0),
             EDGE_IMU_PVR: self.dims.get("M", 0), EDGE_IMU_BIAS: self.dims.get("M", 0), EDGE_PRIOR: 1}[kind]
        chi = np.zeros(n); dp = np.zeros(n, np.uint8)
        self.call("get_edge_chi2", kind, _dp(chi), _up(dp))
        return chi, dp

    def trace(self):
        n = C.c_int(0)
        self.call("get_trace", None, 0, C.byref(n))
        rows = (TraceRow * max(n.value, 1))()
        self.call("get_trace", rows, n.value, C.byref(n))
        return [dict(iteration=r.iteration, trial=r.trial, accepted=r.accepted, solver_ok=r.solver_ok,
                     lam=r.lam, chi2_current=r.chi2_current, chi2_trial=r.chi2_trial, scale=r.scale, rho=r.rho)
                for r in rows[:n.value]]

    # -- results -----------------------------------------------------------------------------
    def get_keyframes(self):
        K = self.dims["K"]
        P, V, q, dbg, dba = np.zeros((K, 3)), np.zeros((K, 3)), np.zeros((K, 4)), np.zeros((K, 3)), np.zeros((K, 3))
        self.call("get_keyframes", _dp(P), _dp(V), _dp(q), _dp(dbg), _dp(dba))
        return dict(P=P, V=V, q=q, dbg=dbg, dba=dba)

    def get_points(self):
        a = np.zeros((self.dims.get("Np", 0), 3))
        if len(a):
            self.call("get_points", _dp(a))
        return a

    def get_lines(self):
        a = np.zeros((self.dims.get("Nl", 0), 6))
        if len(a):
            self.call("get_lines", _dp(a))
        return a

    def save_state(self):
        self.call("save_state")

    def restore_state(self):
        self.call("restore_state")

    def set_marg_eps(self, eps):
        self.call("set_marg_eps", float(eps))

    def marginalize(self, first_kf=0, max_edges=50):
        pr = Prior()
        self.call("marginalize", int(first_kf), int(max_edges), C.byref(pr))
        n, nv = pr.n, pr.nv

        def arr(p, cnt, dt):
            return np.ctypeslib.as_array(p, shape=(cnt,)).astype(dt).copy() if cnt else np.zeros(0, dt)
        size = arr(pr.size, nv, np.int32)
        nx = int(sum(10 if s == 9 else 6 for s in size))
        out = dict(n=n, m=pr.m, vid=arr(pr.vid, nv, np.int32), size=size, idx=arr(pr.idx, nv, np.int32),
                   x0=arr(pr.x0, nx, np.float64),
                   J0=arr(pr.J0, n * n, np.float64).reshape(n, n).T.copy(),  # colmajor -> J0[r, c]
                   r0=arr(pr.r0, n, np.float64),
                   Ar=arr(pr.Ar, n * n, np.float64).reshape(n, n), br=arr(pr.br, n, np.float64))
        self.lib.fn["prior_free"](C.byref(pr))
        return out

    def lba_visual(self, T_kf_w, kf_loc, xyz, pq, po_pt, po_kf, uv, lo_ln, lo_kf, l3, cam, **opts):
        """MapHandler::levMarquardtOptimizationLBA (src/mapHandler.cpp:1441-2098) on the arrays of include/plba.h;
        returns the optimised poses / landmarks, the moved flags and the run's statistics."""
        o = LbaOptions()
        self.lib.fn["lba_default_options"](C.byref(o))
        for k, v in opts.items():
            if not hasattr(o, k):
                raise TypeError("unknown LBA option %r" % k)
            setattr(o, k, v)
        T = _f64(T_kf_w).reshape(-1, 16).copy(); K = T.shape[0]
        xyz = _f64(xyz).reshape(-1, 3).copy(); pq = _f64(pq).reshape(-1, 6).copy()
        po_pt, po_kf, lo_ln, lo_kf, loc = _i32(po_pt), _i32(po_kf), _i32(lo_ln), _i32(lo_kf), _i32(kf_loc)
        uv, l3 = _f64(uv).reshape(-1, 2), _f64(l3).reshape(-1, 3)
        Tout = np.zeros((K, 16)); pm = np.zeros(max(len(xyz), 1), np.uint8); lm = np.zeros(max(len(pq), 1), np.uint8)
        st = LbaStats()
        self.call("lba_visual", C.byref(o), K, _dp(T), _ip(loc), len(xyz), _dp(xyz), len(pq), _dp(pq),
                  len(po_pt), _ip(po_pt), _ip(po_kf), _dp(uv), len(lo_ln), _ip(lo_ln), _ip(lo_kf), _dp(l3),
                  float(cam[0]), float(cam[1]), float(cam[2]), float(cam[3]), _dp(Tout), _up(pm), _up(lm), C.byref(st))
        return dict(T=Tout.reshape(K, 4, 4), xyz=xyz, pq=pq, pt_moved=pm[:len(xyz)].astype(bool), ln_moved=lm[:len(pq)].astype(bool),
                    iterations=st.iterations, updates=st.updates, err_first=st.err_first, err_last=st.err_last, lam=st.lam,
                    solver_failed=st.solver_failed)

    # -- diagnostics -------------------------------------------------------------------------
    def debug_build(self, lam, do_solve=False):
        self.call("debug_build", float(lam), int(do_solve))

    def debug_get(self, what):
        n = C.c_size_t(0)
        self.call("debug_get", what.encode(), None, 0, C.byref(n))
        a = np.zeros(max(n.value, 1))
        self.call("debug_get", what.encode(), _dp(a), n.value, C.byref(n))
        return a[:n.value]

    def dense_solve(self, A, b, entry="dense_solve"):
        A = _f64(A); b = _f64(b)
        n = len(b)
        x = np.zeros(n); ok = C.c_int(0)
        self.call(entry, n, _dp(A), _dp(b), _dp(x), C.byref(ok))
        return x, bool(ok.value)

    def debug_dense_solve(self, A, b):
        return self.dense_solve(A, b, entry="debug_dense_solve")

    def preintegrate(self, sample_start, t, gyr, acc, t_prev, t_curr, bg, ba, gyr_meas_cov, acc_meas_cov):
        """KeyFrame::ComputeIMUPreIntSinceLastFrame for M intervals (plba_preintegrate); time stamps as np.longdouble."""
        ss = np.ascontiguousarray(sample_start, dtype=np.int32)
        M = len(ss) - 1
        ld = lambda v: np.ascontiguousarray(v, dtype=np.longdouble)
        t, t_prev, t_curr = ld(t), ld(t_prev), ld(t_curr)
        gyr, acc, bg, ba = _f64(gyr), _f64(acc), _f64(bg), _f64(ba)
        out = np.zeros((M, 142))
        lp = lambda v: v.ctypes.data_as(C.POINTER(C.c_longdouble))
        self.call("preintegrate", M, ss.ctypes.data_as(c_int32_p), lp(t), _dp(gyr), _dp(acc), lp(t_prev), lp(t_curr), _dp(bg), _dp(ba),
                  float(gyr_meas_cov), float(acc_meas_cov), _dp(out))
        return out

    def slide_window(self, d):
        """plba_slide_window: `d` as window.slide_delta() makes it — n_drop, the appended keyframes / IMU edges / landmarks / observations,
        optional drop masks and the new window's fixed flags.  Returns (point_map, line_map): each old landmark's new index or -1."""
        keep = []      # the arrays behind the struct's pointers stay alive until the call returns

        def f64(a, shape=None):
            if a is None:
                return None
            a = _f64(a, shape); keep.append(a); return _dp(a)

        def i32(a):
            if a is None:
                return None
            a = _i32(a); keep.append(a); return _ip(a)

        def u8(a):
            if a is None:
                return None
            a = _u8(a); keep.append(a); return _up(a)
        s = Slide()
        s.n_drop = int(d.get("n_drop", 0))
        s.drop_point, s.drop_line = u8(d.get("drop_point")), u8(d.get("drop_line"))
        s.drop_point_obs, s.drop_line_obs = u8(d.get("drop_point_obs")), u8(d.get("drop_line_obs"))
        k = d.get("kf")
        s.K_add = 0 if k is None else len(k["vid_pvr"])
        if k is not None:
            s.vid_pvr, s.vid_bias = i32(k["vid_pvr"]), i32(k.get("vid_bias"))
            s.P3, s.V3, s.q_xyzw4 = f64(k["P"]), f64(k["V"]), f64(k["q"])
            s.bg3, s.ba3, s.dbg3, s.dba3 = f64(k.get("bg")), f64(k.get("ba")), f64(k.get("dbg")), f64(k.get("dba"))
        s.fixed_pvr, s.fixed_bias = u8(d.get("fixed_pvr")), u8(d.get("fixed_bias"))
        im = d.get("imu")
        s.M_add = 0 if im is None else len(im["kf_i"])
        if im is not None:
            s.imu_kf_i, s.imu_kf_j = i32(im["kf_i"]), i32(im["kf_j"])
            s.preint142, s.info_pvr81, s.info_bias36 = f64(im["preint"], (-1, 142)), f64(im["info_pvr"], (-1, 81)), f64(im["info_bias"], (-1, 36))
        pts, lns = d.get("points"), d.get("lines")
        s.Np_add = 0 if pts is None else len(pts); s.xyz3 = f64(pts, (-1, 3)) if s.Np_add else None; s.point_fixed = u8(d.get("point_fixed"))
        s.Nl_add = 0 if lns is None else len(lns); s.sPeP6 = f64(lns, (-1, 6)) if s.Nl_add else None; s.line_fixed = u8(d.get("line_fixed"))
        s.Ep_add = len(d["po_pt"]) if d.get("po_pt") is not None else 0
        if s.Ep_add:
            s.po_pt, s.po_kf, s.uv2, s.po_inv_sigma2 = i32(d["po_pt"]), i32(d["po_kf"]), f64(d["po_uv"], (-1, 2)), f64(d.get("po_w"))
        s.El_add = len(d["lo_ln"]) if d.get("lo_ln") is not None else 0
        if s.El_add:
            s.lo_ln, s.lo_kf, s.l3, s.lo_inv_sigma2 = i32(d["lo_ln"]), i32(d["lo_kf"]), f64(d["lo_l"], (-1, 3)), f64(d.get("lo_w"))
        pm = np.zeros(max(self.dims.get("Np", 0), 1), np.int32); lm = np.zeros(max(self.dims.get("Nl", 0), 1), np.int32)
        self.call("slide_window", C.byref(s), _ip(pm), _ip(lm))
        pm, lm = pm[:self.dims.get("Np", 0)], lm[:self.dims.get("Nl", 0)]
        sz = np.zeros(6, np.int32)
        self.call("get_sizes", _ip(sz))      # (the library merged the lists: its counts serve the getters)
        for k, v in zip(("K", "Np", "Nl", "Ep", "El", "M"), sz):
            self.dims[k] = int(v)
        return pm, lm

    # -- convenience -------------------------------------------------------------------------
    def upload_window(self, w):
        """Upload a synthetic window (window.make_window) following the reference's graph
        construction order (mapHandler.cpp:5799-6034)."""
        c = w["cam"]
        self.set_camera(c["fx"], c["fy"], c["cx"], c["cy"], c["Rbc"], c["Pbc"])
        self.set_gravity(w["gw"])
        k = w["kf"]
        self.set_keyframes(k["vid_pvr"], k["vid_bias"], k["P"], k["V"], k["q"], k["bg"], k["ba"], k["dbg"], k["dba"],
                           k["fixed_pvr"], k["fixed_bias"])
        self.set_points(w["points"], w.get("point_fixed"))
        self.set_lines(w["lines"], w.get("line_fixed"))
        self.set_point_obs(w["po_pt"], w["po_kf"], w["po_uv"], w["po_w"])
        self.set_line_obs(w["lo_ln"], w["lo_kf"], w["lo_l"], w["lo_w"])
        if w.get("imu") is not None:
            im = w["imu"]
            self.set_imu_edges(im["kf_i"], im["kf_j"], im["preint"], im["info_pvr"], im["info_bias"])
        else:      # a handle keeps its arrays until they are set again: a window without IMU edges clears the previous window's
            self.set_imu_edges(np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros((0, 142)), np.zeros((0, 81)), np.zeros((0, 36)))
        self.set_prior(w.get("prior"))
        for kind, d in w["huber"].items():
            self.set_robust(kind, True, d)
