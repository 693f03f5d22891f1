"""Loader of the CPU oracle (oracle/plba_oracle.c) — TEST INFRASTRUCTURE ONLY.

May be imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  It binds
the orc_* entry points with the same ctypes signature table the product uses for plba_*.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_build", "libplba_oracle.so")
_lib = None


def _abi():
    sys.path.insert(0, os.path.dirname(_HERE))
    try:
        import __graft_entry__ as g
    finally:
        sys.path.pop(0)
    return g.load_package().abi


def build(force=False):
    src = os.path.join(_HERE, "plba_oracle.c")
    if force or not os.path.exists(LIB_PATH) or not os.path.exists(os.path.join(_HERE, "_build", "libplba_oracle_fast.so")) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "all"] + (["-B"] if force else []))
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = _abi().Lib(LIB_PATH, "orc_")
        c = _lib.cdll
        dp = C.POINTER(C.c_double)
        ip = C.POINTER(C.c_int)
        sigs = {
            "orc_eval_point_edge": [dp, dp, dp, dp, dp, dp, dp, ip],
            "orc_eval_line_edge": [dp, dp, dp, dp, C.c_int, dp, dp, dp, ip],
            "orc_eval_linepoint_edge": [dp, dp, dp, dp, dp],
            "orc_eval_pvr_edge": [dp] * 9,
            "orc_nav_oplus_pvr": [dp, dp, dp],
            "orc_nav_oplus_bias": [dp, dp, dp],
            "orc_so3_exp": [dp, dp], "orc_so3_log": [dp, dp], "orc_so3_jr": [dp, dp], "orc_so3_jrinv": [dp, dp],
            "orc_quat_to_R": [dp, dp], "orc_R_to_quat": [dp, dp],
            "orc_huber": [C.c_double, C.c_double, dp],
            "orc_sym_eig": [dp, C.c_int, dp, dp],
            "orc_preint_update": [dp, dp, dp, C.c_double, C.c_double, C.c_double],
            "orc_se3_exp": [dp, dp, dp], "orc_se3_mul": [dp] * 6, "orc_se3_map": [dp] * 4, "orc_se3_log": [dp, dp, dp],
            "orc_se3_oplus": [dp] * 5,
            "orc_eval_gyrbias_edge": [dp] * 7,
            "orc_gyrbias_estimate": [C.c_int, dp, dp, dp, dp, dp, C.c_int, C.c_int, dp, dp],
            "orc_pgo": [C.c_int, C.POINTER(C.c_int32), dp, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32), dp, dp, C.c_int, C.c_double, C.c_int, C.c_int, dp],
            "orc_se3_edge_error": [dp, dp, dp, dp], "orc_se3_vertex_oplus": [dp, dp, dp],
            "orc_eval_se3_edge": [C.c_int, dp, dp, dp, dp, dp, dp, dp, dp, ip],
        }
        for name, args in sigs.items():
            f = getattr(c, name)
            f.restype = None
            f.argtypes = args
    return _lib


def new_problem(**opts):
    return _abi().Problem(lib(), **opts)


FAST_LIB_PATH = os.path.join(_HERE, "_build", "libplba_oracle_fast.so")
_fast = None


def new_fast_problem(threads=None, **opts):
    """bench.py's stronger CPU leg: the same source built -O3 -march=native -fopenmp with an envelope (sparse) Cholesky of the reduced
    system (oracle/Makefile, PLBA_ORACLE_FAST).  Never the parity checker: its summation orders differ from the faithful build."""
    global _fast
    if _fast is None:
        build()
        _fast = _abi().Lib(FAST_LIB_PATH, "orc_")
    if threads is not None:
        omp = C.CDLL("libgomp.so.1")
        omp.omp_set_num_threads(int(threads))
    return _abi().Problem(_fast, **opts)


QUAD_LIB_PATH = os.path.join(_HERE, "_build", "libplba_oracle_quad.so")
_quad = None


def new_quad_problem(**opts):
    """The arbiter of ill-conditioned cases: the same source in __float128 (oracle/make_quad.py builds it; real-double entry points
    orcq_* for upload_window / optimize / gate_outliers / the getters only).  Container only: tests use the fixtures it generated."""
    global _quad
    if _quad is None:
        src = os.path.join(_HERE, "plba_oracle.c")
        if not os.path.exists(QUAD_LIB_PATH) or os.path.getmtime(QUAD_LIB_PATH) < os.path.getmtime(src):
            subprocess.run([sys.executable, os.path.join(_HERE, "make_quad.py")], check=True)
        _quad = _abi().Lib(QUAD_LIB_PATH, "orcq_", optional=True)
    return _abi().Problem(_quad, **opts)


def _d(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def cam_vec(cam):
    return np.concatenate([[cam["fx"], cam["fy"], cam["cx"], cam["cy"]], np.asarray(cam["Rbc"]).ravel(),
                           np.asarray(cam["Pbc"]).ravel()]).astype(np.float64)


def nav_vec(P, V, q, bg=None, ba=None, dbg=None, dba=None):
    z = np.zeros(3)
    return np.concatenate([P, V, q, z if bg is None else bg, z if ba is None else ba,
                           z if dbg is None else dbg, z if dba is None else dba]).astype(np.float64)


def eval_point_edge(cam, nav, Pw, obs, jac=True):
    c = lib().cdll
    cam, nav, Pw, obs = (np.ascontiguousarray(a, dtype=np.float64) for a in (cam, nav, Pw, obs))
    e, Ji, Jj, d = np.zeros(2), np.zeros(6), np.zeros(18), C.c_int(0)
    c.orc_eval_point_edge(_d(cam), _d(nav), _d(Pw), _d(obs), _d(e), _d(Ji) if jac else None, _d(Jj) if jac else None, C.byref(d))
    return e, Ji.reshape(2, 3), Jj.reshape(2, 9), bool(d.value)


def eval_line_edge(cam, nav, L, obs, jac=True, fix_q1=0):
    c = lib().cdll
    cam, nav, L, obs = (np.ascontiguousarray(a, dtype=np.float64) for a in (cam, nav, L, obs))
    e, Ji, Jj, d = np.zeros(3), np.zeros(18), np.zeros(27), C.c_int(0)
    c.orc_eval_line_edge(_d(cam), _d(nav), _d(L), _d(obs), fix_q1, _d(e), _d(Ji) if jac else None, _d(Jj) if jac else None, C.byref(d))
    return e, Ji.reshape(3, 6), Jj.reshape(3, 9), bool(d.value)


def eval_linepoint_edge(cam, nav, Pw, obs):
    c = lib().cdll
    cam, nav, Pw, obs = (np.ascontiguousarray(a, dtype=np.float64) for a in (cam, nav, Pw, obs))
    e = np.zeros(3)
    c.orc_eval_linepoint_edge(_d(cam), _d(nav), _d(Pw), _d(obs), _d(e))
    return e


def eval_pvr_edge(gw, navi, navj, navb, pre, jac=True):
    c = lib().cdll
    gw, navi, navj, navb, pre = (np.ascontiguousarray(a, dtype=np.float64) for a in (gw, navi, navj, navb, pre))
    e, J0, J1, J2 = np.zeros(9), np.zeros(81), np.zeros(81), np.zeros(54)
    c.orc_eval_pvr_edge(_d(gw), _d(navi), _d(navj), _d(navb), _d(pre), _d(e), _d(J0) if jac else None, _d(J1), _d(J2))
    return e, J0.reshape(9, 9), J1.reshape(9, 9), J2.reshape(9, 6)


def nav_oplus_pvr(nav, u):
    c = lib().cdll
    nav, u = np.ascontiguousarray(nav, dtype=np.float64), np.ascontiguousarray(u, dtype=np.float64)
    o = np.zeros(22)
    c.orc_nav_oplus_pvr(_d(nav), _d(u), _d(o))
    return o


def nav_oplus_bias(nav, u):
    c = lib().cdll
    nav, u = np.ascontiguousarray(nav, dtype=np.float64), np.ascontiguousarray(u, dtype=np.float64)
    o = np.zeros(22)
    c.orc_nav_oplus_bias(_d(nav), _d(u), _d(o))
    return o


def _v2v(name, a, nout):
    c = lib().cdll
    a = np.ascontiguousarray(a, dtype=np.float64)
    o = np.zeros(nout)
    getattr(c, name)(_d(a), _d(o))
    return o


def so3_exp(w): return _v2v("orc_so3_exp", w, 4)
def so3_log(q): return _v2v("orc_so3_log", q, 3)
def so3_jr(w): return _v2v("orc_so3_jr", w, 9).reshape(3, 3)
def so3_jrinv(w): return _v2v("orc_so3_jrinv", w, 9).reshape(3, 3)
def quat_to_R(q): return _v2v("orc_quat_to_R", q, 9).reshape(3, 3)
def R_to_quat(R): return _v2v("orc_R_to_quat", np.asarray(R).ravel(), 4)


def huber(e, delta):
    r = np.zeros(3)
    lib().cdll.orc_huber(float(e), float(delta), _d(r))
    return r


def sym_eig(A):
    A = np.ascontiguousarray(A, dtype=np.float64)
    n = A.shape[0]
    w, V = np.zeros(n), np.zeros((n, n))
    lib().cdll.orc_sym_eig(_d(A), n, _d(w), _d(V))
    return w, V


def preint_update(pre142, omega, acc, dt, gyr_cov, acc_cov):
    p = np.ascontiguousarray(pre142, dtype=np.float64).copy()
    o, a = np.ascontiguousarray(omega, dtype=np.float64), np.ascontiguousarray(acc, dtype=np.float64)
    lib().cdll.orc_preint_update(_d(p), _d(o), _d(a), float(dt), float(gyr_cov), float(acc_cov))
    return p
