#!/usr/bin/env python
"""Independent golden vectors for the per-edge arithmetic of the path, evaluated with mpmath at 40 digits and written to
tests/golden/mp_vectors.json.      python tests/golden/make_mp_vectors.py

SURVEY 8(c): the reference holds no fixture for this path and cannot be built here, so the oracle (oracle/plba_oracle.c) was pinned
only by finite differences and by vectors it produced itself.  This script is a SECOND restatement of the same formulas, written from
the reference's source text alone (file:line cited at each function), in a different language and number system, sharing no code with
the oracle or the kernels: SO3 exp / log / Jr / Jr^-1 with the reference's thresholds and branch structure (IMU/so3.cpp:32-89, 199-280:
theta < 1e-5, n < 1e-10, the overwritten |w| < eps branch, atan not atan2, the truncated series of exp), the point edge
(IMU/g2otypes.h:230-275, .cpp:286-341), the line edge including the world-frame position block of SURVEY B-Q1 (.h:783-825,
.cpp:1306-1359), the IMU PVR residual (.cpp:27-92), the bias residual (.cpp:236-262), the prior residual (.cpp:1423-1475) and the
NavState oplus (IMU/NavState.cpp:69-121).  Inputs are doubles (exact in mpmath); outputs are rounded to double at the end.
tests/test_mp_vectors.py holds the oracle, the device formulas compiled for the host, and the HIP kernels to these values."""
import json
import os
import sys

import mpmath as mp
import numpy as np

mp.mp.dps = 40
HERE = os.path.dirname(os.path.abspath(__file__))
SMALL_EPS = mp.mpf("1e-10")      # IMU/so3.h:35


def M(a):
    return mp.matrix(a)


def hat(v):      # IMU/so3.cpp:283-290
    return mp.matrix([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])


def vnorm(v):
    return mp.sqrt(sum(x * x for x in v))


def quat_normalized(q):      # (x, y, z, w), Eigen coeffs() /= norm()
    n = mp.sqrt(sum(x * x for x in q))
    return [x / n for x in q]


def quat_to_R(q):      # Eigen QuaternionBase::toRotationMatrix
    x, y, z, w = q
    tx, ty, tz = 2 * x, 2 * y, 2 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    return mp.matrix([[1 - (tyy + tzz), txy - twz, txz + twy], [txy + twz, 1 - (txx + tzz), tyz - twx], [txz - twy, tyz + twx, 1 - (txx + tyy)]])


def R_to_quat(m):      # Eigen quaternionbase_assign_impl<Matrix3d>
    t = m[0, 0] + m[1, 1] + m[2, 2]
    q = [mp.mpf(0)] * 4
    if t > 0:
        t = mp.sqrt(t + 1)
        q[3] = t / 2
        t = mp.mpf(1) / 2 / t
        q[0] = (m[2, 1] - m[1, 2]) * t; q[1] = (m[0, 2] - m[2, 0]) * t; q[2] = (m[1, 0] - m[0, 1]) * t
    else:
        i = 0
        if m[1, 1] > m[0, 0]: i = 1
        if m[2, 2] > m[i, i]: i = 2
        j, k = (i + 1) % 3, (i + 2) % 3
        t = mp.sqrt(m[i, i] - m[j, j] - m[k, k] + 1)
        q[i] = t / 2
        t = mp.mpf(1) / 2 / t
        q[3] = (m[k, j] - m[j, k]) * t; q[j] = (m[j, i] + m[i, j]) * t; q[k] = (m[k, i] + m[i, k]) * t
    return q


def quat_mul(a, b):      # Eigen product, (x, y, z, w)
    ax, ay, az, aw = a; bx, by, bz, bw = b
    return [aw * bx + ax * bw + ay * bz - az * by, aw * by + ay * bw + az * bx - ax * bz, aw * bz + az * bw + ax * by - ay * bx, aw * bw - ax * bx - ay * by - az * bz]


def quat_conj(q):
    return [-q[0], -q[1], -q[2], q[3]]


def so3_exp(w):      # IMU/so3.cpp:257-280 (then the SO3(Quaterniond) constructor normalises)
    theta = vnorm(w)
    half = theta / 2
    real = mp.cos(half)
    if theta < SMALL_EPS:
        t2 = theta * theta
        imag = mp.mpf("0.5") - mp.mpf("0.0208333") * t2 + mp.mpf("0.000260417") * t2 * t2
    else:
        imag = mp.sin(half) / theta
    return quat_normalized([imag * w[0], imag * w[1], imag * w[2], real])


def so3_log(q):      # IMU/so3.cpp:206-247: the |w| < eps result is overwritten unconditionally; atan, not atan2 (SURVEY B-Q13)
    n = mp.sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2])
    w = q[3]
    if n < SMALL_EPS:
        f = 2 / w - 2 * (n * n) / (w * w * w)
    else:
        f = 2 * mp.atan(n / w) / n
    return [f * q[0], f * q[1], f * q[2]]


def so3_Jr(w):      # IMU/so3.cpp:32-49
    theta = vnorm(w)
    I = mp.eye(3)
    if theta < mp.mpf("0.00001"):
        return I
    K = hat([x / theta for x in w])
    return I - (1 - mp.cos(theta)) / theta * K + (1 - mp.sin(theta) / theta) * K * K


def so3_JrInv(w):      # IMU/so3.cpp:50-68
    theta = vnorm(w)
    I = mp.eye(3)
    if theta < mp.mpf("0.00001"):
        return I
    K = hat([x / theta for x in w])
    return I + mp.mpf("0.5") * hat(w) + (1 - (1 + mp.cos(theta)) * theta / (2 * mp.sin(theta))) * K * K


def col(v):
    return mp.matrix([[x] for x in v])


def to_f(x):
    if isinstance(x, mp.matrix):
        return [[float(x[i, j]) for j in range(x.cols)] for i in range(x.rows)]
    if isinstance(x, (list, tuple)):
        return [to_f(y) for y in x]
    return float(x)


def mpv(a):
    return [mp.mpf(float(x)) for x in a]


# ---- edges -------------------------------------------------------------------------------------------------------------------------
def cam_Pc(cam, nav, Pw):      # IMU/g2otypes.h:243-260: Pc = Rcb Rwb^T (Pw - Pwb) - Rcb Pbc, Rwb = Get_RotMatrix() of the stored quaternion
    Rbc, Pbc = cam["Rbc"], cam["Pbc"]
    Rwb = quat_to_R(nav["q"])
    Rcb = Rbc.T
    return Rcb * Rwb.T * (col(Pw) - col(nav["P"])) - Rcb * col(Pbc), Rcb, Rwb


def point_edge(cam, nav, Pw, obs):      # error: g2otypes.h:230-236; Jacobians: g2otypes.cpp:286-341
    Pc, Rcb, Rwb = cam_Pc(cam, nav, Pw)
    x, y, z = Pc[0], Pc[1], Pc[2]
    e = [obs[0] - (x / z * cam["fx"] + cam["cx"]), obs[1] - (y / z * cam["fy"] + cam["cy"])]
    Jpi = mp.matrix([[cam["fx"], 0, -x / z * cam["fx"]], [0, cam["fy"], -y / z * cam["fy"]]]) / z
    Ji = -Jpi * Rcb * Rwb.T
    JdP = -Jpi * (-Rcb)
    Paux = Rcb * Rwb.T * (col(Pw) - col(nav["P"]))
    JdR = -Jpi * (hat([Paux[0], Paux[1], Paux[2]]) * Rcb)
    return e, Ji, JdP, JdR, bool(z > 0)


def line_edge(cam, nav, L, obs):      # error: g2otypes.h:783-792; Jacobians: g2otypes.cpp:1306-1359 (position block in the WORLD frame: SURVEY B-Q1)
    Ps, Rcb, Rwb = cam_Pc(cam, nav, L[:3])
    Pe, _, _ = cam_Pc(cam, nav, L[3:])
    def proj(P):
        return [P[0] / P[2] * cam["fx"] + cam["cx"], P[1] / P[2] * cam["fy"] + cam["cy"]]
    us, ue = proj(Ps), proj(Pe)
    e = [obs[0] * us[0] + obs[1] * us[1] + obs[2], obs[0] * ue[0] + obs[1] * ue[1] + obs[2]]
    de_p = mp.matrix([[obs[0], obs[1]]])
    def dproj(P):
        return mp.matrix([[cam["fx"] / P[2], 0, -cam["fx"] * P[0] / (P[2] * P[2])], [0, cam["fy"] / P[2], -cam["fy"] * P[1] / (P[2] * P[2])]])
    M3 = Rcb * Rwb.T
    Jl_s = de_p * dproj(Ps) * M3      # on sP
    Jl_e = de_p * dproj(Pe) * M3      # on eP
    JP_s = de_p * dproj(Ps) * (-M3)
    JP_e = de_p * dproj(Pe) * (-M3)
    ds = Rwb.T * (col(L[:3]) - col(nav["P"])); de_ = Rwb.T * (col(L[3:]) - col(nav["P"]))
    JR_s = de_p * dproj(Ps) * (Rcb * hat([ds[0], ds[1], ds[2]]))
    JR_e = de_p * dproj(Pe) * (Rcb * hat([de_[0], de_[1], de_[2]]))
    return e, Jl_s, Jl_e, JP_s, JP_e, JR_s, JR_e, bool(Ps[2] > 0 and Pe[2] > 0)


def so3_of_R(Rm):      # Sophus::SO3(Matrix3d): Quaterniond(R), normalised
    return quat_normalized(R_to_quat(Rm))


def pvr_error(gw, ni, nj, nb, pre):      # IMU/g2otypes.cpp:27-92; Get_R() returns a copy-normalised SO3
    qi, qj = quat_normalized(ni["q"]), quat_normalized(nj["q"])
    Ri, Rj = quat_to_R(qi), quat_to_R(qj)
    dT = pre["dt"]; dT2 = dT * dT
    RiT = quat_normalized(quat_conj(qi))
    RiTm = quat_to_R(RiT)
    g = col(gw)
    rP = RiTm * (col(nj["P"]) - col(ni["P"]) - col(ni["V"]) * dT - g * dT2 / 2) - (col(pre["dP"]) + pre["JPg"] * col(nb["dbg"]) + pre["JPa"] * col(nb["dba"]))
    rV = RiTm * (col(nj["V"]) - col(ni["V"]) - g * dT) - (col(pre["dV"]) + pre["JVg"] * col(nb["dbg"]) + pre["JVa"] * col(nb["dba"]))
    dRij = so3_of_R(pre["dR"])
    v = pre["JRg"] * col(nb["dbg"])
    dR_dbg = so3_exp([v[0], v[1], v[2]])
    A = quat_normalized(quat_mul(dRij, dR_dbg))
    r = quat_normalized(quat_mul(quat_normalized(quat_mul(quat_normalized(quat_conj(A)), RiT)), qj))
    rPhi = so3_log(r)
    return [rP[0], rP[1], rP[2], rV[0], rV[1], rV[2], rPhi[0], rPhi[1], rPhi[2]]


def bias_error(ni, nj):      # IMU/g2otypes.cpp:236-262
    return [(nj["bg"][t] + nj["dbg"][t]) - (ni["bg"][t] + ni["dbg"][t]) for t in range(3)] + [(nj["ba"][t] + nj["dba"][t]) - (ni["ba"][t] + ni["dba"][t]) for t in range(3)]


def oplus_pvr(n, u):      # IMU/NavState.cpp:69-98
    q = quat_normalized(n["q"])
    R = quat_to_R(q)
    P = col(n["P"]) + R * col(u[0:3])
    V = [n["V"][t] + u[3 + t] for t in range(3)]
    qn = quat_normalized(quat_mul(q, so3_exp(u[6:9])))
    return [P[0], P[1], P[2]] + V + qn


def prior_dx_pvr(n, x0):      # IMU/g2otypes.cpp:1464-1466: world-frame P - P0, V - V0, 2 vec(q0^-1 q(Rwb)) without sign normalisation (SURVEY B-Q5)
    q0 = [x0[6], x0[7], x0[8], x0[9]]
    n2 = sum(x * x for x in q0)
    q0inv = [-q0[0] / n2, -q0[1] / n2, -q0[2] / n2, q0[3] / n2]      # Eigen inverse(): conjugate / squaredNorm
    qR = R_to_quat(quat_to_R(n["q"]))      # Quaterniond(Get_RotMatrix())
    d = quat_mul(q0inv, qR)
    return [n["P"][t] - x0[t] for t in range(3)] + [n["V"][t] - x0[3 + t] for t in range(3)] + [2 * d[0], 2 * d[1], 2 * d[2]]


# ---- deterministic inputs ------------------------------------------------------------------------------------------------------------
def nav_state(rng, small_rot=None):
    w = rng.normal(size=3) * 0.8 if small_rot is None else small_rot
    q = to_f(so3_exp(mpv(w)))
    # perturb the stored quaternion's norm a little: the reference normalises copies, not the stored value
    q = (np.array(q) * (1 + 1e-9 * rng.normal())).tolist()
    return dict(P=rng.normal(size=3).tolist(), V=rng.normal(size=3).tolist(), q=q, bg=(rng.normal(size=3) * 1e-2).tolist(), ba=(rng.normal(size=3) * 1e-1).tolist(),
                dbg=(rng.normal(size=3) * 1e-3).tolist(), dba=(rng.normal(size=3) * 1e-2).tolist())


def nav_mp(n):
    return {k: mpv(v) for k, v in n.items()}


def main():
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    import __graft_entry__ as ge
    win = ge.load_package().window
    rng = np.random.default_rng(0x601D10)
    out = {"note": "mpmath 40 digits, rounded to double; see make_mp_vectors.py", "so3": [], "point": [], "line": [], "pvr": [], "bias": [], "oplus": [], "prior_dx": []}
    camf = dict(fx=win.FX, fy=win.FY, cx=win.CX, cy=win.CY, Rbc=win.T_BS[:3, :3].tolist(), Pbc=win.T_BS[:3, 3].tolist())
    cam = dict(fx=mp.mpf(camf["fx"]), fy=mp.mpf(camf["fy"]), cx=mp.mpf(camf["cx"]), cy=mp.mpf(camf["cy"]), Rbc=mp.matrix([[mp.mpf(float(x)) for x in r] for r in camf["Rbc"]]), Pbc=mpv(camf["Pbc"]))
    out["cam"] = camf
    # SO3: norms across every threshold of the reference (1e-10 for exp / log, 1e-5 for Jr / JrInv), generic and near-pi angles
    dirs = [np.array(d, float) / np.linalg.norm(d) for d in ([1, 0, 0], [0.3, -0.5, 0.8], [-0.6, 0.7, 0.2], [0.1, 0.9, -0.4])]
    norms = [0.0, 1e-13, 5e-11, 9.9e-11, 1.1e-10, 3e-8, 9e-6, 1.1e-5, 1e-3, 0.3, 1.5, 3.0, 3.1]
    for k, th in enumerate(norms):
        w = (dirs[k % 4] * th).tolist()
        wm = mpv(w)
        q = so3_exp(wm)
        qf = to_f(q)      # the log is evaluated on the DOUBLE quaternion handed to the implementations under test
        out["so3"].append(dict(w=w, exp=qf, log_of_exp=to_f(so3_log(mpv(qf))), Jr=to_f(so3_Jr(wm)), JrInv=to_f(so3_JrInv(wm))))
    # quaternions with a tiny vector part (the n < 1e-10 branch of log) and a negative real part
    for q in ([3e-11, -2e-11, 1e-11, 1.0], [1e-12, 0.0, 0.0, -1.0], [0.6, -0.3, 0.2, -0.7141428428542850]):
        out["so3"].append(dict(q=q, log=to_f(so3_log(mpv(q)))))
    for k in range(14):
        small = None if k < 10 else dirs[k % 4] * [1e-12, 5e-11, 2e-6, 2e-5][k - 10]
        n = nav_state(rng, small)
        nm = nav_mp(n)
        R = np.array(to_f(quat_to_R(nm["q"])))
        Rbc, Pbc = np.array(camf["Rbc"]), np.array(camf["Pbc"])
        Pc = np.array([rng.uniform(-2, 2), rng.uniform(-1, 1), rng.uniform(0.5, 8)])
        Pw = (R @ (Rbc @ Pc + Pbc) + np.array(n["P"])).tolist()
        obs = rng.uniform(0, 700, 2).tolist()
        e, Ji, JdP, JdR, dpos = point_edge(cam, nm, mpv(Pw), mpv(obs))
        out["point"].append(dict(nav=n, Pw=Pw, obs=obs, e=to_f(e), Jl=to_f(Ji), JdP=to_f(JdP), JdR=to_f(JdR), depth_positive=dpos))
        Pc2 = Pc + rng.normal(size=3) * 0.4; Pc2[2] = abs(Pc2[2]) + 0.3
        L = Pw + (R @ (Rbc @ Pc2 + Pbc) + np.array(n["P"])).tolist()
        l = rng.normal(size=3); l[:2] /= np.hypot(l[0], l[1]); l[2] *= 50
        e, Js, Je, JPs, JPe, JRs, JRe, dpos = line_edge(cam, nm, mpv(L), mpv(l))
        out["line"].append(dict(nav=n, L=L, obs=l.tolist(), e=to_f(e), Jl_s=to_f(Js), Jl_e=to_f(Je), JP_s=to_f(JPs), JP_e=to_f(JPe), JR_s=to_f(JRs), JR_e=to_f(JRe), depth_positive=dpos))
    gw = [0.0, 0.0, -9.81]
    for k in range(12):
        ni = nav_state(rng)
        # keyframe j: a plausible 0.25 s later (so that the rotation residual is small), or identical rotation (residual in the n < 1e-10 branch)
        dR = {0: np.zeros(3), 1: dirs[1] * 3e-11, 2: dirs[2] * 4e-6}.get(k, rng.normal(size=3) * 0.05)
        nj = nav_state(rng)
        nj["q"] = to_f(quat_normalized(quat_mul(quat_normalized(mpv(ni["q"])), so3_exp(mpv(dR)))))
        nj["bg"], nj["ba"] = ni["bg"], ni["ba"]
        dt = 0.25
        pre = dict(dP=(rng.normal(size=3) * 0.1).tolist(), dV=(rng.normal(size=3) * 0.3).tolist(), dR=np.eye(3).tolist() if k < 3 else to_f(quat_to_R(so3_exp(mpv(rng.normal(size=3) * 0.05)))),
                   JPg=(rng.normal(size=(3, 3)) * 0.01).tolist(), JPa=(rng.normal(size=(3, 3)) * 0.03).tolist(), JVg=(rng.normal(size=(3, 3)) * 0.05).tolist(),
                   JVa=(rng.normal(size=(3, 3)) * 0.25).tolist(), JRg=(-np.eye(3) * 0.25 + rng.normal(size=(3, 3)) * 0.01).tolist(), dt=dt)
        if k < 3: ni["dbg"] = [0.0, 0.0, 0.0]
        prem = dict(dP=mpv(pre["dP"]), dV=mpv(pre["dV"]), dR=mp.matrix([[mp.mpf(float(x)) for x in r] for r in pre["dR"]]), dt=mp.mpf(dt),
                    **{key: mp.matrix([[mp.mpf(float(x)) for x in r] for r in pre[key]]) for key in ("JPg", "JPa", "JVg", "JVa", "JRg")})
        out["pvr"].append(dict(gw=gw, navi=ni, navj=nj, pre=pre, e=to_f(pvr_error(mpv(gw), nav_mp(ni), nav_mp(nj), nav_mp(ni), prem))))
        out["bias"].append(dict(navi=ni, navj=nj, e=to_f(bias_error(nav_mp(ni), nav_mp(nj)))))
        u = (rng.normal(size=9) * np.array([0.05] * 3 + [0.1] * 3 + [[0.02] * 3, [1e-11] * 3, [3e-6] * 3][k % 3])).tolist()
        out["oplus"].append(dict(nav=ni, u=u, out=to_f(oplus_pvr(nav_mp(ni), mpv(u)))))
        x0 = (np.array(ni["P"] + ni["V"] + ni["q"]) + np.concatenate([rng.normal(size=6) * 0.01, np.zeros(4)])).tolist()
        x0[6:10] = to_f(quat_normalized(quat_mul(quat_normalized(mpv(ni["q"])), so3_exp(mpv(rng.normal(size=3) * 0.01)))))
        if k % 2: x0[6:10] = [-x for x in x0[6:10]]      # q0 with the opposite sign: the reference does not normalise it
        out["prior_dx"].append(dict(nav=nj, x0=x0, dx=to_f(prior_dx_pvr(nav_mp(nj), mpv(x0)))))
    with open(os.path.join(HERE, "mp_vectors.json"), "w") as f:
        json.dump(out, f)
    print("wrote mp_vectors.json:", {k: len(v) for k, v in out.items() if isinstance(v, list)})


if __name__ == "__main__":
    main()
