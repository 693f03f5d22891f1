#!/usr/bin/env python
"""A THIRD implementation of the g2o control flow of the path — independent of oracle/plba_oracle.c and of the HIP library — run on the
3-keyframe / 20-point / 5-line / IMU window SURVEY 8(c) names, written to tests/golden/lm_trace.json.

    python tests/golden/make_lm_trace.py

What the per-edge vectors of make_mp_vectors.py did for the arithmetic of one edge, this does for everything ABOVE the edges: one whole
`optimize(5)` -> gate -> `optimize(10)` of MapHandler::localBundleAdjustmentWithImuAndMarg (src/mapHandler.cpp:6038-6069) with
  * g2o's Levenberg-Marquardt (computeLambdaInit = tau * max |H_jj| over pose AND landmark diagonals, the rho
    test with computeScale = sum x_j (lambda x_j + b_j) + 1e-3, the lambda schedule 1 - (2 rho - 1)^3 clipped to [1/3, 2/3], nu doubling,
    at most 10 trials, Terminate when rho == 0 or the trials ran out, lambda re-initialised by every optimize() call),
  * buildSystem / constructQuadraticForm with the Huber kernel as g2o applies it (A.4, A.8: rho' scales Omega and the gradient, the
    second-order term is not used; activeRobustChi2 sums rho_0), fixed vertices skipped,
  * push / pop, the per-edge error CACHE that the gate reads (A.7: after a rejected last trial the chi2 is the rejected point's while
    isDepthPositive looks at the restored estimate), level-1 edges and landmarks without active edges leaving stage 2 (A.1),
  * the call-site protocol (Huber off on point / line edges in stage 2, kept on the IMU edges: SURVEY B-Q11).
It is written from SURVEY.md App. A.3 - A.8 and the reference's edge sources alone (IMU/g2otypes.cpp:94-234 for the PVR edge's Jacobians;
the residuals, the point / line Jacobians and the vertex oplus are make_mp_vectors.py's restatements), in mpmath at 40 digits, on DENSE
normal equations over [pose vertices by id | points | lines] solved by one Cholesky factorisation — no Schur complement, no block
structure, no chain elimination: mathematically what g2o's BlockSolverX + LinearSolverEigen compute (A.5 / A.6), structurally nothing
like the oracle's or the device's solver.  Inputs (the window) are doubles and exact in mpmath; every trial's lambda, chi2, scale, rho
and decision, the gate's counts and the final estimates are rounded to double at the end.  tests/test_lm_trace.py holds the oracle
(CPU) and the HIP path — record-based and fused passes — (GPU) to them."""
import json
import os
import sys

import mpmath as mp
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_mp_vectors as E  # noqa: E402  (the 40-digit per-edge restatements; sets mp.dps = 40)

CASE = dict(K=3, Np=20, Nl=5, seed=0x601D31)
TAU, GOOD_LO, GOOD_HI, MAX_TRIALS = mp.mpf("1e-5"), mp.mpf(1) / 3, mp.mpf(2) / 3, 10      # g2o defaults, SURVEY App. A.3
CHI2_GATE = mp.mpf("5.991")      # src/mapHandler.cpp:6051,6061


def f2mp(x):
    return mp.mpf(float(x))


def mat(a):
    a = np.asarray(a, float)
    return mp.matrix([[f2mp(v) for v in r] for r in a])


# ---- the PVR edge's Jacobians: IMU/g2otypes.cpp:94-234 ---------------------------------------------------------------------------------
def pvr_jacobians(gw, ni, nj, nb, pre, err):
    Ri, Rj = E.quat_to_R(ni["q"]), E.quat_to_R(nj["q"])      # Get_RotMatrix(): the stored quaternion's matrix (no copy-normalisation)
    dT = pre["dt"]; dT2 = dT * dT
    g = E.col(gw)
    RiT, RjT = Ri.T, Rj.T
    rPhi = err[6:9]
    JrInv = E.so3_JrInv(rPhi)
    JRg = pre["JRg"]
    I3, Z9 = mp.eye(3), mp.zeros(9, 9)
    Ji = Z9.copy()
    a = RiT * (E.col(nj["P"]) - E.col(ni["P"]) - E.col(ni["V"]) * dT - g * dT2 / 2)
    c = RiT * (E.col(nj["V"]) - E.col(ni["V"]) - g * dT)
    Ji[0:3, 0:3] = -I3
    Ji[0:3, 3:6] = -RiT * dT
    Ji[0:3, 6:9] = E.hat([a[0], a[1], a[2]])
    Ji[3:6, 3:6] = -RiT
    Ji[3:6, 6:9] = E.hat([c[0], c[1], c[2]])
    Ji[6:9, 6:9] = -JrInv * RjT * Ri
    Jj = Z9.copy()
    Jj[0:3, 0:3] = RiT * Rj
    Jj[3:6, 3:6] = RiT
    Jj[6:9, 6:9] = JrInv
    ExpT = E.quat_to_R(E.quat_normalized(E.quat_conj(E.so3_exp(rPhi))))      # SO3::exp(rPhi).inverse().matrix()
    v = JRg * E.col(nb["dbg"])
    JrB = E.so3_Jr([v[0], v[1], v[2]])
    Jb = mp.zeros(9, 6)
    Jb[0:3, 0:3] = -pre["JPg"]; Jb[0:3, 3:6] = -pre["JPa"]
    Jb[3:6, 0:3] = -pre["JVg"]; Jb[3:6, 3:6] = -pre["JVa"]
    Jb[6:9, 0:3] = -JrInv * ExpT * JrB * JRg
    return Ji, Jj, Jb


# ---- the graph ---------------------------------------------------------------------------------------------------------------------------
class Graph:
    """State = per keyframe (P, V, q) [the PVR vertex] and (dbg, dba) [the bias vertex]; points; lines.  Hessian order (A.1): non-fixed
    pose-side vertices by id (PVR 2k, bias 2k + 1), then the landmarks (points, then lines)."""

    def __init__(self, w):
        self.w = w
        kf = w["kf"]
        self.K = len(kf["P"])
        self.nav = [dict(P=E.mpv(kf["P"][k]), V=E.mpv(kf["V"][k]), q=E.mpv(kf["q"][k]), bg=E.mpv(kf["bg"][k]), ba=E.mpv(kf["ba"][k]),
                         dbg=E.mpv(kf["dbg"][k]), dba=E.mpv(kf["dba"][k])) for k in range(self.K)]
        self.pts = [E.mpv(p) for p in w["points"]]
        self.lns = [E.mpv(l) for l in w["lines"]]
        c = w["cam"]
        self.cam = dict(fx=f2mp(c["fx"]), fy=f2mp(c["fy"]), cx=f2mp(c["cx"]), cy=f2mp(c["cy"]), Rbc=mat(c["Rbc"]), Pbc=E.mpv(c["Pbc"]))
        self.gw = E.mpv(w["gw"])
        im = w["imu"]
        self.imu = []
        for m in range(len(im["kf_i"])):
            p = im["preint"][m]
            pre = dict(dP=E.mpv(p[0:3]), dV=E.mpv(p[3:6]), dR=mat(p[6:15].reshape(3, 3)), JPg=mat(p[15:24].reshape(3, 3)), JPa=mat(p[24:33].reshape(3, 3)),
                       JVg=mat(p[33:42].reshape(3, 3)), JVa=mat(p[42:51].reshape(3, 3)), JRg=mat(p[51:60].reshape(3, 3)), dt=f2mp(p[141]))
            self.imu.append(dict(i=int(im["kf_i"][m]), j=int(im["kf_j"][m]), pre=pre, info_pvr=mat(im["info_pvr"][m].reshape(9, 9)), info_bias=mat(im["info_bias"][m].reshape(6, 6))))
        self.po = [(int(w["po_pt"][e]), int(w["po_kf"][e]), E.mpv(w["po_uv"][e]), f2mp(np.float32(w["po_w"][e]))) for e in range(len(w["po_pt"]))]      # invSigma2 is a float (SURVEY B-Q12)
        self.lo = [(int(w["lo_ln"][e]), int(w["lo_kf"][e]), E.mpv(w["lo_l"][e]), f2mp(np.float32(w["lo_w"][e]))) for e in range(len(w["lo_ln"]))]
        self.level_p = [0] * len(self.po)
        self.level_l = [0] * len(self.lo)
        self.huber = {k: f2mp(v) for k, v in w["huber"].items()}      # kinds 0 point, 1 line, 2 PVR, 3 bias; removed from 0 / 1 by the gate
        self.fixed_pvr = [bool(x) for x in kf["fixed_pvr"]]
        self.fixed_bias = [bool(x) for x in kf["fixed_bias"]]
        self.cache = {}      # per-edge error cache (A.7)
        self.layout()

    def layout(self):
        """Hessian indices of the active non-fixed vertices (A.1)."""
        off = 0
        self.ix_pvr, self.ix_bias = [-1] * self.K, [-1] * self.K
        for k in range(self.K):
            if not self.fixed_pvr[k]:
                self.ix_pvr[k] = off; off += 9
            if not self.fixed_bias[k]:
                self.ix_bias[k] = off; off += 6
        self.n_pose = off
        act_p = set(pt for e, (pt, _, _, _) in enumerate(self.po) if self.level_p[e] == 0)
        act_l = set(ln for e, (ln, _, _, _) in enumerate(self.lo) if self.level_l[e] == 0)
        self.ix_pt, self.ix_ln = [-1] * len(self.pts), [-1] * len(self.lns)
        for i in range(len(self.pts)):
            if i in act_p:
                self.ix_pt[i] = off; off += 3
        for i in range(len(self.lns)):
            if i in act_l:
                self.ix_ln[i] = off; off += 6
        self.n = off

    # -- g2o computeActiveErrors: refresh the cache of every ACTIVE edge
    def compute_errors(self):
        for e, (pt, kf, uv, w) in enumerate(self.po):
            if self.level_p[e] == 0:
                self.cache[("p", e)] = E.point_edge(self.cam, self.nav[kf], self.pts[pt], uv)[0]
        for e, (ln, kf, l, w) in enumerate(self.lo):
            if self.level_l[e] == 0:
                self.cache[("l", e)] = E.line_edge(self.cam, self.nav[kf], self.lns[ln], l)[0]
        for m, ed in enumerate(self.imu):
            self.cache[("v", m)] = E.pvr_error(self.gw, self.nav[ed["i"]], self.nav[ed["j"]], self.nav[ed["i"]], ed["pre"])
            self.cache[("b", m)] = E.bias_error(self.nav[ed["i"]], self.nav[ed["j"]])

    def edge_chi2(self, key):
        kind, e = key
        err = self.cache[key]
        if kind == "p":
            return self.po[e][3] * (err[0] * err[0] + err[1] * err[1])
        if kind == "l":
            return self.lo[e][3] * (err[0] * err[0] + err[1] * err[1])      # third component is identically zero (SURVEY B-Q2)
        info = self.imu[e]["info_pvr"] if kind == "v" else self.imu[e]["info_bias"]
        v = E.col(err)
        return (v.T * info * v)[0, 0]

    def robustify(self, kind, chi2):
        """(rho_0, rho_1) of RobustKernelHuber (A.8) or (chi2, 1) without a kernel."""
        d = self.huber.get(kind)
        if d is None or chi2 <= d * d:
            return chi2, mp.mpf(1)
        s = mp.sqrt(chi2)
        return 2 * s * d - d * d, d / s

    KIND = {"p": 0, "l": 1, "v": 2, "b": 3}

    def active_robust_chi2(self):
        return mp.fsum(self.robustify(self.KIND[key[0]], self.edge_chi2(key))[0] for key in self.active_keys())

    def active_keys(self):
        ks = [("v", m) for m in range(len(self.imu))] + [("b", m) for m in range(len(self.imu))]
        ks += [("p", e) for e in range(len(self.po)) if self.level_p[e] == 0]
        ks += [("l", e) for e in range(len(self.lo)) if self.level_l[e] == 0]
        return ks

    # -- buildSystem: linearizeOplus + constructQuadraticForm of every active edge (A.4)
    def build_system(self):
        n = self.n
        H, b = mp.zeros(n, n), mp.zeros(n, 1)

        def add(blocks, info, err, rho1):
            """blocks: [(hessian index or -1, Jacobian)], err as a column; Omega' = rho_1 Omega, b += J^T (-Omega' e), H += J^T Omega' J"""
            om = info * rho1
            g = -(om * err)
            for a, (ia, Ja) in enumerate(blocks):
                if ia < 0:
                    continue
                JtO = Ja.T * om
                bb = Ja.T * g
                for r in range(Ja.cols):
                    b[ia + r] += bb[r]
                for (ic, Jc) in blocks[a:]:
                    if ic < 0:
                        continue
                    Hb = JtO * Jc
                    for r in range(Hb.rows):
                        for c in range(Hb.cols):
                            H[ia + r, ic + c] += Hb[r, c]
                            if ic != ia:
                                H[ic + c, ia + r] += Hb[r, c]
        for m, ed in enumerate(self.imu):
            i, j = ed["i"], ed["j"]
            err = self.cache[("v", m)]
            Ji, Jj, Jb = pvr_jacobians(self.gw, self.nav[i], self.nav[j], self.nav[i], ed["pre"], err)
            _, r1 = self.robustify(2, self.edge_chi2(("v", m)))
            add([(self.ix_pvr[i], Ji), (self.ix_pvr[j], Jj), (self.ix_bias[i], Jb)], ed["info_pvr"], E.col(err), r1)
            _, r1 = self.robustify(3, self.edge_chi2(("b", m)))
            add([(self.ix_bias[i], -mp.eye(6)), (self.ix_bias[j], mp.eye(6))], ed["info_bias"], E.col(self.cache[("b", m)]), r1)      # IMU/g2otypes.cpp:264-284
        for e, (pt, kf, uv, wgt) in enumerate(self.po):
            if self.level_p[e]:
                continue
            _, Jl, JdP, JdR, _ = E.point_edge(self.cam, self.nav[kf], self.pts[pt], uv)
            Jp = mp.zeros(2, 9)
            Jp[:, 0:3] = JdP; Jp[:, 6:9] = JdR
            _, r1 = self.robustify(0, self.edge_chi2(("p", e)))
            add([(self.ix_pt[pt], Jl), (self.ix_pvr[kf], Jp)], mp.eye(2) * wgt, E.col(self.cache[("p", e)]), r1)
        for e, (ln, kf, l, wgt) in enumerate(self.lo):
            if self.level_l[e]:
                continue
            _, Jls, Jle, JPs, JPe, JRs, JRe, _ = E.line_edge(self.cam, self.nav[kf], self.lns[ln], l)
            Jl = mp.zeros(2, 6)
            Jl[0, 0:3] = Jls; Jl[1, 3:6] = Jle
            Jp = mp.zeros(2, 9)
            Jp[0, 0:3] = JPs; Jp[1, 0:3] = JPe; Jp[0, 6:9] = JRs; Jp[1, 6:9] = JRe
            _, r1 = self.robustify(1, self.edge_chi2(("l", e)))
            add([(self.ix_ln[ln], Jl), (self.ix_pvr[kf], Jp)], mp.eye(2) * wgt, E.col(self.cache[("l", e)]), r1)
        return H, b

    # -- push / pop / update
    def snapshot(self):
        return ([dict((k, list(v)) for k, v in n.items()) for n in self.nav], [list(p) for p in self.pts], [list(l) for l in self.lns])

    def restore(self, s):
        self.nav = [dict((k, list(v)) for k, v in n.items()) for n in s[0]]
        self.pts = [list(p) for p in s[1]]
        self.lns = [list(l) for l in s[2]]

    def update(self, x):
        for k in range(self.K):
            if self.ix_pvr[k] >= 0:
                u = [x[self.ix_pvr[k] + t] for t in range(9)]
                o = E.oplus_pvr(self.nav[k], u)      # IMU/NavState.cpp:69-98
                self.nav[k]["P"], self.nav[k]["V"], self.nav[k]["q"] = o[0:3], o[3:6], o[6:10]
            if self.ix_bias[k] >= 0:      # IMU/NavState.cpp:100-121: the DELTA bias moves
                for t in range(3):
                    self.nav[k]["dbg"][t] += x[self.ix_bias[k] + t]
                    self.nav[k]["dba"][t] += x[self.ix_bias[k] + 3 + t]
        for i, ix in enumerate(self.ix_pt):
            if ix >= 0:
                for t in range(3):
                    self.pts[i][t] += x[ix + t]
        for i, ix in enumerate(self.ix_ln):
            if ix >= 0:
                for t in range(6):
                    self.lns[i][t] += x[ix + t]


def optimize(g, iters, rows, stage, user_lambda_init=0.0):
    """SparseOptimizer::optimize(n) with OptimizationAlgorithmLevenberg::solve."""
    g.layout()
    lam, nu = None, mp.mpf(2)
    for it in range(iters):
        g.compute_errors()
        cur = g.active_robust_chi2()
        H, b = g.build_system()
        if it == 0:
            lam = f2mp(user_lambda_init) if user_lambda_init > 0 else TAU * max(abs(H[j, j]) for j in range(g.n))      # g2o computeLambdaInit
            nu = mp.mpf(2)
        qmax, rho = 0, mp.mpf(0)
        while True:
            saved = g.snapshot()
            A = H.copy()
            for j in range(g.n):
                A[j, j] += lam
            try:
                x = mp.cholesky_solve(A, b)
                ok = True
            except (ValueError, ZeroDivisionError):
                x, ok = mp.zeros(g.n, 1), False
            g.update([x[j] for j in range(g.n)])
            g.compute_errors()
            tmp = g.active_robust_chi2()
            scale = mp.fsum(x[j] * (lam * x[j] + b[j]) for j in range(g.n)) + mp.mpf("1e-3")
            rho = (cur - tmp) / scale if ok else mp.mpf(-1)
            row = dict(stage=stage, iteration=it, trial=qmax, solver_ok=int(ok), **{"lambda": float(lam)}, chi2_current=float(cur), chi2_trial=float(tmp), scale=float(scale), rho=float(rho))
            if ok and rho > 0:
                alpha = 1 - (2 * rho - 1) ** 3
                alpha = min(alpha, GOOD_HI)
                lam *= max(GOOD_LO, alpha)
                nu = mp.mpf(2)
                cur = tmp
                row["accepted"] = 1
            else:
                lam *= nu
                nu *= 2
                g.restore(saved)
                row["accepted"] = 0
            rows.append(row)
            qmax += 1
            if not (rho < 0 and qmax < MAX_TRIALS):
                break
        if qmax == MAX_TRIALS or rho == 0:
            return it + 1, cur
    return iters, cur


def gate(g):
    """src/mapHandler.cpp:6047-6066: chi2() of the CACHED error (A.7) above 5.991 or a non-positive depth at the current estimate
    puts a point / line edge at level 1; the robust kernel comes off every point / line edge."""
    n_p = n_l = 0
    for e, (pt, kf, uv, w) in enumerate(g.po):
        dpos = E.point_edge(g.cam, g.nav[kf], g.pts[pt], uv)[4]
        if g.edge_chi2(("p", e)) > CHI2_GATE or not dpos:
            g.level_p[e] = 1; n_p += 1
    for e, (ln, kf, l, w) in enumerate(g.lo):
        dpos = E.line_edge(g.cam, g.nav[kf], g.lns[ln], l)[7]
        if g.edge_chi2(("l", e)) > CHI2_GATE or not dpos:
            g.level_l[e] = 1; n_l += 1
    del g.huber[0], g.huber[1]
    return n_p, n_l


def make_case_window(pkg, lm_sigma=None):
    w = pkg.window.make_window(CASE["K"], CASE["Np"], CASE["Nl"], imu=True, seed=CASE["seed"])
    return w


def run(w, stage1=5, stage2=10, user_lambda_init=0.0):
    g = Graph(w)
    rows = []
    it1, chi1 = optimize(g, stage1, rows, 1, user_lambda_init)
    n_p, n_l = gate(g)
    it2, chi2 = optimize(g, stage2, rows, 2, user_lambda_init)
    return dict(rows=rows, iterations=[it1, it2], chi2_final=[float(chi1), float(chi2)], gated=[n_p, n_l],
                P=[E.to_f(n["P"]) for n in g.nav], V=[E.to_f(n["V"]) for n in g.nav], q=[E.to_f(n["q"]) for n in g.nav],
                dbg=[E.to_f(n["dbg"]) for n in g.nav], dba=[E.to_f(n["dba"]) for n in g.nav],
                points=[E.to_f(p) for p in g.pts], lines=[E.to_f(l) for l in g.lns])


def perturbed(w, scale):
    """the same window with its landmarks moved `scale` times further from the truth: stage 1 then starts with rejected trials"""
    w2 = dict(w)
    w2["points"] = w["truth"]["points"] + (w["points"] - w["truth"]["points"]) * scale
    w2["lines"] = w["truth"]["lines"] + (w["lines"] - w["truth"]["lines"]) * scale
    return w2


OVERSHOOT = [("overshoot_a", dict(seed=0x601D31), 100.0), ("overshoot_b", dict(seed=77), 1.0)]      # (name, overshoot_cases.overshoot_window arguments, userLambdaInit)


def overshoot_window(pkg, spec):
    import overshoot_cases as oc
    return oc.overshoot_window(pkg, K=CASE["K"], Np=CASE["Np"], Nl=CASE["Nl"], **spec)


def main():
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    import __graft_entry__ as ge
    pkg = ge.load_package()
    w = make_case_window(pkg)
    out = {"note": "mpmath 40 digits, dense normal equations; see make_lm_trace.py", "case": CASE, "meta": dict(Ep=int(w["meta"]["Ep"]), El=int(w["meta"]["El"]))}

    def report(name):
        r = out[name]
        print(name + ":", r["iterations"], r["gated"], r["chi2_final"], "rejected trials by stage", [sum(1 - t["accepted"] for t in r["rows"] if t["stage"] == s) for s in (1, 2)])
    out["nominal"] = run(w)
    report("nominal")
    out["perturbed_scale"] = 12.0
    out["perturbed"] = run(perturbed(w, out["perturbed_scale"]))
    report("perturbed")
    # rotations off by 0.6 rad and velocities by 20 m/s, started at a damping that is too small: damped steps overshoot with the Huber
    # kernels still on (stage 1) and after the gate (stage 2)
    for name, spec, lam in OVERSHOOT:
        out[name] = dict(run(overshoot_window(pkg, spec), user_lambda_init=lam), user_lambda_init=lam)
        report(name)
    with open(os.path.join(HERE, "lm_trace.json"), "w") as f:
        json.dump(out, f)
    print("wrote lm_trace.json")


if __name__ == "__main__":
    main()
