#!/usr/bin/env python
"""Regenerates the golden vectors under tests/golden/ from the CPU oracle.

The reference ships no fixtures for this path and cannot be built or run here, so these
vectors are produced by this repository's own oracle (oracle/plba_oracle.c) on deterministic windows
(window.make_window, seeds below) — they pin the oracle against regressions and give the GPU tests a
second, frozen, comparison point.  Inputs are regenerated from the seed; only expected outputs are stored.

    python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402
from oracle import oracle as orc  # noqa: E402

CASES = {
    "imu_small": dict(K=6, Np=60, Nl=15, imu=True, seed=0x601D01),
    "noimu_small": dict(K=5, Np=50, Nl=12, imu=False, seed=0x601D02),
}


def run_case(pkg, c):
    w = pkg.window.make_window(c["K"], c["Np"], c["Nl"], imu=c["imu"], seed=c["seed"])
    p = orc.new_problem()
    p.upload_window(w)
    p.debug_build(2.5, True)
    build = dict(chi2=float(p.debug_get("chi2")[0]), maxdiag=float(p.debug_get("maxdiag")[0]),
                 x=p.debug_get("x").tolist(), bp=p.debug_get("bp").tolist())
    p.close()
    p = orc.new_problem()
    p.upload_window(w)
    out = pkg.protocol.local_ba(p)
    trace1 = None
    res = pkg.protocol.results(p)
    tr = p.trace()
    g = dict(meta=dict(c, Ep=int(w["meta"]["Ep"]), El=int(w["meta"]["El"])), build=build,
             gated=list(out["gated"]), stage2_trace=tr,
             chi2=[out["stage1"].chi2_initial, out["stage1"].chi2_final, out["stage2"].chi2_initial, out["stage2"].chi2_final],
             P=res["P"].tolist(), V=res["V"].tolist(), q=res["q"].tolist(), dbg=res["dbg"].tolist(), dba=res["dba"].tolist(),
             points_checksum=float(np.abs(res["points"]).sum()), lines_checksum=float(np.abs(res["lines"]).sum()),
             points_head=res["points"][:5].tolist())
    if c["imu"]:
        pr = p.marginalize(0, 50)
        g["marg"] = dict(n=int(pr["n"]), m=int(pr["m"]), vid=pr["vid"].tolist(), idx=pr["idx"].tolist(),
                         Ar_diag=np.diag(pr["Ar"]).tolist(), br=pr["br"].tolist(), r0_sq=float(pr["r0"] @ pr["r0"]))
    p.close()
    return g


def preint_stream(seed, M=4):
    """EuRoC-shaped IMU stream for the preintegration producer (ns stamps near 1.4e9 s: long double territory), with
    samples before the first image and past the second one so every branch of src/keyFrame.cpp:147-170 runs."""
    LD = np.longdouble
    rng = np.random.default_rng(seed)
    t_prev = LD("1403636579.763555527") + LD(0.25) * np.arange(M, dtype=LD)
    t_curr = t_prev + LD(0.25)
    ts, starts = [], [0]
    for m in range(M):
        k = np.arange(-2, 50 + 2, dtype=LD)
        tt = t_prev[m] + LD(0.0007) + k / LD(200.0) + LD(1e-6) * rng.normal(size=len(k)).astype(LD)
        ts.append(np.sort(tt)); starts.append(starts[-1] + len(tt))
    t = np.concatenate(ts)
    S = len(t)
    return dict(sample_start=np.array(starts, dtype=np.int32), t=t, gyr=rng.normal(size=(S, 3)) * 0.3,
                acc=rng.normal(size=(S, 3)) * 2.0 + np.array([0, 0, 9.81]), t_prev=t_prev, t_curr=t_curr,
                bg=rng.normal(size=(M, 3)) * 1e-3, ba=rng.normal(size=(M, 3)) * 1e-2)


def run_preint(pkg):
    s = preint_stream(0x601D03)
    p = orc.new_problem()
    out = p.preintegrate(s["sample_start"], s["t"], s["gyr"], s["acc"], s["t_prev"], s["t_curr"], s["bg"], s["ba"],
                         pkg.window.GYR_MEAS_COV, pkg.window.ACC_MEAS_COV)
    p.close()
    return dict(meta=dict(seed=0x601D03, M=4), dP=out[:, 0:3].tolist(), dV=out[:, 3:6].tolist(), dR=out[:, 6:15].tolist(),
                JRg=out[:, 51:60].tolist(), cov_diag=[np.diag(o[60:141].reshape(9, 9)).tolist() for o in out], dt=out[:, 141].tolist(),
                checksum=float(np.abs(out).sum()))


LBA_CASES = {"lba": dict(K=6, Np=80, Nl=16, n_fixed=2, seed=0x601D04, variant=0, max_iters=15),
             "gba": dict(K=6, Np=80, Nl=16, n_fixed=1, seed=0x601D05, variant=1, max_iters=6)}


def lba_opts(c):
    eps = float(np.finfo(float).eps)
    return dict(variant=1, min_error=eps, min_error_change=eps, max_iters=c["max_iters"]) if c["variant"] else dict(max_iters=c["max_iters"])


def run_lba(pkg):
    """the pre-init visual-only optimiser (MapHandler::levMarquardtOptimizationLBA / ...GBA) on make_visual_window"""
    out = {}
    for name, c in LBA_CASES.items():
        w = pkg.window.make_visual_window(K=c["K"], Np=c["Np"], Nl=c["Nl"], n_fixed=c["n_fixed"], seed=c["seed"])
        p = orc.new_problem()
        r = p.lba_visual(w["T_kf_w"], w["kf_loc"], w["xyz"], w["pq"], w["po_pt"], w["po_kf"], w["uv"], w["lo_ln"], w["lo_kf"], w["l3"], w["cam"], **lba_opts(c))
        p.close()
        out[name] = dict(meta=dict(c, Ep=int(len(w["po_pt"])), El=int(len(w["lo_ln"]))), iterations=int(r["iterations"]), updates=int(r["updates"]),
                         err_first=float(r["err_first"]), err_last=(None if not np.isfinite(r["err_last"]) else float(r["err_last"])), lam=float(r["lam"]),
                         T=r["T"].tolist(), xyz_head=r["xyz"][:5].tolist(), xyz_checksum=float(np.abs(r["xyz"]).sum()), pq_checksum=float(np.abs(r["pq"]).sum()),
                         pt_moved=int(r["pt_moved"].sum()), ln_moved=int(r["ln_moved"].sum()))
    return out


def main():
    pkg = ge.load_package()
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "lba_small.json"), "w") as f:
        json.dump(run_lba(pkg), f, indent=0)
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "preint_small.json"), "w") as f:
        json.dump(run_preint(pkg), f, indent=0)
    for name, c in CASES.items():
        g = run_case(pkg, c)
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), name + ".json"), "w") as f:
            json.dump(g, f, indent=0)
        print("wrote", name, "stage2 chi2", g["chi2"][3])


if __name__ == "__main__":
    main()
