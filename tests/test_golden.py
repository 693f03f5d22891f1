"""Golden vectors (tests/golden/*.json, made by tests/golden/make_golden.py from the CPU oracle):
the oracle must keep reproducing them (CPU), and the HIP path must match them (GPU)."""
import json
import os

import numpy as np
import pytest

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["imu_small", "noimu_small"]


def _load(name):
    with open(os.path.join(HERE, name + ".json")) as f:
        return json.load(f)


def _check(pkg, new_problem, g, tol):
    c = g["meta"]
    w = pkg.window.make_window(c["K"], c["Np"], c["Nl"], imu=c["imu"], seed=c["seed"])
    assert (w["meta"]["Ep"], w["meta"]["El"]) == (c["Ep"], c["El"])          # the generator itself is pinned
    p = new_problem(); p.upload_window(w)
    p.debug_build(2.5, True)
    assert p.debug_get("chi2")[0] == pytest.approx(g["build"]["chi2"], rel=tol)
    assert p.debug_get("maxdiag")[0] == pytest.approx(g["build"]["maxdiag"], rel=tol)
    xs = np.array(g["build"]["x"])
    assert np.abs(p.debug_get("x") - xs).max() < max(tol, 1e-9) * 10 * np.abs(xs).max()
    assert np.abs(p.debug_get("bp") - np.array(g["build"]["bp"])).max() < tol * np.abs(g["build"]["bp"]).max()
    p.close()
    p = new_problem(); p.upload_window(w)
    out = pkg.protocol.local_ba(p)
    assert list(out["gated"]) == g["gated"]
    tr = p.trace()
    assert [t["accepted"] for t in tr] == [t["accepted"] for t in g["stage2_trace"]]
    for a, b in zip(tr, g["stage2_trace"]):
        assert a["lam"] == pytest.approx(b["lam"], rel=1e3 * tol) and a["chi2_trial"] == pytest.approx(b["chi2_trial"], rel=1e3 * tol)
    res = pkg.protocol.results(p)
    ptol = max(1e3 * tol, 1e-9)
    for k in ("P", "V", "q", "dbg", "dba"):
        assert np.abs(res[k] - np.array(g[k])).max() < ptol, k
    assert np.abs(res["points"][:5] - np.array(g["points_head"])).max() < 10 * ptol
    assert np.abs(res["points"]).sum() == pytest.approx(g["points_checksum"], rel=ptol)
    if "marg" in g:
        pr = p.marginalize(0, 50)
        m = g["marg"]
        assert (pr["n"], pr["m"], pr["vid"].tolist(), pr["idx"].tolist()) == (m["n"], m["m"], m["vid"], m["idx"])
        assert np.abs(np.diag(pr["Ar"]) - np.array(m["Ar_diag"])).max() < 1e-6 * np.abs(m["Ar_diag"]).max()
        assert np.abs(pr["br"] - np.array(m["br"])).max() < 1e-6 * max(np.abs(m["br"]).max(), 1)
    p.close()


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_golden(pkg, orc, name):
    _check(pkg, orc.new_problem, _load(name), 1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_matches_golden(pkg, hip, name):
    _check(pkg, pkg.new_problem, _load(name), 1e-9)


def _check_preint(pkg, new_problem, tol):
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec); spec.loader.exec_module(mg)
    g = _load("preint_small")
    s = mg.preint_stream(g["meta"]["seed"], g["meta"]["M"])
    p = new_problem()
    out = p.preintegrate(s["sample_start"], s["t"], s["gyr"], s["acc"], s["t_prev"], s["t_curr"], s["bg"], s["ba"],
                         pkg.window.GYR_MEAS_COV, pkg.window.ACC_MEAS_COV)
    p.close()
    for key, sl in (("dP", slice(0, 3)), ("dV", slice(3, 6)), ("dR", slice(6, 15)), ("JRg", slice(51, 60))):
        ref = np.array(g[key])
        assert np.abs(out[:, sl] - ref).max() <= tol * max(1.0, np.abs(ref).max()), key
    for m in range(g["meta"]["M"]):
        ref = np.array(g["cov_diag"][m])
        assert np.abs(np.diag(out[m, 60:141].reshape(9, 9)) - ref).max() <= tol * np.abs(ref).max()
    assert np.allclose(out[:, 141], g["dt"], rtol=0, atol=1e-15)
    assert np.abs(out).sum() == pytest.approx(g["checksum"], rel=tol)


def test_oracle_reproduces_the_preintegration_golden(pkg, orc):
    _check_preint(pkg, orc.new_problem, 1e-13)


@pytest.mark.gpu
def test_hip_matches_the_preintegration_golden(pkg, hip):
    _check_preint(pkg, pkg.new_problem, 1e-11)


def _check_lba(pkg, new_problem, tol):
    from tests.golden.make_golden import LBA_CASES, lba_opts
    g = _load("lba_small")
    for name, c in LBA_CASES.items():
        e = g[name]
        assert {k: e["meta"][k] for k in c} == c
        w = pkg.window.make_visual_window(K=c["K"], Np=c["Np"], Nl=c["Nl"], n_fixed=c["n_fixed"], seed=c["seed"])
        assert (len(w["po_pt"]), len(w["lo_ln"])) == (e["meta"]["Ep"], e["meta"]["El"])          # the generator itself is pinned
        p = new_problem()
        r = p.lba_visual(w["T_kf_w"], w["kf_loc"], w["xyz"], w["pq"], w["po_pt"], w["po_kf"], w["uv"], w["lo_ln"], w["lo_kf"], w["l3"], w["cam"], **lba_opts(c))
        p.close()
        assert (r["iterations"], r["updates"]) == (e["iterations"], e["updates"])
        assert r["err_first"] == pytest.approx(e["err_first"], rel=tol) and r["lam"] == pytest.approx(e["lam"], rel=tol)
        assert (e["err_last"] is None and not np.isfinite(r["err_last"])) or r["err_last"] == pytest.approx(e["err_last"], rel=1e3 * tol)
        ptol = max(1e3 * tol, 1e-9)
        assert np.abs(r["T"] - np.array(e["T"])).max() < ptol
        assert np.abs(r["xyz"][:5] - np.array(e["xyz_head"])).max() < 10 * ptol
        assert np.abs(r["xyz"]).sum() == pytest.approx(e["xyz_checksum"], rel=ptol) and np.abs(r["pq"]).sum() == pytest.approx(e["pq_checksum"], rel=ptol)
        assert (int(r["pt_moved"].sum()), int(r["ln_moved"].sum())) == (e["pt_moved"], e["ln_moved"])


def test_oracle_reproduces_the_lba_golden(pkg, orc):
    _check_lba(pkg, orc.new_problem, 1e-12)


@pytest.mark.gpu
def test_hip_matches_the_lba_golden(pkg, hip):
    _check_lba(pkg, pkg.new_problem, 1e-9)
