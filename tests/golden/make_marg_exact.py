"""Extended-precision golden values for the marginalization step (IMU/marginalization.cpp:330-372), written to
tests/golden/marg_exact.npz.  Run on a GPU box:  python tests/golden/make_marg_exact.py gpurun_out/marg_exact.npz

Why: the reference thresholds the eigenvalues of the whole dropped block Amm at 1e-8 while |Amm| ~ 1e7, so the outcome hangs on
eigenvalues 15 orders below the norm.  Forming Amm = Jm^T Jm in fp64 already perturbs those by ~macheps |Amm|: the fp64 oracle
(which, like the reference, forms Amm and diagonalises it) is itself only good to 1e-4 on windows with far landmarks, so it cannot
arbitrate there.  These values can: the stacked Jacobian J and residual r of the step (fetched from the device with
options.diag bit 1; the same J, r the oracle's factors hold — A' agrees with the oracle to 1e-11 on the near-landmark cases) are
taken as exact inputs, and A = J^T J, the eigen-decomposition of Amm, the thresholded pseudo-inverse, the Schur complement A', b'
and r0^T r0 = b'^T A'^+ b' are evaluated with mpmath at 40 digits.
Cases: the far-landmark windows of tests/test_gpu_parity.py::_far_window (far = 1, 1e2, 1e3, 1e6) and a 12-keyframe window with
tracks over the whole window (kept block n = 105), all marginalized at their initial estimates (no optimisation in between, so
the inputs depend on the window generator alone)."""
import os, sys
import numpy as np
import mpmath as mp
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package()
mp.mp.dps = 40

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from marg_cases import CASES, case_window


def exact(J, r, m, n, eps):
    R, pos = J.shape
    Jm = mp.matrix(J.tolist()); rm = mp.matrix(r.tolist())
    A = Jm.T * Jm; b = Jm.T * rm
    E, Q = mp.eigsy(A[0:m, 0:m])
    W = mp.matrix(m, m)
    for i in range(m):
        if E[i] > eps:
            W[i, i] = 1 / E[i]
    Ainv = Q * W * Q.T
    Ar = A[m:pos, m:pos] - A[m:pos, 0:m] * Ainv * A[0:m, m:pos]
    br = b[m:pos, 0] - A[m:pos, 0:m] * Ainv * b[0:m, 0]
    E2, Q2 = mp.eigsy(Ar)
    r0r0 = mp.mpf(0)
    for i in range(n):
        if E2[i] > eps:
            vb = sum(Q2[k, i] * br[k] for k in range(n))
            r0r0 += vb * vb / E2[i]
    return (np.array(Ar.tolist(), dtype=float), np.array(br.tolist(), dtype=float).ravel(), float(r0r0),
            np.array([float(E[i]) for i in range(m)]), np.array([float(E2[i]) for i in range(n)]))


if __name__ == "__main__":
    out = {}
    for name, spec in CASES:
        w = case_window(pkg, spec)
        g = pkg.new_problem(marg_exact=0, diag=2); g.upload_window(w); pr = g.marginalize(0, 50)
        d = g.debug_get("marg_J"); g.close()
        R, pos, m, n = (int(x) for x in d[:4])
        J = d[4:4 + R * pos].reshape(pos, R).T; r = d[4 + R * pos:]
        Ar, br, r0r0, lam_mm, lam_r = exact(J, r, m, n, 1e-8)
        out[name + "_Ar"] = Ar; out[name + "_br"] = br; out[name + "_r0r0"] = np.array([r0r0]); out[name + "_dims"] = np.array([m, n, R])
        out[name + "_lam_mm"] = lam_mm; out[name + "_lam_r"] = lam_r; out[name + "_vid"] = np.asarray(pr["vid"])
        print(name, "m", m, "n", n, "R", R, "eigenvalues of Amm within a decade of the threshold:", np.sort(lam_mm[(lam_mm > 1e-9) & (lam_mm < 1e-7)]), flush=True)
    np.savez_compressed(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "marg_exact.npz"), **out)
