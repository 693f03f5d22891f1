"""Who is right where the device's dense pseudo-inverse and the oracle's differ (far landmarks, eigenvalues of Amm around the
1e-8 threshold)?  The stacked Jacobian J and residual r of the marginalization step are fetched from the device (options.diag bit 1),
A = J^T J, the eigen-decomposition of Amm, the thresholded pseudo-inverse and the Schur complement are evaluated with mpmath at 40
digits, and both fp64 results are compared with that.  Also prints the eigenvalues of Amm next to the threshold."""
import os, sys
import numpy as np
import mpmath as mp
sys.path.insert(0, ".")
import __graft_entry__ as ge
pkg = ge.load_package()
from oracle import oracle as orc
from tools.check_marg import far_window
mp.mp.dps = 40


def exact(J, r, m, n, eps):
    R, pos = J.shape
    Jm = mp.matrix(J.tolist()); rm = mp.matrix(r.tolist())
    A = Jm.T * Jm; b = Jm.T * rm
    Amm = A[0:m, 0:m]
    E, Q = mp.eigsy(Amm)
    lam = [E[i] for i in range(m)]
    W = mp.matrix(m, m)
    for i in range(m):
        if lam[i] > eps:
            W[i, i] = 1 / lam[i]
    Ainv = Q * W * Q.T
    Ar = A[m:pos, m:pos] - A[m:pos, 0:m] * Ainv * A[0:m, m:pos]
    br = b[m:pos, 0] - A[m:pos, 0:m] * Ainv * b[0:m, 0]
    # eigen square root of the kept block (cpp:364-372): r0^T r0 = b'^T A'^+ b' over the eigenvalues above eps
    E2, Q2 = mp.eigsy(Ar)
    r0r0 = mp.mpf(0)
    for i in range(n):
        if E2[i] > eps:
            vb = sum(Q2[k, i] * br[k] for k in range(n))
            r0r0 += vb * vb / E2[i]
    lam2 = np.array([float(E2[i]) for i in range(n)])
    return np.array(Ar.tolist(), dtype=float), np.array(br.tolist(), dtype=float).ravel(), np.array([float(x) for x in lam]), float(r0r0), lam2


def compare(w, label, iters=0):
    o = orc.new_problem(); o.upload_window(w)
    if iters: o.optimize(iters)
    po = o.marginalize(0, 50); o.close()
    res = {}
    for mode in (0, 2):
        g = pkg.new_problem(marg_exact=mode, diag=2); g.upload_window(w)
        if iters: g.optimize(iters)
        res[mode] = g.marginalize(0, 50)
        if mode == 0:
            d = g.debug_get("marg_J")
        g.close()
    R, pos, m, n = (int(x) for x in d[:4])
    J = d[4:4 + R * pos].reshape(pos, R).T; r = d[4 + R * pos:]
    Ar, br, lam, r0r0, lam2 = exact(J, r, m, n, 1e-8)
    sc = np.abs(Ar).max()
    near = np.sort(lam[(lam > 1e-11) & (lam < 1e-5)])
    near2 = np.sort(lam2[(lam2 > 1e-11) & (lam2 < 1e-4)])
    print("%s  m %d n %d  eigenvalues of Amm in (1e-11, 1e-5): %s | of A' in (1e-11, 1e-4): %s" % (label, m, n, " ".join("%.3e" % x for x in near), " ".join("%.3e" % x for x in near2)))
    for name, pr in (("oracle (fp64, two-sided Jacobi on Amm)", po), ("device block-wise", res[0]), ("device dense (Hestenes on Jm)", res[2])):
        print("   %-40s |A' - exact| / max|A'| = %.2e   |b' - exact| = %.2e   r0^T r0 rel %.2e" % (
            name, np.abs(pr["Ar"] - Ar).max() / sc, np.abs(pr["br"] - br).max() / max(np.abs(br).max(), 1.0), abs(pr["r0"] @ pr["r0"] - r0r0) / r0r0), flush=True)


if __name__ == "__main__":
    fars = [float(x) for x in sys.argv[1:]] or [1e2, 1e3, 1e6]
    for far in fars:
        compare(far_window(far), "far %.0e" % far)
    compare(pkg.window.make_window(12, 260, 50, imu=True, seed=21), "K12 seed21", iters=3)
