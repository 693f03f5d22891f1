"""The 32 x 32 tile factorisation of the reduced-camera solve (csrc/plba_factor32_dev.h: two 16-column DPP sweeps of one wave, MFMA coupling
blocks) on its own: tools/test_factor32.hip factors a graded SPD tile (entries over four decades) with it and with the look-ahead
pipeline it replaced (csrc/plba_dense_dev.h, still used by the dataflow and 64-column forms) and prints the residuals of both.
Replaces the diagonal-tile step of g2o's LinearSolverEigen."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tools", "test_factor32.hip")
EXE = os.path.join(ROOT, "tools", "_build_test_factor32")


def _build():
    hipcc = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else shutil.which("hipcc")
    assert hipcc, "hipcc not found"
    deps = [SRC] + [os.path.join(ROOT, "pl-inertial-slam_amd", "csrc", f) for f in ("plba_factor32_dev.h", "plba_dense_dev.h", "plba_internal.h")]
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-I", os.path.join(ROOT, "include"),
                               "-I", os.path.join(ROOT, "pl-inertial-slam_amd", "csrc"), SRC, "-o", EXE])
    return EXE


def test_tile_tool_builds():
    """cross-compiles for gfx950 without a GPU (the sweep's DPP / permlane builtins and the MFMA products)"""
    assert os.access(_build(), os.X_OK)


@pytest.mark.gpu
def test_tile_factor_and_inverse():
    out = subprocess.run([_build()], capture_output=True, text=True, timeout=120).stdout
    rows = re.findall(r"mode (\d) \((\w+)\): err ([\w ]+), solver_ok (\d), \|L L\^T - A\| \(scaled\) (\S+), \|Linv L - I\| (\S+), \|Linv - kept\| (\S+), upper part (\S+), cycles per tile (\d+)", out)
    assert len(rows) == 2, out
    for mode, name, err, ok, e1, e2, e3, up, cyc in rows:
        assert err == "no error" and ok == "1", (name, err, ok)
        assert float(e1) < 1e-14 and float(e2) < 1e-13, (name, e1, e2)      # L L^T = A entrywise against sqrt(a_ii a_jj); L^-1 L = I
        assert float(e3) == 0.0 and float(up) == 0.0, (name, e3, up)        # the copy kept in LDS is the published one; strictly lower triangular outputs
    new, old = int(rows[1][8]), int(rows[0][8])
    assert new < old, (new, old)      # the point of the exercise (10.7 k against 13.3 k cycles when written)
