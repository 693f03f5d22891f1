"""FETCH_SIZE calibration on known byte counts (tools/calib_fetch.hip): run under `rocprofv3 --pmc FETCH_SIZE -- python3 tools/calib_fetch.py`
(and once more with the request-size counters, see tools/collect_profiles_r05.sh); tools/rocpd_extract.py `calib` turns the run into
profiles/r05_fetch_calibration.json.  Source size 1 GiB: four times the Infinity Cache, so every element comes from HBM."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BYTES = 1 << 30
src = os.path.join(ROOT, "tools", "calib_fetch.hip")
out = os.path.join(ROOT, "tools", "_build_calib", "libcalib.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", src, "-o", out])
lib = ctypes.CDLL(out)
lib.calib_run.argtypes = [ctypes.c_size_t, ctypes.c_int]
rc = lib.calib_run(BYTES, 3)
print("calib_run rc=%d bytes=%d" % (rc, BYTES))
sys.exit(rc)
