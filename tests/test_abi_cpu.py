"""CPU-side checks of the drop-in boundary: the HIP shared object loads without a GPU, exports
exactly the entry points include/plba.h declares, and refuses to compute without a device
(no CPU fallback).  No kernel is launched here."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "plba.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = set(re.findall(r"\b(plba_[a-z0-9_]+)\s*\(", txt))
    names -= {"plba_allreduce_fn"}
    return names


def test_header_and_signature_table_agree(pkg):
    declared = _declared()
    table = {"plba_" + n for n in pkg.abi.SIGNATURES}
    assert declared == table, (declared ^ table)


def test_hip_library_exports_every_declared_symbol(pkg):
    import __graft_entry__ as g
    g.build_hip()
    lib = C.CDLL(g.HIP_LIB)
    for name in sorted(_declared()):
        assert hasattr(lib, name), name
    lib.plba_backend_name.restype = C.c_char_p
    assert lib.plba_backend_name() == b"hip-gfx950"


def test_oracle_exports_the_same_surface(pkg, orc):
    lib = orc.lib()
    assert set(lib.fn) == set(pkg.abi.SIGNATURES) - pkg.abi.PRODUCT_ONLY      # (memory management of the device-resident window is not the reference's algorithm)
    assert lib.backend_name() == "cpu-oracle"


def test_product_path_fails_loudly_without_gpu(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(pkg.abi.PlbaError, match="no HIP device|no CPU fallback"):
        pkg.new_problem()


def test_default_options_match_reference_constants(pkg):
    import __graft_entry__ as g
    g.build_hip()
    o = pkg.abi.Lib(g.HIP_LIB, "plba_").default_options()
    assert (o.tau, o.max_trials, o.marg_eps) == (1e-5, 10, 1e-8)          # g2o LM defaults; IMU/marginalization.h:99
    assert o.good_step_lower == pytest.approx(1 / 3) and o.good_step_upper == pytest.approx(2 / 3)
    assert o.fix_line_position_jacobian == 0 and not hasattr(o, "whiten_marg_factors")
    assert (o.lm_fused, o.lm_fused_min_obs, o.lm_group_steps, o.chain_seg, o.twin_max_tiles, o.diag) == (1, 40000, 0, 0, 0, 0)


def test_product_package_never_imports_the_oracle():
    """A product path that routes through the oracle voids every parity claim."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "pl-inertial-slam_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")) and "hostcheck" not in f:
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle", txt, flags=re.M), f
                assert not re.search(r"(CDLL|dlopen)\([^)]*oracle", txt), f


def test_rccl_hook_library_exports_its_header(pkg):
    """include/plba_rccl.h: the multi-GPU exchange hook in C++ (libplba_rccl.so).  It binds RCCL by dlopen at first use, so
    loading it needs neither a GPU nor librccl; the hook has exactly the plba_allreduce_fn signature plba_set_shard takes."""
    import __graft_entry__ as g
    g.build_rccl()
    txt = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "plba_rccl.h")).read(), flags=re.S)
    declared = set(re.findall(r"\b(plba_rccl_[a-z0-9_]+)\s*\(", txt))
    assert declared == {"plba_rccl_unique_id", "plba_rccl_init", "plba_rccl_allreduce", "plba_rccl_destroy", "plba_rccl_last_error"}
    lib = C.CDLL(g.RCCL_LIB)
    for name in sorted(declared):
        assert hasattr(lib, name), name
    # bad arguments are refused before RCCL is touched
    lib.plba_rccl_allreduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    assert lib.plba_rccl_allreduce(None, None, 4, 0, None) != 0
    lib.plba_rccl_last_error.restype = C.c_char_p
    assert b"bad argument" in lib.plba_rccl_last_error()
