"""Host-side call-site protocol of the reference's local mapping thread, over the C ABI.

Mirrors MapHandler::localBundleAdjustmentWithImuAndMarg (src/mapHandler.cpp:5741-6254):
graph upload, optimize(5) with Huber on every edge, chi2/depth gating, optimize(10) without Huber
on the point/line edges (Huber stays on the IMU edges, SURVEY B-Q11), optional marginalization of
the oldest keyframe when the window is full, write-back.  Works with any implementation of
plba.h (abi.Problem); the product path passes the HIP library.
"""
from . import abi
from .window import CHI2_GATE

STAGE1_ITERS = 5      # mapHandler.cpp:6039
STAGE2_ITERS = 10     # mapHandler.cpp:6069
MARG_NUM = 50         # mapHandler.cpp:6073
MAX_KF_IN_WINDOW = 12  # include/mapHandler.h:217


def local_ba(prob, abort=None, stage1=STAGE1_ITERS, stage2=STAGE2_ITERS, chi2_gate=CHI2_GATE,
             marginalize=False, first_kf=0, marg_num=MARG_NUM):
    """Run the two-stage local BA on an uploaded problem.  Returns a dict of per-stage stats, the
    gating counts and (if requested) the new prior."""
    out = {}
    out["stage1"] = prob.optimize(stage1, abort)
    do_more = not (abort is not None and abort[0])                       # mapHandler.cpp:6043-6046
    if do_more:
        out["gated"] = prob.gate_outliers(chi2_gate)                     # :6047-6066
        out["stage2"] = prob.optimize(stage2, abort)                     # :6068-6069
    if marginalize:                                                      # :6075-6199
        out["prior"] = prob.marginalize(first_kf, marg_num)
    return out


def results(prob):
    """Write-back payload of mapHandler.cpp:6202-6239."""
    r = prob.get_keyframes()
    r["points"] = prob.get_points()
    r["lines"] = prob.get_lines()
    return r


def next_window_prior_ok(prior, vid_pvr, vid_bias):
    """The prior edge attaches to vertices by id (mapHandler.cpp:6020-6027); every kept vertex
    must exist in the next window."""
    ids = set(int(v) for v in vid_pvr) | set(int(v) for v in vid_bias if v >= 0)
    return all(int(v) in ids for v in prior["vid"])
