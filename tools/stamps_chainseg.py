"""cycle stamps of the velocity / bias chain code riding in the fused launches (diagnostic build: PLBA_EXTRA_FLAGS=-DPLBA_STAMPS_LM):
chain_elim_segment (segment 1, in k_lm_schur) and chain_back_segment (segments 0 / 1, in k_lm_trial)"""
import sys
sys.path.insert(0, '.')
import __graft_entry__ as ge
pkg = ge.load_package()
w = pkg.window.make_config(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
g = pkg.new_problem(); g.upload_window(w)
g.optimize(4)
v = g.debug_get("dbgbuf")
n = int(v[75])
print("chain_elim_segment 1 (%d blocks): staged %.0f; chain wave after block: %s; column waves drained %.0f cycles" % (n, v[65], " ".join("%.0f" % v[66 + i] for i in range(n)), v[76]))
for s in (0, 1):
    o = 48 + 6 * s
    print("chain_back_segment %d: staged %.0f, W x done %.0f, substitution done %.0f, keyframes updated + drained %.0f cycles" % (s, v[o + 1], v[o + 2], v[o + 3], v[o + 4]))
print("IMU edge block 1, lane 0: error done %.0f, past the barrier %.0f, residual-dependent Jacobian blocks %.0f, chi / weights barrier %.0f, Omega J %.0f, J^T Omega J + atomics issued %.0f cycles" % tuple(v[25:31]))
