"""cycle stamps of three workgroups of k_landmark_hll<true> (diagnostic build: PLBA_EXTRA_FLAGS=-DPLBA_STAMPS_LM)"""
import sys, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as ge
pkg = ge.load_package()
w = pkg.window.make_config(3)
g = pkg.new_problem(); g.upload_window(w)
g.optimize(4)
v = g.debug_get("dbgbuf")
for name, o in (("first block", 0), ("middle block", 8), ("last block", 16)):
    print("%-13s after barrier %6.0f  after edge loop %6.0f  before stores %6.0f  stores drained %6.0f cycles;  edges of lane 0's landmark %d; realtime %.0f" % (name, v[o + 1], v[o + 2], v[o + 3], v[o + 4], v[o + 6], v[o + 5]))
print("realtime spread first..last block end (100 MHz ticks):", v[16 + 5] - v[5], v[8 + 5] - v[5])

print("k_linearize<true>: IMU block 0 / mid: %.0f / %.0f cycles; observation blocks first / mid / last: %.0f / %.0f / %.0f cycles" % (v[32], v[33], v[40], v[42], v[44]))
t = [v[34], v[35], v[41], v[43], v[45]]
print("  end times relative to the earliest (100 MHz ticks): IMU0 %.0f IMUmid %.0f obs first %.0f mid %.0f last %.0f" % tuple(x - min(t) for x in t))

for g in (0, 1):
    o = 48 + 6 * g
    print("chain back segment %d: staged %.0f, W x done %.0f, substitution done %.0f, keyframes updated + drained %.0f cycles" % (g, v[o + 1], v[o + 2], v[o + 3], v[o + 4]))
print("  end (100 MHz ticks, relative to segment 0): segment 1 %+.0f, first landmark block %+.0f, last landmark block %+.0f" % (v[59] - v[53], v[60] - v[53], v[61] - v[53]))

print("IMU edge block 1, lane 0: error done %.0f, past the barrier %.0f, residual-dependent Jacobian blocks %.0f, chi / weights barrier %.0f, Omega J %.0f, J^T Omega J + atomics issued %.0f cycles" % tuple(v[25:31]))
