"""cycle stamps of three chunk workgroups of k_schur_pairs (diagnostic build: PLBA_EXTRA_FLAGS=-DPLBA_STAMPS_LM; the hll stamps
of the same build write to the same slots afterwards, so this tool stops after the first Schur launch: debug_build)"""
import sys, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as ge
pkg = ge.load_package()
w = pkg.window.make_config(3)
g = pkg.new_problem(); g.upload_window(w)
g.debug_build(100.0, True)
v = g.debug_get("dbgbuf")
t0 = min(v[5], v[13], v[21])
for name, o in (("first chunk", 0), ("middle chunk", 8), ("last chunk", 16)):
    print("%-13s loads done %6.0f  reduced %6.0f  partials published + counted %6.0f  end %6.0f cycles | start %+5.0f end %+5.0f (100 MHz ticks)" % (name, v[o + 1], v[o + 2], v[o + 3], v[o + 4], v[o + 5] - t0, v[o + 6] - t0))
