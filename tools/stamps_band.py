import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
w = pkg.window.make_config(3, scale=0.2)
g = pkg.new_problem(); g.upload_window(w)
g.optimize(3)
d = g.debug_get("dbgbuf")
print("band", g.debug_get("band"))
print("cycles: lookahead %.0f panels %.0f write+barrier %.0f rhs+trailing %.0f barrier %.0f total %.0f" % (d[1]-d[0], d[2]-d[1], d[3]-d[2], d[4]-d[3], d[5]-d[4], d[5]-d[0]))
