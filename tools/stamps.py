import sys, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as ge
pkg = ge.load_package()
w = pkg.window.make_config(3)
g = pkg.new_problem(); g.upload_window(w)
g.debug_build(100.0, True)
g.debug_build(100.0, True)
st = g.debug_get("stamps")
names = ["start", "LT/x loaded+barrier", "trsm+XT write+barrier", "mfma+C rmw+barrier", "sC->regs", "potrf", "store_factor"]
print("cycle stamps (100 MHz s_memtime ticks? or shader clock):", st[:7])
for i in range(1, 7):
    print("%-28s %8.0f" % (names[i], st[i] - st[i - 1]))
