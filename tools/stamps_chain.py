import sys, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as ge
pkg = ge.load_package()
w = pkg.window.make_config(3)
g = pkg.new_problem(chain_elim=1); g.upload_window(w)
g.debug_build(100.0, True)
g.debug_build(100.0, True)
full = g.debug_get("stamps")
st = full[40:]
b = st[0]
names = {0: "w0 step 20 start", 1: "w0 C_ii ready", 2: "w0 potrf done", 3: "w0 inverse done", 4: "w0 published", 5: "w0 step 21 published", 8: "col step 20 start", 9: "col saw step 20", 10: "col step 20 done"}
for i in sorted(names, key=lambda i: st[i]): print("%-24s %8.0f" % (names[i], st[i] - b))
print("wave 0 steps 20..23 (start, published):", [(full[60 + 2 * q] - b, full[61 + 2 * q] - b) for q in range(4)])
print("column waves done with step 20:", [full[70 + wv] - b for wv in range(1, 6)])
