import sys, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as ge
pkg = ge.load_package()
w = pkg.window.make_config(3)
g = pkg.new_problem(); g.upload_window(w)
g.debug_build(100.0, True)
g.debug_build(100.0, True)
st = g.debug_get("stamps")[40:]
b = st[0]
names = {0: "w0 step 20 start", 1: "w0 C_ii ready", 2: "w0 potrf done", 3: "w0 inverse done", 4: "w0 published", 5: "w0 step 21 published", 8: "col step 20 start", 9: "col saw step 20", 10: "col step 20 done"}
for i in sorted(names, key=lambda i: st[i]): print("%-24s %8.0f" % (names[i], st[i] - b))
