import sys, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as ge
pkg = ge.load_package()
w = pkg.window.make_config(3)
g = pkg.new_problem(); g.upload_window(w)
g.debug_build(100.0, True)
g.debug_build(100.0, True)
st = g.debug_get("stamps")
names = {16: "step 5 start", 17: "X + update done", 18: "barrier, pipeline start", 0: "w0 sweep start", 1: "w0 sweep end", 4: "w1 start", 5: "w1 I11 done", 6: "w1 M out", 7: "w1 end",
         8: "w2 start", 10: "w2 all rs out", 11: "w2 end", 21: "w2 next tiles fetched", 12: "w3 subst done", 13: "w3 Wu done", 14: "w3 end", 19: "after final barrier", 20: "step 6 start"}
for i in sorted(names, key=lambda i: st[i]): print("%-26s %8.0f" % (names[i], st[i]))
