import sys, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as ge
pkg = ge.load_package()
w = pkg.window.make_config(3)
g = pkg.new_problem(); g.upload_window(w)
g.debug_build(100.0, True)
g.debug_build(100.0, True)
st = g.debug_get("stamps")
st = g.debug_get("stamps")
print("per wave: [start, panel products+barrier, update+barrier, look-ahead end] (cycles)")
for w in range(4):
    print("wave", w, st[4 * w:4 * w + 4])
names = {0: "w0 start", 1: "w0 potrf end", 4: "w1 start", 5: "w1 I11 done", 6: "w1 M published", 7: "w1 end", 8: "w2 start", 10: "w2 all rs out", 11: "w2 end", 12: "w3 subst done", 13: "w3 Wu done", 14: "w3 end"}
for i in sorted(names): print("%-18s %8.0f" % (names[i], st[16 + i]))
