"""cycle stamps of three workgroups of k_lm_schur<0> (diagnostic build: PLBA_EXTRA_FLAGS=-DPLBA_STAMPS_LMF)"""
import sys, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as ge
pkg = ge.load_package()
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
w = pkg.window.make_config(cfg)
g = pkg.new_problem(); g.upload_window(w)
g.optimize(4)
print("fused:", g.debug_get("lm_fused"))
v = g.debug_get("dbgbuf")
names = ["start", "staged+first loads", "eval done", "hll reduced", "operand written", "mfma done", None, "step 0 end", "loop end", "waves combined", "outputs drained"]
for name, o in (("first group", 0), ("middle group", 16), ("last group", 32)):
    print(name, "steps", int(v[o + 11]))
    prev = 0.0
    for q in range(1, 11):
        if names[q] is None: continue
        print("   %-28s %8.0f  (+%.0f)" % (names[q], v[o + q], v[o + q] - prev)); prev = v[o + q]
h = g.debug_get("lm_groups")
print("groups by steps, points:", [int(x) for x in h[:8]], "lines:", [int(x) for x in h[8:16]], "mean window %.2f" % (h[16] / max(sum(h[:16]), 1)), "workgroup steps", int(h[17]))
