"""Wall time of plba_preintegrate (upload + kernel + download) for a window's worth of intervals."""
import sys, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import __graft_entry__ as ge
pkg = ge.load_package()
from test_preintegration import _stream, _call
rng = np.random.default_rng(0)
for M in (49, 199):
    s = _stream(pkg, M, rng)
    g = pkg.new_problem()
    _call(g, pkg, s)
    t = time.perf_counter()
    for _ in range(20): _call(g, pkg, s)
    print("M=%d intervals x ~52 samples: %.3f ms per call (host staging included)" % (M, (time.perf_counter() - t) / 20 * 1e3))
    g.close()
