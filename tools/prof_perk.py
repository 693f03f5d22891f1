"""per-block-step average duration of k_chol32 (position after each k_potrf0_32) from a rocprofv3 rocpd database"""
import sqlite3, collections, sys
for db in sys.argv[1:]:
    c = sqlite3.connect(db).cursor()
    rows = list(c.execute("select name, start, end from kernels order by start"))
    pos = None; agg = collections.defaultdict(list); gap = collections.defaultdict(list); prev_end = None
    for n, s, e in rows:
        if 'k_potrf0_32' in n: pos = 0; agg['p0'].append(e - s); prev_end = e; continue
        if 'k_chol32' in n and pos is not None:
            agg[pos].append(e - s); gap[pos].append(s - prev_end); prev_end = e; pos += 1
    print(db)
    print("  dur ", " ".join("%5.0f" % (sum(v) / len(v)) for k, v in agg.items()))
    print("  gap ", "      " + " ".join("%5.0f" % (sum(v) / len(v)) for k, v in gap.items()))
