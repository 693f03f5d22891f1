import sqlite3, collections, sys
for db in sys.argv[1:]:
    c=sqlite3.connect(db).cursor()
    rows=list(c.execute("select name, start, end from kernels order by start"))
    names=collections.defaultdict(list)
    for n,s,e in rows: names[n.split('(')[0]].append(e-s)
    print(db)
    for n,v in sorted(names.items(), key=lambda x:-sum(x[1]))[:16]: print("  %-50s %5d %9.1f"%(n[:50],len(v),sum(v)/len(v)))
