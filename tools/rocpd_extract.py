"""Turn rocprofv3's rocpd sqlite output into the small CSV / JSON summaries kept under profiles/.

  python tools/rocpd_extract.py stats  gpurun_out/prof7/run_results.db  profiles/r01_v7_kernel_stats.csv
  python tools/rocpd_extract.py pmc    gpurun_out/pmc_fetch/run_results.db gpurun_out/pmc_write/run_results.db profiles/r01_pmc_traffic.json

`pmc` applies the gfx950 correction of MI355X_MICROARCH.md (HBM section): FETCH_SIZE (KB) counts 128-byte read
requests at 64 bytes, so it is doubled; WRITE_SIZE (KB) is taken as is.  Both counters sit on the memory side of L2
(Infinity-Cache hits included), one --pmc pass each."""
import csv, json, sqlite3, sys


def stats(db, out):
    c = sqlite3.connect(db).cursor()
    rows = list(c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc"))
    tot = sum(r[2] for r in rows)
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([r[0], r[1], r[2], "%.1f" % r[3], "%.2f" % (100.0 * r[2] / tot), r[4], r[5]])


def pmc(db_fetch, db_write, out, bench_json=None, command=None):
    res = {}
    for db, name in ((db_fetch, "FETCH_SIZE"), (db_write, "WRITE_SIZE")):
        c = sqlite3.connect(db).cursor()
        for k, n, avg in c.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name=? group by kernel_name", (name,)):
            res.setdefault(k, {})[name + "_KB_avg"] = avg
            res[k]["launches_" + name] = n
    for k, v in res.items():
        f, w = v.get("FETCH_SIZE_KB_avg"), v.get("WRITE_SIZE_KB_avg")
        if f is not None and w is not None:
            v["traffic_bytes_per_launch"] = (2.0 * f + w) * 1024.0
    import glob, hashlib, os
    h = hashlib.sha256()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for f in sorted(glob.glob(os.path.join(root, "pl-inertial-slam_amd", "csrc", "*.h*"))):
        h.update(open(f, "rb").read())
    # algorithmic bytes per launch (32 B per point observation, 40 B per line observation, 24 / 48 B per landmark) of the
    # kernels that pass over every observation once, from the workload the bench line names
    if bench_json:
        try:
            cfg = json.loads([l for l in open(bench_json).read().splitlines() if l.startswith("{")][-1])["config"]
            alg = 32 * cfg["point_obs"] + 40 * cfg["line_obs"] + 24 * cfg["points"] + 48 * cfg["lines"]
            for k, v in res.items():
                if any(t in k for t in ("k_lm_schur<0", "k_lm_trial", "k_linearize<true>", "k_linearize<false>")) and "traffic_bytes_per_launch" in v:
                    v["algorithmic_bytes_per_launch"] = alg
                    v["traffic_over_algorithmic"] = v["traffic_bytes_per_launch"] / alg
        except (OSError, ValueError, KeyError, IndexError):
            pass
    json.dump({"note": "per-launch averages; traffic = (2 x FETCH_SIZE + WRITE_SIZE) KB, gfx950 correction of MI355X_MICROARCH.md; "
                       "command: " + (command or "rocprofv3 --pmc <counter> -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-config5-leg") + " (one pass per counter)",
               "csrc_sha16": h.hexdigest()[:16],      # bench.py quotes these numbers only for the kernel sources they were measured on
               "kernels": res}, open(out, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5] if len(sys.argv) > 5 else None, sys.argv[6] if len(sys.argv) > 6 else None)
