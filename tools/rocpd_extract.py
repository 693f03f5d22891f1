"""Turn rocprofv3's rocpd sqlite output into the small CSV / JSON summaries kept under profiles/.

  python tools/rocpd_extract.py stats  gpurun_out/prof7/run_results.db  profiles/r01_v7_kernel_stats.csv
  python tools/rocpd_extract.py pmc    gpurun_out/pmc_fetch/run_results.db gpurun_out/pmc_write/run_results.db profiles/r01_pmc_traffic.json

`pmc` applies the gfx950 correction of MI355X_MICROARCH.md (HBM section): FETCH_SIZE (KB) counts 128-byte read
requests at 64 bytes, so it is doubled; WRITE_SIZE (KB) is taken as is.  Both counters sit on the memory side of L2
(Infinity-Cache hits included), one --pmc pass each.

Round 5: the blanket doubling is replaced by the L2's own request-size counters when a third pass is given —
`rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum` of the same command:
fetched bytes = 128 x RDREQ_128B + 64 x RDREQ_64B + 32 x RDREQ_32B per launch, exact for every access width.  `calib` summarises
tools/calib_fetch.py (known byte counts in the fused kernels' access widths) the same way: FETCH_SIZE x 2 is exact for contiguous
1 / 4 / 8 / 16-byte-per-lane streams (every request is a 128-byte line) and 5 % high for k_lm_schur / k_lm_trial (one request in ten is a
64-byte one); 8-byte gathers are NOT fetched at 64 bytes: a 48-byte landmark slot read in random order costs a 128-byte line.

  python tools/rocpd_extract.py pmc   fetch.db write.db out.json [bench.log] [command] [reqsize.db]
  python tools/rocpd_extract.py calib cal_fetch.db cal_reqsize.db out.json"""
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


def _reqsize(db):
    """per kernel: average 32 / 64 / 128-byte read requests per launch and the bytes they fetch"""
    out = {}
    c = sqlite3.connect(db).cursor()
    for k, name, avg in c.execute("select kernel_name, counter_name, avg(value) from counters_collection group by kernel_name, counter_name"):
        out.setdefault(k, {})[name] = avg
    for k, v in out.items():
        r32, r64, r128 = v.get("TCC_EA0_RDREQ_32B_sum", 0.0), v.get("TCC_EA0_RDREQ_64B_sum", 0.0), v.get("TCC_EA0_RDREQ_128B_sum", 0.0)
        v["fetch_bytes_exact"] = 32.0 * r32 + 64.0 * r64 + 128.0 * r128
        v["requests_accounted"] = (r32 + r64 + r128) / v["TCC_EA0_RDREQ_sum"] if v.get("TCC_EA0_RDREQ_sum") else None
    return out


def calib(db_fetch, db_req, out):
    known = {"k_calib_stream<HIP_vector_type<double, 2u>": ("16 B per lane, contiguous", 1 << 30), "k_calib_stream<double>": ("8 B per lane, contiguous", 1 << 30),
             "k_calib_stream<int>": ("4 B per lane, contiguous", 1 << 30), "k_calib_stream<unsigned char>": ("1 B per lane, contiguous", 1 << 30),
             "k_calib_slot24": ("24 B of a 48-byte slot, slots in random order (8-byte loads)", ((1 << 30) // 48) * 24)}
    c = sqlite3.connect(db_fetch).cursor()
    fs = {k: avg for k, avg in c.execute("select kernel_name, avg(value) from counters_collection where counter_name='FETCH_SIZE' group by kernel_name")}
    rq = _reqsize(db_req)
    res = {}
    for key, (what, nbytes) in known.items():
        kf = [k for k in fs if key in k]; kr = [k for k in rq if key in k]
        if not kf or not kr:
            continue
        f, r = fs[kf[0]] * 1024.0, rq[kr[0]]
        res[what] = dict(bytes_read_by_the_kernel=nbytes, FETCH_SIZE_bytes=f, exact_fetch_bytes=r["fetch_bytes_exact"], FETCH_SIZE_factor_to_exact=r["fetch_bytes_exact"] / f,
                         exact_over_known=r["fetch_bytes_exact"] / nbytes, requests_128B=r.get("TCC_EA0_RDREQ_128B_sum"), requests_64B=r.get("TCC_EA0_RDREQ_64B_sum"), requests_32B=r.get("TCC_EA0_RDREQ_32B_sum"))
    json.dump({"note": "tools/calib_fetch.py under rocprofv3 --pmc (1 GiB source, 4 x the Infinity Cache): FETCH_SIZE against the L2's request-size counters and the bytes the kernels read",
               "patterns": res}, open(out, "w"), indent=1, sort_keys=True)


def pmc(db_fetch, db_write, out, bench_json=None, command=None, db_req=None):
    res = {}
    for db, name in ((db_fetch, "FETCH_SIZE"), (db_write, "WRITE_SIZE")):
        c = sqlite3.connect(db).cursor()
        for k, n, avg in c.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name=? group by kernel_name", (name,)):
            res.setdefault(k, {})[name + "_KB_avg"] = avg
            res[k]["launches_" + name] = n
    rq = _reqsize(db_req) if db_req else {}
    for k, v in res.items():
        f, w = v.get("FETCH_SIZE_KB_avg"), v.get("WRITE_SIZE_KB_avg")
        if k in rq and rq[k]["fetch_bytes_exact"] > 0 and w is not None:      # exact: the L2's read requests by size
            v["fetch_bytes_exact"] = rq[k]["fetch_bytes_exact"]
            v["read_requests"] = {n: rq[k].get("TCC_EA0_RDREQ_%s_sum" % n) for n in ("32B", "64B", "128B")}
            if f:
                v["FETCH_SIZE_factor_to_exact"] = rq[k]["fetch_bytes_exact"] / (f * 1024.0)
            v["traffic_bytes_per_launch"] = rq[k]["fetch_bytes_exact"] + w * 1024.0
            v["traffic_method"] = "128 x RDREQ_128B + 64 x RDREQ_64B + 32 x RDREQ_32B + WRITE_SIZE"
        elif f is not None and w is not None:
            v["traffic_bytes_per_launch"] = (2.0 * f + w) * 1024.0
            v["traffic_method"] = "2 x FETCH_SIZE + WRITE_SIZE"
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
    json.dump({"note": "per-launch averages; traffic = fetched bytes from the L2's read-request counters by size (round 5) + WRITE_SIZE — or (2 x FETCH_SIZE + WRITE_SIZE) KB, the gfx950 correction of MI355X_MICROARCH.md, where no request-size pass was made; "
                       "command: " + (command or "rocprofv3 --pmc <counter> -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-config5-leg") + " (one pass per counter)",
               "csrc_sha16": h.hexdigest()[:16],      # bench.py quotes these numbers only for the kernel sources they were measured on
               "kernels": res}, open(out, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "calib":
        calib(sys.argv[2], sys.argv[3], sys.argv[4])
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5] if len(sys.argv) > 5 else None, sys.argv[6] if len(sys.argv) > 6 else None, sys.argv[7] if len(sys.argv) > 7 else None)
