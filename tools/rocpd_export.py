"""Export rocprofv3 (ROCm 7.2 rocpd .db output) into the files kept under profiles/.

  rocpd_export.py stats   <kt_results.db> <out.csv>              per-kernel calls / total / average (us)
  rocpd_export.py traffic <fetch.db> <write.db> <out.json> [k=v ...]   HBM bytes per launch from the two PMC passes
  rocpd_export.py pmc     <out.json> <pass.db> [<pass.db> ...] [k=v ...] per-kernel means of every counter in the passes

Trailing k=v pairs are stored under "workload" (scene, film size, spp, git revision, ...): bench.py only
attaches a traffic / PMC figure to its JSON line when the workload recorded here matches the run.

Traffic follows MI355X_MICROARCH.md's HBM section: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
reports half of the bytes of a wide coalesced read stream, so the read side is doubled (an upper bound for
narrow accesses); WRITE_SIZE is taken as is.
"""
import collections, csv, json, sqlite3, sys


def short(name):
    return name.split("(")[0].replace("void ", "").replace("lr::", "")


def workload(args):
    out = {}
    for a in args:
        if "=" in a:
            k, v = a.split("=", 1)
            try:
                v = int(v)
            except ValueError:
                pass
            out[k] = v
    return out


def stats(db, out):
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
        for r in rows:
            w.writerow([r[0], r[1], round(r[2], 3), round(r[3], 3), round(r[4], 4)])
            print(f"{short(r[0]):28s} calls={r[1]:5d} avg_us={r[3]:12.3f} pct={r[4]:.3f}")


def per_kernel(db, counter, full_name=False):
    c = sqlite3.connect(db)
    acc = collections.defaultdict(dict)
    for disp, name, val in c.execute(
            "select dispatch_id, kernel_name, value from counters_collection where counter_name = ?", (counter,)):
        key = short(name) if full_name else short(name).split("<")[0]
        d = acc[key]
        d[disp] = d.get(disp, 0.0) + float(val)          # sum over XCD / channel instances of one dispatch
    return {k: list(v.values()) for k, v in acc.items()}


def traffic(fdb, wdb, out, extra):
    fetch, write = per_kernel(fdb, "FETCH_SIZE"), per_kernel(wdb, "WRITE_SIZE")
    res = {"note": "HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes; fetch side doubled "
                   "per the gfx950 correction", "workload": workload(extra)}
    dur = durations(fdb)
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith("k_"):
            continue
        fv, wv = fetch.get(k, [0.0]), write.get(k, [0.0])
        f, w = sum(fv) / len(fv) * 1024.0, sum(wv) / len(wv) * 1024.0
        res[k + "_fetch_bytes_raw_per_launch"] = round(f)
        res[k + "_write_bytes_per_launch"] = round(w)
        res[k + "_hbm_bytes_per_launch"] = round(2 * f + w)
        res[k + "_launches_sampled"] = len(fv)
        if k in dur and dur[k][0] > 0:
            # duration of the same launches in the (counter-collecting, serialised) fetch pass: a LOWER bound of the rate
            res[k + "_avg_us_in_pmc_pass"] = round(dur[k][0], 3)
            res[k + "_hbm_GBps_in_pmc_pass"] = round((2 * f + w) / (dur[k][0] * 1e-6) / 1e9, 2)
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


def durations(db, full_name=False):
    """mean dispatch duration (us) and count per kernel, from the kernel-dispatch records of a pass"""
    c = sqlite3.connect(db)
    acc = collections.defaultdict(list)
    try:
        rows = c.execute("select name, start, end from kernels")
    except sqlite3.Error:
        return {}
    for name, s, e in rows:
        key = short(name) if full_name else short(name).split("<")[0]
        acc[key].append((e - s) / 1e3)
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def pmc(out, dbs, extra):
    res = {"note": "rocprofv3 --pmc passes, mean per launch (summed over XCDs / SEs of a dispatch); SQ_*_CYCLES and SQ_WAIT_* "
                   "count quad-cycles per wave (MI355X_MICROARCH.md cycle-constants table)", "workload": workload(extra), "kernels": {}}
    for db in dbs:
        c = sqlite3.connect(db)
        names = [r[0] for r in c.execute("select distinct counter_name from counters_collection")]
        dur = durations(db, full_name=True)
        for cn in names:
            for k, vals in per_kernel(db, cn, full_name=True).items():
                if not k.startswith("k_"):
                    continue
                e = res["kernels"].setdefault(k, {})
                e[cn] = round(sum(vals) / len(vals), 1)
                e.setdefault("launches_sampled", len(vals))
                if k in dur:
                    e.setdefault("avg_us_in_pmc_pass", round(dur[k][0], 3))
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "pmc":
        rest = sys.argv[3:]
        pmc(sys.argv[2], [a for a in rest if "=" not in a], [a for a in rest if "=" in a])
    else:
        traffic(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5:])
