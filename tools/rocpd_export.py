"""Export rocprofv3 (ROCm 7.2 rocpd .db output) into the files kept under profiles/.

  rocpd_export.py stats   <kt_results.db> <out.csv>              per-kernel calls / total / average (us)
  rocpd_export.py traffic <fetch.db> <write.db> <out.json>       HBM bytes per launch from the two PMC passes

Traffic follows MI355X_MICROARCH.md's HBM section: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
reports half of the bytes of a wide coalesced read stream, so the read side is doubled (an upper bound for
narrow accesses); WRITE_SIZE is taken as is.
"""
import collections, csv, json, sqlite3, sys


def short(name):
    return name.split("(")[0].replace("void ", "").replace("lr::", "")


def stats(db, out):
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
        for r in rows:
            w.writerow([r[0], r[1], round(r[2], 3), round(r[3], 3), round(r[4], 4)])
            print(f"{short(r[0]):28s} calls={r[1]:5d} avg_us={r[3]:12.3f} pct={r[4]:.3f}")


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    acc = collections.defaultdict(dict)
    for disp, name, val in c.execute(
            "select dispatch_id, kernel_name, value from counters_collection where counter_name = ?", (counter,)):
        d = acc[short(name).split("<")[0]]
        d[disp] = d.get(disp, 0.0) + float(val)          # sum over XCD / channel instances of one dispatch
    return {k: list(v.values()) for k, v in acc.items()}


def traffic(fdb, wdb, out):
    fetch, write = per_kernel(fdb, "FETCH_SIZE"), per_kernel(wdb, "WRITE_SIZE")
    res = {"note": "HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes; fetch side doubled "
                   "per the gfx950 correction"}
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith("k_"):
            continue
        fv, wv = fetch.get(k, [0.0]), write.get(k, [0.0])
        f, w = sum(fv) / len(fv) * 1024.0, sum(wv) / len(wv) * 1024.0
        res[k + "_fetch_bytes_raw_per_launch"] = round(f)
        res[k + "_write_bytes_per_launch"] = round(w)
        res[k + "_hbm_bytes_per_launch"] = round(2 * f + w)
        res[k + "_launches_sampled"] = len(fv)
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    else:
        traffic(sys.argv[2], sys.argv[3], sys.argv[4])
