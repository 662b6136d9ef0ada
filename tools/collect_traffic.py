"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into profiles/traffic.json.

Per MI355X_MICROARCH.md (HBM section): counters are in KiB; on gfx950 FETCH_SIZE reports half of the bytes
of a wide coalesced read stream, so the read side is doubled (an upper bound for narrow accesses); WRITE_SIZE
is exact for 16-B-per-lane stores.  usage: collect_traffic.py <fetch_dir> <write_dir> <out.json>
"""
import collections, csv, glob, json, sys


def per_kernel(d, counter):
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("lr::", "").split("<")[0]
            acc[name].append(float(r["Counter_Value"]))
    return acc


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
out = {"note": "HBM bytes per launch from rocprofv3 PMC passes; fetch side doubled per the gfx950 correction"}
for k in sorted(set(fetch) | set(write)):
    if not k.startswith("k_"):
        continue
    f = sum(fetch.get(k, [0])) / max(len(fetch.get(k, [0])), 1) * 1024.0
    w = sum(write.get(k, [0])) / max(len(write.get(k, [0])), 1) * 1024.0
    out[k + "_fetch_bytes_raw_per_launch"] = round(f)
    out[k + "_write_bytes_per_launch"] = round(w)
    out[k + "_hbm_bytes_per_launch"] = round(2 * f + w)
    out[k + "_launches_sampled"] = len(fetch.get(k, []))
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
