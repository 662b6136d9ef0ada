"""Summarise a rocprofv3 --pmc run: per kernel name, mean of each counter per dispatch (steady-state
dispatches only: the largest 50 % by SQ_WAVES or all if absent)."""
import csv, sys, collections, glob
d = sys.argv[1]
f = glob.glob(d + "/*/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].split("(")[0].replace("void lr::", "").replace("lr::", "")
    acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, cs in sorted(acc.items()):
    n = len(next(iter(cs.values())))
    if n < 3: continue
    print(f"{name:28s} n={n:5d} " + "  ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(cs.items())))
