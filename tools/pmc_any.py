"""Mean of arbitrary PMC counters per kernel from one rocprofv3 --pmc pass (--output-format csv):
    python tools/pmc_any.py <dir> '<kernel regex>'      -> one line per (kernel, grid): launches, counter means"""
import collections, csv, glob, re, sys

f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
rx = re.compile(sys.argv[2])
d = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    m = rx.search(r["Kernel_Name"])
    if m:
        d[(m.group(0), r.get("Grid_Size", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key in sorted(d):
    c = d[key]
    n = max(len(v) for v in c.values())
    print(key[0], "grid", key[1], "launches", n, " ".join(f"{k}={sum(v) / len(v):.0f}" for k, v in sorted(c.items())))
