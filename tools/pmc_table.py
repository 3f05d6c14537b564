"""Per-kernel table of every counter in a rocprofv3 --pmc csv dump: usage pmc_table.py DIR [name-filter]
(mean per launch; kernels keyed by name + grid size)."""
import csv, glob, sys, collections, re
d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
ctrs = []
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(anonymous namespace\)::|^void ", "", r["Kernel_Name"])
        if flt and flt not in name: continue
        key = (name.split("(")[0][:70], r.get("Grid_Size", ""))
        c = r["Counter_Name"]
        if c not in ctrs: ctrs.append(c)
        a = agg[key][c]; a[0] += 1; a[1] += float(r["Counter_Value"])
for key, cs in sorted(agg.items()):
    print(f"{key[0]} grid {key[1]}")
    for c in ctrs:
        if c in cs: print(f"    {c:28s} {cs[c][1] / cs[c][0]:16.1f}  (x{cs[c][0]})")
