"""Aggregate rocprofv3 --pmc results (csv) per kernel name: usage pmc_summary.py DIR COUNTER"""
import csv, glob, sys, collections, re
d, ctr = sys.argv[1], sys.argv[2]
files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: [0, 0.0])
for f in files:
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") != ctr: continue
        name = re.sub(r"\(anonymous namespace\)::|^void ", "", r["Kernel_Name"])
        fam = "gemm" if ("gemm_" in name or "ln_mlp_kernel" in name) else ("attn" if "attn_" in name else name.split("(")[0][:60])
        agg[fam][0] += 1; agg[fam][1] += float(r["Counter_Value"])
for fam, (n, v) in sorted(agg.items(), key=lambda x: -x[1][1])[:12]:
    print(f"{fam:60s} launches {n:6d} total {v:.4g} per-launch {v/n:.4g}")
