"""HBM traffic of the GEMM family from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, kernel trace only).
usage: pmc_traffic.py FETCH_DIR WRITE_DIR OUT.json dtype batch videos frames size pooling
FETCH_SIZE is doubled (gfx950 counts half of wide coalesced reads: MI355X_MICROARCH.md, HBM section; calibration in
profiles/r01_c_pmc_calibration.txt); both counters are in KB."""
import csv, glob, json, re, sys, collections

fd, wd, out = sys.argv[1:4]
workload = [sys.argv[4], int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7]), int(sys.argv[8]), sys.argv[9]]


def family_totals(d, ctr):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != ctr:
                continue
            name = re.sub(r"\(anonymous namespace\)::|^void ", "", r["Kernel_Name"])
            fam = "gemm" if ("gemm_" in name or "ln_mlp_kernel" in name) else ("attn" if "attn_" in name else name.split("(")[0][:60])
            agg[fam][0] += 1
            agg[fam][1] += float(r["Counter_Value"])
    return agg


f, w = family_totals(fd, "FETCH_SIZE"), family_totals(wd, "WRITE_SIZE")
n = f["gemm"][0]
rec = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over bench.py; FETCH_SIZE doubled (gfx950 rule); KB units",
       "workload": workload, "family": "gemm (all gemm_* launches)", "launches": n, "fetch_kb_raw": f["gemm"][1], "write_kb": w["gemm"][1],
       "hbm_bytes_per_launch": int((2.0 * f["gemm"][1] + w["gemm"][1]) * 1024 / max(n, 1)),
       "attn_launches": f["attn"][0], "attn_hbm_bytes_per_launch": int((2.0 * f["attn"][1] + w["attn"][1]) * 1024 / max(f["attn"][0], 1))}
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps(rec))
for name, agg in (("FETCH_SIZE", f), ("WRITE_SIZE", w)):
    print(f"--- {name} by kernel family (KB)")
    for fam, (c, v) in sorted(agg.items(), key=lambda x: -x[1][1])[:14]:
        if c == 0:
            continue
        print(f"{fam:60s} launches {c:6d} total {v:.4g} per-launch {v / c:.4g}")
