"""Summarise a rocprofv3 --kernel-trace result (rocpd .db or csv): per (kernel, grid) time per step.
usage: prof_summary.py DIR STEPS [TOP]"""
import csv, glob, sys, collections, re, sqlite3
d, steps = sys.argv[1], int(sys.argv[2]); top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rows = []
dbs = glob.glob(d + "/**/*.db", recursive=True)
if dbs:
    c = sqlite3.connect(dbs[0])
    rows = [(n, (e - s) / 1e3, gx, gy, v) for n, s, e, gx, gy, v in c.execute("select name, start, end, grid_x, grid_y, vgpr_count from kernels")]
else:
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    for r in csv.DictReader(open(f)):
        rows.append((r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size_X", ""), r.get("Grid_Size_Y", ""), r.get("VGPR_Count", "")))
agg = collections.defaultdict(lambda: [0, 0.0]); fam = collections.defaultdict(float)
for name, dur, gx, gy, v in rows:
    name = re.sub(r"^void |\(anonymous namespace\)::", "", name)[:100]
    agg[(name, gx, gy, v)][0] += 1; agg[(name, gx, gy, v)][1] += dur; fam[name.split("(")[0][:70]] += dur
tot = sum(v[1] for v in agg.values())
print(f"total kernel time {tot/steps/1e3:.2f} ms/step over {steps} steps (incl. warmup)")
for n, t in sorted(fam.items(), key=lambda x: -x[1])[:30]: print(f"  {t/steps:9.1f} us/step  {n}")
print("--- per grid")
for (n, gx, gy, v), (c, t) in sorted(agg.items(), key=lambda x: -x[1][1])[:top]:
    print(f"{t/steps:9.1f} us/step {c/steps:6.1f}x avg {t/c:8.1f} us grid {gx}x{gy} vgpr {v} {n}")
