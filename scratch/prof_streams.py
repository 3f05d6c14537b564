"""Per-stream kernel time per step from a rocprofv3 rocpd .db: which stream is the critical chain?  usage: DIR"""
import glob, sys, sqlite3, collections, re
db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
rows = c.execute("select start, end, name, stream_id, queue_id from kernels order by start").fetchall()
steps = [e for s, e, n, *_ in rows if "adam_kernel" in n]
a, b = steps[3], steps[-2]; ns = len(steps) - 5
per = collections.defaultdict(float); cnt = collections.Counter(); fam = collections.defaultdict(lambda: collections.defaultdict(float))
for s, e, n, sid, qid in rows:
    if s < a or e > b: continue
    key = (sid, qid); per[key] += e - s; cnt[key] += 1
    n = re.sub(r"^void |\(anonymous namespace\)::|_ZN12_GLOBAL__N_1\d+", "", n)[:40]
    fam[key][n] += e - s
print(f"window {(b-a)/ns/1e6:.2f} ms/step")
for k, v in sorted(per.items(), key=lambda x: -x[1]):
    print(f"stream {k}: {v/ns/1e6:6.2f} ms/step in {cnt[k]/ns:6.0f} kernels; top: " + ", ".join(f"{n} {t/ns/1e3:.0f}us" for n, t in sorted(fam[k].items(), key=lambda x: -x[1])[:6]))
