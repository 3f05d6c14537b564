"""GPU busy time (union of kernel intervals) vs wall per step from a rocprofv3 rocpd .db: usage prof_union.py DIR"""
import glob, sys, sqlite3
db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
rows = sorted(c.execute("select start, end, name from kernels").fetchall())
# find adam_kernel launches = step boundaries
steps = [e for s, e, n in rows if "adam_kernel" in n]
print("steps seen", len(steps))
for a, b in list(zip(steps[:-1], steps[1:]))[2:8]:
    iv = [(s, e) for s, e, n in rows if s >= a and e <= b]
    busy = 0; cur_s, cur_e = iv[0]
    gaps = []
    for s, e in iv[1:]:
        if s > cur_e:
            busy += cur_e - cur_s; gaps.append(s - cur_e); cur_s, cur_e = s, e
        else: cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    gaps.sort(reverse=True)
    print(f"step {(b-a)/1e6:.2f} ms  busy {busy/1e6:.2f} ms  idle {(b-a-busy)/1e6:.2f} ms in {len(gaps)} gaps; top gaps us {[round(g/1e3) for g in gaps[:8]]}; kernels {len(iv)} sum {sum(e-s for s,e in iv)/1e6:.2f} ms")
