"""Time per step during which the GPU runs fewer than THR workgroups in total (under-filled: 256 CUs), attributed to
the kernels running then.  usage: prof_underfill.py DIR [THR]"""
import glob, sys, sqlite3, collections, re
db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
thr = int(sys.argv[2]) if len(sys.argv) > 2 else 128
c = sqlite3.connect(db)
rows = c.execute("select start, end, name, grid_x, grid_y, grid_z, workgroup_x, workgroup_y, workgroup_z from kernels order by start").fetchall()
steps = [e for s, e, n, *_ in rows if "adam_kernel" in n]
a, b = steps[3], steps[-2]; nsteps = len(steps) - 5
ev = []
for s, e, n, gx, gy, gz, wx, wy, wz in rows:
    if s < a or e > b: continue
    wgs = max(1, (gx // max(1, wx)) * (gy // max(1, wy)) * (gz // max(1, wz)))
    n = re.sub(r"^void |\(anonymous namespace\)::|_ZN12_GLOBAL__N_1\d+", "", n)[:60]
    ev.append((s, 1, wgs, n)); ev.append((e, -1, wgs, n))
ev.sort()
cur = collections.Counter(); tot = 0; last = a
under = collections.defaultdict(float); idle = 0.0; under_t = 0.0
for t, kind, wgs, n in ev:
    dt = t - last
    if dt > 0:
        if tot == 0: idle += dt
        elif tot < thr:
            under_t += dt
            for k in cur:
                if cur[k] > 0: under[k] += dt
    last = t
    tot += kind * wgs; cur[n] += kind
print(f"per step: idle {idle/nsteps/1e6:.2f} ms, under-filled (<{thr} workgroups) {under_t/nsteps/1e6:.2f} ms, window {(b-a)/nsteps/1e6:.2f} ms")
for k, v in sorted(under.items(), key=lambda x: -x[1])[:25]:
    print(f"  {v/nsteps/1e3:8.1f} us/step  {k}")
