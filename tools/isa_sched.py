"""Compile one HIP source to gfx950 assembly and print, per kernel whose name contains PATTERN: registers, scratch operations per
barrier interval, and the compressed opcode order of its hottest basic block (most MFMAs) — to check what the scheduler did.
usage: python tools/isa_sched.py csrc_file.hip PATTERN [extra hipcc flags...]"""
import os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "deepcoro_clip_amd", "csrc")
src, pat, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
out = "/tmp/isa_sched.s"
sys.path.insert(0, ROOT)
from deepcoro_clip_amd.build import FILE_FLAGS  # the per-file flags of the real build (e.g. attention.hip: MFMA in VGPR form)
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-fPIC", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-result", "-ffp-contract=fast", *FILE_FLAGS.get(src, []), "-I", CSRC,
       "-S", "--cuda-device-only", os.path.join(CSRC, src), "-o", out, *extra]
r = subprocess.run(cmd, capture_output=True, text=True)
if r.returncode:
    print(r.stderr[-3000:]); sys.exit(1)
s = open(out).read()
SHORT = {"v_mfma_f32_32x32x16_bf16": "MFMA", "ds_read_b128": "dsr", "ds_write_b128": "dsw", "buffer_load_dwordx4": "DMA", "s_waitcnt": "w", "v_accvgpr_read_b32": "ar",
         "v_accvgpr_write_b32": "aw", "global_store_dwordx4": "GST", "global_load_dwordx4": "GLD", "s_barrier": "BARRIER"}
for f in re.split(r"\n(?=_Z\w+:)", s):
    name = f.split(":")[0]
    if pat not in name or not name.startswith("_Z"):
        continue
    lines = f.split("\n")
    seg, cnt = 0, {}
    for l in lines:
        if "s_barrier" in l: seg += 1
        if "scratch_" in l: cnt[seg] = cnt.get(seg, 0) + 1
    print(name[:90], "scratch ops per barrier interval:", cnt, [l.strip() for l in lines if "NumVgprs" in l or "NumAgprs" in l][:2])
    blocks, cur, bn = [], [], "entry"
    for l in lines:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blocks.append((bn, cur)); bn, cur = m.group(1), []
        else:
            cur.append(l)
    blocks.append((bn, cur))
    top = sorted(blocks, key=lambda t: -sum("v_mfma" in l for l in t[1]))
    nshow = int(os.environ.get("ISA_BLOCKS", "1"))
    for bn, b in top[:nshow]:
        ops = [l.split()[0] for l in b if l.strip() and not l.strip().startswith((";", "."))]
        outl, prev, c = [], None, 0
        for o in ops:
            o = SHORT.get(o, o)
            if o == prev: c += 1
            else:
                if prev: outl.append(f"{prev}x{c}" if c > 1 else prev)
                prev, c = o, 1
        outl.append(f"{prev}x{c}" if c > 1 else prev)
        print(" ", bn, len(b), "lines,", sum("v_mfma" in l for l in b), "MFMAs:")
        print("  " + " ".join(outl))
