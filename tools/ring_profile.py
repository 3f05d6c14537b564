"""Crew timers of the ring GEMM kernels (build with -DDCC_RING_PROFILE: python deepcoro_clip_amd/build.py --variant prof
gemm_ring.hip=-DDCC_RING_PROFILE; run with DCC_LIB=<that .so>): where a K-step's time goes for one shape.
Counters (s_memtime ticks summed over one consumer wave and one loader wave per workgroup): [0] consumer barrier wait, [1] MFMA
section, [2] epilogue hand-off, [3] K-steps, [4] loader vmcnt wait, [5] loader barrier wait, [6] loader issue.
usage: DCC_LIB=... python tools/ring_profile.py M N K [b_km]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepcoro_clip_amd import _lib as L, kernels as K

M, N, Kd = (int(v) for v in sys.argv[1:4])
b_km = len(sys.argv) > 4 and sys.argv[4] == "1"
g = torch.Generator(device="cuda").manual_seed(0)
a = (torch.rand((M, Kd), device="cuda", generator=g) * 2 - 1).bfloat16()
b = (torch.rand((Kd, N) if b_km else (N, Kd), device="cuda", generator=g) * 2 - 1).bfloat16()
out = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
K.GEMM_VARIANT = 10 << 8   # ring kernels only
lib = L.lib()
lib.dcc_gemm_ring_profile.argtypes = [ctypes.c_void_p, ctypes.c_int]
for _ in range(3):
    K.gemm(a, b, b_km=b_km, out=out)
torch.cuda.synchronize()
lib.dcc_gemm_ring_profile(None, 1)
reps = 10
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(reps):
    K.gemm(a, b, b_km=b_km, out=out)
e.record()
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 8)()
lib.dcc_gemm_ring_profile(buf, 0)
v = list(buf)
steps = max(1, v[3])
print(f"M{M} N{N} K{Kd} {'NT' if b_km else 'NN'}: {s.elapsed_time(e) / reps * 1e3:.1f} us per call (profile build)")
names = ["consumer barrier wait", "consumer MFMA section", "consumer epilogue hand-off", "steps", "loader vmcnt wait", "loader barrier wait", "loader issue", "storer"]
for i, n in enumerate(names):
    if i == 3:
        print(f"  K-steps counted (per consumer wave, summed over workgroups and calls): {v[3]}")
    else:
        print(f"  {n:28s} {v[i] / steps:9.1f} ticks per K-step")
