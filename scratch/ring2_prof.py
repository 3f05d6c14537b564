"""Phase profile of the bf16-output ring GEMM (crews: consumers / loaders / storers); needs DCC_CFLAGS=-DDCC_RING_PROFILE."""
import sys, ctypes; sys.path.insert(0, '.')
import torch
from deepcoro_clip_amd import kernels as K, _lib as L
lib = L.lib()
prof = lib.dcc_gemm_ring_profile
prof.argtypes = [ctypes.c_void_p, ctypes.c_int]; prof.restype = ctypes.c_int
def run(M, N, Kd, **kw):
    a = torch.randn(M, Kd, device='cuda').bfloat16(); b = torch.randn(N, Kd, device='cuda').bfloat16()
    out = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    for _ in range(2): K.gemm(a, b, out=out, **kw)
    torch.cuda.synchronize(); prof(None, 1)
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record(); reps = 5
    for _ in range(reps): K.gemm(a, b, out=out, **kw)
    e.record(); torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 8)(); prof(buf, 0)
    v = [x / reps / 256 for x in buf]
    us = s.elapsed_time(e) / reps * 1e3
    print(f"M{M} N{N} K{Kd}: {us:.1f} us; per workgroup (memtime ticks, 100 MHz): steps {v[3]:.0f} | consumer: barrier-wait {v[0]:.0f} mfma {v[1]:.0f} dump {v[2]:.0f} | loader: vmwait {v[4]:.0f} barrier {v[5]:.0f} issue {v[6]:.0f} | storer barrier-wait {v[7]:.0f}")
for v in (0, 7, 4):
    K.GEMM_VARIANT = v << 8
    print("variant", v, "(7 = storers do not store, 4 = loaders do not load)")
    run(37656, 1536, 384); run(37656, 1152, 384); run(37656, 384, 1536)
