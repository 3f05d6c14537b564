"""Phase profile of the ring GEMM (needs a library built with DCC_CFLAGS=-DDCC_RING_PROFILE)."""
import sys, ctypes; sys.path.insert(0, '.')
import torch
from deepcoro_clip_amd import kernels as K, _lib as L
lib = L.lib()
prof = lib.dcc_gemm_ring_profile
prof.argtypes = [ctypes.c_void_p, ctypes.c_int]; prof.restype = ctypes.c_int
def run(M, N, Kd, lay, **kw):
    a_km, b_km = lay[0] == 'T', lay[1] == 'T'
    a = torch.randn((Kd, M) if a_km else (M, Kd), device='cuda').bfloat16()
    b = torch.randn((Kd, N) if b_km else (N, Kd), device='cuda').bfloat16()
    out = torch.empty(M, N, device='cuda', dtype=torch.float32)  # fp32 output -> the v1 ring kernel (profiling hooks)
    aux = None
    if kw.pop('pre', False): kw.update(save_pre=True, aux=aux)
    for _ in range(2): K.gemm(a, b, a_km=a_km, b_km=b_km, out=out, **kw)
    torch.cuda.synchronize(); prof(None, 1)
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record(); reps = 5
    for _ in range(reps): K.gemm(a, b, a_km=a_km, b_km=b_km, out=out, **kw)
    e.record(); torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 8)(); prof(buf, 0)
    v = [x / reps for x in buf]
    us = s.elapsed_time(e) / reps * 1e3
    nblk = 256
    steps = v[3] / nblk
    print(f"M{M} N{N} K{Kd} {lay} {kw.keys()}: {us:.1f} us; per block: steps {steps:.0f}; consumer ticks: barrier-wait {v[0]/nblk:.0f} mfma {v[1]/nblk:.0f} epilogue {v[2]/nblk:.0f} | loader: vmwait {v[4]/nblk:.0f} barrier {v[5]/nblk:.0f} issue {v[6]/nblk:.0f}")
import os
for v in (0, 4):
    K.GEMM_VARIANT = v << 8
    print('variant', v, '(4 = no DMA)')
    run(8192, 8192, 8192, 'NT')
    run(4096, 4096, 4096, 'NN')
    run(37656, 384, 1536, 'NN')
