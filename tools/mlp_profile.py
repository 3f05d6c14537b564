"""Phase timers of the fused LayerNorm + MLP kernel (build with -DDCC_MLP_PROFILE: python deepcoro_clip_amd/build.py --variant
mprof mlp_fused.hip=-DDCC_MLP_PROFILE; run with DCC_LIB=<that .so>): s_memtime ticks of wave 0 of every workgroup, per iteration
(= one 32-wide hidden tile: GEMM1 of tile i, GEMM2 of tile i-1 under the epilogue of tile i).
usage: DCC_LIB=... python tools/mlp_profile.py M C [save 0|1|2]    (0: inference, 1: pre + act, 2: pre only)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepcoro_clip_amd import _lib as L, kernels as K

M, C = int(sys.argv[1]), int(sys.argv[2])
mode = int(sys.argv[3]) if len(sys.argv) > 3 else 1
H = 4 * C
x = (torch.randn(M, C, device="cuda") * 1.2).bfloat16()
g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
w1 = (torch.randn(H, C, device="cuda") * C ** -0.5).bfloat16()
w2 = (torch.randn(C, H, device="cuda") * H ** -0.5).bfloat16()
b1, b2 = torch.zeros(H, device="cuda"), torch.zeros(C, device="cuda")
pk = K.mlp_pack(w1, w2)
run = lambda: K.ln_mlp_fwd(x, g, b, 1e-6, pk, b1, b2, save=mode > 0, save_act=mode == 1)  # noqa: E731
lib = L.lib()
lib.dcc_mlp_profile.argtypes = [ctypes.c_void_p, ctypes.c_int]
for _ in range(3):
    run()
torch.cuda.synchronize()
lib.dcc_mlp_profile(None, 1)
reps = 10
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(reps):
    run()
e.record()
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 8)()
lib.dcc_mlp_profile(buf, 0)
v = list(buf)
it = max(1, v[5])
print(f"M{M} C{C} mode {mode}: {s.elapsed_time(e) / reps * 1e3:.1f} us per call (profile build); {it} iterations timed")
for i, n in enumerate(["vmcnt wait", "barrier", "GEMM1 + DMA issue", "GEMM2 || epilogue", "swaps + stores", "", "tile head (x, LN, xn)", "y store"]):
    if n:
        print(f"  {n:24s} {v[i] / it:9.1f} ticks per iteration")
print(f"  sum {sum(v[i] for i in (0, 1, 2, 3, 4, 6, 7)) / it:9.1f} ticks per iteration (MFMA floor: {(C // 16 + C // 16) * 32} cycles)")
