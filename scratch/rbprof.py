import sys, os, ctypes; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepcoro_clip_amd import functional as Fn, kernels as K, _lib as L
sys.argv = ["x"]
import importlib.util
spec = importlib.util.spec_from_file_location("rb", "tools/relpos_bench.py"); rb = importlib.util.module_from_spec(spec)
lib = L.lib()
fn = lib.dcc_relpos_prof_read; fn.argtypes = [ctypes.c_void_p, ctypes.c_int]; fn.restype = ctypes.c_int
D = 96
for name in ("blk4-13", "blk0"):
    H, q_thw, k_thw = {"blk4-13": (4, (8, 14, 14), (8, 7, 7)), "blk0": (1, (8, 56, 56), (8, 7, 7))}[name]
    B = 32; A = Fn.relpos_aug_width(k_thw); DQK = D + A; Nq = q_thw[0]*q_thw[1]*q_thw[2] + 1
    idx, lens = [], []
    for a in (1, 2, 0):
        qs, ks = q_thw[a], k_thw[a]; d = 2*max(qs, ks) - 1
        idx.append(Fn.relpos_index(qs, ks).cuda()); lens.append(d)
    Lp = 8*((sum(lens)+7)//8)
    Rcat = (torch.randn(Lp, D, device="cuda")*0.5).bfloat16()
    dims = K.relpos_dims(B, H, D, A, Lp, q_thw, k_thw, lens)
    Qf = (torch.randn(B, Nq, H, DQK, device="cuda")*0.5).bfloat16(); dQf = torch.randn_like(Qf)
    dR = torch.zeros(Lp, D, device="cuda")
    for _ in range(2): K.relpos_aug_bwd(dQf, Qf, Rcat, idx, dims, D**-0.5, dR)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong*8)(); fn(buf, 1)
    n = 5
    for _ in range(n): K.relpos_aug_bwd(dQf, Qf, Rcat, idx, dims, D**-0.5, dR)
    torch.cuda.synchronize(); fn(buf, 1)
    rows = B*Nq*H; tiles = (rows+63)//64
    print(name, "tiles", tiles, "ticks/launch: prologue %d  top-barrier %d  zero+stage %d  scatter+prefetch %d  product2 %d  product1+store %d  ws-write %d" % tuple(b//n for b in list(buf)[:7]))
