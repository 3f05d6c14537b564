"""Phase timers of the fused rel-pos backward kernel (wave 0 of every block, s_memtime ticks at 100 MHz), with and without the
LayerNorm epilogue.  Builds a -DRB_PROF copy of the library; usage: python tools/relpos_prof.py [blk ...]"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if "DCC_LIB" not in os.environ:
    lib = subprocess.run([sys.executable, os.path.join(ROOT, "deepcoro_clip_amd", "build.py"), "--variant", "rbprof", "relpos.hip=-DRB_PROF"],
                         capture_output=True, text=True, check=True).stdout.strip().splitlines()[-1]
    os.environ["DCC_LIB"] = lib
import torch
from deepcoro_clip_amd import functional as Fn, kernels as K, _lib as L
from tools.relpos_bench import BLOCKS, CLIPS

NAMES = ["prologue", "barrier0", "stage", "scatter", "product2", "product1", "final", "epi1(+bfly)", "barrierLN", "epi2"]


def main():
    D = 96
    lib = ctypes.CDLL(os.environ["DCC_LIB"])
    buf = (ctypes.c_ulonglong * 12)()
    for name in (sys.argv[1:] or ["blk4-13"]):
        H, q_thw, k_thw = BLOCKS[name]
        B = CLIPS
        A = Fn.relpos_aug_width(k_thw); DQK = D + A
        Nq = q_thw[0] * q_thw[1] * q_thw[2] + 1
        idx, lens = [], []
        for a in (1, 2, 0):
            qs, ks = q_thw[a], k_thw[a]
            idx.append(Fn.relpos_index(qs, ks).cuda()); lens.append(2 * max(qs, ks) - 1)
        Lp = 8 * ((sum(lens) + 7) // 8)
        Rcat = (torch.randn(Lp, D, device="cuda") * 0.5).bfloat16()
        dims = K.relpos_dims(B, H, D, A, Lp, q_thw, k_thw, lens)
        Qf = (torch.randn(B, Nq, H, DQK, device="cuda") * 0.5).bfloat16()
        dQf = torch.randn_like(Qf)
        rows = B * Nq * H
        conv = torch.randn(B, Nq, H, D, device="cuda").bfloat16()
        gamma = torch.ones(D, device="cuda"); beta = torch.zeros(D, device="cuda")
        _, mean, rstd = K.layernorm_fwd(conv.view(rows, D), gamma, beta, 1e-6)
        dg = torch.zeros(D, device="cuda"); db = torch.zeros(D, device="cuda"); dR = torch.zeros(Lp, D, device="cuda")
        for label, ln in (("plain", None), ("LN epilogue", (conv, mean, rstd, gamma, dg, db))):
            for _ in range(2):
                K.relpos_aug_bwd(dQf, Qf, Rcat, idx, dims, D ** -0.5, dR, ln=ln)
            torch.cuda.synchronize()
            lib.dcc_relpos_prof_read(buf, 1)
            reps = 5
            for _ in range(reps):
                K.relpos_aug_bwd(dQf, Qf, Rcat, idx, dims, D ** -0.5, dR, ln=ln)
            torch.cuda.synchronize()
            lib.dcc_relpos_prof_read(buf, 1)
            tiles = (rows + 63) // 64
            per = [buf[i] / reps / tiles for i in range(10)]
            # 5, 7, 9 are cumulative from the start of product 1
            p1, e1, e2 = per[5], per[7] - per[5], per[9] - per[7] - per[8]
            out = {"barrier0": per[1], "stage": per[2], "scatter": per[3], "product2": per[4], "product1": p1, "epi1(+bfly)": e1,
                   "barrierLN": per[8], "epi2": e2}
            print(f"{name} {label}: ticks per tile (10 ns): " + ", ".join(f"{k} {v:.0f}" for k, v in out.items())
                  + f";  tile total {sum(out.values()):.0f}", flush=True)


if __name__ == "__main__":
    main()
