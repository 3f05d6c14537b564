"""Which element of A / B sits in which lane / byte of the gfx950 fp8 MFMA operands?  Exact integer data (fp8 e4m3
represents 0..16 exactly), candidate layouts checked against A @ B^T.  Build: hipcc -shared (see bottom), run on the GPU box."""
import ctypes, itertools, os, subprocess, sys
import numpy as np, torch

here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "probe", "libfp8probe.so"))
dev = "cuda"


def e4m3(v):  # torch does the encoding (exact for small ints)
    return torch.tensor(v, dtype=torch.float32).to(torch.float8_e4m3fn).view(torch.uint8).numpy()


rng = np.random.default_rng(0)
A = rng.integers(0, 5, size=(32, 64)).astype(np.float32)   # [row][k]
B = rng.integers(0, 5, size=(32, 64)).astype(np.float32)   # [col][k]
A8, B8 = e4m3(A), e4m3(B)
C_ref = A @ B.T


def run_scaled(a_lane_bytes, b_lane_bytes, sa=0x7f, sb=0x7f, fa=0, fb=0):
    a = torch.from_numpy(np.ascontiguousarray(a_lane_bytes)).view(torch.int32).to(dev)
    b = torch.from_numpy(np.ascontiguousarray(b_lane_bytes)).view(torch.int32).to(dev)
    out = torch.zeros(64 * 16, device=dev)
    assert lib.fp8_probe_scaled(ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(b.data_ptr()), ctypes.c_void_p(out.data_ptr()), sa, sb, fa, fb) == 0
    o = out.cpu().numpy().reshape(64, 16)
    C = np.zeros((32, 32), np.float32)
    for l in range(64):
        for r in range(16):
            C[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5), l & 31] = o[l, r]   # standard 32x32 C/D map
    return C


def lanes_from(M8, kmap):
    """lane l gets bytes M8[l & 31][kmap(l >> 5, j)] for j = 0..31"""
    out = np.zeros((64, 32), np.uint8)
    for l in range(64):
        for j in range(32):
            out[l, j] = M8[l & 31, kmap(l >> 5, j)]
    return out


hyp = {
    "k = 32h + j": lambda h, j: 32 * h + j,
    "k = 16h + j (j<16), 32 + 16h + j-16": lambda h, j: (16 * h + j) if j < 16 else (32 + 16 * h + j - 16),
    "k = 8h + (j%8) + 16*(j//8)": lambda h, j: 8 * h + (j % 8) + 16 * (j // 8),
    "k = 2j + h": lambda h, j: 2 * j + h,
}
for name, km in hyp.items():
    C = run_scaled(lanes_from(A8, km), lanes_from(B8, km))
    print(f"scaled 32x32x64 [{name}]: max |C - ref| = {np.abs(C - C_ref).max()}  (C^T match: {np.abs(C.T - C_ref).max()})")
# scale semantics with the working layout: scale byte 0x80 on A doubles?
km = hyp["k = 32h + j"]
for sa, sb in ((0x7f, 0x7f), (0x80, 0x7f), (0x7f, 0x7e), (0x80808080 - (1 << 32), 0x7f7f7f7f)):
    C = run_scaled(lanes_from(A8, km), lanes_from(B8, km), sa, sb)
    print(f"scales a={sa & 0xffffffff:#x} b={sb & 0xffffffff:#x}: C[0,0]/ref = {C[0, 0] / max(C_ref[0, 0], 1)}")

# plain 32x32x16 fp8: lane (r, h) holds k = 8h + j ?
A16, B16 = A8[:, :16], B8[:, :16]
def lanes8(M8):
    out = np.zeros((64, 8), np.uint8)
    for l in range(64):
        out[l] = M8[l & 31, 8 * (l >> 5): 8 * (l >> 5) + 8]
    return out
a = torch.from_numpy(lanes8(A16)).view(torch.int64).to(dev); b = torch.from_numpy(lanes8(B16)).view(torch.int64).to(dev)
out = torch.zeros(64 * 16, device=dev)
lib.fp8_probe_plain(ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(b.data_ptr()), ctypes.c_void_p(out.data_ptr()))
o = out.cpu().numpy().reshape(64, 16); C = np.zeros((32, 32), np.float32)
for l in range(64):
    for r in range(16):
        C[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5), l & 31] = o[l, r]
print("plain 32x32x16 fp8 [k = 8h + j]: max err", np.abs(C - A[:, :16] @ B[:, :16].T).max())

# conversion semantics vs torch (round-to-nearest-even, saturation?)
x = torch.tensor([0.1, -0.3, 1.0, 1.7, 3.3, 100.0, 447.0, 448.0, 449.0, 480.0, 1000.0, -1e5, 1e-3, 2e-3, 0.0019, 0.017], dtype=torch.float32)
for bf8, dt in ((0, torch.float8_e4m3fn), (1, torch.float8_e5m2)):
    xo = torch.zeros(8, dtype=torch.int32, device=dev)
    lib.fp8_probe_cvt(ctypes.c_void_p(x.to(dev).data_ptr()), ctypes.c_void_p(xo.data_ptr()), 8, bf8)
    got = xo.cpu().numpy().astype(np.uint32)
    hw = np.stack([got & 0xff, (got >> 8) & 0xff], 1).reshape(-1).astype(np.uint8)
    ref = x.to(dt).view(torch.uint8).numpy()
    print("cvt", "bf8" if bf8 else "fp8", "hw", [hex(v) for v in hw], "torch", [hex(v) for v in ref])
