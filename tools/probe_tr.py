import ctypes, os, sys, torch
here = os.path.dirname(os.path.abspath(__file__))
L = ctypes.CDLL(os.path.join(here, "libprobe_tr.so"))
def run(addrs):
    a = torch.tensor(addrs, dtype=torch.int32, device="cuda"); out = torch.zeros(256, dtype=torch.int16, device="cuda")
    L.run_probe(ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    return out.cpu().view(64, 4).tolist()
# experiment 1: the guide's layout: lane l -> byte address of u16 index (l&15)*?  Use row-major [k][16 cols] tile: rows of 16 u16 (32 B)
# lane i of a 16-lane group points at row (i>>2) [k], 4-col chunk (i&3): u16 index = (i>>2)*16 + (i&3)*4 ; groups offset by 64 u16
addrs = [(((l & 15) >> 2) * 16 + ((l & 15) & 3) * 4 + (l >> 4) * 64) * 2 for l in range(64)]
r = run(addrs)
print("exp1 (lane i -> row i>>2, chunk i&3 of a 4x16 tile per 16-lane group; values are u16 indices = k*16+n within group*64)")
for l in range(64): print(l, addrs[l] // 2, r[l])
# experiment 2: row stride 128 u16 instead of 16 (free row stride?)
addrs = [(((l & 15) >> 2) * 128 + ((l & 15) & 3) * 4 + (l >> 4) * 512) * 2 for l in range(64)]
r = run(addrs)
print("exp2 (row stride 128 u16)")
for l in range(0, 64, 1): print(l, addrs[l] // 2, r[l])
