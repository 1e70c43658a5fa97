import ctypes, os, torch
here = os.path.dirname(os.path.abspath(__file__))
L = ctypes.CDLL(os.path.join(here, "libprobe_dma.so"))
src = torch.arange(4096, dtype=torch.int32, device="cuda")          # word i holds i
def run(voffs, soff, nbytes=4096 * 4):
    v = torch.tensor(voffs, dtype=torch.int32, device="cuda"); out = torch.zeros(2048, dtype=torch.int32, device="cuda")
    L.run_probe(ctypes.c_void_p(src.data_ptr()), ctypes.c_uint(nbytes), ctypes.c_void_p(v.data_ptr()), ctypes.c_uint(soff),
                ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    return out.cpu().tolist()
# lane l reads 16 B at byte offset 64*l (words 16l..16l+3); lanes 5 and 9 out of range (offset >= nbytes)
voffs = [64 * l for l in range(64)]; voffs[5] = 4096 * 4; voffs[9] = 0x7fffff00
o = run(voffs, 0)
print("untouched before dst:", all(x == -559038737 for x in o[:256]), " after:", all(x == -559038737 for x in o[512:]))
for l in (0, 1, 4, 5, 6, 9, 10, 63): print("lane", l, "lds words", o[256 + 4 * l: 256 + 4 * l + 4])
# soffset = 32 bytes: expect words shifted by 8; range check with soffset: voffset near the end
voffs = [64 * l for l in range(64)]; voffs[63] = 4096 * 4 - 16
o = run(voffs, 32)
for l in (0, 1, 63): print("soff=32 lane", l, o[256 + 4 * l: 256 + 4 * l + 4])
