// Probe: buffer_load_dwordx4 ... lds (LDS-DMA) semantics on gfx950: destination = M0 base + lane*16, out-of-range lanes write zeros.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef int v4i __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16(unsigned voff, v4i rsrc, unsigned soff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_dst) : "memory");
}

__global__ void probe(const unsigned* src, unsigned nbytes, const unsigned* voffs, unsigned soff, unsigned* out) {
    __shared__ __attribute__((aligned(16))) unsigned lds[2048];
    const int lane = threadIdx.x;
    for (int i = lane; i < 2048; i += 64) lds[i] = 0xdeadbeefu;
    __syncthreads();
    const uint64_t a = (uint64_t)src;
    v4i rs;
    rs.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    rs.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32) & 0xffff);
    rs.z = __builtin_amdgcn_readfirstlane((int)nbytes);
    rs.w = 0x00020000;
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned*)lds;
    dma16(voffs[lane], rs, soff, __builtin_amdgcn_readfirstlane(base + 1024));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 2048; i += 64) out[i] = lds[i];
}

extern "C" void run_probe(const void* src, unsigned nbytes, const void* voffs, unsigned soff, void* out, void* stream) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, (hipStream_t)stream, (const unsigned*)src, nbytes, (const unsigned*)voffs, soff, (unsigned*)out);
}
