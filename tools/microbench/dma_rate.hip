// Delivery-rate micro-benchmark for the conv k-loops' operand path on gfx950 (round 4; numbers in profiles/r04_dma_rate.txt).
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/microbench/dma_rate tools/microbench/dma_rate.hip && tools/microbench/dma_rate
//
// Every workgroup streams a private, L2-resident region (default 40 KB: larger than the 32 KB vector L1, so nothing hits there; three workgroups per CU = 3.75 MB per XCD, inside its 4 MB L2)
// round and round with 16-byte-per-lane loads, 1 KB per wave instruction, and nothing else.  Swept: waves per workgroup, workgroups per CU,
// the instruction (buffer_load_dwordx4 ... lds  |  global_load_dwordx4 into VGPRs) and how many a wave keeps in flight.
// Output: bytes per clock and CU (clock = s_memtime / 100 MHz-independent: wall time x the shader clock reported by the runtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int i32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ i32x4_t make_rsrc(const void* p, unsigned bytes) {
    const unsigned long long a = (unsigned long long)p;
    i32x4_t r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
    r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    r[3] = 0x00020000;
    return r;
}

__device__ __forceinline__ void lds_dma16(unsigned voff, i32x4_t rsrc, unsigned soff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" :: "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_dst) : "memory");
}

// MODE 0: LDS-DMA.  MODE 1: loads into registers (consumed by an xor so that they are not dropped).  INFL instructions per wait.
template <int MODE, int INFL, int SEGB>
__global__ void __launch_bounds__(1024) rate_kernel(const unsigned char* __restrict__ src, unsigned region, int iters, unsigned* sink, unsigned lds_bytes) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const unsigned char* base = src + (size_t)blockIdx.x * region;
    const i32x4_t rsrc = make_rsrc(base, region);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const unsigned unit0 = (unsigned)__builtin_amdgcn_readfirstlane(wave) * INFL;      // destination KB units wrap inside the workgroup's allocation
    unsigned off = (unsigned)__builtin_amdgcn_readfirstlane(wave) * 1024u;       // wave w takes the KB units w, w + nw, ...
    const unsigned stride = (unsigned)nw * 1024u;
    // SEGB < 1024: the instruction's 1 KB is 1024 / SEGB pieces of SEGB bytes from as many rows of a [rows][512 B] matrix (the conv_igemm
    // operand tiles: 64 B = 32 bf16 channels of one pixel / filter row per k-step); unit u -> row group u % G, k-piece u / G
    constexpr unsigned PITCH = 512u, LPR = SEGB / 16u, RPI = 1024u / SEGB;
    const unsigned rows = region / PITCH, G = rows / RPI, KP = PITCH / SEGB;
    unsigned g = (unsigned)__builtin_amdgcn_readfirstlane(wave) % G, kp = ((unsigned)__builtin_amdgcn_readfirstlane(wave) / G) % KP;
    const unsigned lane_off = SEGB >= 1024 ? (unsigned)lane * 16u : ((unsigned)lane / LPR) * PITCH + ((unsigned)lane % LPR) * 16u;
    u32x4_t acc = {0u, 0u, 0u, 0u};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < INFL; ++q) {
            if (SEGB < 1024) {
                off = g * (RPI * PITCH) + kp * SEGB;
                g += (unsigned)nw;                       // counters, not divisions: the address arithmetic must not be what is measured
                while (g >= G) { g -= G; kp = (kp + 1 == KP) ? 0u : kp + 1; }
            }
            if (MODE == 0) {
                lds_dma16(lane_off, rsrc, off, lds0 + ((unit0 + q) * 1024u) % lds_bytes);
            } else {
                u32x4_t v;
                asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(v) : "v"(lane_off), "s"(rsrc), "s"(off) : "memory");
                asm volatile("s_waitcnt vmcnt(%0)" :: "n"(INFL - 1) : "memory");
                acc ^= v;
            }
            if (SEGB >= 1024) { off += stride; if (off >= region) off -= region; }
        }
        if (MODE == 0) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(INFL) : "memory");      // keep <= 2 * INFL in flight
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (MODE == 1 && (acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) sink[0] = 1;
    if (MODE == 0 && lds[lds_bytes & 1023] == 77 && lane == 63) sink[1] = 1;
}

template <int MODE, int INFL, int SEGB>
double run(const unsigned char* src, unsigned region, int wgs, int waves, int lds_bytes, int iters, unsigned* sink, int clk_khz, int cus) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)rate_kernel<MODE, INFL, SEGB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((rate_kernel<MODE, INFL, SEGB>), dim3(wgs), dim3(waves * 64), lds_bytes, 0, src, region, iters, sink, (unsigned)lds_bytes);
    hipEventRecord(e0);
    hipLaunchKernelGGL((rate_kernel<MODE, INFL, SEGB>), dim3(wgs), dim3(waves * 64), lds_bytes, 0, src, region, iters, sink, (unsigned)lds_bytes);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)wgs * waves * iters * INFL * 1024.0;
    const double clocks = ms * 1e-3 * clk_khz * 1e3;
    return bytes / clocks / cus;        // bytes per clock and CU
}

int main(int argc, char** argv) {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount, clk = prop.clockRate;      // kHz
    const unsigned region = argc > 1 ? (unsigned)atoi(argv[1]) * 1024u : 40u * 1024u;
    const bool pieces = argc > 2;        // second argument: the row-piece sweep instead of the instruction sweep
    const int max_wgs = cus * 8;
    unsigned char* src; hipMalloc(&src, (size_t)max_wgs * region); hipMemset(src, 1, (size_t)max_wgs * region);
    unsigned* sink; hipMalloc(&sink, 64); hipMemset(sink, 0, 64);
    printf("# %s, %d CUs, %.2f GHz; region per workgroup %u KB (%.1f MB per XCD at one workgroup per CU)\n", prop.gcnArchName, cus, clk * 1e-6, region / 1024,
           region / 1048576.0 * cus / 8);
    printf("# bytes per clock and CU;  x %d CUs x %.2f GHz = chip-wide\n", cus, clk * 1e-6);
    printf("%-44s %8s %8s %8s %8s %8s\n", "instruction, in flight per wave", "1 wave", "2", "4", "8", "12/16");
    const int iters = 4000;
    auto row = [&](const char* name, auto fn, int wg_per_cu, int lds_per_wg) {
        printf("%-44s", name);
        const int waves[5] = {1, 2, 4, 8, 16};
        for (int k = 0; k < 5; ++k) {
            int w = waves[k];
            if (w * wg_per_cu > 32) { printf(" %8s", "-"); continue; }
            printf(" %8.1f", fn(w, wg_per_cu, lds_per_wg));
        }
        printf("   (%d workgroup%s per CU)\n", wg_per_cu, wg_per_cu > 1 ? "s" : "");
        fflush(stdout);
    };
#define FN(MODE, INFL, SEGB) [&](int w, int wpc, int ldsb) { return run<MODE, INFL, SEGB>(src, region, cus * wpc, w, ldsb > 0 ? ldsb : w * INFL * 1024, iters / INFL, sink, clk, cus); }
    for (int wpc = 1; wpc <= 3; ++wpc) {
        const int ldsb = 48 * 1024;      // the conv kernels' ring: caps residency at three workgroups per CU like them
        if (pieces) {
            row("lds, 4 in flight, 1 KB contiguous", FN(0, 4, 1024), wpc, ldsb);
            row("lds, 4 in flight, 4 rows x 256 B", FN(0, 4, 256), wpc, ldsb);
            row("lds, 4 in flight, 8 rows x 128 B", FN(0, 4, 128), wpc, ldsb);
            row("lds, 4 in flight, 16 rows x 64 B", FN(0, 4, 64), wpc, ldsb);
            row("lds, 8 in flight, 16 rows x 64 B", FN(0, 8, 64), wpc, ldsb);
            continue;
        }
        row("buffer_load_dwordx4 lds, 1", FN(0, 1, 1024), wpc, ldsb);
        row("buffer_load_dwordx4 lds, 4", FN(0, 4, 1024), wpc, ldsb);
        row("buffer_load_dwordx4 lds, 8", FN(0, 8, 1024), wpc, ldsb);
        row("buffer_load_dwordx4 -> VGPR, 4", FN(1, 4, 1024), wpc, ldsb);
        row("buffer_load_dwordx4 -> VGPR, 8", FN(1, 8, 1024), wpc, ldsb);
    }
    return 0;
}
