// conv_pw.hip — pointwise (1x1, stride 1) convolution with the pixel tile RESIDENT in LDS, for short contractions (Cin <= 256).
//
// Call sites: the "expand" 1x1 convolutions of every Bottleneck — conv3 forward (fpn.py:18,30: 256 -> 1024 at 30x30 for the 23
// layer-3 blocks, 64 -> 256, 128 -> 512) and the input gradient of conv1 (fpn.py:14,28: dY has 256 channels, dX 1024, accumulated
// onto the shortcut's gradient, with the BatchNorm-backward statistics of the block below in the epilogue).  Both have a SHORT
// contraction (2 - 8 k-steps of 32) and a WIDE output, which is the worst case for the generic implicit-GEMM tile (conv_igemm.hip):
// every 128x128 output tile is its own workgroup with a DMA prologue, a handful of k-steps and a 32 KB store burst, each pixel tile
// is re-fetched by Cout/128 workgroups and each weight tile by every pixel tile (2.0 TB/s, 18.6 % of MFMA at 30x30).
//
// Here ONE workgroup owns a 128-pixel tile for ALL output channels:
//   * the tile's whole K extent (128 px x Cin x 2 B <= 64 KB) lands in LDS once (LDS-DMA, source-side XOR swizzle) and is read-only
//     afterwards: no ring, no barrier in the main loop;
//   * each of the 8 (4) waves walks its own 64-output-channel strips: A fragments (weights, L2-resident) come straight from global
//     memory into registers one k-step ahead, B fragments from the resident tile, 32 MFMAs per k-step (wave tile 64 ch x 128 px);
//   * the waves are independent after the prologue, so they drift apart: one wave's epilogue (loads of residual / accumulate /
//     BatchNorm operands, stores, statistics) runs under the other waves' MFMAs instead of in a workgroup-wide burst;
//   * the rows of the A fragments are permuted (fragment pair (2m, 2m+1), row 4g+c <-> channel m*32 + g*8 + (i&1)*4 + c) so that a
//     lane's accumulators of a fragment pair are 8 CONSECUTIVE output channels of one pixel: results leave as 16-byte stores straight
//     from registers (64 contiguous bytes per pixel and instruction), no LDS staging;
//   * a wave covers all 128 pixels of the tile for its channels, so the per-tile BatchNorm partials (forward sum / sum^2, backward
//     sum g / sum g*xhat) need one DPP row reduction and no cross-wave step.
// Traffic per launch (256 -> 1024 at 28 800 pixels): pixels once (14.7 MB), weights 225 x 512 KB through L2 (115 MB; 230 MB before),
// output once (59 MB).
#include "common.h"
#include <stdlib.h>

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef int i32x4_t __attribute__((ext_vector_type(4)));

template <typename T> struct Mma16;
template <> struct Mma16<bf16_t> {
    __device__ static __forceinline__ void run(f32x4_t& acc, const u32x4_t& a, const u32x4_t& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
    }
};
template <> struct Mma16<f16_t> {
    __device__ static __forceinline__ void run(f32x4_t& acc, const u32x4_t& a, const u32x4_t& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), acc, 0, 0, 0);
    }
};

__device__ __forceinline__ i32x4_t pw_rsrc(const void* base, unsigned bytes) {
    const uint64_t a = (uint64_t)base;
    i32x4_t r;
    r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));
    r.z = __builtin_amdgcn_readfirstlane((int)bytes);
    r.w = 0x00020000;
    return r;
}

// buffer_load_dwordx4 ... lds: 16 bytes per lane to LDS at M0 + lane*16; lanes whose offset is past num_records deliver zeros
__device__ __forceinline__ void pw_dma16(unsigned voff, i32x4_t rsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_dst) : "memory");
}

__device__ __forceinline__ float pw_row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    return v;
}

constexpr int PW_TP = 128;             // pixels per workgroup (== the statistics tile of mpn_conv_stats_tiles)
constexpr int PW_LDS = PW_TP * 256 * 2;

// NW waves.  EPI selects the epilogue, each with its own (small) live register set beside the 128 accumulators:
//   0  conv (+ forward BatchNorm tile statistics): the training forward of conv3;
//   1  per-channel scale / bias, ReLU, same-size residual, ReLU after it: conv3 with folded BatchNorm (inference), biased 1x1 layers;
//   2  accumulate onto the existing output AND BatchNorm-backward tile statistics with the ReLU mask taken from z: the training input
//      gradient of conv1 (the block below ends in relu(bn3(.) + shortcut)).
template <typename T, int NW, int EPI>
__global__ void __launch_bounds__(NW * 64, 2) conv_pw_kernel(const MpnConvParams p, const int dbg, unsigned long long* __restrict__ stamps) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[PW_LDS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tp = blockIdx.x;
    // tools/pw_timeline.py: s_memtime stamps per wave at the phase boundaries (null in production launches)
    unsigned long long* my_stamps = stamps ? stamps + ((long)(blockIdx.y * gridDim.x + blockIdx.x) * NW + wave) * 8 : nullptr;
    int stamp_i = 0;
    auto stamp = [&]() { if (my_stamps && lane == 0 && stamp_i < 8) my_stamps[stamp_i] = __builtin_readcyclecounter(); ++stamp_i; };
    stamp();
    const unsigned P = (unsigned)p.B * (unsigned)p.Ho * (unsigned)p.Wo;
    const unsigned p0 = (unsigned)tp * PW_TP;
    const int K = p.Cin;
    const int ppr = K >> 3;                                      // 16-byte pieces per pixel row (8 / 16 / 32)
    const int ppr_log = 31 - __builtin_clz((unsigned)ppr);
    const int key_mask = ppr > 16 ? 15 : ppr - 1;
    const unsigned row_bytes = (unsigned)K * 2u;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;

    // ---- the pixel tile: 128 rows x K, piece (row, q) stored at slot q ^ (row & key_mask) ------------------------------------
    {
        const unsigned x_bytes = (unsigned)((long)P * p.x_sW * 2);
        const i32x4_t rsrc_x = pw_rsrc(p.x, x_bytes);
        const int ninstr = 2 * ppr;                              // 1 KiB (64 pieces) per DMA instruction
        for (int q = wave; q < ninstr; q += NW) {
            const int L = q * 64 + lane;
            const int row = L >> ppr_log, slot = L & (ppr - 1);
            const unsigned pix = p0 + (unsigned)row;
            const unsigned voff = pix < P ? (unsigned)((long)pix * p.x_sW * 2) + (unsigned)((slot ^ (row & key_mask)) << 4) : x_bytes;
            pw_dma16(voff, rsrc_x, (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_base + (unsigned)q * 1024u)));
        }
    }

    // ---- per-wave strips of 64 output channels -------------------------------------------------------------------------------
    const T* __restrict__ Wt = (const T*)p.w;
    const int r16 = lane & 15, kg = lane >> 4;
    const int G = kg;                                           // accumulator row group of this lane
    // A fragment i, lane row r16 = 4g + c  <->  channel  (i >> 1) * 32 + g * 8 + (i & 1) * 4 + c  of the strip
    int a_row[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) a_row[i] = ((i >> 1) * 32 + (r16 >> 2) * 8 + (i & 1) * 4 + (r16 & 3)) * K;
    const int nstrips = p.Cout >> 6;
    const int ksteps = K >> 5;
    const int key = r16 & key_mask;
    unsigned b_row[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) b_row[j] = (unsigned)(j * 16 + r16) * row_bytes;

    auto load_a = [&](u32x4_t (&a)[4], int strip, int ks) {
        const T* base = Wt + (long)(strip * 64) * K + ks * 32 + kg * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const u32x4_t*>(base + a_row[i]);
    };

    u32x4_t a0[4], a1[4];
    // blockIdx.y selects a group of strips: with gridDim.y = nstrips / NW every wave owns exactly ONE strip and never issues a load
    // after its stores (stores and loads share vmcnt: a load behind a store burst waits for the burst to drain)
    const int nstrips_wg = nstrips / (int)gridDim.y;
    int strip = blockIdx.y * nstrips_wg + wave;
    const int strip_end = (blockIdx.y + 1) * nstrips_wg;
    if (strip < strip_end) load_a(a0, strip, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                            // the pixel tile is complete and read-only from here on
    stamp();

    T* __restrict__ Y = (T*)p.y;
    // Waves w and w + NW/2 share a SIMD (a workgroup's waves are dealt to the SIMDs cyclically).  With equal priority both halves run
    // their main loops and then their store bursts at the same time; with the first half prioritised it takes the matrix pipe first,
    // and from then on one half computes while the other waits for its stores to drain.
    if (NW == 8 && (dbg & 64) && wave < NW / 2) __builtin_amdgcn_s_setprio(1);

    for (; strip < strip_end; strip += NW) {
        f32x4_t acc[4][8];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

        auto step = [&](const u32x4_t (&a)[4], int ks) {
            const unsigned poff = (unsigned)((((ks << 2) | kg) ^ key) << 4);
            u32x4_t b[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) b[j] = *reinterpret_cast<const u32x4_t*>(lds + b_row[j] + poff);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) Mma16<T>::run(acc[i][j], a[i], b[j]);
        };
        // Two k-steps per trip, fragments of the k-step after next always in flight.  The prefetch is UNCONDITIONAL (past the end of
        // the strip it fetches the first weights of the wave's next strip, or harmlessly re-fetches): a load that exists on one path
        // only makes the compiler's vmcnt bookkeeping assume the shorter queue, i.e. wait for the newest load before every k-step.
        const int nxt_strip = strip + NW < strip_end ? strip + NW : strip;
        for (int ks = 0; ks < ksteps; ks += 2) {                // ksteps is even (Cin = 64 / 128 / 256)
            load_a(a1, strip, ks + 1);
            step(a0, ks);
            const bool more = ks + 2 < ksteps;
            load_a(a0, more ? strip : nxt_strip, more ? ks + 2 : 0);
            step(a1, ks + 1);
        }
        stamp();
        const int cb = strip * 64;

        // ---- epilogue of the strip: lane = 8 consecutive channels (cb + m*32 + G*8 ..) of pixel p0 + j*16 + r16 ----------------
        // The accumulators are first rounded to the element type and packed (64 registers instead of 128; the generic kernel rounds
        // at the same point: its staging tile).  Loads of the loaded variants then go out in batches of 8 pixel fragments: stores and
        // loads share vmcnt, so every load issued behind a store waits for that store to drain — two such waits per strip, not sixteen.
        u32x4_t pk[2][8];
        {
            const float relu_lo = (EPI == 1 && p.act == 1) ? 0.f : -INFINITY;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int co = cb + m * 32 + G * 8;
                float sc[8], bs[8];
                if constexpr (EPI == 1) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { sc[e] = 1.f; bs[e] = 0.f; }
                    if (p.scale) {
                        const float4 t0 = *reinterpret_cast<const float4*>(p.scale + co), t1 = *reinterpret_cast<const float4*>(p.scale + co + 4);
                        sc[0] = t0.x; sc[1] = t0.y; sc[2] = t0.z; sc[3] = t0.w; sc[4] = t1.x; sc[5] = t1.y; sc[6] = t1.z; sc[7] = t1.w;
                    }
                    if (p.bias) {
                        const float4 t0 = *reinterpret_cast<const float4*>(p.bias + co), t1 = *reinterpret_cast<const float4*>(p.bias + co + 4);
                        bs[0] = t0.x; bs[1] = t0.y; bs[2] = t0.z; bs[3] = t0.w; bs[4] = t1.x; bs[5] = t1.y; bs[6] = t1.z; bs[7] = t1.w;
                    }
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    Vec16<T> o;
#pragma unroll
                    for (int c = 0; c < 4; ++c) { o.v[c] = acc[2 * m][j][c]; o.v[4 + c] = acc[2 * m + 1][j][c]; }
                    if constexpr (EPI == 1) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) o.v[e] = fmaxf(o.v[e] * sc[e] + bs[e], relu_lo);
                    }
                    o.store(reinterpret_cast<T*>(&pk[m][j]));
                }
            }
        }
        const T* __restrict__ Rz = (const T*)p.res;
        const T* __restrict__ Ybn = (const T*)p.bnb_y;
        const T* __restrict__ Zbn = (const T*)p.bnb_z;
        constexpr bool bnb = EPI == 2;
        float* part = EPI == 0 ? p.stats : (bnb ? p.bnb_partial : nullptr);
        // Stores go out pixel fragment by pixel fragment with the two 64-byte halves (m = 0, 1) of every 128-byte line back to back:
        // issued eight stores apart, the halves reached HBM as separate partial-line writes (+30 % write requests, PMC).
        constexpr int BJ = EPI == 0 ? 8 : (EPI == 1 ? 4 : 2);                    // pixel fragments per batch of epilogue loads
        const int co0 = cb + G * 8;                                             // + m * 32
        const long ystep = 16L * p.y_sP;
        const long yo0 = (long)(p0 + (unsigned)r16) * p.y_sP + co0;             // + j * ystep + m * 32
        float bmu[2][8], bis[2][8], s1[2][8], s2[2][8];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { s1[m][e] = 0.f; s2[m][e] = 0.f; bmu[m][e] = 0.f; bis[m][e] = 0.f; }
            if (bnb) {
                const int co = co0 + m * 32;
                const float4 t0 = *reinterpret_cast<const float4*>(p.bnb_mean + co), t1 = *reinterpret_cast<const float4*>(p.bnb_mean + co + 4);
                const float4 u0 = *reinterpret_cast<const float4*>(p.bnb_invstd + co), u1 = *reinterpret_cast<const float4*>(p.bnb_invstd + co + 4);
                bmu[m][0] = t0.x; bmu[m][1] = t0.y; bmu[m][2] = t0.z; bmu[m][3] = t0.w; bmu[m][4] = t1.x; bmu[m][5] = t1.y; bmu[m][6] = t1.z; bmu[m][7] = t1.w;
                bis[m][0] = u0.x; bis[m][1] = u0.y; bis[m][2] = u0.z; bis[m][3] = u0.w; bis[m][4] = u1.x; bis[m][5] = u1.y; bis[m][6] = u1.z; bis[m][7] = u1.w;
            }
        }
#pragma unroll
        for (int j0 = 0; j0 < 8; j0 += BJ) {
            asm volatile("" ::: "memory");                                       // keep the batches' loads apart (register budget)
            u32x4_t l_a[BJ][2], l_y[BJ][2], l_z[BJ][2];                          // residual or previous output / bnb_y / bnb_z
            // every load is issued on every path (dead pixels read the tile's first pixel): exact vmcnt bookkeeping
            if (EPI == 1 && p.res_mode == 1) {
#pragma unroll
                for (int g = 0; g < BJ; ++g) {
                    const bool live = p0 + (unsigned)((j0 + g) * 16 + r16) < P;
                    const long ro = (live ? (long)(p0 + (unsigned)((j0 + g) * 16 + r16)) : (long)p0) * p.res_sP + co0;
#pragma unroll
                    for (int m = 0; m < 2; ++m) l_a[g][m] = *reinterpret_cast<const u32x4_t*>(Rz + ro + m * 32);
                }
            }
            if (EPI == 2) {
#pragma unroll
                for (int g = 0; g < BJ; ++g) {
                    const bool live = p0 + (unsigned)((j0 + g) * 16 + r16) < P;
                    const long yo = live ? yo0 + (j0 + g) * ystep : (long)p0 * p.y_sP + co0;
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        l_a[g][m] = *reinterpret_cast<const u32x4_t*>(Y + yo + m * 32);
                        l_y[g][m] = *reinterpret_cast<const u32x4_t*>(Ybn + yo + m * 32);
                        l_z[g][m] = *reinterpret_cast<const u32x4_t*>(Zbn + yo + m * 32);
                    }
                }
            }
#pragma unroll
            for (int g = 0; g < BJ; ++g) {
                const int j = j0 + g;
                const bool live = p0 + (unsigned)(j * 16 + r16) < P;
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    u32x4_t v = pk[m][j];
                    Vec16<T> o;
                    if ((EPI == 1 && p.res_mode == 1) || EPI == 2) {
                        Vec16<T> t;
                        o.load(reinterpret_cast<const T*>(&v));
                        t.load(reinterpret_cast<const T*>(&l_a[g][m]));
#pragma unroll
                        for (int e = 0; e < 8; ++e) o.v[e] += t.v[e];
                        if (EPI == 1 && p.act == 3) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) o.v[e] = fmaxf(o.v[e], 0.f);
                        }
                        o.store(reinterpret_cast<T*>(&v));
                    }
                    if (live) *reinterpret_cast<u32x4_t*>(Y + yo0 + j * ystep + m * 32) = v;
                    if (part && live) {
                        o.load(reinterpret_cast<const T*>(&v));                  // the value as stored
                        if (EPI == 0) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) { s1[m][e] += o.v[e]; s2[m][e] += o.v[e] * o.v[e]; }
                        } else {
                            Vec16<T> yy, zz;
                            yy.load(reinterpret_cast<const T*>(&l_y[g][m]));
                            zz.load(reinterpret_cast<const T*>(&l_z[g][m]));
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                float gg = o.v[e];
                                if (!(zz.v[e] > 0.f)) gg = 0.f;
                                s1[m][e] += gg;
                                s2[m][e] += gg * ((yy.v[e] - bmu[m][e]) * bis[m][e]);
                            }
                        }
                    }
                }
            }
        }
        stamp();
        if (part) {
#pragma unroll
            for (int m = 0; m < 2; ++m) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { s1[m][e] = pw_row16_sum(s1[m][e]); s2[m][e] = pw_row16_sum(s2[m][e]); }
                if (r16 == 0) {
                    float* dst = part + ((long)tp * p.Cout + co0 + m * 32) * 2;
#pragma unroll
                    for (int e = 0; e < 8; e += 2)
                        *reinterpret_cast<float4*>(dst + e * 2) = make_float4(s1[m][e], s2[m][e], s1[m][e + 1], s2[m][e + 1]);
                }
            }
        }
    }
}

int g_pw_min_tiles = -1;
unsigned long long* g_pw_stamps = nullptr;

inline int pw_min_tiles() {
    if (g_pw_min_tiles < 0) g_pw_min_tiles = getenv("MPN_PW_MIN_TILES") ? atoi(getenv("MPN_PW_MIN_TILES")) : 96;
    return g_pw_min_tiles;
}

inline bool pw_shape_ok(const MpnConvParams& p) {
    if (p.dtype == MPN_F32 || p.nseg != 0 || p.R != 1 || p.S != 1 || p.stride != 1 || p.pad != 0) return false;
    if (!(p.Cin == 64 || p.Cin == 128 || p.Cin == 256)) return false;
    if (p.Cout < 256 || (p.Cout & 63) || p.Cout_store != p.Cout || p.out_f32) return false;
    if (p.res_mode > 1 || p.act == 2 || p.fin_counters || p.res_mask || p.relu_y || p.y2 || p.kseg_n || p.stats_atomic) return false;
    const bool affine = p.scale || p.bias || p.act || p.res_mode;                 // epilogue 1
    const bool grad = p.accumulate || p.bnb_partial;                             // epilogue 2: accumulate AND statistics with the ReLU mask
    if (affine && grad) return false;                                            // read from z — the training gradient of conv1 (fpn.py:14,28);
    if (grad && !(p.accumulate && p.bnb_partial && p.bnb_relu && p.bnb_z)) return false;      // other mixes: generic kernel
    if (p.H != p.Ho || p.W != p.Wo) return false;
    const int64_t hw = (int64_t)p.Ho * p.Wo;
    if (p.x_sH != (int64_t)p.W * p.x_sW || p.x_sB != hw * p.x_sW || p.y_sB != hw * p.y_sP) return false;
    if (p.res_mode == 1 && p.res_sB != hw * p.res_sP) return false;
    if ((p.x_sW & 7) || (p.y_sP & 7) || (p.res_mode == 1 && (p.res_sP & 7))) return false;
    const int64_t P = (int64_t)p.B * hw;
    if (P * p.x_sW * 2 >= 0xfffffff0LL) return false;
    return true;
}

template <typename T>
int pw_launch(const MpnConvParams& p, hipStream_t st) {
    const long P = (long)p.B * p.Ho * p.Wo;
    const unsigned grid = (unsigned)((P + PW_TP - 1) / PW_TP);
    static const int dbg = getenv("MPN_PW_DEBUG") ? atoi(getenv("MPN_PW_DEBUG")) : 0;       // micro-benchmark ablations only
    const int epi = (p.accumulate || p.bnb_partial) ? 2 : ((p.scale || p.bias || p.act || p.res_mode) ? 1 : 0);
    // grid shape: 2 = one workgroup per pixel tile walks ALL strips (default; the others measured slower),
    // 0 = 4 waves x one strip each (gridDim.y = strips / 4), 1 = 8 waves x one strip each
    static const int cfg = getenv("MPN_PW_CFG") ? atoi(getenv("MPN_PW_CFG")) : 2;
    const int nstrips = p.Cout / 64;
    const int nw = (cfg == 0 || nstrips < 8) ? 4 : 8;
    const unsigned gy = cfg == 2 ? 1u : (unsigned)(nstrips / nw);
    const dim3 g(grid, gy);
    if (nw == 8) {
        if (epi == 2) hipLaunchKernelGGL((conv_pw_kernel<T, 8, 2>), g, dim3(512), 0, st, p, dbg, g_pw_stamps);
        else if (epi == 1) hipLaunchKernelGGL((conv_pw_kernel<T, 8, 1>), g, dim3(512), 0, st, p, dbg, g_pw_stamps);
        else hipLaunchKernelGGL((conv_pw_kernel<T, 8, 0>), g, dim3(512), 0, st, p, dbg, g_pw_stamps);
    } else {
        if (epi == 2) hipLaunchKernelGGL((conv_pw_kernel<T, 4, 2>), g, dim3(256), 0, st, p, dbg, g_pw_stamps);
        else if (epi == 1) hipLaunchKernelGGL((conv_pw_kernel<T, 4, 1>), g, dim3(256), 0, st, p, dbg, g_pw_stamps);
        else hipLaunchKernelGGL((conv_pw_kernel<T, 4, 0>), g, dim3(256), 0, st, p, dbg, g_pw_stamps);
    }
    return mpn_launch_status();
}

}  // namespace

// 1 when mpn_conv_forward will run this problem on conv_pw_kernel (shape served AND enough pixel tiles to fill the chip)
extern "C" int mpn_conv_pw_selected(const MpnConvParams* p) {
    if (!p || !pw_shape_ok(*p)) return 0;
    // Which epilogues mpn_conv_forward routes here by itself: bit 0 = plain / forward statistics, bit 1 = affine / residual,
    // bit 2 = accumulate + backward statistics.  Default 0 — NONE: measured in isolation the kernel is 1.15 - 1.22x the generic one
    // on the plain epilogue and at parity on the others, and inside the training step it buys nothing (39.5 vs 39.6 ms, DESIGN.md
    // section 5); MPN_PW_EPI_MASK=1..7 turns classes on, mpn_conv_pw_set_min_tiles(0) (the parity tests) routes everything served.
    static const int mask = getenv("MPN_PW_EPI_MASK") ? atoi(getenv("MPN_PW_EPI_MASK")) : 0;
    const int epi = (p->accumulate || p->bnb_partial) ? 2 : ((p->scale || p->bias || p->act || p->res_mode) ? 1 : 0);
    if (pw_min_tiles() > 0 && !((mask >> epi) & 1)) return 0;
    const long P = (long)p->B * p->Ho * p->Wo;
    return (P + PW_TP - 1) / PW_TP >= pw_min_tiles() ? 1 : 0;
}

extern "C" int mpn_conv_pw_set_min_tiles(int tiles) {
    const int old = pw_min_tiles();
    if (tiles >= 0) g_pw_min_tiles = tiles;
    return old;
}

// tools only: device buffer of [workgroups][waves][8] s_memtime stamps written by the next launches (nullptr = off)
extern "C" int mpn_conv_pw_debug_stamps(void* buf) { g_pw_stamps = (unsigned long long*)buf; return 0; }

extern "C" int mpn_conv_pw_supported(const MpnConvParams* p) { return p && pw_shape_ok(*p) ? 1 : 0; }

// launches conv_pw_kernel for a problem mpn_conv_pw_supported() accepts (whatever its size); mpn_conv_forward routes here by itself
extern "C" int mpn_conv_pw_forward(const MpnConvParams* pp, void* stream) {
    if (!pp) return MPN_E_BADARG;
    if (!pw_shape_ok(*pp)) return MPN_E_UNSUPPORTED;
    MPN_CHECK_ARG(pp->x && pp->w && pp->y && pp->B > 0 && pp->Ho > 0 && pp->Wo > 0);
    MPN_CHECK_ARG(!(pp->accumulate && pp->act != 0) && (pp->act != 3 || pp->res_mode == 1) && (pp->res_mode == 0 || pp->res));
    MPN_CHECK_ARG(!(pp->stats && (pp->bias || pp->scale || pp->res_mode || pp->accumulate || pp->act || pp->bnb_partial)));
    MPN_CHECK_ARG(!pp->bnb_partial || (pp->bnb_y && pp->bnb_mean && pp->bnb_invstd && !pp->act &&
                                       (!pp->bnb_relu || pp->bnb_z || (pp->bnb_scale && pp->bnb_shift))));
    if (pp->dtype == MPN_F16) return pw_launch<f16_t>(*pp, (hipStream_t)stream);
    return pw_launch<bf16_t>(*pp, (hipStream_t)stream);
}
