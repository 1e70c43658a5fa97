// conv_igemm.hip — im2col-free implicit-GEMM convolution on MFMA for gfx950 (forward + dgrad).
//
// Replaces the nn.Conv2d / nn.Linear forward call sites of network/fpn.py:14-26,42-76 and
// network/posenet.py:36-46,78-89,133-135,165-186, and the input-gradient half of their autograd
// backward (training/trainer.py:251).
//
// GEMM view:  D[cout][pixel] = sum_{r,s,c} W[cout][r][s][c] * X[gather(pixel, r, s)][c]
//   * MFMA "A" operand rows  = output channels (weights, K-contiguous [Cout][R][S][Cin])
//   * MFMA "B" operand cols  = output pixels   (NHWC activations: Cin contiguous per pixel/tap)
//   so each lane ends up with 4 CONSECUTIVE output channels of one pixel -> 8/16-byte NHWC stores.
//   * K is walked in 64-byte chunks (32 bf16 / 16 f32 channels of one tap); a chunk never straddles
//     a tap, so the gather is one predicated 16-byte load per lane and halo pixels are zero-filled
//     in registers (no im2col buffer, no padded copy of the activations).
//   * 256 threads = 4 waves; block tile TC x 128 pixels; tiles travel HBM -> LDS by DMA (buffer_load ... lds)
//     into a 3-deep ring with hand-counted vmcnt; 64-byte rows, 16-byte pieces XOR-swizzled against bank conflicts.
//   * bf16: v_mfma_f32_16x16x32_bf16 (fp32 accumulate); f32: v_mfma_f32_16x16x4_f32 (exact fp32,
//     bit-identical to an fmaf chain) — the f32 instantiation is the parity path.
//   * epilogue fuses per-channel scale, bias, residual (same-size or nearest-upsampled: the FPN
//     top-down add of fpn.py:84-95), ReLU / sigmoid, zero-fill of pad channels, and the per-tile
//     (sum, sum^2) partials BatchNorm needs in train mode.
// dgrad reuses the kernel with mode=1 (transposed gather hi = (ho + pad - r)/stride) and the
// [Cin][R][S][Cout_pad] weight copy made by mpn_weight_transpose.
// 3x3 / stride 1 / pad 1 launches with 16-bit operands take conv_igemm_s3_kernel (below): same tile, ring and epilogue, but the pixel
// tile of a kernel row lands once for its three taps (the k-loop is bound by operand delivery, not by MFMA issue).
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;


template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    __device__ static __forceinline__ void run(f32x4_t& acc, const u32x4_t& a, const u32x4_t& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                      __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
    }
};
template <> struct Mma<f16_t> {
    __device__ static __forceinline__ void run(f32x4_t& acc, const u32x4_t& a, const u32x4_t& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), acc, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    __device__ static __forceinline__ void run(f32x4_t& acc, const u32x4_t& a, const u32x4_t& b) {
        const f32x4_t fa = __builtin_bit_cast(f32x4_t, a);
        const f32x4_t fb = __builtin_bit_cast(f32x4_t, b);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[0], fb[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[1], fb[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[2], fb[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[3], fb[3], acc, 0, 0, 0);
    }
};

template <typename OT> struct OutVec4;
template <> struct OutVec4<float> {
    __device__ static __forceinline__ void load(const float* p, float v[4]) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    __device__ static __forceinline__ void store(float* p, const float v[4]) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    }
    __device__ static __forceinline__ float round(float v) { return v; }
};
template <> struct OutVec4<bf16_t> {
    __device__ static __forceinline__ void load(const bf16_t* p, float v[4]) {
        const uint2 t = *reinterpret_cast<const uint2*>(p);
        v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
        v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
    }
    __device__ static __forceinline__ void store(bf16_t* p, const float v[4]) {
        uint2 t;
        t.x = pack_bf16x2(v[0], v[1]);
        t.y = pack_bf16x2(v[2], v[3]);
        *reinterpret_cast<uint2*>(p) = t;
    }
    __device__ static __forceinline__ float round(float v) { return bf2f(f2bf(v)); }
};

template <> struct OutVec4<f16_t> {
    typedef __attribute__((ext_vector_type(4))) _Float16 f16x4_t;
    __device__ static __forceinline__ void load(const f16_t* p, float v[4]) {
        const f16x4_t t = *reinterpret_cast<const f16x4_t*>(p);
        v[0] = (float)t[0]; v[1] = (float)t[1]; v[2] = (float)t[2]; v[3] = (float)t[3];
    }
    __device__ static __forceinline__ void store(f16_t* p, const float v[4]) {
        f16x4_t t;
        t[0] = (f16_t)v[0]; t[1] = (f16_t)v[1]; t[2] = (f16_t)v[2]; t[3] = (f16_t)v[3];
        *reinterpret_cast<f16x4_t*>(p) = t;
    }
    __device__ static __forceinline__ float round(float v) { return (float)(f16_t)v; }
};

template <typename T, int TC, int TP>
struct ConvCfg {
    static constexpr int KC = 64 / (int)sizeof(T);
    static constexpr int V = 16 / (int)sizeof(T);
    static constexpr int WAVES_C = (TC >= 64) ? 2 : 1;
    static constexpr int WAVES_P = 4 / WAVES_C;
    static constexpr int WTC = TC / WAVES_C;     // wave tile, channels
    static constexpr int WTP = TP / WAVES_P;     // wave tile, pixels
    static constexpr int MC = WTC / 16;
    static constexpr int MP = WTP / 16;
    // LDS ring: per stage an A tile (TA_ROWS couts x 64 B) and a B tile (TP pixels x 64 B), rows back to back
    static constexpr int TA_ROWS = TC < 64 ? 64 : TC;        // one 1-KiB DMA instruction per wave at least
    static constexpr int A_PER_W = TA_ROWS / 64;             // DMA instructions per wave per k-step
    static constexpr int B_PER_W = TP / 64;
    static constexpr int LPS = A_PER_W + B_PER_W;
    static constexpr int A_BYTES = TA_ROWS * 64;
    static constexpr int STAGE_BYTES = (TA_ROWS + TP) * 64;
    static constexpr int NST = 3;
};

// Epilogue.  Phase A (accumulator layout: lane = 4 consecutive couts of one pixel): scale, bias, residual,
// accumulate, activation, BN partial sums.  Phase B: the finished tile goes through a per-wave LDS staging
// area and leaves as 16-byte stores with 8..16 lanes covering one pixel's contiguous channels (full 128-byte
// lines per wave-instruction) instead of 16 strided 8-byte stores per lane.
// sum over the 16 lanes of a DPP row (lanes sharing lane>>4); every lane ends with the total.
// quad_perm xor-1 / xor-2 butterflies, then row_half_mirror and row_mirror (cdna4 DPP controls).
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
    return v;
}

// A per-tile statistics pair.  When the launch finalizes in place (fin_counters) it is published write-through (one 8-byte
// agent-scope store: cdna_hip_programming.md, Guideline 16 recipe R1) so that the last-arriving workgroup — on any XCD — reads it
// after its acquire.
__device__ __forceinline__ void store_partial(float* dst, float a, float b, bool publish) {
    if (publish) {
        const unsigned long long bits = ((unsigned long long)__float_as_uint(b) << 32) | __float_as_uint(a);
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(dst), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        *reinterpret_cast<float2*>(dst) = make_float2(a, b);
    }
}

// In-launch finalize (mpn.h: fin_*).  Every workgroup of channel tile `tc` has published its partial pair per channel; the one that
// draws the last ticket reduces the column [ntiles][TC] in a fixed order (SL interleaved slices per channel in double precision,
// combined in slice order) — the same numbers whichever workgroup arrives last — and writes the BatchNorm coefficients.
// (A two-level form — groups of ~sqrt(tiles) pixel tiles reduced by their own last arrivers — and a finalize inside the bn_act launch
// were built in round 3, measured 0.2 - 0.3 ms/step SLOWER than the separate finalize launch for the larger layers, and removed in
// round 4: DESIGN.md section 5.)

// true for the workgroup that drew ticket `last` of `counter` (which it leaves at zero again)
__device__ __forceinline__ bool fin_ticket(unsigned* counter, unsigned last, volatile int* flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // every storing wave drains its write-through stores
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *flag = (old == last) ? 1 : 0;
        if (old == last) {
            __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
    }
    __syncthreads();
    return *flag != 0;
}

template <int TC>
__device__ __forceinline__ void fin_last_arriver(const MpnConvParams& p, int c0, int tc, int tp, int ntiles, unsigned char* lds) {
    // Round 5: the column block [ntiles][TC] is read as 16-byte pieces (two channels' (sum, sum^2)) by 256 / (TC / 2) row slices, eight
    // loads in flight per thread — half the dependent round trips of the 8-byte / (256 / TC)-slice form it replaces (225 tiles at
    // TC = 128: 8 instead of 15), which is the whole cost of this tail.  Fixed order (slice-sequential rows, slices combined in order):
    // the same numbers whichever workgroup arrives last.
    constexpr int QL = TC / 2, SL = 256 / QL;
    volatile int* flag = reinterpret_cast<volatile int*>(lds + 12288);
    double* lds_d = reinterpret_cast<double*>(lds);            // [SL][TC][2] = 8 KB
    const int t = threadIdx.x;
    const float* __restrict__ part = p.stats ? p.stats : p.bnb_partial;
    const int cl = t % TC, sl0 = t / TC;                       // finishing role: one thread per channel (sl0 == 0)
    const int c = c0 + cl;
    double s1 = 0.0, s2 = 0.0;
    if (!fin_ticket(p.fin_counters + tc, (unsigned)(ntiles - 1), flag)) return;
    {
        const int q = t % QL, sl = t / QL;
        const bool on = c0 + 2 * q + 1 < p.Cout;                // Cout is even (launcher): the pair exists or does not
        const float4* __restrict__ col = reinterpret_cast<const float4*>(part + (long)c0 * 2) + q;
        const long rs = (long)p.Cout / 2;                       // float4 per table row (launcher: in-launch finalize needs an even Cout)
        double a1 = 0.0, a2 = 0.0, b1 = 0.0, b2 = 0.0;
        if (on) {
            for (int r0 = sl; r0 < ntiles; r0 += 8 * SL) {
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int r = r0 + u * SL;
                    v[u] = r < ntiles ? col[(long)r * rs] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) { a1 += (double)v[u].x; a2 += (double)v[u].y; b1 += (double)v[u].z; b2 += (double)v[u].w; }
            }
        }
        lds_d[(sl * TC + 2 * q) * 2 + 0] = a1; lds_d[(sl * TC + 2 * q) * 2 + 1] = a2;
        lds_d[(sl * TC + 2 * q + 1) * 2 + 0] = b1; lds_d[(sl * TC + 2 * q + 1) * 2 + 1] = b2;
        __syncthreads();
        if (t < TC) {
#pragma unroll
            for (int k = 0; k < SL; ++k) { s1 += lds_d[(k * TC + cl) * 2 + 0]; s2 += lds_d[(k * TC + cl) * 2 + 1]; }
        }
    }
    const int sl = sl0;
    if (sl == 0 && c < p.Cout) {
        const int C = p.Cout;
        const float gm = p.fin_gamma ? p.fin_gamma[c] : 1.f;
        if (p.stats) {                                          // mpn_bn_finalize_train's arithmetic
            const double mu = s1 / p.fin_count;
            double var = s2 / p.fin_count - mu * mu;
            if (var < 0.0) var = 0.0;
            const float is = (float)(1.0 / sqrt(var + (double)p.fin_eps));
            const float b = p.fin_beta ? p.fin_beta[c] : 0.f;
            const float sc = gm * is;
            p.fin_out[c] = (float)mu; p.fin_out[C + c] = is; p.fin_out[2 * C + c] = sc; p.fin_out[3 * C + c] = b - (float)mu * sc;
            if (p.fin_rm) p.fin_rm[c] = (1.f - p.fin_momentum) * p.fin_rm[c] + p.fin_momentum * (float)mu;
            if (p.fin_rv) {
                const double unb = p.fin_count > 1.0 ? var * p.fin_count / (p.fin_count - 1.0) : var;
                p.fin_rv[c] = (1.f - p.fin_momentum) * p.fin_rv[c] + p.fin_momentum * (float)unb;
            }
        } else {                                                // mpn_bn_bwd_finalize's arithmetic
            if (p.fin_dbeta) p.fin_dbeta[c] += (float)s1;
            if (p.fin_dgamma) p.fin_dgamma[c] += (float)s2;
            if (p.fin_out) {
                const float is = p.bnb_invstd[c], mu = p.bnb_mean[c];
                const float a = (float)(s1 / p.fin_count), b = (float)(s2 / p.fin_count);
                p.fin_out[c] = gm * is;
                p.fin_out[C + c] = p.fin_train ? -gm * is * is * b : 0.f;
                p.fin_out[2 * C + c] = p.fin_train ? gm * is * (mu * is * b - a) : 0.f;
            }
        }
    }
}

// Epilogue.  Phase A (accumulator layout: lane = 4 consecutive couts of one pixel): scale, bias, residual,
// accumulate, activation — branch-free per element (uniform conditions only); skipped entirely for the plain
// conv+BN-stats case.  BN partial sums use DPP row reductions.  Phase B: the finished tile goes through a
// per-wave LDS staging area and leaves as 16-byte stores with 8..16 lanes covering one pixel's contiguous
// channels (full 128-byte lines per wave-instruction).
// EXT (general kernels only) compiles in the rarely used epilogue features — both a residual AND accumulate, BatchNorm-backward
// statistics with the mask read from the z tensor, the virtual-concatenation gather (kseg).  The standard general
// instantiation serves everything the training / inference steps launch by default with ONE added tensor (residual, optionally
// through mask bits, OR the previous output) and mask bits / recomputation for the statistics: with all features compiled into
// one kernel the 128-row tile needed 80 - 97 spilled registers at its three-waves-per-SIMD budget (+3.3 ms/step, r03 trace).
template <typename T, typename OT, int TC, int TP, bool GENERAL, bool EXT, bool YSTEP = false>      // YSTEP: parity-class output (conv_igemm_kernel's extended instantiation only)
__device__ __forceinline__ void conv_epilogue(const MpnConvParams& p, f32x4_t (&acc)[ConvCfg<T, TC, TP>::MC][ConvCfg<T, TC, TP>::MP],
                                              int c0, long p0, int wc, int wp, int lane, int tp, unsigned char* lds, const int dbg,
                                              int tc, int ntiles, const MpnConvParams& pk) {
    // bnb_* / fin_* are read from the kernel-argument struct `pk` (late scalar loads), never from the patched working copy `p`:
    // fields the copy never reads are not kept in scalar registers through the main loop
    using C = ConvCfg<T, TC, TP>;
    constexpr int OSZ = (int)sizeof(OT);
    constexpr int PASS_TILES = (OSZ == 4) ? ((C::MP >= 2) ? C::MP / 2 : 1) : C::MP;   // pixel tiles staged per pass
    constexpr int NPASS = C::MP / PASS_TILES;
    constexpr int SROW = C::WTC * OSZ + 16;                   // staging row stride (bytes)
    constexpr int REGION = PASS_TILES * 16 * SROW;            // per wave
    constexpr int LPP = C::WTC * OSZ / 16;                    // lanes per pixel in the store phase
    constexpr int PPI = 64 / LPP;                             // pixels per store instruction
    constexpr int EV = 16 / OSZ;                              // elements per 16-byte store
    static_assert(4 * REGION + C::WAVES_P * TC * 8 <= C::NST * C::STAGE_BYTES, "staging area + statistics scratch must fit the main-loop LDS");
    const unsigned HoWo = (unsigned)p.Ho * (unsigned)p.Wo;
    const unsigned P = (unsigned)p.B * HoWo;           // launcher guarantees P < 2^31
    OT* __restrict__ Y = (OT*)p.y;
    const OT* __restrict__ Rz = (const OT*)p.res;
    const int lrow4 = (lane >> 4) * 4;
    const int lcol = lane & 15;
    float* lds_f = reinterpret_cast<float*>(lds);

    // ---- phase A (general kernels only): per-channel scale / bias / activation in registers ----------
    if (GENERAL) {
        const float relu_lo = (p.act == 1) ? 0.f : -INFINITY;
#pragma unroll
        for (int i = 0; i < C::MC; ++i) {
            const int cout0 = c0 + wc * C::WTC + i * 16 + lrow4;
            float sc[4] = {1.f, 1.f, 1.f, 1.f}, bs[4] = {0.f, 0.f, 0.f, 0.f}, lm[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) lm[c] = (cout0 + c) < p.Cout ? 1.f : 0.f;
            if (cout0 + 4 <= p.Cout) {                                   // whole group in range: one float4 per array
                if (p.scale) { const float4 t = *reinterpret_cast<const float4*>(p.scale + cout0); sc[0] = t.x; sc[1] = t.y; sc[2] = t.z; sc[3] = t.w; }
                if (p.bias) { const float4 t = *reinterpret_cast<const float4*>(p.bias + cout0); bs[0] = t.x; bs[1] = t.y; bs[2] = t.z; bs[3] = t.w; }
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int ci = (cout0 + c) < p.Cout ? cout0 + c : p.Cout - 1;
                    if (p.scale) sc[c] = p.scale[ci];
                    if (p.bias) bs[c] = p.bias[ci];
                }
            }
#pragma unroll
            for (int j = 0; j < C::MP; ++j)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float x = acc[i][j][c] * sc[c] + bs[c];
                    x = fmaxf(x, relu_lo);
                    acc[i][j][c] = x * lm[c];
                }
        }
        if (p.act == 2) {
#pragma unroll
            for (int i = 0; i < C::MC; ++i)
#pragma unroll
                for (int j = 0; j < C::MP; ++j)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float lv = (c0 + wc * C::WTC + i * 16 + lrow4 + c) < p.Cout ? 1.f : 0.f;
                        acc[i][j][c] = lv / (1.0f + expf(-acc[i][j][c]));
                    }
        }
    }

    // ---- BatchNorm tile statistics: per-channel (sum, sum^2) of the stored values: over this lane's MP pixels, then over the 16
    // pixel lanes of the DPP row, then over the WAVES_P waves via LDS scratch `sf`.  The plain kernels run this AFTER the stores were
    // issued (the reductions overlap the tile's drain to HBM; scratch behind the staging area, so no barrier against its readers);
    // the general kernels before them (the accumulators must be dead while the grouped epilogue loads hold their registers).
    auto tile_stats = [&](float* sf) {
        float pm[C::MP];
#pragma unroll
        for (int j = 0; j < C::MP; ++j) pm[j] = (((unsigned)p0 + wp * C::WTP + j * 16 + lcol) < P) ? 1.f : 0.f;
#pragma unroll
        for (int i = 0; i < C::MC; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float a = 0.f, q = 0.f;
#pragma unroll
                for (int j = 0; j < C::MP; ++j) {
                    const float xr = OutVec4<OT>::round(acc[i][j][c]) * pm[j];
                    a += xr; q += xr * xr;
                }
                a = row16_sum(a);
                q = row16_sum(q);
                if (lcol == 0) {
                    const int row = wc * C::WTC + i * 16 + lrow4 + c;      // 0..TC-1
                    *reinterpret_cast<float2*>(sf + (wp * TC + row) * 2) = make_float2(a, q);
                }
            }
        __syncthreads();
        const int t = threadIdx.x;
        if (t < TC) {
            const int cout = c0 + t;
            if (cout < p.Cout) {
                float a = 0.f, q = 0.f;
#pragma unroll
                for (int w = 0; w < C::WAVES_P; ++w) {
                    a += sf[(w * TC + t) * 2 + 0];
                    q += sf[(w * TC + t) * 2 + 1];
                }
                store_partial(p.stats + ((long)tp * p.Cout + cout) * 2, a, q, pk.fin_counters != nullptr);
            }
        }
    };
    constexpr bool stats_first = GENERAL;
    if (stats_first && p.stats && !MPN_DBG(64)) {
        tile_stats(lds_f);
        __syncthreads();
    }

    // ---- phase B: stage through LDS, store whole rows ----------------------------------------------
    if (MPN_DBG(256)) { if (acc[0][0][0] == 123.456f) ((float*)p.y)[0] = 1.f; return; }
    unsigned char* stage = lds + (threadIdx.x >> 6) * REGION;
    const int sp = lane / LPP, sc_ = lane % LPP;               // store phase: pixel slot, 16-byte chunk
    const int ccol = c0 + wc * C::WTC + sc_ * EV;              // first channel of this lane's chunk
    // BatchNorm-backward statistics of the tensor being completed (mpn.h: bnb_*): per-lane sums over the lane's pixels of
    // its EV channels, reduced over the tile after the store loop
    const bool bnb = GENERAL && pk.bnb_partial != nullptr;
    const OT* __restrict__ Ybn = (const OT*)pk.bnb_y;
    const unsigned char* __restrict__ Mbn = pk.bnb_mask;         // sign bits of z (one byte per 16-byte chunk) instead of z itself
    const OT* __restrict__ Zbn = (EXT && !Mbn) ? (const OT*)pk.bnb_z : nullptr;
    const int mrow = p.Cout_store / EV;                           // mask bytes per pixel
    float bs1[EV], bs2[EV], bmu[EV], bis[EV], bsc[EV], bsf[EV];
    if (bnb) {
#pragma unroll
        for (int e = 0; e < EV; ++e) {
            const int c = ccol + e;
            const bool live = c < p.Cout;
            bs1[e] = 0.f; bs2[e] = 0.f;
            bmu[e] = live ? pk.bnb_mean[c] : 0.f;
            bis[e] = live ? pk.bnb_invstd[c] : 0.f;
            bsc[e] = (live && pk.bnb_scale) ? pk.bnb_scale[c] : 0.f;
            bsf[e] = (live && pk.bnb_shift) ? pk.bnb_shift[c] : 0.f;
        }
    }
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
#pragma unroll
        for (int jj = 0; jj < PASS_TILES; ++jj) {
            const int j = ps * PASS_TILES + jj;
#pragma unroll
            for (int i = 0; i < C::MC; ++i) {
                const float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                OutVec4<OT>::store(reinterpret_cast<OT*>(stage + (jj * 16 + lcol) * SROW + (i * 16 + lrow4) * OSZ), v);
            }
        }
        __syncthreads();
        // first pixel handled by this lane in this pass, then advance PPI pixels per store
        unsigned pix = (unsigned)p0 + wp * C::WTP + ps * PASS_TILES * 16 + sp;
        unsigned b = (pix < P ? pix : 0u) / HoWo;
        unsigned rem = (pix < P ? pix : 0u) - b * HoWo;
        // The store loop runs in groups of G iterations: every global load of a group (residual, accumulate, the BatchNorm operands)
        // is in flight before the first result of the group is needed — one memory round trip per group instead of one per
        // iteration (the accumulators are dead by now, their registers hold the loads).  Groups of 8 (round 4; 4 before): the loaded
        // epilogues are HBM-latency bound — the phase profile shows 54 k cycles of epilogue against a 17 k-cycle k-loop in the conv1
        // input-gradient launch that adds dz through the mask bits and collects the BatchNorm statistics (profiles/
        // r04_kloop_phase_profile.txt) — so twice the loads in flight per wave: that launch 102.6 -> 96.6 us, step 37.93 -> 37.55 ms,
        // and the 128-row kernel needs FEWER registers (one group = the whole pass: 152 instead of 166 VGPRs).
        constexpr int NIT = PASS_TILES * 16 / PPI;
        // the extended epilogue carries three more load arrays per group: 8 would spill (70 VGPRs).  (Round 5: groups of 4 for the f32-output
        // instantiations — 8 - 10 spilled VGPRs in <float,128,128,general> / <*,128,128,OUTF32,general> — made it 78: a group that is
        // the whole pass compiles to the simplest loop; groups of 2 for the 256-row extended tile changed nothing, its spills are in the prologue.)
        constexpr int GMAX = EXT ? 4 : 8;
        constexpr int G = NIT < GMAX ? NIT : GMAX;
        static_assert(NIT % G == 0, "store groups");
#pragma unroll
        for (int k0 = 0; k0 < NIT; k0 += G) {
            bool live[G];
            unsigned yo[G];                                     // element offsets (launcher: output-shaped tensors < 2^31 elements)
            constexpr int GX = EXT ? G : 1;                      // arrays of the extended features (dead when !EXT)
            u32x4_t l_add[G], l_y[G];                            // the added tensor (residual, or the previous output) / bnb_y
            unsigned l_m[G], l_rm[G];                            // mask bytes: bnb_mask / res_mask
            u32x4_t l_acc[GX], l_z[GX];
            const bool add_acc = !EXT && p.res_mode == 0 && p.accumulate;      // standard kernel: accumulate rides in l_add
#pragma unroll
            for (int g = 0; g < G; ++g) {
                live[g] = pix < P && ccol < p.Cout_store && !MPN_DBG(128);
                yo[g] = b * (unsigned)p.y_sB + rem * (unsigned)p.y_sP + (unsigned)ccol;
                unsigned mpix = pix;                             // pixel index into the output-shaped mask tensors
                if (YSTEP && pk.y_step > 1) {                      // parity-class launch (mpn.h): output pixel (i, j) lives at (step i + oh, step j + ow)
                    const unsigned ho = rem / (unsigned)p.Wo, wo = rem - ho * (unsigned)p.Wo;
                    mpix = (b * (unsigned)pk.y_H + ho * (unsigned)pk.y_step + (unsigned)pk.y_oh) * (unsigned)pk.y_W + wo * (unsigned)pk.y_step + (unsigned)pk.y_ow;
                    yo[g] = mpix * (unsigned)p.y_sP + (unsigned)ccol;
                }
                if (GENERAL && live[g]) {
                    if (p.res_mode != 0) {
                        long ro;
                        if (p.res_mode == 1) {
                            ro = (long)b * p.res_sB + (long)rem * p.res_sP;
                        } else {
                            const unsigned ho = rem / (unsigned)p.Wo, wo = rem - ho * (unsigned)p.Wo;
                            const unsigned rh = (ho * (unsigned)p.res_H) / (unsigned)p.Ho, rw = (wo * (unsigned)p.res_W) / (unsigned)p.Wo;
                            ro = (long)b * p.res_sB + (long)(rh * (unsigned)p.res_W + rw) * p.res_sP;
                        }
                        l_add[g] = *reinterpret_cast<const u32x4_t*>(Rz + ro + ccol);
                        if (pk.res_mask) l_rm[g] = pk.res_mask[mpix * (unsigned)mrow + (unsigned)(ccol / EV)];
                    }
                    if (add_acc) l_add[g] = *reinterpret_cast<const u32x4_t*>(Y + yo[g]);
                    if (EXT && p.accumulate) l_acc[g] = *reinterpret_cast<const u32x4_t*>(Y + yo[g]);
                    if (bnb) {
                        l_y[g] = *reinterpret_cast<const u32x4_t*>(Ybn + yo[g]);
                        if (EXT && Zbn) l_z[g] = *reinterpret_cast<const u32x4_t*>(Zbn + yo[g]);
                        if (Mbn) l_m[g] = Mbn[mpix * (unsigned)mrow + (unsigned)(ccol / EV)];
                    }
                }
                pix += PPI; rem += PPI;
                while (rem >= HoWo) { rem -= HoWo; ++b; }
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if (!live[g]) continue;
                u32x4_t v = *reinterpret_cast<const u32x4_t*>(stage + ((k0 + g) * PPI + sp) * SROW + sc_ * 16);
                if (GENERAL && (p.res_mode != 0 || p.accumulate)) {
                    // residual (same-size or nearest-upsampled source) and accumulate: 16-byte coalesced loads
                    Vec16<OT> a; a.load(reinterpret_cast<const OT*>(&v));
                    if (p.res_mode != 0 || add_acc) {
                        Vec16<OT> r; r.load(reinterpret_cast<const OT*>(&l_add[g]));
                        if (p.res_mode != 0 && pk.res_mask) {
#pragma unroll
                            for (int e = 0; e < EV; ++e) r.v[e] = ((l_rm[g] >> e) & 1u) ? r.v[e] : 0.f;
                        }
#pragma unroll
                        for (int e = 0; e < EV; ++e) a.v[e] += r.v[e];
                    }
                    if (EXT && p.accumulate) {
                        Vec16<OT> r; r.load(reinterpret_cast<const OT*>(&l_acc[g]));
#pragma unroll
                        for (int e = 0; e < EV; ++e) a.v[e] += r.v[e];
                    }
                    if (p.act == 3) {                            // ReLU AFTER the residual add: relu(bn3(conv3(.)) + shortcut), fpn.py:30-33
#pragma unroll
                        for (int e = 0; e < EV; ++e) a.v[e] = fmaxf(a.v[e], 0.f);
                    }
                    a.store(reinterpret_cast<OT*>(&v));
                }
                *reinterpret_cast<u32x4_t*>(Y + yo[g]) = v;
                if (bnb) {
                    Vec16<OT> dzv, yy, zz;
                    dzv.load(reinterpret_cast<const OT*>(&v));                  // the value as stored
                    yy.load(reinterpret_cast<const OT*>(&l_y[g]));
                    if (EXT && Zbn) zz.load(reinterpret_cast<const OT*>(&l_z[g]));
#pragma unroll
                    for (int e = 0; e < EV; ++e) {
                        float gg = dzv.v[e];
                        if (pk.bnb_relu) {
                            const float zv = Mbn ? (((l_m[g] >> e) & 1u) ? 1.f : 0.f) : ((EXT && Zbn) ? zz.v[e] : yy.v[e] * bsc[e] + bsf[e]);
                            if (!(zv > 0.f)) gg = 0.f;
                        }
                        bs1[e] += gg;
                        bs2[e] += gg * ((yy.v[e] - bmu[e]) * bis[e]);
                    }
                }
            }
        }
        if (ps + 1 < NPASS) __syncthreads();
    }
    if (!stats_first && p.stats && !MPN_DBG(64)) tile_stats(reinterpret_cast<float*>(lds + 4 * REGION));

    if (bnb) {
        // lanes sharing a channel chunk (same sc_, different pixel slot sp): butterfly over the sp bits
#pragma unroll
        for (int e = 0; e < EV; ++e) {
#pragma unroll
            for (int m = LPP; m < 64; m <<= 1) {
                bs1[e] += __shfl_xor(bs1[e], m, 64);
                bs2[e] += __shfl_xor(bs2[e], m, 64);
            }
        }
        __syncthreads();                                        // staging area no longer read
        if (sp == 0) {
#pragma unroll
            for (int e = 0; e < EV; ++e) {
                const int row = wc * C::WTC + sc_ * EV + e;      // 0..TC-1
                *reinterpret_cast<float2*>(lds_f + (wp * TC + row) * 2) = make_float2(bs1[e], bs2[e]);
            }
        }
        __syncthreads();
        const int t = threadIdx.x;
        if (t < TC && c0 + t < p.Cout) {
            float a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int w = 0; w < C::WAVES_P; ++w) {
                a1 += lds_f[(w * TC + t) * 2 + 0];
                a2 += lds_f[(w * TC + t) * 2 + 1];
            }
            store_partial(pk.bnb_partial + ((long)tp * p.Cout + c0 + t) * 2, a1, a2, pk.fin_counters != nullptr);
        }
    }
    if (pk.fin_counters) fin_last_arriver<TC>(pk, c0, tc, tp, ntiles, lds);
}

// LDS-DMA plumbing (see conv_wgrad.hip for the probe-verified semantics): `buffer_load_dwordx4 ... lds` writes
// 16 bytes per lane at M0 + lane*16, zero for lanes whose voffset + soffset is beyond num_records.
typedef int i32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ i32x4_t make_rsrc(const void* base, unsigned bytes) {
    const uint64_t a = (uint64_t)base;
    i32x4_t r;
    r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));
    r.z = __builtin_amdgcn_readfirstlane((int)bytes);
    r.w = 0x00020000;
    return r;
}

__device__ __forceinline__ void lds_dma16(unsigned voff, i32x4_t rsrc, unsigned soff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_dst) : "memory");
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// tools/kloop_profile.py only (PROF instantiations; production kernels compile none of it): per-wave s_memtime cycle sums of the
// k-loop phases — own-DMA wait, barrier wait, DMA issue, fragment reads + MFMA issue — written to [workgroup][wave][8] at the end
__device__ unsigned long long* d_igemm_prof = nullptr;
struct KProf {
    unsigned long long pt[4] = {0ull, 0ull, 0ull, 0ull}, t[4], start;
    __device__ __forceinline__ void begin() { start = __builtin_readcyclecounter(); }
    template <int I> __device__ __forceinline__ void mark() { t[I] = __builtin_readcyclecounter(); }
    __device__ __forceinline__ void step() {
        const unsigned long long e = __builtin_readcyclecounter();
        pt[0] += t[1] - t[0]; pt[1] += t[2] - t[1]; pt[2] += t[3] - t[2]; pt[3] += e - t[3];
    }
    __device__ __forceinline__ void finish(int nsteps, unsigned long long loop_end) {
        if ((threadIdx.x & 63) == 0 && d_igemm_prof) {
            unsigned long long* d = d_igemm_prof + ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8;
            d[0] = pt[0]; d[1] = pt[1]; d[2] = pt[2]; d[3] = pt[3];
            d[4] = loop_end - start; d[5] = __builtin_readcyclecounter() - loop_end; d[6] = (unsigned long long)nsteps; d[7] = start;
        }
    }
};

// Main kernel.  Operand tiles travel HBM -> LDS by DMA into a 3-deep ring, two k-steps ahead of the MFMAs; no
// staging registers, no ds_write pass.  A tile row is one 64-byte k-chunk of a cout (A) or of a gathered pixel (B);
// rows are stored back to back (the DMA destination is lane-linear) and the four 16-byte pieces of row `i` are
// XOR-swizzled with g(i >> 2), g(q) = (-q) & 3 — applied on the source side: the lane that owns LDS slot (row, j)
// fetches piece j ^ g(row >> 2).  ds_read_b128 is serviced in four NON-contiguous 16-lane groups
// ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ...; MI355X_MICROARCH.md, LDS table): with this g every group's 16
// (row, piece) pairs land on 16 distinct 4-bank slots.
// Halo pixels, rows past Cout / past the last pixel carry an out-of-range offset and arrive as zeros.  The loads
// are invisible to the compiler's waitcnt pass: completion is counted by hand (vmcnt(LPS) = everything but the
// newest k-step has landed) and the barrier is the raw s_barrier, so the ring never drains inside the loop.
template <typename T, int TC, int TP, bool OUTF32, bool GENERAL, bool EXT = false, bool PROF = false>
__global__ void __launch_bounds__(256, TC > 128 ? 2 : 3) conv_igemm_kernel(const MpnConvParams pk, const int dbg) {
    using C = ConvCfg<T, TC, TP>;
    __shared__ __attribute__((aligned(16))) unsigned char lds[C::NST * C::STAGE_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wc = wave / C::WAVES_P, wp = wave % C::WAVES_P;
    const int tilesC = (pk.Cout_store + TC - 1) / TC;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    int tp = bid / tilesC;
    const int tc = bid - tp * tilesC;
    MpnConvParams p = pk;                        // wave-uniform working copy; pyramid mode patches in the level's tensors
    if (pk.nseg > 0) {
        int l = 0;
#pragma unroll
        for (int k = 1; k < MPN_MAX_SEG; ++k) l += (k < pk.nseg && tp >= pk.seg_tile0[k]) ? 1 : 0;
        l = __builtin_amdgcn_readfirstlane(l);                  // block-uniform: keep the level's geometry in scalar registers
        tp -= pk.seg_tile0[l];
        p.x = pk.seg_x[l]; p.y = pk.seg_y[l];
        p.H = p.Ho = pk.seg_H[l]; p.W = p.Wo = pk.seg_W[l];
        p.x_sH = (int64_t)p.W * pk.x_sW; p.x_sB = (int64_t)p.H * p.x_sH;
        p.y_sB = (int64_t)p.H * p.W * pk.y_sP;
    }
    const unsigned HoWo = (unsigned)p.Ho * (unsigned)p.Wo;
    const unsigned P = (unsigned)p.B * HoWo;
    const long p0 = (long)tp * TP;
    const int c0 = tc * TC;
    constexpr unsigned TS = (unsigned)sizeof(T);
    const long KW = (long)(pk.w_taps > 0 ? pk.w_taps : p.R * p.S) * p.Cin;      // w_taps: a tap subset of a larger filter (mpn.h: parity classes)
    const int sh = p.stride - 1;          // dgrad supports stride 1 or 2
    // 32-bit per-lane byte offsets into buffer descriptors; the k-chunk offset rides in the scalar soffset (the
    // launcher guarantees tensor bytes + one row < 4 GB so marker + soffset cannot wrap)
    const unsigned x_bytes = (unsigned)((long)p.B * p.x_sB * TS);
    const unsigned w_bytes = (unsigned)((long)p.Cout * KW * TS);
    const i32x4_t rsrc_x = make_rsrc(p.x, x_bytes);
    const i32x4_t rsrc_w = make_rsrc(p.w, w_bytes);
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const unsigned wave_u = (unsigned)__builtin_amdgcn_readfirstlane(wave);

    // DMA units: one instruction fills 16 consecutive tile rows (1 KiB); lane -> row (lane >> 2), slot (lane & 3)
    const int u_row = lane >> 2;
    const int u_piece = (lane & 3) ^ ((0 - (lane >> 4)) & 3);    // source piece for this lane's slot: j ^ g(row >> 2)
    unsigned a_voff[C::A_PER_W];
#pragma unroll
    for (int q = 0; q < C::A_PER_W; ++q) {
        const int row = ((int)wave_u * C::A_PER_W + q) * 16 + u_row;
        const int cout = c0 + row;
        const bool ok = (row < TC) && (cout < p.Cout);
        a_voff[q] = ok ? (unsigned)(((long)cout * KW + u_piece * C::V) * TS) : w_bytes;
    }
    unsigned b_base[C::B_PER_W], b_voff[C::B_PER_W];
    int b_h[C::B_PER_W], b_w[C::B_PER_W];
    bool b_ok[C::B_PER_W];
#pragma unroll
    for (int q = 0; q < C::B_PER_W; ++q) {
        const int row = ((int)wave_u * C::B_PER_W + q) * 16 + u_row;
        const unsigned pix = (unsigned)p0 + row;
        const bool ok = pix < P;
        const unsigned pc = ok ? pix : 0u;
        const unsigned b = pc / HoWo;
        const unsigned rem = pc - b * HoWo;
        const int ho = (int)(rem / (unsigned)p.Wo), wo = (int)(rem - (unsigned)ho * (unsigned)p.Wo);
        b_ok[q] = ok;
        if (p.mode == 0) { b_h[q] = ho * p.stride - p.pad; b_w[q] = wo * p.stride - p.pad; }
        else             { b_h[q] = ho + p.pad;            b_w[q] = wo + p.pad; }
        b_base[q] = (unsigned)(((long)b * p.x_sB + u_piece * C::V) * TS);
        b_voff[q] = x_bytes;
    }
    const unsigned sH_b = (unsigned)(p.x_sH * TS), sW_b = (unsigned)(p.x_sW * TS);

    f32x4_t acc[C::MC][C::MP];
#pragma unroll
    for (int i = 0; i < C::MC; ++i)
#pragma unroll
        for (int j = 0; j < C::MP; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const int nsteps = p.R * p.S * (p.Cin / C::KC);
    int r = 0, s = 0, cc = 0;
    int klin = 0;

    auto issue = [&](unsigned stage) {       // queue one k-step (LPS DMA instructions per wave) into ring slot `stage`
        if (cc == 0) {                       // new tap (uniform): per-row gather offsets, OOB offset for halo / dead rows
#pragma unroll
            for (int q = 0; q < C::B_PER_W; ++q) {
                int hi, wi;
                bool ok = b_ok[q];
                if (p.mode == 0) {
                    hi = b_h[q] + r; wi = b_w[q] + s;
                } else {
                    const int th = b_h[q] - r, tw = b_w[q] - s;
                    ok = ok && th >= 0 && tw >= 0 && (((th | tw) & sh) == 0);
                    hi = th >> sh; wi = tw >> sh;
                }
                ok = ok && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
                b_voff[q] = ok ? b_base[q] + (unsigned)hi * sH_b + (unsigned)wi * sW_b : x_bytes;
            }
            if (pk.w_taps > 0) klin = (pk.wtap0 + r * pk.wtap_dr + s * pk.wtap_ds) * p.Cin;      // tap (r, s) of this launch = that tap of the filter
        }
        const unsigned so_w = (unsigned)klin * TS, so_x = (unsigned)cc * TS;
        const unsigned st = lds_base + stage * C::STAGE_BYTES;
#pragma unroll
        for (int q = 0; q < C::A_PER_W; ++q)
            lds_dma16(a_voff[q], rsrc_w, so_w, __builtin_amdgcn_readfirstlane(st + (wave_u * C::A_PER_W + q) * 1024u));
#pragma unroll
        for (int q = 0; q < C::B_PER_W; ++q)
            lds_dma16(b_voff[q], rsrc_x, so_x, __builtin_amdgcn_readfirstlane(st + C::A_BYTES + (wave_u * C::B_PER_W + q) * 1024u));
        klin += C::KC; cc += C::KC;
        if (cc == p.Cin) { cc = 0; if (++s == p.S) { s = 0; ++r; } }
    };
    // fragment gather: lane -> row (lane & 15) of a 16-row group, k-piece (lane >> 4), un-swizzled by g(row >> 2)
    const int f_off = (lane & 15) * 64 + (((lane >> 4) ^ ((0 - ((lane >> 2) & 3)) & 3)) * 16);
    const int fa_off = wc * C::WTC * 64 + f_off;
    const int fb_off = C::A_BYTES + wp * C::WTP * 64 + f_off;
    auto compute = [&](unsigned stage) {
        const unsigned char* base = lds + stage * C::STAGE_BYTES;
        u32x4_t fa[C::MC], fb[C::MP];
#pragma unroll
        for (int i = 0; i < C::MC; ++i)
            fa[i] = *reinterpret_cast<const u32x4_t*>(base + fa_off + i * 1024);
#pragma unroll
        for (int j = 0; j < C::MP; ++j)
            fb[j] = *reinterpret_cast<const u32x4_t*>(base + fb_off + j * 1024);
#pragma unroll
        for (int i = 0; i < C::MC; ++i)
#pragma unroll
            for (int j = 0; j < C::MP; ++j) Mma<T>::run(acc[i][j], fa[i], fb[j]);
    };
    if (MPN_DBG(32)) return;                 // ablation: prologue only

    KProf kp;
    if (PROF) kp.begin();
    issue(0u);
    if (nsteps > 1) issue(1u);
    unsigned cur = 0u, nxt = 2u;
    for (int it = 0; it < nsteps; ++it) {
        if (PROF) kp.mark<0>();
        if (it + 1 < nsteps) wait_vmcnt<C::LPS>(); else wait_vmcnt<0>();     // k-step `it` has landed (this wave's part)
        if (PROF) kp.mark<1>();
        __builtin_amdgcn_s_barrier();                                          // ... everyone's; slot `nxt` is free again
        if (PROF) kp.mark<2>();
        if (it + 2 < nsteps) issue(nxt);
        if (PROF) kp.mark<3>();
        compute(cur);
        if (PROF) kp.step();
        cur = (cur == C::NST - 1) ? 0u : cur + 1u;
        nxt = (nxt == C::NST - 1) ? 0u : nxt + 1u;
    }
    const unsigned long long loop_end = PROF ? __builtin_readcyclecounter() : 0ull;
    __syncthreads();                          // the epilogue re-uses the ring as its staging area

    if (MPN_DBG(16)) { if (acc[0][0][0] == 123.456f) ((float*)p.y)[0] = 1.f; return; }   // ablation: no epilogue
    const int ntiles = (int)(gridDim.x / (unsigned)tilesC);       // pixel tiles of the launch (in-launch finalize)
    if (OUTF32) conv_epilogue<T, float, TC, TP, GENERAL, EXT>(p, acc, c0, p0, wc, wp, lane, tp, lds, dbg, tc, ntiles, pk);
    else        conv_epilogue<T, T, TC, TP, GENERAL, EXT, EXT>(p, acc, c0, p0, wc, wp, lane, tp, lds, dbg, tc, ntiles, pk);
    if (PROF) kp.finish(nsteps, loop_end);
}

// 3x3 / stride 1 / pad 1 variant (forward and stride-1 dgrad), 16-bit types: the k-loop is bound by operand delivery (global -> LDS
// DMA at ~45 of the CU's 64 B/clk), and the three horizontal taps of a kernel row read the SAME pixels shifted by one.  The
// pixel tile of a (kernel row r, channel chunk) group therefore lands ONCE, as TP + 2 rows (pixels p0-1 .. p0+TP of the row r input
// line; padded to 192 DMA rows), and the taps s = 0..2 read it at row offsets 0 / 1 / 2.  What the shared rows cannot encode — the
// left / right image border, which depends on the OUTPUT pixel's column — is a per-lane select on the fragments of the two outer
// taps; the vertical border is a property of the row itself (every consumer of a row sits in the same output line) and stays a
// DMA-time zero fill.  Per group: 3 weight tiles + 1 pixel tile instead of 3 + 3 (-31 % DMA bytes, -25 % DMA instructions).
// k order: (r, channel chunk, s).  LDS: weight ring 3 x A_BYTES, pixel ring 2 x 12 KB (same total as the generic kernel).
template <typename T, int TC, int TP, bool OUTF32, bool GENERAL, bool EXT = false, bool PROF = false>
__global__ void __launch_bounds__(256, TC > 128 ? 2 : 3) conv_igemm_s3_kernel(const MpnConvParams pk, const int dbg) {
    using C = ConvCfg<T, TC, TP>;
    static_assert(sizeof(T) == 2 && TP == 128, "16-bit operands, 128-pixel tiles");
    constexpr int B_ROWS = 192, B_BYTES = B_ROWS * 64, LB = 3, LA = C::A_PER_W;
    constexpr int A_RING = 3 * C::A_BYTES;
    static_assert(A_RING + 2 * B_BYTES == C::NST * C::STAGE_BYTES, "same LDS footprint as the generic kernel");
    __shared__ __attribute__((aligned(16))) unsigned char lds[A_RING + 2 * B_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wc = wave / C::WAVES_P, wp = wave % C::WAVES_P;
    const int tilesC = (pk.Cout_store + TC - 1) / TC;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    int tp = bid / tilesC;
    const int tc = bid - tp * tilesC;
    MpnConvParams p = pk;
    if (pk.nseg > 0) {
        int l = 0;
#pragma unroll
        for (int k = 1; k < MPN_MAX_SEG; ++k) l += (k < pk.nseg && tp >= pk.seg_tile0[k]) ? 1 : 0;
        l = __builtin_amdgcn_readfirstlane(l);
        tp -= pk.seg_tile0[l];
        p.x = pk.seg_x[l]; p.y = pk.seg_y[l];
        p.H = p.Ho = pk.seg_H[l]; p.W = p.Wo = pk.seg_W[l];
        p.x_sH = (int64_t)p.W * pk.x_sW; p.x_sB = (int64_t)p.H * p.x_sH;
        p.y_sB = (int64_t)p.H * p.W * pk.y_sP;
    }
    const unsigned HoWo = (unsigned)p.Ho * (unsigned)p.Wo;
    const unsigned P = (unsigned)p.B * HoWo;
    const long p0 = (long)tp * TP;
    const int c0 = tc * TC;
    constexpr unsigned TS = 2u;
    const long KW = 9L * p.Cin;
    const unsigned x_bytes = (unsigned)((long)p.B * p.x_sB * TS);
    const unsigned w_bytes = (unsigned)((long)p.Cout * KW * TS);
    const i32x4_t rsrc_x = make_rsrc(p.x, x_bytes);
    const i32x4_t rsrc_w = make_rsrc(p.w, w_bytes);
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const unsigned wave_u = (unsigned)__builtin_amdgcn_readfirstlane(wave);

    const int u_row = lane >> 2;
    const int u_piece = (lane & 3) ^ ((0 - (lane >> 4)) & 3);
    // The pixel tile is read at row offsets 0 / 1 / 2 (the three taps).  The generic key g(row >> 2) = (-q) & 3 is conflict-free only
    // for 16-row-aligned fragments: shifted by one or two rows, two lane pairs of every ds_read_b128 service group ({0-3,12-15,20-27},
    // ...) meet on a bank quad (PMC: bank-conflict cycles 6.7 % of the kernel against 3.4 % in the generic kernel).  The key
    // 2 * ((row >> 2) & 1) keeps all 16 (row, piece) pairs of every group on distinct quads for ALL three offsets (exhaustive check
    // over keys of the row index mod 16: tools/lds_swizzle_search.py).
    const int u_piece_b = (lane & 3) ^ (((lane >> 4) & 1) << 1);
    unsigned a_voff[LA];
#pragma unroll
    for (int q = 0; q < LA; ++q) {
        const int row = ((int)wave_u * LA + q) * 16 + u_row;
        const int cout = c0 + row;
        const bool ok = (row < TC) && (cout < p.Cout);
        a_voff[q] = ok ? (unsigned)(((long)cout * KW + u_piece * C::V) * TS) : w_bytes;
    }
    // pixel-tile DMA units: LDS row i holds input pixel (linear index) p0 - 1 + i of the line the current kernel row selects
    unsigned b_base[LB];
    int b_h[LB];
    unsigned b_wb[EXT ? LB : 1];                                  // (image << 16) | column: only for the virtual concatenation (EXT instantiation; launcher: B, W < 65 536)
    bool b_ok[LB];
    const bool vcat = EXT && pk.kseg_n > 0;                       // virtual channel concatenation of up-sampled sources (mpn.h: kseg_*)
#pragma unroll
    for (int q = 0; q < LB; ++q) {
        const int i = ((int)wave_u * LB + q) * 16 + u_row;
        const long pc = p0 - 1 + i;
        const bool ok = i < TP + 2 && pc >= 0 && pc < (long)P;
        const unsigned pcu = ok ? (unsigned)pc : 0u;
        const unsigned bi = pcu / HoWo;
        const unsigned rem = pcu - bi * HoWo;
        b_h[q] = (int)(rem / (unsigned)p.Wo);
        if (EXT) b_wb[q] = (bi << 16) | (rem - (unsigned)b_h[q] * (unsigned)p.Wo);
        b_ok[q] = ok;
        b_base[q] = (unsigned)(((long)pcu * p.x_sW + u_piece_b * C::V) * TS);
    }
    const int line_b = (int)(p.W * p.x_sW * TS);                  // bytes between two image lines

    f32x4_t acc[C::MC][C::MP];
#pragma unroll
    for (int i = 0; i < C::MC; ++i)
#pragma unroll
        for (int j = 0; j < C::MP; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // fragment gather.  Weights: as in the generic kernel.  Pixels: fragment j, lane -> tile row  l + 1 + dx  (l = local pixel index,
    // dx = -1 / 0 / +1 the tap's column shift), un-swizzled with that ROW's key.
    const int f_off = (lane & 15) * 64 + (((lane >> 4) ^ ((0 - ((lane >> 2) & 3)) & 3)) * 16);
    const int fa_off = wc * C::WTC * 64 + f_off;
    // (fragment j sits 16 rows = 1 KiB further on and 16 rows do not change bit 2 of the row index, i.e. the key: one offset per tap
    // shift, j * 1024 rides in the ds_read's immediate — 3 registers instead of 3 * MP; round 5, after the 256-row extended
    // instantiation had started to reload spilled gather state from scratch — with a vmcnt(0) — inside its k-loop)
    int fb_off[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int row = wp * C::WTP + (lane & 15) + d;
        fb_off[d] = row * 64 + (((lane >> 4) ^ (((row >> 2) & 1) << 1)) * 16);
    }
    unsigned edge_l = 0u, edge_r = 0u;                            // bit j: the lane's pixel of fragment j sits in the first / last column
#pragma unroll
    for (int j = 0; j < C::MP; ++j) {
        const int l = wp * C::WTP + j * 16 + (lane & 15);
        const unsigned pix = (unsigned)p0 + (unsigned)l;
        const unsigned wo = (pix < P ? pix : 0u) % (unsigned)p.Wo;
        edge_l |= (wo == 0u ? 1u : 0u) << j;
        edge_r |= (wo == (unsigned)p.Wo - 1u ? 1u : 0u) << j;
    }

    const int chunks = p.Cin / C::KC;
    const int groups = 3 * chunks;                                // (r, channel chunk) groups of three taps
    // issue state: the next weight tile to queue is tap (ar, as_) of chunk acc_; the next pixel tile is group (br, bcc)
    int ar = 0, acc_ = 0, as_ = 0, br = 0, bcc = 0;
    auto issue_a = [&](unsigned slot) {
        const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)((ar * 3 + as_) * p.Cin + acc_) * TS));
        const unsigned st = lds_base + slot * C::A_BYTES;
#pragma unroll
        for (int q = 0; q < LA; ++q)
            lds_dma16(a_voff[q], rsrc_w, so, __builtin_amdgcn_readfirstlane(st + (wave_u * LA + q) * 1024u));
        if (++as_ == 3) { as_ = 0; acc_ += C::KC; if (acc_ == p.Cin) { acc_ = 0; ++ar; } }
    };
    auto issue_b = [&](unsigned slot) {
        const int dy = (p.mode == 0) ? br - 1 : 1 - br;           // input line relative to the output line
        const unsigned st = lds_base + A_RING + slot * B_BYTES;
        bool done = false;
        if constexpr (EXT) {
            if (vcat) {
                // channel chunk bcc lives in segment sg = bcc / kseg_c: a dense [B][H >> sh][W >> sh][kseg_c] tensor read at (h >> sh, w >> sh)
                const int sg = __builtin_amdgcn_readfirstlane(bcc / pk.kseg_c);
                const int sh = __builtin_amdgcn_readfirstlane(pk.kseg_shift[sg]);
                const int Hs = p.H >> sh, Ws = p.W >> sh;
                const i32x4_t rs = make_rsrc(pk.kseg_x[sg], (unsigned)((long)p.B * Hs * Ws * pk.kseg_c * TS));
                const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(bcc - sg * pk.kseg_c) * TS));
#pragma unroll
                for (int q = 0; q < LB; ++q) {
                    const int hh = b_h[q] + dy;
                    const bool ok = b_ok[q] && (unsigned)hh < (unsigned)p.H;
                    const unsigned src = (unsigned)(((int)(b_wb[q] >> 16) * Hs + (hh >> sh)) * Ws + (int)((b_wb[q] & 0xffffu) >> sh));
                    const unsigned voff = ok ? (src * (unsigned)pk.kseg_c + (unsigned)(u_piece_b * C::V)) * TS : 0x80000000u;
                    lds_dma16(voff, rs, so, __builtin_amdgcn_readfirstlane(st + (wave_u * LB + q) * 1024u));
                }
                done = true;
            }
        }
        if (!done) {
            const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)bcc * TS));
#pragma unroll
            for (int q = 0; q < LB; ++q) {
                const bool ok = b_ok[q] && (unsigned)(b_h[q] + dy) < (unsigned)p.H;
                const unsigned voff = ok ? (unsigned)((int)b_base[q] + dy * line_b) : x_bytes;
                lds_dma16(voff, rsrc_x, so, __builtin_amdgcn_readfirstlane(st + (wave_u * LB + q) * 1024u));
            }
        }
        bcc += C::KC;
        if (bcc == p.Cin) { bcc = 0; ++br; }
    };
    auto compute = [&](unsigned aslot, unsigned bslot, int s) {
        const int dx = (p.mode == 0) ? s : 2 - s;                 // tile row offset 0 / 1 / 2  <=>  column shift -1 / 0 / +1
        const unsigned char* abase = lds + aslot * C::A_BYTES;
        const unsigned char* bbase = lds + A_RING + bslot * B_BYTES;
        u32x4_t fa[C::MC], fb[C::MP];
#pragma unroll
        for (int i = 0; i < C::MC; ++i)
            fa[i] = *reinterpret_cast<const u32x4_t*>(abase + fa_off + i * 1024);
        const int off = dx == 0 ? fb_off[0] : (dx == 1 ? fb_off[1] : fb_off[2]);
#pragma unroll
        for (int j = 0; j < C::MP; ++j) fb[j] = *reinterpret_cast<const u32x4_t*>(bbase + off + j * 1024);
        if (dx != 1) {                                           // outer taps: pixels beyond the left / right border read zeros
            const unsigned edge = dx == 0 ? edge_l : edge_r;
#pragma unroll
            for (int j = 0; j < C::MP; ++j)
                if ((edge >> j) & 1u) fb[j] = (u32x4_t){0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int i = 0; i < C::MC; ++i)
#pragma unroll
            for (int j = 0; j < C::MP; ++j) Mma<T>::run(acc[i][j], fa[i], fb[j]);
    };

    // prologue: pixel tile of group 0, weight tiles of its first two taps
    KProf kp;
    if (PROF) kp.begin();
    issue_b(0u);
    issue_a(0u);
    issue_a(1u);
    unsigned acur = 0u, anxt = 2u;
    for (int g = 0; g < groups; ++g) {
        const unsigned bcur = (unsigned)(g & 1);
        const bool last = g + 1 == groups;
        // tap 0: needs weights (g,0) and pixel tile g; younger in flight: weights (g,1)
        if (PROF) kp.mark<0>();
        if (last) wait_vmcnt<0>(); else wait_vmcnt<LA>();
        if (PROF) kp.mark<1>();
        __builtin_amdgcn_s_barrier();
        if (PROF) kp.mark<2>();
        if (!last) issue_b(bcur ^ 1u);                            // pixel tile of the next group (its slot was last read one step ago)
        issue_a(anxt);                                            // weights (g,2)
        if (PROF) kp.mark<3>();
        compute(acur, bcur, 0);
        if (PROF) kp.step();
        acur = (acur == 2u) ? 0u : acur + 1u; anxt = (anxt == 2u) ? 0u : anxt + 1u;
        // tap 1: needs weights (g,1); younger: pixel tile g+1 and weights (g,2)
        if (PROF) kp.mark<0>();
        if (last) wait_vmcnt<0>(); else wait_vmcnt<LB + LA>();
        if (PROF) kp.mark<1>();
        __builtin_amdgcn_s_barrier();
        if (PROF) kp.mark<2>();
        if (!last) issue_a(anxt);                                 // weights (g+1,0)
        if (PROF) kp.mark<3>();
        compute(acur, bcur, 1);
        if (PROF) kp.step();
        acur = (acur == 2u) ? 0u : acur + 1u; anxt = (anxt == 2u) ? 0u : anxt + 1u;
        // tap 2: needs weights (g,2); younger: weights (g+1,0)
        if (PROF) kp.mark<0>();
        if (last) wait_vmcnt<0>(); else wait_vmcnt<LA>();
        if (PROF) kp.mark<1>();
        __builtin_amdgcn_s_barrier();
        if (PROF) kp.mark<2>();
        if (!last) issue_a(anxt);                                 // weights (g+1,1)
        if (PROF) kp.mark<3>();
        compute(acur, bcur, 2);
        if (PROF) kp.step();
        acur = (acur == 2u) ? 0u : acur + 1u; anxt = (anxt == 2u) ? 0u : anxt + 1u;
    }
    const unsigned long long loop_end = PROF ? __builtin_readcyclecounter() : 0ull;
    __syncthreads();

    const int ntiles = (int)(gridDim.x / (unsigned)tilesC);
    if (OUTF32) conv_epilogue<T, float, TC, TP, GENERAL, EXT>(p, acc, c0, p0, wc, wp, lane, tp, lds, dbg, tc, ntiles, pk);
    else        conv_epilogue<T, T, TC, TP, GENERAL, EXT>(p, acc, c0, p0, wc, wp, lane, tp, lds, dbg, tc, ntiles, pk);
    if (PROF) kp.finish(groups * 3, loop_end);
}

constexpr int kTP = 128;
#if MPN_EXP
bool g_igemm_prof = false;          // tools/kloop_profile.py: route 128-row bf16 launches to the PROF instantiations
#endif

// Block tile height (output channels).  The k-loop is bound by the DMA/LDS path, so the tallest tile that still
// fills the chip wins: 256 rows (2 workgroups per CU) for long contractions with enough workgroups, else 128 rows
// down to ~200 workgroups (measured: 450 x 128x128 beats 900 x 64x128 at 30x30), else 64.
inline int pick_tc(const MpnConvParams& p, long tilesP) {
    const int cout_store = p.Cout_store;
    if (cout_store <= 32) return 32;
    if (cout_store <= 64) return 64;
    static const long min_blocks = mpn_tune("MPN_TC_MIN_BLOCKS", 200);
    static const long min_blocks256 = mpn_tune("MPN_TC256_MIN_BLOCKS", 400);
    static const long min_ksteps256 = mpn_tune("MPN_TC256_MIN_KSTEPS", 16);
    const long ksteps = (long)p.R * p.S * p.Cin / (p.dtype == MPN_F32 ? 16 : 32);
    // (f32: the exact-fp32 MFMA needs 8 passes per 16x16x4 — the loop is bound by the matrix pipe, not by operand delivery, and three
    //  128-row workgroups per CU keep it busier than two 256-row ones: cfg2 55.7 -> 54.7 ms, profiles/r05_cfg2_tile_rule_sweep.txt)
    if (p.dtype != MPN_F32 && cout_store >= 256 && tilesP * ((cout_store + 255) / 256) >= min_blocks256 && ksteps >= min_ksteps256) return 256;
    const long blocks128 = tilesP * ((cout_store + 127) / 128);
    return blocks128 >= min_blocks ? 128 : 64;
}

inline bool conv_uses_s3(const MpnConvParams& p, int tc) {
    static const bool on = mpn_tune("MPN_IGEMM_S3", 1) != 0;
    if (!on || p.dtype == MPN_F32 || p.R != 3 || p.S != 3 || p.stride != 1 || p.pad != 1 || tc < 64) return false;
    if (p.nseg > 0 || p.kseg_n > 0) return true;                  // pyramid levels / concatenation members are dense by construction
    return p.H == p.Ho && p.W == p.Wo && p.x_sH == (int64_t)p.W * p.x_sW && p.x_sB == (int64_t)p.H * p.x_sH;
}

template <typename T, bool OUTF32, bool GENERAL, bool EXT = false>
int launch_conv_k(const MpnConvParams& p, int tc, long grid, int dbg, hipStream_t st) {
#if MPN_EXP
    if constexpr (std::is_same<T, bf16_t>::value && !OUTF32 && !EXT) {
        if (g_igemm_prof && tc == 128 && !p.fin_counters) {
            if (conv_uses_s3(p, tc)) hipLaunchKernelGGL((conv_igemm_s3_kernel<T, 128, kTP, false, GENERAL, false, true>), dim3((unsigned)grid), dim3(256), 0, st, p, dbg);
            else hipLaunchKernelGGL((conv_igemm_kernel<T, 128, kTP, false, GENERAL, false, true>), dim3((unsigned)grid), dim3(256), 0, st, p, dbg);
            return mpn_launch_status();
        }
    }
#endif
    if constexpr (sizeof(T) == 2) {
        if (conv_uses_s3(p, tc)) {
            if (tc == 256) hipLaunchKernelGGL((conv_igemm_s3_kernel<T, 256, kTP, OUTF32, GENERAL, EXT>), dim3((unsigned)grid), dim3(256), 0, st, p, dbg);
            else if (tc == 128) hipLaunchKernelGGL((conv_igemm_s3_kernel<T, 128, kTP, OUTF32, GENERAL, EXT>), dim3((unsigned)grid), dim3(256), 0, st, p, dbg);
            else hipLaunchKernelGGL((conv_igemm_s3_kernel<T, 64, kTP, OUTF32, GENERAL, EXT>), dim3((unsigned)grid), dim3(256), 0, st, p, dbg);
            return mpn_launch_status();
        }
    }
    if (tc == 256) {
        // pick_tc() never chooses the 256-row tile for f32 (the exact-fp32 MFMA loop is matrix-pipe bound: three 128-row workgroups per
        // CU beat two 256-row ones, profiles/r05_cfg2_tile_rule_sweep.txt): no <float, 256, ...> instantiation exists
        if constexpr (sizeof(T) == 2) hipLaunchKernelGGL((conv_igemm_kernel<T, 256, kTP, OUTF32, GENERAL, EXT>), dim3((unsigned)grid), dim3(256), 0, st, p, dbg);
        else return MPN_E_UNSUPPORTED;
    } else if (tc == 128) hipLaunchKernelGGL((conv_igemm_kernel<T, 128, kTP, OUTF32, GENERAL, EXT>), dim3((unsigned)grid), dim3(256), 0, st, p, dbg);
    else if (tc == 64) hipLaunchKernelGGL((conv_igemm_kernel<T, 64, kTP, OUTF32, GENERAL, EXT>), dim3((unsigned)grid), dim3(256), 0, st, p, dbg);
    else hipLaunchKernelGGL((conv_igemm_kernel<T, 32, kTP, OUTF32, GENERAL, EXT>), dim3((unsigned)grid), dim3(256), 0, st, p, dbg);
    return mpn_launch_status();
}

// the extended epilogue (see conv_epilogue): only when a launch asks for one of its features
inline bool conv_needs_ext(const MpnConvParams& p) {
    return (p.bnb_partial && p.bnb_relu && p.bnb_z && !p.bnb_mask) ||
           (p.res_mode && p.accumulate) || p.kseg_n > 0 || p.y_step > 1;
}

template <typename T, bool OUTF32>
int launch_conv(const MpnConvParams& p, hipStream_t st) {
    const long P = p.nseg > 0 ? (long)p.seg_tile0[p.nseg] * kTP : (long)p.B * p.Ho * p.Wo;
    const long tilesP = p.nseg > 0 ? (long)p.seg_tile0[p.nseg] : (P + kTP - 1) / kTP;
    const int tc = pick_tc(p, tilesP);
    const long tilesC = (p.Cout_store + tc - 1) / tc;
    const long grid = tilesP * tilesC;
    if (grid <= 0 || grid > 0x7fffffffL || P >= 0x7fffffffL) return MPN_E_BADARG;
    static const int dbg = (int)mpn_tune("MPN_DEBUG_FLAGS", 0);   // microbenchmark ablations (experiments build only)
    // "plain" = conv (+ BN tile statistics): no per-element epilogue math at all, lighter register footprint
    const bool general = p.scale || p.bias || p.res_mode || p.accumulate || p.act || (p.Cout % tc) != 0 || p.bnb_partial;
    if (conv_needs_ext(p)) {
        if constexpr (OUTF32) return MPN_E_UNSUPPORTED;          // none of the extended features writes f32 from 16-bit operands
        else return launch_conv_k<T, false, true, true>(p, tc, grid, dbg, st);
    }
    return general ? launch_conv_k<T, OUTF32, true>(p, tc, grid, dbg, st) : launch_conv_k<T, OUTF32, false>(p, tc, grid, dbg, st);
}

}  // namespace

extern "C" int mpn_debug_igemm_prof(void* buf) {
#if MPN_EXP
    g_igemm_prof = buf != nullptr;
    unsigned long long* q = (unsigned long long*)buf;
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(d_igemm_prof), &q, sizeof(q));
#else
    (void)buf;
    return MPN_E_UNSUPPORTED;          // the PROF instantiations live in the experiments build only (common.h)
#endif
}

extern "C" int mpn_conv_stats_tiles(const MpnConvParams* p) {
    if (!p) return MPN_E_BADARG;
    const long P = (long)p->B * p->Ho * p->Wo;
    return (int)((P + kTP - 1) / kTP);
}

extern "C" int mpn_conv_tile_rows(const MpnConvParams* p) {
    if (!p) return MPN_E_BADARG;
    const long P = (long)p->B * p->Ho * p->Wo;
    return pick_tc(*p, p->nseg > 0 ? (long)p->seg_tile0[p->nseg] : (P + kTP - 1) / kTP);
}

extern "C" int mpn_conv_shared_tile(const MpnConvParams* p) {
    if (!p) return MPN_E_BADARG;
    const long P = (long)p->B * p->Ho * p->Wo;
    return conv_uses_s3(*p, pick_tc(*p, p->nseg > 0 ? (long)p->seg_tile0[p->nseg] : (P + kTP - 1) / kTP)) ? 1 : 0;
}

extern "C" int mpn_conv_forward(const MpnConvParams* pp, void* stream) {
    if (!pp) return MPN_E_BADARG;
    const MpnConvParams& p = *pp;
    if (p.nseg > 0) {       // pyramid mode: per-level tensors, shared weights; stride 1, same-size output, no statistics / residual
        MPN_CHECK_ARG(p.nseg <= MPN_MAX_SEG && p.w && p.B > 0 && p.stride == 1 && !p.stats && p.res_mode == 0 && p.seg_tile0[0] == 0);
        MPN_CHECK_ARG(2 * p.pad == p.R - 1 && p.R == p.S);
        for (int l = 0; l < p.nseg; ++l) {
            MPN_CHECK_ARG(p.seg_x[l] && p.seg_y[l] && p.seg_H[l] > 0 && p.seg_W[l] > 0);
            const long tiles = ((long)p.B * p.seg_H[l] * p.seg_W[l] + kTP - 1) / kTP;
            MPN_CHECK_ARG(p.seg_tile0[l + 1] - p.seg_tile0[l] == tiles);
        }
    } else {
        MPN_CHECK_ARG((p.x || p.kseg_n > 0) && p.y && p.Ho > 0 && p.Wo > 0 && p.H > 0 && p.W > 0);
    }
    if (p.kseg_n > 0) {     // virtual concatenation: served by the shared-tile 3x3 kernel only
        MPN_CHECK_ARG(p.kseg_n <= 4 && p.kseg_c > 0 && p.kseg_c % 32 == 0 && p.Cin == p.kseg_n * p.kseg_c && p.mode == 0 && !p.nseg);
        MPN_CHECK_ARG(p.dtype != MPN_F32 && p.R == 3 && p.S == 3 && p.stride == 1 && p.pad == 1 && p.H == p.Ho && p.W == p.Wo && p.Cout_store > 32);
        MPN_CHECK_ARG(p.B < 65536 && p.W < 65536);               // the gather keeps (image, column) in one 32-bit register
        for (int k = 0; k < p.kseg_n; ++k) {
            MPN_CHECK_ARG(p.kseg_x[k] && p.kseg_shift[k] >= 0 && p.kseg_shift[k] < 8 && ((p.H >> p.kseg_shift[k]) << p.kseg_shift[k]) == p.H &&
                          ((p.W >> p.kseg_shift[k]) << p.kseg_shift[k]) == p.W);
            if ((int64_t)p.B * (p.H >> p.kseg_shift[k]) * (p.W >> p.kseg_shift[k]) * p.kseg_c * 2 >= 0x7ffffff0LL) return MPN_E_UNSUPPORTED;
        }
    }
    MPN_CHECK_ARG(p.w && p.B > 0);
    MPN_CHECK_ARG(mpn_dtype_ok(p.dtype));
    const int kc = p.dtype == MPN_F32 ? 16 : 32;
    MPN_CHECK_ARG(p.Cin > 0 && p.Cin % kc == 0);
    MPN_CHECK_ARG(p.Cout > 0 && p.Cout_store >= p.Cout && p.Cout_store % 4 == 0);
    MPN_CHECK_ARG(p.R > 0 && p.S > 0 && p.stride >= 1);
    MPN_CHECK_ARG(p.mode == 0 || (p.mode == 1 && (p.stride == 1 || p.stride == 2)));
    MPN_CHECK_ARG(!(p.accumulate && p.act != 0));
    MPN_CHECK_ARG(p.act >= 0 && p.act <= 3 && (p.act != 3 || p.res_mode == 1));
    MPN_CHECK_ARG(p.res_mode == 0 || p.res != nullptr);
    MPN_CHECK_ARG(!p.res_mask || (p.res_mode == 1 && !p.nseg && p.y_sB == (int64_t)p.Ho * p.Wo * p.y_sP && p.y_sP == p.Cout_store));
    MPN_CHECK_ARG(!(p.stats && (p.bias || p.scale || p.res_mode || p.accumulate || p.act)));
    MPN_CHECK_ARG(!((p.res_mode || p.accumulate) && p.act && p.act != 3));
    MPN_CHECK_ARG(!p.fin_counters || (p.Cout % 2) == 0);       // the last arriver reads the partial table in 16-byte pieces
    MPN_CHECK_ARG(!p.fin_counters || (((p.stats != nullptr) != (p.bnb_partial != nullptr)) && p.fin_count > 0 && !p.nseg &&
                                      (p.stats ? p.fin_out != nullptr : true)));
    MPN_CHECK_ARG(!p.bnb_partial || (p.bnb_y && p.bnb_mean && p.bnb_invstd && !p.out_f32 && !p.nseg && !p.stats && !p.act &&
                                     (!p.bnb_relu || p.bnb_z || p.bnb_mask || (p.bnb_scale && p.bnb_shift)) &&
                                     (!p.bnb_mask || ((p.y_step > 1 || p.y_sB == (int64_t)p.Ho * p.Wo * p.y_sP) && p.y_sP == p.Cout_store))));
    if (p.y_step != 0 || p.w_taps != 0) {      // parity-class launch of a strided input gradient (mpn.h)
        MPN_CHECK_ARG(p.y_step == 2 && p.w_taps > 0 && !p.out_f32 && p.mode == 0 && p.stride == 1 && p.pad == 0 && !p.nseg && !p.kseg_n);
        MPN_CHECK_ARG(!p.res_mode && !p.fin_counters && !p.stats && !p.bias && !p.scale && !p.act);
        MPN_CHECK_ARG(p.y_oh >= 0 && p.y_oh < 2 && p.y_ow >= 0 && p.y_ow < 2 && p.R >= 1 && p.R <= 2 && p.S >= 1 && p.S <= 2);
        MPN_CHECK_ARG(p.y_H > 0 && p.y_W > 0 && 2 * (p.Ho - 1) + p.y_oh < p.y_H && 2 * (p.Wo - 1) + p.y_ow < p.y_W);
        MPN_CHECK_ARG(p.y_sB == (int64_t)p.y_H * p.y_W * p.y_sP && p.y_sP == p.Cout_store);
        for (int r = 0; r < p.R; ++r)
            for (int s = 0; s < p.S; ++s) {
                const int t = p.wtap0 + r * p.wtap_dr + s * p.wtap_ds;
                MPN_CHECK_ARG(t >= 0 && t < p.w_taps);
            }
    }
    {   // buffer descriptors address at most 4 GB per operand
        const int64_t ts = p.dtype == MPN_F32 ? 4 : 2;
        const int64_t row = (int64_t)(p.w_taps > 0 ? p.w_taps : p.R * p.S) * p.Cin * ts;
        int64_t xb = (int64_t)p.B * p.x_sB * ts;
        for (int l = 0; l < p.nseg; ++l) { const int64_t v = (int64_t)p.B * p.seg_H[l] * p.seg_W[l] * p.x_sW * ts; if (l == 0 || v > xb) xb = v; }
        if (xb + row >= 0xfffffff0LL || (int64_t)(p.Cout + 1) * row >= 0xfffffff0LL) return MPN_E_UNSUPPORTED;
        int64_t ye = (int64_t)p.B * p.y_sB;                       // the epilogue indexes output-shaped tensors with 32-bit element offsets
        for (int l = 0; l < p.nseg; ++l) { const int64_t v = (int64_t)p.B * p.seg_H[l] * p.seg_W[l] * p.y_sP; if (l == 0 || v > ye) ye = v; }
        if (ye >= 0x7fffffffLL) return MPN_E_UNSUPPORTED;
    }          // activation is applied before the residual stage
    hipStream_t st = (hipStream_t)stream;
    if (p.dtype == MPN_F32) return launch_conv<float, false>(p, st);       // OT == T == float
    if (p.dtype == MPN_F16) return p.out_f32 ? launch_conv<f16_t, true>(p, st) : launch_conv<f16_t, false>(p, st);
    return p.out_f32 ? launch_conv<bf16_t, true>(p, st) : launch_conv<bf16_t, false>(p, st);
}
