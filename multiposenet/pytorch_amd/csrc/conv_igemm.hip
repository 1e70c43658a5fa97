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
//   * 256 threads = 4 waves; block tile TC x 128 pixels; register-prefetched, double-buffered LDS
//     (80-byte rows: 64 B data + 16 B pad so the 16-lane ds_read_b128 groups spread over banks).
//   * bf16: v_mfma_f32_16x16x32_bf16 (fp32 accumulate); f32: v_mfma_f32_16x16x4_f32 (exact fp32,
//     bit-identical to an fmaf chain) — the f32 instantiation is the parity path.
//   * epilogue fuses per-channel scale, bias, residual (same-size or nearest-upsampled: the FPN
//     top-down add of fpn.py:84-95), ReLU / sigmoid, zero-fill of pad channels, and the per-tile
//     (sum, sum^2) partials BatchNorm needs in train mode.
// dgrad reuses the kernel with mode=1 (transposed gather hi = (ho + pad - r)/stride) and the
// [Cin][R][S][Cout_pad] weight copy made by mpn_weight_transpose.
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

constexpr int LDS_ROW = 80;

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    __device__ static __forceinline__ void run(f32x4_t& acc, const u32x4_t& a, const u32x4_t& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                      __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
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
        t.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
        t.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
        *reinterpret_cast<uint2*>(p) = t;
    }
    __device__ static __forceinline__ float round(float v) { return bf2f(f2bf(v)); }
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
    static constexpr int A_PER_T = (TC * 4 + 255) / 256;
    static constexpr int B_PER_T = (TP * 4 + 255) / 256;
    static constexpr int BUF_BYTES = (TC + TP) * LDS_ROW;
};

// Epilogue.  Phase A (accumulator layout: lane = 4 consecutive couts of one pixel): scale, bias, residual,
// accumulate, activation, BN partial sums.  Phase B: the finished tile goes through a per-wave LDS staging
// area and leaves as 16-byte stores with 8..16 lanes covering one pixel's contiguous channels (full 128-byte
// lines per wave-instruction) instead of 16 strided 8-byte stores per lane.
template <typename T, typename OT, int TC, int TP>
__device__ __forceinline__ void conv_epilogue(const MpnConvParams& p, f32x4_t (&acc)[ConvCfg<T, TC, TP>::MC][ConvCfg<T, TC, TP>::MP],
                                              int c0, long p0, int wc, int wp, int lane, int tp, unsigned char* lds) {
    using C = ConvCfg<T, TC, TP>;
    constexpr int OSZ = (int)sizeof(OT);
    constexpr int PASS_TILES = (OSZ == 4) ? ((C::MP >= 2) ? C::MP / 2 : 1) : C::MP;   // pixel tiles staged per pass
    constexpr int NPASS = C::MP / PASS_TILES;
    constexpr int SROW = C::WTC * OSZ + 16;                   // staging row stride (bytes)
    constexpr int REGION = PASS_TILES * 16 * SROW;            // per wave
    constexpr int LPP = C::WTC * OSZ / 16;                    // lanes per pixel in the store phase
    constexpr int PPI = 64 / LPP;                             // pixels per store instruction
    constexpr int EV = 16 / OSZ;                              // elements per 16-byte store
    static_assert(4 * REGION <= 2 * C::BUF_BYTES, "staging area must fit the main-loop LDS");
    const unsigned HoWo = (unsigned)p.Ho * (unsigned)p.Wo;
    const unsigned P = (unsigned)p.B * HoWo;           // launcher guarantees P < 2^31
    OT* __restrict__ Y = (OT*)p.y;
    const OT* __restrict__ Rz = (const OT*)p.res;
    const int lrow4 = (lane >> 4) * 4;
    const int lcol = lane & 15;
    float* lds_f = reinterpret_cast<float*>(lds);

    // ---- phase A: finish the values in registers -------------------------------------------------
    const bool simple = (p.res_mode == 0) && !p.accumulate;
#pragma unroll
    for (int j = 0; j < C::MP; ++j) {
        const unsigned pix = (unsigned)p0 + wp * C::WTP + j * 16 + lcol;
        const bool pok = pix < P;
        long yoff = 0, roff = 0;
        if (!simple) {
            const unsigned pc = pok ? pix : 0u;
            const unsigned b = pc / HoWo;
            const unsigned rem = pc - b * HoWo;
            yoff = (long)b * p.y_sB + (long)rem * p.y_sP;
            if (p.res_mode == 1) {
                roff = (long)b * p.res_sB + (long)rem * p.res_sP;
            } else if (p.res_mode == 2) {
                const unsigned ho = rem / (unsigned)p.Wo, wo = rem - ho * (unsigned)p.Wo;
                const unsigned rh = (ho * (unsigned)p.res_H) / (unsigned)p.Ho, rw = (wo * (unsigned)p.res_W) / (unsigned)p.Wo;
                roff = (long)b * p.res_sB + (long)(rh * (unsigned)p.res_W + rw) * p.res_sP;
            }
        }
#pragma unroll
        for (int i = 0; i < C::MC; ++i) {
            const int cout0 = c0 + wc * C::WTC + i * 16 + lrow4;
            float rv[4] = {0.f, 0.f, 0.f, 0.f};
            float yv[4] = {0.f, 0.f, 0.f, 0.f};
            const bool live = pok && cout0 < p.Cout_store;
            if (live && p.res_mode != 0) OutVec4<OT>::load(Rz + roff + cout0, rv);
            if (live && p.accumulate) OutVec4<OT>::load(Y + yoff + cout0, yv);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int cout = cout0 + c;
                float x = acc[i][j][c];
                if (cout < p.Cout) {
                    if (p.scale) x *= p.scale[cout];
                    if (p.bias) x += p.bias[cout];
                    x += rv[c];
                    x += yv[c];
                    if (p.act == 1) x = fmaxf(x, 0.f);
                    else if (p.act == 2) x = 1.0f / (1.0f + expf(-x));
                } else {
                    x = 0.f;
                }
                acc[i][j][c] = x;
            }
        }
    }

    if (p.stats) {
        // per-channel (sum, sum^2) of the stored values: over this lane's MP pixels, then over the 16 pixel
        // lanes that share (lane>>4) by xor-shuffles, then over the WAVES_P waves via LDS
#pragma unroll
        for (int i = 0; i < C::MC; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float a = 0.f, q = 0.f;
#pragma unroll
                for (int j = 0; j < C::MP; ++j) {
                    const bool pok = ((unsigned)p0 + wp * C::WTP + j * 16 + lcol) < P;
                    const float xr = pok ? OutVec4<OT>::round(acc[i][j][c]) : 0.f;
                    a += xr; q += xr * xr;
                }
#pragma unroll
                for (int m = 1; m < 16; m <<= 1) {
                    a += __shfl_xor(a, m, 64);
                    q += __shfl_xor(q, m, 64);
                }
                if (lcol == 0) {
                    const int row = wc * C::WTC + i * 16 + lrow4 + c;      // 0..TC-1
                    lds_f[(wp * TC + row) * 2 + 0] = a;
                    lds_f[(wp * TC + row) * 2 + 1] = q;
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
                    a += lds_f[(w * TC + t) * 2 + 0];
                    q += lds_f[(w * TC + t) * 2 + 1];
                }
                p.stats[((long)tp * p.Cout + cout) * 2 + 0] = a;
                p.stats[((long)tp * p.Cout + cout) * 2 + 1] = q;
            }
        }
        __syncthreads();
    }

    // ---- phase B: stage through LDS, store whole rows ----------------------------------------------
    unsigned char* stage = lds + (threadIdx.x >> 6) * REGION;
    const int sp = lane / LPP, sc = lane % LPP;                // store phase: pixel slot, 16-byte chunk
    const int ccol = c0 + wc * C::WTC + sc * EV;               // first channel of this lane's chunk
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
#pragma unroll
        for (int k = 0; k < PASS_TILES * 16 / PPI; ++k) {
            if (pix < P && ccol < p.Cout_store) {
                const u32x4_t v = *reinterpret_cast<const u32x4_t*>(stage + (k * PPI + sp) * SROW + sc * 16);
                *reinterpret_cast<u32x4_t*>(Y + (long)b * p.y_sB + (long)rem * p.y_sP + ccol) = v;
            }
            pix += PPI; rem += PPI;
            while (rem >= HoWo) { rem -= HoWo; ++b; }
        }
        if (ps + 1 < NPASS) __syncthreads();
    }
}

template <typename T, int TC, int TP>
__global__ void __launch_bounds__(256, 3) conv_igemm_kernel(const MpnConvParams p) {
    using C = ConvCfg<T, TC, TP>;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * C::BUF_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wc = wave / C::WAVES_P, wp = wave % C::WAVES_P;
    const int tilesC = (p.Cout_store + TC - 1) / TC;
    const unsigned HoWo = (unsigned)p.Ho * (unsigned)p.Wo;
    const unsigned P = (unsigned)p.B * HoWo;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tp = bid / tilesC, tc = bid - tp * tilesC;
    const long p0 = (long)tp * TP;
    const int c0 = tc * TC;
    const T* __restrict__ X = (const T*)p.x;
    const T* __restrict__ Wg = (const T*)p.w;
    const long KW = (long)p.R * p.S * p.Cin;
    const int sh = p.stride - 1;          // dgrad supports stride 1 or 2

    long a_off[C::A_PER_T];
    bool a_ok[C::A_PER_T];
    int a_lds[C::A_PER_T];
#pragma unroll
    for (int q = 0; q < C::A_PER_T; ++q) {
        const int u = tid + 256 * q;
        const int row = u >> 2, ch = u & 3;
        const int cout = c0 + row;
        a_ok[q] = (u < TC * 4) && (cout < p.Cout);
        a_off[q] = (long)cout * KW + ch * C::V;
        a_lds[q] = (u < TC * 4) ? row * LDS_ROW + ch * 16 : -1;
    }
    long b_base[C::B_PER_T];
    int b_h[C::B_PER_T], b_w[C::B_PER_T];
    bool b_ok[C::B_PER_T];
    int b_lds[C::B_PER_T];
#pragma unroll
    for (int q = 0; q < C::B_PER_T; ++q) {
        const int u = tid + 256 * q;
        const int row = u >> 2, ch = u & 3;
        const unsigned pix = (unsigned)p0 + row;
        const bool ok = (u < TP * 4) && (pix < P);
        const unsigned pc = ok ? pix : 0u;
        const unsigned b = pc / HoWo;
        const unsigned rem = pc - b * HoWo;
        const int ho = (int)(rem / (unsigned)p.Wo), wo = (int)(rem - (unsigned)ho * (unsigned)p.Wo);
        b_ok[q] = ok;
        if (p.mode == 0) { b_h[q] = ho * p.stride - p.pad; b_w[q] = wo * p.stride - p.pad; }
        else             { b_h[q] = ho + p.pad;            b_w[q] = wo + p.pad; }
        b_base[q] = (long)b * p.x_sB + ch * C::V;
        b_lds[q] = (u < TP * 4) ? (TC + row) * LDS_ROW + ch * 16 : -1;
    }

    f32x4_t acc[C::MC][C::MP];
#pragma unroll
    for (int i = 0; i < C::MC; ++i)
#pragma unroll
        for (int j = 0; j < C::MP; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const int nsteps = p.R * p.S * (p.Cin / C::KC);
    int r = 0, s = 0, cc = 0;
    long klin = 0;
    // two register stages: loads run TWO k-steps ahead of the MFMAs (global latency ~1 us per hop is the
    // critical path of a short k-step; one stage in flight while the other is written to LDS)
    u32x4_t ra0[C::A_PER_T], rb0[C::B_PER_T], ra1[C::A_PER_T], rb1[C::B_PER_T];
    const u32x4_t zero4 = (u32x4_t){0u, 0u, 0u, 0u};

    auto gload = [&](u32x4_t (&ra)[C::A_PER_T], u32x4_t (&rb)[C::B_PER_T]) {
#pragma unroll
        for (int q = 0; q < C::A_PER_T; ++q)
            ra[q] = a_ok[q] ? *reinterpret_cast<const u32x4_t*>(Wg + a_off[q] + klin) : zero4;
#pragma unroll
        for (int q = 0; q < C::B_PER_T; ++q) {
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
            const long off = b_base[q] + (long)hi * p.x_sH + (long)wi * p.x_sW + cc;
            rb[q] = ok ? *reinterpret_cast<const u32x4_t*>(X + off) : zero4;
        }
        // advance (tap, channel-chunk)
        klin += C::KC; cc += C::KC;
        if (cc == p.Cin) { cc = 0; if (++s == p.S) { s = 0; ++r; } }
    };
    auto lstore = [&](int buf, const u32x4_t (&ra)[C::A_PER_T], const u32x4_t (&rb)[C::B_PER_T]) {
        unsigned char* base = lds + buf * C::BUF_BYTES;
#pragma unroll
        for (int q = 0; q < C::A_PER_T; ++q)
            if (a_lds[q] >= 0) *reinterpret_cast<u32x4_t*>(base + a_lds[q]) = ra[q];
#pragma unroll
        for (int q = 0; q < C::B_PER_T; ++q)
            if (b_lds[q] >= 0) *reinterpret_cast<u32x4_t*>(base + b_lds[q]) = rb[q];
    };
    const int fa_off = (wc * C::WTC + (lane & 15)) * LDS_ROW + (lane >> 4) * 16;
    const int fb_off = (TC + wp * C::WTP + (lane & 15)) * LDS_ROW + (lane >> 4) * 16;
    auto compute = [&](int buf) {
        const unsigned char* base = lds + buf * C::BUF_BYTES;
        u32x4_t fa[C::MC], fb[C::MP];
#pragma unroll
        for (int i = 0; i < C::MC; ++i)
            fa[i] = *reinterpret_cast<const u32x4_t*>(base + fa_off + i * 16 * LDS_ROW);
#pragma unroll
        for (int j = 0; j < C::MP; ++j)
            fb[j] = *reinterpret_cast<const u32x4_t*>(base + fb_off + j * 16 * LDS_ROW);
#pragma unroll
        for (int i = 0; i < C::MC; ++i)
#pragma unroll
            for (int j = 0; j < C::MP; ++j) Mma<T>::run(acc[i][j], fa[i], fb[j]);
    };

    gload(ra0, rb0);
    if (nsteps > 1) gload(ra1, rb1);
    lstore(0, ra0, rb0);
    __syncthreads();
    int it = 0;
    for (; it + 1 < nsteps; it += 2) {
        if (it + 2 < nsteps) gload(ra0, rb0);       // step it+2
        compute(0);                                  // step it
        lstore(1, ra1, rb1);                         // step it+1 (loaded one half-iteration ago)
        __syncthreads();
        if (it + 3 < nsteps) gload(ra1, rb1);       // step it+3
        compute(1);                                  // step it+1
        if (it + 2 < nsteps) lstore(0, ra0, rb0);   // step it+2
        __syncthreads();
    }
    if (it < nsteps) compute(0);
    __syncthreads();

    if (p.out_f32) conv_epilogue<T, float, TC, TP>(p, acc, c0, p0, wc, wp, lane, tp, lds);
    else           conv_epilogue<T, T, TC, TP>(p, acc, c0, p0, wc, wp, lane, tp, lds);
}

constexpr int kTP = 128;

inline int pick_tc(int cout_store, long tilesP) {
    if (cout_store <= 32) return 32;
    if (cout_store <= 64) return 64;
    // 128-row tiles unless that leaves the 256 CUs with fewer than ~3 workgroups each
    const long blocks128 = tilesP * ((cout_store + 127) / 128);
    return blocks128 >= 768 ? 128 : 64;
}

template <typename T>
int launch_conv(const MpnConvParams& p, hipStream_t st) {
    const long P = (long)p.B * p.Ho * p.Wo;
    const long tilesP = (P + kTP - 1) / kTP;
    const int tc = pick_tc(p.Cout_store, tilesP);
    const long tilesC = (p.Cout_store + tc - 1) / tc;
    const long grid = tilesP * tilesC;
    if (grid <= 0 || grid > 0x7fffffffL || P >= 0x7fffffffL) return MPN_E_BADARG;
    if (tc == 128) hipLaunchKernelGGL((conv_igemm_kernel<T, 128, kTP>), dim3((unsigned)grid), dim3(256), 0, st, p);
    else if (tc == 64) hipLaunchKernelGGL((conv_igemm_kernel<T, 64, kTP>), dim3((unsigned)grid), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((conv_igemm_kernel<T, 32, kTP>), dim3((unsigned)grid), dim3(256), 0, st, p);
    return mpn_launch_status();
}

}  // namespace

extern "C" int mpn_conv_stats_tiles(const MpnConvParams* p) {
    if (!p) return MPN_E_BADARG;
    const long P = (long)p->B * p->Ho * p->Wo;
    return (int)((P + kTP - 1) / kTP);
}

extern "C" int mpn_conv_tile_rows(const MpnConvParams* p) {
    if (!p) return MPN_E_BADARG;
    const long P = (long)p->B * p->Ho * p->Wo;
    return pick_tc(p->Cout_store, (P + kTP - 1) / kTP);
}

extern "C" int mpn_conv_forward(const MpnConvParams* pp, void* stream) {
    if (!pp) return MPN_E_BADARG;
    const MpnConvParams& p = *pp;
    MPN_CHECK_ARG(p.x && p.w && p.y);
    MPN_CHECK_ARG(p.B > 0 && p.Ho > 0 && p.Wo > 0 && p.H > 0 && p.W > 0);
    MPN_CHECK_ARG(p.dtype == MPN_F32 || p.dtype == MPN_BF16);
    const int kc = p.dtype == MPN_F32 ? 16 : 32;
    MPN_CHECK_ARG(p.Cin > 0 && p.Cin % kc == 0);
    MPN_CHECK_ARG(p.Cout > 0 && p.Cout_store >= p.Cout && p.Cout_store % 4 == 0);
    MPN_CHECK_ARG(p.R > 0 && p.S > 0 && p.stride >= 1);
    MPN_CHECK_ARG(p.mode == 0 || (p.mode == 1 && (p.stride == 1 || p.stride == 2)));
    MPN_CHECK_ARG(!(p.accumulate && p.act != 0));
    MPN_CHECK_ARG(p.res_mode == 0 || p.res != nullptr);
    MPN_CHECK_ARG(!(p.stats && (p.bias || p.scale || p.res_mode || p.accumulate || p.act)));
    hipStream_t st = (hipStream_t)stream;
    if (p.dtype == MPN_F32) return launch_conv<float>(p, st);
    return launch_conv<bf16_t>(p, st);
}
