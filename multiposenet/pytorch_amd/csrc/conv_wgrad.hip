// conv_wgrad.hip — weight-gradient convolution on MFMA for gfx950.
//
// Replaces the weight half of autograd's conv backward (training/trainer.py:251) for every
// nn.Conv2d / nn.Linear of network/fpn.py and network/posenet.py.
//
//   dW[cout][r][s][cin] += sum_{pixels p} dY[p][cout] * X[gather(p, r, s)][cin]
//
// GEMM view per tap (r,s):  D[cin][cout] = sum_p X^T[cin][p] * dY[p][cout]  (contraction = pixels)
//   * MFMA "A" rows = cin, "B" cols = cout  -> each lane owns 4 consecutive cin of one cout, i.e.
//     a float4 of the [Cout][R][S][Cin] gradient (the master-weight memory layout).
//   * both operands arrive pixel-major (channels contiguous), i.e. K-strided.  f32 path: the
//     16x16x4 f32 MFMA takes ONE k per lane, so tiles stay [k][channel] in LDS and fragments are
//     conflict-free ds_read_b32.  bf16 path: 16x16x32 wants 8 consecutive k per lane; two pixel rows
//     are interleaved in registers into (k, k+1) dwords and written channel-major ([channel][k],
//     80-byte rows) so fragments are one ds_read_b128 exactly like the forward kernel.
//   * the pixel contraction is split into `chunks` slices across workgroups; slices write f32
//     partials to a workspace and mpn_reduce_partials adds them into dW in a fixed order
//     (deterministic; no atomics).  chunks == 1 accumulates straight into dW.
#include "common.h"
#include <stdlib.h>

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

// 16x16x32 MFMA on 8 packed 16-bit values per lane, by element type
template <typename T> __device__ __forceinline__ f32x4_t mma16(const u32x4_t& a, const u32x4_t& b, const f32x4_t& acc);
template <> __device__ __forceinline__ f32x4_t mma16<bf16_t>(const u32x4_t& a, const u32x4_t& b, const f32x4_t& acc) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4_t mma16<f16_t>(const u32x4_t& a, const u32x4_t& b, const f32x4_t& acc) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), acc, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4_t mma16<float>(const u32x4_t&, const u32x4_t&, const f32x4_t& acc) { return acc; }   // never taken
template <typename T> struct Ones16;
template <> struct Ones16<bf16_t> { static constexpr unsigned kPair = 0x3F803F80u; };
template <> struct Ones16<f16_t> { static constexpr unsigned kPair = 0x3C003C00u; };
template <> struct Ones16<float> { static constexpr unsigned kPair = 0u; };        // unused: the f32 bias sum multiplies by a scalar 1.0f

struct PixState {      // incremental (b, ho, wo) walker over the dense output-pixel index
    int b, ho, wo;
    __device__ __forceinline__ void init(long pix, int Ho, int Wo) {      // pix < 2^31 (checked by the launcher)
        const unsigned hw = (unsigned)Ho * (unsigned)Wo;
        const unsigned px = (unsigned)pix;
        const unsigned bb = px / hw;
        const unsigned rem = px - bb * hw;
        const unsigned hh = rem / (unsigned)Wo;
        b = (int)bb; ho = (int)hh; wo = (int)(rem - hh * (unsigned)Wo);
    }
    __device__ __forceinline__ void advance(int n, int Ho, int Wo) {
        wo += n;
        while (wo >= Wo) { wo -= Wo; if (++ho == Ho) { ho = 0; ++b; } }
    }
    // branch-free advance by a fixed pixel count pre-split into mixed-radix digits (db, dh, dw): n = (db*Ho + dh)*Wo + dw
    __device__ __forceinline__ void advance_digits(int db, int dh, int dw, int Ho, int Wo) {
        wo += dw;
        const int c0 = wo >= Wo ? 1 : 0;
        wo -= c0 ? Wo : 0;
        ho += dh + c0;
        const int c1 = ho >= Ho ? 1 : 0;
        ho -= c1 ? Ho : 0;
        b += db + c1;
    }
    __device__ __forceinline__ PixState next(int Ho, int Wo) const {
        PixState q = *this;
        if (++q.wo == Wo) { q.wo = 0; if (++q.ho == Ho) { q.ho = 0; ++q.b; } }
        return q;
    }
};

template <typename T, int TM, int TN>
struct WgCfg {
    static constexpr bool kBf16 = sizeof(T) == 2;
    static constexpr int KP = kBf16 ? 32 : 16;            // pixels per k-step
    static constexpr int WTM = TM / 2, WTN = TN / 2;      // 2 x 2 waves
    static constexpr int MM = (WTM + 15) / 16, MN = (WTN + 15) / 16;
    static constexpr int RS_A = kBf16 ? 80 : (TM + 16) * 4;   // LDS row stride (bytes)
    static constexpr int RS_B = kBf16 ? 80 : (TN + 16) * 4;
    static constexpr int A_BYTES = kBf16 ? TM * 80 : 16 * RS_A;
    static constexpr int B_BYTES = kBf16 ? TN * 80 : 16 * RS_B;
    static constexpr int BUF_BYTES = A_BYTES + B_BYTES;
    // load units per operand: bf16 -> (16 k-pairs) x (TM/8 groups); f32 -> (16 k) x (TM/4 vecs)
    static constexpr int UA = kBf16 ? 16 * (TM / 8) : 16 * (TM / 4);
    static constexpr int UB = kBf16 ? 16 * (TN / 8) : 16 * (TN / 4);
    static constexpr int A_PER_T = (UA + 255) / 256;
    static constexpr int B_PER_T = (UB + 255) / 256;
};

template <typename T, int TM, int TN>
__global__ void __launch_bounds__(256, 3) conv_wgrad_kernel(const MpnWgradParams p, long chunk_pixels) {
    using C = WgCfg<T, TM, TN>;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * C::BUF_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tilesM = (p.Cin + TM - 1) / TM, tilesN = (p.Cout + TN - 1) / TN;
    const int taps = p.R * p.S;
    // XCD-aware order: the tile-blocks of one pixel slice (which re-read the same X / dY rows) share an L2
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tn = bid % tilesN; bid /= tilesN;
    const int tm = bid % tilesM; bid /= tilesM;
    const int tap = bid % taps; bid /= taps;
    const int chunk = bid;
    const int r = tap / p.S, s = tap - r * p.S;
    const int m0 = tm * TM, n0 = tn * TN;
    const long P = (long)p.B * p.Ho * p.Wo;
    const long k_begin = (long)chunk * chunk_pixels;
    long k_end = k_begin + chunk_pixels; if (k_end > P) k_end = P;
    const T* __restrict__ X = (const T*)p.x;
    const T* __restrict__ DY = (const T*)p.dy;
    const int dy_cs = ((p.Cout + 31) / 32) * 32;    // dY channel storage (pad lanes are zero)
    constexpr int V = 16 / (int)sizeof(T);

    // ---------------- per-thread load descriptors ----------------
    // A (X, gathered): unit -> (k index within step, channel offset)
    int a_k[C::A_PER_T], a_c[C::A_PER_T]; bool a_on[C::A_PER_T]; PixState a_px[C::A_PER_T];
    int b_k[C::B_PER_T], b_c[C::B_PER_T]; bool b_on[C::B_PER_T];
#pragma unroll
    for (int q = 0; q < C::A_PER_T; ++q) {
        const int u = tid + 256 * q;
        if (C::kBf16) { a_k[q] = (u & 15) * 2; a_c[q] = (u >> 4) * 8; }
        else          { a_k[q] = u / (TM / 4); a_c[q] = (u % (TM / 4)) * 4; }
        a_on[q] = (u < C::UA) && (m0 + a_c[q] < p.Cin);
        long pix = k_begin + a_k[q]; if (pix >= P) pix = P - 1;
        a_px[q].init(pix, p.Ho, p.Wo);
    }
#pragma unroll
    for (int q = 0; q < C::B_PER_T; ++q) {
        const int u = tid + 256 * q;
        if (C::kBf16) { b_k[q] = (u & 15) * 2; b_c[q] = (u >> 4) * 8; }
        else          { b_k[q] = u / (TN / 4); b_c[q] = (u % (TN / 4)) * 4; }
        b_on[q] = (u < C::UB) && (n0 + b_c[q] < dy_cs);
    }

    f32x4_t acc[C::MM][C::MN];
#pragma unroll
    for (int i = 0; i < C::MM; ++i)
#pragma unroll
        for (int j = 0; j < C::MN; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    constexpr int NR = C::kBf16 ? 2 : 1;      // pixel rows per unit
    u32x4_t ra0[C::A_PER_T][NR], rb0[C::B_PER_T][NR], ra1[C::A_PER_T][NR], rb1[C::B_PER_T][NR];   // loads run two k-steps ahead
    const u32x4_t zero4 = (u32x4_t){0u, 0u, 0u, 0u};
    long k0 = k_begin;

    auto xload = [&](const PixState& ps, long pix, int c) -> u32x4_t {
        const int hi = ps.ho * p.stride - p.pad + r, wi = ps.wo * p.stride - p.pad + s;
        const bool ok = (pix < k_end) && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
        const long off = (long)ps.b * p.x_sB + (long)hi * p.x_sH + (long)wi * p.x_sW + m0 + c;
        return ok ? *reinterpret_cast<const u32x4_t*>(X + off) : zero4;
    };
    auto gload = [&](u32x4_t (&ra)[C::A_PER_T][NR], u32x4_t (&rb)[C::B_PER_T][NR]) {
#pragma unroll
        for (int q = 0; q < C::A_PER_T; ++q) {
            if (a_on[q]) {
                const long pix = k0 + a_k[q];
                ra[q][0] = xload(a_px[q], pix, a_c[q]);
                if (NR == 2) ra[q][NR - 1] = xload(a_px[q].next(p.Ho, p.Wo), pix + 1, a_c[q]);
                a_px[q].advance(C::KP, p.Ho, p.Wo);
            } else {
#pragma unroll
                for (int e = 0; e < NR; ++e) ra[q][e] = zero4;
            }
        }
#pragma unroll
        for (int q = 0; q < C::B_PER_T; ++q) {
#pragma unroll
            for (int e = 0; e < NR; ++e) {
                const long pix = k0 + b_k[q] + e;
                const bool ok = b_on[q] && pix < k_end;
                rb[q][e] = ok ? *reinterpret_cast<const u32x4_t*>(DY + pix * p.dy_sP + n0 + b_c[q]) : zero4;
            }
        }
        k0 += C::KP;
    };
    auto lstore = [&](int buf, const u32x4_t (&ra)[C::A_PER_T][NR], const u32x4_t (&rb)[C::B_PER_T][NR]) {
        unsigned char* la = lds + buf * C::BUF_BYTES;
        unsigned char* lb = la + C::A_BYTES;
        if (C::kBf16) {
#pragma unroll
            for (int q = 0; q < C::A_PER_T; ++q) {
                if (tid + 256 * q < C::UA) {
                    const u32x4_t e = ra[q][0], o = ra[q][NR - 1];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const unsigned ew = e[i >> 1], ow = o[i >> 1];
                        const unsigned v = (i & 1) ? ((ew >> 16) | (ow & 0xffff0000u)) : ((ew & 0xffffu) | (ow << 16));
                        *reinterpret_cast<unsigned*>(la + (a_c[q] + i) * 80 + a_k[q] * 2) = v;
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < C::B_PER_T; ++q) {
                if (tid + 256 * q < C::UB) {
                    const u32x4_t e = rb[q][0], o = rb[q][NR - 1];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const unsigned ew = e[i >> 1], ow = o[i >> 1];
                        const unsigned v = (i & 1) ? ((ew >> 16) | (ow & 0xffff0000u)) : ((ew & 0xffffu) | (ow << 16));
                        *reinterpret_cast<unsigned*>(lb + (b_c[q] + i) * 80 + b_k[q] * 2) = v;
                    }
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < C::A_PER_T; ++q)
                if (tid + 256 * q < C::UA)
                    *reinterpret_cast<u32x4_t*>(la + a_k[q] * C::RS_A + a_c[q] * 4) = ra[q][0];
#pragma unroll
            for (int q = 0; q < C::B_PER_T; ++q)
                if (tid + 256 * q < C::UB)
                    *reinterpret_cast<u32x4_t*>(lb + b_k[q] * C::RS_B + b_c[q] * 4) = rb[q][0];
        }
    };

    const long span = k_end - k_begin;
    const int nsteps = span > 0 ? (int)((span + C::KP - 1) / C::KP) : 0;
    auto compute = [&](int buf) {
        const unsigned char* la = lds + buf * C::BUF_BYTES;
        const unsigned char* lb = la + C::A_BYTES;
        if (C::kBf16) {
            u32x4_t fa[C::MM], fb[C::MN];
#pragma unroll
            for (int i = 0; i < C::MM; ++i)
                fa[i] = *reinterpret_cast<const u32x4_t*>(la + (wm * C::WTM + i * 16 + (lane & 15)) * 80 + (lane >> 4) * 16);
#pragma unroll
            for (int j = 0; j < C::MN; ++j)
                fb[j] = *reinterpret_cast<const u32x4_t*>(lb + (wn * C::WTN + j * 16 + (lane & 15)) * 80 + (lane >> 4) * 16);
#pragma unroll
            for (int i = 0; i < C::MM; ++i)
#pragma unroll
                for (int j = 0; j < C::MN; ++j)
                    acc[i][j] = mma16<T>(fa[i], fb[j], acc[i][j]);
        } else {
#pragma unroll
            for (int kq = 0; kq < 4; ++kq) {
                float fa[C::MM], fb[C::MN];
                const int krow = kq * 4 + (lane >> 4);
#pragma unroll
                for (int i = 0; i < C::MM; ++i)
                    fa[i] = *reinterpret_cast<const float*>(la + krow * C::RS_A + (wm * C::WTM + i * 16 + (lane & 15)) * 4);
#pragma unroll
                for (int j = 0; j < C::MN; ++j)
                    fb[j] = *reinterpret_cast<const float*>(lb + krow * C::RS_B + (wn * C::WTN + j * 16 + (lane & 15)) * 4);
#pragma unroll
                for (int i = 0; i < C::MM; ++i)
#pragma unroll
                    for (int j = 0; j < C::MN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
            }
        }
    };
    if (nsteps > 0) {
        gload(ra0, rb0);
        if (nsteps > 1) gload(ra1, rb1);
        lstore(0, ra0, rb0);
    }
    __syncthreads();
    int it = 0;
    for (; it + 1 < nsteps; it += 2) {
        if (it + 2 < nsteps) gload(ra0, rb0);
        compute(0);
        lstore(1, ra1, rb1);
        __syncthreads();
        if (it + 3 < nsteps) gload(ra1, rb1);
        compute(1);
        if (it + 2 < nsteps) lstore(0, ra0, rb0);
        __syncthreads();
    }
    if (it < nsteps) compute(0);

    // ---------------- epilogue ----------------
    const long NW = (long)p.Cout * taps * p.Cin;
    float* __restrict__ dst = (p.chunks > 1) ? (p.ws + (long)chunk * NW) : p.dw;
    const bool add = (p.chunks == 1);
#pragma unroll
    for (int i = 0; i < C::MM; ++i) {
        const int ml = wm * C::WTM + i * 16 + (lane >> 4) * 4;
        const int cin = m0 + ml;
        if (cin >= p.Cin) continue;
#pragma unroll
        for (int j = 0; j < C::MN; ++j) {
            const int nl = wn * C::WTN + j * 16 + (lane & 15);
            const int cout = n0 + nl;
            if (cout >= p.Cout) continue;
            float* q = dst + ((long)cout * taps + tap) * p.Cin + cin;
            float4 v = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            if (add) {
                const float4 o = *reinterpret_cast<const float4*>(q);
                v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
            }
            *reinterpret_cast<float4*>(q) = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// bf16 fast path: LDS-DMA + CDNA4 LDS transpose read.  Both operands arrive pixel-major ([k][channel]) while the
// MFMA wants 8 consecutive k per lane.  Instead of re-packing in registers (generic kernel above), tiles land in
// LDS exactly as they lie in memory and fragments are gathered with ds_read_b64_tr_b16, which hands lane i column
// i of a 4(k) x 16(channel) block.  Tiles go HBM -> LDS with `buffer_load_dwordx4 ... lds` (no staging registers,
// no ds_write pass) into a 3-deep ring, two k-steps ahead of the MFMAs.  The DMA destination is lane-linear
// (M0 base + lane*16), so the bank swizzle of the tile image is applied on the SOURCE side: the lane that owns LDS
// slot j of row k fetches channel chunk j ^ swz(k).  Out-of-range lanes (halo taps, channel tail, pixels past the
// slice) carry an offset beyond num_records and the hardware writes zeros — the dY descriptor is clipped to the
// slice end, so tail pixels need no per-step mask.  The loads are inline asm, invisible to the compiler's waitcnt
// bookkeeping: completion is counted by hand (`s_waitcnt vmcnt(n)`, n = one k-step's loads per wave: everything but
// the newest k-step has landed) and the barrier is the raw s_barrier, so the ring never drains inside the loop.
// (semantics of the DMA and of the transpose read were pinned with tools/probe_dma.* and tools/probe_tr.*)
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
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

constexpr unsigned DMA_OOB = 0x80000000u;     // tensors are < 2 GB (launcher check): marker + soffset never wraps

// Tile image in LDS: 32 pixel rows of ROWB = tile_channels * 2 bytes, back to back.  ds_read_b64_tr_b16 is serviced in
// two 32-lane halves; one half touches 8 rows (k = {0..3} + 8*{0,1} + 4h) x 32 bytes, which must land on 8 distinct
// 32-byte bank slots of the 256-byte bank row.  256-byte rows: every row starts on slot 0, so the 32-byte piece index
// is XORed with (k&3)|((k>>3)&1)<<2 (512-byte rows likewise: the row stride is a multiple of the bank row).  128-byte rows: rows k and k+1 already differ by 4 slots, the remaining four
// rows of equal parity are separated by XORing the piece index with ((k>>1)&1)|((k>>3)&1)<<1.  Swizzles are in
// 16-byte chunk units (piece << 1) and are applied on the DMA source side.
template <int ROWB> __device__ __forceinline__ int dma_swz(int k) {
    return ROWB >= 256 ? 2 * ((k & 3) | (((k >> 3) & 1) << 2)) : 2 * (((k >> 1) & 1) | (((k >> 3) & 1) << 1));
}

// f32 tiles (round 5): the exact-fp32 MFMA takes ONE k per lane, so fragments are ds_read_b32 — 16 lanes read 16 consecutive channels
// (one 64-byte group) of a row, the four 16-lane groups of an instruction read rows k, k+1, k+2, k+3.  Rows are 256 / 512 bytes (a
// multiple of the 256-byte bank row), so the four rows would meet on the same banks: the 64-byte group index is XORed with k & 3
// (16-byte chunk units: (k & 3) << 2), which puts the four groups of every instruction on four distinct quarter bank rows.
template <typename T, int ROWB> __device__ __forceinline__ int dma_swz_t(int k) {
    if (sizeof(T) == 4) return (k & 3) << 2;
    return dma_swz<ROWB>(k);
}

__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ const void* rfl_ptr(const void* q) {
    const uint64_t a = (uint64_t)q;
    return (const void*)(((uint64_t)(unsigned)rfl((int)(a >> 32)) << 32) | (unsigned)rfl((int)a));
}

// SEG = pyramid mode (MpnWgradParams::nseg > 0): a separate instantiation, so the single-tensor kernels keep their register
// allocation (the level lookup costs ~30 VGPRs of address arithmetic that the compiler no longer proves uniform)
// PROF (tools/kloop_profile.py only): a separate instantiation that accumulates, per wave, the s_memtime cycles spent in the four phases
// of a k-step — waiting for its own DMA (s_waitcnt vmcnt), waiting at the barrier, issuing the next k-step's DMA, fragment reads + MFMA
// issue — and writes them to `prof` [workgroup][wave][8] at the end.  Production instantiations compile none of it.
// What it showed (profiles/r04_kloop_phase_profile.txt): a k-step of a wave takes ~980 cycles — ~10 waiting for its DMA, ~45 at the
// barrier, ~365 ISSUING four buffer_load ... lds (back-pressure of the CU's one texture path, 16 KB per k-step and workgroup), ~570 in
// fragment reads + 16 MFMAs (256 cycles of matrix pipe).  Issuing the DMA between the MFMAs instead (built, measured, removed) moves
// the stalls into the MFMA phase and leaves the k-step at ~920 cycles: the loop is bound by the texture path's ~45 B/clk, i.e. by the
// tile's 64 FLOP per DMA byte, not by latency (the ring covers it) or by the wave's instruction order.
// LIN (round 4): stride-1 "same" convolutions over a dense x (the launcher checks: wgrad_lin_ok) read x at pixel + tap offset — LINEAR in the
// output pixel.  The tap offset moves into the buffer descriptor's base, the k-step advance into the scalar offset, and what is left per DMA
// instruction is the halo predicate on incrementally tracked (ho, wo) (nothing at all for 1x1): ~35 VALU instructions per k-step and wave
// instead of ~65 with ten quarter-rate v_mul_lo_u32 (the general gather: strided / virtually concatenated / up-sampled operands).
template <typename T, int TM, int TN, bool SEG, bool PROF = false, bool LIN = false>
__device__ __forceinline__ void conv_wgrad_dma_body(const MpnWgradParams& pk, long chunk_pixels, unsigned long long* prof = nullptr) {
    const bool ablate_stores = (chunk_pixels >> 62) & 1;     // MPN_WGRAD_ABLATE=2 (tools only): how much do the partial stores cost?
    chunk_pixels &= ~(1L << 62);
    constexpr int ES = (int)sizeof(T), EV = 16 / ES;         // element bytes, elements per 16-byte DMA chunk
    constexpr int KP = 64 / ES, NST = 3;                     // pixels per k-step: 32 (16-bit) / 16 (f32) — 16 KB per stage at 128 x 128 either way
    constexpr int ROWA = TM * ES, ROWB = TN * ES;            // bytes per pixel row of each tile
    constexpr int A_BYTES = KP * ROWA, B_BYTES = KP * ROWB;
    constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int QA = A_BYTES / 4096, QB = B_BYTES / 4096;  // DMA instructions (1 KiB each) per wave per k-step
    constexpr int RPA = 1024 / ROWA, RPB = 1024 / ROWB;      // tile rows covered by one instruction
    constexpr int MM = TM / 32, MN = TN / 32;                // 16x16 fragments per wave (2 x 2 waves)
    static_assert((TM == 256 || TM == 128 || TM == 64) && (TN == 128 || TN == 64), "tile widths");
    static_assert(ES == 2 || (TM <= 128 && !SEG), "f32: 64 / 128-wide tiles, single tensor");
    __shared__ __attribute__((aligned(16))) unsigned char lds[NST * STAGE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tilesM = (pk.Cin + TM - 1) / TM, tilesN = (pk.Cout + TN - 1) / TN;
    const int taps = pk.R * pk.S;
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tn = bid % tilesN; bid /= tilesN;
    const int tm = bid % tilesM; bid /= tilesM;
    const int tap = bid % taps; bid /= taps;
    const int chunk = bid;                        // global slice index (partial-sum slot)
    MpnWgradParams p = pk;                        // working copy; pyramid mode patches in the level's tensors
    int chunk_local = chunk;
    if (SEG) {
        int l = 0;
#pragma unroll
        for (int k = 1; k < 5; ++k) l += (k < pk.nseg && chunk >= pk.seg_chunk0[k]) ? 1 : 0;
        l = rfl(l);
        chunk_local = rfl(chunk - pk.seg_chunk0[l]);
        p.x = rfl_ptr(pk.seg_x[l]); p.dy = rfl_ptr(pk.seg_dy[l]);
        p.H = p.Ho = rfl(pk.seg_H[l]); p.W = p.Wo = rfl(pk.seg_W[l]);
        p.x_sH = (int64_t)p.W * pk.x_sW; p.x_sB = (int64_t)p.H * p.x_sH;
    }
    const int r = tap / p.S, s = tap - r * p.S;
    const int m0 = tm * TM, n0 = tn * TN;
    const long P = (long)p.B * p.Ho * p.Wo;
    const long k_begin = (long)chunk_local * chunk_pixels;
    long k_end = k_begin + chunk_pixels; if (k_end > P) k_end = P;
    const int dy_cs = ((p.Cout + 31) / 32) * 32;
    // virtual channel concatenation of x (mpn.h: kseg_*): this workgroup's cin tile is ONE member — a dense, nearest-up-sampled
    // tensor of kseg_c channels read at (hi >> vsh, wi >> vsh)
    int vsh = 0, vc0 = 0;
    if (!SEG && pk.kseg_n > 0) {
        const int sg = rfl(m0 / pk.kseg_c);
        vsh = rfl(pk.kseg_shift[sg]);
        vc0 = sg * pk.kseg_c;
        p.x = rfl_ptr(pk.kseg_x[sg]);
        p.x_sW = pk.kseg_c;
        p.x_sH = (int64_t)(p.W >> vsh) * p.x_sW;
        p.x_sB = (int64_t)(p.H >> vsh) * p.x_sH;
    }

    const int dr = r - p.pad, dsx = s - p.pad;                       // LIN: tap offset in lines / pixels
    const long tap_bytes = LIN ? ((long)dr * p.W + dsx) * p.x_sW * ES : 0;      // base moves by the tap; a valid (ho + dr, wo + dsx) never reads before x
    long x_bytes_l = (long)p.B * p.x_sB * ES - tap_bytes;
    if (x_bytes_l < 0) x_bytes_l = 0;
    if (x_bytes_l > 0x7fffffffL) x_bytes_l = 0x7fffffffL;
    const i32x4_t rsrc_x = make_rsrc((const char*)p.x + tap_bytes, (unsigned)x_bytes_l);
    const bool halo = LIN && (p.R > 1 || p.S > 1 || p.pad != 0);     // uniform: 1x1 needs no predicate and no pixel digits
    const i32x4_t rsrc_dy = make_rsrc(p.dy, (unsigned)((k_end > k_begin ? k_end : 0) * (long)p.dy_sP * ES));
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const unsigned wave_u = (unsigned)__builtin_amdgcn_readfirstlane(wave);

    // DMA units: instruction q of wave w fills rows (w*Q + q)*RP .. +RP-1 of a tile; lane -> row (lane / chunks-per-row),
    // slot (lane % chunks-per-row); the slot's source chunk is slot ^ swz(row)
    unsigned b_voff[QB]; int a_chan[QA]; bool a_on[QA]; PixState a_px[QA];
#pragma unroll
    for (int q = 0; q < QA; ++q) {
        const int row = ((int)wave_u * QA + q) * RPA + lane / (ROWA / 16);
        const int chan = ((lane % (ROWA / 16)) ^ dma_swz_t<T, ROWA>(row)) * EV;
        a_chan[q] = m0 + chan;
        a_on[q] = (m0 + chan) < p.Cin;
        long pix = k_begin + row;
        if (LIN) {           // a_chan holds the lane's byte offset from the tap-shifted base (rows past the end run out of the descriptor or meet zero dY rows)
            a_chan[q] = a_on[q] ? (int)(unsigned)((pix * p.x_sW + m0 + chan) * ES) : (int)DMA_OOB;
        } else if (pix >= P) pix = P - 1;     // beyond-the-end rows meet zero dY rows
        a_px[q].init(pix, p.Ho, p.Wo);
    }
#pragma unroll
    for (int q = 0; q < QB; ++q) {
        const int row = ((int)wave_u * QB + q) * RPB + lane / (ROWB / 16);
        const int chan = ((lane % (ROWB / 16)) ^ dma_swz_t<T, ROWB>(row)) * EV;
        b_voff[q] = ((n0 + chan) < dy_cs) ? (unsigned)(((k_begin + row) * (long)p.dy_sP + n0 + chan) * ES) : DMA_OOB;
    }
    const unsigned b_step = (unsigned)__builtin_amdgcn_readfirstlane((int)(KP * p.dy_sP * ES));
    unsigned b_soff = 0;
    const unsigned a_step = (unsigned)__builtin_amdgcn_readfirstlane((int)(KP * p.x_sW * ES));
    unsigned a_soff = 0;
    unsigned a_sel[QA];                  // LIN: the lane's offset or the out-of-range marker for the k-step about to be queued
#pragma unroll
    for (int q = 0; q < QA; ++q) {
        const bool ok = !halo || ((unsigned)(a_px[q].ho + dr) < (unsigned)p.H && (unsigned)(a_px[q].wo + dsx) < (unsigned)p.W);
        a_sel[q] = ok ? (unsigned)a_chan[q] : DMA_OOB;
    }
    const int adv_t = KP / p.Wo, adv_w = KP - adv_t * p.Wo, adv_b = adv_t / p.Ho, adv_h = adv_t - adv_b * p.Ho;

    f32x4_t acc[MM][MN];
#pragma unroll
    for (int i = 0; i < MM; ++i)
#pragma unroll
        for (int j = 0; j < MN; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    auto issue = [&](unsigned stage) {       // queue one k-step (QA + QB DMA instructions per wave) into ring slot `stage`
        const unsigned st = lds_base + stage * STAGE_BYTES;
        if constexpr (LIN) {
#pragma unroll
            for (int q = 0; q < QA; ++q)
                lds_dma16(a_sel[q], rsrc_x, a_soff, __builtin_amdgcn_readfirstlane(st + (wave_u * QA + q) * 1024u));
            a_soff += a_step;
        } else
#pragma unroll
        for (int q = 0; q < QA; ++q) {
            const int hi = a_px[q].ho * p.stride - p.pad + r, wi = a_px[q].wo * p.stride - p.pad + s;
            const bool ok = a_on[q] && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
            const unsigned off = ((unsigned)a_px[q].b * (unsigned)p.x_sB + (unsigned)(hi >> vsh) * (unsigned)p.x_sH +
                                  (unsigned)(wi >> vsh) * (unsigned)p.x_sW + (unsigned)(a_chan[q] - vc0)) * (unsigned)ES;
            lds_dma16(ok ? off : DMA_OOB, rsrc_x, 0u, __builtin_amdgcn_readfirstlane(st + (wave_u * QA + q) * 1024u));
            a_px[q].advance_digits(adv_b, adv_h, adv_w, p.Ho, p.Wo);
        }
#pragma unroll
        for (int q = 0; q < QB; ++q)
            lds_dma16(b_voff[q], rsrc_dy, b_soff, __builtin_amdgcn_readfirstlane(st + A_BYTES + (wave_u * QB + q) * 1024u));
        b_soff += b_step;
        if constexpr (LIN) {
            if (halo) {      // the NEXT k-step's halo predicate, computed behind this one's queue (it runs under the MFMA phase)
#pragma unroll
                for (int q = 0; q < QA; ++q) {
                    a_px[q].advance_digits(0, adv_h, adv_w, p.Ho, p.Wo);
                    const bool ok = (unsigned)(a_px[q].ho + dr) < (unsigned)p.H && (unsigned)(a_px[q].wo + dsx) < (unsigned)p.W;
                    a_sel[q] = ok ? (unsigned)a_chan[q] : DMA_OOB;
                }
            }
        }
    };

    // fragment gather: lane l -> rows k = 8*(l>>4) + 4*h + ((l&15)>>2), 8-byte piece (l&3) of the 16-channel block
    const int g8 = (lane >> 4) * 8, li = lane & 15;
    const int piece_chunk = (li & 3) >> 1, piece_half = (li & 1) * 8;
    int a_row[2], a_swz[2], b_row[2], b_swz[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int k = g8 + 4 * h + (li >> 2);
        a_row[h] = k * ROWA; a_swz[h] = dma_swz<ROWA>(k);
        b_row[h] = k * ROWB; b_swz[h] = dma_swz<ROWB>(k);
    }
    auto frag = [&](const unsigned char* tile, const int (&row)[2], const int (&swz)[2], int c0) -> u32x4_t {
        const int lc = (c0 >> 3) + piece_chunk;          // c0: first channel of the 16-wide block (multiple of 16)
        s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4_t*)(tile + row[0] + ((lc ^ swz[0]) * 16) + piece_half));
        s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4_t*)(tile + row[1] + ((lc ^ swz[1]) * 16) + piece_half));
        struct { s16x4_t a, b; } pr = {lo, hi};
        return __builtin_bit_cast(u32x4_t, pr);
    };
    // bias gradient = column sums of dY: the workgroups of the first cin tile / first tap multiply the dY fragments
    // they hold anyway by a vector of ones (one extra MFMA per fragment in two of the four waves)
    const bool do_bias = p.db != nullptr && tm == 0 && tap == 0 && wm == 0;
    f32x4_t accb[MN];
#pragma unroll
    for (int j = 0; j < MN; ++j) accb[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    auto compute = [&](unsigned stage) {
        const unsigned char* la = lds + stage * STAGE_BYTES;
        const unsigned char* lb = la + A_BYTES;
        if constexpr (ES == 4) {
            // exact-fp32 MFMA (16x16x4, one k per lane): lane -> channel (lane & 15) of a 16-wide block, row kq * 4 + (lane >> 4)
#pragma unroll
            for (int kq = 0; kq < 4; ++kq) {
                const int krow = kq * 4 + (lane >> 4);
                const int key = (lane >> 4) << 2;                            // == dma_swz_t<float>(krow): krow & 3 == lane >> 4
                float fa[MM], fb[MN];
#pragma unroll
                for (int i = 0; i < MM; ++i) {
                    const int c = wm * (TM / 2) + i * 16 + (lane & 15);
                    fa[i] = *reinterpret_cast<const float*>(la + krow * ROWA + (((c >> 2) ^ key) * 16) + (c & 3) * 4);
                }
#pragma unroll
                for (int j = 0; j < MN; ++j) {
                    const int c = wn * (TN / 2) + j * 16 + (lane & 15);
                    fb[j] = *reinterpret_cast<const float*>(lb + krow * ROWB + (((c >> 2) ^ key) * 16) + (c & 3) * 4);
                }
#pragma unroll
                for (int i = 0; i < MM; ++i)
#pragma unroll
                    for (int j = 0; j < MN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
                if (do_bias) {
#pragma unroll
                    for (int j = 0; j < MN; ++j) accb[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, fb[j], accb[j], 0, 0, 0);
                }
            }
            return;
        }
        const u32x4_t ones = {Ones16<T>::kPair, Ones16<T>::kPair, Ones16<T>::kPair, Ones16<T>::kPair};
        u32x4_t fa[MM], fb[MN];
#pragma unroll
        for (int i = 0; i < MM; ++i) fa[i] = frag(la, a_row, a_swz, wm * (TM / 2) + i * 16);
#pragma unroll
        for (int j = 0; j < MN; ++j) fb[j] = frag(lb, b_row, b_swz, wn * (TN / 2) + j * 16);
#pragma unroll
        for (int i = 0; i < MM; ++i)
#pragma unroll
            for (int j = 0; j < MN; ++j)
                acc[i][j] = mma16<T>(fa[i], fb[j], acc[i][j]);
        if (do_bias) {
#pragma unroll
            for (int j = 0; j < MN; ++j) accb[j] = mma16<T>(ones, fb[j], accb[j]);
        }
    };

    const long span = k_end - k_begin;
    const int nsteps = span > 0 ? (int)((span + KP - 1) / KP) : 0;
    // steps past the slice end are still queued (their dY rows are out of range -> zeros, never consumed) so the
    // outstanding-load count is the same in every iteration
    unsigned long long pt[4] = {0ull, 0ull, 0ull, 0ull};
    const unsigned long long p_start = PROF ? __builtin_readcyclecounter() : 0ull;
    issue(0u);
    issue(1u);
    unsigned cur = 0u, nxt = 2u;
    for (int it = 0; it < nsteps; ++it) {
        unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0;
        if (PROF) t0 = __builtin_readcyclecounter();
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(QA + QB) : "memory");      // k-step `it` has landed (this wave's part)
        if (PROF) t1 = __builtin_readcyclecounter();
        __builtin_amdgcn_s_barrier();                          // ... everyone's part; and slot `nxt` is no longer being read
        if (PROF) t2 = __builtin_readcyclecounter();
        issue(nxt);
        if (PROF) t3 = __builtin_readcyclecounter();
        compute(cur);
        if (PROF) {
            const unsigned long long t4 = __builtin_readcyclecounter();
            pt[0] += t1 - t0; pt[1] += t2 - t1; pt[2] += t3 - t2; pt[3] += t4 - t3;
        }
        cur = (cur == NST - 1) ? 0u : cur + 1u;
        nxt = (nxt == NST - 1) ? 0u : nxt + 1u;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long p_loop_end = PROF ? __builtin_readcyclecounter() : 0ull;

    const long NW = (long)p.Cout * taps * p.Cin;
    float* __restrict__ dst = (p.chunks > 1) ? (p.ws + (long)chunk * NW) : p.dw;
    const bool add = (p.chunks == 1);
    if (do_bias && lane < 16) {                 // every row of the ones-product holds the column sum: row 0 = lanes 0..15, reg 0
        float* __restrict__ bdst = (p.chunks > 1) ? (p.db_ws + (long)chunk * p.Cout) : p.db;
#pragma unroll
        for (int j = 0; j < MN; ++j) {
            const int cout = n0 + wn * (TN / 2) + j * 16 + lane;
            if (cout < p.Cout) bdst[cout] = (add ? bdst[cout] : 0.f) + accb[j][0];
        }
    }
#pragma unroll
    for (int i = 0; i < MM; ++i) {
        const int cin = m0 + wm * (TM / 2) + i * 16 + (lane >> 4) * 4;
        if (cin >= p.Cin) continue;
#pragma unroll
        for (int j = 0; j < MN; ++j) {
            const int cout = n0 + wn * (TN / 2) + j * 16 + (lane & 15);
            if (cout >= p.Cout) continue;
            float* q = dst + ((long)cout * taps + tap) * p.Cin + cin;
            float4 v = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            if (add) {
                const float4 o = *reinterpret_cast<const float4*>(q);
                v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
            }
            if (!ablate_stores || v.x == 123.456f) *reinterpret_cast<float4*>(q) = v;
        }
    }
    if (PROF && prof && lane == 0) {
        const unsigned long long p_end = __builtin_readcyclecounter();
        unsigned long long* d = prof + ((long)blockIdx.x * 4 + wave) * 8;
        d[0] = pt[0]; d[1] = pt[1]; d[2] = pt[2]; d[3] = pt[3];
        d[4] = p_loop_end - p_start; d[5] = p_end - p_loop_end; d[6] = (unsigned long long)nsteps; d[7] = p_start;
    }
}

template <int TM, int TN>
__global__ void __launch_bounds__(256, TM > 128 ? 2 : 3) conv_wgrad_dma_kernel(const MpnWgradParams p, long chunk_pixels) {
    conv_wgrad_dma_body<bf16_t, TM, TN, false>(p, chunk_pixels);
}
template <int TM, int TN>
__global__ void __launch_bounds__(256, TM > 128 ? 2 : 3) conv_wgrad_dma_prof_kernel(const MpnWgradParams p, long chunk_pixels, unsigned long long* prof) {
    conv_wgrad_dma_body<bf16_t, TM, TN, false, true>(p, chunk_pixels, prof);
}
template <int TM, int TN>
__global__ void __launch_bounds__(256, TM > 128 ? 2 : 3) conv_wgrad_dma_lin_kernel(const MpnWgradParams p, long chunk_pixels) {
    conv_wgrad_dma_body<bf16_t, TM, TN, false, false, true>(p, chunk_pixels);
}
template <int TM, int TN>
__global__ void __launch_bounds__(256, TM > 128 ? 2 : 3) conv_wgrad_dma_lin_f16_kernel(const MpnWgradParams p, long chunk_pixels) {
    conv_wgrad_dma_body<f16_t, TM, TN, false, false, true>(p, chunk_pixels);
}
template <int TM, int TN>
__global__ void __launch_bounds__(256, TM > 128 ? 2 : 3) conv_wgrad_dma_lin_prof_kernel(const MpnWgradParams p, long chunk_pixels, unsigned long long* prof) {
    conv_wgrad_dma_body<bf16_t, TM, TN, false, true, true>(p, chunk_pixels, prof);
}
template <int TM, int TN>
__global__ void __launch_bounds__(256, TM > 128 ? 2 : 3) conv_wgrad_dma_f16_kernel(const MpnWgradParams p, long chunk_pixels) {
    conv_wgrad_dma_body<f16_t, TM, TN, false>(p, chunk_pixels);
}
template <int TM, int TN>
__global__ void __launch_bounds__(256, 3) conv_wgrad_dma_f32_kernel(const MpnWgradParams p, long chunk_pixels) {
    conv_wgrad_dma_body<float, TM, TN, false>(p, chunk_pixels);
}
template <int TM, int TN>
__global__ void __launch_bounds__(256, 3) conv_wgrad_dma_lin_f32_kernel(const MpnWgradParams p, long chunk_pixels) {
    conv_wgrad_dma_body<float, TM, TN, false, false, true>(p, chunk_pixels);
}
template <int TM, int TN>
__global__ void __launch_bounds__(256, TM > 128 ? 2 : 3) conv_wgrad_dma_seg_kernel(const MpnWgradParams p, long chunk_pixels) {
    conv_wgrad_dma_body<bf16_t, TM, TN, true, false, true>(p, chunk_pixels);     // pyramid levels are dense stride-1 "same" convolutions (mpn_conv_wgrad checks)
}
template <int TM, int TN>
__global__ void __launch_bounds__(256, TM > 128 ? 2 : 3) conv_wgrad_dma_seg_f16_kernel(const MpnWgradParams p, long chunk_pixels) {
    conv_wgrad_dma_body<f16_t, TM, TN, true, false, true>(p, chunk_pixels);
}

// dst[i] (+)= sum_c ws[c][i], chunks added in index order (deterministic).  (A variant that split the chunks over four
// slice-groups per column with an LDS combine measured slower: 23 vs 13 us on the layer3 shapes.)
__global__ void reduce_partials_kernel(const float* __restrict__ ws, int chunks, long n, float* __restrict__ dst, int accumulate) {
    const long i4 = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i4 >= n) return;
    if (i4 + 3 < n) {
        float4 a = accumulate ? *reinterpret_cast<const float4*>(dst + i4) : make_float4(0.f, 0.f, 0.f, 0.f);
        for (int c = 0; c < chunks; ++c) {
            const float4 v = *reinterpret_cast<const float4*>(ws + (long)c * n + i4);
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        *reinterpret_cast<float4*>(dst + i4) = a;
    } else {
        for (long i = i4; i < n; ++i) {
            float a = accumulate ? dst[i] : 0.f;
            for (int c = 0; c < chunks; ++c) a += ws[(long)c * n + i];
            dst[i] = a;
        }
    }
}

inline int launch_reduce_partials(const float* ws, int chunks, long n, float* dst, int accumulate, hipStream_t st) {
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)(((n + 3) / 4 + 255) / 256)), dim3(256), 0, st, ws, chunks, n, dst, accumulate);
    return mpn_launch_status();
}

// small n (bias gradients: n = channels, many chunks): 16 outputs x 16 chunk-slices per block
__global__ void reduce_partials_small_kernel(const float* __restrict__ ws, int chunks, long n, float* __restrict__ dst, int accumulate) {
    __shared__ float sh[16][17];
    const int ol = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const long i = (long)blockIdx.x * 16 + ol;
    float a = 0.f;
    if (i < n)
        for (int c = sl; c < chunks; c += 16) a += ws[(long)c * n + i];
    sh[sl][ol] = a;
    __syncthreads();
    if (sl == 0 && i < n) {
        a = accumulate ? dst[i] : 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) a += sh[k][ol];
        dst[i] = a;
    }
}

#if MPN_EXP
unsigned long long* g_wgrad_prof = nullptr;        // tools/kloop_profile.py: [workgroups][4 waves][8] cycle sums of the next 128x128 launches
#endif

inline int pick_tile(int n) { return n > 64 ? 128 : (n > 32 ? 64 : 32); }

template <typename T, int TM>
int launch_wgrad_n(const MpnWgradParams& p, int tn, long grid, long chunk_pixels, hipStream_t st) {
    if (tn == 128) hipLaunchKernelGGL((conv_wgrad_kernel<T, TM, 128>), dim3((unsigned)grid), dim3(256), 0, st, p, chunk_pixels);
    else if (tn == 64) hipLaunchKernelGGL((conv_wgrad_kernel<T, TM, 64>), dim3((unsigned)grid), dim3(256), 0, st, p, chunk_pixels);
    else hipLaunchKernelGGL((conv_wgrad_kernel<T, TM, 32>), dim3((unsigned)grid), dim3(256), 0, st, p, chunk_pixels);
    return mpn_launch_status();
}

// bf16 launches whose tensors fit 32-bit buffer offsets take the LDS-DMA kernel (tiles of 64 or 128 channels; narrower
// operands ride in a zero-filled 64-wide tile); everything else (f32 parity path, > 2 GB tensors) the generic kernel.
inline long wgrad_total_pixels(const MpnWgradParams& p) {
    if (p.nseg <= 0) return (long)p.B * p.Ho * p.Wo;
    long t = 0;
    for (int l = 0; l < p.nseg; ++l) t += (long)p.B * p.seg_H[l] * p.seg_W[l];
    return t;
}

inline bool wgrad_uses_dma(const MpnWgradParams& p) {
    static const bool use_dma = mpn_tune("MPN_WGRAD_NO_DMA", 0) == 0;
    long P = (long)p.B * p.Ho * p.Wo, xb = (long)p.B * p.x_sB;
    if (p.kseg_n > 0) xb = (long)p.B * p.H * p.W * p.kseg_c;
    if (p.nseg > 0) {
        P = 0; xb = 0;
        for (int l = 0; l < p.nseg; ++l) {
            const long pl = (long)p.B * p.seg_H[l] * p.seg_W[l];
            if (pl > P) P = pl;
            if (pl * p.x_sW > xb) xb = pl * p.x_sW;
        }
    }
    const long es = p.dtype == MPN_F32 ? 4 : 2;
    const bool small = xb * es < 0x7fffffffL && P * p.dy_sP * es < 0x7fffffffL;
    // f32 (round 5): the same LDS-DMA ring feeding the exact-fp32 MFMA; single dense tensors only (no pyramid / virtual concatenation)
    if (p.dtype == MPN_F32) return use_dma && mpn_tune("MPN_WGRAD_F32_DMA", 1) != 0 && small && p.nseg == 0 && p.kseg_n == 0 && p.Cin % 4 == 0;
    return (p.dtype == MPN_BF16 || p.dtype == MPN_F16) && use_dma && small && p.Cin % 8 == 0;
}

// LIN instantiations (linear x addressing): stride-1 convolutions whose output has the input's extent, over a dense x, no virtual concatenation
inline bool wgrad_lin_ok(const MpnWgradParams& p) {
    static const bool on = mpn_tune("MPN_WGRAD_LIN", 1) != 0;
    if (p.nseg > 0) return true;
    // the tap offset moves into the x descriptor (base + tap, num_records = bytes - tap): a NEGATIVE tap lengthens the range by up to
    // (pad W + pad) pixels, and the descriptor holds 2^31 - 1 bytes at most — beyond that the launch takes the gather kernel instead of
    // a clamped descriptor that would zero-fill valid rows at the tail
    const int64_t es = p.dtype == MPN_F32 ? 4 : 2;
    const int64_t reach = (int64_t)p.B * p.x_sB * es + ((int64_t)p.pad * p.W + p.pad) * p.x_sW * es;
    if (reach >= 0x7fffffffLL) return false;
    return on && p.kseg_n == 0 && p.stride == 1 && p.Ho == p.H && p.Wo == p.W && p.x_sH == (int64_t)p.W * p.x_sW && p.x_sB == (int64_t)p.H * p.x_sH;
}

constexpr long kWgradTarget = 512;       // workgroups per launch (~2 per CU): long slices, little partial-sum traffic

inline void wgrad_tiles(const MpnWgradParams& p, int& tm, int& tn) {
    tm = pick_tile(p.Cin); tn = pick_tile(p.Cout);
    if (!wgrad_uses_dma(p)) return;
    if (tm < 64) tm = 64;
    if (tn < 64) tn = 64;
    // A 256 x 128 tile moves a third less data per FLOP through the DMA/LDS path (it pays in conv_igemm), but here it
    // measured SLOWER (3x3 256->256 @60x60: 243 vs 208 us, 512->256 @120x120: 1595 vs 1431 us): with two workgroups
    // per CU the transpose reads are no longer hidden and the slice count (partial-sum traffic) doubles.  Kept behind
    // MPN_WGRAD_TM256_MIN_STEPS (minimum k-steps per workgroup) for experiments, off by default.
    static const long min_steps = mpn_tune("MPN_WGRAD_TM256_MIN_STEPS", 1L << 40);
    if (p.dtype != MPN_F32 && p.Cin >= 256 && tn == 128) {        // (the f32 ring kernel has no 256-row instantiation)
        const long tiles = (long)((p.Cin + 255) / 256) * ((p.Cout + 127) / 128) * p.R * p.S;
        const long chunks = (kWgradTarget + tiles - 1) / tiles;
        const long P = (long)p.B * p.Ho * p.Wo;
        if (P / 32 / chunks >= min_steps) tm = 256;
    }
}

template <typename T>
int launch_wgrad(const MpnWgradParams& p, hipStream_t st, bool reduce = true) {
    int tm, tn;
    wgrad_tiles(p, tm, tn);
    const long tilesM = (p.Cin + tm - 1) / tm, tilesN = (p.Cout + tn - 1) / tn;
    const long P = wgrad_total_pixels(p);
    const int kp = sizeof(T) == 2 ? 32 : 16;
    long chunk_pixels = (P + p.chunks - 1) / p.chunks;
    chunk_pixels = ((chunk_pixels + kp - 1) / kp) * kp;
    if (p.nseg > 0) chunk_pixels = p.seg_chunk_pixels;
    const long grid = tilesM * tilesN * p.R * p.S * p.chunks;
    if (grid <= 0 || grid > 0x7fffffffL || P >= 0x7fffffffL) return MPN_E_BADARG;
    // ablations for tools/ (results are WRONG with either bit): 1 = no reduction launch, 2 = the slices do not store their partials
    static const int ablate = (int)mpn_tune("MPN_WGRAD_ABLATE", 0);      // experiments build only
    if ((ablate & 2) && p.chunks > 1) chunk_pixels |= (1L << 62);
    if ((ablate & 1) && p.chunks > 1) reduce = false;
    int rc;
    const dim3 g((unsigned)grid), blk(256);
    if (sizeof(T) == 4 && wgrad_uses_dma(p)) {
#define MPN_WGRAD_DMA_LAUNCH_F32(KERNEL)                                                                                 \
        if (tm == 128 && tn == 128) hipLaunchKernelGGL((KERNEL<128, 128>), g, blk, 0, st, p, chunk_pixels);              \
        else if (tm == 128) hipLaunchKernelGGL((KERNEL<128, 64>), g, blk, 0, st, p, chunk_pixels);                       \
        else if (tn == 128) hipLaunchKernelGGL((KERNEL<64, 128>), g, blk, 0, st, p, chunk_pixels);                       \
        else hipLaunchKernelGGL((KERNEL<64, 64>), g, blk, 0, st, p, chunk_pixels)
        if (wgrad_lin_ok(p)) { MPN_WGRAD_DMA_LAUNCH_F32(conv_wgrad_dma_lin_f32_kernel); }
        else { MPN_WGRAD_DMA_LAUNCH_F32(conv_wgrad_dma_f32_kernel); }
#undef MPN_WGRAD_DMA_LAUNCH_F32
        rc = mpn_launch_status();
    } else if (sizeof(T) == 2 && wgrad_uses_dma(p)) {
#define MPN_WGRAD_DMA_LAUNCH(KERNEL)                                                                                     \
        if (tm == 256) hipLaunchKernelGGL((KERNEL<256, 128>), g, blk, 0, st, p, chunk_pixels);                           \
        else if (tm == 128 && tn == 128) hipLaunchKernelGGL((KERNEL<128, 128>), g, blk, 0, st, p, chunk_pixels);         \
        else if (tm == 128) hipLaunchKernelGGL((KERNEL<128, 64>), g, blk, 0, st, p, chunk_pixels);                       \
        else if (tn == 128) hipLaunchKernelGGL((KERNEL<64, 128>), g, blk, 0, st, p, chunk_pixels);                       \
        else hipLaunchKernelGGL((KERNEL<64, 64>), g, blk, 0, st, p, chunk_pixels)
        const bool lin = wgrad_lin_ok(p);
#if MPN_EXP
        if (g_wgrad_prof && p.nseg == 0 && p.dtype == MPN_BF16 && tm == 128 && tn == 128) {
            if (lin) hipLaunchKernelGGL((conv_wgrad_dma_lin_prof_kernel<128, 128>), g, blk, 0, st, p, chunk_pixels, g_wgrad_prof);
            else hipLaunchKernelGGL((conv_wgrad_dma_prof_kernel<128, 128>), g, blk, 0, st, p, chunk_pixels, g_wgrad_prof);
        } else
#endif
        if (p.nseg == 0 && lin) {
            if (p.dtype == MPN_F16) { MPN_WGRAD_DMA_LAUNCH(conv_wgrad_dma_lin_f16_kernel); }
            else { MPN_WGRAD_DMA_LAUNCH(conv_wgrad_dma_lin_kernel); }
        } else if (p.nseg > 0) {
            if (p.dtype == MPN_F16) { MPN_WGRAD_DMA_LAUNCH(conv_wgrad_dma_seg_f16_kernel); }
            else { MPN_WGRAD_DMA_LAUNCH(conv_wgrad_dma_seg_kernel); }
        } else if (p.dtype == MPN_F16) { MPN_WGRAD_DMA_LAUNCH(conv_wgrad_dma_f16_kernel); }
        else { MPN_WGRAD_DMA_LAUNCH(conv_wgrad_dma_kernel); }
#undef MPN_WGRAD_DMA_LAUNCH
        rc = mpn_launch_status();
    } else if (tm == 128) rc = launch_wgrad_n<T, 128>(p, tn, grid, chunk_pixels, st);
    else if (tm == 64) rc = launch_wgrad_n<T, 64>(p, tn, grid, chunk_pixels, st);
    else rc = launch_wgrad_n<T, 32>(p, tn, grid, chunk_pixels, st);
    if (rc != 0) return rc;
    if (p.chunks > 1 && reduce) {
        if (p.db) {
            hipLaunchKernelGGL(reduce_partials_small_kernel, dim3((unsigned)((p.Cout + 15) / 16)), dim3(256), 0, st,
                               (const float*)p.db_ws, p.chunks, (long)p.Cout, p.db, 1);
            rc = mpn_launch_status();
            if (rc != 0) return rc;
        }
        rc = launch_reduce_partials((const float*)p.ws, p.chunks, (long)p.Cout * p.R * p.S * p.Cin, p.dw, 1, st);
    }
    return rc;
}

}  // namespace

extern "C" int mpn_debug_wgrad_prof(void* buf) {
#if MPN_EXP
    g_wgrad_prof = (unsigned long long*)buf;
    return 0;
#else
    (void)buf;
    return MPN_E_UNSUPPORTED;          // the PROF instantiations live in the experiments build only (common.h)
#endif
}

extern "C" int mpn_conv_wgrad_seg_plan(MpnWgradParams* p) {
    if (!p || p->nseg <= 0 || p->nseg > 5) return MPN_E_BADARG;
    const int want = mpn_conv_wgrad_chunks(p);
    if (want < 1) return want;
    const long P = wgrad_total_pixels(*p);
    // Every level rounds its slice count up, so `want` slices of the whole pyramid become more (14 -> 17 on p3..p7 at 480x480: 612
    // workgroups on the 512 slots mpn_conv_wgrad_chunks aimed at).  Round 6 measured the obvious remedy — lengthen the slices until the
    // total fits `want` again, the never-one-more rule of the single-tensor launches — in the step: 36.09 / 36.10 -> 36.24 / 36.29 ms
    // (profiles/r06_small_items_ab.txt): under the two-stream schedule the tail round is filled by the other stream and the longer
    // slices only lengthen the launch.  Off; MPN_WGRAD_SEG_FIT=1 in the experiments build.
    static const bool fit = mpn_tune("MPN_WGRAD_SEG_FIT", 0) != 0;
    long cp = 0;
    int c = 0;
    for (long w = want; w >= 1; --w) {
        cp = (P + w - 1) / w;
        cp = ((cp + 31) / 32) * 32;
        c = 0;
        for (int l = 0; l < p->nseg; ++l) {
            p->seg_chunk0[l] = c;
            const long pl = (long)p->B * p->seg_H[l] * p->seg_W[l];
            c += (int)((pl + cp - 1) / cp);
        }
        if (!fit || c <= want || c <= p->nseg) break;
    }
    p->seg_chunk0[p->nseg] = c;
    p->seg_chunk_pixels = (int)cp;
    p->chunks = c;
    return c;
}

extern "C" int mpn_conv_wgrad_chunks(const MpnWgradParams* p) {
    if (!p) return MPN_E_BADARG;
    int tm, tn;
    wgrad_tiles(*p, tm, tn);
    const long tiles = (long)((p->Cin + tm - 1) / tm) * ((p->Cout + tn - 1) / tn) * p->R * p->S;
    const long P = wgrad_total_pixels(*p);
    static const long target16 = mpn_tune("MPN_WGRAD_TARGET", kWgradTarget);
    // f32 launches that fall back to the register-staged generic kernel (matrix-pipe bound, no DMA ring): three workgroups per CU's worth
    // of slices (cfg2 55.7 -> 55.1 ms; 256: 62.3); on the ring 512 and 768 measure the same (52.06 / 52.08 ms), 1024 worse
    const long target = (p->dtype == MPN_F32 && !wgrad_uses_dma(*p) && target16 == kWgradTarget) ? 768 : target16;
    static const long minpix = mpn_tune("MPN_WGRAD_MINPIX", 512);
    // ~2 workgroups per CU (long slices run near peak, partial-sum traffic dominates beyond) and NEVER one more than that: rounding the
    // slice count up put 540 workgroups on the 512 slots of the 3x3 256-channel layers — the 28 that share a CU three ways finish last,
    // and in isolation the launch takes 15 - 20 % longer (3x3 256->256 @30x30 64 -> 53 us, @60x60 195 -> 161, 512->256 @120x120
    // 1 374 -> 1 216; in the step, where the other stream fills the tail, neutral: profiles/r04_kloop_phase_profile.txt).  MPN_WGRAD_CEIL=1: the old rounding
    static const bool round_up = mpn_tune("MPN_WGRAD_CEIL", 0) != 0;
    long want = round_up ? (target + tiles - 1) / tiles : target / tiles;
    const long maxc = (P + minpix - 1) / minpix;       // keep >= 512 pixels per slice
    if (want > maxc) want = maxc;
    if (want < 1) want = 1;
    if (want > 256) want = 256;
    return (int)want;
}

extern "C" int mpn_conv_wgrad(const MpnWgradParams* pp, void* stream) {
    if (!pp) return MPN_E_BADARG;
    const MpnWgradParams& p = *pp;
    MPN_CHECK_ARG(p.dw && mpn_dtype_ok(p.dtype) && p.B > 0 && p.Cin > 0 && p.Cout > 0);
    if (p.nseg > 0) {
        MPN_CHECK_ARG(p.nseg <= 5 && wgrad_uses_dma(p) && p.stride == 1 && p.R == p.S && 2 * p.pad == p.R - 1);
        for (int l = 0; l < p.nseg; ++l)         // pyramid levels always take the linear-addressing kernel: same descriptor reach rule
            if (((int64_t)p.B * p.seg_H[l] * p.seg_W[l] + (int64_t)p.pad * p.seg_W[l] + p.pad) * p.x_sW * 2 >= 0x7fffffffLL) return MPN_E_UNSUPPORTED;
        MPN_CHECK_ARG(p.seg_chunk_pixels > 0 && p.seg_chunk_pixels % 32 == 0 && p.seg_chunk0[0] == 0 && p.seg_chunk0[p.nseg] == p.chunks);
        for (int l = 0; l < p.nseg; ++l) MPN_CHECK_ARG(p.seg_x[l] && p.seg_dy[l] && p.seg_H[l] > 0 && p.seg_W[l] > 0);
    } else {
        MPN_CHECK_ARG((p.x || p.kseg_n > 0) && p.dy && p.Ho > 0 && p.Wo > 0);
    }
    if (p.kseg_n > 0) {
        int tm, tn;
        wgrad_tiles(p, tm, tn);
        MPN_CHECK_ARG(p.kseg_n <= 4 && p.kseg_c == 128 && tm == 128 && p.Cin == p.kseg_n * p.kseg_c && wgrad_uses_dma(p) && p.stride == 1);
        for (int k = 0; k < p.kseg_n; ++k)
            MPN_CHECK_ARG(p.kseg_x[k] && p.kseg_shift[k] >= 0 && p.kseg_shift[k] < 8 && ((p.H >> p.kseg_shift[k]) << p.kseg_shift[k]) == p.H &&
                          ((p.W >> p.kseg_shift[k]) << p.kseg_shift[k]) == p.W);
    }
    MPN_CHECK_ARG(p.Cin % 8 == 0);
    MPN_CHECK_ARG(p.chunks >= 1 && (p.chunks == 1 || p.ws));
    MPN_CHECK_ARG(!p.db || (wgrad_uses_dma(p) && (p.chunks == 1 || p.db_ws)));
    hipStream_t st = (hipStream_t)stream;
    if (p.dtype == MPN_F32) return launch_wgrad<float>(p, st);
    if (p.dtype == MPN_F16) return launch_wgrad<f16_t>(p, st);
    return launch_wgrad<bf16_t>(p, st);
}

extern "C" int mpn_conv_wgrad_partials(const MpnWgradParams* pp, void* stream) {
    if (!pp) return MPN_E_BADARG;
    const MpnWgradParams& p = *pp;
    MPN_CHECK_ARG(p.dw && p.ws && p.chunks > 1 && (p.nseg > 0 || ((p.x || p.kseg_n > 0) && p.dy && p.Ho > 0 && p.Wo > 0)));
    MPN_CHECK_ARG(mpn_dtype_ok(p.dtype));
    MPN_CHECK_ARG(p.B > 0 && p.Cin > 0 && p.Cout > 0 && p.Cin % 8 == 0);
    hipStream_t st = (hipStream_t)stream;
    if (p.dtype == MPN_F32) return launch_wgrad<float>(p, st, false);
    if (p.dtype == MPN_F16) return launch_wgrad<f16_t>(p, st, false);
    return launch_wgrad<bf16_t>(p, st, false);
}

extern "C" int mpn_conv_wgrad_kernel_id(const MpnWgradParams* p) {
    if (!p) return MPN_E_BADARG;
    int tm, tn;
    wgrad_tiles(*p, tm, tn);
    const bool dma = wgrad_uses_dma(*p);
    return (tm << 16) | (tn << 4) | (dma && p->nseg == 0 && wgrad_lin_ok(*p) ? 2 : 0) | (dma ? 1 : 0);      // bit 1: the linear-addressing instantiation
}

extern "C" int mpn_reduce_partials(const float* ws, int chunks, int64_t n, float* dst, int accumulate, void* stream) {
    MPN_CHECK_ARG(ws && dst && chunks >= 1 && n > 0);
    if (n <= 8192 && chunks >= 32) {
        hipLaunchKernelGGL(reduce_partials_small_kernel, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, (hipStream_t)stream,
                           ws, chunks, (long)n, dst, accumulate);
        return mpn_launch_status();
    }
    return launch_reduce_partials(ws, chunks, (long)n, dst, accumulate, (hipStream_t)stream);
}
