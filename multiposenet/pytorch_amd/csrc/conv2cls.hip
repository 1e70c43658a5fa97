// conv2cls.hip — the keypoint head's conv2 (posenet.py:311-315) by POSITION CLASSES for its x8 / x4 up-sampled members.
//
// conv2 convolves cat(up8(q5), up4(q4), up2(q3), q2) (4 x 128 channels at H x W) with a 3x3 filter.  For a member up-sampled by
// s the output pixel (s i + a, s j + c) reads, per axis, the low-resolution sources
//      a = 0         ("first"):  taps r = 0 -> i - 1,  r = 1, 2 -> i
//      0 < a < s - 1 ("mid")  :  taps r = 0, 1, 2      -> i
//      a = s - 1     ("last") :  taps r = 0, 1 -> i,   r = 2 -> i + 1
// so its contribution is one of 3 x 3 = 9 CLASS MAPS at the member's own resolution, each a 3x3 / pad-1 convolution of the
// low-resolution tensor with a filter whose frame tap (u, v) is the SUM of the original taps that land on source (i + u - 1,
// j + v - 1) (zero where none does): the zero border of the up-sampled image is the zero border of the low-resolution one.
// Work per output pixel of such a member: 81 frame taps per s^2 pixels instead of 9 per pixel (x8: 14 %, x4: 56 %).
//
//   forward :  M_s   = conv3x3(q_s, Wc_s)            [B, H/s, W/s, 9 * Cout]   (plain launches of conv_igemm.hip, f32 out)
//              E     = expand(M_8, M_4)              [B, H, W, Cout]           (this file)
//              y     = relu(conv3x3(cat(up2(q3), q2), W[:, 256:]) + bias + E)   (virtual concatenation of TWO members, E = residual)
//   backward:  P_s   = class-pool(dy)                [B, H/s, W/s, 9 * Cout]   (this file: one pass over dy for both s)
//              G_s   = tap-sums(P_s)                 [B, H/s, W/s, 9 * Cout]   (this file; G_s[t] = sum of dy over the output pixels
//                                                                               whose filter tap t reads the low-resolution pixel)
//              dq_s  = dgrad1x1(G_s, Wtap_s^T),  dWtap_s = wgrad1x1(q_s, G_s)   (plain launches: per TAP the member is a 1x1 convolution
//                                                                               of the low-resolution tensor — 9x fewer FLOPs than the frames)
//              dW   += fold(dWtap_8, dWtap_4, dW_main)                          (this file)
// (The forward keeps the frame convolutions: the per-tap form would hand the 9-term sums to the full-resolution expansion.)
// Everything here is streaming / tiny; the arithmetic that matters stays in the convolution kernels.
#include "common.h"

namespace {

// frame tap u of row class ca collects the original taps in this bit set (bit r)
__device__ __forceinline__ unsigned cls_taps(int ca, int u) {
    //            u = 0   u = 1   u = 2
    // first      {0}     {1,2}   {}
    // mid        {}      {0,1,2} {}
    // last       {}      {0,1}   {2}
    const unsigned tbl = (0x1u) | (0x6u << 3) | (0x0u << 6) | (0x0u << 9) | (0x7u << 12) | (0x0u << 15) | (0x0u << 18) | (0x3u << 21) | (0x4u << 24);
    return (tbl >> ((ca * 3 + u) * 3)) & 7u;
}
// the frame tap of class ca that original tap r lands on
__device__ __forceinline__ int cls_frame(int ca, int r) { return ca == 0 ? (r == 0 ? 0 : 1) : (ca == 1 ? 1 : (r == 2 ? 2 : 1)); }
// class of offset a inside an s-pixel block
__device__ __forceinline__ int cls_of(int a, int s) { return a == 0 ? 0 : (a == s - 1 ? 2 : 1); }

// W f32 [O][3][3][4 * C] (members q5, q4, q3, q2 along the input channels) ->
//   comb + 0            : Wm  [O][3][3][2 C]      (members q3, q2)
//   comb + O*9*2C       : Wc8 [9 O][3][3][C]      (class k = ca * 3 + cc of member 0, frame filters)
//   comb + O*9*2C + 81OC: Wc4 [9 O][3][3][C]      (member 1)
//   then                : Wtap8 [9 O][C], Wtap4 [9 O][C]   (row t * O + o = tap t of output channel o: the member as nine 1x1 convolutions)
__global__ void conv2cls_combine_kernel(const float* __restrict__ w, float* __restrict__ comb, int O, int C) {
    const long nm = (long)O * 9 * 2 * C, nc = (long)9 * O * 9 * C, nt = (long)9 * O * C;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nm + 2 * nc + 2 * nt) return;
    const int K = 4 * C;
    if (i >= nm + 2 * nc) {
        long j = i - nm - 2 * nc;
        const int member = j >= nt ? 1 : 0;
        if (member) j -= nt;
        const int c = (int)(j % C);
        const long to = j / C;
        const int o = (int)(to % O), t = (int)(to / O);
        comb[i] = w[((long)o * 9 + t) * K + member * C + c];
        return;
    }
    if (i < nm) {
        const int c = (int)(i % (2 * C));
        const long ot = i / (2 * C);                              // o * 9 + tap
        comb[i] = w[ot * K + 2 * C + c];
        return;
    }
    long j = i - nm;
    const int member = j >= nc ? 1 : 0;
    if (member) j -= nc;
    const int c = (int)(j % C);
    long q = j / C;
    const int v = (int)(q % 3); q /= 3;
    const int u = (int)(q % 3); q /= 3;
    const int o = (int)(q % O);
    const int k = (int)(q / O);
    const unsigned rs = cls_taps(k / 3, u), ss = cls_taps(k % 3, v);
    float acc = 0.f;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s = 0; s < 3; ++s)
            if (((rs >> r) & 1u) && ((ss >> s) & 1u)) acc += w[((long)o * 9 + r * 3 + s) * K + member * C + c];
    comb[i] = acc;
}

// four elements as floats: 8 bytes of a 16-bit type, 16 bytes of float
template <typename T> struct Vec8 {
    float v[4];
    __device__ __forceinline__ void load(const T* p) {
        if constexpr (sizeof(T) == 4) {
            const float4 t = *reinterpret_cast<const float4*>(p);
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        } else {
            const uint2 t = *reinterpret_cast<const uint2*>(p);
            T e[4];
            __builtin_memcpy(e, &t, 8);
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = Elem<T>::ld(&e[k]);
        }
    }
    __device__ __forceinline__ void store(T* p) const {
        if constexpr (sizeof(T) == 4) {
            *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
            T e[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) Elem<T>::st(&e[k], v[k]);
            uint2 t;
            __builtin_memcpy(&t, e, 8);
            *reinterpret_cast<uint2*>(p) = t;
        }
    }
};

// E[b, y, x, o] = M8[b, y / 8, x / 8, cls * O + o] + M4[b, y / 4, x / 4, cls' * O + o]; 4 channels per thread, f32 sums, one rounding
template <typename T>
__global__ void __launch_bounds__(256) conv2cls_expand_kernel(const float* __restrict__ m8, const float* __restrict__ m4, T* __restrict__ e, int B, int H, int W, int O) {
    const int og = O / 4;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * H * W * og) return;
    const int g = (int)(i % og);
    long p = i / og;
    const int x = (int)(p % W); p /= W;
    const int y = (int)(p % H);
    const int b = (int)(p / H);
    const int k8 = cls_of(y & 7, 8) * 3 + cls_of(x & 7, 8), k4 = cls_of(y & 3, 4) * 3 + cls_of(x & 3, 4);
    const float4 a = *reinterpret_cast<const float4*>(m8 + (((long)b * (H >> 3) + (y >> 3)) * (W >> 3) + (x >> 3)) * 9 * O + (long)k8 * O + g * 4);
    const float4 c = *reinterpret_cast<const float4*>(m4 + (((long)b * (H >> 2) + (y >> 2)) * (W >> 2) + (x >> 2)) * 9 * O + (long)k4 * O + g * 4);
    Vec8<T> v;
    v.v[0] = a.x + c.x; v.v[1] = a.y + c.y; v.v[2] = a.z + c.z; v.v[3] = a.w + c.w;
    v.store(e + i * 4);
}

// The forward per TAP (the f32 path, where the zero taps of the frame filters would cost real matrix time): T[b, i, j, t * O + o] is the
// 1x1 convolution of the low-resolution tensor with filter tap t; class map k = (ca, cc) collects, for every original tap (r, s), the
// product at the source pixel that tap reads — frame tap (u, v) = (cls_frame(ca, r), cls_frame(cc, s)) -> pixel (i + u - 1, j + v - 1),
// zero outside.  Nine terms in (r, s) order; f32 in, f32 out, 4 channels per thread.
__global__ void __launch_bounds__(256) conv2cls_classsum_kernel(const float* __restrict__ t, float* __restrict__ m, int B, int h, int w, int O) {
    const int og = O / 4;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)B * h * w * 9 * og) return;
    const int g = (int)(idx % og);
    long q = idx / og;
    const int k = (int)(q % 9); q /= 9;
    const int j = (int)(q % w); q /= w;
    const int i = (int)(q % h);
    const int b = (int)(q / h);
    const int ca = k / 3, cc = k - ca * 3;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int ii = i + cls_frame(ca, r) - 1;
        if (ii < 0 || ii >= h) continue;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int jj = j + cls_frame(cc, s) - 1;
            if (jj < 0 || jj >= w) continue;
            const float4 v = *reinterpret_cast<const float4*>(t + (((long)b * h + ii) * w + jj) * 9 * O + (long)(r * 3 + s) * O + g * 4);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    *reinterpret_cast<float4*>(m + (((long)b * h + i) * w + j) * 9 * O + (long)k * O + g * 4) = acc;
}

// Class pooling of dy [B, H, W, O]: P4[b, i, j, k * O + o] = sum over the pixels (4 i + a, 4 j + c) of class k of dy; P8 likewise over 8 x 8
// blocks.  One thread per (4 x 4 quadrant, 4 channels): its 16 pixels (all sixteen 8-byte loads in flight) give the nine P4 sums
// outright; the four quadrants of an 8 x 8 block sit in four adjacent lanes and their partial P8 sums meet through two xor-shuffles
// (fixed order: deterministic).  f32 sums, rounded once on store.
template <typename T>
__global__ void __launch_bounds__(256) conv2cls_pool_kernel(const T* __restrict__ dy, T* __restrict__ p8, T* __restrict__ p4, int B, int H, int W, int O) {
    constexpr int V = 4;
    const int og = O / V;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long nthreads = (long)B * (H >> 3) * (W >> 3) * og * 4;
    const bool live = gid < nthreads;                             // (whole 4-lane groups are live or dead together: nthreads % 4 == 0)
    const long gi = live ? gid : 0;
    const int quad = (int)(gi & 3), qa = quad >> 1, qc = quad & 1;
    const int g = (int)((gi >> 2) % og);
    long blk = (gi >> 2) / og;
    const int bj = (int)(blk % (W >> 3)); blk /= (W >> 3);
    const int bi = (int)(blk % (H >> 3));
    const int b = (int)(blk / (H >> 3));
    const int y0 = bi * 8 + qa * 4, x0 = bj * 8 + qc * 4;
    Vec8<T> px[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) px[a][c].load(dy + (((long)b * H + y0 + a) * W + x0 + c) * O + g * V);
    float s4[9][V];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int e = 0; e < V; ++e) s4[k][e] = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int k = (a == 0 ? 0 : (a == 3 ? 2 : 1)) * 3 + (c == 0 ? 0 : (c == 3 ? 2 : 1));      // compile-time after unrolling
#pragma unroll
            for (int e = 0; e < V; ++e) s4[k][e] += px[a][c].v[e];
        }
    if (live) {
        T* __restrict__ o4 = p4 + (((long)b * (H >> 2) + (y0 >> 2)) * (W >> 2) + (x0 >> 2)) * 9 * O + g * V;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            Vec8<T> v;
#pragma unroll
            for (int e = 0; e < V; ++e) v.v[e] = s4[k][e];
            v.store(o4 + (long)k * O);
        }
    }
    // 8 x 8 classes from the quadrant's own nine sums.  Rows of the block: quadrant row qa = 0 holds block rows 0 (first) and 1..3 (mid);
    // qa = 1 holds 4..6 (mid) and 7 (last): block class R takes the quadrant's row classes in `rsel` (bit = quadrant row class).
    T* __restrict__ o8 = p8 + (((long)b * (H >> 3) + bi) * (W >> 3) + bj) * 9 * O + g * V;
#pragma unroll
    for (int R = 0; R < 3; ++R) {
        // quadrant row classes (first = 1, mid = 2, last = 4) that belong to block row class R
        const unsigned rsel = R == 0 ? (qa == 0 ? 1u : 0u) : (R == 2 ? (qa == 1 ? 4u : 0u) : (qa == 0 ? 6u : 3u));
#pragma unroll
        for (int Cc = 0; Cc < 3; ++Cc) {
            const unsigned csel = Cc == 0 ? (qc == 0 ? 1u : 0u) : (Cc == 2 ? (qc == 1 ? 4u : 0u) : (qc == 0 ? 6u : 3u));
            float part[V];
#pragma unroll
            for (int e = 0; e < V; ++e) part[e] = 0.f;
#pragma unroll
            for (int ra = 0; ra < 3; ++ra)
#pragma unroll
                for (int rc = 0; rc < 3; ++rc) {
                    const bool in = ((rsel >> ra) & 1u) && ((csel >> rc) & 1u);
#pragma unroll
                    for (int e = 0; e < V; ++e) part[e] += in ? s4[ra * 3 + rc][e] : 0.f;
                }
#pragma unroll
            for (int e = 0; e < V; ++e) {
                part[e] += __shfl_xor(part[e], 1);
                part[e] += __shfl_xor(part[e], 2);
            }
            if (live && quad == 0) {
                Vec8<T> v;
#pragma unroll
                for (int e = 0; e < V; ++e) v.v[e] = part[e];
                v.store(o8 + (long)(R * 3 + Cc) * O);
            }
        }
    }
}

// dW[o][t][c] += dWtap_member[t * O + o][c] (members 0, 1: their per-tap 1x1 filter gradients) or the main part's (members 2, 3);
// dcomb = dWm [O][9][2C] | dWtap8 [9 O][C] | dWtap4 [9 O][C]
__global__ void conv2cls_fold_kernel(const float* __restrict__ dcomb, float* __restrict__ dw, int O, int C) {
    const int K = 4 * C;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)O * 9 * K) return;
    const int c = (int)(i % K);
    const long ot = i / K;
    const int t = (int)(ot % 9), o = (int)(ot / 9);
    const long nm = (long)O * 9 * 2 * C, nt = (long)9 * O * C;
    float g;
    if (c >= 2 * C) {
        g = dcomb[ot * 2 * C + (c - 2 * C)];
    } else {
        const int member = c / C;
        g = dcomb[nm + (long)member * nt + ((long)t * O + o) * C + (c - member * C)];
    }
    dw[i] += g;
}

// G[b, i, j, t * O + o] = sum of dy over the output pixels whose filter tap t = (r, s) reads the low-resolution pixel (i, j), from the
// class sums P[b, i', j', k * O + o] of the neighbouring blocks.  Per axis (F / M / L = first / mid / last class of a block):
//   r = 0:  M[i] + L[i] + F[i + 1]      r = 1:  F[i] + M[i] + L[i]      r = 2:  L[i - 1] + F[i] + M[i]
// f32 sums in this fixed order, one rounding.  8 channels per thread.
template <typename T>
__global__ void __launch_bounds__(256) conv2cls_tapsum_kernel(const T* __restrict__ pc, T* __restrict__ g, int B, int h, int w, int O) {
    const int og = O / 4;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)B * h * w * 9 * og) return;
    const int gch = (int)(idx % og);
    long q = idx / og;
    const int t = (int)(q % 9); q /= 9;
    const int j = (int)(q % w); q /= w;
    const int i = (int)(q % h);
    const int b = (int)(q / h);
    const int r = t / 3, s = t - r * 3;
    // per axis: up to three (block offset, class) terms
    int di[3], ci[3], dj[3], cj[3];
    if (r == 0) { di[0] = 0; ci[0] = 1; di[1] = 0; ci[1] = 2; di[2] = 1; ci[2] = 0; }
    else if (r == 1) { di[0] = 0; ci[0] = 0; di[1] = 0; ci[1] = 1; di[2] = 0; ci[2] = 2; }
    else { di[0] = -1; ci[0] = 2; di[1] = 0; ci[1] = 0; di[2] = 0; ci[2] = 1; }
    if (s == 0) { dj[0] = 0; cj[0] = 1; dj[1] = 0; cj[1] = 2; dj[2] = 1; cj[2] = 0; }
    else if (s == 1) { dj[0] = 0; cj[0] = 0; dj[1] = 0; cj[1] = 1; dj[2] = 0; cj[2] = 2; }
    else { dj[0] = -1; cj[0] = 2; dj[1] = 0; cj[1] = 0; dj[2] = 0; cj[2] = 1; }
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int ii = i + di[a];
        if (ii < 0 || ii >= h) continue;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int jj = j + dj[c];
            if (jj < 0 || jj >= w) continue;
            Vec8<T> v;
            v.load(pc + (((long)b * h + ii) * w + jj) * 9 * O + (long)(ci[a] * 3 + cj[c]) * O + gch * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += v.v[e];
        }
    }
    Vec8<T> out;
#pragma unroll
    for (int e = 0; e < 4; ++e) out.v[e] = acc[e];
    out.store(g + (((long)b * h + i) * w + j) * 9 * O + (long)t * O + gch * 4);
}

inline unsigned nblk(long n, int t) { return (unsigned)((n + t - 1) / t); }

}  // namespace

extern "C" int64_t mpn_conv2cls_comb_elems(int O, int C) {
    if (O <= 0 || C <= 0) return 0;
    return (int64_t)O * 9 * 2 * C + 2 * (int64_t)9 * O * 9 * C + 2 * (int64_t)9 * O * C;
}

extern "C" int mpn_conv2cls_combine(const float* w, float* comb, int O, int C, void* stream) {
    MPN_CHECK_ARG(w && comb && O > 0 && C > 0);
    const long n = mpn_conv2cls_comb_elems(O, C);
    hipLaunchKernelGGL(conv2cls_combine_kernel, dim3(nblk(n, 256)), dim3(256), 0, (hipStream_t)stream, w, comb, O, C);
    return mpn_launch_status();
}

extern "C" int mpn_conv2cls_expand(const float* m8, const float* m4, void* e, int B, int H, int W, int O, int dtype, void* stream) {
    MPN_CHECK_ARG(m8 && m4 && e && B > 0 && H > 0 && W > 0 && H % 8 == 0 && W % 8 == 0 && O > 0 && O % 8 == 0 && mpn_dtype_ok(dtype));
    const long n = (long)B * H * W * (O / 4);
    MPN_CHECK_ARG(n < 0x7fffffffL * 256L);
    MPN_DISPATCH_T(dtype, hipLaunchKernelGGL((conv2cls_expand_kernel<T>), dim3(nblk(n, 256)), dim3(256), 0, (hipStream_t)stream, m8, m4, (T*)e, B, H, W, O));
    return mpn_launch_status();
}

extern "C" int mpn_conv2cls_classsum(const float* t, float* m, int B, int h, int w, int O, void* stream) {
    MPN_CHECK_ARG(t && m && B > 0 && h > 0 && w > 0 && O > 0 && O % 4 == 0);
    const long n = (long)B * h * w * 9 * (O / 4);
    MPN_CHECK_ARG(n < 0x7fffffffL * 256L);
    hipLaunchKernelGGL(conv2cls_classsum_kernel, dim3(nblk(n, 256)), dim3(256), 0, (hipStream_t)stream, t, m, B, h, w, O);
    return mpn_launch_status();
}

extern "C" int mpn_conv2cls_pool(const void* dy, void* p8, void* p4, int B, int H, int W, int O, int dtype, void* stream) {
    MPN_CHECK_ARG(dy && p8 && p4 && B > 0 && H > 0 && W > 0 && H % 8 == 0 && W % 8 == 0 && O > 0 && O % 8 == 0 && mpn_dtype_ok(dtype));
    const long n = (long)B * (H / 8) * (W / 8) * (O / 4) * 4;
    MPN_CHECK_ARG(n < 0x7fffffffL * 256L);
    MPN_DISPATCH_T(dtype, hipLaunchKernelGGL((conv2cls_pool_kernel<T>), dim3(nblk(n, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)dy, (T*)p8, (T*)p4, B, H, W, O));
    return mpn_launch_status();
}

extern "C" int mpn_conv2cls_tapsum(const void* pc, void* g, int B, int h, int w, int O, int dtype, void* stream) {
    MPN_CHECK_ARG(pc && g && B > 0 && h > 0 && w > 0 && O > 0 && O % 8 == 0 && mpn_dtype_ok(dtype));
    const long n = (long)B * h * w * 9 * (O / 4);
    MPN_CHECK_ARG(n < 0x7fffffffL * 256L);
    MPN_DISPATCH_T(dtype, hipLaunchKernelGGL((conv2cls_tapsum_kernel<T>), dim3(nblk(n, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)pc, (T*)g, B, h, w, O));
    return mpn_launch_status();
}

extern "C" int mpn_conv2cls_fold(const float* dcomb, float* dw, int O, int C, void* stream) {
    MPN_CHECK_ARG(dcomb && dw && O > 0 && C > 0);
    const long n = (long)O * 9 * 4 * C;
    hipLaunchKernelGGL(conv2cls_fold_kernel, dim3(nblk(n, 256)), dim3(256), 0, (hipStream_t)stream, dcomb, dw, O, C);
    return mpn_launch_status();
}
