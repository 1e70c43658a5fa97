// bn.hip — BatchNorm2d (train / frozen) fused with ReLU and the bottleneck residual add, gfx950.
//
// Reference: nn.BatchNorm2d call sites network/fpn.py:15-26,43 used as F.relu(bn(conv(x)))
// (fpn.py:28-33,99) and bn3(...) += shortcut; relu (fpn.py:30-33); freeze_bn = eval mode
// (network/posenet.py:220-224).  Semantics: eps 1e-5, momentum 0.1, biased variance to normalise,
// unbiased variance into running_var (torch.nn.functional.batch_norm).
//
// All HBM-bound streaming kernels over dense pixel-major [P][Cs] tensors.  Thread mapping (all
// kernels): a block is GB channel-groups (16 bytes = 8 bf16 / 4 f32 each) x (256/GB) pixel lanes;
// a thread keeps ITS channel group's per-channel coefficients in registers (loaded once, as
// float4s) and walks several pixels, so the stream is pure 16-byte coalesced loads/stores.
//   train fwd : conv epilogue already produced per-tile (sum, sum^2) -> finalize -> bn_act pass
//   backward  : reduce pass (sum g, sum g*xhat) -> finalize (k1,k2,k3) -> apply pass dy=k1*g+k2*y+k3
// Algorithmic bytes per element (bf16): bn_act 2+2(+2 res); bwd_reduce 6; bwd_apply 6+2(+2 dres).
#include "common.h"
#include <stdlib.h>

namespace {

template <typename T> struct Geo {
    static constexpr int V = Vec16<T>::N;
    int G, GB, lanes, g, pl, c0;
    __device__ __forceinline__ Geo(int Cs) {
        G = Cs / V;
        GB = G < 256 ? G : 256;
        lanes = 256 / GB;
        g = blockIdx.y * GB + (threadIdx.x % GB);
        pl = threadIdx.x / GB;
        c0 = g * V;
    }
};

template <int V>
__device__ __forceinline__ void load_coef(const float* __restrict__ p, int c0, int C, float (&out)[V], float fill) {
#pragma unroll
    for (int k = 0; k < V; k += 4) {
        if (p != nullptr && c0 + k + 3 < C) {
            const float4 t = *reinterpret_cast<const float4*>(p + c0 + k);
            out[k] = t.x; out[k + 1] = t.y; out[k + 2] = t.z; out[k + 3] = t.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) out[k + j] = (p != nullptr && c0 + k + j < C) ? p[c0 + k + j] : fill;
        }
    }
}

// slice `sl` of 64: rows sl, sl+64, ... of the [n][C][2] partials of channel c, eight loads in flight (a dependent-load loop
// costs a memory round trip per row: 4-15 round trips on the 225..900-tile layers), added in row order
__device__ __forceinline__ void reduce_slices(const float* __restrict__ part, int n, int C, int c, int sl, double& s1, double& s2) {
    const float* col = part + (long)c * 2;
    const long stride = (long)C * 2;
    int t = sl;
    for (; t + 7 * 64 < n; t += 8 * 64) {
        float2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float2*>(col + (long)(t + u * 64) * stride);
#pragma unroll
        for (int u = 0; u < 8; ++u) { s1 += (double)v[u].x; s2 += (double)v[u].y; }
    }
    float2 w[8];
    int m = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int tt = t + u * 64;
        w[u] = tt < n ? *reinterpret_cast<const float2*>(col + (long)tt * stride) : make_float2(0.f, 0.f);
        m += tt < n ? 1 : 0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) if (u < m) { s1 += (double)w[u].x; s2 += (double)w[u].y; }
}

// One block of the training-mode finalize: 4 channels x 64 tile-slices (the reduction over up to ~14k tiles is the only work: spread
// it wide).
__device__ __forceinline__ void finalize_train_block(int blk, const float* __restrict__ stats, int tiles, int C, double count,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float* __restrict__ rm, float* __restrict__ rv, float momentum, float eps,
                                                     float* __restrict__ mean, float* __restrict__ invstd,
                                                     float* __restrict__ scale, float* __restrict__ shift, double (*sh)[4][4]) {
    const int cl = threadIdx.x & 3, sl = threadIdx.x >> 2;
    const int c = blk * 4 + cl;
    double s1 = 0.0, s2 = 0.0;
    // per-channel parameters do not depend on the reduction: fetch them first so their latency hides behind it
    const bool fin = sl == 0 && c < C;
    const float g = (fin && gamma) ? gamma[c] : 1.f, b = (fin && beta) ? beta[c] : 0.f;
    const float rm0 = (fin && rm) ? rm[c] : 0.f, rv0 = (fin && rv) ? rv[c] : 0.f;
    if (c < C) reduce_slices(stats, tiles, C, c, sl, s1, s2);
    // the 16 slices a wave holds for each channel: xor-butterfly over lane bits 2..5, then 4 wave partials through LDS
#pragma unroll
    for (int m = 4; m < 64; m <<= 1) { s1 += __shfl_xor(s1, m); s2 += __shfl_xor(s2, m); }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane < 4) { sh[0][wave][cl] = s1; sh[1][wave][cl] = s2; }
    __syncthreads();
    if (sl == 0 && c < C) {
        s1 = (sh[0][0][cl] + sh[0][1][cl]) + (sh[0][2][cl] + sh[0][3][cl]);
        s2 = (sh[1][0][cl] + sh[1][1][cl]) + (sh[1][2][cl] + sh[1][3][cl]);
        const double mu = s1 / count;
        double var = s2 / count - mu * mu;
        if (var < 0.0) var = 0.0;
        const float is = (float)(1.0 / sqrt(var + (double)eps));
        const float sc = g * is;
        mean[c] = (float)mu; invstd[c] = is;
        scale[c] = sc; shift[c] = b - (float)mu * sc;
        if (rm) rm[c] = (1.f - momentum) * rm0 + momentum * (float)mu;
        if (rv) {
            const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
            rv[c] = (1.f - momentum) * rv0 + momentum * (float)unb;
        }
    }
}

__global__ void bn_finalize_train_kernel(const float* __restrict__ stats, int tiles, int C, double count,
                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                         float* __restrict__ rm, float* __restrict__ rv, float momentum, float eps,
                                         float* __restrict__ mean, float* __restrict__ invstd,
                                         float* __restrict__ scale, float* __restrict__ shift) {
    __shared__ double sh[2][4][4];
    finalize_train_block(blockIdx.x, stats, tiles, C, count, gamma, beta, rm, rv, momentum, eps, mean, invstd, scale, shift, sh);
}

__global__ void bn_finalize_eval_kernel(int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                                        const float* __restrict__ rm, const float* __restrict__ rv, float eps,
                                        float* __restrict__ mean, float* __restrict__ invstd,
                                        float* __restrict__ scale, float* __restrict__ shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float is = 1.0f / sqrtf(rv[c] + eps);
    const float sc = gamma[c] * is;
    mean[c] = rm[c]; invstd[c] = is; scale[c] = sc; shift[c] = beta[c] - rm[c] * sc;
}

// z = act(y*scale + shift + res)
template <typename T>
__global__ void __launch_bounds__(256) bn_act_kernel(const T* __restrict__ y, const T* __restrict__ res, T* __restrict__ z,
                                                     const float* __restrict__ scale, const float* __restrict__ shift,
                                                     long P, int C, int Cs, int relu, int iters, unsigned char* __restrict__ mask) {
    constexpr int V = Vec16<T>::N;
    const Geo<T> q(Cs);
    float sc[V], sf[V];
    load_coef<V>(scale, q.c0, C, sc, 0.f);
    load_coef<V>(shift, q.c0, C, sf, 0.f);
    const long p0 = (long)blockIdx.x * q.lanes * iters + q.pl;
    for (int it = 0; it < iters; ++it) {
        const long p = p0 + (long)it * q.lanes;
        if (p >= P) break;
        const long off = p * Cs + q.c0;
        Vec16<T> a; a.load(y + off);
        Vec16<T> r;
        if (res) r.load(res + off);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            float x = a.v[k] * sc[k] + sf[k];
            if (res) x += r.v[k];
            if (relu) x = fmaxf(x, 0.f);
            a.v[k] = (q.c0 + k < C) ? x : 0.f;
        }
        a.store(z + off);
        if (mask) {                                   // sign bits of the value as stored (rounded to the element type)
            unsigned bits = 0;
#pragma unroll
            for (int k = 0; k < V; ++k) bits |= (Elem<T>::round(a.v[k]) > 0.f ? 1u : 0u) << k;
            mask[p * q.G + q.g] = (unsigned char)bits;
        }
    }
}

// stage 1 of backward: per (pixel-chunk, channel) partial sums of g and g*xhat
template <typename T>
__global__ void __launch_bounds__(256) bn_bwd_reduce_kernel(const T* __restrict__ dz, const T* __restrict__ z, const T* __restrict__ y,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd,
                                                            const float* __restrict__ mscale, const float* __restrict__ mshift,
                                                            float* __restrict__ partial, long P, int C, int Cs, int relu, int chunk_pix) {
    constexpr int V = Vec16<T>::N;
    __shared__ float sh[256][2 * V + 1];
    const Geo<T> q(Cs);
    const long p_begin = (long)blockIdx.x * chunk_pix;
    long p_end = p_begin + chunk_pix; if (p_end > P) p_end = P;
    float s1[V], s2[V], mu[V], is[V], sc[V], sf[V];
    load_coef<V>(mean, q.c0, C, mu, 0.f);
    load_coef<V>(invstd, q.c0, C, is, 0.f);
    const bool remask = relu && z == nullptr;       // ReLU mask recomputed from y with bn_act's own expression (no z read)
    if (remask) { load_coef<V>(mscale, q.c0, C, sc, 0.f); load_coef<V>(mshift, q.c0, C, sf, 0.f); }
#pragma unroll
    for (int k = 0; k < V; ++k) { s1[k] = 0.f; s2[k] = 0.f; }
    for (long p = p_begin + q.pl; p < p_end; p += q.lanes) {
        const long off = p * Cs + q.c0;
        Vec16<T> d, o, x;
        d.load(dz + off); x.load(y + off);
        if (relu && !remask) o.load(z + off);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            float gk = d.v[k];
            if (remask) o.v[k] = x.v[k] * sc[k] + sf[k];
            if (relu && !(o.v[k] > 0.f)) gk = 0.f;
            s1[k] += gk;
            s2[k] += gk * ((x.v[k] - mu[k]) * is[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < V; ++k) { sh[threadIdx.x][k] = s1[k]; sh[threadIdx.x][V + k] = s2[k]; }
    __syncthreads();
    if (q.pl == 0) {
        for (int l = 1; l < q.lanes; ++l) {
#pragma unroll
            for (int k = 0; k < V; ++k) {
                s1[k] += sh[threadIdx.x + l * q.GB][k];
                s2[k] += sh[threadIdx.x + l * q.GB][V + k];
            }
        }
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const int c = q.c0 + k;
            if (c < C) *reinterpret_cast<float2*>(partial + ((long)blockIdx.x * C + c) * 2) = make_float2(s1[k], s2[k]);
        }
    }
}

// reduce partials; dgamma/dbeta += ; coefficient vectors so that dy = k1*g + k2*y + k3:
//   train : k1 = gamma*is, k2 = -gamma*is*is*b, k3 = gamma*is*(mu*is*b - a)   (a = sum g / n, b = sum g*xhat / n)
//   frozen: k1 = gamma*is, k2 = k3 = 0
__global__ void bn_bwd_finalize_kernel(const float* __restrict__ partial, int chunks, int C, double count,
                                       const float* __restrict__ gamma, const float* __restrict__ mean, const float* __restrict__ invstd,
                                       int train, float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ coef) {
    __shared__ double sh[2][4][4];
    const int cl = threadIdx.x & 3, sl = threadIdx.x >> 2;
    const int c = blockIdx.x * 4 + cl;
    double s1 = 0.0, s2 = 0.0;
    const bool fin = sl == 0 && c < C;           // parameters first: their latency hides behind the reduction
    const float is = fin ? invstd[c] : 0.f, gm = (fin && gamma) ? gamma[c] : 1.f, mu = fin ? mean[c] : 0.f;
    const float db0 = (fin && dbeta) ? dbeta[c] : 0.f, dg0 = (fin && dgamma) ? dgamma[c] : 0.f;
    if (c < C) reduce_slices(partial, chunks, C, c, sl, s1, s2);
#pragma unroll
    for (int m = 4; m < 64; m <<= 1) { s1 += __shfl_xor(s1, m); s2 += __shfl_xor(s2, m); }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane < 4) { sh[0][wave][cl] = s1; sh[1][wave][cl] = s2; }
    __syncthreads();
    if (sl == 0 && c < C) {
        s1 = (sh[0][0][cl] + sh[0][1][cl]) + (sh[0][2][cl] + sh[0][3][cl]);
        s2 = (sh[1][0][cl] + sh[1][1][cl]) + (sh[1][2][cl] + sh[1][3][cl]);
        if (dbeta) dbeta[c] = db0 + (float)s1;
        if (dgamma) dgamma[c] = dg0 + (float)s2;
        if (coef) {
            const float a = (float)(s1 / count), b = (float)(s2 / count);
            coef[c] = gm * is;
            coef[C + c] = train ? -gm * is * is * b : 0.f;
            coef[2 * C + c] = train ? gm * is * (mu * is * b - a) : 0.f;
        }
    }
}

// dy = k1*g + k2*y + k3 ; dres (+)= g ; g = dz * (z > 0)
template <typename T>
__global__ void __launch_bounds__(256) bn_bwd_apply_kernel(const T* __restrict__ dz, const T* __restrict__ z, const T* __restrict__ y,
                                                           const float* __restrict__ k1p, const float* __restrict__ k2p,
                                                           const float* __restrict__ k3p, const float* __restrict__ mscale,
                                                           const float* __restrict__ mshift, T* __restrict__ dy, T* __restrict__ dres,
                                                           int dres_acc, long P, int C, int Cs, int relu, int iters,
                                                           const unsigned char* __restrict__ mask) {
    constexpr int V = Vec16<T>::N;
    const Geo<T> q(Cs);
    float k1[V], k2[V], k3[V];
    load_coef<V>(k1p, q.c0, C, k1, 0.f);
    load_coef<V>(k2p, q.c0, C, k2, 0.f);
    load_coef<V>(k3p, q.c0, C, k3, 0.f);
    const bool bits = relu && mask != nullptr;      // ReLU mask from the forward's sign bits (1 byte per 16-byte chunk)
    const bool remask = relu && !bits && z == nullptr;
    float sc[V], sf[V];
    if (remask) { load_coef<V>(mscale, q.c0, C, sc, 0.f); load_coef<V>(mshift, q.c0, C, sf, 0.f); }
    const bool need_y = ((dy != nullptr) && (k2p != nullptr)) || remask;
    const long p0 = (long)blockIdx.x * q.lanes * iters + q.pl;
    for (int it = 0; it < iters; ++it) {
        const long p = p0 + (long)it * q.lanes;
        if (p >= P) break;
        const long off = p * Cs + q.c0;
        Vec16<T> d, o, x, out, r;
        d.load(dz + off);
        if (need_y) x.load(y + off);
        if (relu && !remask && !bits) o.load(z + off);
        unsigned mb = 0;
        if (bits) mb = mask[p * q.G + q.g];
        if (dres && dres_acc) r.load(dres + off);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            float gk = d.v[k];
            if (remask) o.v[k] = x.v[k] * sc[k] + sf[k];
            if (bits) o.v[k] = ((mb >> k) & 1u) ? 1.f : 0.f;
            if (relu && !(o.v[k] > 0.f)) gk = 0.f;
            const bool live = q.c0 + k < C;
            out.v[k] = live ? (k1[k] * gk + (k2p ? k2[k] * x.v[k] : 0.f) + k3[k]) : 0.f;
            if (dres) r.v[k] = (dres_acc ? r.v[k] : 0.f) + (live ? gk : 0.f);
        }
        if (dy) out.store(dy + off);
        if (dres) r.store(dres + off);
    }
}

inline int geo_lanes(int Cs, int V) { const int G = Cs / V; return 256 / (G < 256 ? G : 256); }
inline int geo_yblocks(int Cs, int V) { const int G = Cs / V; return G <= 256 ? 1 : G / 256; }
inline bool geo_ok(int Cs, int V) { const int G = Cs / V; return Cs % V == 0 && G > 0 && (G & (G - 1)) == 0; }

// pixels per block for the streaming kernels: ~8192 blocks per launch (32 per CU; measured -0.14 ms/step against 2048), <= 32 pixels per thread
inline int pick_iters(long P, int lanes) {
    static const long blocks = mpn_tune("MPN_BN_BLOCKS", 8192);   // blocks per launch (swept 1k..32k at step level)
    long it = P / ((long)lanes * blocks);
    if (it < 1) it = 1;
    if (it > 32) it = 32;
    return (int)it;
}
inline int reduce_chunk(long P, int lanes) {
    long c = P / 512;                  // ~512 chunks
    const long lo = (long)lanes * 4, hi = 4096;
    if (c < lo) c = lo;
    if (c > hi) c = hi;
    return (int)((c + lanes - 1) / lanes * lanes);
}

}  // namespace

extern "C" int mpn_bn_finalize_train(const float* stats, int tiles, int C, int64_t count, const float* gamma,
                                     const float* beta, float* running_mean, float* running_var, float momentum,
                                     float eps, float* mean, float* invstd, float* scale, float* shift, void* stream) {
    MPN_CHECK_ARG(stats && tiles > 0 && C > 0 && count > 0 && mean && invstd && scale && shift);
    hipLaunchKernelGGL(bn_finalize_train_kernel, dim3((C + 3) / 4), dim3(256), 0, (hipStream_t)stream, stats, tiles, C,
                       (double)count, gamma, beta, running_mean, running_var, momentum, eps, mean, invstd, scale, shift);
    return mpn_launch_status();
}

extern "C" int mpn_bn_finalize_eval(int C, const float* gamma, const float* beta, const float* running_mean,
                                    const float* running_var, float eps, float* mean, float* invstd,
                                    float* scale, float* shift, void* stream) {
    MPN_CHECK_ARG(C > 0 && gamma && beta && running_mean && running_var && mean && invstd && scale && shift);
    hipLaunchKernelGGL(bn_finalize_eval_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, C, gamma, beta,
                       running_mean, running_var, eps, mean, invstd, scale, shift);
    return mpn_launch_status();
}

extern "C" int mpn_bn_act_forward(const void* y, const void* res, void* z, const float* scale, const float* shift,
                                  int64_t P, int C, int Cs, int relu, int dtype, uint8_t* mask, void* stream) {
    MPN_CHECK_ARG(y && z && scale && shift && P > 0 && C > 0 && Cs >= C && (!mask || relu));
    const int V = dtype == MPN_F32 ? 4 : 8;
    MPN_CHECK_ARG(geo_ok(Cs, V) && C % 4 == 0);
    const int lanes = geo_lanes(Cs, V), iters = pick_iters(P, lanes);
    dim3 grid((unsigned)((P + (long)lanes * iters - 1) / ((long)lanes * iters)), (unsigned)geo_yblocks(Cs, V));
    MPN_DISPATCH_T(dtype, hipLaunchKernelGGL((bn_act_kernel<T>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)y, (const T*)res,
                           (T*)z, scale, shift, (long)P, C, Cs, relu, iters, (unsigned char*)mask));
    return mpn_launch_status();
}

extern "C" int mpn_bn_bwd_chunks(int64_t P, int Cs, int dtype) {
    const int V = dtype == MPN_F32 ? 4 : 8;
    if (!geo_ok(Cs, V) || P <= 0) return MPN_E_BADARG;
    const int chunk = reduce_chunk(P, geo_lanes(Cs, V));
    return (int)((P + chunk - 1) / chunk);
}

extern "C" int mpn_bn_bwd_reduce(const void* dz, const void* z, const void* y, const float* mean, const float* invstd,
                                 const float* mask_scale, const float* mask_shift, float* partial, int chunks, int64_t P, int C, int Cs, int relu, int dtype, void* stream) {
    MPN_CHECK_ARG(dz && y && mean && invstd && partial && P > 0 && C > 0 && Cs >= C);
    MPN_CHECK_ARG(!relu || z || (mask_scale && mask_shift && y));
    const int V = dtype == MPN_F32 ? 4 : 8;
    MPN_CHECK_ARG(geo_ok(Cs, V) && C % 4 == 0);
    const int chunk = reduce_chunk(P, geo_lanes(Cs, V));
    MPN_CHECK_ARG(chunks == (int)((P + chunk - 1) / chunk));
    dim3 grid((unsigned)chunks, (unsigned)geo_yblocks(Cs, V));
    MPN_DISPATCH_T(dtype, hipLaunchKernelGGL((bn_bwd_reduce_kernel<T>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)dz, (const T*)z,
                           (const T*)y, mean, invstd, mask_scale, mask_shift, partial, (long)P, C, Cs, relu, chunk));
    return mpn_launch_status();
}

extern "C" int mpn_bn_bwd_finalize(const float* partial, int chunks, int C, int64_t count, const float* gamma, const float* mean,
                                   const float* invstd, int train, float* dgamma, float* dbeta, float* coef, void* stream) {
    MPN_CHECK_ARG(partial && chunks > 0 && C > 0 && count > 0);
    MPN_CHECK_ARG(!coef || (mean && invstd));
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, (hipStream_t)stream, partial, chunks, C,
                       (double)count, gamma, mean, invstd, train, dgamma, dbeta, coef);
    return mpn_launch_status();
}

extern "C" int mpn_bn_bwd_apply(const void* dz, const void* z, const void* y, const float* k1, const float* k2, const float* k3,
                                const float* mask_scale, const float* mask_shift, void* dy, void* dres, int dres_accumulate, int64_t P, int C, int Cs, int relu, int dtype,
                                const uint8_t* mask, void* stream) {
    MPN_CHECK_ARG(dz && (dy || dres) && P > 0 && C > 0 && Cs >= C);
    MPN_CHECK_ARG(!dy || k1);
    MPN_CHECK_ARG(!(dy && k2) || y);
    MPN_CHECK_ARG(!relu || z || mask || (mask_scale && mask_shift && y));
    const int V = dtype == MPN_F32 ? 4 : 8;
    MPN_CHECK_ARG(geo_ok(Cs, V) && C % 4 == 0);
    const int lanes = geo_lanes(Cs, V), iters = pick_iters(P, lanes);
    dim3 grid((unsigned)((P + (long)lanes * iters - 1) / ((long)lanes * iters)), (unsigned)geo_yblocks(Cs, V));
    MPN_DISPATCH_T(dtype, hipLaunchKernelGGL((bn_bwd_apply_kernel<T>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)dz, (const T*)z,
                           (const T*)y, k1, k2, k3, mask_scale, mask_shift, (T*)dy, (T*)dres, dres_accumulate, (long)P, C, Cs, relu, iters,
                           (const unsigned char*)mask));
    return mpn_launch_status();
}
