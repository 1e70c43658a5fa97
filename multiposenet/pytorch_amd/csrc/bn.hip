// bn.hip — BatchNorm2d (train / frozen) fused with ReLU and the bottleneck residual add, gfx950.
//
// Reference: nn.BatchNorm2d call sites network/fpn.py:15-26,43 used as F.relu(bn(conv(x)))
// (fpn.py:28-33,99) and bn3(...) += shortcut; relu (fpn.py:30-33); freeze_bn = eval mode
// (network/posenet.py:220-224).  Semantics: eps 1e-5, momentum 0.1, biased variance to normalise,
// unbiased variance into running_var (torch.nn.functional.batch_norm).
//
// All of these are HBM-bound streaming kernels: 16-byte vector loads/stores, one pass each.
//   train fwd : conv epilogue already produced per-tile (sum, sum^2) -> finalize -> bn_act pass
//   backward  : reduce pass (sum g, sum g*xhat) -> finalize -> apply pass
// Algorithmic bytes per element (bf16): bn_act 2+2(+2 res); bwd_reduce 6; bwd_apply 6+2(+2 dres).
#include "common.h"

namespace {

constexpr int BN_CHUNK_PIX = 2048;

__global__ void bn_finalize_train_kernel(const float* __restrict__ stats, int tiles, int C, double count,
                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                         float* __restrict__ rm, float* __restrict__ rv, float momentum, float eps,
                                         float* __restrict__ mean, float* __restrict__ invstd,
                                         float* __restrict__ scale, float* __restrict__ shift) {
    // block = 64 channels x 4 tile-slices
    __shared__ double sh[2][4][64];
    const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    double s1 = 0.0, s2 = 0.0;
    if (c < C) {
        for (int t = sl; t < tiles; t += 4) {
            s1 += (double)stats[((long)t * C + c) * 2 + 0];
            s2 += (double)stats[((long)t * C + c) * 2 + 1];
        }
    }
    sh[0][sl][cl] = s1; sh[1][sl][cl] = s2;
    __syncthreads();
    if (sl == 0 && c < C) {
        s1 = sh[0][0][cl] + sh[0][1][cl] + sh[0][2][cl] + sh[0][3][cl];
        s2 = sh[1][0][cl] + sh[1][1][cl] + sh[1][2][cl] + sh[1][3][cl];
        const double mu = s1 / count;
        double var = s2 / count - mu * mu;
        if (var < 0.0) var = 0.0;
        const float is = (float)(1.0 / sqrt(var + (double)eps));
        const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
        mean[c] = (float)mu; invstd[c] = is;
        const float sc = g * is;
        scale[c] = sc; shift[c] = b - (float)mu * sc;
        if (rm) rm[c] = (1.f - momentum) * rm[c] + momentum * (float)mu;
        if (rv) {
            const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
            rv[c] = (1.f - momentum) * rv[c] + momentum * (float)unb;
        }
    }
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

// z = act(y*scale + shift + res); one 16-byte vector per thread
template <typename T>
__global__ void bn_act_kernel(const T* __restrict__ y, const T* __restrict__ res, T* __restrict__ z,
                              const float* __restrict__ scale, const float* __restrict__ shift,
                              long nvec, int C, int Cs, int relu) {
    constexpr int V = Vec16<T>::N;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nvec) return;
    const int G = Cs / V;
    const int c0 = (int)(i % G) * V;
    Vec16<T> a; a.load(y + i * V);
    Vec16<T> r;
    if (res) r.load(res + i * V);
#pragma unroll
    for (int k = 0; k < V; ++k) {
        const int c = c0 + k;
        float x = 0.f;
        if (c < C) {
            x = a.v[k] * scale[c] + shift[c];
            if (res) x += r.v[k];
            if (relu) x = fmaxf(x, 0.f);
        }
        a.v[k] = x;
    }
    a.store(z + i * V);
}

// stage 1 of backward: per (pixel-chunk, channel) partial sums of g and g*xhat
template <typename T>
__global__ void bn_bwd_reduce_kernel(const T* __restrict__ dz, const T* __restrict__ z, const T* __restrict__ y,
                                     const float* __restrict__ mean, const float* __restrict__ invstd,
                                     float* __restrict__ partial, long P, int C, int Cs, int relu) {
    constexpr int V = Vec16<T>::N;
    __shared__ float sh[256][2 * V + 1];
    const int G = Cs / V;                       // channel groups per pixel (power of two)
    const int GB = G < 256 ? G : 256;           // groups handled by this block
    const int lanes = 256 / GB;                 // pixel lanes
    const int g = blockIdx.y * GB + (threadIdx.x % GB);
    const int pl = threadIdx.x / GB;
    const int c0 = g * V;
    const long p_begin = (long)blockIdx.x * BN_CHUNK_PIX;
    long p_end = p_begin + BN_CHUNK_PIX; if (p_end > P) p_end = P;
    float s1[V], s2[V], mu[V], is[V];
#pragma unroll
    for (int k = 0; k < V; ++k) {
        s1[k] = 0.f; s2[k] = 0.f;
        const int c = c0 + k;
        mu[k] = c < C ? mean[c] : 0.f; is[k] = c < C ? invstd[c] : 0.f;
    }
    for (long p = p_begin + pl; p < p_end; p += lanes) {
        const long off = p * Cs + c0;
        Vec16<T> d, o, x;
        d.load(dz + off); x.load(y + off);
        if (relu) o.load(z + off);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            float gk = d.v[k];
            if (relu && !(o.v[k] > 0.f)) gk = 0.f;
            s1[k] += gk;
            s2[k] += gk * ((x.v[k] - mu[k]) * is[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < V; ++k) { sh[threadIdx.x][k] = s1[k]; sh[threadIdx.x][V + k] = s2[k]; }
    __syncthreads();
    if (pl == 0) {
        for (int l = 1; l < lanes; ++l) {
#pragma unroll
            for (int k = 0; k < V; ++k) {
                s1[k] += sh[threadIdx.x + l * GB][k];
                s2[k] += sh[threadIdx.x + l * GB][V + k];
            }
        }
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const int c = c0 + k;
            if (c < C) {
                partial[((long)blockIdx.x * C + c) * 2 + 0] = s1[k];
                partial[((long)blockIdx.x * C + c) * 2 + 1] = s2[k];
            }
        }
    }
}

__global__ void bn_bwd_finalize_kernel(const float* __restrict__ partial, int chunks, int C, double count,
                                       float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ coef) {
    __shared__ double sh[2][4][64];
    const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    double s1 = 0.0, s2 = 0.0;
    if (c < C) {
        for (int t = sl; t < chunks; t += 4) {
            s1 += (double)partial[((long)t * C + c) * 2 + 0];
            s2 += (double)partial[((long)t * C + c) * 2 + 1];
        }
    }
    sh[0][sl][cl] = s1; sh[1][sl][cl] = s2;
    __syncthreads();
    if (sl == 0 && c < C) {
        s1 = sh[0][0][cl] + sh[0][1][cl] + sh[0][2][cl] + sh[0][3][cl];
        s2 = sh[1][0][cl] + sh[1][1][cl] + sh[1][2][cl] + sh[1][3][cl];
        if (dbeta) dbeta[c] += (float)s1;
        if (dgamma) dgamma[c] += (float)s2;
        if (coef) { coef[c * 2 + 0] = (float)(s1 / count); coef[c * 2 + 1] = (float)(s2 / count); }
    }
}

template <typename T>
__global__ void bn_bwd_apply_kernel(const T* __restrict__ dz, const T* __restrict__ z, const T* __restrict__ y,
                                    const float* __restrict__ mean, const float* __restrict__ invstd,
                                    const float* __restrict__ gamma, const float* __restrict__ coef,
                                    T* __restrict__ dy, T* __restrict__ dres, int dres_acc,
                                    long nvec, int C, int Cs, int relu) {
    constexpr int V = Vec16<T>::N;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nvec) return;
    const int G = Cs / V;
    const int c0 = (int)(i % G) * V;
    Vec16<T> d, o, x, out, r;
    d.load(dz + i * V);
    if (dy) x.load(y + i * V);
    if (relu) o.load(z + i * V);
    if (dres && dres_acc) r.load(dres + i * V);
#pragma unroll
    for (int k = 0; k < V; ++k) {
        const int c = c0 + k;
        float gk = d.v[k];
        if (relu && !(o.v[k] > 0.f)) gk = 0.f;
        float v = 0.f;
        if (c < C && dy) {
            const float is = invstd[c];
            const float gm = gamma ? gamma[c] : 1.f;
            if (coef) {
                const float xh = (x.v[k] - mean[c]) * is;
                v = gm * is * (gk - coef[c * 2 + 0] - xh * coef[c * 2 + 1]);
            } else {
                v = gk * gm * is;
            }
        }
        out.v[k] = v;
        if (dres) r.v[k] = (dres_acc ? r.v[k] : 0.f) + (c < C ? gk : 0.f);
    }
    if (dy) out.store(dy + i * V);
    if (dres) r.store(dres + i * V);
}

}  // namespace

extern "C" int mpn_bn_finalize_train(const float* stats, int tiles, int C, int64_t count, const float* gamma,
                                     const float* beta, float* running_mean, float* running_var, float momentum,
                                     float eps, float* mean, float* invstd, float* scale, float* shift, void* stream) {
    MPN_CHECK_ARG(stats && tiles > 0 && C > 0 && count > 0 && mean && invstd && scale && shift);
    hipLaunchKernelGGL(bn_finalize_train_kernel, dim3((C + 63) / 64), dim3(256), 0, (hipStream_t)stream, stats, tiles, C,
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
                                  int64_t P, int C, int Cs, int relu, int dtype, void* stream) {
    MPN_CHECK_ARG(y && z && scale && shift && P > 0 && C > 0 && Cs >= C && Cs % 8 == 0);
    if (dtype == MPN_F32) {
        const long nvec = (long)P * Cs / 4;
        hipLaunchKernelGGL(bn_act_kernel<float>, dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                           (const float*)y, (const float*)res, (float*)z, scale, shift, nvec, C, Cs, relu);
    } else {
        const long nvec = (long)P * Cs / 8;
        hipLaunchKernelGGL(bn_act_kernel<bf16_t>, dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)y, (const bf16_t*)res, (bf16_t*)z, scale, shift, nvec, C, Cs, relu);
    }
    return mpn_launch_status();
}

extern "C" int mpn_bn_bwd_chunks(int64_t P, int C) {
    (void)C;
    return (int)((P + BN_CHUNK_PIX - 1) / BN_CHUNK_PIX);
}

extern "C" int mpn_bn_bwd_reduce(const void* dz, const void* z, const void* y, const float* mean, const float* invstd,
                                 float* partial, int chunks, int64_t P, int C, int Cs, int relu, int dtype, void* stream) {
    MPN_CHECK_ARG(dz && y && mean && invstd && partial && P > 0 && C > 0 && Cs >= C);
    MPN_CHECK_ARG(!relu || z);
    MPN_CHECK_ARG(chunks == (int)((P + BN_CHUNK_PIX - 1) / BN_CHUNK_PIX));
    const int V = dtype == MPN_F32 ? 4 : 8;
    const int G = Cs / V;
    MPN_CHECK_ARG(Cs % V == 0 && (G & (G - 1)) == 0);
    const int GB = G < 256 ? G : 256;
    dim3 grid((unsigned)chunks, (unsigned)(G / GB));
    if (dtype == MPN_F32)
        hipLaunchKernelGGL(bn_bwd_reduce_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)dz, (const float*)z,
                           (const float*)y, mean, invstd, partial, (long)P, C, Cs, relu);
    else
        hipLaunchKernelGGL(bn_bwd_reduce_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dz, (const bf16_t*)z,
                           (const bf16_t*)y, mean, invstd, partial, (long)P, C, Cs, relu);
    return mpn_launch_status();
}

extern "C" int mpn_bn_bwd_finalize(const float* partial, int chunks, int C, int64_t count, float* dgamma, float* dbeta,
                                   float* coef, void* stream) {
    MPN_CHECK_ARG(partial && chunks > 0 && C > 0 && count > 0);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 63) / 64), dim3(256), 0, (hipStream_t)stream, partial, chunks, C,
                       (double)count, dgamma, dbeta, coef);
    return mpn_launch_status();
}

extern "C" int mpn_bn_bwd_apply(const void* dz, const void* z, const void* y, const float* mean, const float* invstd,
                                const float* gamma, const float* coef, void* dy, void* dres, int dres_accumulate,
                                int64_t P, int C, int Cs, int relu, int dtype, void* stream) {
    MPN_CHECK_ARG(dz && (dy || dres) && P > 0 && C > 0 && Cs >= C && Cs % 8 == 0);
    MPN_CHECK_ARG(!dy || (y && mean && invstd));
    MPN_CHECK_ARG(!relu || z);
    if (dtype == MPN_F32) {
        const long nvec = (long)P * Cs / 4;
        hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                           (const float*)dz, (const float*)z, (const float*)y, mean, invstd, gamma, coef, (float*)dy, (float*)dres,
                           dres_accumulate, nvec, C, Cs, relu);
    } else {
        const long nvec = (long)P * Cs / 8;
        hipLaunchKernelGGL(bn_bwd_apply_kernel<bf16_t>, dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)dz, (const bf16_t*)z, (const bf16_t*)y, mean, invstd, gamma, coef, (bf16_t*)dy,
                           (bf16_t*)dres, dres_accumulate, nvec, C, Cs, relu);
    }
    return mpn_launch_status();
}
