// weight_prep.hip — per-step parameter preparation + fused Adam for gfx950.
//
// Master parameters live in ONE flat f32 arena whose per-tensor physical layout is
// [Cout][R][S][Cin] (a channels_last view of the reference's [Cout,Cin,R,S] state_dict entry, so
// checkpoint keys/shapes are unchanged: network/net_utils.py:32-34).  Each step the compute-dtype
// operand copies are refreshed from it:
//   * forward operand  : same layout, cast to bf16 (one launch over the whole arena)
//   * dgrad operand    : Wt[Cin][R][S][Cout_pad]  (LDS-tiled 32x32 transpose, one launch per conv)
//   * stem (7x7x3)     : [64][7][32] packing that turns the Cin=3 stem into a Cin=32 "row" conv
// Adam follows torch.optim.Adam (training/multipose_keypoint_train.py:106-110, trainer.py:259).
#include "common.h"
#include "build_id.h"

namespace {

__global__ void cast_f32_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, long n) {
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (i >= n) return;
    if (i + 7 < n) {
        const float4 a = *reinterpret_cast<const float4*>(src + i);
        const float4 b = *reinterpret_cast<const float4*>(src + i + 4);
        uint4 o;
        o.x = pack_bf16x2(a.x, a.y);
        o.y = pack_bf16x2(a.z, a.w);
        o.z = pack_bf16x2(b.x, b.y);
        o.w = pack_bf16x2(b.z, b.w);
        *reinterpret_cast<uint4*>(dst + i) = o;
    } else {
        for (long k = i; k < n; ++k) dst[k] = f2bf(src[k]);
    }
}

__global__ void cast_f32_f16_kernel(const float* __restrict__ src, f16_t* __restrict__ dst, long n) {
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (i >= n) return;
    if (i + 7 < n) {
        Vec16<f16_t> o;
        const float4 a = *reinterpret_cast<const float4*>(src + i);
        const float4 b = *reinterpret_cast<const float4*>(src + i + 4);
        o.v[0] = a.x; o.v[1] = a.y; o.v[2] = a.z; o.v[3] = a.w; o.v[4] = b.x; o.v[5] = b.y; o.v[6] = b.z; o.v[7] = b.w;
        o.store(dst + i);
    } else {
        for (long k = i; k < n; ++k) dst[k] = (f16_t)src[k];
    }
}

// grid (ceil(Cin/32), ceil(Cout_pad/32), RS), block (32, 8)
template <typename T>
__global__ void weight_transpose_kernel(const float* __restrict__ w, T* __restrict__ wt, int Cout, int RS, int Cin, int Cout_pad) {
    __shared__ float tile[32][33];
    const int rs = blockIdx.z;
    const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
    for (int k = threadIdx.y; k < 32; k += 8) {
        const int co = co0 + k, ci = ci0 + threadIdx.x;
        float v = 0.f;
        if (co < Cout && ci < Cin) v = w[((long)co * RS + rs) * Cin + ci];
        tile[k][threadIdx.x] = v;
    }
    __syncthreads();
    for (int k = threadIdx.y; k < 32; k += 8) {
        const int ci = ci0 + k, co = co0 + threadIdx.x;
        if (ci < Cin && co < Cout_pad) Elem<T>::st(wt + ((long)ci * RS + rs) * Cout_pad + co, tile[threadIdx.x][k]);
    }
}

// every layer's dgrad operand in one launch: table row l = {src offset (floats), dst offset (elements), Cout, RS, Cin,
// Cout_pad, first block, blocks along Cin}; block -> layer by binary search over the first-block column
template <typename T>
__global__ void weight_transpose_batched_kernel(const float* __restrict__ arena, T* __restrict__ dst, const long* __restrict__ table, int nlayers) {
    __shared__ float tile[32][33];
    const long b = blockIdx.x;
    int lo = 0, hi = nlayers - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid * 8 + 6] <= b) lo = mid; else hi = mid - 1;
    }
    const long* row = table + lo * 8;
    const float* __restrict__ w = arena + row[0];
    T* __restrict__ wt = dst + row[1];
    const int Cout = (int)row[2], RS = (int)row[3], Cin = (int)row[4], Cout_pad = (int)row[5], gx = (int)row[7];
    const int gy = (Cout_pad + 31) / 32;
    long lb = b - row[6];
    const int bx = (int)(lb % gx); lb /= gx;
    const int by = (int)(lb % gy); lb /= gy;
    const int rs = (int)lb;
    const int ci0 = bx * 32, co0 = by * 32;
    for (int k = threadIdx.y; k < 32; k += 8) {
        const int co = co0 + k, ci = ci0 + threadIdx.x;
        float v = 0.f;
        if (co < Cout && ci < Cin) v = w[((long)co * RS + rs) * Cin + ci];
        tile[k][threadIdx.x] = v;
    }
    __syncthreads();
    for (int k = threadIdx.y; k < 32; k += 8) {
        const int ci = ci0 + k, co = co0 + threadIdx.x;
        if (ci < Cin && co < Cout_pad) Elem<T>::st(wt + ((long)ci * RS + rs) * Cout_pad + co, tile[threadIdx.x][k]);
    }
}

template <typename T>
__global__ void weight_pad_k_kernel(const float* __restrict__ w, T* __restrict__ dst, int Cout, int K, int Kpad) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)Cout * Kpad) return;
    const int co = (int)(i / Kpad), k = (int)(i - (long)co * Kpad);
    Elem<T>::st(dst + i, k < K ? w[(long)co * K + k] : 0.f);
}

// W[co][r][s][c] (7x7x3) -> packed[co][r][s*4 + c], slots with c==3 or s==7 are zero
template <typename T>
__global__ void stem_pack_weight_kernel(const float* __restrict__ w, T* __restrict__ dst, int Cout) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Cout * 7 * 32) return;
    const int slot = i & 31, r = (i >> 5) % 7, co = i / (7 * 32);
    const int s = slot >> 2, c = slot & 3;
    float v = 0.f;
    if (s < 7 && c < 3) v = w[((co * 7 + r) * 7 + s) * 3 + c];
    Elem<T>::st(dst + i, v);
}

__global__ void stem_unpack_wgrad_kernel(const float* __restrict__ dp, float* __restrict__ dw, int Cout) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Cout * 147) return;
    const int c = i % 3, s = (i / 3) % 7, r = (i / 21) % 7, co = i / 147;
    dw[i] += dp[(co * 7 + r) * 32 + s * 4 + c];
}

// NCHW f32 [B,3,H,W] (arbitrary strides) -> [B][H+6][W+8][4] zero bordered (3 top/left)
template <typename T>
__global__ void stem_pack_image_kernel(const float* __restrict__ img, long sB, long sC, long sH, long sW,
                                       T* __restrict__ dst, int B, int H, int W) {
    const int Hp = H + 6, Wp = W + 8;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;      // one padded pixel per thread
    if (i >= (long)B * Hp * Wp) return;
    const int wp = (int)(i % Wp), hp = (int)((i / Wp) % Hp), b = (int)(i / ((long)Wp * Hp));
    const int h = hp - 3, w = wp - 3;
    float v[3] = {0.f, 0.f, 0.f};
    if (h >= 0 && h < H && w >= 0 && w < W) {
        const float* q = img + b * sB + h * sH + w * sW;
        v[0] = q[0]; v[1] = q[sC]; v[2] = q[2 * sC];
    }
    T* d = dst + i * 4;
    Elem<T>::st(d + 0, v[0]); Elem<T>::st(d + 1, v[1]); Elem<T>::st(d + 2, v[2]); Elem<T>::st(d + 3, 0.f);
}

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            long n, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt, float gscale) {
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    const int cnt = (i + 3 < n) ? 4 : (int)(n - i);
    for (int k = 0; k < cnt; ++k) {
        float gr = g[i + k] * gscale;
        const float pv = p[i + k];
        if (wd != 0.f) gr += wd * pv;
        const float mk = b1 * m[i + k] + (1.f - b1) * gr;
        const float vk = b2 * v[i + k] + (1.f - b2) * gr * gr;
        m[i + k] = mk; v[i + k] = vk;
        const float denom = sqrtf(vk) / bc2_sqrt + eps;
        p[i + k] = pv - (lr / bc1) * (mk / denom);
    }
}

// Device-resident optimizer state (so that a captured hipGraph replays correct steps): hyper[0..7] =
// {lr, beta1, beta2, eps, weight_decay, grad_scale, bias_correction1, sqrt(bias_correction2)}, hyper[8] = step (as float
// bits of an int32).  adam_advance increments the step and refreshes the two bias corrections in double precision,
// exactly the values torch.optim.Adam computes on the host.
__global__ void adam_advance_kernel(float* __restrict__ hyper) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int* stepp = reinterpret_cast<int*>(hyper + 8);
    const int t = *stepp + 1;
    *stepp = t;
    const double b1 = (double)hyper[1], b2 = (double)hyper[2];
    hyper[6] = (float)(1.0 - pow(b1, (double)t));
    hyper[7] = (float)sqrt(1.0 - pow(b2, (double)t));
}

__global__ void adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                long n, const float* __restrict__ hyper) {
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], wd = hyper[4], gscale = hyper[5], bc1 = hyper[6], bc2_sqrt = hyper[7];
    const float step_size = lr / bc1;
    if (i + 3 < n) {
        const float4 g4 = *reinterpret_cast<const float4*>(g + i);
        float4 p4 = *reinterpret_cast<float4*>(p + i), m4 = *reinterpret_cast<float4*>(m + i), v4 = *reinterpret_cast<float4*>(v + i);
        float gr[4] = {g4.x, g4.y, g4.z, g4.w}, pv[4] = {p4.x, p4.y, p4.z, p4.w}, mk[4] = {m4.x, m4.y, m4.z, m4.w}, vk[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float gk = gr[k] * gscale;
            if (wd != 0.f) gk += wd * pv[k];
            mk[k] = b1 * mk[k] + (1.f - b1) * gk;
            vk[k] = b2 * vk[k] + (1.f - b2) * gk * gk;
            const float denom = sqrtf(vk[k]) / bc2_sqrt + eps;
            pv[k] = pv[k] - step_size * (mk[k] / denom);
        }
        *reinterpret_cast<float4*>(p + i) = make_float4(pv[0], pv[1], pv[2], pv[3]);
        *reinterpret_cast<float4*>(m + i) = make_float4(mk[0], mk[1], mk[2], mk[3]);
        *reinterpret_cast<float4*>(v + i) = make_float4(vk[0], vk[1], vk[2], vk[3]);
    } else {
        for (long k = i; k < n; ++k) {
            float gk = g[k] * gscale;
            const float pv = p[k];
            if (wd != 0.f) gk += wd * pv;
            const float mk = b1 * m[k] + (1.f - b1) * gk;
            const float vk = b2 * v[k] + (1.f - b2) * gk * gk;
            m[k] = mk; v[k] = vk;
            const float denom = sqrtf(vk) / bc2_sqrt + eps;
            p[k] = pv - step_size * (mk / denom);
        }
    }
}

__global__ void fill_kernel(float* __restrict__ dst, float v, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = v;
}

inline unsigned nblk(long n, int per) { return (unsigned)((n + per - 1) / per); }

}  // namespace

extern "C" int mpn_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream) {
    MPN_CHECK_ARG(src && dst && n > 0);
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(nblk(n, 2048)), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, (long)n);
    return mpn_launch_status();
}

extern "C" int mpn_cast_f32(const float* src, void* dst, int64_t n, int dtype, void* stream) {
    MPN_CHECK_ARG(src && dst && n > 0 && (dtype == MPN_BF16 || dtype == MPN_F16));
    if (dtype == MPN_BF16) hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(nblk(n, 2048)), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, (long)n);
    else hipLaunchKernelGGL(cast_f32_f16_kernel, dim3(nblk(n, 2048)), dim3(256), 0, (hipStream_t)stream, src, (f16_t*)dst, (long)n);
    return mpn_launch_status();
}

extern "C" int mpn_weight_transpose(const float* w, void* wt, int Cout, int RS, int Cin, int Cout_pad, int dtype, void* stream) {
    MPN_CHECK_ARG(w && wt && Cout > 0 && RS > 0 && Cin > 0 && Cout_pad >= Cout);
    dim3 grid((Cin + 31) / 32, (Cout_pad + 31) / 32, RS), block(32, 8);
    MPN_DISPATCH_T(dtype, hipLaunchKernelGGL((weight_transpose_kernel<T>), grid, block, 0, (hipStream_t)stream, w, (T*)wt, Cout, RS, Cin, Cout_pad));
    return mpn_launch_status();
}

extern "C" int mpn_weight_transpose_batched(const float* arena, void* dst, const int64_t* table, int nlayers, int64_t nblocks,
                                            int dtype, void* stream) {
    MPN_CHECK_ARG(arena && dst && table && nlayers > 0 && nblocks > 0 && nblocks < 0x7fffffffLL);
    dim3 grid((unsigned)nblocks), block(32, 8);
    MPN_DISPATCH_T(dtype, hipLaunchKernelGGL((weight_transpose_batched_kernel<T>), grid, block, 0, (hipStream_t)stream, arena, (T*)dst, (const long*)table, nlayers));
    return mpn_launch_status();
}

extern "C" int mpn_weight_pad_k(const float* w, void* dst, int Cout, int K, int Kpad, int dtype, void* stream) {
    MPN_CHECK_ARG(w && dst && Cout > 0 && K > 0 && Kpad >= K);
    const long n = (long)Cout * Kpad;
    MPN_DISPATCH_T(dtype, hipLaunchKernelGGL((weight_pad_k_kernel<T>), dim3(nblk(n, 256)), dim3(256), 0, (hipStream_t)stream, w, (T*)dst, Cout, K, Kpad));
    return mpn_launch_status();
}

extern "C" int mpn_stem_pack_weight(const float* w, void* packed, int Cout, int dtype, void* stream) {
    MPN_CHECK_ARG(w && packed && Cout > 0);
    const long n = (long)Cout * 7 * 32;
    MPN_DISPATCH_T(dtype, hipLaunchKernelGGL((stem_pack_weight_kernel<T>), dim3(nblk(n, 256)), dim3(256), 0, (hipStream_t)stream, w, (T*)packed, Cout));
    return mpn_launch_status();
}

extern "C" int mpn_stem_unpack_wgrad(const float* dpacked, float* dw, int Cout, void* stream) {
    MPN_CHECK_ARG(dpacked && dw && Cout > 0);
    hipLaunchKernelGGL(stem_unpack_wgrad_kernel, dim3(nblk((long)Cout * 147, 256)), dim3(256), 0, (hipStream_t)stream, dpacked, dw, Cout);
    return mpn_launch_status();
}

extern "C" int mpn_stem_pack_image(const float* img, int64_t sB, int64_t sC, int64_t sH, int64_t sW, void* dst,
                                   int B, int H, int W, int dtype, void* stream) {
    MPN_CHECK_ARG(img && dst && B > 0 && H > 0 && W > 0);
    const long n = (long)B * (H + 6) * (W + 8);
    MPN_DISPATCH_T(dtype, hipLaunchKernelGGL((stem_pack_image_kernel<T>), dim3(nblk(n, 256)), dim3(256), 0, (hipStream_t)stream, img, (long)sB, (long)sC, (long)sH, (long)sW, (T*)dst, B, H, W));
    return mpn_launch_status();
}

extern "C" int mpn_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                             float beta1, float beta2, float eps, float weight_decay, float bias_correction1,
                             float bias_correction2_sqrt, float grad_scale, void* stream) {
    MPN_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && n > 0);
    hipLaunchKernelGGL(adam_kernel, dim3(nblk(n, 1024)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq,
                       (long)n, lr, beta1, beta2, eps, weight_decay, bias_correction1, bias_correction2_sqrt, grad_scale);
    return mpn_launch_status();
}

extern "C" int mpn_adam_advance(float* hyper, void* stream) {
    MPN_CHECK_ARG(hyper);
    hipLaunchKernelGGL(adam_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, hyper);
    return mpn_launch_status();
}

extern "C" int mpn_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, const float* hyper,
                                 void* stream) {
    MPN_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && hyper && n > 0);
    MPN_CHECK_ARG(((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) % 16 == 0);
    hipLaunchKernelGGL(adam_dev_kernel, dim3(nblk(n, 1024)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, (long)n, hyper);
    return mpn_launch_status();
}

extern "C" int mpn_copy_bytes(void* dst, const void* src, int64_t nbytes, void* stream) {
    MPN_CHECK_ARG(dst && src && nbytes > 0);
    return (int)hipMemcpyAsync(dst, src, (size_t)nbytes, hipMemcpyDeviceToDevice, (hipStream_t)stream);
}

extern "C" int mpn_fill_f32(float* dst, float v, int64_t n, void* stream) {
    MPN_CHECK_ARG(dst && n > 0);
    hipLaunchKernelGGL(fill_kernel, dim3(nblk(n, 256)), dim3(256), 0, (hipStream_t)stream, dst, v, (long)n);
    return mpn_launch_status();
}

extern "C" const char* mpn_version(void) { return "mpn-hip 0.2 (gfx950) src:" MPN_BUILD_ID; }
