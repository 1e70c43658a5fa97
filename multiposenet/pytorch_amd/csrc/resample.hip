// resample.hip — max-pool, nearest up/down-sampling, concat slices, layout/API-edge conversion.
//
// Reference call sites: F.max_pool2d(3,2,1) network/fpn.py:100; F.upsample(nearest)+add
// fpn.py:84-95; nn.Upsample x8/x4/x2 + Concat network/posenet.py:180-184,296-299,311-315.
// All HBM-bound streaming kernels on dense pixel-major tensors (16-byte vectors along channels).
#include "common.h"

namespace {

// -------------------------------------------------------------------- max-pool 3x3 / s2 / p1
template <typename T>
__global__ void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, uint8_t* __restrict__ idx,
                                   int B, int H, int W, int Cs, int Ho, int Wo) {
    constexpr int V = Vec16<T>::N;
    const int G = Cs / V;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * Ho * Wo * G) return;
    const int g = (int)(i % G);
    long pix = i / G;
    const int wo = (int)(pix % Wo); pix /= Wo;
    const int ho = (int)(pix % Ho);
    const int b = (int)(pix / Ho);
    float best[V]; int bi[V];
#pragma unroll
    for (int k = 0; k < V; ++k) { best[k] = -INFINITY; bi[k] = 0; }
    bool first = true;
    for (int r = 0; r < 3; ++r) {
        const int h = ho * 2 - 1 + r;
        if ((unsigned)h >= (unsigned)H) continue;
        for (int s = 0; s < 3; ++s) {
            const int w = wo * 2 - 1 + s;
            if ((unsigned)w >= (unsigned)W) continue;
            Vec16<T> v; v.load(x + (((long)b * H + h) * W + w) * Cs + g * V);
#pragma unroll
            for (int k = 0; k < V; ++k)
                if (first || v.v[k] > best[k]) { best[k] = v.v[k]; bi[k] = r * 3 + s; }
            first = false;
        }
    }
    Vec16<T> o;
#pragma unroll
    for (int k = 0; k < V; ++k) o.v[k] = best[k];
    o.store(y + i * V);
    if (idx) {
#pragma unroll
        for (int k = 0; k < V; ++k) idx[i * V + k] = (uint8_t)bi[k];
    }
}

template <typename T>
__global__ void maxpool_bwd_kernel(const T* __restrict__ dy, const uint8_t* __restrict__ idx, T* __restrict__ dx,
                                   int B, int H, int W, int Cs, int Ho, int Wo) {
    constexpr int V = Vec16<T>::N;
    const int G = Cs / V;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * H * W * G) return;
    const int g = (int)(i % G);
    long pix = i / G;
    const int w = (int)(pix % W); pix /= W;
    const int h = (int)(pix % H);
    const int b = (int)(pix / H);
    float acc[V];
#pragma unroll
    for (int k = 0; k < V; ++k) acc[k] = 0.f;
    // windows (ho, wo) that contain (h, w): ho*2-1 <= h <= ho*2+1
    const int ho_lo = h / 2, ho_hi = (h + 1) / 2;       // ceil((h-1)/2) == h/2 for h >= 0
    const int wo_lo = w / 2, wo_hi = (w + 1) / 2;
    for (int ho = ho_lo; ho <= ho_hi; ++ho) {
        if (ho >= Ho) continue;
        const int r = h - (ho * 2 - 1);
        for (int wo = wo_lo; wo <= wo_hi; ++wo) {
            if (wo >= Wo) continue;
            const int s = w - (wo * 2 - 1);
            const int tap = r * 3 + s;
            const long o = ((((long)b * Ho + ho) * Wo + wo) * G + g) * V;
            Vec16<T> d; d.load(dy + o);
#pragma unroll
            for (int k = 0; k < V; ++k)
                if (idx[o + k] == tap) acc[k] += d.v[k];
        }
    }
    Vec16<T> out;
#pragma unroll
    for (int k = 0; k < V; ++k) out.v[k] = acc[k];
    out.store(dx + i * V);
}

// ------------------------------------------------- nearest upsample backward (sum over children)
__device__ __forceinline__ int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

template <typename T>
__global__ void upsample_bwd_kernel(const T* __restrict__ dfine, long f_sP, int f_coff, T* __restrict__ dcoarse,
                                    int B, int Hf, int Wf, int Hc, int Wc, int Cs, int C, int accumulate) {
    constexpr int V = Vec16<T>::N;
    const int G = Cs / V;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * Hc * Wc * G) return;
    const int g = (int)(i % G);
    long pix = i / G;
    const int w = (int)(pix % Wc); pix /= Wc;
    const int h = (int)(pix % Hc);
    const int b = (int)(pix / Hc);
    float acc[V];
    Vec16<T> o;
    if (accumulate) o.load(dcoarse + i * V);
#pragma unroll
    for (int k = 0; k < V; ++k) acc[k] = accumulate ? o.v[k] : 0.f;
    // fine rows whose nearest source (fh*Hc)/Hf == h
    const int fh0 = ceil_div((long)h * Hf, Hc), fh1 = ceil_div((long)(h + 1) * Hf, Hc);
    const int fw0 = ceil_div((long)w * Wf, Wc), fw1 = ceil_div((long)(w + 1) * Wf, Wc);
    for (int fh = fh0; fh < fh1 && fh < Hf; ++fh)
        for (int fw = fw0; fw < fw1 && fw < Wf; ++fw) {
            Vec16<T> d; d.load(dfine + (((long)b * Hf + fh) * Wf + fw) * f_sP + f_coff + g * V);
#pragma unroll
            for (int k = 0; k < V; ++k) acc[k] += d.v[k];
        }
#pragma unroll
    for (int k = 0; k < V; ++k) o.v[k] = (g * V + k < C) ? acc[k] : 0.f;
    o.store(dcoarse + i * V);
}

// out[b, oh, ow, c_off + c] = src[b, (oh*Hs)/Ho, (ow*Ws)/Wo, c]
template <typename T>
__global__ void upsample_slice_kernel(const T* __restrict__ src, T* __restrict__ dst, int B, int Hs, int Ws, int Cs_src,
                                      int Ho, int Wo, int Cs_dst, int c_off) {
    constexpr int V = Vec16<T>::N;
    const int G = Cs_src / V;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * Ho * Wo * G) return;
    const int g = (int)(i % G);
    long pix = i / G;
    const int ow = (int)(pix % Wo); pix /= Wo;
    const int oh = (int)(pix % Ho);
    const int b = (int)(pix / Ho);
    const int sh = (int)(((long)oh * Hs) / Ho), sw = (int)(((long)ow * Ws) / Wo);
    Vec16<T> v; v.load(src + (((long)b * Hs + sh) * Ws + sw) * Cs_src + g * V);
    v.store(dst + (((long)b * Ho + oh) * Wo + ow) * Cs_dst + c_off + g * V);
}

// ----------------------------------------------- API edge: padded internal -> exact f32 and back
template <typename T>
__global__ void export_f32_kernel(const T* __restrict__ src, float* __restrict__ dst, int B, int Hs, int Ws, int Cs, int C,
                                  int Ho, int Wo, long dst_sB, long dst_sP) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * Ho * Wo * C) return;
    const int c = (int)(i % C);
    long pix = i / C;
    const int ow = (int)(pix % Wo); pix /= Wo;
    const int oh = (int)(pix % Ho);
    const int b = (int)(pix / Ho);
    const int sh = (int)(((long)oh * Hs) / Ho), sw = (int)(((long)ow * Ws) / Wo);
    dst[b * dst_sB + ((long)oh * Wo + ow) * dst_sP + c] = Elem<T>::ld(src + (((long)b * Hs + sh) * Ws + sw) * Cs + c);
}

template <typename T>
__global__ void import_grad_kernel(const float* __restrict__ ddst, long d_sB, long d_sP, T* __restrict__ dsrc,
                                   int B, int Hs, int Ws, int Cs, int C, int Ho, int Wo) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * Hs * Ws * Cs) return;
    const int c = (int)(i % Cs);
    long pix = i / Cs;
    const int w = (int)(pix % Ws); pix /= Ws;
    const int h = (int)(pix % Hs);
    const int b = (int)(pix / Hs);
    float acc = 0.f;
    if (c < C) {
        const int fh0 = ceil_div((long)h * Ho, Hs), fh1 = ceil_div((long)(h + 1) * Ho, Hs);
        const int fw0 = ceil_div((long)w * Wo, Ws), fw1 = ceil_div((long)(w + 1) * Wo, Ws);
        for (int fh = fh0; fh < fh1 && fh < Ho; ++fh)
            for (int fw = fw0; fw < fw1 && fw < Wo; ++fw)
                acc += ddst[b * d_sB + ((long)fh * Wo + fw) * d_sP + c];
    }
    Elem<T>::st(dsrc + i, acc);
}

// strided f32 [B,C,H,W] -> dense NHWC f32
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, long sB, long sC, long sH, long sW,
                                    float* __restrict__ dst, int B, int C, int H, int W) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * H * W * C) return;
    const int c = (int)(i % C);
    long pix = i / C;
    const int w = (int)(pix % W); pix /= W;
    const int h = (int)(pix % H);
    const int b = (int)(pix / H);
    dst[i] = src[b * sB + c * sC + h * sH + w * sW];
}

// dense f32 [rows, C] (e.g. [B*A, 4]) <-> internal padded [rows', Cs]: det head pack / unpack
//   pack:   dst[b*dst_sB + (pix*C + c)] = src[(b*HW + pix)*Cs + c]           (f32 -> f32)
//   unpack: dsrc[(b*HW + pix)*Cs + c] = ddst[b*dst_sB + pix*C + c], pad lanes 0  (f32 -> T)
template <typename T>
__global__ void det_pack_kernel(const T* __restrict__ src, float* __restrict__ dst, int B, long HW, int Cs, int C, long dst_sB) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * HW * C) return;
    const int c = (int)(i % C);
    const long pix = (i / C) % HW;
    const int b = (int)(i / (C * HW));
    dst[b * dst_sB + pix * C + c] = Elem<T>::ld(src + ((long)b * HW + pix) * Cs + c);
}
template <typename T>
__global__ void det_unpack_kernel(const float* __restrict__ ddst, T* __restrict__ dsrc, int B, long HW, int Cs, int C, long dst_sB) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * HW * Cs) return;
    const int c = (int)(i % Cs);
    const long pix = (i / Cs) % HW;
    const int b = (int)(i / (Cs * HW));
    Elem<T>::st(dsrc + i, c < C ? ddst[b * dst_sB + pix * C + c] : 0.f);
}

template <typename T>
__global__ void relu_bwd_kernel(const T* __restrict__ dz, const T* __restrict__ z, T* __restrict__ dx, long nvec, int accumulate) {
    constexpr int V = Vec16<T>::N;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nvec) return;
    Vec16<T> d, o, a;
    d.load(dz + i * V); o.load(z + i * V);
    if (accumulate) a.load(dx + i * V);
#pragma unroll
    for (int k = 0; k < V; ++k) a.v[k] = (accumulate ? a.v[k] : 0.f) + ((o.v[k] > 0.f) ? d.v[k] : 0.f);
    a.store(dx + i * V);
}

template <typename T>
__global__ void relu_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, long nvec) {
    constexpr int V = Vec16<T>::N;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nvec) return;
    Vec16<T> a; a.load(x + i * V);
#pragma unroll
    for (int k = 0; k < V; ++k) a.v[k] = fmaxf(a.v[k], 0.f);
    a.store(y + i * V);
}

template <typename T>
__global__ void add_inplace_kernel(T* __restrict__ dst, const T* __restrict__ src, long nvec) {
    constexpr int V = Vec16<T>::N;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nvec) return;
    Vec16<T> a, b;
    a.load(dst + i * V); b.load(src + i * V);
#pragma unroll
    for (int k = 0; k < V; ++k) a.v[k] += b.v[k];
    a.store(dst + i * V);
}

// per-chunk column sums of dy[P][Cs] (bias gradients); partial[chunk][C].  Same thread mapping as the
// BN kernels: GB channel groups (16 B) x 256/GB pixel lanes, 16-byte coalesced loads.
template <typename T>
__global__ void __launch_bounds__(256) channel_sum_kernel(const T* __restrict__ dy, long P, int C, int Cs, float* __restrict__ partial, int chunk_pix) {
    constexpr int V = Vec16<T>::N;
    __shared__ float sh[256][V + 1];
    const int G = Cs / V;
    const int GB = G < 256 ? G : 256;
    const int lanes = 256 / GB;
    const int g = blockIdx.y * GB + (threadIdx.x % GB);
    const int pl = threadIdx.x / GB;
    const int c0 = g * V;
    const long p_begin = (long)blockIdx.x * chunk_pix;
    long p_end = p_begin + chunk_pix; if (p_end > P) p_end = P;
    float s[V];
#pragma unroll
    for (int k = 0; k < V; ++k) s[k] = 0.f;
    for (long p = p_begin + pl; p < p_end; p += lanes) {
        Vec16<T> d; d.load(dy + p * Cs + c0);
#pragma unroll
        for (int k = 0; k < V; ++k) s[k] += d.v[k];
    }
#pragma unroll
    for (int k = 0; k < V; ++k) sh[threadIdx.x][k] = s[k];
    __syncthreads();
    if (pl == 0) {
        for (int l = 1; l < lanes; ++l)
#pragma unroll
            for (int k = 0; k < V; ++k) s[k] += sh[threadIdx.x + l * GB][k];
#pragma unroll
        for (int k = 0; k < V; ++k)
            if (c0 + k < C) partial[(long)blockIdx.x * C + c0 + k] = s[k];
    }
}

// few rows, any channel count (bias gradient of the PRN's Linear layers): one thread per channel
template <typename T>
__global__ void colsum_rows_kernel(const T* __restrict__ dy, long P, int C, int Cs, float* __restrict__ db) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float s = 0.f;
    for (long p = 0; p < P; ++p) s += Elem<T>::ld(dy + p * Cs + c);
    db[c] += s;
}

inline int cs_lanes(int Cs, int V) { const int G = Cs / V; return 256 / (G < 256 ? G : 256); }
inline int cs_chunk(long P, int lanes) {
    long c = P / 512;
    const long lo = (long)lanes * 4, hi = 4096;
    if (c < lo) c = lo;
    if (c > hi) c = hi;
    return (int)((c + lanes - 1) / lanes * lanes);
}

inline unsigned nb(long n) { return (unsigned)((n + 255) / 256); }

}  // namespace

extern "C" int mpn_maxpool3x3s2_forward(const void* x, void* y, uint8_t* idx, int B, int H, int W, int Cs, int Ho, int Wo,
                                        int dtype, void* stream) {
    MPN_CHECK_ARG(x && y && B > 0 && Cs % 8 == 0 && Ho == (H + 2 - 3) / 2 + 1 && Wo == (W + 2 - 3) / 2 + 1);
    const int V = dtype == MPN_F32 ? 4 : 8;
    const long n = (long)B * Ho * Wo * (Cs / V);
    MPN_DISPATCH_T(dtype, hipLaunchKernelGGL((maxpool_fwd_kernel<T>), dim3(nb(n)), dim3(256), 0, (hipStream_t)stream, (const T*)x, (T*)y, idx, B, H, W, Cs, Ho, Wo));
    return mpn_launch_status();
}

extern "C" int mpn_maxpool3x3s2_backward(const void* dy, const uint8_t* idx, void* dx, int B, int H, int W, int Cs, int Ho, int Wo,
                                         int dtype, void* stream) {
    MPN_CHECK_ARG(dy && idx && dx && B > 0 && Cs % 8 == 0);
    const int V = dtype == MPN_F32 ? 4 : 8;
    const long n = (long)B * H * W * (Cs / V);
    MPN_DISPATCH_T(dtype, hipLaunchKernelGGL((maxpool_bwd_kernel<T>), dim3(nb(n)), dim3(256), 0, (hipStream_t)stream, (const T*)dy, idx, (T*)dx, B, H, W, Cs, Ho, Wo));
    return mpn_launch_status();
}

extern "C" int mpn_upsample_nearest_backward(const void* dfine, void* dcoarse, int B, int Hf, int Wf, int Hc, int Wc, int Cs,
                                             int accumulate, int dtype, void* stream) {
    MPN_CHECK_ARG(dfine && dcoarse && B > 0 && Cs % 8 == 0);
    const int V = dtype == MPN_F32 ? 4 : 8;
    const long n = (long)B * Hc * Wc * (Cs / V);
    MPN_DISPATCH_T(dtype, hipLaunchKernelGGL((upsample_bwd_kernel<T>), dim3(nb(n)), dim3(256), 0, (hipStream_t)stream, (const T*)dfine, (long)Cs, 0, (T*)dcoarse, B, Hf, Wf, Hc, Wc, Cs, Cs, accumulate));
    return mpn_launch_status();
}

extern "C" int mpn_upsample_nearest_slice(const void* src, void* dst, int B, int Hs, int Ws, int Cs_src, int Ho, int Wo,
                                          int Cs_dst, int c_off, int dtype, void* stream) {
    MPN_CHECK_ARG(src && dst && B > 0 && Cs_src % 8 == 0 && Cs_dst % 8 == 0 && c_off % 8 == 0);
    const int V = dtype == MPN_F32 ? 4 : 8;
    const long n = (long)B * Ho * Wo * (Cs_src / V);
    MPN_DISPATCH_T(dtype, hipLaunchKernelGGL((upsample_slice_kernel<T>), dim3(nb(n)), dim3(256), 0, (hipStream_t)stream, (const T*)src, (T*)dst, B, Hs, Ws, Cs_src, Ho, Wo, Cs_dst, c_off));
    return mpn_launch_status();
}

extern "C" int mpn_upsample_nearest_slice_backward(const void* ddst, void* dsrc, int B, int Hs, int Ws, int Cs_src, int Ho, int Wo,
                                                   int Cs_dst, int c_off, int dtype, void* stream) {
    MPN_CHECK_ARG(ddst && dsrc && B > 0 && Cs_src % 8 == 0 && Cs_dst % 8 == 0 && c_off % 8 == 0);
    const int V = dtype == MPN_F32 ? 4 : 8;
    const long n = (long)B * Hs * Ws * (Cs_src / V);
    MPN_DISPATCH_T(dtype, hipLaunchKernelGGL((upsample_bwd_kernel<T>), dim3(nb(n)), dim3(256), 0, (hipStream_t)stream, (const T*)ddst, (long)Cs_dst, c_off, (T*)dsrc, B, Ho, Wo, Hs, Ws, Cs_src, Cs_src, 0));
    return mpn_launch_status();
}

extern "C" int mpn_export_f32(const void* src, int src_dtype, float* dst, int B, int Hs, int Ws, int Cs, int C, int Ho, int Wo,
                              int64_t dst_sB, int64_t dst_sP, void* stream) {
    MPN_CHECK_ARG(src && dst && B > 0 && C > 0 && Cs >= C);
    const long n = (long)B * Ho * Wo * C;
    MPN_DISPATCH_T(src_dtype, hipLaunchKernelGGL((export_f32_kernel<T>), dim3(nb(n)), dim3(256), 0, (hipStream_t)stream, (const T*)src, dst, B, Hs, Ws, Cs, C, Ho, Wo, (long)dst_sB, (long)dst_sP));
    return mpn_launch_status();
}

extern "C" int mpn_import_grad(const float* ddst, int64_t ddst_sB, int64_t ddst_sP, void* dsrc, int dst_dtype, int B, int Hs, int Ws,
                               int Cs, int C, int Ho, int Wo, void* stream) {
    MPN_CHECK_ARG(ddst && dsrc && B > 0 && C > 0 && Cs >= C);
    const long n = (long)B * Hs * Ws * Cs;
    MPN_DISPATCH_T(dst_dtype, hipLaunchKernelGGL((import_grad_kernel<T>), dim3(nb(n)), dim3(256), 0, (hipStream_t)stream, ddst, (long)ddst_sB, (long)ddst_sP, (T*)dsrc, B, Hs, Ws, Cs, C, Ho, Wo));
    return mpn_launch_status();
}

extern "C" int mpn_nchw_to_nhwc_f32(const float* src, int64_t sB, int64_t sC, int64_t sH, int64_t sW, float* dst,
                                    int B, int C, int H, int W, void* stream) {
    MPN_CHECK_ARG(src && dst && B > 0 && C > 0);
    const long n = (long)B * C * H * W;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(nb(n)), dim3(256), 0, (hipStream_t)stream, src, (long)sB, (long)sC, (long)sH, (long)sW, dst, B, C, H, W);
    return mpn_launch_status();
}

extern "C" int mpn_det_pack(const void* src, int src_dtype, float* dst, int B, int64_t HW, int Cs, int C, int64_t dst_sB, void* stream) {
    MPN_CHECK_ARG(src && dst && B > 0 && C > 0 && Cs >= C);
    const long n = (long)B * HW * C;
    MPN_DISPATCH_T(src_dtype, hipLaunchKernelGGL((det_pack_kernel<T>), dim3(nb(n)), dim3(256), 0, (hipStream_t)stream, (const T*)src, dst, B, (long)HW, Cs, C, (long)dst_sB));
    return mpn_launch_status();
}

extern "C" int mpn_det_unpack(const float* ddst, void* dsrc, int dst_dtype, int B, int64_t HW, int Cs, int C, int64_t dst_sB, void* stream) {
    MPN_CHECK_ARG(ddst && dsrc && B > 0 && C > 0 && Cs >= C);
    const long n = (long)B * HW * Cs;
    MPN_DISPATCH_T(dst_dtype, hipLaunchKernelGGL((det_unpack_kernel<T>), dim3(nb(n)), dim3(256), 0, (hipStream_t)stream, ddst, (T*)dsrc, B, (long)HW, Cs, C, (long)dst_sB));
    return mpn_launch_status();
}

extern "C" int mpn_relu_backward(const void* dz, const void* z, void* dx, int64_t n, int accumulate, int dtype, void* stream) {
    MPN_CHECK_ARG(dz && z && dx && n > 0 && n % 8 == 0);
    const int V = dtype == MPN_F32 ? 4 : 8;
    const long nvec = n / V;
    MPN_DISPATCH_T(dtype, hipLaunchKernelGGL((relu_bwd_kernel<T>), dim3(nb(nvec)), dim3(256), 0, (hipStream_t)stream, (const T*)dz, (const T*)z, (T*)dx, nvec, accumulate));
    return mpn_launch_status();
}

extern "C" int mpn_relu_forward(const void* x, void* y, int64_t n, int dtype, void* stream) {
    MPN_CHECK_ARG(x && y && n > 0 && n % 8 == 0);
    const int V = dtype == MPN_F32 ? 4 : 8;
    const long nvec = n / V;
    MPN_DISPATCH_T(dtype, hipLaunchKernelGGL((relu_fwd_kernel<T>), dim3(nb(nvec)), dim3(256), 0, (hipStream_t)stream, (const T*)x, (T*)y, nvec));
    return mpn_launch_status();
}

extern "C" int mpn_add_inplace(void* dst, const void* src, int64_t n, int dtype, void* stream) {
    MPN_CHECK_ARG(dst && src && n > 0 && n % 8 == 0);
    const int V = dtype == MPN_F32 ? 4 : 8;
    const long nvec = n / V;
    MPN_DISPATCH_T(dtype, hipLaunchKernelGGL((add_inplace_kernel<T>), dim3(nb(nvec)), dim3(256), 0, (hipStream_t)stream, (T*)dst, (const T*)src, nvec));
    return mpn_launch_status();
}

extern "C" int mpn_colsum_rows(const void* dy, int dy_dtype, int64_t P, int C, int Cs, float* db, void* stream) {
    MPN_CHECK_ARG(dy && db && P > 0 && C > 0 && Cs >= C);
    MPN_DISPATCH_T(dy_dtype, hipLaunchKernelGGL((colsum_rows_kernel<T>), dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const T*)dy, (long)P, C, Cs, db));
    return mpn_launch_status();
}

extern "C" int mpn_channel_sum_chunks(int64_t P, int Cs, int dtype) {
    const int V = dtype == MPN_F32 ? 4 : 8;
    const int G = Cs / V;
    if (P <= 0 || Cs % V != 0 || G <= 0 || (G & (G - 1)) != 0) return MPN_E_BADARG;
    const int chunk = cs_chunk(P, cs_lanes(Cs, V));
    return (int)((P + chunk - 1) / chunk);
}

extern "C" int mpn_channel_sum(const void* dy, int dy_dtype, int64_t P, int C, int Cs, float* partial, int chunks, void* stream) {
    MPN_CHECK_ARG(dy && partial && P > 0 && C > 0 && Cs >= C);
    const int V = dy_dtype == MPN_F32 ? 4 : 8;
    const int G = Cs / V;
    MPN_CHECK_ARG(Cs % V == 0 && G > 0 && (G & (G - 1)) == 0);
    const int chunk = cs_chunk(P, cs_lanes(Cs, V));
    MPN_CHECK_ARG(chunks == (int)((P + chunk - 1) / chunk));
    dim3 grid((unsigned)chunks, (unsigned)(G <= 256 ? 1 : G / 256));
    MPN_DISPATCH_T(dy_dtype, hipLaunchKernelGGL((channel_sum_kernel<T>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)dy, (long)P, C, Cs, partial, chunk));
    return mpn_launch_status();
}
