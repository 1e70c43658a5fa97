// losses.hip — heat-map MSE, focal + smooth-L1 detection loss, PRN softmax / BCE for gfx950.
//
// Reference: build_keypoint_loss network/posenet.py:367-403 (5 x MSELoss(size_average=True) over
// pred[:, :18]*w vs w*gt, plus max/min of the final heat-map), FocalLoss network/losses.py:27-137
// (alpha .25, gamma 2, IoU<.4 negative / >=.5 positive, smooth-L1 beta 1/9, per-image python loop
// replaced by one thread per (image, anchor)), calc_iou losses.py:5-22, PRN softmax + BCELoss
// posenet.py:345-347,427-445.  All reductions are two-stage and deterministic (no atomics).
#include "common.h"

namespace {

constexpr int MSE_CHUNK = 256 * 16;     // elements (pixel*18 + c) per block

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
    return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fminf(v, __shfl_xor(v, m, 64));
    return v;
}

struct PredPtrs { const float* p[5]; long sP[5]; };
struct DPredPtrs { float* p[5]; long sP[5]; int C[5]; };

// partial[block][8] = {sum_j (5), unused, max, min}
__global__ void mse_fwd_kernel(PredPtrs pr, const float* __restrict__ gt, const float* __restrict__ wgt, long nelem,
                               float* __restrict__ partial) {
    __shared__ float sh[4][8];
    float s[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    float mx = -INFINITY, mn = INFINITY;
    const long base = (long)blockIdx.x * MSE_CHUNK;
    for (int it = 0; it < 16; ++it) {
        const long e = base + it * 256 + threadIdx.x;
        if (e < nelem) {
            const long pix = e / 18; const int c = (int)(e - pix * 18);
            const float w = wgt[e], g = gt[e];
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const float pv = pr.p[j][pix * pr.sP[j] + c];
                const float d = pv * w - w * g;
                s[j] += d * d;
                if (j == 4) { mx = fmaxf(mx, pv); mn = fminf(mn, pv); }
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < 5; ++j) s[j] = wave_sum(s[j]);
    mx = wave_max(mx); mn = wave_min(mn);
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < 5; ++j) sh[wave][j] = s[j];
        sh[wave][6] = mx; sh[wave][7] = mn;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float* o = partial + (long)blockIdx.x * 8;
#pragma unroll
        for (int j = 0; j < 5; ++j) o[j] = sh[0][j] + sh[1][j] + sh[2][j] + sh[3][j];
        o[5] = 0.f;
        o[6] = fmaxf(fmaxf(sh[0][6], sh[1][6]), fmaxf(sh[2][6], sh[3][6]));
        o[7] = fminf(fminf(sh[0][7], sh[1][7]), fminf(sh[2][7], sh[3][7]));
    }
}

__global__ void mse_finalize_kernel(const float* __restrict__ partial, int chunks, double nelem, float* __restrict__ out) {
    __shared__ double sh[256][5];
    __shared__ float shm[256][2];
    double s[5] = {0, 0, 0, 0, 0};
    float mx = -INFINITY, mn = INFINITY;
    for (int i = threadIdx.x; i < chunks; i += 256) {
        const float* o = partial + (long)i * 8;
        for (int j = 0; j < 5; ++j) s[j] += (double)o[j];
        mx = fmaxf(mx, o[6]); mn = fminf(mn, o[7]);
    }
    for (int j = 0; j < 5; ++j) sh[threadIdx.x][j] = s[j];
    shm[threadIdx.x][0] = mx; shm[threadIdx.x][1] = mn;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot[5] = {0, 0, 0, 0, 0};
        for (int i = 0; i < 256; ++i) {
            for (int j = 0; j < 5; ++j) tot[j] += sh[i][j];
            mx = fmaxf(mx, shm[i][0]); mn = fminf(mn, shm[i][1]);
        }
        float total = 0.f;
        for (int j = 0; j < 5; ++j) { const float l = (float)(tot[j] / nelem); out[j] = l; total += l; }
        out[5] = total; out[6] = mx; out[7] = mn;
    }
}

// d(total)/d(pred_j) at one full-resolution element.  Contraction is off so that mse_bwd_kernel (API path: full-resolution f32
// gradients, summed onto the coarser levels by import_grad) and mse_train_kernel (recorded step: the same sums inside one launch)
// round identically; the recorded step's parameters then stay bit-identical to the eager step's.
__device__ __forceinline__ float mse_grad(float k, float w, float g, float pv) {
#pragma clang fp contract(off)
    const float a = pv * w;
    const float b = w * g;
    const float d = a - b;
    const float kw = k * w;
    return kw * d;
}

// one thread per (pixel, channel < Cmax); dpred_j[pix*sP + c] = gs*2*w*(p*w - w*g)/N, 0 for c >= 18
__global__ void mse_bwd_kernel(PredPtrs pr, DPredPtrs dp, const float* __restrict__ gt, const float* __restrict__ wgt,
                               long npix, int Cmax, const float* __restrict__ gscale) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix * Cmax) return;
    const long pix = i / Cmax; const int c = (int)(i - pix * Cmax);
    const float gs = gscale ? gscale[0] : 1.f;
    const float k = gs * 2.0f / (float)((double)npix * 18.0);
    float w = 0.f, g = 0.f;
    if (c < 18) { w = wgt[pix * 18 + c]; g = gt[pix * 18 + c]; }
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        if (c >= dp.C[j] || dp.p[j] == nullptr) continue;
        float v = 0.f;
        if (c < 18) v = mse_grad(k, w, g, pr.p[j][pix * pr.sP[j] + c]);
        dp.p[j][pix * dp.sP[j] + c] = v;
    }
}

// ---- recorded training step: loss AND gradients in one pass over the INTERNAL tensors (no exported f32 copies) --------------
// Levels 0..3 are the intermediate maps k2..k5 at 1, 1/2, 1/4, 1/8 of the heat-map resolution (posenet.py:243-257 up-samples them
// with nearest neighbours, so a full-resolution pixel (y, x) reads level s at (y >> s, x >> s) and the gradient of a coarse cell is
// the sum over its 4^s children); level 4 is the final prediction.  One thread owns (8x8 pixel cell, channel): it walks its 64
// pixels in row-major order, which is the order import_grad_kernel sums children in, so the coarse gradients round identically.
// Block = 8 cells x 32 channel lanes; lanes >= 18 write the zero padding of the gradient tensors.
struct TrainPtrs { const float* p[5]; void* d[5]; int Cs[5]; };

template <typename T>
__global__ void __launch_bounds__(256) mse_train_kernel(TrainPtrs tp, const float* __restrict__ gt, const float* __restrict__ wgt,
                                                        long g_sB, long g_sC, long g_sH, int B, int H, int W, long ncell,
                                                        const float* __restrict__ gscale, float* __restrict__ partial) {
    __shared__ float sh[4][8];
    const int c = threadIdx.x & 31;
    const long cell = (long)blockIdx.x * 8 + (threadIdx.x >> 5);
    float s[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    float mx = -INFINITY, mn = INFINITY;
    if (cell < ncell) {
        const int cw = W >> 3, ch = H >> 3;
        const int cx = (int)(cell % cw);
        const int cy = (int)((cell / cw) % ch);
        const int b = (int)(cell / ((long)cw * ch));
        const int y0 = cy * 8, x0 = cx * 8;
        const float gs = gscale ? gscale[0] : 1.f;
        const float k = gs * 2.0f / (float)((double)B * H * W * 18.0);
        const bool live = c < 18;
        T* d0 = (T*)tp.d[0]; T* d1 = (T*)tp.d[1]; T* d2 = (T*)tp.d[2]; T* d3 = (T*)tp.d[3]; T* d4 = (T*)tp.d[4];
        float pv1[4] = {0.f, 0.f, 0.f, 0.f}, pv2[2] = {0.f, 0.f}, pv3 = 0.f;
        float a1[4], a2[2], a3 = 0.f;
        if (live) pv3 = tp.p[3][(((long)b * ch + cy) * cw + cx) * tp.Cs[3] + c];
        for (int r = 0; r < 8; ++r) {
            const int y = y0 + r;
            float g[8], w[8];
            if (live) {
                const float* gp = gt + b * g_sB + c * g_sC + y * g_sH + x0;
                const float* wp = wgt + b * g_sB + c * g_sC + y * g_sH + x0;
                const float4 ga = *(const float4*)gp, gb = *(const float4*)(gp + 4);
                const float4 wa = *(const float4*)wp, wb = *(const float4*)(wp + 4);
                g[0] = ga.x; g[1] = ga.y; g[2] = ga.z; g[3] = ga.w; g[4] = gb.x; g[5] = gb.y; g[6] = gb.z; g[7] = gb.w;
                w[0] = wa.x; w[1] = wa.y; w[2] = wa.z; w[3] = wa.w; w[4] = wb.x; w[5] = wb.y; w[6] = wb.z; w[7] = wb.w;
                if ((r & 1) == 0) {
                    const float* q = tp.p[1] + (((long)b * (H >> 1) + (y >> 1)) * (W >> 1) + (x0 >> 1)) * tp.Cs[1] + c;
#pragma unroll
                    for (int i = 0; i < 4; ++i) { pv1[i] = q[(long)i * tp.Cs[1]]; a1[i] = 0.f; }
                }
                if ((r & 3) == 0) {
                    const float* q = tp.p[2] + (((long)b * (H >> 2) + (y >> 2)) * (W >> 2) + (x0 >> 2)) * tp.Cs[2] + c;
#pragma unroll
                    for (int i = 0; i < 2; ++i) { pv2[i] = q[(long)i * tp.Cs[2]]; a2[i] = 0.f; }
                }
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) { g[i] = 0.f; w[i] = 0.f; }
#pragma unroll
                for (int i = 0; i < 4; ++i) a1[i] = 0.f;
                a2[0] = a2[1] = 0.f;
            }
            const long row = ((long)b * H + y) * W + x0;
#pragma unroll
            for (int px = 0; px < 8; ++px) {
                float v0 = 0.f, v4 = 0.f;
                if (live) {
                    const float p0 = tp.p[0][(row + px) * tp.Cs[0] + c];
                    const float p4 = tp.p[4][(row + px) * tp.Cs[4] + c];
                    const float wv = w[px], gv = g[px];
                    const float wg = wv * gv;
                    float e;
                    e = p0 * wv - wg;      s[0] += e * e;
                    e = pv1[px >> 1] * wv - wg; s[1] += e * e;
                    e = pv2[px >> 2] * wv - wg; s[2] += e * e;
                    e = pv3 * wv - wg;     s[3] += e * e;
                    e = p4 * wv - wg;      s[4] += e * e;
                    mx = fmaxf(mx, p4); mn = fminf(mn, p4);
                    v0 = 0.f + mse_grad(k, wv, gv, p0);
                    v4 = 0.f + mse_grad(k, wv, gv, p4);
                    a1[px >> 1] += mse_grad(k, wv, gv, pv1[px >> 1]);
                    a2[px >> 2] += mse_grad(k, wv, gv, pv2[px >> 2]);
                    a3 += mse_grad(k, wv, gv, pv3);
                }
                Elem<T>::st(d0 + (row + px) * tp.Cs[0] + c, v0);
                Elem<T>::st(d4 + (row + px) * tp.Cs[4] + c, v4);
            }
            if (r & 1) {
                T* q = d1 + (((long)b * (H >> 1) + (y >> 1)) * (W >> 1) + (x0 >> 1)) * tp.Cs[1] + c;
#pragma unroll
                for (int i = 0; i < 4; ++i) Elem<T>::st(q + (long)i * tp.Cs[1], a1[i]);
            }
            if ((r & 3) == 3) {
                T* q = d2 + (((long)b * (H >> 2) + (y >> 2)) * (W >> 2) + (x0 >> 2)) * tp.Cs[2] + c;
#pragma unroll
                for (int i = 0; i < 2; ++i) Elem<T>::st(q + (long)i * tp.Cs[2], a2[i]);
            }
        }
        Elem<T>::st(d3 + (((long)b * ch + cy) * cw + cx) * tp.Cs[3] + c, a3);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < 5; ++j) s[j] = wave_sum(s[j]);
    mx = wave_max(mx); mn = wave_min(mn);
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < 5; ++j) sh[wave][j] = s[j];
        sh[wave][6] = mx; sh[wave][7] = mn;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float* o = partial + (long)blockIdx.x * 8;
#pragma unroll
        for (int j = 0; j < 5; ++j) o[j] = sh[0][j] + sh[1][j] + sh[2][j] + sh[3][j];
        o[5] = 0.f;
        o[6] = fmaxf(fmaxf(sh[0][6], sh[1][6]), fmaxf(sh[2][6], sh[3][6]));
        o[7] = fminf(fminf(sh[0][7], sh[1][7]), fminf(sh[2][7], sh[3][7]));
    }
}

// ------------------------------------------------------------------------------- focal loss
struct Assign { float iou_max; int arg; };

__device__ __forceinline__ Assign assign_anchor(const float4 a, const float* __restrict__ anno, int maxN) {
    // losses.py:5-22 (no +1; union clamped at 1e-8), argmax = first maximum (losses.py:59)
    Assign r; r.iou_max = -1.f; r.arg = -1;
    const float area_a = (a.z - a.x) * (a.w - a.y);
    for (int n = 0; n < maxN; ++n) {
        const float* g = anno + n * 5;
        if (g[4] == -1.f) continue;
        const float area_b = (g[2] - g[0]) * (g[3] - g[1]);
        float iw = fminf(a.z, g[2]) - fmaxf(a.x, g[0]);
        float ih = fminf(a.w, g[3]) - fmaxf(a.y, g[1]);
        iw = fmaxf(iw, 0.f); ih = fmaxf(ih, 0.f);
        float ua = area_a + area_b - iw * ih;
        ua = fmaxf(ua, 1e-8f);
        const float iou = (iw * ih) / ua;
        if (iou > r.iou_max) { r.iou_max = iou; r.arg = n; }
    }
    return r;
}

__device__ __forceinline__ void reg_targets(const float4 a, const float* g, float t[4]) {
    // losses.py:97-121
    const float aw = a.z - a.x, ah = a.w - a.y;
    const float acx = a.x + 0.5f * aw, acy = a.y + 0.5f * ah;
    float gw = g[2] - g[0], gh = g[3] - g[1];
    const float gcx = g[0] + 0.5f * gw, gcy = g[1] + 0.5f * gh;
    gw = fmaxf(gw, 1.f); gh = fmaxf(gh, 1.f);
    t[0] = ((gcx - acx) / aw) / 0.1f;
    t[1] = ((gcy - acy) / ah) / 0.1f;
    t[2] = logf(gw / aw) / 0.2f;
    t[3] = logf(gh / ah) / 0.2f;
}

// grid (blocksA, B); partial[(b*blocksA + blk)*4] = {cls_sum, reg_sum, npos, nvalid_anno}
__global__ void focal_fwd_kernel(const float* __restrict__ cls, const float* __restrict__ reg, const float* __restrict__ anchors,
                                 const float* __restrict__ anno, int A, int maxN, float* __restrict__ partial) {
    __shared__ float sh[4][3];
    const int b = blockIdx.y;
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    const float* an = anno + (long)b * maxN * 5;
    float cl = 0.f, rl = 0.f, np = 0.f;
    int nvalid = 0;
    for (int n = 0; n < maxN; ++n) nvalid += (an[n * 5 + 4] != -1.f);
    if (a < A && nvalid > 0) {
        const float4 box = *reinterpret_cast<const float4*>(anchors + (long)a * 4);
        const Assign as = assign_anchor(box, an, maxN);
        float p = cls[(long)b * A + a];
        p = fminf(fmaxf(p, 1e-4f), 1.0f - 1e-4f);
        if (as.iou_max >= 0.5f) {
            const float om = 1.f - p;
            cl = 0.25f * om * om * (-logf(p));
            np = 1.f;
            float t[4]; reg_targets(box, an + as.arg * 5, t);
            const float4 r = *reinterpret_cast<const float4*>(reg + ((long)b * A + a) * 4);
            const float rr[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float d = fabsf(t[k] - rr[k]);
                rl += (d <= 1.0f / 9.0f) ? 0.5f * 9.0f * d * d : d - 0.5f / 9.0f;
            }
        } else if (as.iou_max < 0.4f) {
            cl = 0.75f * p * p * (-logf(1.f - p));
        }
    }
    cl = wave_sum(cl); rl = wave_sum(rl); np = wave_sum(np);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { sh[wave][0] = cl; sh[wave][1] = rl; sh[wave][2] = np; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float* o = partial + ((long)b * gridDim.x + blockIdx.x) * 4;
        o[0] = sh[0][0] + sh[1][0] + sh[2][0] + sh[3][0];
        o[1] = sh[0][1] + sh[1][1] + sh[2][1] + sh[3][1];
        o[2] = sh[0][2] + sh[1][2] + sh[2][2] + sh[3][2];
        o[3] = (float)nvalid;
    }
}

// one block; per_img[b] = {cls_sum, reg_sum, npos, nvalid}; out = {cls_loss, reg_loss}
__global__ void focal_finalize_kernel(const float* __restrict__ partial, int B, int blocksA, float* __restrict__ per_img,
                                      float* __restrict__ out) {
    __shared__ double shc[256], shr[256];
    double csum = 0.0, rsum = 0.0;
    for (int b = threadIdx.x; b < B; b += 256) {
        double c = 0.0, r = 0.0, n = 0.0; float nv = 0.f;
        for (int k = 0; k < blocksA; ++k) {
            const float* o = partial + ((long)b * blocksA + k) * 4;
            c += o[0]; r += o[1]; n += o[2]; nv = o[3];
        }
        per_img[b * 4 + 0] = (float)c; per_img[b * 4 + 1] = (float)r; per_img[b * 4 + 2] = (float)n; per_img[b * 4 + 3] = nv;
        if (nv > 0.f) {
            csum += (double)((float)c / fmaxf((float)n, 1.f));
            if (n > 0.0) rsum += (double)((float)r / (4.f * (float)n));
        }
    }
    shc[threadIdx.x] = csum; shr[threadIdx.x] = rsum;
    __syncthreads();
    if (threadIdx.x == 0) {
        double c = 0.0, r = 0.0;
        for (int i = 0; i < 256; ++i) { c += shc[i]; r += shr[i]; }
        out[0] = (float)(c / B); out[1] = (float)(r / B);
    }
}

__global__ void focal_bwd_kernel(const float* __restrict__ cls, const float* __restrict__ reg, const float* __restrict__ anchors,
                                 const float* __restrict__ anno, int B, int A, int maxN, const float* __restrict__ per_img,
                                 const float* __restrict__ gscale, float* __restrict__ dcls, float* __restrict__ dreg) {
    const int b = blockIdx.y;
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= A) return;
    const float gs = gscale ? gscale[0] : 1.f;        // upstream grad of the classification loss
    const float gsr = gscale ? gscale[1] : 1.f;       // upstream grad of the regression loss
    const float* an = anno + (long)b * maxN * 5;
    const float npos = per_img[b * 4 + 2], nvalid = per_img[b * 4 + 3];
    float dc = 0.f;
    float dr[4] = {0.f, 0.f, 0.f, 0.f};
    if (nvalid > 0.f) {
        const float4 box = *reinterpret_cast<const float4*>(anchors + (long)a * 4);
        const Assign as = assign_anchor(box, an, maxN);
        const float praw = cls[(long)b * A + a];
        const bool inrange = (praw >= 1e-4f) && (praw <= 1.0f - 1e-4f);     // clamp passes gradient inside [min, max]
        const float p = fminf(fmaxf(praw, 1e-4f), 1.0f - 1e-4f);
        const float kc = gs / ((float)B * fmaxf(npos, 1.f));
        if (as.iou_max >= 0.5f) {
            const float om = 1.f - p;
            if (inrange) dc = kc * 0.25f * (2.f * om * logf(p) - om * om / p);
            float t[4]; reg_targets(box, an + as.arg * 5, t);
            const float4 r = *reinterpret_cast<const float4*>(reg + ((long)b * A + a) * 4);
            const float rr[4] = {r.x, r.y, r.z, r.w};
            const float kr = gsr / ((float)B * 4.f * npos);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float e = rr[k] - t[k];
                const float d = fabsf(e);
                dr[k] = kr * ((d <= 1.0f / 9.0f) ? 9.0f * e : (e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f)));
            }
        } else if (as.iou_max < 0.4f) {
            if (inrange) dc = kc * 0.75f * (-2.f * p * logf(1.f - p) + p * p / (1.f - p));
        }
    }
    dcls[(long)b * A + a] = dc;
    *reinterpret_cast<float4*>(dreg + ((long)b * A + a) * 4) = make_float4(dr[0], dr[1], dr[2], dr[3]);
}

__global__ void sigmoid_bwd_kernel(const float* __restrict__ dp, const float* __restrict__ p, float* __restrict__ dl, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const float pv = p[i]; dl[i] = dp[i] * pv * (1.f - pv); }
}

__global__ void sigmoid_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = 1.0f / (1.0f + expf(-x[i]));
}

// ------------------------------------------------------------------------------- PRN pieces
__global__ void add_softmax_rows_kernel(const float* __restrict__ a, long a_stride, const float* __restrict__ res, float* __restrict__ out, int cols, int relu) {
    __shared__ float sh[4];
    __shared__ float bc;
    const long row = blockIdx.x;
    const float* pa = a + row * a_stride; const float* pr = res + row * cols; float* po = out + row * cols;
    float mx = -INFINITY;
    const float lo = relu ? 0.f : -INFINITY;      // F.relu(dens2(.)) is folded in (posenet.py:343)
    for (int i = threadIdx.x; i < cols; i += 256) mx = fmaxf(mx, fmaxf(pa[i], lo) + pr[i]);
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) bc = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
    __syncthreads();
    mx = bc;
    float s = 0.f;
    for (int i = threadIdx.x; i < cols; i += 256) s += expf(fmaxf(pa[i], lo) + pr[i] - mx);
    s = wave_sum(s);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) bc = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    const float inv = 1.0f / bc;
    for (int i = threadIdx.x; i < cols; i += 256) po[i] = expf(fmaxf(pa[i], lo) + pr[i] - mx) * inv;
}

// dlogit[r][i] = p * (dp - sum_j p_j dp_j)  (softmax backward, one block per row)
__global__ void softmax_rows_bwd_kernel(const float* __restrict__ p, const float* __restrict__ dp, const float* __restrict__ pre, long pre_stride, float* __restrict__ dl, int cols) {
    __shared__ float sh[4];
    __shared__ float bc;
    const long row = blockIdx.x;
    const float* pp = p + row * cols; const float* pd = dp + row * cols; float* po = dl + row * cols;
    float s = 0.f;
    for (int i = threadIdx.x; i < cols; i += 256) s += pp[i] * pd[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) bc = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    const float dot = bc;
    for (int i = threadIdx.x; i < cols; i += 256) {
        const float g = pp[i] * (pd[i] - dot);
        po[i] = (pre == nullptr || pre[row * pre_stride + i] > 0.f) ? g : 0.f;      // relu mask of the pre-activation
    }
}

// BCELoss(mean) backward, torch semantics: d/dp = (p - y) / max(p (1 - p), 1e-12) / N
__global__ void bce_bwd_kernel(const float* __restrict__ p, const float* __restrict__ y, float* __restrict__ dp, long n,
                               const float* __restrict__ gscale) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float gs = (gscale ? gscale[0] : 1.f) / (float)n;
    const float pv = p[i];
    dp[i] = gs * (pv - y[i]) / fmaxf(pv * (1.f - pv), 1e-12f);
}

// counter-based dropout: keep element i iff hash(seed, i) >= p; kept values are scaled by 1/(1-p).
__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
template <typename T>
__global__ void dropout_kernel(const T* __restrict__ x, T* __restrict__ y, long n, unsigned long long seed, float p, float scale) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long h = mix64(mix64(seed) ^ (unsigned long long)i);
    const float u = (float)(h >> 40) * (1.0f / 16777216.0f);
    Elem<T>::st(y + i, u >= p ? Elem<T>::ld(x + i) * scale : 0.f);
}

__global__ void bce_partial_kernel(const float* __restrict__ p, const float* __restrict__ y, long n, float* __restrict__ partial) {
    __shared__ float sh[4];
    float s = 0.f;
    const long base = (long)blockIdx.x * 4096;
    for (int it = 0; it < 16; ++it) {
        const long i = base + it * 256 + threadIdx.x;
        if (i < n) {
            const float lp = fmaxf(logf(p[i]), -100.f), lq = fmaxf(logf(1.f - p[i]), -100.f);
            s += -(y[i] * lp + (1.f - y[i]) * lq);
        }
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ void mean_finalize_kernel(const float* __restrict__ partial, int chunks, double n, float* __restrict__ out) {
    __shared__ double sh[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < chunks; i += 256) s += (double)partial[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) { double t = 0.0; for (int i = 0; i < 256; ++i) t += sh[i]; out[0] = (float)(t / n); }
}

}  // namespace

extern "C" int mpn_mse_chunks(int64_t npix) { return (int)((npix * 18 + MSE_CHUNK - 1) / MSE_CHUNK); }

extern "C" int mpn_mse_heatmap_forward(const float* const* preds, const int64_t* pred_sP, const float* gt, const float* wgt,
                                       int64_t npix, float* partial, int chunks, float* out, void* stream) {
    MPN_CHECK_ARG(preds && pred_sP && gt && wgt && partial && out && npix > 0);
    MPN_CHECK_ARG(chunks == mpn_mse_chunks(npix));
    PredPtrs pr;
    for (int j = 0; j < 5; ++j) { pr.p[j] = preds[j]; pr.sP[j] = (long)pred_sP[j]; MPN_CHECK_ARG(preds[j] != nullptr); }
    hipLaunchKernelGGL(mse_fwd_kernel, dim3((unsigned)chunks), dim3(256), 0, (hipStream_t)stream, pr, gt, wgt, (long)npix * 18, partial);
    hipLaunchKernelGGL(mse_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)partial, chunks, (double)npix * 18.0, out);
    return mpn_launch_status();
}

extern "C" int mpn_mse_heatmap_backward(const float* const* preds, float* const* dpreds, const int64_t* pred_sP,
                                        const int64_t* dpred_sP, const int32_t* pred_C, const float* gt, const float* wgt,
                                        int64_t npix, const float* gscale, void* stream) {
    MPN_CHECK_ARG(preds && dpreds && pred_sP && dpred_sP && pred_C && gt && wgt && npix > 0);
    PredPtrs pr; DPredPtrs dp; int cmax = 0;
    for (int j = 0; j < 5; ++j) {
        pr.p[j] = preds[j]; pr.sP[j] = (long)pred_sP[j];
        dp.p[j] = dpreds[j]; dp.sP[j] = (long)dpred_sP[j]; dp.C[j] = pred_C[j];
        if (pred_C[j] > cmax) cmax = pred_C[j];
    }
    const long n = (long)npix * cmax;
    hipLaunchKernelGGL(mse_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pr, dp, gt, wgt, (long)npix, cmax, gscale);
    return mpn_launch_status();
}

extern "C" int mpn_mse_train_blocks(int B, int H, int W) { return (int)(((long)B * (H / 8) * (W / 8) + 7) / 8); }

extern "C" int mpn_mse_heatmap_train(const float* const* levels, void* const* dlevels, const int32_t* level_Cs, int dtype,
                                     const float* gt, const float* wgt, int64_t g_sB, int64_t g_sC, int64_t g_sH,
                                     int B, int H, int W, const float* gscale, float* partial, int blocks, float* out, void* stream) {
    MPN_CHECK_ARG(levels && dlevels && level_Cs && gt && wgt && partial && out && B > 0);
    MPN_CHECK_ARG(H > 0 && W > 0 && H % 8 == 0 && W % 8 == 0 && blocks == mpn_mse_train_blocks(B, H, W));
    MPN_CHECK_ARG(g_sH % 4 == 0 && g_sC % 4 == 0 && g_sB % 4 == 0 && ((uintptr_t)gt & 15) == 0 && ((uintptr_t)wgt & 15) == 0);
    TrainPtrs tp;
    for (int j = 0; j < 5; ++j) {
        MPN_CHECK_ARG(levels[j] && dlevels[j] && level_Cs[j] == 32);
        tp.p[j] = levels[j]; tp.d[j] = dlevels[j]; tp.Cs[j] = level_Cs[j];
    }
    const long ncell = (long)B * (H / 8) * (W / 8);
    MPN_DISPATCH_T(dtype, hipLaunchKernelGGL((mse_train_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, tp, gt, wgt,
                                             (long)g_sB, (long)g_sC, (long)g_sH, B, H, W, ncell, gscale, partial));
    hipLaunchKernelGGL(mse_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)partial, blocks,
                       (double)B * H * W * 18.0, out);
    return mpn_launch_status();
}

extern "C" int mpn_focal_blocks(int A) { return (A + 255) / 256; }

extern "C" int mpn_focal_forward(const float* cls, const float* reg, const float* anchors, const float* anno, int B, int A,
                                 int maxN, float* partial, float* per_img, float* out, void* stream) {
    MPN_CHECK_ARG(cls && reg && anchors && anno && partial && per_img && out && B > 0 && A > 0 && maxN > 0);
    const int blocksA = (A + 255) / 256;
    hipLaunchKernelGGL(focal_fwd_kernel, dim3(blocksA, B), dim3(256), 0, (hipStream_t)stream, cls, reg, anchors, anno, A, maxN, partial);
    hipLaunchKernelGGL(focal_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)partial, B, blocksA, per_img, out);
    return mpn_launch_status();
}

extern "C" int mpn_focal_backward(const float* cls, const float* reg, const float* anchors, const float* anno, int B, int A,
                                  int maxN, const float* per_img, const float* gscale, float* dcls, float* dreg, void* stream) {
    MPN_CHECK_ARG(cls && reg && anchors && anno && per_img && dcls && dreg && B > 0 && A > 0 && maxN > 0);
    hipLaunchKernelGGL(focal_bwd_kernel, dim3((A + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, cls, reg, anchors, anno, B, A, maxN,
                       per_img, gscale, dcls, dreg);
    return mpn_launch_status();
}

extern "C" int mpn_sigmoid_backward(const float* dp, const float* p, float* dlogit, int64_t n, void* stream) {
    MPN_CHECK_ARG(dp && p && dlogit && n > 0);
    hipLaunchKernelGGL(sigmoid_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dp, p, dlogit, (long)n);
    return mpn_launch_status();
}

extern "C" int mpn_sigmoid_forward(const float* x, float* y, int64_t n, void* stream) {
    MPN_CHECK_ARG(x && y && n > 0);
    hipLaunchKernelGGL(sigmoid_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, (long)n);
    return mpn_launch_status();
}

extern "C" int mpn_add_softmax_rows(const float* a, int64_t a_stride, const float* res, float* out, int rows, int cols, int relu, void* stream) {
    MPN_CHECK_ARG(a && res && out && rows > 0 && cols > 0 && a_stride >= cols);
    hipLaunchKernelGGL(add_softmax_rows_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, a, (long)a_stride, res, out, cols, relu);
    return mpn_launch_status();
}

extern "C" int mpn_softmax_rows_backward(const float* p, const float* dp, const float* pre_relu, int64_t pre_stride, float* dlogit, int rows, int cols, void* stream) {
    MPN_CHECK_ARG(p && dp && dlogit && rows > 0 && cols > 0 && (!pre_relu || pre_stride >= cols));
    hipLaunchKernelGGL(softmax_rows_bwd_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, p, dp, pre_relu, (long)pre_stride, dlogit, cols);
    return mpn_launch_status();
}

extern "C" int mpn_bce_mean_backward(const float* p, const float* label, float* dp, int64_t n, const float* gscale, void* stream) {
    MPN_CHECK_ARG(p && label && dp && n > 0);
    hipLaunchKernelGGL(bce_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, label, dp, (long)n, gscale);
    return mpn_launch_status();
}

extern "C" int mpn_dropout(const void* x, void* y, int64_t n, uint64_t seed, float p, int dtype, void* stream) {
    MPN_CHECK_ARG(x && y && n > 0 && p >= 0.f && p < 1.f);
    const float scale = 1.0f / (1.0f - p);
    MPN_DISPATCH_T(dtype, hipLaunchKernelGGL((dropout_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const T*)x, (T*)y, (long)n, (unsigned long long)seed, p, scale));
    return mpn_launch_status();
}

// log vector of the combined step: [0..4] level losses, [5] heat-map total, [6] max_ht, [7] min_ht, [8] detection total
// (cls + reg), [9] cls, [10] reg, [11] loss = heat-map total + detection total — the additions torch would do on the host side
__global__ void step_log_kernel(const float* __restrict__ kp8, const float* __restrict__ det2, float* __restrict__ logv) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float kp_total = 0.f, det_total = 0.f;
    if (kp8) {
        for (int i = 0; i < 8; ++i) logv[i] = kp8[i];
        kp_total = kp8[5];
    }
    if (det2) {
        det_total = det2[0] + det2[1];
        logv[8] = det_total; logv[9] = det2[0]; logv[10] = det2[1];
    }
    logv[11] = (kp8 && det2) ? kp_total + det_total : (kp8 ? kp_total : det_total);
}

extern "C" int mpn_step_log(const float* kp8, const float* det2, float* logv, void* stream) {
    MPN_CHECK_ARG(logv && (kp8 || det2));
    hipLaunchKernelGGL(step_log_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, kp8, det2, logv);
    return mpn_launch_status();
}

extern "C" int mpn_bce_chunks(int64_t n) { return (int)((n + 4095) / 4096); }

extern "C" int mpn_bce_mean_forward(const float* p, const float* label, int64_t n, float* partial, int chunks, float* out, void* stream) {
    MPN_CHECK_ARG(p && label && partial && out && n > 0 && chunks == mpn_bce_chunks(n));
    hipLaunchKernelGGL(bce_partial_kernel, dim3((unsigned)chunks), dim3(256), 0, (hipStream_t)stream, p, label, (long)n, partial);
    hipLaunchKernelGGL(mean_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)partial, chunks, (double)n, out);
    return mpn_launch_status();
}
