// peaks.hip — heat-map peak extraction on device (first half of the inference post-processing).
//
// Replaces network/joint_utils.py:19-31 (find_peaks: 3x3-cross maximum filter == value && value > thre1) and :61-138
// (NMS: per joint type, peaks in row-major order, optional refinement on a bicubically up-sampled 5x5 patch, running
// peak id) for a batch of images: one workgroup per (image, joint plane), a second tiny kernel turns plane-local slots
// into the running ids.
// The refinement restates cv2.resize(INTER_CUBIC) (float32, A = -0.75, s = (d + 0.5)/f - 0.5, taps clamped to the patch,
// horizontal then vertical pass) op for op with oracle/joint_oracle.py; built with -ffp-contract=off.
#include "common.h"

namespace {

constexpr int PK_THREADS = 256;

__device__ __forceinline__ void cubic_coeffs(float x, float (&c)[4]) {
    const float A = -0.75f;
    c[0] = ((A * (x + 1.f) - 5.f * A) * (x + 1.f) + 8.f * A) * (x + 1.f) - 4.f * A;
    c[1] = ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f;
    c[2] = ((A + 2.f) * (1.f - x) - (A + 3.f)) * (1.f - x) * (1.f - x) + 1.f;
    c[3] = 1.f - c[0] - c[1] - c[2];
}

__device__ __forceinline__ void cubic_taps(int d, double scale, int n_src, int (&idx)[4], float (&co)[4]) {
    const float fx = (float)(((double)d + 0.5) * scale - 0.5);
    const int sx = (int)floorf(fx);
    cubic_coeffs(fx - (float)sx, co);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int s = sx - 1 + k;
        idx[k] = s < 0 ? 0 : (s > n_src - 1 ? n_src - 1 : s);
    }
}

// grid (J, B): one workgroup per joint plane.  Phase 1: every thread owns a contiguous run of cells, counts its peaks,
// a wave-shuffle + 4-entry LDS scan gives the output slot (row-major order), the cell coordinates are parked in the
// slot.  Phase 2: the four waves take the plane's peaks in turn; the 64 lanes of a wave evaluate the up-sampled patch
// cells in parallel and a (value, lowest index) wave reduction picks np.argmax's cell.
__global__ void __launch_bounds__(PK_THREADS) heatmap_peaks_kernel(const float* __restrict__ heat, long sB, long sJ, long sY, long sX,
                                                                   int J, int H, int W, float thre1, double f, int refine,
                                                                   double* __restrict__ peaks, int* __restrict__ counts, int cap) {
    __shared__ int wave_tot[PK_THREADS / 64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int j = blockIdx.x, b = blockIdx.y;
    const float* __restrict__ m = heat + (long)b * sB + (long)j * sJ;
    auto at = [&](int y, int x) { return m[(long)y * sY + (long)x * sX]; };
    auto is_peak = [&](int c) {
        const int y = c / W, x = c - y * W;
        const float v = at(y, x);
        if (!(v > thre1)) return false;
        // out-of-range neighbours reflect onto the cell itself (scipy 'reflect'): they never raise the maximum
        if (y > 0 && at(y - 1, x) > v) return false;
        if (y + 1 < H && at(y + 1, x) > v) return false;
        if (x > 0 && at(y, x - 1) > v) return false;
        if (x + 1 < W && at(y, x + 1) > v) return false;
        return true;
    };
    const int cells = H * W;
    const int cpt = (cells + PK_THREADS - 1) / PK_THREADS;
    const int c_lo = t * cpt, c_hi = (c_lo + cpt < cells) ? c_lo + cpt : cells;
    int local = 0;
    for (int c = c_lo; c < c_hi; ++c) local += is_peak(c) ? 1 : 0;
    int incl = local;                                            // inclusive scan inside the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off);
        if (lane >= off) incl += v;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < PK_THREADS / 64; ++w) { if (w < wave) base += wave_tot[w]; total += wave_tot[w]; }
    int slot = base + incl - local;
    double* __restrict__ out = peaks + ((long)b * J + j) * cap * 4;
    for (int c = c_lo; c < c_hi && local > 0; ++c) {
        if (!is_peak(c)) continue;
        --local;
        if (slot < cap) {
            const int py = c / W, px = c - py * W;
            out[slot * 4 + 0] = (double)px;
            out[slot * 4 + 1] = (double)py;
            out[slot * 4 + 2] = (double)at(py, px);
            out[slot * 4 + 3] = (double)slot;                    // plane-local; heatmap_peak_ids_kernel adds the joint prefix
        }
        ++slot;
    }
    if (t == 0) counts[(long)b * J + j] = total;
    __threadfence_block();
    __syncthreads();
    const int stored = total < cap ? total : cap;
    for (int pk = wave; pk < stored; pk += PK_THREADS / 64) {
        const int px = (int)out[pk * 4 + 0], py = (int)out[pk * 4 + 1];
        double rx = 0.0, ry = 0.0;
        float score = at(py, px);
        if (refine) {
            const int x_min = px - 2 < 0 ? 0 : px - 2, y_min = py - 2 < 0 ? 0 : py - 2;
            const int x_max = px + 2 > W - 1 ? W - 1 : px + 2, y_max = py + 2 > H - 1 ? H - 1 : py + 2;
            const int sw = x_max - x_min + 1, sh = y_max - y_min + 1;
            const int dw = (int)rint((double)sw * f), dh = (int)rint((double)sh * f);
            const double scale = 1.0 / f;
            float best = -INFINITY;
            int bi = 0x7fffffff;
            for (int d = lane; d < dh * dw; d += 64) {
                const int dy = d / dw, dx = d - dy * dw;
                int yi[4], xi[4]; float yc[4], xc[4];
                cubic_taps(dy, scale, sh, yi, yc);
                cubic_taps(dx, scale, sw, xi, xc);
                float r[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int yy = y_min + yi[k];
                    float a = at(yy, x_min + xi[0]) * xc[0];
                    a = a + at(yy, x_min + xi[1]) * xc[1];
                    a = a + at(yy, x_min + xi[2]) * xc[2];
                    a = a + at(yy, x_min + xi[3]) * xc[3];
                    r[k] = a;
                }
                float v = r[0] * yc[0];
                v = v + r[1] * yc[1];
                v = v + r[2] * yc[2];
                v = v + r[3] * yc[3];
                if (v > best) { best = v; bi = d; }              // d ascends per lane: keeps the lane's first maximum
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {             // first maximum in row-major order = (max value, min index)
                const float ov = __shfl_xor(best, off);
                const int oi = __shfl_xor(bi, off);
                if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
            }
            const int by = bi / dw, bx = bi - by * dw;
            const double cy = ((double)(py - y_min) + 0.5) * f - 0.5, cx = ((double)(px - x_min) + 0.5) * f - 0.5;
            ry = (double)by - cy; rx = (double)bx - cx;
            score = best;
        }
        if (lane == 0) {
            out[pk * 4 + 0] = rint((((double)px + 0.5) * f - 0.5) + rx);
            out[pk * 4 + 1] = rint((((double)py + 0.5) * f - 0.5) + ry);
            out[pk * 4 + 2] = (double)score;
        }
    }
}

// cv2.resize of a float32 [Hs, Ws, C] image to [Hd, Wd, C] (evaluate/tester.py:67,213,296-299): INTER_CUBIC (cubic != 0: A = -0.75,
// four clamped taps per axis) or INTER_LINEAR (two taps; source coordinate clamped as OpenCV does), horizontal pass then
// vertical pass, float32, left-to-right sums.  One thread per destination element.
__global__ void resize_kernel(const float* __restrict__ src, long sY, long sX, long sC, int Hs, int Ws, int C,
                              float* __restrict__ dst, int Hd, int Wd, int cubic, double inv_fy, double inv_fx) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)Hd * Wd * C) return;
    const int c = (int)(i % C);
    const int dx = (int)((i / C) % Wd);
    const int dy = (int)(i / ((long)C * Wd));
    // dsize form: scale = src / dst; fx / fy form (cv2.resize(img, None, fx=, fy=)): scale = 1 / f exactly
    const double scale_x = inv_fx > 0.0 ? inv_fx : (double)Ws / (double)Wd, scale_y = inv_fy > 0.0 ? inv_fy : (double)Hs / (double)Hd;
    auto at = [&](int y, int x) { return src[(long)y * sY + (long)x * sX + (long)c * sC]; };
    if (cubic) {
        int yi[4], xi[4]; float yc[4], xc[4];
        cubic_taps(dy, scale_y, Hs, yi, yc);
        cubic_taps(dx, scale_x, Ws, xi, xc);
        float r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float a = at(yi[k], xi[0]) * xc[0];
            a = a + at(yi[k], xi[1]) * xc[1];
            a = a + at(yi[k], xi[2]) * xc[2];
            a = a + at(yi[k], xi[3]) * xc[3];
            r[k] = a;
        }
        float v = r[0] * yc[0];
        v = v + r[1] * yc[1];
        v = v + r[2] * yc[2];
        v = v + r[3] * yc[3];
        dst[i] = v;
    } else {
        float fx = (float)(((double)dx + 0.5) * scale_x - 0.5), fy = (float)(((double)dy + 0.5) * scale_y - 0.5);
        int sx = (int)floorf(fx), sy = (int)floorf(fy);
        fx -= (float)sx; fy -= (float)sy;
        if (sx < 0) { fx = 0.f; sx = 0; }
        if (sx >= Ws - 1) { fx = 0.f; sx = Ws - 1; }
        if (sy < 0) { fy = 0.f; sy = 0; }
        if (sy >= Hs - 1) { fy = 0.f; sy = Hs - 1; }
        const int sx1 = sx + 1 < Ws ? sx + 1 : Ws - 1, sy1 = sy + 1 < Hs ? sy + 1 : Hs - 1;
        const float r0 = at(sy, sx) * (1.f - fx) + at(sy, sx1) * fx;
        const float r1 = at(sy1, sx) * (1.f - fx) + at(sy1, sx1) * fx;
        dst[i] = r0 * (1.f - fy) + r1 * fy;
    }
}

// ids run over the joint types of an image in order: id = (peaks of earlier joints) + slot
__global__ void heatmap_peak_ids_kernel(double* __restrict__ peaks, const int* __restrict__ counts, int J, int cap) {
    const int b = blockIdx.x;
    int prefix = 0;
    for (int j = 0; j < J; ++j) {
        const int n = counts[(long)b * J + j];
        const int stored = n < cap ? n : cap;
        for (int s = threadIdx.x; s < stored; s += blockDim.x) peaks[(((long)b * J + j) * cap + s) * 4 + 3] = (double)(prefix + s);
        prefix += n;
    }
}

}  // namespace

extern "C" int mpn_heatmap_peaks(const float* heat, int64_t sB, int64_t sJ, int64_t sY, int64_t sX, int B, int J, int H, int W,
                                 float thre1, double upsamp, int refine, double* peaks, int32_t* counts, int cap, void* stream) {
    MPN_CHECK_ARG(heat && peaks && counts && B > 0 && J > 0 && H > 0 && W > 0 && cap > 0 && upsamp > 0.0);
    MPN_CHECK_ARG((long)H * W < 0x7fffffffL);
    hipLaunchKernelGGL(heatmap_peaks_kernel, dim3((unsigned)J, (unsigned)B), dim3(PK_THREADS), 0, (hipStream_t)stream, heat, (long)sB,
                       (long)sJ, (long)sY, (long)sX, J, H, W, thre1, upsamp, refine, peaks, (int*)counts, cap);
    int rc = mpn_launch_status();
    if (rc != 0) return rc;
    hipLaunchKernelGGL(heatmap_peak_ids_kernel, dim3((unsigned)B), dim3(64), 0, (hipStream_t)stream, peaks, (const int*)counts, J, cap);
    return mpn_launch_status();
}

extern "C" int mpn_resize(const float* src, int64_t sY, int64_t sX, int64_t sC, int Hs, int Ws, int C, float* dst, int Hd, int Wd,
                          int cubic, double inv_fy, double inv_fx, void* stream) {
    MPN_CHECK_ARG(src && dst && Hs > 0 && Ws > 0 && C > 0 && Hd > 0 && Wd > 0 && inv_fy >= 0.0 && inv_fx >= 0.0);
    const long n = (long)Hd * Wd * C;
    MPN_CHECK_ARG(n < 0x7fffffffL * 256L);
    hipLaunchKernelGGL(resize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, (long)sY, (long)sX, (long)sC,
                       Hs, Ws, C, dst, Hd, Wd, cubic, inv_fy, inv_fx);
    return mpn_launch_status();
}
