// peaks.hip — heat-map peak extraction on device (first half of the inference post-processing).
//
// Replaces network/joint_utils.py:19-31 (find_peaks: 3x3-cross maximum filter == value && value > thre1) and :61-138
// (NMS: per joint type, peaks in row-major order, optional refinement on a bicubically up-sampled 5x5 patch, running
// peak id) for a batch of images: flags -> per-plane compaction -> refinement over a device-wide peak list (three launches).
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

// ---- round 6: three bandwidth-shaped launches instead of one workgroup per plane --------------------------------------------
//   flags   : every cell's peak test, lanes on CONSECUTIVE addresses of whichever layout the caller passes (channels-last: the
//             J values of a pixel are adjacent, a workgroup takes whole rows of all planes and transposes the flag bytes through
//             LDS; planar / anything else: lanes along x).  Output: one bit per cell, 64-bit words [b][j][y][ceil(W / 64)].
//   compact : one workgroup per plane walks the plane's words (a few KB): popcount + scan = row-major slots, writes the cell
//             coordinates and count, appends its stored slots to ONE device-wide list.
//   refine  : every wave of a fixed grid takes list entries in turn (a peak's up-sampled patch is evaluated by the 64 lanes) —
//             the refinement load spreads over the chip however unevenly the peaks are distributed over planes — and writes the
//             final (x, y, score, id); the id prefix over an image's earlier joint types is 17 loads.
// The peak test is evaluated once per cell; the arithmetic of the refinement is unchanged (bit-exact with round 5's kernel).

__device__ __forceinline__ bool peak_test(float v, float thre1, bool has_u, float u, bool has_d, float d, bool has_l, float l, bool has_r, float r) {
    if (!(v > thre1)) return false;
    // out-of-range neighbours reflect onto the cell itself (scipy 'reflect'): they never raise the maximum
    if (has_u && u > v) return false;
    if (has_d && d > v) return false;
    if (has_l && l > v) return false;
    if (has_r && r > v) return false;
    return true;
}

constexpr int PK_ROWS = 4;            // rows per workgroup (channels-last flags kernel; its row index arithmetic assumes 4)
constexpr int PK_ROW_W = 64;          // widest up-sampled patch row the refinement stages in LDS (5 * upsamp <= 64; wider: direct form)

// channels-last (sJ == 1, sX == J, sY == W * J): grid (ceil(H / PK_ROWS), B).  A row of all planes is W * J consecutive floats.
__global__ void __launch_bounds__(PK_THREADS) peaks_flags_cl_kernel(const float* __restrict__ heat, long sB, int J, int H, int W, float thre1,
                                                                     unsigned long long* __restrict__ bits, int wpr) {
    extern __shared__ unsigned char fl[];                        // [PK_ROWS][W * J] flag bytes
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int b = blockIdx.y, y0 = blockIdx.x * PK_ROWS;
    const int rowlen = W * J;
    const float* __restrict__ m = heat + (long)b * sB;
    const int rows = (H - y0 < PK_ROWS) ? H - y0 : PK_ROWS;
    // the tile's rows are one contiguous range of rows * rowlen floats: eight independent (coalesced) loads in flight per thread
    const float* __restrict__ tile = m + (long)y0 * rowlen;
    const int total = rows * rowlen;
    constexpr int U = 8;
    for (int i0 = t; i0 < total; i0 += U * PK_THREADS) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * PK_THREADS;
            v[u] = i < total ? tile[i] : -INFINITY;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * PK_THREADS;
            if (i >= total) break;
            bool f = false;
            if (v[u] > thre1) {                                  // sparse maps: one load per cell
                const int r = (i >= rowlen) + (i >= 2 * rowlen) + (i >= 3 * rowlen);
                const int e = i - r * rowlen, y = y0 + r;
                const bool hu = y > 0, hd = y + 1 < H, hl = e >= J, hr = e + J < rowlen;
                f = peak_test(v[u], thre1, hu, hu ? tile[i - rowlen] : 0.f, hd, hd ? tile[i + rowlen] : 0.f, hl, hl ? tile[i - J] : 0.f,
                              hr, hr ? tile[i + J] : 0.f);
            }
            fl[i] = f ? 1 : 0;
        }
    }
    __syncthreads();
    const int pairs = J * wpr;
    for (int q = wave; q < rows * pairs; q += PK_THREADS / 64) {
        const int r = q / pairs, pq = q - r * pairs;
        const int j = pq / wpr, xw = pq - j * wpr;
        const int x = xw * 64 + lane;
        const bool f = x < W && fl[r * rowlen + x * J + j] != 0;
        const unsigned long long word = __ballot(f);
        if (lane == 0) bits[(((long)b * J + j) * H + (y0 + r)) * wpr + xw] = word;
    }
}

// any strides (coalesced when sX == 1): grid (ceil(H / (4 * PK_ROWS)), J, B); a wave takes a row, lanes along x.
__global__ void __launch_bounds__(PK_THREADS) peaks_flags_kernel(const float* __restrict__ heat, long sB, long sJ, long sY, long sX, int J, int H, int W,
                                                                  float thre1, unsigned long long* __restrict__ bits, int wpr) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int j = blockIdx.y, b = blockIdx.z;
    const float* __restrict__ m = heat + (long)b * sB + (long)j * sJ;
    const int y_end = ((int)blockIdx.x + 1) * 4 * PK_ROWS < H ? ((int)blockIdx.x + 1) * 4 * PK_ROWS : H;
    for (int y = blockIdx.x * 4 * PK_ROWS + wave; y < y_end; y += PK_THREADS / 64) {
        const float* __restrict__ row = m + (long)y * sY;
        constexpr int U = 4;
        for (int xw0 = 0; xw0 < wpr; xw0 += U) {
            float v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int x = (xw0 + u) * 64 + lane;
                v[u] = x < W ? row[(long)x * sX] : -INFINITY;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (xw0 + u >= wpr) break;
                const int x = (xw0 + u) * 64 + lane;
                bool f = false;
                if (v[u] > thre1) {
                    const float* c = row + (long)x * sX;
                    const bool hu = y > 0, hd = y + 1 < H, hl = x > 0, hr = x + 1 < W;
                    f = peak_test(v[u], thre1, hu, hu ? c[-sY] : 0.f, hd, hd ? c[sY] : 0.f, hl, hl ? c[-sX] : 0.f, hr, hr ? c[sX] : 0.f);
                }
                const unsigned long long word = __ballot(f);
                if (lane == 0) bits[(((long)b * J + j) * H + y) * wpr + xw0 + u] = word;
            }
        }
    }
}

// grid (J, B): row-major slots of a plane from its flag words; cell coordinates + provisional slot parked in the output rows,
// the plane's stored slots appended to the device-wide work list of the refinement launch.
__global__ void __launch_bounds__(PK_THREADS) peaks_compact_kernel(const float* __restrict__ heat, long sB, long sJ, long sY, long sX, int J, int H, int W,
                                                                    const unsigned long long* __restrict__ bits, int wpr, double* __restrict__ peaks,
                                                                    int* __restrict__ counts, int cap, int* __restrict__ list, int* __restrict__ list_n) {
    __shared__ int wave_tot[PK_THREADS / 64];
    __shared__ int gbase;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int j = blockIdx.x, b = blockIdx.y;
    const long plane = (long)b * J + j;
    const float* __restrict__ m = heat + (long)b * sB + (long)j * sJ;
    const unsigned long long* __restrict__ pw = bits + plane * H * wpr;
    double* __restrict__ out = peaks + plane * cap * 4;
    const int nwords = H * wpr;
    int running = 0;
    for (int w0 = 0; w0 < nwords; w0 += PK_THREADS) {
        const int wi = w0 + t;
        unsigned long long word = wi < nwords ? pw[wi] : 0ull;
        const int local = __popcll(word);
        int incl = local;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int v = __shfl_up(incl, off);
            if (lane >= off) incl += v;
        }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        int base = running, total = 0;
#pragma unroll
        for (int w = 0; w < PK_THREADS / 64; ++w) { if (w < wave) base += wave_tot[w]; total += wave_tot[w]; }
        int slot = base + incl - local;
        const int y = wi / wpr, xw = wi - y * wpr;
        while (word && slot < cap) {
            const int bit = __ffsll((long long)word) - 1;
            word &= word - 1;
            const int x = xw * 64 + bit;
            out[slot * 4 + 0] = (double)x;
            out[slot * 4 + 1] = (double)y;
            out[slot * 4 + 2] = (double)m[(long)y * sY + (long)x * sX];
            out[slot * 4 + 3] = (double)slot;
            ++slot;
        }
        running += total;
        __syncthreads();                                         // wave_tot is rewritten by the next chunk
    }
    const int stored = running < cap ? running : cap;
    if (t == 0) {
        counts[plane] = running;
        gbase = stored > 0 ? atomicAdd(list_n, stored) : 0;
    }
    __syncthreads();
    const int g0 = gbase;
    for (int s = t; s < stored; s += PK_THREADS) list[g0 + s] = (int)(plane * cap + s);
}

// fixed grid; wave-granular stride over the work list.  One wave per peak: final coordinates, refined score, running id.
__global__ void __launch_bounds__(PK_THREADS) peaks_refine_kernel(const float* __restrict__ heat, long sB, long sJ, long sY, long sX, int J, int H, int W,
                                                                   double f, int refine, double* __restrict__ peaks, const int* __restrict__ counts,
                                                                   int cap, const int* __restrict__ list, const int* __restrict__ list_n) {
    __shared__ float rows_all[PK_THREADS / 64][5 * PK_ROW_W];
    const int lane = threadIdx.x & 63;
    float* __restrict__ rows = rows_all[threadIdx.x >> 6];
    const int nwaves = gridDim.x * (PK_THREADS / 64);
    const int n = *list_n;
    for (int e = blockIdx.x * (PK_THREADS / 64) + (threadIdx.x >> 6); e < n; e += nwaves) {
        const int entry = list[e];
        const int plane = entry / cap, pk = entry - plane * cap;
        const int b = plane / J, j = plane - b * J;
        const float* __restrict__ m = heat + (long)b * sB + (long)j * sJ;
        auto at = [&](int y, int x) { return m[(long)y * sY + (long)x * sX]; };
        double* __restrict__ out = peaks + (long)plane * cap * 4;
        const int px = (int)out[pk * 4 + 0], py = (int)out[pk * 4 + 1];
        int prefix = 0;                                          // ids run over the joint types of an image in order
        for (int jj = lane; jj < j; jj += 64) prefix += counts[(long)b * J + jj];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) prefix += __shfl_xor(prefix, off);
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
            // cv2's resize is separable: the horizontal pass of a source row serves every destination row that taps it.  rows[sy][dx]
            // (<= 5 x dw values) is built once per peak in this wave's LDS slice; each up-sampled cell then costs four LDS reads instead
            // of sixteen global loads.  Same products and the same left-to-right sums as the direct form: bit-identical.
            const bool staged = dw <= PK_ROW_W;
            if (staged) {
                for (int q = lane; q < sh * dw; q += 64) {
                    const int sy = q / dw, dx = q - sy * dw;
                    int xi[4]; float xc[4];
                    cubic_taps(dx, scale, sw, xi, xc);
                    const int yy = y_min + sy;
                    float a = at(yy, x_min + xi[0]) * xc[0];
                    a = a + at(yy, x_min + xi[1]) * xc[1];
                    a = a + at(yy, x_min + xi[2]) * xc[2];
                    a = a + at(yy, x_min + xi[3]) * xc[3];
                    rows[sy * PK_ROW_W + dx] = a;
                }
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_s_waitcnt(0xc07f);              // lgkmcnt(0): this wave's LDS writes have landed before it reads them
            }
            for (int d = lane; d < dh * dw; d += 64) {
                const int dy = d / dw, dx = d - dy * dw;
                int yi[4]; float yc[4];
                cubic_taps(dy, scale, sh, yi, yc);
                float r[4];
                if (staged) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) r[k] = rows[yi[k] * PK_ROW_W + dx];
                } else {
                    int xi[4]; float xc[4];
                    cubic_taps(dx, scale, sw, xi, xc);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int yy = y_min + yi[k];
                        float a = at(yy, x_min + xi[0]) * xc[0];
                        a = a + at(yy, x_min + xi[1]) * xc[1];
                        a = a + at(yy, x_min + xi[2]) * xc[2];
                        a = a + at(yy, x_min + xi[3]) * xc[3];
                        r[k] = a;
                    }
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
            __builtin_amdgcn_wave_barrier();                     // the next peak's rows[] must not overtake this one's reads
        }
        if (lane == 0) {
            out[pk * 4 + 0] = rint((((double)px + 0.5) * f - 0.5) + rx);
            out[pk * 4 + 1] = rint((((double)py + 0.5) * f - 0.5) + ry);
            out[pk * 4 + 2] = (double)score;
            out[pk * 4 + 3] = (double)(prefix + pk);
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

}  // namespace

static inline int64_t pk_align(int64_t v) { return (v + 255) / 256 * 256; }

extern "C" int64_t mpn_heatmap_peaks_workspace_bytes(int B, int J, int H, int W, int cap) {
    if (B <= 0 || J <= 0 || H <= 0 || W <= 0 || cap <= 0) return 0;
    const int64_t wpr = (W + 63) / 64;
    return pk_align((int64_t)B * J * H * wpr * 8) + pk_align((int64_t)B * J * cap * 4) + 256;
}

extern "C" int mpn_heatmap_peaks(const float* heat, int64_t sB, int64_t sJ, int64_t sY, int64_t sX, int B, int J, int H, int W,
                                 float thre1, double upsamp, int refine, double* peaks, int32_t* counts, int cap, void* workspace,
                                 void* stream) {
    MPN_CHECK_ARG(heat && peaks && counts && workspace && B > 0 && J > 0 && H > 0 && W > 0 && cap > 0 && upsamp > 0.0);
    MPN_CHECK_ARG((long)H * W < 0x7fffffffL && (long)B * J * cap < 0x7fffffffL && B <= 65535 && J <= 65535);
    const int wpr = (W + 63) / 64;
    unsigned long long* bits = (unsigned long long*)workspace;
    int* list = (int*)((char*)workspace + pk_align((int64_t)B * J * H * wpr * 8));
    int* list_n = (int*)((char*)list + pk_align((int64_t)B * J * cap * 4));
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(list_n, 0, sizeof(int), st) != hipSuccess) return (int)hipGetLastError();
    const size_t lds = (size_t)PK_ROWS * W * J;
    if (sJ == 1 && sX == J && sY == (int64_t)W * J && lds <= 60 * 1024) {
        hipLaunchKernelGGL(peaks_flags_cl_kernel, dim3((unsigned)((H + PK_ROWS - 1) / PK_ROWS), (unsigned)B), dim3(PK_THREADS), lds, st, heat, (long)sB,
                           J, H, W, thre1, bits, wpr);
    } else {
        hipLaunchKernelGGL(peaks_flags_kernel, dim3((unsigned)((H + 4 * PK_ROWS - 1) / (4 * PK_ROWS)), (unsigned)J, (unsigned)B), dim3(PK_THREADS), 0, st,
                           heat, (long)sB, (long)sJ, (long)sY, (long)sX, J, H, W, thre1, bits, wpr);
    }
    int rc = mpn_launch_status();
    if (rc != 0) return rc;
    hipLaunchKernelGGL(peaks_compact_kernel, dim3((unsigned)J, (unsigned)B), dim3(PK_THREADS), 0, st, heat, (long)sB, (long)sJ, (long)sY, (long)sX,
                       J, H, W, bits, wpr, peaks, (int*)counts, cap, list, list_n);
    rc = mpn_launch_status();
    if (rc != 0) return rc;
    // enough waves to fill the chip on a dense (noise) workload; idle waves of a sparse one read the list length and leave
    long want = ((long)B * J * (cap < 16 ? cap : 16) + PK_THREADS / 64 - 1) / (PK_THREADS / 64);
    const unsigned grid = (unsigned)(want < 1 ? 1 : (want > 4096 ? 4096 : want));
    hipLaunchKernelGGL(peaks_refine_kernel, dim3(grid), dim3(PK_THREADS), 0, st, heat, (long)sB, (long)sJ, (long)sY, (long)sX, J, H, W, upsamp, refine,
                       peaks, (const int*)counts, cap, (const int*)list, (const int*)list_n);
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
