// prn_assign.hip — device half of the pose-residual-network person assignment (evaluate/tester.py:333-513).
//
// The reference builds, per detected box, a 56x36x17 one-hot map of the heat-map peaks that fall inside the box
// (tester.py:363-392), blurs every joint plane with skimage's gaussian (sigma 1, 'nearest' edges; :395-397), runs the PRN
// once PER BOX at batch 1 (:399-406), and scores every peak by the 15x15 window sum of the PRN output around its cell
// (:414-430) — python triple loops on the host.  Here:
//   prn_build_maps_kernel : one workgroup per (box, joint type): cell assignment in the reference's overwrite order with its
//                           exact float64 arithmetic and its one-branch clamp chain (including Python's negative-index wrap),
//                           then the separable 9-tap blur in scipy.ndimage.correlate1d's summation order, written as the
//                           float32 [box][y][x][joint] tensor the PRN consumes — all boxes of all images in ONE launch, so the
//                           PRN forward is one batched call.
//   prn_scores_kernel     : per (box, joint type): window sums at the occupied cells in numpy's float32 pairwise order, and
//                           the first arg-max of the plane (the fallback of tester.py:471-483).
// Compiled with -ffp-contract=off: results are bit-identical to the numpy/scipy arithmetic of the reference.
#include "common.h"

namespace {

constexpr int kMaxCells = 4096;        // >= 56*36 (coeff 2, the only size the reference's reshape at tester.py:404 allows)

__global__ void __launch_bounds__(256) prn_build_maps_kernel(const double* __restrict__ peaks,      // [n][2] (x, y), grouped by image then joint type
                                                             const int* __restrict__ joint_off,   // [nimg][18] offsets into peaks
                                                             const double* __restrict__ boxes,    // [nb][4] (x, y, w, h)
                                                             const int* __restrict__ box_img,     // [nb]
                                                             int H, int W, double in_thres, const double* __restrict__ wts /*[9]*/,
                                                             int* __restrict__ occ,               // [nb][17][H][W]: 1 + peak id, 0 = empty
                                                             float* __restrict__ prn_in,          // [nb][H][W][17]
                                                             int* __restrict__ err) {
    __shared__ int occ_s[kMaxCells];
    __shared__ double pa[kMaxCells];
    __shared__ double pb[kMaxCells];
    const int t = blockIdx.x, b = blockIdx.y;
    const int img = box_img[b];
    const int n = H * W;
    for (int i = threadIdx.x; i < n; i += blockDim.x) occ_s[i] = 0;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double bx = boxes[b * 4 + 0], by = boxes[b * 4 + 1], bw = boxes[b * 4 + 2], bh = boxes[b * 4 + 3];
        const int* off = joint_off + img * 18;
        const int first = off[0];
        const double lo_x = bx - bw * in_thres, lo_y = by - bh * in_thres;
        const double hi_x = bx + bw * (1.0 + in_thres), hi_y = by + bh * (1.0 + in_thres);
        const double x_scale = (double)W / ceil(bw), y_scale = (double)H / ceil(bh);       // tester.py:374-375
        for (int p = off[t]; p < off[t + 1]; ++p) {                                        // instances in annotation order: later ones overwrite
            const double px = peaks[p * 2 + 0], py = peaks[p * 2 + 1];
            if (!(px > lo_x && py > lo_y && px < hi_x && py < hi_y)) continue;             // tester.py:369-372
            int x0 = (int)((px - bx) * x_scale), y0 = (int)((py - by) * y_scale);          // int(): truncation toward zero
            if (x0 >= W && y0 >= H) { x0 = W - 1; y0 = H - 1; }                            // ONE branch of the chain applies (tester.py:378-390)
            else if (x0 >= W) x0 = W - 1;
            else if (y0 >= H) y0 = H - 1;
            else if (x0 < 0 && y0 < 0) { x0 = 0; y0 = 0; }
            else if (x0 < 0) x0 = 0;
            else if (y0 < 0) y0 = 0;
            if (x0 < 0) x0 += W;                                                           // numpy negative index wraps
            if (y0 < 0) y0 += H;
            if (x0 < 0 || x0 >= W || y0 < 0 || y0 >= H) { atomicExch(err, 1); continue; }   // IndexError in the reference
            occ_s[y0 * W + x0] = p - first + 1;
        }
    }
    __syncthreads();
    double w[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) w[k] = wts[k];
    for (int i = threadIdx.x; i < n; i += blockDim.x) pa[i] = occ_s[i] ? 1.0 : 0.0;
    __syncthreads();
    // axis 0 (rows), 'nearest' extension, scipy's symmetric-kernel order: centre, then pairs from the far tap inwards
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int y = i / W, x = i - y * W;
        double tmp = pa[i] * w[4];
#pragma unroll
        for (int jj = -4; jj < 0; ++jj) {
            int ya = y + jj, yb = y - jj;
            ya = ya < 0 ? 0 : ya; yb = yb >= H ? H - 1 : yb;
            tmp += (pa[ya * W + x] + pa[yb * W + x]) * w[4 + jj];
        }
        pb[i] = tmp;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int y = i / W, x = i - y * W;
        double tmp = pb[i] * w[4];
#pragma unroll
        for (int jj = -4; jj < 0; ++jj) {
            int xa = x + jj, xb = x - jj;
            xa = xa < 0 ? 0 : xa; xb = xb >= W ? W - 1 : xb;
            tmp += (pb[y * W + xa] + pb[y * W + xb]) * w[4 + jj];
        }
        prn_in[((long)b * n + i) * 17 + t] = (float)tmp;
        occ[((long)b * 17 + t) * n + i] = occ_s[i];
    }
}

// np.sum of a float32 window = numpy's pairwise sum over the window flattened in row-major order (verified against numpy
// 2.2 on 400 random windows, tools-free: tests/test_prn_assign.py): blocks of <= 128 values go through eight running
// accumulators combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) plus a sequential tail; longer inputs split at
// n/2 rounded down to a multiple of 8 (numpy/core/src/umath/loops_utils.h.src, @TYPE@_pairwise_sum).
struct WinView {
    const float* plane; int W, r0, c0, ncols;
    __device__ __forceinline__ float at(int k) const {
        const int r = k / ncols, c = k - r * ncols;
        return plane[(long)((r0 + r) * W + c0 + c) * 17];
    }
};
__device__ __forceinline__ float np_pairwise_block(const WinView& v, int k0, int n) {
    if (n < 8) {
        float res = 0.f;
        for (int i = 0; i < n; ++i) res += v.at(k0 + i);
        return res;
    }
    float r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = v.at(k0 + j);
    int i;
    for (i = 8; i < n - (n % 8); i += 8)
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] += v.at(k0 + i + j);
    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += v.at(k0 + i);
    return res;
}
__device__ __forceinline__ float np_sum_window(const WinView& v, int n) {      // n <= 256
    if (n <= 128) return np_pairwise_block(v, 0, n);
    int n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise_block(v, 0, n2) + np_pairwise_block(v, n2, n - n2);
}

__global__ void __launch_bounds__(64) prn_scores_kernel(const float* __restrict__ prn_out,     // [nb][H][W][17]
                                                        const int* __restrict__ occ,          // [nb][17][H][W]
                                                        int H, int W, int N,
                                                        float* __restrict__ score,            // [nb][17][H][W], valid where occ > 0
                                                        int* __restrict__ argmax) {           // [nb][17] first row-major maximum
    const int t = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    const int n = H * W;
    const float* plane = prn_out + (long)b * n * 17 + t;           // element (y, x) at plane[(y*W + x) * 17]
    const int* oc = occ + ((long)b * 17 + t) * n;
    float* sc = score + ((long)b * 17 + t) * n;
    const int half = (N - 1) / 2;
    float best = -INFINITY;
    int best_i = 0x7fffffff;
    for (int i = lane; i < n; i += 64) {
        const float v = plane[(long)i * 17];
        if (v > best) { best = v; best_i = i; }                    // lanes scan ascending indices: first maximum per lane
        if (oc[i] > 0) {
            const int y = i / W, x = i - y * W;
            // crop(img, (y, x), N) of prn_gaussian.py:134-158: rows [y-half, y+half], cols [x-half, x+half], clipped
            const int r0 = y - half < 0 ? 0 : y - half, r1 = y + half + 1 > H - 1 ? H : y + half + 1;
            const int c0 = x - half < 0 ? 0 : x - half, c1 = x + half + 1 > W - 1 ? W : x + half + 1;
            // exact ties between candidates occur (windows clipped by the same border) and are resolved by np.argsort's order
            // in the reference, so the sum itself must be numpy's, bit for bit
            WinView v = {plane, W, r0, c0, c1 - c0};
            sc[i] = np_sum_window(v, (r1 - r0) * (c1 - c0));
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const float ov = __shfl_xor(best, m, 64);
        const int oi = __shfl_xor(best_i, m, 64);
        if (ov > best || (ov == best && oi < best_i)) { best = ov; best_i = oi; }
    }
    if (lane == 0) argmax[b * 17 + t] = best_i;
}

// prn_scores_kernel with a COMPACT result: instead of the full [nb][17][H][W] score plane (274 KB per box with its occupancy map:
// 140 MB per 64-image batch over PCIe), the occupied cells of every (box, joint type) plane leave as a short list in row-major cell
// order — exactly the order np.argwhere visits them at tester.py:415 — of (peak id, window score).
__global__ void __launch_bounds__(64) prn_scores_compact_kernel(const float* __restrict__ prn_out, const int* __restrict__ occ,
                                                                int H, int W, int N, int cap,
                                                                int* __restrict__ cand_n,          // [nb][17]
                                                                int* __restrict__ cand_id,         // [nb][17][cap] peak id (occ - 1)
                                                                float* __restrict__ cand_score,    // [nb][17][cap]
                                                                int* __restrict__ argmax, int* __restrict__ overflow) {
    const int t = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    const int n = H * W;
    const float* plane = prn_out + (long)b * n * 17 + t;
    const int* oc = occ + ((long)b * 17 + t) * n;
    const long slot0 = ((long)b * 17 + t) * cap;
    const int half = (N - 1) / 2;
    float best = -INFINITY;
    int best_i = 0x7fffffff;
    int count = 0;                                                 // wave-uniform
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + lane;
        const bool in = i < n;
        const float v = in ? plane[(long)i * 17] : -INFINITY;
        if (in && v > best) { best = v; best_i = i; }
        const int o = in ? oc[i] : 0;
        const bool hit = o > 0;
        const unsigned long long m = __ballot(hit);
        if (hit) {
            const int y = i / W, x = i - y * W;
            const int r0 = y - half < 0 ? 0 : y - half, r1 = y + half + 1 > H - 1 ? H : y + half + 1;
            const int c0 = x - half < 0 ? 0 : x - half, c1 = x + half + 1 > W - 1 ? W : x + half + 1;
            WinView wv = {plane, W, r0, c0, c1 - c0};
            const float sc = np_sum_window(wv, (r1 - r0) * (c1 - c0));
            const int pos = count + __popcll(m & ((1ull << lane) - 1ull));
            if (pos < cap) { cand_id[slot0 + pos] = o - 1; cand_score[slot0 + pos] = sc; }
        }
        count += __popcll(m);
    }
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) {
        const float ov = __shfl_xor(best, k, 64);
        const int oi = __shfl_xor(best_i, k, 64);
        if (ov > best || (ov == best && oi < best_i)) { best = ov; best_i = oi; }
    }
    if (lane == 0) {
        argmax[b * 17 + t] = best_i;
        cand_n[b * 17 + t] = count;
        if (count > cap) atomicExch(overflow, 1);
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------------
// Host half of the assignment (tester.py:432-485) on the compact candidate lists: plain C++, no device work.  For every image and
// joint type the (boxes x peaks) score table is matched greedily exactly as the reference's numpy loops do.  Where the outcome
// would depend on how np.argsort orders EQUAL positive scores (numpy's small-array sorts are SIMD networks, not stable: the
// order is an implementation detail of the numpy build) the function does not guess: it flags that (image, joint type) in
// `tie_flags` and leaves it to the caller's numpy path.
// ---------------------------------------------------------------------------------------------------------------------------
extern "C" int mpn_prn_match_host(int nimg, const int32_t* box_start /*[nimg+1]*/, const double* boxes /*[nb][4] x,y,w,h*/,
                                  const int32_t* joint_off /*[nimg][18]*/, const double* peaks /*[np][2]*/,
                                  const int32_t* cand_n /*[nb][17]*/, const int32_t* cand_id, const float* cand_score, int cap,
                                  const int32_t* argmax /*[nb][17]*/, int H, int W,
                                  double* out_kp /*[nb][17][3], zero-filled by the caller*/, uint8_t* tie_flags /*[nimg][17]*/) {
    if (!box_start || !boxes || !joint_off || !peaks || !cand_n || !cand_id || !cand_score || !argmax || !out_kp || !tie_flags) return MPN_E_BADARG;
    for (int im = 0; im < nimg; ++im) {
        const int b0 = box_start[im], n = box_start[im + 1] - b0;
        if (n <= 0) continue;
        const int first = joint_off[im * 18];
        bool any_empty = false;
        for (int t = 0; t < 17; ++t) {
            // columns = distinct peak ids among the candidates of this joint type (a peak id names one peak of this image)
            int ids[512];
            int ncol = 0;
            int total = 0;
            for (int bi = 0; bi < n; ++bi) {
                const int c = cand_n[(b0 + bi) * 17 + t];
                total += c;
                if (c > cap) { tie_flags[im * 17 + t] = 1; break; }    // list truncated on the device: the caller's full-table path
                for (int k = 0; k < c && k < cap; ++k) {
                    const int id = cand_id[((long)(b0 + bi) * 17 + t) * cap + k];
                    int j = 0;
                    while (j < ncol && ids[j] != id) ++j;
                    if (j == ncol) { if (ncol >= 512) { tie_flags[im * 17 + t] = 1; break; } ids[ncol++] = id; }
                }
            }
            if (total == 0) { any_empty = true; continue; }
            if (tie_flags[im * 17 + t]) continue;
            // table[bi][col]
            double table[64 * 64];
            if (n > 64 || ncol > 64) { tie_flags[im * 17 + t] = 1; continue; }
            for (int k = 0; k < n * ncol; ++k) table[k] = 0.0;
            for (int bi = 0; bi < n; ++bi) {
                const int c = cand_n[(b0 + bi) * 17 + t];
                for (int k = 0; k < c; ++k) {
                    const long q = ((long)(b0 + bi) * 17 + t) * cap + k;
                    int j = 0;
                    while (ids[j] != cand_id[q]) ++j;
                    table[bi * ncol + j] = (double)cand_score[q];
                }
            }
            // any two equal positive entries in one row or one column could make the answer depend on argsort's tie order
            bool tie = false;
            for (int bi = 0; bi < n && !tie; ++bi)
                for (int a = 0; a < ncol && !tie; ++a)
                    for (int c2 = a + 1; c2 < ncol; ++c2)
                        if (table[bi * ncol + a] > 0.0 && table[bi * ncol + a] == table[bi * ncol + c2]) { tie = true; break; }
            for (int a = 0; a < ncol && !tie; ++a)
                for (int bi = 0; bi < n && !tie; ++bi)
                    for (int b2 = bi + 1; b2 < n; ++b2)
                        if (table[bi * ncol + a] > 0.0 && table[bi * ncol + a] == table[b2 * ncol + a]) { tie = true; break; }
            if (tie) { tie_flags[im * 17 + t] = 1; continue; }
            for (int bi = 0; bi < n; ++bi) {                       // tester.py:453-470
                bool used[64];
                for (int a = 0; a < ncol; ++a) used[a] = false;
                for (int step = 0; step < ncol; ++step) {          // columns of this row by descending score
                    int r = -1;
                    for (int a = 0; a < ncol; ++a)
                        if (!used[a] && (r < 0 || table[bi * ncol + a] > table[bi * ncol + r])) r = a;
                    used[r] = true;
                    if (!(table[bi * ncol + r] > 0.0)) break;      // the rest of the row is empty
                    int bestb = 0;                                  // the box that scores highest for this peak
                    for (int b2 = 1; b2 < n; ++b2) if (table[b2 * ncol + r] > table[bestb * ncol + r]) bestb = b2;
                    bool take = bestb == bi;
                    if (!take) {                                    // ... or this peak is that box's WORST column (np.argsort(table[best])[0] == r):
                        bool is_min = true;                         // impossible while that row holds an empty (zero) cell, since table[best][r] > 0
                        for (int a = 0; a < ncol; ++a)
                            if (a != r && table[bestb * ncol + a] <= table[bestb * ncol + r]) { is_min = false; break; }
                        take = is_min;
                    }
                    if (take) {
                        const int pid = first + ids[r];
                        double* o = out_kp + ((long)(b0 + bi) * 17 + t) * 3;
                        o[0] = peaks[(long)pid * 2]; o[1] = peaks[(long)pid * 2 + 1]; o[2] = 1.0;
                        break;
                    }
                }
            }
        }
        if (any_empty) {                                            // tester.py:471-483: joint types nobody has -> the PRN's arg-max cell
            for (int bi = 0; bi < n; ++bi) {
                const double* bx = boxes + (long)(b0 + bi) * 4;
                const double x_scale = (double)W / ceil(bx[2]), y_scale = (double)H / ceil(bx[3]);
                for (int t = 0; t < 17; ++t) {
                    if (cand_n[(b0 + bi) * 17 + t] != 0) continue;
                    const int am = argmax[(b0 + bi) * 17 + t];
                    const int my = am / W, mx = am - my * W;
                    double* o = out_kp + ((long)(b0 + bi) * 17 + t) * 3;
                    o[0] = (double)mx / x_scale + bx[0]; o[1] = (double)my / y_scale + bx[1]; o[2] = 0.0;
                }
            }
        }
    }
    return 0;
}

extern "C" int mpn_prn_scores_compact(const float* prn_out, const int32_t* occ, int nboxes, int H, int W, int N, int cap,
                                      int32_t* cand_n, int32_t* cand_id, float* cand_score, int32_t* argmax, int32_t* overflow, void* stream) {
    MPN_CHECK_ARG(prn_out && occ && cand_n && cand_id && cand_score && argmax && overflow && nboxes > 0 && H > 0 && W > 0 && cap > 0 &&
                  N > 0 && N <= 15 && (N & 1));
    hipLaunchKernelGGL(prn_scores_compact_kernel, dim3(17, nboxes), dim3(64), 0, (hipStream_t)stream, prn_out, occ, H, W, N, cap, cand_n,
                       cand_id, cand_score, argmax, overflow);
    return mpn_launch_status();
}

extern "C" int mpn_prn_build_maps(const double* peaks, const int32_t* joint_off, const double* boxes, const int32_t* box_img, int nboxes,
                                  int H, int W, double in_thres, const double* weights9, int32_t* occ, float* prn_in, int32_t* err,
                                  void* stream) {
    MPN_CHECK_ARG(peaks && joint_off && boxes && box_img && weights9 && occ && prn_in && err && nboxes > 0 && H > 0 && W > 0);
    if ((long)H * W > kMaxCells) return MPN_E_UNSUPPORTED;
    hipLaunchKernelGGL(prn_build_maps_kernel, dim3(17, nboxes), dim3(256), 0, (hipStream_t)stream, peaks, joint_off, boxes, box_img, H, W,
                       in_thres, weights9, occ, prn_in, err);
    return mpn_launch_status();
}

extern "C" int mpn_prn_scores(const float* prn_out, const int32_t* occ, int nboxes, int H, int W, int N, float* score, int32_t* argmax,
                              void* stream) {
    MPN_CHECK_ARG(prn_out && occ && score && argmax && nboxes > 0 && H > 0 && W > 0 && N > 0 && N <= 15 && (N & 1));
    hipLaunchKernelGGL(prn_scores_kernel, dim3(17, nboxes), dim3(64), 0, (hipStream_t)stream, prn_out, occ, H, W, N, score, argmax);
    return mpn_launch_status();
}
