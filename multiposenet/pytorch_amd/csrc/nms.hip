// nms.hip — box decode/clip, score filter and wavefront-level greedy NMS for gfx950.
//
// Replaces lib/nms (the reference's only native code):
//   pth_nms            lib/nms/pth_nms.py:5-45        (areas, descending sort, order[keep])
//   gpu_nms            lib/nms/src/nms_cuda.c:17-67   (mask alloc, D2H of the mask, serial host scan)
//   nms_kernel/devIoU  lib/nms/src/cuda/nms_kernel.cu:16-70
//   cpu_nms            lib/nms/src/nms.c:4-69         (mode 1: '>=' comparison)
// and BBoxTransform / ClipBoxes (network/utils.py:19-61) + the score>0.05 gather
// (network/posenet.py:269-279).
//
// Design (CDNA4: one wave == 64 lanes == one u64 mask word):
//   1. rank sort: rank[i] = #{j : s_j > s_i or (s_j == s_i and j < i)} — O(N^2) compares, unique
//      ranks, deterministic tie rule (lower index first); boxes are scattered to sorted order.
//   2. mask: one wave per 64x64 tile of the UPPER triangle; lane t owns row box t, column boxes are
//      broadcast lane-to-lane (v_readlane) — no LDS; the reference also fills the lower triangle,
//      which its scan never reads.
//      Tile pairs are enumerated triangularly, so no workgroup is launched for the lower half.
//   3. scan ON DEVICE (the reference copies the whole N*N/64 mask to the host): one workgroup walks
//      the 64-box blocks; the diagonal word resolves intra-block suppression with scalar bit ops,
//      kept rows are OR-ed into the removal vector in parallel.  Only `keep`/`num` leave the GPU.
// IoU arithmetic is op-for-op devIoU (+1 pixel convention); this file is compiled with
// -ffp-contract=off so mask bits equal the CPU oracle's bit for bit.
#include "common.h"

namespace {

__global__ void box_decode_clip_kernel(const float* __restrict__ anchors, const float* __restrict__ deltas,
                                       float* __restrict__ boxes, int B, int A, float img_w, float img_h, float4 mean, float4 std) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * A) return;
    const int a = (int)(i % A);
    const float4 an = *reinterpret_cast<const float4*>(anchors + (long)a * 4);
    const float4 d = *reinterpret_cast<const float4*>(deltas + i * 4);
    // utils.py:21-41: deltas * std + mean (defaults mean 0, std .1 .1 .2 .2, utils.py:10-17)
    const float w = an.z - an.x, h = an.w - an.y;
    const float cx = an.x + 0.5f * w, cy = an.y + 0.5f * h;
    const float dx = d.x * std.x + mean.x, dy = d.y * std.y + mean.y, dw = d.z * std.z + mean.z, dh = d.w * std.w + mean.w;
    const float pcx = cx + dx * w, pcy = cy + dy * h;
    const float pw = expf(dw) * w, ph = expf(dh) * h;
    float x1 = pcx - 0.5f * pw, y1 = pcy - 0.5f * ph, x2 = pcx + 0.5f * pw, y2 = pcy + 0.5f * ph;
    // utils.py:55-59
    if (img_w >= 0.f) { x1 = fmaxf(x1, 0.f); y1 = fmaxf(y1, 0.f); x2 = fminf(x2, img_w); y2 = fminf(y2, img_h); }
    *reinterpret_cast<float4*>(boxes + i * 4) = make_float4(x1, y1, x2, y2);
}

__global__ void clip_boxes_kernel(float* __restrict__ boxes, long n, float img_w, float img_h) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 b = *reinterpret_cast<float4*>(boxes + i * 4);
    b.x = fmaxf(b.x, 0.f); b.y = fmaxf(b.y, 0.f); b.z = fminf(b.z, img_w); b.w = fminf(b.w, img_h);
    *reinterpret_cast<float4*>(boxes + i * 4) = b;
}

// one workgroup (1024 threads) per image, order-preserving compaction of the candidates with score > thresh
__global__ void __launch_bounds__(1024) score_filter_kernel(const float* __restrict__ boxes_all, const float* __restrict__ scores_all, int A,
                                                            float thresh, float* __restrict__ dets_all, int* __restrict__ src_all,
                                                            int* __restrict__ count) {
    __shared__ int wave_cnt[16];
    __shared__ int base_s;
    const int img = blockIdx.x;
    const float* __restrict__ boxes = boxes_all + (long)img * A * 4;
    const float* __restrict__ scores = scores_all + (long)img * A;
    float* __restrict__ dets = dets_all + (long)img * A * 5;
    int* __restrict__ src_idx = src_all ? src_all + (long)img * A : nullptr;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) base_s = 0;
    __syncthreads();
    for (int a0 = 0; a0 < A; a0 += 1024) {
        const int a = a0 + threadIdx.x;
        const bool pass = (a < A) && (scores[a] > thresh);
        const unsigned long long m = __ballot(pass);
        if (lane == 0) wave_cnt[wave] = __popcll(m);
        __syncthreads();
        int off = base_s;
        for (int w = 0; w < wave; ++w) off += wave_cnt[w];
        if (pass) {
            const int pos = off + __popcll(m & ((1ull << lane) - 1ull));
            const float4 b = *reinterpret_cast<const float4*>(boxes + (long)a * 4);
            float* o = dets + (long)pos * 5;
            o[0] = b.x; o[1] = b.y; o[2] = b.z; o[3] = b.w; o[4] = scores[a];
            if (src_idx) src_idx[pos] = a;
        }
        __syncthreads();
        if (threadIdx.x == 0) { int t = 0; for (int w = 0; w < 16; ++w) t += wave_cnt[w]; base_s += t; }
        __syncthreads();
    }
    if (threadIdx.x == 0) count[img] = base_s;
}

// ---------------------------------------------------------------------------------- NMS
// Every kernel below serves one image per blockIdx.z (or .x for the scan): image b has n = counts[b] candidates (counts ==
// NULL: n_host for the single image), its rows start at dets + b*dets_stride, its scratch at b*ws_stride bytes.
struct NmsBatch {
    const float* dets; long dets_stride;         // candidate rows [n][5] per image (floats between images)
    const int* counts; int n_host;
    char* ws; long ws_stride;                    // per image: sorted [nmax*5 f32] | order [nmax i32] | remv [cb u64] | mask (upper triangle, packed)
    long off_order, off_remv, off_mask;
    int cb;                                      // mask words per row = ceil(nmax / 64)
    int cap;                                     // 0, or: only the `cap` best-scored candidates of an image enter the suppression
};
// candidates of image b / the ones that take part in the suppression (the best `cap` of them after the sort)
__device__ __forceinline__ int nb_all(const NmsBatch& q, int b) { return q.counts ? q.counts[b] : q.n_host; }
__device__ __forceinline__ int nb_count(const NmsBatch& q, int b) {
    const int n = nb_all(q, b);
    return (q.cap > 0 && n > q.cap) ? q.cap : n;
}

// The suppression mask is stored as its UPPER TRIANGLE only (the lower half is never written by nms_mask_kernel nor read by the scan —
// nms_kernel.cu allocates it all the same, nms_cuda.c:28): row r (tile row rb = r / 64) keeps the words of column tiles rb .. cb - 1,
// rows of a tile row back to back.  Word (r, j >= rb) sits at nms_tri_row(r, cb) + (j - rb); 64 * cb (cb + 1) / 2 words per image.
__device__ __forceinline__ long nms_tri_row(int r, int cb) {
    const long rb = r >> 6;
    return 64 * (rb * cb - rb * (rb - 1) / 2) + (long)(r & 63) * (cb - rb);
}

__global__ void nms_rank_kernel(const NmsBatch q) {
    __shared__ float ssc[1024];
    const int b = blockIdx.z;
    const int n = nb_all(q, b);
    const int live = nb_count(q, b);
    if ((int)(blockIdx.x * blockDim.x) >= n) return;
    const float* __restrict__ dets = q.dets + (long)b * q.dets_stride;
    float* __restrict__ sorted = reinterpret_cast<float*>(q.ws + (long)b * q.ws_stride);
    int* __restrict__ order = reinterpret_cast<int*>(q.ws + (long)b * q.ws_stride + q.off_order);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const float si = (i < n) ? dets[(long)i * 5 + 4] : 0.f;
    int rank = 0;
    for (int j0 = 0; j0 < n; j0 += 1024) {
        for (int t = threadIdx.x; t < 1024; t += blockDim.x) ssc[t] = (j0 + t < n) ? dets[(long)(j0 + t) * 5 + 4] : 0.f;
        __syncthreads();
        const int lim = (n - j0) < 1024 ? (n - j0) : 1024;
        for (int t = 0; t < lim; ++t) {
            const float sj = ssc[t];
            rank += (sj > si) || (sj == si && (j0 + t) < i);
        }
        __syncthreads();
    }
    if (i < n && rank < live) {
        order[rank] = i;
        const float* s = dets + (long)i * 5;
        float* d = sorted + (long)rank * 5;
        d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; d[3] = s[3]; d[4] = s[4];
    }
}

__device__ __forceinline__ float dev_iou(const float a0, const float a1, const float a2, const float a3,
                                         const float b0, const float b1, const float b2, const float b3) {
    // nms_kernel.cu:16-24, op for op
    const float left = fmaxf(a0, b0), right = fminf(a2, b2);
    const float top = fmaxf(a1, b1), bottom = fminf(a3, b3);
    const float width = fmaxf(right - left + 1, 0.f), height = fmaxf(bottom - top + 1, 0.f);
    const float interS = width * height;
    const float Sa = (a2 - a0 + 1) * (a3 - a1 + 1);
    const float Sb = (b2 - b0 + 1) * (b3 - b1 + 1);
    return interS / (Sa + Sb - interS);
}

// One wave per 64x64 tile of the UPPER triangle (tile pairs are enumerated so that no workgroup is launched for the
// lower half): blockIdx.x = row_blk * cb - row_blk*(row_blk-1)/2 + (col_blk - row_blk).
__global__ void __launch_bounds__(64) nms_mask_kernel(const NmsBatch q, float thresh, int mode) {
    const int b = blockIdx.z;
    const int n = nb_count(q, b);
    const int cbn = (n + 63) / 64;               // live tiles per side for this image
    // invert the triangular index with the launch-wide cb
    const int cb = q.cb;
    int row_blk = (int)((2.0 * cb + 1.0 - sqrt((2.0 * cb + 1.0) * (2.0 * cb + 1.0) - 8.0 * (double)blockIdx.x)) * 0.5);
    while (row_blk > 0 && (long)row_blk * cb - (long)row_blk * (row_blk - 1) / 2 > (long)blockIdx.x) --row_blk;
    while ((long)(row_blk + 1) * cb - (long)(row_blk + 1) * row_blk / 2 <= (long)blockIdx.x) ++row_blk;
    const int col_blk = row_blk + (int)((long)blockIdx.x - ((long)row_blk * cb - (long)row_blk * (row_blk - 1) / 2));
    if (row_blk >= cbn || col_blk >= cbn) return;
    const float* __restrict__ sorted = reinterpret_cast<const float*>(q.ws + (long)b * q.ws_stride);
    unsigned long long* __restrict__ mask = reinterpret_cast<unsigned long long*>(q.ws + (long)b * q.ws_stride + q.off_mask);
    const int t = threadIdx.x;
    const int row = row_blk * 64 + t, col = col_blk * 64 + t;
    float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f, r0 = 0.f, r1 = 0.f, r2 = 0.f, r3 = 0.f;
    if (col < n) { const float* p = sorted + (long)col * 5; c0 = p[0]; c1 = p[1]; c2 = p[2]; c3 = p[3]; }
    if (row < n) { const float* p = sorted + (long)row * 5; r0 = p[0]; r1 = p[1]; r2 = p[2]; r3 = p[3]; }
    const int col_size = (n - col_blk * 64) < 64 ? (n - col_blk * 64) : 64;
    unsigned long long bits = 0ull;
    const int start = (row_blk == col_blk) ? t + 1 : 0;       // nms_kernel.cu:58-61
    for (int i = 0; i < col_size; ++i) {
        const float b0 = __shfl(c0, i, 64), b1 = __shfl(c1, i, 64), b2 = __shfl(c2, i, 64), b3 = __shfl(c3, i, 64);
        const float iou = dev_iou(r0, r1, r2, r3, b0, b1, b2, b3);
        const bool hit = (mode == 0) ? (iou > thresh) : (iou >= thresh);
        if (hit && i >= start) bits |= 1ull << i;
    }
    if (row < n) mask[nms_tri_row(row, cb) + (col_blk - row_blk)] = bits;
}

// one workgroup of 1024 threads per image
__global__ void __launch_bounds__(1024) nms_scan_kernel(const NmsBatch q, int64_t* __restrict__ keep_all, long keep_stride,
                                                        int64_t* __restrict__ num_out) {
    __shared__ unsigned long long keep_word;
    __shared__ int base_cnt;
    const int b = blockIdx.x;
    const int n = nb_count(q, b);
    const int cb = q.cb;
    const int cbn = (n + 63) / 64;
    const int* __restrict__ order = reinterpret_cast<const int*>(q.ws + (long)b * q.ws_stride + q.off_order);
    unsigned long long* __restrict__ remv = reinterpret_cast<unsigned long long*>(q.ws + (long)b * q.ws_stride + q.off_remv);
    const unsigned long long* __restrict__ mask = reinterpret_cast<const unsigned long long*>(q.ws + (long)b * q.ws_stride + q.off_mask);
    int64_t* __restrict__ keep_out = keep_all + (long)b * keep_stride;
    const int tid = threadIdx.x;
    for (int j = tid; j < cbn; j += 1024) remv[j] = 0ull;
    if (tid == 0) base_cnt = 0;
    __syncthreads();
    for (int bi = 0; bi < cbn; ++bi) {
        if (tid < 64) {
            const int row = bi * 64 + tid;
            const unsigned long long diag = (row < n) ? mask[nms_tri_row(row, cb)] : 0ull;
            unsigned long long cur = remv[bi];
            unsigned long long kept = 0ull;
            const int lim = (n - bi * 64) < 64 ? (n - bi * 64) : 64;
            for (int t = 0; t < lim; ++t) {
                const unsigned long long dt = __shfl(diag, t, 64);
                if (!((cur >> t) & 1ull)) { kept |= 1ull << t; cur |= dt; }
            }
            if ((kept >> tid) & 1ull) {
                const int pos = base_cnt + __popcll(kept & ((1ull << tid) - 1ull));
                keep_out[pos] = (int64_t)order[row];
            }
            if (tid == 0) keep_word = kept;
        }
        __syncthreads();
        const unsigned long long kept = keep_word;
        if (kept) {
            for (int j = bi + 1 + tid; j < cbn; j += 1024) {
                unsigned long long acc = remv[j];
                unsigned long long k = kept;
                while (k) {
                    const int t = __ffsll((long long)k) - 1;
                    k &= k - 1ull;
                    acc |= mask[nms_tri_row(bi * 64 + t, cb) + (j - bi)];
                }
                remv[j] = acc;
            }
        }
        __syncthreads();
        if (tid == 0) base_cnt += __popcll(kept);
        __syncthreads();
    }
    if (tid == 0) num_out[b] = (int64_t)base_cnt;
}

// rows of dets[n,5] selected by keep[k] -> boxes[k,4], scores[k]  (posenet.py:283-285 gathers); image = blockIdx.y,
// k = num[image] when num != NULL
__global__ void gather_dets_kernel(const float* __restrict__ dets_all, long dets_stride, const int64_t* __restrict__ keep_all, long keep_stride,
                                   const int64_t* __restrict__ num, int k_host, float* __restrict__ boxes_all, float* __restrict__ scores_all,
                                   long out_stride) {
    const int b = blockIdx.y;
    const int k = num ? (int)num[b] : k_host;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= k) return;
    const float* s = dets_all + (long)b * dets_stride + keep_all[(long)b * keep_stride + i] * 5;
    float* boxes = boxes_all + (long)b * out_stride * 4;
    boxes[i * 4 + 0] = s[0]; boxes[i * 4 + 1] = s[1]; boxes[i * 4 + 2] = s[2]; boxes[i * 4 + 3] = s[3];
    scores_all[(long)b * out_stride + i] = s[4];
}

inline long align_up(long v, long a) { return (v + a - 1) / a * a; }

}  // namespace

extern "C" int mpn_box_decode_clip(const float* anchors, const float* deltas, float* boxes, int B, int A, float img_w,
                                   float img_h, void* stream) {
    MPN_CHECK_ARG(anchors && deltas && boxes && B > 0 && A > 0);
    const long n = (long)B * A;
    hipLaunchKernelGGL(box_decode_clip_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, anchors, deltas,
                       boxes, B, A, img_w, img_h, make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.1f, 0.1f, 0.2f, 0.2f));
    return mpn_launch_status();
}

// BBoxTransform(mean, std) with non-default coefficients (utils.py:8-17): mean_std = {mean[0..3], std[0..3]} on the HOST
extern "C" int mpn_box_decode_clip_ms(const float* anchors, const float* deltas, float* boxes, int B, int A, float img_w,
                                      float img_h, const float* mean_std, void* stream) {
    MPN_CHECK_ARG(anchors && deltas && boxes && mean_std && B > 0 && A > 0);
    const long n = (long)B * A;
    hipLaunchKernelGGL(box_decode_clip_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, anchors, deltas,
                       boxes, B, A, img_w, img_h, make_float4(mean_std[0], mean_std[1], mean_std[2], mean_std[3]),
                       make_float4(mean_std[4], mean_std[5], mean_std[6], mean_std[7]));
    return mpn_launch_status();
}

extern "C" int mpn_clip_boxes(float* boxes, int64_t n, float img_w, float img_h, void* stream) {
    MPN_CHECK_ARG(boxes && n > 0);
    hipLaunchKernelGGL(clip_boxes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, boxes, (long)n, img_w, img_h);
    return mpn_launch_status();
}

extern "C" int mpn_score_filter(const float* boxes, const float* scores, int A, float thresh, float* dets, int32_t* src_idx,
                                int32_t* count, void* stream) {
    MPN_CHECK_ARG(boxes && scores && dets && src_idx && count && A > 0);
    hipLaunchKernelGGL(score_filter_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, boxes, scores, A, thresh, dets, src_idx, count);
    return mpn_launch_status();
}

extern "C" int mpn_score_filter_batched(const float* boxes, const float* scores, int B, int A, float thresh, float* dets,
                                        int32_t* src_idx, int32_t* counts, void* stream) {
    MPN_CHECK_ARG(boxes && scores && dets && counts && B > 0 && A > 0);
    hipLaunchKernelGGL(score_filter_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, boxes, scores, A, thresh, dets, src_idx, counts);
    return mpn_launch_status();
}

extern "C" int mpn_gather_dets(const float* dets, const int64_t* keep, int k, float* boxes, float* scores, void* stream) {
    MPN_CHECK_ARG(dets && keep && boxes && scores && k > 0);
    hipLaunchKernelGGL(gather_dets_kernel, dim3((k + 255) / 256, 1), dim3(256), 0, (hipStream_t)stream, dets, 0L, keep, 0L,
                       (const int64_t*)nullptr, k, boxes, scores, 0L);
    return mpn_launch_status();
}

extern "C" int mpn_gather_dets_batched(const float* dets, int64_t dets_stride, const int64_t* keep, int64_t keep_stride, const int64_t* num,
                                       int B, int kmax, float* boxes, float* scores, int64_t out_stride, void* stream) {
    MPN_CHECK_ARG(dets && keep && num && boxes && scores && B > 0 && kmax > 0 && out_stride >= kmax);
    hipLaunchKernelGGL(gather_dets_kernel, dim3((kmax + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, dets, (long)dets_stride, keep,
                       (long)keep_stride, num, 0, boxes, scores, (long)out_stride);
    return mpn_launch_status();
}

namespace {
struct NmsLayout { long off_order, off_remv, off_mask, total; int cb; };
inline NmsLayout nms_layout(long nmax) {
    NmsLayout l;
    l.cb = (int)((nmax + 63) / 64);
    l.off_order = align_up(nmax * 5 * 4, 256);
    l.off_remv = l.off_order + align_up(nmax * 4, 256);
    l.off_mask = l.off_remv + align_up((long)l.cb * 8, 256);
    l.total = l.off_mask + align_up(64L * ((long)l.cb * (l.cb + 1) / 2) * 8, 256);      // upper triangle only (nms_tri_row)
    return l;
}
inline int launch_nms(const float* dets, long dets_stride, const int* counts, int n_host, int B, long nmax, float thresh, int mode,
                      int64_t* keep_out, long keep_stride, int64_t* num_out, void* workspace, hipStream_t st, long top_k = 0) {
    const long nlive = (top_k > 0 && top_k < nmax) ? top_k : nmax;      // rows of the sorted / mask scratch
    const NmsLayout l = nms_layout(nlive);
    NmsBatch q;
    q.dets = dets; q.dets_stride = dets_stride; q.counts = counts; q.n_host = n_host;
    q.ws = (char*)workspace; q.ws_stride = l.total;
    q.off_order = l.off_order; q.off_remv = l.off_remv; q.off_mask = l.off_mask; q.cb = l.cb;
    q.cap = top_k > 0 ? (int)nlive : 0;
    const long tri = (long)l.cb * (l.cb + 1) / 2;
    if (tri > 0x7fffffffL) return MPN_E_UNSUPPORTED;
    hipLaunchKernelGGL(nms_rank_kernel, dim3((unsigned)((nmax + 255) / 256), 1, B), dim3(256), 0, st, q);
    hipLaunchKernelGGL(nms_mask_kernel, dim3((unsigned)tri, 1, B), dim3(64), 0, st, q, thresh, mode);
    hipLaunchKernelGGL(nms_scan_kernel, dim3(B), dim3(1024), 0, st, q, keep_out, keep_stride, num_out);
    return mpn_launch_status();
}
}  // namespace

extern "C" int64_t mpn_nms_workspace_bytes(int64_t n) {
    if (n <= 0) return 256;
    return nms_layout(n).total;
}

extern "C" int mpn_nms(const float* dets, int64_t n, float thresh, int mode, int64_t* keep_out, int64_t* num_out,
                       void* workspace, void* stream) {
    MPN_CHECK_ARG(num_out && (mode == 0 || mode == 1));
    hipStream_t st = (hipStream_t)stream;
    if (n <= 0) return (int)hipMemsetAsync(num_out, 0, sizeof(int64_t), st);
    MPN_CHECK_ARG(dets && keep_out && workspace && n < (1 << 30));
    return launch_nms(dets, 0, nullptr, (int)n, 1, (long)n, thresh, mode, keep_out, 0, num_out, workspace, st);
}

extern "C" int64_t mpn_nms_batched_workspace_bytes(int B, int64_t nmax) {
    if (B <= 0 || nmax <= 0) return 256;
    return (int64_t)B * nms_layout(nmax).total;
}

extern "C" int mpn_nms_batched(const float* dets, int64_t dets_stride, const int32_t* counts, int B, int64_t nmax, float thresh, int mode,
                               int64_t* keep_out, int64_t keep_stride, int64_t* num_out, void* workspace, void* stream) {
    MPN_CHECK_ARG(dets && counts && keep_out && num_out && workspace && B > 0 && (mode == 0 || mode == 1));
    MPN_CHECK_ARG(nmax > 0 && nmax < (1 << 30) && keep_stride >= nmax && dets_stride >= nmax * 5);
    return launch_nms(dets, (long)dets_stride, counts, 0, B, (long)nmax, thresh, mode, keep_out, (long)keep_stride, num_out, workspace,
                      (hipStream_t)stream);
}

// mpn_nms_batched with a cap ahead of the suppression: per image only the top_k best-scored candidates (ties by index, the
// order of the sort) take part; the rest are dropped as if they had not passed the score filter.  Not in the reference
// (posenet.py:269-285 hands every candidate above 0.05 to nms), off unless asked for: it bounds the N x N/64 mask of a dense
// image (A = 76 725 at 640x640: 736 MB) to top_k x top_k/64.  workspace: mpn_nms_batched_workspace_bytes(B, min(nmax, top_k)).
extern "C" int mpn_nms_batched_topk(const float* dets, int64_t dets_stride, const int32_t* counts, int B, int64_t nmax, int64_t top_k,
                                    float thresh, int mode, int64_t* keep_out, int64_t keep_stride, int64_t* num_out, void* workspace,
                                    void* stream) {
    MPN_CHECK_ARG(dets && counts && keep_out && num_out && workspace && B > 0 && (mode == 0 || mode == 1) && top_k > 0);
    const int64_t nlive = top_k < nmax ? top_k : nmax;
    MPN_CHECK_ARG(nmax > 0 && nmax < (1 << 30) && keep_stride >= nlive && dets_stride >= nmax * 5);
    return launch_nms(dets, (long)dets_stride, counts, 0, B, (long)nmax, thresh, mode, keep_out, (long)keep_stride, num_out, workspace,
                      (hipStream_t)stream, (long)top_k);
}
