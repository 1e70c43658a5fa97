/*
 * oracle/nms_oracle.c — CPU restatement of the reference's box NMS (TEST INFRASTRUCTURE).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call this.  The product
 * path (multiposenet/pytorch_amd/csrc) never links or loads it.
 *
 * The reference's native NMS (lib/nms/src: nms.c, nms_cuda.c, cuda/nms_kernel.cu) cannot be compiled here: it includes
 * <TH/TH.h>/<THC/THC.h> (gone from torch >= 1.0) and needs nvcc.  It is therefore restated:
 *
 *   oracle_nms_gpu  follows  lib/nms/src/cuda/nms_kernel.cu:16-24  (devIoU: +1 pixel convention)
 *                            lib/nms/src/cuda/nms_kernel.cu:53-68  (64-wide bit mask, strict '>',
 *                                                                   diagonal tile starts at t+1)
 *                            lib/nms/src/nms_cuda.c:47-58          (serial scan over the mask)
 *                            lib/nms/pth_nms.py:25-44              (areas, descending sort,
 *                                                                   order[keep] mapping)
 *   oracle_nms_cpu  follows  lib/nms/src/nms.c:35-63               (greedy, suppress on '>=')
 *                            lib/nms/pth_nms.py:9-24
 *
 * Sort: descending score, ties -> lower original index (pth_nms.py uses torch's unstable sort, so
 * the reference itself does not define tie order; fixtures avoid ties and this rule makes the
 * oracle and the HIP kernel agree when ties do occur).
 *
 * Build with -ffp-contract=off (see oracle/Makefile): every float op below is a separately rounded
 * IEEE binary32 operation in exactly the order of the reference source.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float key; int64_t idx; } sort_item;

static int cmp_desc(const void* pa, const void* pb) {
    const sort_item* a = (const sort_item*)pa;
    const sort_item* b = (const sort_item*)pb;
    if (a->key > b->key) return -1;
    if (a->key < b->key) return 1;
    if (a->idx < b->idx) return -1;
    if (a->idx > b->idx) return 1;
    return 0;
}

/* order[i] = original index of the i-th highest score */
void oracle_sort_desc(const float* dets, int64_t n, int64_t* order) {
    sort_item* it = (sort_item*)malloc(sizeof(sort_item) * (size_t)(n > 0 ? n : 1));
    for (int64_t i = 0; i < n; ++i) { it[i].key = dets[i * 5 + 4]; it[i].idx = i; }
    qsort(it, (size_t)n, sizeof(sort_item), cmp_desc);
    for (int64_t i = 0; i < n; ++i) order[i] = it[i].idx;
    free(it);
}

/* nms_kernel.cu:16-24 */
static float dev_iou(const float* a, const float* b) {
    float left = fmaxf(a[0], b[0]), right = fminf(a[2], b[2]);
    float top = fmaxf(a[1], b[1]), bottom = fminf(a[3], b[3]);
    float width = fmaxf(right - left + 1, 0.f), height = fmaxf(bottom - top + 1, 0.f);
    float interS = width * height;
    float Sa = (a[2] - a[0] + 1) * (a[3] - a[1] + 1);
    float Sb = (b[2] - b[0] + 1) * (b[3] - b[1] + 1);
    return interS / (Sa + Sb - interS);
}

/* GPU-path semantics.  dets [n,5] unsorted; keep_out receives ORIGINAL indices in descending-score
 * order; returns the count.  mask_words_out (optional, may be NULL) receives the n*ceil(n/64) u64
 * mask in sorted space exactly as nms_kernel.cu writes it (upper triangle incl. diagonal tile;
 * lower-triangle words are computed by the reference too and are reproduced here). */
int64_t oracle_nms_gpu(const float* dets, int64_t n, float thresh, int64_t* keep_out,
                       uint64_t* mask_words_out) {
    if (n <= 0) return 0;
    int64_t* order = (int64_t*)malloc(sizeof(int64_t) * (size_t)n);
    float* sorted = (float*)malloc(sizeof(float) * 5 * (size_t)n);
    oracle_sort_desc(dets, n, order);
    for (int64_t i = 0; i < n; ++i) memcpy(sorted + i * 5, dets + order[i] * 5, 5 * sizeof(float));
    const int64_t cb = (n + 63) / 64;
    uint64_t* mask = (uint64_t*)calloc((size_t)(n * cb), sizeof(uint64_t));
    for (int64_t i = 0; i < n; ++i) {
        const int64_t rb = i / 64, rt = i % 64;
        for (int64_t c = 0; c < cb; ++c) {
            const int64_t csz = (n - c * 64) < 64 ? (n - c * 64) : 64;
            uint64_t t = 0;
            int64_t start = (rb == c) ? rt + 1 : 0;          /* nms_kernel.cu:58-61 */
            for (int64_t k = start; k < csz; ++k)
                if (dev_iou(sorted + i * 5, sorted + (c * 64 + k) * 5) > thresh) t |= 1ULL << k;
            mask[i * cb + c] = t;
        }
    }
    uint64_t* remv = (uint64_t*)calloc((size_t)cb, sizeof(uint64_t));
    int64_t num = 0;
    for (int64_t i = 0; i < n; ++i) {                        /* nms_cuda.c:47-58 */
        const int64_t nb = i / 64, ib = i % 64;
        if (!(remv[nb] & (1ULL << ib))) {
            keep_out[num++] = order[i];                      /* pth_nms.py:44 order[keep] */
            const uint64_t* p = mask + i * cb;
            for (int64_t j = nb; j < cb; ++j) remv[j] |= p[j];
        }
    }
    if (mask_words_out) memcpy(mask_words_out, mask, sizeof(uint64_t) * (size_t)(n * cb));
    free(remv); free(mask); free(sorted); free(order);
    return num;
}

/* CPU-path semantics (nms.c:35-63): greedy in sorted order, suppress when ovr >= thresh. */
int64_t oracle_nms_cpu(const float* dets, int64_t n, float thresh, int64_t* keep_out) {
    if (n <= 0) return 0;
    int64_t* order = (int64_t*)malloc(sizeof(int64_t) * (size_t)n);
    float* areas = (float*)malloc(sizeof(float) * (size_t)n);
    unsigned char* sup = (unsigned char*)calloc((size_t)n, 1);
    oracle_sort_desc(dets, n, order);
    for (int64_t i = 0; i < n; ++i)                          /* pth_nms.py:16 */
        areas[i] = (dets[i * 5 + 2] - dets[i * 5 + 0] + 1) * (dets[i * 5 + 3] - dets[i * 5 + 1] + 1);
    int64_t num = 0;
    for (int64_t _i = 0; _i < n; ++_i) {
        const int64_t i = order[_i];
        if (sup[i]) continue;
        keep_out[num++] = i;
        const float ix1 = dets[i * 5], iy1 = dets[i * 5 + 1], ix2 = dets[i * 5 + 2], iy2 = dets[i * 5 + 3];
        const float iarea = areas[i];
        for (int64_t _j = _i + 1; _j < n; ++_j) {
            const int64_t j = order[_j];
            if (sup[j]) continue;
            float xx1 = fmaxf(ix1, dets[j * 5]);
            float yy1 = fmaxf(iy1, dets[j * 5 + 1]);
            float xx2 = fminf(ix2, dets[j * 5 + 2]);
            float yy2 = fminf(iy2, dets[j * 5 + 3]);
            float w = fmaxf(0.0f, xx2 - xx1 + 1);
            float h = fmaxf(0.0f, yy2 - yy1 + 1);
            float inter = w * h;
            float ovr = inter / (iarea + areas[j] - inter);
            if (ovr >= thresh) sup[j] = 1;
        }
    }
    free(sup); free(areas); free(order);
    return num;
}
