// gt.hip — ground-truth heat-map generation on device.
//
// Replaces datasets/coco_data/heatmap.py:20-41 (putGaussianMaps) and the per-keypoint loop of
// datasets/coco_data/COCO_data_pipeline.py:218-236 for a whole batch: one thread per output cell (b, keypoint, y, x)
// walks that image's people in annotation order and does the reference's float64 arithmetic op for op
// (exponent = d2/2/sigma/sigma, cut at 4.6052, accumulate, clamp at 1.0), then rounds to float32 like
// COCO_data_pipeline.py:283.  Built with -ffp-contract=off so no multiply-add is fused that numpy would not fuse.
// Traffic: the keypoint table (B*maxP*18*3 doubles, read through L2) in, 4 bytes per cell out.
#include "common.h"

namespace {

__global__ void __launch_bounds__(256) gt_heatmaps_kernel(const double* __restrict__ joints, const int* __restrict__ num_people,
                                                          int maxP, float* __restrict__ out, int gh, int gw, double stride,
                                                          double sigma, long cells_per_image) {
    // grid = (cell blocks, 18 keypoint channels, B): a workgroup serves ONE (image, channel), so the keypoint walk
    // below is wave-uniform (scalar loads, one copy per wave instead of 36 vector loads per lane)
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int k = blockIdx.y, b = blockIdx.z;
    if (c >= gh * gw) return;
    const int x = c % gw, y = c / gw;
    const long i = (long)k * gh * gw + c;
    const double start = stride / 2.0 - 0.5;
    const double xx = (double)x * stride + start, yy = (double)y * stride + start;
    int np = num_people[b];
    if (np > maxP) np = maxP;
    const double* jp = joints + ((long)b * maxP * 18 + k) * 3;
    // cells farther than the Gaussian's support radius along one axis cannot pass the 4.6052 cut (d2 >= dx*dx and all
    // later operations are monotone); the 1e-6 relative margin dwarfs any rounding, so the prefilter never changes a
    // result — it only skips the three float64 divisions and the exp for the ~97 % of (cell, person) pairs out of range
    const double rmax = sqrt(2.0 * 4.6052) * sigma * (1.0 + 1e-6);
    double acc = 0.0;
    for (int j = 0; j < np; ++j, jp += 18 * 3) {
        if (!(jp[2] <= 1.0)) continue;
        const double dx = xx - jp[0], dy = yy - jp[1];
        if (fabs(dx) > rmax || fabs(dy) > rmax) continue;
        const double d2 = dx * dx + dy * dy;
        const double e = d2 / 2.0 / sigma / sigma;
        if (e <= 4.6052) {
            acc += exp(-e);
            if (acc > 1.0) acc = 1.0;
        }
    }
    out[(long)b * cells_per_image + i] = (float)acc;
}

}  // namespace

extern "C" int mpn_gt_heatmaps(const double* joints, const int32_t* num_people, int B, int maxP, float* out, int gh, int gw,
                               double stride, double sigma, void* stream) {
    MPN_CHECK_ARG(joints && num_people && out && B > 0 && maxP > 0 && gh > 0 && gw > 0 && stride > 0.0 && sigma > 0.0);
    const long cells = 18L * gh * gw;
    dim3 grid((unsigned)(((long)gh * gw + 255) / 256), 18u, (unsigned)B);
    hipLaunchKernelGGL(gt_heatmaps_kernel, grid, dim3(256), 0, (hipStream_t)stream, joints, (const int*)num_people, maxP, out, gh, gw,
                       stride, sigma, cells);
    return mpn_launch_status();
}
