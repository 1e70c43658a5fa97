/*
 * mpn.h — C ABI of libmpn_hip.so: the MI355X (gfx950) kernels behind the MultiPoseNet hot path.
 *
 * Boundary rules: extern "C", plain pointers + sizes, no torch types.  Every entry
 * point enqueues work on the caller's stream (`hipStream_t` passed as void*), never allocates,
 * never synchronises (except where stated), and returns 0 on success or the hipError_t code /
 * a negative MPN_E* code on a bad argument.  All tensors are device pointers unless stated.
 *
 * Tensor model: activations are NHWC ("pixel-major"): element (b, h, w, c) of a tensor lives at
 *     base + b*sB + (h*W + w)*sP + c            (strides in ELEMENTS, channel stride 1)
 * Internal tensors keep their channel count padded to a multiple of 32 (pad lanes hold zeros) so
 * every 64-byte K-chunk load is in bounds and 16-byte aligned.  dtype codes: 0 = f32, 1 = bf16, 2 = f16.
 *
 * Each entry point cites the reference interface it replaces (paths relative to the reference).
 */
#ifndef MPN_H_
#define MPN_H_
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPN_F32 0
#define MPN_BF16 1
#define MPN_F16 2      /* IEEE half: same kernels, v_mfma_f32_16x16x32_f16 (BASELINE config 5 inference arithmetic) */

#define MPN_E_BADARG (-2)
#define MPN_E_UNSUPPORTED (-3)

/* ---------------------------------------------------------------------------------------------
 * Convolution (implicit GEMM on MFMA).  Replaces every nn.Conv2d / nn.Linear call site of
 * network/fpn.py:14-26,42-76, network/posenet.py:36-46,78-89,133-135,165-186 (forward) and the
 * autograd backward torch derives for them (training/trainer.py:251).
 * -------------------------------------------------------------------------------------------*/
typedef struct MpnConvParams {
    const void* x;        /* gathered operand: fwd = input activations, dgrad = dY                */
    const void* w;        /* fwd: [Cout][R][S][Cin]; dgrad: [Cout'][R][S][Cin'] = Wt (see below)    */
    void* y;              /* output tensor                                                        */
    const float* bias;    /* optional [Cout] f32                                                  */
    const float* scale;   /* optional [Cout] f32 per-channel multiplier applied before bias       */
    const void* res;      /* optional residual (same element type as y)                           */
    const uint8_t* res_mask; /* optional, res_mode 1 only: the residual is multiplied by these mask bits (layout of bnb_mask /
                              * mpn_bn_act_forward's mask, dense [pixels][Cout_store / V]) before it is added — the gradient of
                              * relu(bn3(.) + shortcut) w.r.t. the shortcut is dz * (z > 0): the dgrad launch that completes the
                              * shortcut's gradient reads dz and the bits instead of a materialised copy (network/fpn.py:30-33)  */
    float* stats;         /* optional [tilesP][Cout][2] per-tile (sum, sumsq) partials (BN train) */
    int64_t x_sB, x_sH, x_sW;
    int64_t y_sB, y_sP;
    int64_t res_sB, res_sP;
    int32_t B, H, W, Cin; /* source dims; Cin = contraction channels per tap (mult. of 64 bytes)  */
    int32_t Ho, Wo, Cout, Cout_store; /* output dims; channels [Cout, Cout_store) are written 0   */
    int32_t R, S, stride, pad;
    int32_t mode;         /* 0: hi = ho*stride - pad + r ; 1 (dgrad): hi = (ho + pad - r)/stride   */
    int32_t act;          /* 0 none, 1 relu, 2 sigmoid (both before the residual stage), 3 relu AFTER a same-size
                           * residual add: with scale/bias = folded BatchNorm this is a whole Bottleneck tail
                           * relu(bn3(conv3(x)) + shortcut) (network/fpn.py:30-33) in one launch (frozen statistics) */
    int32_t res_mode;     /* 0 none, 1 same size, 2 nearest-upsampled from [res_H, res_W]          */
    int32_t res_H, res_W;
    int32_t accumulate;   /* y = y + result (act must be 0)                                       */
    int32_t dtype;        /* element type of x, w                                                 */
    int32_t out_f32;      /* 1: y (and res) are f32 even when dtype is bf16                       */
    /* Pyramid mode (nseg > 0): ONE launch applies the same weights to nseg <= MPN_MAX_SEG feature maps of different sizes —
     * the shared RetinaNet towers of network/posenet.py:33-117,327-328, which the reference runs level by level.  Level l
     * has its own dense tensors seg_x[l] / seg_y[l] ([B][seg_H][seg_W][channels], same channel strides x_sW / y_sP as the
     * single-tensor fields; stride 1, output size = input size) and owns the pixel tiles [seg_tile0[l], seg_tile0[l+1]);
     * a workgroup never straddles two levels.  x, y, H, W, Ho, Wo, x_sB, x_sH, y_sB are ignored in this mode.           */
    int32_t nseg;
    int32_t seg_H[5], seg_W[5];
    int32_t seg_tile0[6];
    const void* seg_x[5];
    void* seg_y[5];
    /* BatchNorm-backward statistics in the epilogue (bnb_partial != NULL; dgrad launches): the tensor this launch completes
     * is dz, the gradient w.r.t. the OUTPUT of a BatchNorm(+ReLU) layer z = act(bn(y_bn) [+ residual]) (network/fpn.py:28-34).
     * With g = dz * (z > 0) the launch also writes, per pixel tile, (sum g, sum g * xhat), xhat = (y_bn - mean) * invstd —
     * exactly what mpn_bn_bwd_reduce would produce in an extra pass over dz, y_bn (and z): bnb_partial [tiles][Cout][2] with
     * tiles = mpn_conv_stats_tiles(), consumed by mpn_bn_bwd_finalize(partial, tiles, ...).  bnb_y / bnb_z have the geometry,
     * element type and strides of y; bnb_z == NULL with bnb_relu: the mask is recomputed as y_bn*bnb_scale + bnb_shift > 0 (valid
     * when the forward had no residual input).  Statistics use the values as stored (after rounding to the element type).    */
    const void* bnb_y;
    const void* bnb_z;
    const uint8_t* bnb_mask; /* alternative to bnb_z: the ReLU mask as bits written by mpn_bn_act_forward(mask=...) — one byte per
                              * 16-byte chunk of z (8 channels of a 16-bit type, 4 of f32), bit k = (z[chunk*V + k] > 0), dense
                              * [pixels][Cout_store / V]; 1/16 of z's bytes.  Takes precedence over bnb_z.                       */
    const float* bnb_mean;
    const float* bnb_invstd;
    const float* bnb_scale;
    const float* bnb_shift;
    float* bnb_partial;
    int32_t bnb_relu;
    /* In-launch finalize (fin_counters != NULL, together with `stats` or `bnb_partial`): the LAST workgroup to finish a tile
     * of output channels reduces that tile's per-pixel-tile partials (fixed order, double precision — deterministic whatever
     * the arrival order) and does the work of mpn_bn_finalize_train (stats: fin_out = [4][Cout] mean, invstd, scale, shift;
     * running statistics updated when fin_rm / fin_rv are given) or of mpn_bn_bwd_finalize (bnb_partial: fin_dgamma += ,
     * fin_dbeta += , fin_out = [3][Cout] k1, k2, k3 with fin_train selecting batch-statistics or frozen coefficients; mean /
     * invstd are bnb_mean / bnb_invstd).  fin_counters: one zeroed uint32 per output-channel tile (64 entries); the launch leaves
     * them zero again.  Launches sharing a counter array must be ordered (one stream).           */
    /* Virtual channel concatenation of the gathered operand (kseg_n > 0; 3x3 / stride 1 / pad 1 launches with 16-bit operands):
     * the input is cat_s(nearest_upsample(kseg_x[s])) over kseg_n <= 4 segments of kseg_c channels each (Cin = kseg_n * kseg_c) —
     * torch.cat((up8(q5), up4(q4), up2(q3), q2), 1) feeding conv2 (network/posenet.py:311-315) — and is never materialised:
     * segment s is the dense tensor [B][H >> kseg_shift[s]][W >> kseg_shift[s]][kseg_c] and pixel (h, w) of the virtual input reads
     * its pixel (h >> shift, w >> shift).  H, W are the virtual (output) size; x, x_sB, x_sH, x_sW are ignored.            */
    int32_t kseg_n, kseg_c;
    int32_t kseg_shift[4];
    const void* kseg_x[4];
    uint32_t* fin_counters;
    const float* fin_gamma;
    const float* fin_beta;
    float* fin_rm;
    float* fin_rv;
    float* fin_out;
    float* fin_dgamma;
    float* fin_dbeta;
    double fin_count;
    float fin_momentum, fin_eps;
    int32_t fin_train;
    /* Parity classes of a stride-2 input gradient (y_step == 2; round 4).  dx[h][w] of a 3x3 / stride 2 / pad 1 convolution only receives
     * the taps r = (h + 1) mod 2 (+ 2), s likewise: of the nine taps 1, 2, 2 or 4 are live, depending on the parities (a, c) of (h, w) — the
     * gather form (mode 1, stride 2) multiplies the other 6.75 of 9 with zeros.  One launch per class computes
     *   dx[b][2 i + a][2 j + c] = sum_{t_r <= a, t_s <= c} dy[b][i + t_r][j + t_s] . W[a + 1 - 2 t_r][c + 1 - 2 t_s]
     * as a FORWARD gather (mode 0, stride 1, pad 0, R = 1 + a, S = 1 + c) over dy whose output pixel (i, j) is stored at
     * (y_step i + y_oh, y_step j + y_ow) of the dense [B][y_H][y_W] tensor y (and of every output-shaped operand: accumulate, bnb_y,
     * bnb_mask), and whose filter tap (t_r, t_s) is tap wtap0 + t_r wtap_dr + t_s wtap_ds of a weight tensor with w_taps taps per output
     * channel ([Cout][w_taps][Cin]).  Ho, Wo = the class grid (ceil((y_H - a) / 2), ceil((y_W - c) / 2)); rows of dy past its end read zeros.
     * Extended epilogue; not combined with nseg, kseg_n, res, fin_counters, stats, out_f32.  (torch.nn.Conv2d(stride=2) backward:
     * network/layers.py, the first Bottleneck of layer2-4; posenet.py P6 / P7.)                                                        */
    int32_t y_step, y_oh, y_ow, y_H, y_W;
    int32_t w_taps, wtap0, wtap_dr, wtap_ds;
} MpnConvParams;
#define MPN_MAX_SEG 5

/* number of pixel tiles (rows of `stats`) mpn_conv_forward will use for this problem */
int mpn_conv_stats_tiles(const MpnConvParams* p);
/* output-channel rows of the tile (256 / 128 / 64 / 32) the launcher will pick: names the kernel instantiation */
int mpn_conv_tile_rows(const MpnConvParams* p);
/* 1 when the launcher will take conv_igemm_s3_kernel (3x3, stride 1, pad 1, 16-bit operands, dense input: the pixel tile of a kernel
 * row lands once and serves its three taps), 0 for conv_igemm_kernel: names the kernel instantiation */
int mpn_conv_shared_tile(const MpnConvParams* p);
int mpn_conv_forward(const MpnConvParams* p, void* stream);

typedef struct MpnWgradParams {
    const void* x;        /* forward input activations (gathered)                                 */
    const void* dy;       /* output gradient, dense pixel-major: dy[p*dy_sP + cout]               */
    float* dw;            /* [Cout][R][S][Cin] f32, ACCUMULATED into (dw += result)               */
    float* ws;            /* workspace, >= chunks * Cout*R*S*Cin floats when chunks > 1           */
    int64_t x_sB, x_sH, x_sW;
    int64_t dy_sP;
    int32_t B, H, W, Cin;
    int32_t Ho, Wo, Cout;
    int32_t R, S, stride, pad;
    int32_t dtype;
    int32_t chunks;       /* split of the B*Ho*Wo contraction; use mpn_conv_wgrad_chunks()        */
    float* db;            /* optional bias gradient [Cout] f32, ACCUMULATED into: sum over pixels of dy.  Served
                           * only by the bf16 LDS-DMA kernel (mpn_conv_wgrad_kernel_id & 1), where it costs one
                           * extra MFMA against a vector of ones per dY fragment; otherwise must be NULL          */
    float* db_ws;         /* workspace, >= chunks * Cout floats, when db != NULL and chunks > 1                    */
    /* Pyramid mode (nseg > 0, LDS-DMA kernel only): the contraction runs over the pixels of nseg feature maps that share the
     * weights (see MpnConvParams); level l = dense tensors seg_x[l] / seg_dy[l] of size [B][seg_H][seg_W][.] (stride 1, same
     * size in and out) and owns the pixel slices [seg_chunk0[l], seg_chunk0[l+1]) of seg_chunk_pixels pixels each.  Fill
     * chunks / seg_chunk0 / seg_chunk_pixels with mpn_conv_wgrad_seg_plan().                                              */
    int32_t nseg;
    int32_t seg_H[5], seg_W[5];
    int32_t seg_chunk0[6];
    int32_t seg_chunk_pixels;
    const void* seg_x[5];
    const void* seg_dy[5];
    /* Virtual channel concatenation of x (see MpnConvParams.kseg_*): LDS-DMA kernel with 128-channel cin tiles, kseg_c == 128 —
     * every workgroup's cin tile is one segment.  H, W are the virtual size; x, x_sB, x_sH, x_sW are ignored.              */
    int32_t kseg_n, kseg_c;
    int32_t kseg_shift[4];
    const void* kseg_x[4];
} MpnWgradParams;
/* tools/kloop_profile.py only: device buffer [workgroups][4 waves][8] of uint64 that the following 128 x 128-tile bf16 weight-gradient
 * launches fill with per-wave s_memtime cycle sums of their k-loop phases (own-DMA wait, barrier wait, DMA issue, fragment reads +
 * MFMA issue, whole loop, epilogue, k-steps, start stamp); NULL = off (production kernels contain none of this) */
int mpn_debug_wgrad_prof(void* buf);
/* the same for the 128-row bf16 forward / input-gradient launches (conv_igemm_kernel, conv_igemm_s3_kernel); synchronises the device */
int mpn_debug_igemm_prof(void* buf);

int mpn_conv_wgrad_chunks(const MpnWgradParams* p);
/* pyramid mode: chooses the slice length for the summed pixel count, writes p->chunks, p->seg_chunk0, p->seg_chunk_pixels;
 * returns the number of slices (or a negative error) */
int mpn_conv_wgrad_seg_plan(MpnWgradParams* p);
int mpn_conv_wgrad(const MpnWgradParams* p, void* stream);
/* first stage only (chunks > 1): the per-slice partial gradients go to ws and the caller finishes with
 * mpn_reduce_partials(ws, chunks, Cout*R*S*Cin, dw, 1, stream) — lets a profiler bracket the MFMA kernel alone */
int mpn_conv_wgrad_partials(const MpnWgradParams* p, void* stream);
/* which kernel mpn_conv_wgrad launches for p: (tile_cin << 16) | (tile_cout << 4) | linear_x_addressing << 1 | uses_lds_dma
 * (bit 1: the instantiation for stride-1 same-extent convolutions over a dense x — halo predicate only, no per-lane address arithmetic) */
int mpn_conv_wgrad_kernel_id(const MpnWgradParams* p);

/* dst[i] (+)= sum_{c<chunks} ws[c*n + i]  — deterministic second stage of split reductions */
int mpn_reduce_partials(const float* ws, int chunks, int64_t n, float* dst, int accumulate, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Parameter preparation (once per step): master f32 [Cout][R][S][Cin] -> compute-dtype copies.
 * -------------------------------------------------------------------------------------------*/
int mpn_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream);
/* same for either 16-bit type (dtype = MPN_BF16 or MPN_F16) */
int mpn_cast_f32(const float* src, void* dst, int64_t n, int dtype, void* stream);
/* Wt[ci][r][s][co_pad] = W[co][r][s][ci] (co >= Cout -> 0); ci_pad rows beyond Cin are zero too */
int mpn_weight_transpose(const float* w, void* wt, int Cout, int RS, int Cin, int Cout_pad,
                         int dtype, void* stream);
/* all layers at once.  table[l] = {src offset in `arena` (floats), dst offset in `dst` (elements), Cout, RS, Cin, Cout_pad,
 * first block of the layer, ceil(Cin/32)} as int64; nblocks = sum over layers of ceil(Cin/32)*ceil(Cout_pad/32)*RS */
int mpn_weight_transpose_batched(const float* arena, void* dst, const int64_t* table, int nlayers, int64_t nblocks,
                                 int dtype, void* stream);
/* copy f32 [Cout][K] -> dtype [Cout][Kpad] zero padded (used for narrow-Cin / linear layers) */
int mpn_weight_pad_k(const float* w, void* dst, int Cout, int K, int Kpad, int dtype, void* stream);
/* stem 7x7x3: W[64][7][7][3] f32 <-> packed [64][7][32] (slot s*4+c, c<3, s<7; rest zero) */
int mpn_stem_pack_weight(const float* w, void* packed, int Cout, int dtype, void* stream);
int mpn_stem_unpack_wgrad(const float* dpacked, float* dw, int Cout, void* stream);
/* image NCHW f32 [B,3,H,W] -> zero-bordered NHWC4 [B][H+6][W+8][4] (pad 3 top/left) */
int mpn_stem_pack_image(const float* img, int64_t sB, int64_t sC, int64_t sH, int64_t sW, void* dst,
                        int B, int H, int W, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * BatchNorm2d (network/fpn.py:15-26,43) in train / eval mode, fused with ReLU and residual add
 * (Bottleneck.forward, fpn.py:28-34).
 * -------------------------------------------------------------------------------------------*/
/* train: reduce conv-epilogue partials -> mean/invstd, scale/shift, update running stats */
int mpn_bn_finalize_train(const float* stats, int tiles, int C, int64_t count, const float* gamma,
                          const float* beta, float* running_mean, float* running_var, float momentum,
                          float eps, float* mean, float* invstd, float* scale, float* shift, void* stream);
/* eval / frozen: scale/shift from running statistics */
int mpn_bn_finalize_eval(int C, const float* gamma, const float* beta, const float* running_mean,
                         const float* running_var, float eps, float* mean, float* invstd,
                         float* scale, float* shift, void* stream);
/* z = act(y*scale + shift [+ res]);  all [P][Cs] dense pixel-major with pixel stride Cs.  mask (optional, with relu): the sign
 * bits of the STORED z, one byte per 16-byte chunk ([P][Cs / V], V = 8 for 16-bit types, 4 for f32; bit k = z[chunk*V + k] > 0) —
 * what the backward of relu(bn(.) + shortcut) (network/fpn.py:30-33) needs of z, at 1/16 of its bytes. */
int mpn_bn_act_forward(const void* y, const void* res, void* z, const float* scale, const float* shift,
                       int64_t P, int C, int Cs, int relu, int dtype, uint8_t* mask, void* stream);
/* backward, stage 1: g = dz * (z > 0 if relu); partial sums of g and g*xhat per channel.
 * With relu and z == NULL the mask is recomputed as (y*mask_scale + mask_shift) > 0 — the forward's own expression
 * (valid when the forward had no residual input), which saves reading z. */
int mpn_bn_bwd_reduce(const void* dz, const void* z, const void* y, const float* mean, const float* invstd,
                      const float* mask_scale, const float* mask_shift, float* partial, int chunks, int64_t P, int C, int Cs, int relu, int dtype, void* stream);
/* backward, stage 2: reduce partials; dgamma += , dbeta += (if non-null); coef [3][C] = k1,k2,k3 with
 * dy = k1*g + k2*y + k3  (train: full batch-stat backward; train=0: k1 = gamma*invstd, k2 = k3 = 0) */
int mpn_bn_bwd_finalize(const float* partial, int chunks, int C, int64_t count, const float* gamma, const float* mean,
                        const float* invstd, int train, float* dgamma, float* dbeta, float* coef, void* stream);
/* backward, stage 3: dy = k1*g + k2*y + k3 (k2/k3 may be NULL = 0); dres (optional) receives / accumulates g.  With relu the mask
 * comes from `mask` bits (mpn_bn_act_forward) when given, else from z, else it is recomputed from y (mask_scale / mask_shift). */
int mpn_bn_bwd_apply(const void* dz, const void* z, const void* y, const float* k1, const float* k2, const float* k3,
                     const float* mask_scale, const float* mask_shift, void* dy, void* dres, int dres_accumulate, int64_t P, int C, int Cs, int relu, int dtype,
                     const uint8_t* mask, void* stream);
int mpn_bn_bwd_chunks(int64_t P, int Cs, int dtype);

/* ---------------------------------------------------------------------------------------------
 * Pooling / resampling / layout (fpn.py:84-100, posenet.py:180-184,296-315)
 * -------------------------------------------------------------------------------------------*/
int mpn_maxpool3x3s2_forward(const void* x, void* y, uint8_t* idx, int B, int H, int W, int Cs,
                             int Ho, int Wo, int dtype, void* stream);
int mpn_maxpool3x3s2_backward(const void* dy, const uint8_t* idx, void* dx, int B, int H, int W, int Cs,
                              int Ho, int Wo, int dtype, void* stream);
/* dcoarse[b,h,w,:] (+)= sum over the fine pixels whose nearest source is (h,w) of dfine */
int mpn_upsample_nearest_backward(const void* dfine, void* dcoarse, int B, int Hf, int Wf, int Hc, int Wc,
                                  int Cs, int accumulate, int dtype, void* stream);
/* out[b, oh, ow, c_off + c] = src[b, oh*Hs/Ho, ow*Ws/Wo, c]  (nearest; writes a channel slice) */
int mpn_upsample_nearest_slice(const void* src, void* dst, int B, int Hs, int Ws, int Cs_src,
                               int Ho, int Wo, int Cs_dst, int c_off, int dtype, void* stream);
/* dsrc[b,h,w,c] = sum over fine pixels of ddst[b,oh,ow,c_off+c] */
int mpn_upsample_nearest_slice_backward(const void* ddst, void* dsrc, int B, int Hs, int Ws, int Cs_src,
                                        int Ho, int Wo, int Cs_dst, int c_off, int dtype, void* stream);
/* API edge: padded internal f32/bf16 [B,h,w,Cs] -> exact f32 [B,H,W,C] (nearest up by H/h) and back */
int mpn_export_f32(const void* src, int src_dtype, float* dst, int B, int Hs, int Ws, int Cs, int C,
                   int Ho, int Wo, int64_t dst_sB, int64_t dst_sP, void* stream);
int mpn_import_grad(const float* ddst, int64_t ddst_sB, int64_t ddst_sP, void* dsrc, int dst_dtype,
                    int B, int Hs, int Ws, int Cs, int C, int Ho, int Wo, void* stream);
/* generic strided f32 NCHW -> dense NHWC f32 [B,H,W,C] (ground-truth maps) */
int mpn_nchw_to_nhwc_f32(const float* src, int64_t sB, int64_t sC, int64_t sH, int64_t sW, float* dst,
                         int B, int C, int H, int W, void* stream);
/* detection-head edge: padded internal [B,HW,Cs] <-> dense f32 [B, HW*C] slice of the [B,A,C/9] output
 * (the permute+view+cat of network/posenet.py:67-69,111-117,327-328 without any copy kernels in between) */
int mpn_det_pack(const void* src, int src_dtype, float* dst, int B, int64_t HW, int Cs, int C, int64_t dst_sB, void* stream);
int mpn_det_unpack(const float* ddst, void* dsrc, int dst_dtype, int B, int64_t HW, int Cs, int C, int64_t dst_sB, void* stream);
int mpn_relu_forward(const void* x, void* y, int64_t n, int dtype, void* stream);
int mpn_relu_backward(const void* dz, const void* z, void* dx, int64_t n, int accumulate, int dtype, void* stream);
int mpn_add_inplace(void* dst, const void* src, int64_t n, int dtype, void* stream);
int mpn_channel_sum(const void* dy, int dy_dtype, int64_t P, int C, int Cs, float* partial, int chunks, void* stream);
int mpn_channel_sum_chunks(int64_t P, int Cs, int dtype);
/* db[c] += sum over the P rows of dy[p][c] — small P, any C (Linear-layer bias gradients) */
int mpn_colsum_rows(const void* dy, int dy_dtype, int64_t P, int C, int Cs, float* db, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Losses (posenet.py:367-445, losses.py:5-137)
 * -------------------------------------------------------------------------------------------*/
/* preds: 5 device pointers (f32 NHWC [B,h,w,*] with pixel stride pred_sP[j]); gt/wgt dense NHWC f32
 * [B,h,w,18].  out[0..4] = per-level means, out[5] = total, out[6] = max_ht, out[7] = min_ht. */
int mpn_mse_heatmap_forward(const float* const* preds, const int64_t* pred_sP, const float* gt, const float* wgt,
                            int64_t npix, float* partial, int chunks, float* out, void* stream);
/* dpred_j[p, c] = gscale * 2*w^2*(pred - gt)/N for c < 18, 0 for c in [18, Cj) */
int mpn_mse_heatmap_backward(const float* const* preds, float* const* dpreds, const int64_t* pred_sP,
                             const int64_t* dpred_sP, const int32_t* pred_C, const float* gt, const float* wgt,
                             int64_t npix, const float* gscale, void* stream);
int mpn_mse_chunks(int64_t npix);
/* The recorded training step's form of the two calls above (replay.py): loss and gradients in ONE pass over the network's
 * internal tensors, with no exported f32 copies.  levels[0..3] = the intermediate maps k2..k5 (f32, [B, H>>s, W>>s, 32], s = 0..3:
 * posenet.py:243-257 up-samples them by nearest neighbours to the heat-map size, so pixel (y, x) reads level s at (y>>s, x>>s) and
 * a coarse cell's gradient is the sum over its children), levels[4] = the final prediction at [B, H, W, 32]; dlevels[j] receives
 * d(total)/d(levels[j]) in `dtype` with the padding channels zero.  gt / wgt are the reference's NCHW f32 targets read in place
 * (strides g_sB, g_sC, g_sH in elements, unit stride along x).  H and W must be multiples of 8.  out as for
 * mpn_mse_heatmap_forward; partial needs blocks * 8 floats, blocks = mpn_mse_train_blocks(B, H, W).  The gradients are
 * bit-identical to mpn_mse_heatmap_backward + mpn_import_grad. */
int mpn_mse_train_blocks(int B, int H, int W);
int mpn_mse_heatmap_train(const float* const* levels, void* const* dlevels, const int32_t* level_Cs, int dtype,
                          const float* gt, const float* wgt, int64_t g_sB, int64_t g_sC, int64_t g_sH,
                          int B, int H, int W, const float* gscale, float* partial, int blocks, float* out, void* stream);
/* focal + smooth-L1.  cls [B,A] f32 (post-sigmoid), reg [B,A,4], anchors [A,4], anno [B,maxN,5].
 * out[0] = cls loss, out[1] = reg loss (batch means).  per_img [B][4] scratch. */
int mpn_focal_blocks(int A);   /* partial needs B * mpn_focal_blocks(A) * 4 floats */
int mpn_focal_forward(const float* cls, const float* reg, const float* anchors, const float* anno,
                      int B, int A, int maxN, float* partial, float* per_img, float* out, void* stream);
/* gscale: device float[2] = upstream gradients of {cls loss, reg loss} */
int mpn_focal_backward(const float* cls, const float* reg, const float* anchors, const float* anno,
                       int B, int A, int maxN, const float* per_img, const float* gscale,
                       float* dcls, float* dreg, void* stream);
int mpn_sigmoid_forward(const float* x, float* y, int64_t n, void* stream);   /* in place allowed */
/* dlogit = dp * p * (1-p) */
int mpn_sigmoid_backward(const float* dp, const float* p, float* dlogit, int64_t n, void* stream);
/* PRN: out = softmax(relu?(a) + res) rowwise (posenet.py:345-347); BCE mean (posenet.py:436-439) */
int mpn_add_softmax_rows(const float* a, int64_t a_stride, const float* res, float* out, int rows, int cols, int relu, void* stream);
/* dlogit = p*(dp - sum p*dp), masked by pre_relu > 0 when pre_relu != NULL.  a / pre_relu: rows of `cols` values with row
 * stride a_stride / pre_stride (the conv output they come from is padded to a multiple of 32 columns) */
int mpn_softmax_rows_backward(const float* p, const float* dp, const float* pre_relu, int64_t pre_stride, float* dlogit, int rows, int cols, void* stream);
int mpn_bce_mean_backward(const float* p, const float* label, float* dp, int64_t n, const float* gscale, void* stream);
/* nn.Dropout (posenet.py:139,341-342): y = keep(seed, i) ? x / (1 - p) : 0, counter-based (same call with the same seed
 * applied to a gradient is the backward) */
int mpn_dropout(const void* x, void* y, int64_t n, uint64_t seed, float p, int dtype, void* stream);
int mpn_bce_chunks(int64_t n);
int mpn_bce_mean_forward(const float* p, const float* label, int64_t n, float* partial, int chunks, float* out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Boxes (network/utils.py:19-61, network/posenet.py:266-271)
 * -------------------------------------------------------------------------------------------*/
/* img_w < 0 disables the clip (pure BBoxTransform) */
int mpn_box_decode_clip(const float* anchors, const float* deltas, float* boxes, int B, int A,
                        float img_w, float img_h, void* stream);
/* the same with BBoxTransform's optional coefficients (network/utils.py:8-17): mean_std = {mean[0..3], std[0..3]}, a HOST array */
int mpn_box_decode_clip_ms(const float* anchors, const float* deltas, float* boxes, int B, int A, float img_w,
                           float img_h, const float* mean_std, void* stream);
int mpn_clip_boxes(float* boxes, int64_t n, float img_w, float img_h, void* stream);   /* in place, utils.py:55-59 */
/* compact image-0 candidates with score > thresh into dets[n,5] (x1,y1,x2,y2,score) + src index;
 * order preserved (ascending anchor index).  count[0] receives n. */
int mpn_score_filter(const float* boxes, const float* scores, int A, float thresh, float* dets,
                     int32_t* src_idx, int32_t* count, void* stream);

/* the same for every image of a batch in one launch (BASELINE config 5: batch 64; the reference thresholds image 0 only,
 * posenet.py:271): boxes [B,A,4], scores [B,A]; image b's candidates go to dets[b*A*5 ...] (order preserved),
 * src_idx[b*A ...] (optional), counts[b]. */
int mpn_score_filter_batched(const float* boxes, const float* scores, int B, int A, float thresh, float* dets,
                             int32_t* src_idx, int32_t* counts, void* stream);

/* boxes[k,4], scores[k] = rows keep[0..k) of dets[n,5] */
int mpn_gather_dets(const float* dets, const int64_t* keep, int k, float* boxes, float* scores, void* stream);
/* per image b < B: boxes[b*out_stride*4 ...], scores[b*out_stride ...] = rows keep[b*keep_stride + (0..num[b])) of the
 * image's candidate block dets + b*dets_stride (floats); kmax >= max num[b] sizes the launch */
int mpn_gather_dets_batched(const float* dets, int64_t dets_stride, const int64_t* keep, int64_t keep_stride, const int64_t* num,
                            int B, int kmax, float* boxes, float* scores, int64_t out_stride, void* stream);

/* ---------------------------------------------------------------------------------------------
 * NMS — replaces lib/nms: gpu_nms (src/nms_cuda.c:17-67), _nms/nms_kernel
 * (src/cuda/nms_kernel.cu:26-83) and the sort/gather of pth_nms (pth_nms.py:25-44).
 * dets [n,5] f32 device, unsorted.  mode 0: suppress if IoU > thresh (reference GPU path),
 * mode 1: IoU >= thresh (reference CPU path, nms.c:59).  keep_out [n] i64 device receives ORIGINAL
 * indices in descending-score order (ties: lower index first); num_out [1] i64 device.
 * workspace: mpn_nms_workspace_bytes(n) bytes of device memory.
 * -------------------------------------------------------------------------------------------*/
int64_t mpn_nms_workspace_bytes(int64_t n);
int mpn_nms(const float* dets, int64_t n, float thresh, int mode, int64_t* keep_out, int64_t* num_out,
            void* workspace, void* stream);
/* Segmented NMS: B independent problems in the same three launches, sizes read ON THE DEVICE (counts[b] candidates for
 * image b, rows at dets + b*dets_stride floats), so a batch needs no per-image host round trip.  nmax >= max counts[b]
 * sizes the launches and the workspace (mpn_nms_batched_workspace_bytes(B, nmax)); keep_out [B][keep_stride] i64,
 * num_out [B] i64.  Each image's result equals mpn_nms on that image alone. */
int64_t mpn_nms_batched_workspace_bytes(int B, int64_t nmax);
int mpn_nms_batched(const float* dets, int64_t dets_stride, const int32_t* counts, int B, int64_t nmax, float thresh, int mode,
                    int64_t* keep_out, int64_t keep_stride, int64_t* num_out, void* workspace, void* stream);
/* mpn_nms_batched with a cap ahead of the suppression: per image only the top_k best-scored candidates (ties by index: the
 * order of the sort) take part, the rest are dropped.  NOT in the reference (posenet.py:269-285 hands every candidate above
 * 0.05 to nms) and off unless the caller asks: it bounds the N x N/64 mask of a dense image (A = 76 725 at 640x640: 736 MB)
 * to top_k x top_k/64.  workspace: mpn_nms_batched_workspace_bytes(B, min(nmax, top_k)); keep_stride >= min(nmax, top_k). */
int mpn_nms_batched_topk(const float* dets, int64_t dets_stride, const int32_t* counts, int B, int64_t nmax, int64_t top_k,
                         float thresh, int mode, int64_t* keep_out, int64_t keep_stride, int64_t* num_out, void* workspace,
                         void* stream);

/* ---------------------------------------------------------------------------------------------
 * Optimizer: torch.optim.Adam semantics (training/multipose_keypoint_train.py:106-110), fused over
 * a flat f32 arena.  step_size/bias corrections are computed on the host and passed in.
 * -------------------------------------------------------------------------------------------*/
int mpn_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                  float lr, float beta1, float beta2, float eps, float weight_decay,
                  float bias_correction1, float bias_correction2_sqrt, float grad_scale, void* stream);
/* The same update with every scalar read from DEVICE memory, so that a captured hipGraph of the training step replays
 * correct Adam steps: hyper[0..7] = {lr, beta1, beta2, eps, weight_decay, grad_scale, bias_correction1,
 * sqrt(bias_correction2)} (floats), hyper[8] = step count (int32 bits).  mpn_adam_advance does step += 1 and recomputes
 * hyper[6..7] (double-precision pow, the values torch computes on the host); mpn_adam_step_dev applies one update to a
 * 16-byte aligned run of the arena.  The host only rewrites hyper[0] when the scheduler changes the learning rate. */
int mpn_adam_advance(float* hyper, void* stream);
int mpn_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, const float* hyper,
                      void* stream);
int mpn_fill_f32(float* dst, float v, int64_t n, void* stream);
/* device-to-device copy on the stream (gradient hand-over between two activations of equal geometry) */
int mpn_copy_bytes(void* dst, const void* src, int64_t nbytes, void* stream);
/* log vector of a recorded training step (replay.py): kp8 = out of mpn_mse_heatmap_forward (or NULL), det2 = out of
 * mpn_focal_forward (or NULL); logv[0..7] = kp8, [8] = cls + reg, [9] = cls, [10] = reg, [11] = loss (the sums the reference
 * forms with torch adds at posenet.py:387,421 and the combined step of SURVEY.md 8d) */
int mpn_step_log(const float* kp8, const float* det2, float* logv, void* stream);

/* ---------------------------------------------------------------------------------------------
 * conv2 of the keypoint head by position classes (network/posenet.py:311-315: conv2 over cat(up8(q5), up4(q4), up2(q3), q2)).
 * The x8 / x4 members contribute through nine class maps each at their own resolution (csrc/conv2cls.hip has the derivation);
 * these entry points are the streaming pieces around the ordinary convolution launches.  O = output channels (256), C = channels
 * per member (128); w / dw: f32 [O][3][3][4 C].
 *   comb (f32, mpn_conv2cls_comb_elems(O, C) elements) = Wm [O][3][3][2C] | Wc8 [9 O][3][3][C] | Wc4 [9 O][3][3][C] (frame filters of
 *          the forward class convolutions) | Wtap8 [9 O][C] | Wtap4 [9 O][C] (row t * O + o = tap t of output channel o)
 *   combine: comb <- w;      fold: dw += dcomb, dcomb = dWm [O][3][3][2C] | dWtap8 [9 O][C] | dWtap4 [9 O][C]
 *   tapsum : g[b,i,j,t*O+:] (like pc) = sum of dy over the output pixels whose tap t reads (i, j), from the class sums pc
 *   expand : e[b,y,x,:] ([B,H,W,O] in `dtype`) = m8[b, y/8, x/8, class * O + :] + m4[b, y/4, x/4, class * O + :]   (m: f32 [B,h,w,9 O])
 *   classsum: m (f32 class maps) from t (f32 [B,h,w,9 O], t[..., tap * O + :] = the 1x1 convolution with filter tap `tap`): the forward
 *            per tap, used where the zero taps of the frame filters cost matrix time (f32)
 *   pool   : p8 / p4 ([B,H/s,W/s,9 O] in `dtype`) = per-class sums of dy ([B,H,W,O]) over s x s blocks; H, W multiples of 8
 * -------------------------------------------------------------------------------------------*/
int64_t mpn_conv2cls_comb_elems(int O, int C);
int mpn_conv2cls_combine(const float* w, float* comb, int O, int C, void* stream);
int mpn_conv2cls_expand(const float* m8, const float* m4, void* e, int B, int H, int W, int O, int dtype, void* stream);
int mpn_conv2cls_classsum(const float* t, float* m, int B, int h, int w, int O, void* stream);
int mpn_conv2cls_pool(const void* dy, void* p8, void* p4, int B, int H, int W, int O, int dtype, void* stream);
int mpn_conv2cls_tapsum(const void* pc, void* g, int B, int h, int w, int O, int dtype, void* stream);
int mpn_conv2cls_fold(const float* dcomb, float* dw, int O, int C, void* stream);

/* library self-description */
const char* mpn_version(void);

/* ---------------------------------------------------------------------------------------------
 * Ground-truth heat-maps (datasets/coco_data/heatmap.py:20-41 + COCO_data_pipeline.py:218-236,283):
 * out[b][k][y][x] (f32, 18 keypoint channels) = min(1, sum over people j < num_people[b] with
 * visibility <= 1 of exp(-e) where e = d2/2/sigma/sigma <= 4.6052), float64 arithmetic in annotation
 * order.  joints: double [B][maxP][18][3] = (x, y, visibility) in crop pixels.
 * -------------------------------------------------------------------------------------------*/
int mpn_gt_heatmaps(const double* joints, const int32_t* num_people, int B, int maxP, float* out, int gh, int gw,
                    double stride, double sigma, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Heat-map peak extraction (network/joint_utils.py:19-31 find_peaks, :61-138 NMS).  heat: f32, element
 * (b, j, y, x) at heat[b*sB + j*sJ + y*sY + x*sX].  A cell is a peak if it is > thre1 and no 4-neighbour
 * is larger.  peaks[b][j][slot][4] = (x, y, score, id) as doubles, slots in row-major cell order, ids
 * counted over joints 0..J-1 in that order; with refine != 0 the position/score come from the bicubically
 * (cv2 INTER_CUBIC) `upsamp`-times up-sampled 5x5 patch around the cell, else from the cell itself
 * ((c + 0.5)*upsamp - 0.5, rounded half to even).  counts[b][j] = number of peaks found (only the first
 * `cap` are stored).  workspace: mpn_heatmap_peaks_workspace_bytes(B, J, H, W, cap) bytes of device scratch (one flag bit
 * per cell, the device-wide peak list of the refinement launch); contents need no initialisation.  Three launches: flags
 * (coalesced in channels-last [sJ == 1, sX == J] and planar [sX == 1] layouts), per-plane compaction, refinement.
 * -------------------------------------------------------------------------------------------*/
int64_t mpn_heatmap_peaks_workspace_bytes(int B, int J, int H, int W, int cap);
int mpn_heatmap_peaks(const float* heat, int64_t sB, int64_t sJ, int64_t sY, int64_t sX, int B, int J, int H, int W,
                      float thre1, double upsamp, int refine, double* peaks, int32_t* counts, int cap, void* workspace,
                      void* stream);

/* cv2.resize for float32 images / heat-map stacks (evaluate/tester.py:67,213,296-299): src element (y, x, c) at
 * src[y*sY + x*sX + c*sC]; dst dense [Hd][Wd][C]; cubic != 0 -> INTER_CUBIC, else INTER_LINEAR (OpenCV's coordinate rule
 * (d + 0.5)*scale - 0.5, clamped taps, horizontal then vertical pass, float32).  scale = src/dst (the dsize call form) unless
 * inv_fy / inv_fx > 0: those are 1/fy, 1/fx of the `cv2.resize(img, None, fx=, fy=)` form (tester.py:68), used as the scale exactly. */
int mpn_resize(const float* src, int64_t sY, int64_t sX, int64_t sC, int Hs, int Ws, int C, float* dst, int Hd, int Wd,
               int cubic, double inv_fy, double inv_fx, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Pose-residual-network person assignment, device half (evaluate/tester.py:333-513; crop/gaussian helpers
 * datasets/coco_data/prn_gaussian.py:2,134-158).
 * mpn_prn_build_maps: for nboxes boxes (x, y, w, h doubles; box_img[b] = image of box b) and the heat-map peaks of their
 * images (peaks [n][2] doubles grouped by image then joint type; joint_off [nimg][18] offsets) builds
 *   occ    [nboxes][17][H][W] int32 : 1 + id of the peak assigned to the cell (id = position among its image's peaks), 0 = empty,
 *            with the reference's inside test, float64 cell arithmetic, one-branch clamp chain and overwrite order
 *            (tester.py:363-392); *err is set to 1 where the reference would raise IndexError;
 *   prn_in [nboxes][H][W][17] float32: the one-hot planes blurred by skimage.filters.gaussian (sigma 1, 'nearest', 9 taps;
 *            weights9 = the normalised kernel as scipy computes it), in scipy.ndimage.correlate1d's summation order.
 * mpn_prn_scores: score[b][t][y][x] (where occ > 0) = sum of the N x N window of prn_out[b][:, :, t] around (y, x), clipped
 * as prn_gaussian.crop does, in np.sum's own float32 order (pairwise over the row-major flattened window: exact ties between
 * candidates exist and the reference resolves them by sort order; tester.py:418-419); N odd, <= 15; argmax[b][t] = first row-major maximum of
 * the plane (tester.py:480).  H*W <= 4096.
 * -------------------------------------------------------------------------------------------*/
int mpn_prn_build_maps(const double* peaks, const int32_t* joint_off, const double* boxes, const int32_t* box_img, int nboxes,
                       int H, int W, double in_thres, const double* weights9, int32_t* occ, float* prn_in, int32_t* err,
                       void* stream);
int mpn_prn_scores(const float* prn_out, const int32_t* occ, int nboxes, int H, int W, int N, float* score, int32_t* argmax,
                   void* stream);
/* mpn_prn_scores with a compact result (the full score / occupancy planes are 274 KB per box): per (box, joint type) the occupied
 * cells in row-major order — np.argwhere's order at tester.py:415 — as cand_id (peak id = occ - 1) and cand_score (window sum),
 * at most `cap` per plane (cand_n holds the true count; *overflow is set when one exceeds cap), plus the per-plane arg-max. */
int mpn_prn_scores_compact(const float* prn_out, const int32_t* occ, int nboxes, int H, int W, int N, int cap,
                           int32_t* cand_n, int32_t* cand_id, float* cand_score, int32_t* argmax, int32_t* overflow, void* stream);
/* HOST function (no device work, host pointers): the greedy (boxes x peaks) matching of tester.py:432-485 for a whole batch on the
 * compact candidate lists.  box_start [nimg+1]: the boxes of image i are box_start[i] .. box_start[i+1]; boxes (x, y, w, h);
 * joint_off / peaks as for mpn_prn_build_maps; out_kp [nb][17][3] (x, y, score) must be zero-filled.  Where the result would depend
 * on numpy's ordering of EQUAL positive scores, tie_flags[image*17 + type] is set and that pair is left to the caller. */
int mpn_prn_match_host(int nimg, const int32_t* box_start, const double* boxes, const int32_t* joint_off, const double* peaks,
                       const int32_t* cand_n, const int32_t* cand_id, const float* cand_score, int cap, const int32_t* argmax,
                       int H, int W, double* out_kp, uint8_t* tie_flags);

#ifdef __cplusplus
}
#endif
#endif /* MPN_H_ */
