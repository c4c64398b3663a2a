/*
 * mindpose_hip.h - C ABI of libmindpose_hip.so: the MI355X (gfx950) hot path of the
 * top-down heat-map pose pipeline.
 *
 * The reference (mindspore-lab/mindpose) has no FFI for this path: every function below
 * replaces a MindSpore-op sequence issued from a Python `construct`/`transform` method
 * (file:line cited per entry).  INTEGRATION.md shows the ctypes binding a maintainer adds.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch / HIP types in the signatures
 *     (`mp_stream_t` is a hipStream_t passed as void*; NULL = the default stream).
 *   - every pointer marked `dev` is DEVICE memory owned by the caller, contiguous, NCHW, fp32.
 *   - asynchronous on `stream`; no hidden synchronisation, no internal allocation
 *     (scratch comes from a caller-provided workspace sized by mp_*_workspace_bytes).
 *   - returns MP_OK (0) or a negative MP_ERR_* code; never throws, never aborts.
 *   - re-entrant; thread-safe for distinct streams.
 */
#ifndef MINDPOSE_HIP_H
#define MINDPOSE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* mp_stream_t;

#define MP_OK 0
#define MP_ERR_NULL (-1)        /* required pointer is NULL */
#define MP_ERR_SHAPE (-2)       /* non-positive / inconsistent dimension */
#define MP_ERR_UNSUPPORTED (-3) /* valid request outside what the kernels implement */
#define MP_ERR_HIP (-4)         /* HIP runtime error at launch (see mp_last_hip_error) */
#define MP_ERR_WORKSPACE (-5)   /* workspace missing or too small */

const char* mp_version(void);
const char* mp_error_string(int code);
int mp_last_hip_error(void); /* hipError_t of the most recent MP_ERR_HIP on this thread */

/* ------------------------------------------------------------------------------------------
 * Decoder: TopDownHeatMapDecoder.construct, mindpose/models/decoders/top_down_decoder.py:72-94
 *   _get_max_preds :96-116 (per-joint arg-max, first index on ties, no maxval>0 mask)
 *   _shift_coordinate :118-141 | _dark_udp_refine_coords :171-205 | _transform_preds :143-169
 * heatmap [N,K,H,W]; center,scale [N,2]; score [N] -> preds [N,K,3]=(x,y,maxval), boxes [N,6].
 * argmax_idx (optional, may be NULL): [N,K] int32 flat arg-max index (bit-exact target).
 * blur_kernel: dev [kernel_size^2] normalised Gaussian (:207-215), required for MP_REFINE_DARK.
 * ------------------------------------------------------------------------------------------ */
#define MP_REFINE_NONE 0
#define MP_REFINE_SHIFT 1 /* shift_coordinate=True */
#define MP_REFINE_DARK 2  /* dark_udp_refine=True  */

int mp_decode_topdown(const float* heatmap_dev, const float* center_dev, const float* scale_dev,
                      const float* score_dev, float* preds_dev, float* boxes_dev, int32_t* argmax_idx_dev,
                      int n, int k, int h, int w, int refine_mode, int use_udp, int to_original,
                      float pixel_std, const float* blur_kernel_dev, int kernel_size, mp_stream_t stream);

/* mp_decode_topdown in MP_REFINE_DARK mode that ALSO writes the intermediates of the refinement (top_down_decoder.py:176-204)
 * for parity tests: dark_terms_dev [N*K][16] = the 3x3 neighbourhood of log(clip(blur(heatmap))) around the arg-max, row-major
 * (rows y-1, y, y+1; zero outside the map = the reference's zero pad in log space) [0..8], dx, dy [9,10], dxx, dyy, dxy [11..13],
 * det(Hessian + 1e-7 I) [14], 0 [15].  Same kernel, same arithmetic; the product path passes no such pointer. */
int mp_decode_topdown_debug(const float* heatmap_dev, const float* center_dev, const float* scale_dev,
                            const float* score_dev, float* preds_dev, float* boxes_dev, int32_t* argmax_idx_dev,
                            int n, int k, int h, int w, int refine_mode, int use_udp, int to_original,
                            float pixel_std, const float* blur_kernel_dev, int kernel_size, float* dark_terms_dev,
                            mp_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Flip-test aggregation: _MultiRunNet.construct, mindpose/engine/inferencer/topdown_inferencer.py:165-187
 *   avg = (heatmap + flip_back(flipped)[shifted]) * 0.5 ; decoder(avg, ...)
 * flip_index [K] int32 (:78-80).  avg_out (optional, may be NULL): [N,K,H,W] averaged heat-map.
 * mp_flip_aggregate_decode fuses aggregation and decode in one pass (avg never leaves registers
 * unless avg_out is given).
 * ------------------------------------------------------------------------------------------ */
int mp_flip_aggregate(const float* heatmap_dev, const float* flipped_dev, const int32_t* flip_index_dev,
                      float* avg_out_dev, int n, int k, int h, int w, int shift_heatmap, mp_stream_t stream);

int mp_flip_aggregate_decode(const float* heatmap_dev, const float* flipped_dev, const int32_t* flip_index_dev,
                             int shift_heatmap, float* avg_out_dev, const float* center_dev,
                             const float* scale_dev, const float* score_dev, float* preds_dev, float* boxes_dev,
                             int32_t* argmax_idx_dev, int n, int k, int h, int w, int refine_mode, int use_udp,
                             int to_original, float pixel_std, const float* blur_kernel_dev, int kernel_size,
                             mp_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Target generation: TopDownGenerateTarget._encoding / _udp_encoding,
 * mindpose/data/transform/topdown_transform.py:324-375, :377-430 (batched over samples).
 * keypoints [N,K,3] (x,y,vis) in input-image px -> target [N,K,H,W], target_weight [N,K].
 * feat_stride_x/y: the reference's float64 feature stride (image/heatmap, or (image-1)/(heatmap-1)
 * for UDP), computed by the host mirror exactly as the reference does.
 * patch_dev: plain mode only - the precomputed un-normalised Gaussian patch [patch_side^2] fp32
 * (:335-344), built on the host with the reference's numpy expression so values are bit-identical.
 * joint_weights_dev: optional [K] float64 (use_different_joint_weights), else NULL.
 * ------------------------------------------------------------------------------------------ */
int mp_gaussian_target(const float* keypoints_dev, const float* patch_dev, int patch_side,
                       const double* joint_weights_dev, float* target_dev, float* target_weight_dev, int n,
                       int k, int h, int w, double feat_stride_x, double feat_stride_y, double sigma,
                       int use_udp, mp_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Loss: JointsMSELoss.construct, mindpose/models/loss/mse.py:36-44
 *   L = mean_{n,k,h,w}( w[n,k] * (pred-target)^2 ) ; weight_dev NULL -> use_target_weight=False.
 * Deterministic two-stage fp32/fp64 reduction; workspace >= mp_joints_mse_workspace_bytes(n,k).
 * bwd: grad_pred = grad_out * 2 * w * (pred-target) / (N*K*H*W); grad_out_dev NULL means 1.0.
 * ------------------------------------------------------------------------------------------ */
size_t mp_joints_mse_workspace_bytes(int n, int k);
int mp_joints_mse_fwd(const float* pred_dev, const float* target_dev, const float* weight_dev, float* loss_dev,
                      void* workspace_dev, size_t workspace_bytes, int n, int k, int hw, mp_stream_t stream);
int mp_joints_mse_bwd(const float* pred_dev, const float* target_dev, const float* weight_dev,
                      const float* grad_out_dev, float* grad_pred_dev, int n, int k, int hw, mp_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Convolution family (direct, im2col-free, LDS-tiled, fp32 MFMA v_mfma_f32_16x16x4_f32):
 *   nn.Conv2d + nn.BatchNorm2d(eval) + residual add + nn.ReLU of
 *     BasicBlock  mindpose/models/backbones/hrnet.py:66-83, Bottleneck :126-146,
 *     HRModule fuse rows :318-344 (incl. ResizeNearestNeighbor + add :333-336),
 *     transitions :440-496, stem :568-573, HRNetHead heads/hrnet_head.py:47-49,
 *     ResNet stem/blocks backbones/resnet.py:118-138,:247-264,
 *     SimpleBaselineHead heads/simple_baseline_head.py:80-90 (Conv2dTranspose k4 s2 p1 as four
 *     2x2 sub-pixel phase convolutions) and its final 1x1 conv :57-62.
 *
 * out[n,co,oy,ox] = act( conv(x,w)[n,co,y,x] * scale[co] + shift[co] (+ res1[...]) (+ res2[...]) )
 * with (oy,ox) = (y*out_mul + out_off_y + a, x*out_mul + out_off_x + b), a,b in [0,out_rep):
 *   normal conv          out_mul=1 out_rep=1
 *   nearest-upsample+add out_mul=s out_rep=s   (HRModule fuse, j>i)
 *   deconv phase (py,px) out_mul=2 out_rep=1 out_off=(py,px)
 * res1/res2 (optional) are indexed like `out` ([N,Cout,out_h,out_w]); res1 may alias out.
 * Weights are pre-packed once by mp_conv_pack_weight (device -> device).
 * ------------------------------------------------------------------------------------------ */
typedef struct mp_conv_desc {
    int32_t n, cin, h, w;          /* input [N,Cin,H,W] */
    int32_t cout, kh, kw;          /* kernel (kh == kw in {1,2,3,7}) */
    int32_t stride;                /* 1 or 2 */
    int32_t pad_top, pad_left;     /* zero padding before the first row / column */
    int32_t conv_h, conv_w;        /* conv output extent (before the output mapping) */
    int32_t out_h, out_w;          /* extent of the out / res tensors */
    int32_t out_mul, out_rep, out_off_y, out_off_x;
    int32_t relu;                  /* apply ReLU last */
    int32_t flags;                 /* MP_CONV_* bits; an unknown bit is refused (MP_ERR_UNSUPPORTED) */
} mp_conv_desc;

/* mp_conv_desc.flags: the launch will share the CUs with other kernels of the caller - a training step's BatchNorm and
 * weight-gradient launches on sibling streams.  Forms that take a CU's whole register file (the two-team Winograd
 * workgroup) are then not chosen: they are faster alone and slower in that company. */
#define MP_CONV_SHARES_CUS 1
/* fp16 family only (mp_f16_conv2d_fwd): the FOUR 2x2 sub-pixel phase convs of a stride-2 3x3 data gradient as ONE launch.  The
 * descriptor is one phase's (kh = kw = 2, stride 1, out_mul 2, out_off_* = 0; all four phases share its padding - the tap
 * alignment is in the packing, mp_f16_pack_weight mode 3); phase (py, px) writes output pixels (2 y + py, 2 x + px) and reads its
 * weights from slice 2 py + px of the packed buffer (four slices of mp_f16_packed_weight_bytes(cout, cin, 2, 2) bytes each).
 * One-tile and persistent multi-tile variants only; no residual; statistics: the backward sums (mp_f16_conv2d_fwd_stats mode 2,
 * partial slots = four times a phase's). */
#define MP_CONV_PHASES4 2

/* bytes of the packed weight buffer for a (cout, cin, kh, kw) kernel */
size_t mp_conv_packed_weight_bytes(int cout, int cin, int kh, int kw);
/* Weight source layouts (`transposed` selects how w_dev is read; cout/cin/kh/kw describe the PACKED convolution):
 *   0  w_dev = [Cout,Cin,kh,kw]   forward Conv2d
 *   1  w_dev = [Cin,Cout,4,4]     Conv2dTranspose(k4,s2,p1) sub-pixel phase (py,px), kh=kw=2
 *   2  w_dev = [Cin,Cout,kh,kw]   data gradient of a stride-1 Conv2d whose forward weight is w_dev (roles swapped,
 *                                 taps mirrored): dx = conv(dz, pack2(W)) with the same padding
 *   3  w_dev = [Cin,Cout,3,3]     data gradient of a 3x3 stride-2 pad-1 Conv2d, output parity phase (py,px), kh=kw=2,
 *                                 run with pad 0 and the deconv output mapping (out_mul=2, out_off=(py,px))
 *   5 / 6  (kh=kw=3)              the Winograd forms U = G w G^T of mode 0 / mode 2, for mp_conv2d_winograd_fwd; packed_dev holds
 *                                 mp_conv_winograd_packed_weight_bytes (also as jobs of mp_conv_pack_weight_batch: units_j =
 *                                 Cin_pad4 * Cout_pad16, a thread writes the 16 values of one (cin, cout) pair) */
int mp_conv_pack_weight(const float* w_dev, float* packed_dev, int cout, int cin, int kh, int kw,
                        int transposed, int phase_y, int phase_x, mp_stream_t stream);
int mp_conv2d_fwd(const mp_conv_desc* desc, const float* x_dev, const float* packed_w_dev,
                  const float* scale_dev, const float* shift_dev, const float* res1_dev, const float* res2_dev,
                  float* out_dev, mp_stream_t stream);

/* Same as mp_conv2d_fwd / mp_plan_add_conv with the tile variant forced (0..7 = cout tile x pixel tile builds of the direct
 * kernel; 8 = the streaming 1x1 kernel for the HBM-bound stage-1 / layer1 layers - Cin 64 / 128 / 256 with Cin * Cout <= 16384,
 * image planes a multiple of 64 pixels, one residual tensor, same packed weights; 10 = the blocked-GEMM kernel for 1x1 stride 1 / 2
 * and the 2x2 stride-1 sub-pixel phases of the transposed convolution - Cin a multiple of 16, Cout >= 96, planes a multiple of 4
 * pixels, one residual tensor, same packed weights; 11 = the K-split kernel for SMALL problems - 3x3 pad 1 stride 1 / 2 and 1x1 stride 1 (a handful of
 * crops: 16 couts x 16 pixels per workgroup, its eight waves split the k loop - csrc/conv_small_f32.hip; at most 1024 workgroups,
 * up to two residual tensors, same packed weights), 12 = the same kernel with 48 / 64 pixels per workgroup (a few dozen crops; at
 * most 2048 workgroups; bit-identical to 11); 9 is the host tuner's index of the Winograd form, which has its own entry
 * points below; -1 = library heuristic).  Returns MP_ERR_UNSUPPORTED when that variant cannot run the shape.  Used by the host-side autotuner, which times the candidates once per distinct layer shape. */
int mp_conv2d_fwd_variant(const mp_conv_desc* desc, int variant, const float* x_dev, const float* packed_w_dev,
                          const float* scale_dev, const float* shift_dev, const float* res1_dev, const float* res2_dev,
                          float* out_dev, mp_stream_t stream);

/* The 3x3 stride-1 padding-1 convolutions of the branches (hrnet.py:51-64 BasicBlock, 202-241 _make_one_branch) in Winograd
 * F(2x2,3x3) form, all fp32: 16 instead of 36 multiplications per 2x2 output tile and (cin, cout) pair; same operands, epilogue
 * (folded BatchNorm scale/shift, res1, res2, ReLU) and result as mp_conv2d_fwd up to fp32 rounding (the sums are associated
 * differently).  Needs even H and W <= 96 (W % 4 != 0: whole images of at most 24 tiles with H*W % 4 == 0), Cin % 8 == 0 and the plain
 * output mapping; mp_conv_winograd_supported
 * returns MP_OK or MP_ERR_UNSUPPORTED for a descriptor.  The weights are transformed once: U = G w G^T,
 * [Cin_pad4][Cout_pad16][16] floats (mp_conv_winograd_packed_weight_bytes). */
size_t mp_conv_winograd_packed_weight_bytes(int cout, int cin);
int mp_conv_winograd_pack_weight(const float* w_dev, float* packed_dev, int cout, int cin, mp_stream_t stream);
int mp_conv_winograd_supported(const mp_conv_desc* desc);
int mp_conv2d_winograd_fwd(const mp_conv_desc* desc, const float* x_dev, const float* packed_u_dev, const float* scale_dev,
                           const float* shift_dev, const float* res1_dev, const float* res2_dev, float* out_dev,
                           mp_stream_t stream);

/* nn.MaxPool2d(kernel_size=3, stride=2, pad_mode="same"), resnet.py:190: pads bottom/right only. */
int mp_maxpool3x3s2_same(const float* x_dev, float* out_dev, int n, int c, int h, int w, mp_stream_t stream);

/* HRModule exchange unit, up-sampling side (mindpose/models/backbones/hrnet.py:327-339, j > i terms):
 *   out = act( ((base + up_s1(t1)) + up_s2(t2)) + up_s3(t3) )      (reference summation order)
 * base/out [N,C,H,W]; t_k [N,C,H/s_k,W/s_k] already holds BN(conv1x1(x_j)); up_s = nearest (src = dst / s).
 * t2/t3 may be NULL (s ignored).  One streaming pass: the up-sampled terms are never materialised and the
 * full-resolution sum is read and written once per fuse row.  out may alias base. */
int mp_fuse_upsample_sum(const float* base_dev, const float* t1_dev, int s1, const float* t2_dev, int s2,
                         const float* t3_dev, int s3, float* out_dev, int n, int c, int h, int w, int relu,
                         mp_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Launch plan: a recorded sequence of the calls above (one HRNet / ResNet forward = ~300 launches)
 * replayed by ONE native call, so the per-layer host cost is paid once at build time.
 * ------------------------------------------------------------------------------------------ */
typedef struct mp_plan mp_plan;
mp_plan* mp_plan_create(void);
void mp_plan_destroy(mp_plan* plan);
int mp_plan_add_conv(mp_plan* plan, const mp_conv_desc* desc, const float* x_dev, const float* packed_w_dev,
                     const float* scale_dev, const float* shift_dev, const float* res1_dev,
                     const float* res2_dev, float* out_dev);
int mp_plan_add_conv_variant(mp_plan* plan, const mp_conv_desc* desc, int variant, const float* x_dev,
                             const float* packed_w_dev, const float* scale_dev, const float* shift_dev,
                             const float* res1_dev, const float* res2_dev, float* out_dev);
/* mp_conv2d_winograd_fwd as a plan entry (same operands; MP_ERR_UNSUPPORTED when the descriptor is outside the Winograd form). */
/* Conv2dTranspose(k=4, s=2, pad_mode="pad", padding=1) + scale / shift (+ReLU) - simple_baseline_head.py:80-90 - with ALL FOUR
 * sub-pixel phases in one launch of the blocked-GEMM kernel (variant 10 above; phase = a grid dimension, so the small-map layers
 * of the head fill the chip with efficient tiles).  `phase00_desc` is the phase (0, 0) launch of the per-phase form: kh = kw = 2,
 * stride 1, pad_top = pad_left = 1, conv = input size, out = 2x input size, out_mul 2, offsets 0; `packed4_dev` = the four phase
 * packings of mp_conv_pack_weight(transposed = 1, phase_y, phase_x) back to back in the order (0,0) (0,1) (1,0) (1,1), each
 * mp_conv_packed_weight_bytes(cout, cin, 2, 2) long.  Same arithmetic per output as four mp_conv2d_fwd_variant(10) launches. */
int mp_deconv4x4s2_gemm_supported(const mp_conv_desc* phase00_desc);
int mp_deconv4x4s2_gemm_fwd(const mp_conv_desc* phase00_desc, const float* x_dev, const float* packed4_dev, const float* scale_dev,
                            const float* shift_dev, float* out_dev, mp_stream_t stream);
int mp_plan_add_deconv4x4s2_gemm(mp_plan* plan, const mp_conv_desc* phase00_desc, const float* x_dev, const float* packed4_dev,
                                 const float* scale_dev, const float* shift_dev, float* out_dev);
int mp_plan_add_conv_winograd(mp_plan* plan, const mp_conv_desc* desc, const float* x_dev, const float* packed_u_dev,
                              const float* scale_dev, const float* shift_dev, const float* res1_dev, const float* res2_dev,
                              float* out_dev);
int mp_plan_add_maxpool(mp_plan* plan, const float* x_dev, float* out_dev, int n, int c, int h, int w);
int mp_plan_add_fuse_sum(mp_plan* plan, const float* base_dev, const float* t1_dev, int s1, const float* t2_dev, int s2,
                         const float* t3_dev, int s3, float* out_dev, int n, int c, int h, int w, int relu);
/* Execution lanes: entries added after mp_plan_set_lane(plan, l) (l in 0..3) replay on lane l - lane 0 is the stream
 * passed to mp_plan_run, lanes 1..3 are side streams the plan owns - so independent sub-graphs (the HRNet branches, the
 * rows of an exchange unit) overlap on the chip.  mp_plan_add_barrier orders every lane after everything recorded so far
 * on every other lane; mp_plan_run forks the side lanes from the caller's stream and joins them before returning control
 * to it (event fork/join: also valid under stream capture).  mp_plan_run_range always replays sequentially on one stream. */
int mp_plan_set_lane(mp_plan* plan, int lane);
int mp_plan_add_barrier(mp_plan* plan);
int mp_plan_size(const mp_plan* plan);
int mp_plan_run(const mp_plan* plan, mp_stream_t stream);
/* run entries [first, first+count) only (profiling / per-layer timing) */
int mp_plan_run_range(const mp_plan* plan, int first, int count, mp_stream_t stream);
/* launch geometry of entry `index` (roofline report): info[0]=kind (0 conv, 1 maxpool, 2 fuse-sum), [1]=kernel size,
 * [2]=stride, [3]=tile variant, [4]=workgroups, [5]=LDS bytes per workgroup, [6]=cout tile, [7]=pixel tile,
 * [8]=cin chunk, [9]=images per tile, [10]=rows per tile, [11]=light variant (3 workgroups/CU) */
int mp_plan_entry_info(const mp_plan* plan, int index, int64_t info[12]);

/* ------------------------------------------------------------------------------------------
 * Training-side kernels (reference: mindspore.Model.train over NetWithLoss, tools/train.py:170-233, amp aside:
 * everything here computes in fp32).  All reductions are deterministic (fixed partition, fixed combine order).
 *
 * BatchNorm2d in training mode (nn.BatchNorm2d cells of hrnet.py / resnet.py; eps 1e-5, momentum 0.9 meaning
 * moving = 0.9*moving + 0.1*batch [MS-knowledge]):
 *   fwd: per-channel batch mean / biased variance over (N,H,W) of z [N,C,HW];
 *        y = act(gamma*(z-mean)*invstd + beta (+ res)); saves mean/invstd; updates the moving statistics when given
 *        (moving variance takes the unbiased batch variance).
 *   bwd: g = dy masked by (y > 0) when relu; dgamma = sum g*xhat; dbeta = sum g;
 *        dz = gamma*invstd*(g - dbeta/M - xhat*dgamma/M); dres (optional) = g.
 * workspace >= mp_bn_workspace_bytes(c).
 * ------------------------------------------------------------------------------------------ */
size_t mp_bn_workspace_bytes(int c);
int mp_bn_train_fwd(const float* z_dev, const float* gamma_dev, const float* beta_dev, const float* res_dev, float* y_dev,
                    float* save_mean_dev, float* save_invstd_dev, float* moving_mean_dev, float* moving_var_dev, int n,
                    int c, int hw, float eps, float momentum, int relu, void* workspace_dev, size_t workspace_bytes,
                    mp_stream_t stream);
int mp_bn_train_bwd(const float* dy_dev, const float* z_dev, const float* y_dev, const float* gamma_dev,
                    const float* save_mean_dev, const float* save_invstd_dev, float* dz_dev, float* dres_dev,
                    float* dgamma_dev, float* dbeta_dev, int n, int c, int hw, int relu, void* workspace_dev,
                    size_t workspace_bytes, mp_stream_t stream);

/* backward of mp_fuse_upsample_sum: g = dy*(out>0 when relu); dbase = g; dt_k = s_k x s_k block sums of g.
 * Any of dbase / dt_k may be NULL (not needed). */
int mp_fuse_upsample_sum_bwd(const float* dy_dev, const float* out_dev, float* dbase_dev, float* dt1_dev, int s1,
                             float* dt2_dev, int s2, float* dt3_dev, int s3, int n, int c, int h, int w, int relu,
                             mp_stream_t stream);

/* mp_bn_train_bwd that ALSO adds dgamma / dbeta into dgamma_acc / dbeta_acc (both or neither; the caller's gradient arena) */
int mp_bn_train_bwd_acc(const float* dy_dev, const float* z_dev, const float* y_dev, const float* gamma_dev,
                        const float* save_mean_dev, const float* save_invstd_dev, float* dz_dev, float* dres_dev,
                        float* dgamma_dev, float* dbeta_dev, float* dgamma_acc_dev, float* dbeta_acc_dev, int n, int c, int hw,
                        int relu, void* workspace_dev, size_t workspace_bytes, mp_stream_t stream);

/* mindspore.nn.AdamWeightDecay (selected by mindpose/optim/optim_factory.py:69-72 for "adamw"): Adam WITHOUT bias
 * correction, eps added to sqrt(v), decoupled weight decay:  m=b1*m+(1-b1)*g; v=b2*v+(1-b2)*g*g;
 * p -= lr*(m/(sqrt(v)+eps) + wd*p).  Operates on one flat fp32 arena (params, grads, moments). */
int mp_adamw_step(float* param_dev, const float* grad_dev, float* exp_avg_dev, float* exp_avg_sq_dev, size_t count,
                  float lr, float beta1, float beta2, float eps, float weight_decay, mp_stream_t stream);

/* mp_adamw_step with the gradient read as g * grad_scale (fold 1 / (loss_scale * world_size) here instead of a separate pass over
 * the arena) and an optional device flag: when *skip_flag_dev != 0 the launch leaves parameters and moments untouched (the
 * overflow step of DynamicLossScaleManager, tools/train.py:170-173). */
int mp_adamw_step_scaled(float* param_dev, const float* grad_dev, float* exp_avg_dev, float* exp_avg_sq_dev, size_t count,
                         float lr, float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                         const int* skip_flag_dev, mp_stream_t stream);
/* Overflow check of the dynamic loss scale: *flag_dev |= 1 when any of grad_dev[0..count) is inf / nan (one read of the
 * arena; the caller zeroes the flag).  grad_dev must be 16-byte aligned. */
int mp_grad_finite_check(const float* grad_dev, size_t count, int* flag_dev, mp_stream_t stream);

/* ---- data-parallel gradient mean over RCCL/xGMI (tools/train.py:43-49: parallel_mode=data_parallel, gradients_mean=True) ----
 * A thin layer over the process's RCCL (bound at run time; mp_comm_available() == 0 when no RCCL can be found).
 *   mp_comm_get_unique_id  rank 0 fills 128 bytes (ncclUniqueId) that the host code broadcasts to the other ranks
 *   mp_comm_init_rank      every rank, after hipSetDevice: creates the communicator (collective call)
 *   mp_allreduce_grads     in-place all-reduce of count fp32 values on `stream` (average != 0: mean over ranks, ncclAvg);
 *                          asynchronous, capturable into a hipGraph like any RCCL call
 *   mp_reduce_scatter_allgather_grads   the same result as reduce-scatter + all-gather of count / nranks shards (count must be a
 *                          multiple of nranks): every xGMI link carries 1/nranks of the arena in each phase
 * Errors: MP_ERR_UNSUPPORTED (no RCCL), MP_ERR_HIP with the ncclResult_t in mp_comm_last_error(). */
int mp_comm_available(void);
int mp_comm_last_error(void);
int mp_comm_get_unique_id(void* id128_host);
int mp_comm_init_rank(void** comm_out, int nranks, const void* id128_host, int rank);
int mp_comm_destroy(void* comm);
/* ncclCommCount: the number of ranks the communicator itself reports (-1 on error) */
int mp_comm_count(void* comm);
int mp_allreduce_grads(void* comm, float* arena_dev, size_t count, int average, mp_stream_t stream);
int mp_reduce_scatter_allgather_grads(void* comm, float* arena_dev, size_t count, int nranks, int rank, int average,
                                      mp_stream_t stream);

/* The other optimizers registered by mindpose/optim/optim_factory.py:9-14, on the same flat fp32 arenas.
 * kind 1 = mindspore.nn.Adam (bias-corrected; hyper = {beta1, beta2, eps, beta1^t, beta2^t}; state1 = m, state2 = v),
 * 2 = nn.SGD (hyper = {momentum, dampening, nesterov, first_step}; state1 = momentum buffer, may be NULL when momentum == 0),
 * 3 = nn.Momentum (hyper = {momentum, use_nesterov}; state1 = accumulator), 4 = nn.Adagrad (state1 = accumulator, initialised
 * by the caller).  The gradient enters as g * grad_scale + weight_decay * p (static loss scale and L2 decay of those cells).
 * Update rules from the MindSpore documentation [MS-knowledge; no reference test pins them]. */
int mp_optimizer_step(int kind, float* param_dev, const float* grad_dev, float* state1_dev, float* state2_dev, size_t count,
                      float lr, float grad_scale, float weight_decay, const float hyper[5], mp_stream_t stream);

/* Conv2d weight gradient (training backward of every conv of hrnet.py): dW[co,ci,ky,kx] = sum_{n,y,x}
 * dz[n,co,y,x] * x[n,ci,y*s+ky-p,x*s+kx-p] for k in {1,3}, s in {1,2}, p = k/2 (desc as for the forward conv; its
 * output-mapping fields are ignored).  fp32 MFMA, pixel axis split over workgroups into slabs that are summed in a
 * fixed order (deterministic).  accumulate != 0 adds into dw.  workspace >= mp_conv_wgrad_workspace_bytes(desc). */
size_t mp_conv_wgrad_workspace_bytes(const mp_conv_desc* desc);
int mp_conv_wgrad(const mp_conv_desc* desc, const float* x_dev, const float* dz_dev, float* dw_dev, int accumulate,
                  void* workspace_dev, size_t workspace_bytes, mp_stream_t stream);

/* ---- fp16 matrix-core inference path (amp level O2 / O3; BASELINE.json configs[4] "fp16 MFMA") -----------------------
 * Replaces what mindspore.amp.auto_mixed_precision(network, "O2") does to the cells of mindpose/models/backbones/hrnet.py
 * when they run on the Ascend cube unit: fp16 conv operands, fp32 accumulation, BatchNorm in fp32, fp16 activations.
 * Activations are CHANNEL-BLOCKED fp16, [N][ceil(C/8)][H][W][8] ("c8"; padding channels zero), weights are packed
 * [ceil(Cin/32)][kh*kw][4][Cout_pad16][8] fp16; scale / shift are fp32 arrays of Cout_pad16 entries (zero beyond Cout).
 * Kernel 1x1, 2x2 (stride 1) or 3x3; output mapping plain or strided scatter (out_mul, out_off_*: the sub-pixel phases of
 * the transposed convolution), out_rep must be 1: the exchange-unit up-sampling runs in mp_f16_fuse_upsample_sum.  variant -1 = heuristic, 0..9 = forced tile shape
 * (5..9 = the light builds of 0..4: small chunks, three workgroups per CU), 10..19 = the persistent multi-tile kernel in
 * those shapes (two / one workgroup per CU), 20..23 = the 16-cout x 192-pixel shape (regular, light, multi-tile x2),
 * 24 = 32 couts x 384 pixels for layers whose K fits one chunk (Cin <= 32 at 3x3). */
size_t mp_f16_packed_weight_bytes(int cout, int cin, int kh, int kw);
size_t mp_f16_activation_bytes(int n, int c, int h, int w);
/* transposed = 0: Conv2d weight [Cout,Cin,kh,kw]; 1: the (phase_y, phase_x) 2x2 sub-pixel phase of a
 * Conv2dTranspose(k=4, s=2, pad=1) weight [Cin,Cout,4,4] (kh = kw = 2); 2 / 3: the data-gradient packings of
 * mp_conv_pack_weight (stride-1 conv: roles swapped + taps mirrored; 3x3 stride-2 conv: 2x2 parity phases); 4: the data
 * gradient of transposed-conv phase (phase_y, phase_x) (w = the [Cin_t,Cout_t,4,4] weight, cout := Cin_t, cin := Cout_t) */
int mp_f16_pack_weight(const float* w_dev, void* packed_dev, int cout, int cin, int kh, int kw, int transposed, int phase_y,
                       int phase_x, mp_stream_t stream);
/* All weight packings of one training step in ONE launch (the fp32 master weights change every step, and an HRNet-W32 step
 * needs 700 packings: forward + data-gradient forms).  jobs_dev: device array of n_jobs descriptors with the arguments of
 * mp_f16_pack_weight (same validity rules, checked by the caller); first_block_dev: device array of n_jobs + 1 prefix sums of
 * ceil(units_j / 256), units_j = ceil(cin/32) * 4 * Cout_pad16 (one thread per (8 input channels, cout) pair: it walks the kh * kw
 * taps; a table with MORE blocks per job - e.g. the units of round 4, x kh * kw - is accepted: the surplus blocks leave at once);
 * total_blocks = its last entry.  Both tables are caller-owned and must stay alive until the launch has run. */
typedef struct mp_f16_pack_job {
    const float* w;
    void* packed;
    int cout, cin, kh, kw, transposed, phase_y, phase_x, reserved;
} mp_f16_pack_job;
int mp_f16_pack_weight_batch(const mp_f16_pack_job* jobs_dev, const unsigned* first_block_dev, int n_jobs, unsigned total_blocks,
                             mp_stream_t stream);
/* the fp32 form (mp_conv_pack_weight per job; packed = fp32 [Cin_pad4/4][kh*kw][4][Cout_pad16]): same job descriptor,
 * units_j = Cin_pad4 * kh * kw * Cout_pad16 / 4 (a thread writes four output channels) */
int mp_conv_pack_weight_batch(const mp_f16_pack_job* jobs_dev, const unsigned* first_block_dev, int n_jobs, unsigned total_blocks,
                              mp_stream_t stream);
int mp_f16_to_c8(const float* x_nchw_dev, void* out_c8_dev, int n, int c, int h, int w, mp_stream_t stream);
int mp_f16_from_c8(const void* x_c8_dev, float* out_nchw_dev, int n, int c, int h, int w, mp_stream_t stream);
int mp_f16_conv2d_fwd(const mp_conv_desc* desc, int variant, const void* x_c8_dev, const void* packed_w_dev,
                      const float* scale_dev, const float* shift_dev, const void* res1_c8_dev, const void* res2_c8_dev,
                      void* out_c8_dev, mp_stream_t stream);
/* Fused BasicBlock of the 32-channel and (round 4) the 64-channel HRNet branch (hrnet.py:30-83, 202-241, eval mode; SURVEY 8(b)
 * "mp_basicblock_fused"):
 *   out = relu(bn2(conv3x3(relu(bn1(conv3x3(x))))) + x),  x / out channel-blocked fp16 [n][c/8][h][w][8] (24 < c <= 32, or c == 64),
 * distinct buffers; packed_w1 / packed_w2 = mp_f16_pack_weight(mode 0) of the two [c,c,3,3] weights, scale / shift = folded
 * BatchNorm (c rounded up to 16 fp32 entries each, zero beyond c).  The intermediate tensor stays in LDS; the result is bit-identical
 * to two mp_f16_conv2d_fwd calls.  rows = output rows per workgroup (0 = the largest that fits).  MP_ERR_UNSUPPORTED for other
 * widths / maps too wide for a band; mp_f16_basicblock_supported answers that question without launching (1 / 0). */
int mp_f16_basicblock_supported(int n, int c, int h, int w);
int mp_f16_basicblock_fwd(const void* x_c8_dev, const void* packed_w1_dev, const float* scale1_dev, const float* shift1_dev,
                          const void* packed_w2_dev, const float* scale2_dev, const float* shift2_dev, void* out_c8_dev, int n,
                          int c, int h, int w, int rows, mp_stream_t stream);
/* First convolution of the network in fp32 (hrnet.py:377-385: 3x3, stride 2, padding 1, 3 -> 64 channels, folded BatchNorm, ReLU) as a
 * streaming kernel: out[n][64][h/2][w/2] = act(conv(x; weight) * scale + shift), x fp32 NCHW [n][3][h][w], weight = the [64][3][3][3]
 * tensor itself (no packing).  Same arithmetic as mp_conv2d_fwd on this layer with the 27 products summed in (tap, channel) order.
 * h even, w a multiple of 32: MP_ERR_UNSUPPORTED otherwise. */
int mp_stem_conv_fwd(const float* x_dev, const float* weight_dev, const float* scale_dev, const float* shift_dev, int relu, float* out_dev,
                     int n, int h, int w, mp_stream_t stream);
int mp_plan_add_stem_conv(mp_plan* plan, const float* x_dev, const float* weight_dev, const float* scale_dev, const float* shift_dev,
                          int relu, float* out_dev, int n, int h, int w);
/* First convolution of the network under amp O2, straight from the fp32 NCHW image (hrnet.py:377-385: 3x3, stride 2, padding 1,
 * 3 -> 64 channels, folded BatchNorm, ReLU): out = act(conv(fp16(x); fp16(weight)) * scale + shift), channel-blocked fp16
 * [n][8][h/2][w/2][8].  weight = the fp32 tensor [64][3][3][3] itself (no packing).  Replaces mp_f16_to_c8 + mp_f16_conv2d_fwd: same
 * fp16 operands and fp32 accumulation, the 27 products per output summed in ONE k-step (tap-major) instead of nine - equal within one
 * fp16 rounding of the output.  h even, w a multiple of 32: MP_ERR_UNSUPPORTED otherwise. */
int mp_f16_stem_conv_fwd(const float* x_dev, const float* weight_dev, const float* scale_dev, const float* shift_dev, int relu,
                         void* out_c8_dev, int n, int h, int w, mp_stream_t stream);
int mp_plan_add_stem_conv_f16(mp_plan* plan, const float* x_dev, const float* weight_dev, const float* scale_dev, const float* shift_dev,
                              int relu, void* out_c8_dev, int n, int h, int w);
/* Two chained fp16 1x1 convolutions in ONE launch (hrnet.py:107-123, 126-146, stage 1): the expand conv of Bottleneck i,
 *   y = act3(conv1x1(mid; w3) * scale3 + shift3 + res)      cm -> ce channels, res = the block's identity,
 * and the reduce conv of Bottleneck i + 1 on it,
 *   z = act1(conv1x1(y; w1) * scale1 + shift1)               ce -> cr channels,
 * y written once and never read back from HBM.  Channel-blocked fp16 tensors, weights packed by mp_f16_pack_weight (1x1).  y and z
 * are bit-identical to two mp_f16_conv2d_fwd launches.  Built for cm = 64, ce = 256, cr = 64 and h * w a multiple of 64:
 * MP_ERR_UNSUPPORTED otherwise (the caller launches the two convs). */
int mp_f16_expand_reduce_fwd(const void* mid_c8_dev, const void* res_c8_dev, const void* packed_w3_dev, const float* scale3_dev,
                             const float* shift3_dev, int relu3, const void* packed_w1_dev, const float* scale1_dev,
                             const float* shift1_dev, int relu1, void* y_c8_dev, void* z_c8_dev, int n, int cm, int ce, int cr, int h,
                             int w, mp_stream_t stream);
/* The same kernel with BOTH convs on one 64-channel input (the first Bottleneck of stage 1, hrnet.py:74-81, 107-123: its down-sample
 * conv ya = act_a(conv1x1(x; wa) * scale_a + shift_a), 64 -> 256, and its reduce conv zb = act_b(conv1x1(x; wb) * scale_b + shift_b),
 * 64 -> 64): x read once, one launch; bit-identical to the two launches. */
int mp_f16_dual_pw_fwd(const void* x_c8_dev, const void* packed_wa_dev, const float* scale_a_dev, const float* shift_a_dev, int relu_a,
                       const void* packed_wb_dev, const float* scale_b_dev, const float* shift_b_dev, int relu_b, void* ya_c8_dev,
                       void* zb_c8_dev, int n, int cm, int ce, int cr, int h, int w, mp_stream_t stream);
/* The chain with the FIRST Bottleneck's identity computed inside (hrnet.py:74-81, 107-146): y = act3(conv1x1(mid; w3) * scale3 + shift3 + d),
 * d = conv1x1(x0; wd) * scale_d + shift_d rounded to fp16 (the block's down-sample conv of its 64-channel input, no ReLU), then z as
 * in mp_f16_expand_reduce_fwd.  d is neither written nor read back; y and z are bit-identical to the three mp_f16_conv2d_fwd launches. */
int mp_f16_ds_expand_reduce_fwd(const void* mid_c8_dev, const void* x0_c8_dev, const void* packed_wd_dev, const float* scale_d_dev,
                                const float* shift_d_dev, const void* packed_w3_dev, const float* scale3_dev, const float* shift3_dev,
                                int relu3, const void* packed_w1_dev, const float* scale1_dev, const float* shift1_dev, int relu1,
                                void* y_c8_dev, void* z_c8_dev, int n, int cm, int ce, int cr, int h, int w, mp_stream_t stream);
int mp_plan_add_ds_expand_reduce_f16(mp_plan* plan, const void* mid_c8_dev, const void* x0_c8_dev, const void* packed_wd_dev,
                                     const float* scale_d_dev, const float* shift_d_dev, const void* packed_w3_dev,
                                     const float* scale3_dev, const float* shift3_dev, int relu3, const void* packed_w1_dev,
                                     const float* scale1_dev, const float* shift1_dev, int relu1, void* y_c8_dev, void* z_c8_dev, int n,
                                     int cm, int ce, int cr, int h, int w);
int mp_plan_add_dual_pw_f16(mp_plan* plan, const void* x_c8_dev, const void* packed_wa_dev, const float* scale_a_dev,
                            const float* shift_a_dev, int relu_a, const void* packed_wb_dev, const float* scale_b_dev,
                            const float* shift_b_dev, int relu_b, void* ya_c8_dev, void* zb_c8_dev, int n, int cm, int ce, int cr, int h, int w);
int mp_plan_add_expand_reduce_f16(mp_plan* plan, const void* mid_c8_dev, const void* res_c8_dev, const void* packed_w3_dev,
                                  const float* scale3_dev, const float* shift3_dev, int relu3, const void* packed_w1_dev,
                                  const float* scale1_dev, const float* shift1_dev, int relu1, void* y_c8_dev, void* z_c8_dev, int n,
                                  int cm, int ce, int cr, int h, int w);
/* The fp32 form of the chain (csrc/pwchain_f32.hip; hrnet.py:86-146 Bottleneck, 440-470 layer1): expand conv of Bottleneck i
 *   y = relu(conv1x1(mid; w3) * scale3 + shift3 + r)        64 -> 256 channels
 * and reduce conv of Bottleneck i + 1
 *   z = relu(conv1x1(y; w1) * scale1 + shift1)               256 -> 64 channels
 * in one persistent launch with all weight matrices held in registers; NCHW fp32 tensors, weights packed by mp_conv_pack_weight
 * (1x1), y written once and never read back.  The residual r is EITHER the tensor res_dev (x0_dev / packed_wd_dev / scale_d_dev /
 * shift_d_dev null) OR the block's down-sample conv conv1x1(x0; wd) * scale_d + shift_d of its 64-channel input (hrnet.py:74-81),
 * computed inside the launch (res_dev null).  packed_w1_dev / scale1_dev / shift1_dev / z_dev null = the expand conv alone (the last
 * Bottleneck of the stage; not with the down-sample form).  Same values as the stand-alone mp_conv2d_fwd launches up to fp32
 * rounding (the k sums are associated differently).  Built for cm = 64, ce = 256, cr = 64 and h * w a multiple of 64:
 * MP_ERR_UNSUPPORTED otherwise (the caller launches the convs one by one). */
int mp_expand_reduce_fwd(const float* mid_dev, const float* res_dev, const float* x0_dev, const float* packed_wd_dev,
                         const float* scale_d_dev, const float* shift_d_dev, const float* packed_w3_dev, const float* scale3_dev,
                         const float* shift3_dev, const float* packed_w1_dev, const float* scale1_dev, const float* shift1_dev,
                         float* y_dev, float* z_dev, int n, int cm, int ce, int cr, int h, int w, mp_stream_t stream);
int mp_plan_add_expand_reduce(mp_plan* plan, const float* mid_dev, const float* res_dev, const float* x0_dev, const float* packed_wd_dev,
                              const float* scale_d_dev, const float* shift_d_dev, const float* packed_w3_dev, const float* scale3_dev,
                              const float* shift3_dev, const float* packed_w1_dev, const float* scale1_dev, const float* shift1_dev,
                              float* y_dev, float* z_dev, int n, int cm, int ce, int cr, int h, int w);
int mp_plan_add_basicblock_f16(mp_plan* plan, const void* x_c8_dev, const void* packed_w1_dev, const float* scale1_dev,
                               const float* shift1_dev, const void* packed_w2_dev, const float* scale2_dev, const float* shift2_dev,
                               void* out_c8_dev, int n, int c, int h, int w, int rows);
int mp_f16_fuse_upsample_sum(const void* base_dev, const void* t1_dev, int s1, const void* t2_dev, int s2, const void* t3_dev,
                             int s3, void* out_dev, int n, int c, int h, int w, int relu, mp_stream_t stream);
int mp_plan_add_conv_f16(mp_plan* plan, const mp_conv_desc* desc, int variant, const void* x_c8_dev, const void* packed_w_dev,
                         const float* scale_dev, const float* shift_dev, const void* res1_c8_dev, const void* res2_c8_dev,
                         void* out_c8_dev);
int mp_plan_add_fuse_sum_f16(mp_plan* plan, const void* base_dev, const void* t1_dev, int s1, const void* t2_dev, int s2,
                             const void* t3_dev, int s3, void* out_dev, int n, int c, int h, int w, int relu);
/* to_c8 != 0: NCHW fp32 -> c8 fp16, else c8 fp16 -> NCHW fp32 */
int mp_plan_add_layout_f16(mp_plan* plan, int to_c8, const void* x_dev, void* out_dev, int n, int c, int h, int w);

/* ---- loader side: the crop that feeds the network -----------------------------------------------------
 * cv2.warpAffine(image, trans, (out_w, out_h), flags=cv2.INTER_LINEAR) of TopDownAffine._affine / _udp_affine
 * (mindpose/data/transform/topdown_transform.py:198-262), optionally fused with vision.Normalize(mean, std) +
 * vision.HWC2CHW (mindpose/data/data_factory.py:129-133).
 *   src          device buffer holding the source images, HWC uint8, 3 channels
 *   src_offsets  [n] int64 (device): byte offset of crop i's source image inside src
 *   src_hw       [n][2] int32 (device): height, width of crop i's source image
 *   flip         NULL or [n] int32 (device): non-zero = crop i is taken from the horizontally flipped source image
 *                (cv2.flip(image, 1) of TopDownHorizontalRandomFlip :481, applied while sampling: no flipped copy is made)
 *   trans        [n][6] fp64 (device): the 2x3 FORWARD matrices (source -> crop) exactly as get_affine_transform /
 *                get_warp_matrix return them; the kernel inverts them as cv2.warpAffine does
 *   normalize=1: out = fp32 [n][3][out_h][out_w], (pixel - mean[c]) / stddev[c]   (mean/std already times 255, host arrays)
 *   normalize=0: out = uint8 [n][out_h][out_w][3], the warped image only
 * Border mode BORDER_CONSTANT(0); OpenCV's fixed-point bilinear arithmetic restated (parity unpinned: no cv2 here). */
int mp_warp_affine(const uint8_t* src_dev, const long long* src_offsets_dev, const int* src_hw_dev, const int* flip_dev,
                   const double* trans_dev, void* out_dev, int n, int out_h, int out_w, int normalize, const float mean[3], const float stddev[3],
                   mp_stream_t stream);

/* Horizontal flip of an NCHW fp32 batch, out[n,c,y,x] = in[n,c,y,W-1-x]: the input of the flip test's second run
 * (mindpose/engine/inferencer/topdown_inferencer.py:168-170), one pass; in and out must not overlap. */
int mp_flip_width(const float* in_dev, float* out_dev, int n, int c, int h, int w, mp_stream_t stream);

/* fp16 (amp O2) training passes over channel-blocked fp16 activations; same contracts as mp_bn_train_fwd / _bwd and
 * mp_fuse_upsample_sum_bwd (statistics, gamma / beta gradients and the workspace stay fp32 / fp64; mp_bn_workspace_bytes) */
int mp_f16_bn_train_fwd(const void* z_dev, const float* gamma_dev, const float* beta_dev, const void* res_dev, void* y_dev,
                        float* save_mean_dev, float* save_invstd_dev, float* moving_mean_dev, float* moving_var_dev, int n, int c,
                        int hw, float eps, float momentum, int relu, void* workspace_dev, size_t workspace_bytes, mp_stream_t stream);
/* dgamma_acc / dbeta_acc (both or neither, may be NULL): the gamma / beta gradients are ALSO added into these buffers (the
 * caller's gradient arena), which saves the framework's separate accumulate launch.
 * ReLU mask (relu != 0): taken from the stored forward output y_dev - or, for a layer WITHOUT residual input (dres_dev NULL),
 * re-derived from z with the forward arithmetic when y_dev is NULL and beta_dev is given (y is then not read: two tensor
 * reads less).  beta_dev may be NULL when y_dev is passed. */
int mp_f16_bn_train_bwd(const void* dy_dev, const void* z_dev, const void* y_dev, const float* gamma_dev, const float* beta_dev,
                        const float* save_mean_dev, const float* save_invstd_dev, void* dz_dev, void* dres_dev, float* dgamma_dev,
                        float* dbeta_dev, float* dgamma_acc_dev, float* dbeta_acc_dev, int n, int c, int hw, int relu,
                        void* workspace_dev, size_t workspace_bytes, mp_stream_t stream);
int mp_f16_fuse_upsample_sum_bwd(const void* dy_dev, const void* out_dev, void* dbase_dev, void* dt1_dev, int s1, void* dt2_dev,
                                 int s2, void* dt3_dev, int s3, int n, int c, int h, int w, int relu, mp_stream_t stream);

/* ---- BatchNorm fused out of the amp-O2 training step (round 3) ------------------------------------------------------------------
 * Reference: every conv of mindpose/models/backbones/hrnet.py:51-64 is followed by nn.BatchNorm2d, trained under amp O2
 * (tools/train.py:170-181).  MindSpore runs conv, the batch-statistics reduction and the normalisation as separate cells; here the
 * conv launch that PRODUCES a tensor also produces the per-channel partial sums its BatchNorm needs, so the reduction passes
 * over the tensor disappear in both directions:
 *   mode 1 (forward):  the launch computes z = conv(x); partials of  sum z, sum z*z  over the fp16-rounded output.
 *   mode 2 (backward): the launch computes the gradient dy reaching a BatchNorm's OUTPUT (a stride-1 data-gradient conv; res1 = the
 *                      residual gradient, as in mp_f16_conv2d_fwd).  z_dev = that BatchNorm's input, y_dev its output (relu != 0:
 *                      mask = y > 0), both in the output geometry.  The tensor STORED is g = dy * mask, and the partials are
 *                      sum g, sum g * z (raw z: the consumer forms sum g * xhat = invstd * (sum g z - mean * sum g) in fp64).
 *                      Kernel 1x1 or 3x3, plain output mapping.  mean / invstd / gamma / beta of the struct are unused by the conv.
 * partials: fp32 [ceil(cout/8)][n_parts][8][2], n_parts = mp_f16_conv_stats_parts(desc, variant) (one slot per pixel-tile workgroup;
 * variant -1 = the library's deterministic choice, the same in both calls; 0 = this variant / shape has no statistics build: use
 * the plain entry + the reduction passes).  Fixed summation order, no
 * atomics: bit-reproducible.  The conv output is bit-identical to mp_f16_conv2d_fwd's (mode 2: times the mask). */
typedef struct mp_f16_conv_stats {
    int mode, relu;
    float* partials_dev;
    size_t partials_bytes;
    const void* z_dev;
    const void* y_dev;
    const float* mean_dev;
    const float* invstd_dev;
    const float* gamma_dev;
    const float* beta_dev;
    /* mode 1 only - BatchNorm apply of the layer BELOW on this conv's input operand (hrnet.py:67-72: conv1 -> bn1 -> relu -> conv2
     * without a pass over y1): x_c8_dev holds the RAW output z of the conv below, pre_scale / pre_shift its folded batch statistics
     * (mp_f16_bn_train_finalize: [round_up(cin, 32)] floats, zeros behind cin); the kernel convolves
     * act(z * scale + shift) - the same fp32 fma, ReLU and single fp16 rounding as mp_f16_bn_train_fwd_stats - and, when
     * pre_out_dev is given, also stores that activation tensor (geometry of x; the backward pass reads it).  Bit-identical to the
     * apply pass followed by the plain launch.  3x3 stride-1 weights-in-registers variants only: MP_ERR_UNSUPPORTED otherwise. */
    const float* pre_scale_dev;
    const float* pre_shift_dev;
    void* pre_out_dev;
    int pre_relu;
} mp_f16_conv_stats;
int mp_f16_conv_stats_parts(const mp_conv_desc* desc, int variant);
/* 1 when mp_f16_conv2d_fwd_stats(desc, variant, ...) accepts stats->pre_scale_dev (BatchNorm apply of the layer below on the input
 * operand): an explicit variant of the weights-in-registers family on a 3x3 stride-1 layer staged one image per tile. */
int mp_f16_conv_pre_supported(const mp_conv_desc* desc, int variant);
/* 1 when mp_f16_conv2d_fwd (stats_mode 0; n_res = 0 / 1 / 2 residual tensors) or mp_f16_conv2d_fwd_stats (stats_mode 1 / 2; n_res
 * 0 / 1) would ACCEPT this (shape, tile variant) - the entry's own checks and the kernel family's own dispatch run without the
 * launch; host-only, no device work.  0 = the entry would answer MP_ERR_UNSUPPORTED (or a shape error).  The test matrix is
 * generated from this query, so a pair that stops being served shows up as a failing floor test instead of a skip. */
int mp_f16_conv_supported(const mp_conv_desc* desc, int variant, int n_res, int stats_mode);
int mp_f16_conv2d_fwd_stats(const mp_conv_desc* desc, int variant, const void* x_c8_dev, const void* packed_w_dev,
                            const float* scale_dev, const float* shift_dev, const void* res1_c8_dev, void* out_c8_dev,
                            const mp_f16_conv_stats* stats, mp_stream_t stream);
/* The consumers.  mp_f16_bn_train_fwd_stats: mp_f16_bn_train_fwd without its reduction pass (statistics from the mode-1 partials;
 * one launch: every workgroup folds the partials of its 8 channels itself, in a fixed order; above 512 slots a small fold
 * launch runs first).  mp_f16_bn_train_bwd_stats: dy_dev is the PRE-MASKED gradient g a mode-2 launch stored;
 *   dz = gamma * invstd * (g - mean(g) - xhat * mean(g * xhat));  the residual branch's gradient is g itself (no dres tensor).
 * workspace: mp_bn_workspace_bytes(c). */
int mp_f16_bn_train_fwd_stats(const void* z_dev, const float* gamma_dev, const float* beta_dev, const void* res_dev, void* y_dev,
                              float* save_mean_dev, float* save_invstd_dev, float* moving_mean_dev, float* moving_var_dev, int n,
                              int c, int hw, float eps, float momentum, int relu, const float* partials_dev, int n_parts,
                              void* workspace_dev, size_t workspace_bytes, mp_stream_t stream);
/* The statistics half of mp_f16_bn_train_fwd_stats alone (reference: nn.BatchNorm2d in training mode, hrnet.py:51-64): saved mean /
 * invstd, moving averages and the folded scale_dev / shift_dev ([ceil(c/8)*8] floats, zeros on padding channels) for a consumer
 * that applies the BatchNorm on its own operand (mp_f16_conv_stats.pre_scale_dev).  Same fold order and arithmetic as the apply
 * pass: the two routes give bit-identical activations. */
int mp_f16_bn_train_finalize(const float* gamma_dev, const float* beta_dev, float* save_mean_dev, float* save_invstd_dev,
                             float* moving_mean_dev, float* moving_var_dev, int n, int c, int hw, float eps, float momentum,
                             const float* partials_dev, int n_parts, float* scale_dev, float* shift_dev, void* workspace_dev,
                             size_t workspace_bytes, mp_stream_t stream);
/* The two element-wise producers of a BatchNorm's output gradient with the same statistics (partials [ceil(c/8)][n_parts][8][2],
 * n_parts = mp_f16_ew_stats_parts(n, c, hw of the OUTPUT tensor)):
 *   mp_f16_sum_tensors_stats: out = fp16(((a + b) + c) + d) * [y > 0] - mp_sum_tensors (the gradients of the consumers of a branch
 *     output, hrnet.py:327-339) followed by the mask; sums of g, g * z.
 *   mp_f16_fuse_sum_bwd_term_stats: ONE term of mp_f16_fuse_upsample_sum_bwd (scale s; s = 1: the base term), masked with y_t > 0 when
 *     relu_t; z_t / y_t: input / output of the BatchNorm that produced the term. */
int mp_f16_ew_stats_parts(int n, int c, int hw);
int mp_f16_fuse_term_stats_parts(int n, int c, int h, int w, int s); /* n_parts of mp_f16_fuse_sum_bwd_term_stats (h, w: full resolution) */
int mp_f16_sum_tensors_stats(const void* a_dev, const void* b_dev, const void* c_dev, const void* d_dev, void* out_dev, const void* z_dev,
                             const void* y_dev, int relu, int n, int c, int hw, float* partials_dev, size_t partials_bytes,
                             mp_stream_t stream);
int mp_f16_fuse_sum_bwd_term_stats(const void* dy_dev, const void* out_dev, void* dt_dev, int s, int n, int c, int h, int w, int relu,
                                   const void* z_t_dev, const void* y_t_dev, int relu_t, float* partials_dev, size_t partials_bytes,
                                   mp_stream_t stream);
int mp_f16_bn_train_bwd_stats(const void* g_dev, const void* z_dev, const float* gamma_dev, const float* save_mean_dev,
                              const float* save_invstd_dev, void* dz_dev, float* dgamma_dev, float* dbeta_dev, float* dgamma_acc_dev,
                              float* dbeta_acc_dev, int n, int c, int hw, const float* partials_dev, int n_parts,
                              void* workspace_dev, size_t workspace_bytes, mp_stream_t stream);

/* Up to FOUR apply-only BatchNorm passes as ONE launch: the k-th BatchNorm of every branch of an HRModule (hrnet.py:202-241) -
 * independent tensors of different shapes, each ~10 us of launch, fold and tail around a few us of streaming when launched alone.
 * A job carries exactly the arguments of mp_f16_bn_train_fwd_stats / mp_f16_bn_train_bwd_stats (same results, bit for bit). */
typedef struct mp_f16_bn_fwd_job {
    const void* z_dev; const float* gamma_dev; const float* beta_dev; const void* res_dev; void* y_dev;
    float* save_mean_dev; float* save_invstd_dev; float* moving_mean_dev; float* moving_var_dev;
    const float* partials_dev; void* workspace_dev; size_t workspace_bytes;
    int n, c, hw, relu, n_parts, reserved;
} mp_f16_bn_fwd_job;
typedef struct mp_f16_bn_bwd_job {
    const void* g_dev; const void* z_dev; const float* gamma_dev; const float* save_mean_dev; const float* save_invstd_dev; void* dz_dev;
    float* dgamma_dev; float* dbeta_dev; float* dgamma_acc_dev; float* dbeta_acc_dev;
    const float* partials_dev; void* workspace_dev; size_t workspace_bytes;
    int n, c, hw, n_parts;
} mp_f16_bn_bwd_job;
int mp_f16_bn_train_fwd_stats_grouped(const mp_f16_bn_fwd_job* jobs, int n_jobs, float eps, float momentum, mp_stream_t stream);
int mp_f16_bn_train_bwd_stats_grouped(const mp_f16_bn_bwd_job* jobs, int n_jobs, mp_stream_t stream);

/* out = a + b (+ c) (+ d) over `bytes` bytes (a multiple of 16) of fp32 (half = 0) or fp16 (half = 1, fp32 sums, one rounding):
 * the fan-in of gradients at a tensor with several consumers - every branch output of an HRModule feeds every exchange-unit row
 * (hrnet.py:318-344) - in one pass instead of the framework's k - 1 pairwise adds.  out may alias an input. */
int mp_sum_tensors(const void* a_dev, const void* b_dev, const void* c_dev, const void* d_dev, void* out_dev, size_t bytes, int half,
                   mp_stream_t stream);
/* weight gradient of a conv (kernel 1x1 or 3x3, stride 1 or 2, padding k/2) from channel-blocked fp16 x and dz on the fp16
 * matrix cores, fp32 accumulation and fp32 result dw [Cout,Cin,kh,kw] (times `scale`: pass 1 / loss_scale); accumulate != 0
 * adds into dw (the caller's gradient arena) instead of overwriting; workspace from mp_f16_conv_wgrad_workspace_bytes */
size_t mp_f16_conv_wgrad_workspace_bytes(const mp_conv_desc* desc);
int mp_f16_conv_wgrad(const mp_conv_desc* desc, const void* x_c8_dev, const void* dz_c8_dev, float* dw_dev, float scale,
                      int accumulate, void* workspace_dev, size_t workspace_bytes, mp_stream_t stream);
/* The same for 1 .. 8 layers of ONE shape in a single launch pair (weight gradients are leaves of the backward pass, tools/train.py:233:
 * nothing waits for them before the update, so the layers of an HRNet branch - hrnet.py:202-241, eight 3x3 convs of one shape - are
 * collected and run together): per layer fewer, longer pixel slabs, 1/n of the launches.  x / dz / dw: HOST arrays of n_jobs device
 * pointers (they travel in the kernel arguments: no upload, hipGraph-capturable).  Same fixed-order reduction per layer as the
 * single-layer entry; a layer's result does not depend on its group (slab partition aside: fp32 summation order). */
size_t mp_f16_conv_wgrad_grouped_workspace_bytes(const mp_conv_desc* desc, int n_jobs);
int mp_f16_conv_wgrad_grouped(const mp_conv_desc* desc, const void* const* x_dev, const void* const* dz_dev, float* const* dw_dev,
                              int n_jobs, float scale, int accumulate, void* workspace_dev, size_t workspace_bytes, mp_stream_t stream);

/* ---- training kernels of the SimpleBaseline-ResNet family ---------------------------------------------------------------
 * mp_maxpool3x3s2_same_bwd: backward of nn.MaxPool2d(3, 2, pad_mode="same") (resnet.py:190), gradient to the first maximum of
 *   each window in scan order; fp32 NCHW.
 * mp_stem_conv_wgrad: weight gradient of the k x k (k = 7 or 3) stride-2 padding-k/2 stem conv with <= 4 input channels
 *   (resnet.py:180-188), fp32 NCHW x [n,cin,h,w], dz [n,cout,ho,wo] -> dw [cout,cin,k,k].
 * mp_f16_gather_phase: out[n,c,m,k] = x[n,c,2m+phase_y,2k+phase_x] on channel-blocked fp16 (x is [n,c,2h,2w]): the sub-pixel
 *   phase of a gradient that the 2x2 phase kernels of the transposed convolution (mp_f16_pack_weight mode 4) consume.
 * The transposed convolution's weight gradient is mp_f16_conv_wgrad on a 4x4 stride-2 padding-1 descriptor with the roles
 *   (x := dy at the up-sampled resolution, dz := the layer input): dW[cin_t][cout_t][ky][kx] = sum x_t[m] * dy[2m+ky-1]. */
int mp_maxpool3x3s2_same_bwd(const float* x_dev, const float* dy_dev, float* dx_dev, int n, int c, int h, int w, mp_stream_t stream);
int mp_stem_conv_wgrad(const float* x_dev, const float* dz_dev, float* dw_dev, int n, int cin, int h, int w, int cout, int k,
                       mp_stream_t stream);
int mp_f16_gather_phase(const void* x_c8_dev, void* out_c8_dev, int n, int c, int h, int w, int phase_y, int phase_x,
                        mp_stream_t stream);
/* the same gather on fp32 NCHW; the fp32 transposed-conv gradients use it with mp_conv_pack_weight mode 4 (data gradient of
 * phase (phase_y, phase_x)) and mp_conv_wgrad on a 4x4 stride-2 padding-1 descriptor with the roles exchanged */
int mp_gather_phase(const float* x_dev, float* out_dev, int n, int c, int h, int w, int phase_y, int phase_x, mp_stream_t stream);

/* Diagnostics: only a library built with -DMP_CONV_STAMPS=1 (never the product build) records per-workgroup
 * phase cycle counters (8 x uint64 per workgroup) of each conv launch into this device buffer; the product
 * build returns MP_ERR_UNSUPPORTED. */
int mp_debug_set_stamp_buffer(void* dev_ptr, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* MINDPOSE_HIP_H */
