/*
 * tdrn_hip.h -- C ABI of libtdrn_hip.so: the MI355X (gfx950) inference path of TDRN's
 * dual-refinement detector.  Plain pointers and sizes only (no torch / THC types).
 *
 * Conventions:
 *   - return 0 (TDRN_OK) on success; NEGATIVE = argument / shape error (the reference's
 *     shape_check / THArgCheck cases); POSITIVE = hipError_t of a failed HIP call.  Nothing
 *     is printf'ed-and-ignored (the reference does: deform_conv_cuda_kernel.cu:233-237,
 *     nms_kernel.cu:12-19).
 *   - every device buffer is owned by the caller (PyTorch-ROCm allocates them); functions that
 *     need scratch take a (workspace, workspace_bytes) pair sized by the matching
 *     *_workspace_bytes() query.  No hipMalloc / hipFree / device sync inside the hot calls
 *     (so they can be captured into a hipGraph); the two exceptions are marked COMPAT.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); all work is
 *     enqueued on it and the call returns without synchronising unless stated.
 *   - thread-safe per (handle, stream).  The only process-wide state is idempotent memoisation: environment
 *     switches (TDRN_*) read once, and "dynamic LDS limit raised" per (kernel, device).
 *
 * Each entry point cites the reference interface it replaces (paths relative to the upstream
 * SeanChenxy/TDRN tree).
 */
#ifndef TDRN_HIP_H
#define TDRN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TDRN_API __attribute__((visibility("default")))

/* ---- status codes ---------------------------------------------------------------------- */
#define TDRN_OK 0
#define TDRN_E_ARG (-1)          /* null pointer / bad enum / bad size                          */
#define TDRN_E_SHAPE (-2)        /* shape_check failure (deform_conv_cuda.c:7-96)               */
#define TDRN_E_WORKSPACE (-3)    /* workspace too small                                         */
#define TDRN_E_UNSUPPORTED (-4)  /* configuration outside what the kernels cover                */
#define TDRN_E_PARAM (-5)        /* unknown / missing / mis-shaped state_dict entry             */
#define TDRN_E_STATE (-6)        /* call order (e.g. forward before weights were packed)        */
#define TDRN_E_VALUE (-7)        /* Detect: nms_thresh <= 0 (layers/functions/detection.py:20)  */
#define TDRN_E_DEVICE (-8)       /* a device-side hand-off of an earlier forward timed out: its outputs are invalid (tdrn_net_check) */

TDRN_API const char *tdrn_version(void);
TDRN_API const char *tdrn_error_string(int code);

/* arithmetic type of the dense path (activations + weights in HBM, MFMA input type);
 * accumulation is always fp32; box decoding / NMS / offsets / bilinear weights always fp32. */
typedef enum { TDRN_F32 = 0, TDRN_BF16 = 1, TDRN_F16 = 2 } tdrn_dtype;

/* ========================================================================================
 * (i) Deformable convolution v1 forward -- replaces
 *     int deform_conv_forward_cuda(THCudaTensor *input, *weight, *offset, *output, *columns,
 *         *ones, int kW, int kH, int dW, int dH, int padW, int padH, int dilationH,
 *         int dilationW, int deformable_group)          utils/deformconv/deform_conv_cuda.h:1-7
 *     called from ConvOffset2dFunction.forward          model/networks.py:641-645
 *     kernel semantics                                  utils/deformconv/deform_conv_cuda_kernel.cu:15-51,156-208
 *   input  (N, Cin, H, W)  fp32 NCHW contiguous, device
 *   weight (Cout, Cin, kH, kW) fp32 contiguous, device        (no bias, like the reference)
 *   offset (N, G*2*kH*kW, Ho, Wo) fp32, device
 *   output (N, Cout, Ho, Wo) fp32, device, fully overwritten
 *   `columns` / `ones` of the reference have no counterpart: the sampled columns never leave
 *   LDS.  `compute` selects the MFMA input type (TDRN_F32 reproduces the reference's fp32).
 *   Kernel size, stride, padding and dilation are per axis, as in the reference (any kH x kW).
 *   Shape errors mirror shape_check (deform_conv_cuda.c:7-96, :137) -> TDRN_E_SHAPE.
 * ====================================================================================== */
TDRN_API size_t tdrn_deform_conv_workspace_bytes(int N, int Cin, int H, int W, int Cout, int kH,
                                                 int kW, int dH, int dW, int padH, int padW,
                                                 int dilationH, int dilationW,
                                                 int deformable_group, tdrn_dtype compute);
TDRN_API int tdrn_deform_conv_forward(const float *input, const float *weight,
                                      const float *offset, float *output, int N, int Cin, int H,
                                      int W, int Cout, int kW, int kH, int dW, int dH, int padW,
                                      int padH, int dilationH, int dilationW,
                                      int deformable_group, tdrn_dtype compute, void *workspace,
                                      size_t workspace_bytes, void *stream);

/* ========================================================================================
 * (ii) NMS / box utilities / Detect
 * ====================================================================================== */

/* Greedy NMS with the CPU semantics Detect uses -- replaces
 *     cpu_nms(ndarray[float32, ndim=2] dets, float thresh) -> list   utils/nms/cpu_nms.pyx:17-68
 *     (dispatcher utils/nms_wrapper.py:23-31, called at layers/functions/detection.py:60)
 *   dets (n,5) device fp32 rows [x1,y1,x2,y2,score] in pixel units ("+1" convention), ANY order;
 *   any n (above 16384 boxes the sort keys live in the workspace instead of LDS);
 *   sorted on device by (score desc, index asc).  Suppression when IoU >= thresh
 *   (strict_gt = 0, cpu_nms.pyx:66) or IoU > thresh (strict_gt = 1, nms_kernel.cu:71); the IoU
 *   is fp32 and compared against the double `thresh` exactly as the Cython code does.
 *   keep_out (n) device int32 <- kept indices into dets, descending score; num_out (1) device. */
TDRN_API size_t tdrn_nms_workspace_bytes(int n);
TDRN_API int tdrn_nms(const float *dets, int n, double thresh, int strict_gt, int32_t *keep_out,
                      int32_t *num_out, void *workspace, size_t workspace_bytes, void *stream);

/* The torch NMS that DetectOTA uses -- replaces
 *     nms(boxes, scores, overlap=0.5, top_k=200) -> (keep, count)       layers/box_utils.py:229-293
 *   dets (n,5) device fp32 rows [x1,y1,x2,y2,score], normalised coordinates, NO "+1": area = (x2-x1)*(y2-y1),
 *   IoU = inter / ((area_j - inter) + area_i) in fp32; a candidate survives iff IoU <= overlap (fp32).  Only boxes
 *   with score > min_score are candidates (the caller's c_mask, detection_ota.py:67-68) and only the top_k best
 *   of them enter (box_utils.py:251; 0 = all).  keep_out (n) <- kept indices into dets, descending score. */
TDRN_API int tdrn_nms_topk(const float *dets, int n, float overlap, float min_score, int top_k,
                           int32_t *keep_out, int32_t *num_out, void *workspace,
                           size_t workspace_bytes, void *stream);

/* The per-class loop of DetectOTA.forward (layers/functions/detection_ota.py:61-79: for cl in 1..C-1: c_mask, nms(boxes,
 * scores, nms_thresh, top_k)) as ONE launch: boxes (n,4) shared by all classes, scores (n, num_classes) row-major (the softmax
 * output of one frame); class c in [first_class, num_classes) runs tdrn_nms_topk's rule on (boxes, scores[:, c]).
 * keep_out (num_classes, n) int32 and num_out (num_classes): rows below first_class are not written.  n <= 16384
 * (TDRN_E_UNSUPPORTED beyond: call tdrn_nms_topk per class).  No host round trip: the caller reads num_out once. */
TDRN_API size_t tdrn_nms_topk_classes_workspace_bytes(int n, int num_classes);
TDRN_API int tdrn_nms_topk_classes(const float *boxes, const float *scores, int n, int num_classes, int first_class,
                                   float overlap, float min_score, int top_k, int32_t *keep_out, int32_t *num_out,
                                   void *workspace, size_t workspace_bytes, void *stream);

/* DetectOTA's association arithmetic on the device (the reference does it with torch ops on the GPU; here it is two kernels).
 * tdrn_roi_resample -- detection_ota.py:86-93: for box b the cell range [x0,x1) x [y0,y1) of the (C,H,W) fp32 feature map
 *   (cells (n,4) int32 = x0,y0,x1,y1, computed by the caller as the reference does: floor / ceil of box * size, clipped) is resampled
 *   to S x S with F.upsample(mode='bilinear', align_corners=True)'s arithmetic and flattened as (C,S,S): out (n, C*S*S) fp32.
 * tdrn_ota_similarity -- detection_ota.py:95-99 with box_utils.IoU / cos_similarity (layers/box_utils.py:295-367):
 *   sim[i][j] = exp(IoU(boxes[i], head_j)) * mean_r cos(roi[i], feature of row r of tubelet j); best[i] = max_j, arg[i] = its index.
 *   rows (R, 5+F) fp32: the stored rows [score, x1,y1,x2,y2, feature] of all m tubelets back to back, a tubelet's newest row (its
 *   head) first; row_off (m+1) int32: tubelet j = rows [row_off[j], row_off[j+1]).  No host round trip. */
TDRN_API int tdrn_roi_resample(const float *feature, int C, int H, int W, const int32_t *cells, int n, int S, float *out,
                               void *stream);
TDRN_API int tdrn_ota_similarity(const float *boxes, const float *roi, int n, int F, const float *rows, const int32_t *row_off,
                                 int m, float *best, int32_t *arg, void *stream);

/* COMPAT twin of   void _nms(int* keep_out, int* num_out, const float* boxes_host,
 *     int boxes_num, int boxes_dim, float nms_overlap_thresh, int device_id)
 *                                                          utils/nms/gpu_nms.hpp:1-2
 *   HOST pointers in and out, boxes already sorted descending by the caller
 *   (gpu_nms.pyx:25-28), strict `>` threshold (nms_kernel.cu:71), synchronous, allocates and
 *   frees device memory per call like the reference (nms_kernel.cu:100-143).  Not hot path. */
TDRN_API int tdrn_gpu_nms_host(int *keep_out, int *num_out, const float *boxes_host,
                               int boxes_num, int boxes_dim, float nms_overlap_thresh,
                               int device_id);

/* decode / center_size -- layers/box_utils.py:176-195, :16-25.  Device fp32 (P,4) arrays. */
TDRN_API int tdrn_decode(const float *loc, const float *priors, int P, float var0, float var1,
                         float *boxes_out, void *stream);
TDRN_API int tdrn_center_size(const float *boxes, int P, float *out, void *stream);

/* PriorBox.forward -- layers/functions/prior_box.py:33-64 (host, double arithmetic, cast to
 * fp32, clamp).  aspect_ratios ragged: ar_count[k] values per map, concatenated in `ars`.
 * out == NULL: returns the number of priors only.  Returns P (>= 0) or a negative error. */
TDRN_API int tdrn_prior_box(int n_maps, const int *feature_maps, double image_size,
                            const double *steps, const double *min_sizes,
                            const double *max_sizes, int n_max_sizes, const int *ar_count,
                            const double *ars, int clip, int flip, float *out_host);

/* Detect.forward -- layers/functions/detection.py:25-70, fully on device, batched over
 * (image, class):  two-stage decode (:43-48), score > conf_thresh (:53), boxes*scale (:59),
 * cpu_nms semantics (:60), first top_k survivors packed as [score,x1,y1,x2,y2] (:61-63).
 *   loc (B,P,4)  conf (B*P,C)  priors (P,4)  arm_loc (B,P,4) or NULL  -- device fp32
 *   scale: 4 HOST floats (the caller's [w,h,w,h]; evaluate.py:461)
 *   out (B,C,top_k,5) device fp32, fully overwritten (class 0 and unused slots = 0)
 *   counts_out (B*C) device int32 or NULL: survivors per (image,class), capped at top_k. */
/* Any P (no LDS-derived limit: the 1216-pixel scale of multi_eval.py:24 has P = 92055).
 * tdrn_detect_dev_scale: the same with `scale` as 4 DEVICE floats (16-byte aligned) -- evaluate.py:461 builds
 * the scale as a CUDA tensor; reading it on the device keeps the call free of host synchronisation. */
TDRN_API size_t tdrn_detect_workspace_bytes(int B, int P, int C, int top_k);
TDRN_API int tdrn_detect(const float *loc, const float *conf, const float *priors,
                         const float *arm_loc, const float *scale_host, int B, int P, int C,
                         int top_k, float conf_thresh, double nms_thresh, float *out,
                         int32_t *counts_out, void *workspace, size_t workspace_bytes,
                         void *stream);
TDRN_API int tdrn_detect_dev_scale(const float *loc, const float *conf, const float *priors,
                                   const float *arm_loc, const float *scale_dev, int B, int P, int C,
                                   int top_k, float conf_thresh, double nms_thresh, float *out,
                                   int32_t *counts_out, void *workspace, size_t workspace_bytes,
                                   void *stream);

/* Preprocess -- replaces base_transform + the channel swap of
 *     data/__init__.py:7-12 (cv2.resize(image,(S,S)) -> float32 -> -= mean) and data/voc0712.py:467-468
 *     (BGR -> RGB, HWC -> CHW); test_video.py:103-105 skips the swap (to_rgb = 0).
 *   frames (B, H0, W0, 3) uint8 BGR device; out (B, 3, S, S) fp32 NCHW device.
 *   Resize = OpenCV's INTER_LINEAR for 8-bit images (half-pixel centres, 11-bit fixed-point
 *   coefficients, uint8 result) restated from imgproc/resize.cpp; cv2 itself is not available in
 *   the build image, so this piece is "parity unpinned" by the reference: oracle.base_transform_u8 is
 *   pinned by hand-worked known-answer cases and by staying within one grey level of torch's independent floating-point
 *   bilinear (tests/test_oracle_pin.py); the device kernel is bit-exact against that oracle. */
TDRN_API int tdrn_preprocess(const uint8_t *frames, int B, int H0, int W0, int S, const float mean_bgr[3],
                             int to_rgb, float *out, void *stream);
/* The same resize, result left as cv2.resize returns it -- uint8 -- in planes: out (B, 3, S, S) uint8 device, channel-swapped when
 * to_rgb.  Together with tdrn_net_io.reserved[3] (below) this is SURVEY 8f rank 1 in full: the frame stays uint8 until the first conv's
 * loader reads it and subtracts the mean there ("fused into the first conv's loader"); no fp32 copy of the batch is ever written.
 * float(uint8) - mean is exact, so both routes give the net bit-identical inputs (tests/test_gpu_net.py). */
TDRN_API int tdrn_preprocess_u8(const uint8_t *frames, int B, int H0, int W0, int S, int to_rgb, uint8_t *out, void *stream);

/* ========================================================================================
 * (iii) Whole-network forward -- replaces build_net(...)/RefineSSD.forward:
 *     model/dualrefinedet_vggbn.py:10-117,119-206,217-222   (TDRN_DRN_VGGBN)
 *     model/dualrefinedet_mobilenet.py:8-125,127-199        (TDRN_DRN_MOBILENET)
 *     model/ssd4scale_mobile.py:9-84,86-140                 (TDRN_SSD4SCALE_MOBILE)
 *     model/refinedet_vgg.py:112-219                        (TDRN_REFINEDET_VGG)
 *     model/ssd4scale_vgg.py:71-135                         (TDRN_SSD4SCALE_VGG)
 * ====================================================================================== */
typedef enum {
    TDRN_DRN_VGGBN = 0,
    TDRN_DRN_MOBILENET = 1,
    TDRN_SSD4SCALE_MOBILE = 2,
    TDRN_REFINEDET_VGG = 3,
    TDRN_SSD4SCALE_VGG = 4
} tdrn_model;

typedef struct {
    int model;        /* tdrn_model                                                         */
    int size;         /* 320 or 512 (build_net rejects anything else)                       */
    int num_classes;  /* 21                                                                 */
    int c7_channel;   /* 1024                                                               */
    int def_groups;   /* deformable groups of the ODM heads (1)                             */
    int bn;           /* VGG trunk with BatchNorm                                           */
    int multihead;    /* 5x5 deformable heads summed with the 3x3 ones                      */
    int deform;       /* ssd4scale_*: deformable ARM heads (df_group = 8), TRN temporal net */
    int test_phase;   /* 1: softmax on conf (phase == 'test'); 0: raw logits                */
    int dtype;        /* tdrn_dtype of the dense path                                       */
    int use_refine;   /* refinedet_vgg: ARM loc heads present (4-tuple output)               */
    int plan_flags;   /* TDRN_PLAN_* bits; 0 = the default plan.  Two handles in one process may differ.   */
    int reserved[4];
} tdrn_net_config;

/* Plan-shaping switches (tdrn_net_config.plan_flags).  Every one of them changes the launch plan only: outputs are
 * bit-identical with the bit set or clear, except TDRN_PLAN_NO_DEFORM_TS in the 16-bit modes (where the rounding of the
 * deformable heads sits differently; both stay inside the 16-bit drift bounds of tests/test_gpu_net.py).
 * The environment variables of the same names (TDRN_FUSE_FIRST=0, TDRN_LATE_SIDE=0|1|2, TDRN_STREAMS=1, TDRN_DEFORM_TS=0, TDRN_CHAIN=1)
 * are diagnostics overrides read when a plan is built; a set variable wins over the flag. */
#define TDRN_PLAN_NO_FUSE_FIRST 1   /* keep the first conv a launch of its own (its output tensor is then materialised) */
#define TDRN_PLAN_NO_LATE_SIDE  2   /* release the side-lane convs on their true inputs instead of behind conv5_3      */
#define TDRN_PLAN_ONE_STREAM    4   /* no side lanes: every launch on the caller's stream                              */
#define TDRN_PLAN_NO_DEFORM_TS  8   /* deformable heads as the fused gather kernel (no transform-then-sample)           */
#define TDRN_PLAN_CHAIN         16  /* the small top-of-pyramid layers as ONE queue-driven launch (measured slower: off by default) */
/* Kernel-choice switches (tests/test_gpu_pin16.py holds the default plan bit-identical to each of them): */
#define TDRN_PLAN_NO_CONV_PP    32  /* conv3x3_pp.hip's layers stay on the loader/consumer kernel conv3x3_patch.hip                */
#define TDRN_PLAN_NO_PP_SK      64  /* conv3x3_pp.hip runs whole items only (no chained split)                                     */
#define TDRN_PLAN_NO_CONV_PATCH 128 /* neither 3x3 direct-conv kernel: every conv on the generic implicit GEMM (conv_igemm.hip)    */
#define TDRN_PLAN_DWPW          512 /* MobileNet trunks: eight conv_dw blocks as ONE launch each (dwpw.hip dwpw_kernel: the depthwise output stays in
                                       LDS; bit-identical; measured SLOWER than the two launches -- the depthwise conv on the vector ALU
                                       beside 128 live accumulators -- hence opt-in)                                                    */
#define TDRN_PLAN_NO_PW1X1      1024 /* the wide 1x1 convs stay on conv_igemm.hip instead of dwpw.hip's persistent GEMM (pw1x1_kernel)     */
#define TDRN_PLAN_NO_DW_SLIDE   2048 /* depthwise 3x3 layers on the one-row strip kernel instead of the sliding-window one (same bits)       */
#define TDRN_PLAN_DW_SLIDE_ALL  4096 /* ... the sliding-window kernel (8-row segments) at every batch, also where it leaves CUs idle       */
#define TDRN_PLAN_NO_CONV_WS    8192 /* the pooled Cin = 64 layer (conv1_2, with the first conv fused) stays on conv3x3_patch.hip instead of the weight-stationary conv3x3_ws.hip (conv2_1, full-resolution output, is on conv3x3_patch.hip either way unless TDRN_CONV_WS=2) */
#define TDRN_PLAN_NO_YGEMM_V2  16384 /* transform-then-sample heads: the transform on the round-3 schedule of ygemm_k256 (two barriers per tile, stores behind the
                                       multiply phase) instead of the round-5 one (deform.hip ygemm_k256_v2_kernel); same bits                    */
#define TDRN_PLAN_NO_HEAD3X3   32768 /* the narrow fp32 3x3 heads (ARM loc) stay on conv_igemm.hip instead of head3x3.hip (different K order: the fp32 sums
                                       differ in their last bits)                                                                        */
#define TDRN_PLAN_TS_ONE_RANGE  65536 /* transform-then-sample heads: the whole batch as ONE range (Y of the whole batch in its buffer) instead of ranges
                                       whose Y fits the memory-side cache (192 MiB); same bits -- for tools and tests that read Y back       */
#define TDRN_PLAN_NO_PATCH_TAIL 131072 /* conv3x3_patch.hip and dwpw.hip pw1x1_kernel run whole items only: no 64- / 128-cout sub-items in an XCD's last, sparsely filled
                                        * round (the tail split of round 6: same MFMA rows, same K order -- bit-identical; for A/B runs and the test) */
#define TDRN_PLAN_FAULT_HANDOFF 256 /* fault injection (tests only): producers of the chained split never raise their flag, so the
                                       consumers' bounded polls run out -> the forward is reported failed, it does not hang          */

typedef struct tdrn_net tdrn_net;

/* Builds the layer plan on the host (no device work). */
TDRN_API int tdrn_net_create(const tdrn_net_config *cfg, tdrn_net **out);
/* Frees the host-side plan.  It does NOT destroy the HIP streams / events the net used: they go back to a per-process,
 * per-device pool and are reused by the next net on that device.  Reason (ROCm 7.2, csrc/dev/graph_destroy_repro.hip): a hipGraph
 * captured AFTER streams / events that took part in an earlier capture had been destroyed crashed inside hipGraphLaunch.
 * The caller must not destroy a net while a forward of it (or a graph captured from one) is still executing. */
TDRN_API void tdrn_net_destroy(tdrn_net *net);

/* state_dict interface: names and shapes are the reference's (entries ending in
 * num_batches_tracked are not parameters and are rejected with TDRN_E_PARAM).
 * tdrn_net_param_count/info enumerate what the plan expects: the reference's names and shapes, in the order the plan
 * consumes them (load by name, as load_state_dict does; the order carries no meaning). */
TDRN_API int tdrn_net_param_count(const tdrn_net *net);
TDRN_API int tdrn_net_param_info(const tdrn_net *net, int index, const char **name,
                                 int64_t shape[4], int *ndim);
/* Copies one fp32 HOST tensor into the net's staging area. */
TDRN_API int tdrn_net_set_param(tdrn_net *net, const char *name, const float *data_host,
                                int64_t numel);

/* Device memory the caller must provide. */
TDRN_API size_t tdrn_net_weight_bytes(const tdrn_net *net);
TDRN_API size_t tdrn_net_workspace_bytes(const tdrn_net *net, int batch);
TDRN_API int tdrn_net_num_priors(const tdrn_net *net);

/* Folds BatchNorm into the preceding conv, repacks OIHW -> [Cout_pad][kh][kw][Cin] in the
 * net's dtype, and uploads the blob (synchronous; one-time).  Every expected parameter must
 * have been set.  Ranks that receive the blob by RCCL broadcast skip this call and use
 * tdrn_net_adopt_weights() instead. */
TDRN_API int tdrn_net_pack_weights(tdrn_net *net, void *weights_dev, size_t weights_bytes,
                                   void *stream);
TDRN_API int tdrn_net_adopt_weights(tdrn_net *net);

typedef struct {
    const float *x;        /* (B,3,S,S) fp32 NCHW, mean-subtracted 0..255 range, device      */
    int batch;
    float *arm_loc;        /* (B,P,4) fp32; DRN/RefineDet: ARM loc; ssd4scale: the loc output */
    float *odm_loc;        /* (B,P,4) fp32; NULL for ssd4scale                               */
    float *conf;           /* (B*P,C) fp32 softmax (test phase) or logits                    */
    float *offsets[4];     /* optional (B,G*18,H,W) fp32 NCHW out (arm_offset_list); or NULL */
    const float *ref_loc[4]; /* ssd4scale deform=1: (B,12,H,W) fp32 NCHW loc maps IN        */
    float *loc_maps[4];    /* ssd4scale ret_loc: (B,12,H,W) fp32 NCHW raw loc maps OUT       */
    /* reserved[0] != NULL, ssd4scale deform=1 only: REUSE the deformable offsets the previous forward of this net computed in
     * this workspace at this batch size instead of recomputing them from ref_loc (the reference's cached offset_list of the frames
     * between two key frames, evaluate_trn.py:459-462: offsets are a function of the key frame's loc maps only).  ref_loc may then
     * be NULL; TDRN_E_STATE when no such forward came before (other workspace, other batch).
     * reserved[1] != NULL, ssd4scale deform=1 only: KEY-FRAME BROADCAST -- (intptr_t) reserved[1] = Bk, the number of key frames:
     * ref_loc (and the optional `offsets` outputs) hold Bk samples and frame b of the batch uses the offsets of key frame b % Bk,
     * i.e. the batch is n = B / Bk frames of each of Bk clips in frame-major order.  One forward then does what the reference's
     * loop does frame by frame with its cached offset_list (evaluate_trn.py:452-462): the frames of an interval depend on the key
     * frame only through those offsets.  TDRN_E_ARG unless 1 <= Bk <= B and B % Bk == 0.
     * reserved[2] != NULL, with ref_loc: a hipEvent_t that the stream producing the ref_loc maps (the static net's forward, running
     * on ANOTHER stream beside this one) records when they are complete.  The forward waits for it right before its first read of
     * ref_loc -- behind its trunk, which does not depend on the maps -- instead of the caller serialising the two forwards.
     * reserved[3] != NULL: a `const tdrn_u8_frames *` -- the batch as UINT8 planes instead of `x` (which may then be NULL): the input
     * of the net is float(planes[b][c][y][x]) - mean[c].  16-bit plans of the VGG trunks read the planes inside the first conv, which
     * is computed by conv1_2's producers (conv3x3_ws.hip); every other plan converts them with one small launch into a workspace
     * tensor first.  Same output bits as the fp32 route. */
    void *reserved[4];
} tdrn_net_io;
typedef struct {
    const uint8_t *planes; /* (B, 3, S, S) uint8, device, in the net's channel order (tdrn_preprocess_u8 with to_rgb as the driver swaps) */
    float mean[3];         /* per plane, i.e. in the SAME channel order (BGR means (104,117,123) become (123,117,104) behind to_rgb)      */
} tdrn_u8_frames;

TDRN_API int tdrn_net_forward(tdrn_net *net, const void *weights_dev, void *workspace,
                              size_t workspace_bytes, const tdrn_net_io *io, void *stream);

/* Device-side failures.  Two launches hand data between workgroups through flags with BOUNDED polls (the chained split of
 * conv3x3_pp.hip; the opt-in chain launch of conv_igemm.hip).  A poll that runs out never hangs the GPU and never passes
 * silently: the kernel stores a code into a host-visible (pinned, device-mapped) status word owned by the net.
 * tdrn_net_check returns TDRN_OK, or TDRN_E_DEVICE when any forward enqueued on this net SINCE THE LAST CHECK has reported such a
 * failure (the word is cleared by the call); it does not synchronise -- call it after the stream (or the hipGraph replay) that
 * ran the forward has been synchronised to learn about THAT forward.  tdrn_net_forward performs the same check on entry, so a
 * failed forward makes the next tdrn_net_forward on the net return TDRN_E_DEVICE instead of launching ("never continue after
 * an error"); the call after that runs again (flags and counters are re-zeroed by every forward).
 * `detail` (or NULL) receives the raw word: bit 0 chained-split poll of conv3x3_pp.hip, bit 1 chain-launch poll. */
TDRN_API int tdrn_net_check(tdrn_net *net, unsigned *detail);

/* Per-kernel accounting of the LAST forward for bench.py's roofline line: algorithmic FLOPs
 * and bytes per kernel family, and (when profiling is enabled) hipEvent-measured time.
 * tdrn_net_profile(net, 1) makes the next forwards record an event pair around every launch and run
 * everything on `stream` (no overlap: the kernel's own duration); tdrn_net_profile(net, 2) records the
 * same pairs on the production schedule (side lanes on: the duration a launch has in the real step).
 * Both add launch gaps: use only in a dedicated profiling pass. */
typedef struct {
    char name[48];
    int launches;
    double flops;        /* algorithmic FLOPs (2*MAC) of all launches                        */
    double bytes;        /* algorithmic HBM bytes (inputs + weights + outputs, once each)   */
    double ms;           /* summed launch durations (profiling on), else 0                   */
} tdrn_kernel_stat;
TDRN_API int tdrn_net_profile(tdrn_net *net, int enable);
TDRN_API int tdrn_net_kernel_stats(tdrn_net *net, tdrn_kernel_stat *out, int max_entries);
/* same accounting per launch of the last profiled forward (name = producing parameter / op kind) */
TDRN_API int tdrn_net_op_stats(tdrn_net *net, tdrn_kernel_stat *out, int max_entries);
/* the same launches as a timeline: start / end of every launch of the last profiled forward in ms after the first launch's
 * start, and the stream lane it ran on (0 = the caller's stream); entry i corresponds to entry i of tdrn_net_op_stats */
TDRN_API int tdrn_net_op_timeline(tdrn_net *net, float *start_ms, float *end_ms, int *lane, int max_entries);

/* Test / debug access to the plan's internal activation tensors (NHWC, net dtype) after a
 * forward on the same workspace: tdrn_net_tensor_info names tensor `index` after the parameter
 * that produced it (e.g. "backbone.3", "L2Norm_4_3", "pool:backbone.3"); tdrn_net_read_tensor
 * converts it to fp32 NCHW (B,C,H,W) into out_dev.  Used by tests/ to compare every stage with
 * the oracle; not part of the hot path.  Tensors that a fused launch never materialises are not written: full-resolution
 * maps whose only reader is a fused max-pool, and -- in the 16-bit plans of the VGG trunks -- the first conv's output, which is
 * computed inside the next conv's loader (environment TDRN_FUSE_FIRST=0 keeps it as its own launch). */
/* The plan's ops (test / debug access, like the tensors): what each launch computes and on which tensors, so that a test can
 * recompute ANY stage from the stage's own materialised input (tests/test_gpu_pin16.py: every conv of the 16-bit plans against an
 * fp64 convolution of its device input with the 16-bit-rounded folded weights).  Tensor fields are indices for
 * tdrn_net_tensor_info / tdrn_net_read_tensor, -1 = none. */
typedef struct {
    int kind;          /* 0 first conv (Cin = 3), 1 dense conv, 2 ConvTranspose2d(2,2), 3 depthwise 3x3, 4 MaxPool2d(2,2), 5 L2Norm,
                          6 1x1 offset conv, 7 deformable heads (loc + conf of one pyramid level), 8 anything else            */
    int in, out, res;  /* input, output (-1: a head that writes arm_loc / odm_loc / conf), residual added before the ReLU       */
    int pool;          /* conv: output of the MaxPool2d(2,2) fused into this launch (then `out` is not materialised), or -1    */
    int off, y;        /* deformable heads: the fp32 offset tensor; the transform GEMM's output Y (16-bit one-group plans) or -1 */
    int k, stride, pad, dil, relu, ceil_mode, splitk, groups;
    int out_kind;      /* 0 tensor, 1 arm_loc, 2 odm_loc, 3 conf                                                               */
    int level;         /* heads / offset convs: pyramid level                                                                  */
    int n_branches, k2, pad2, off_c0[2]; /* deformable heads: 3x3 (+ 5x5) branch, first offset channel of each branch          */
    int y_tap_major;   /* deformable heads with y >= 0: 1 = Y is [tap][B*H*W][80], 0 = [B*H*W][y channels] (column of tap t: (t/3)*256 + (t%3)*80) */
    int fused_first;   /* conv: 1 = this launch also computes the first conv (its `in` is then not materialised)               */
    int fused_dw;      /* depthwise: 1 = this launch also computes the pointwise conv behind it (dwpw.hip: its `out` is not materialised
                          and the next op, that conv, is not a launch of its own); conv: 1 = computed by the depthwise op in front   */
    char w[48], b[48], bn[48], w2[48], b2[48];  /* state_dict prefixes ('' = none): weight / bias owner / BatchNorm; second source (merged 5x5+3x3 heads; offset2;
                          deformable heads: w = 3x3 loc, b = 3x3 conf, w2 = 5x5 loc, b2 = 5x5 conf)                             */
    int y_groups;      /* deformable heads with y >= 0: output-column groups of <= 80 (12 loc + 3 * classes conf columns; 21 classes: 1,
                          31: 2, 81: 4).  Tensor y holds one region per group, y channels / y_groups channels each; group g's region
                          starts g * (y channels / y_groups) * H * W * B elements into the buffer and holds columns [80 g, 80 g + 80)   */
} tdrn_op_info;
TDRN_API int tdrn_net_op_count(const tdrn_net *net);
TDRN_API int tdrn_net_op_info(const tdrn_net *net, int index, tdrn_op_info *out);

TDRN_API int tdrn_net_tensor_count(const tdrn_net *net);
TDRN_API int tdrn_net_tensor_info(const tdrn_net *net, int index, const char **label, int *C, int *H,
                                  int *W);
TDRN_API int tdrn_net_read_tensor(const tdrn_net *net, const void *workspace, int batch, int index,
                                  float *out_dev, void *stream);
/* The inverse, and a forward that starts in the middle of the plan -- analysis only (scripts/attribution.py: which stage's
 * 16-bit rounding moves a detection): tdrn_net_write_tensor stores fp32 NCHW values (B,C,H,W) into tensor `index` of the
 * workspace, rounded once to the net dtype; tdrn_net_forward_from runs the plan's ops [first_op, end) assuming everything the
 * earlier ops produce (workspace tensors, and the head outputs in `io` that earlier ops wrote) is already in place.  Needs a
 * TDRN_PLAN_ONE_STREAM plan (TDRN_E_STATE otherwise: the side lanes' event graph assumes a whole forward). */
TDRN_API int tdrn_net_write_tensor(const tdrn_net *net, void *workspace, int batch, int index,
                                   const float *in_dev, void *stream);
TDRN_API int tdrn_net_forward_from(tdrn_net *net, const void *weights_dev, void *workspace,
                                   size_t workspace_bytes, const tdrn_net_io *io, int first_op, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* TDRN_HIP_H */
