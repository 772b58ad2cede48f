// kernels.h -- launch interfaces of the HIP kernels (internal to libtdrn_hip).
#pragma once
#include <cmath>

#include "common.h"

namespace tdrn {

// A device buffer every padding tap / out-of-range row reads instead of branching (16-B aligned,
// >= 256 zero bytes).  It lives at the start of the caller-provided workspace / weight blob.
constexpr size_t kZeroPageBytes = 256;

// Raises a kernel's dynamic-LDS limit to the full 160 KiB.  The attribute is per DEVICE: the call is repeated
// for every device a kernel is used on (a small per-device, per-kernel memo keeps it off the hot path; it is
// idempotent, so a race between threads only repeats the call).  (layers.hip)
int allow_big_lds(const void *kernel);

// ---------------------------------------------------------------------------------------------
// Dense convolution as implicit GEMM on MFMA (conv_igemm.hip).
//   in  : NHWC [B][H][W][Cin]      (DT; Cin a multiple of the 128-byte K-step)
//   w   : [phases][Npad][kh*kw][Cin] (DT; BatchNorm folded; rows >= Cout are zero)
//   out : element (b,ho,wo,c) at  o_base + b*o_bs + ho*o_rs + wo*o_cs + c   (DT or fp32)
//   res : optional residual with the SAME view as out (DT), added before the ReLU
//   phases = 4 turns the launch into ConvTranspose2d(k=2,s=2): phase z=(i,j) uses weight slab z
//   and adds i*o_pr + j*o_pc to o_base.
// ---------------------------------------------------------------------------------------------
struct ConvArgs {
    const void *in = nullptr, *w = nullptr, *res = nullptr, *zero_page = nullptr;
    const float *bias = nullptr;   // [phases? no: shared][Npad] fp32
    void *out = nullptr;
    int B = 0, H = 0, W = 0, Cin = 0;
    int Ho = 0, Wo = 0, Cout = 0, Npad = 0;
    int kh = 1, kw = 1, stride = 1, pad = 0, dil = 1;
    int relu = 0, out_f32 = 0, phases = 1;
    int splitk = 1;                 // > 1: K is cut into slices over blockIdx.y (small-M layers); needs `partial`
    void *partial = nullptr;        // fp32 scratch of conv_splitk_bytes()
    long long o_bs = 0, o_rs = 0, o_cs = 0, o_base = 0, o_pr = 0, o_pc = 0;
    int dtype = TDRN_BF16;
    // patch kernel, first conv fused in (conv3x3_patch.hip FUSE): frames NCHW fp32, the first conv's folded weights [c][27] and bias
    const float *fuse_x = nullptr, *fuse_w = nullptr, *fuse_b = nullptr;
    int fuse_cout = 0;
    // ... or the frames as uint8 planes (B, 3, S, S) with the per-plane mean still to be subtracted (conv3x3_ws.hip only: tdrn_net_io.reserved[3])
    const unsigned char *fuse_x8 = nullptr;
    float fuse_mean[3] = {0.f, 0.f, 0.f};
    int max_wgs = 0;                // patch kernel: > 0 caps the persistent grid (a multiple of 8), leaving CUs to concurrent lanes
    void *sk_ws = nullptr;          // conv3x3_pp.hip: scratch of conv_pp_sk_bytes() for the chained split (one launch at a time), or null
    bool sk_flags_zero = false;     // the first 1024 bytes of sk_ws are zero on entry (every launch leaves them zero): no memset node
    // kernel-choice switches of this launch (tdrn_net_config.plan_flags TDRN_PLAN_NO_CONV_PP / _NO_PP_SK / _NO_CONV_PATCH):
    // bit 0: not conv3x3_pp.hip, bit 1: no chained split, bit 2: neither 3x3 direct-conv kernel, bit 3: not pw1x1 (dwpw.hip), bit 6: not
    // conv3x3_ws.hip (TDRN_PLAN_NO_CONV_WS).  Same output bits either way.
    int kdisable = 0;
    // host-visible status words (pinned, device-mapped; tdrn_net_check): [0] <- 1 when a chained-split poll runs out,
    // [1] <- 1 when a poll of the chain launch does.  Null: a timed-out poll is not reported (dev harness only).
    unsigned *status = nullptr;
    int fault_handoff = 0;          // fault injection (tests): producers never raise their flag, the polls are short
};
int launch_conv(const ConvArgs &a, hipStream_t s);
// Several small dependent layers in ONE launch (conv_igemm.hip, conv_chain_kernel): layer i may depend on up to two EARLIER
// layers of the list (dep = index or -1); everything else a layer reads must be complete when the launch starts.  Every layer
// keeps its own splitk / partial slab.  `ctr`: conv_chain_ctr_bytes() of zeroed device words.  Output bits equal launch_conv's.
struct ChainLayer { ConvArgs a; int dep[2] = {-1, -1}; };
int conv_chain_supported(const ConvArgs &a);
int conv_chain_max_layers();
size_t conv_chain_ctr_bytes();
int launch_conv_chain(const ChainLayer *layers, int n, unsigned *ctr, hipStream_t s, unsigned *status = nullptr);
int conv_splitk_choice(const ConvArgs &a);          // 1 = no split
size_t conv_splitk_bytes(const ConvArgs &a, int splits);
// narrow 3x3/s1/p1 heads with fp32 output (<= 16 columns: the ARM loc heads), head3x3.hip; kdisable bit 8 (TDRN_PLAN_NO_HEAD3X3) declines
int head3x3_supported(const ConvArgs &a);
int launch_head3x3(const ConvArgs &a, hipStream_t s);
// warp-specialised 3x3/s1/p1 kernel (conv3x3_patch.hip); out_pool = optional fused MaxPool2d(2,2) output
int conv_patch_enabled();                       // TDRN_CONV_PATCH (default 1)
int patch_conv_supported(const ConvArgs &a);   // 0 = no, 32/16 = 2-D tiles, -1 = flat tiles
int launch_conv3x3_patch(const ConvArgs &a, void *out_pool, hipStream_t s);
// all-waves-compute ("ping-pong") 3x3/s1/p1 kernel for the 16-bit Cin >= 128, Cout % 256 == 0 layers (conv3x3_pp.hip);
// launch_conv3x3_patch hands those layers over to it (TDRN_CONV_PP=0 keeps the loader/consumer kernel)
int pp_conv_supported(const ConvArgs &a);      // 0 = no, else the tile mode of patch_conv_supported
int launch_conv3x3_pp(const ConvArgs &a, void *out_pool, hipStream_t s);
void conv_pp_force(int v);                      // dev harness: -1 = environment, 0 / 1 = forced
void conv_pp_sk_force(int v);                   // dev harness: the chained split ("stream-K") of conv3x3_pp.hip on / off
int conv_pp_sk_enabled();                       // TDRN_CONV_PP_SK (default 1)
size_t conv_pp_sk_bytes();
// weight-stationary 3x3/s1/p1 kernel for the 16-bit Cin == 64 layers (conv3x3_ws.hip: the whole weight tile resident in LDS, the
// activations in a ring of image rows, the first conv optionally computed by its producer waves); launch_conv3x3_patch hands those
// layers over to it when their output is POOLED -- by default that is conv1_2 alone; a full-resolution output (conv2_1) is declined
// unless TDRN_CONV_WS=2 (it measured slower there) -- and TDRN_CONV_WS=0 / kdisable bit 6 keep the loader/consumer kernel.  Same output bits.
int ws_conv_supported(const ConvArgs &a);
int launch_conv3x3_ws(const ConvArgs &a, void *out_pool, hipStream_t s);
void conv_ws_force(int v);                      // dev harness: -1 = environment, 0 / 1 = forced
// rows of the packed weight matrix must be padded to a multiple of this
int conv_n_pad(int cout);
// channels of every NHWC activation tensor are padded to a multiple of this
constexpr int kChanPad = 64;

// ---------------------------------------------------------------------------------------------
// HBM-bound layer kernels (layers.hip)
// ---------------------------------------------------------------------------------------------
// first conv: x NCHW fp32 [B][3][S][S] -> NHWC DT [B][Ho][Wo][Cpad], 3x3, pad 1, stride 1|2,
// folded BN + ReLU.  w: fp32 [Cout][27] (k = c*9 + r*3 + q), bias fp32 [Cout].
int launch_first_conv(const float *x, const float *w, const float *bias, void *out, int B, int S,
                      int stride, int Cout, int Cpad, int relu, int dtype, hipStream_t s);
// 2x2 stride-2 max pool, NHWC DT, optional ceil_mode
int launch_maxpool2(const void *in, void *out, int B, int H, int W, int C, int ceil_mode, int dtype,
                    hipStream_t s);
// L2Norm over channels: y = w[c] * (x / (sqrt(sum x^2) + 1e-10)), NHWC DT
int launch_l2norm(const void *in, const float *w, void *out, long long pixels, int C, int dtype,
                  hipStream_t s);
// depthwise 3x3, pad 1, stride 1|2, folded BN + ReLU, NHWC DT.  w: fp32 [9][Cpad], bias [Cpad]
// (kdisable bit 4: the one-row strip kernel instead of the sliding-window one, bit 5: the sliding-window one always -- same bits)
int launch_dwconv3(const void *in, const float *w, const float *bias, void *out, int B, int H, int W,
                   int C, int stride, int relu, int dtype, hipStream_t s, int kdisable = 0);
// conv_dw block fused (dwpw.hip): depthwise 3x3 (stride 1, pad 1, fp32 weights [9][Cin] + bias [Cin], ReLU) -> pointwise 1x1
// (DT weights [Npad][Cin], fp32 bias [Npad], ReLU) in one launch; the depthwise output stays in LDS.  16-bit types only.
struct DwPwArgs {
    const void *in = nullptr, *w = nullptr;
    const float *wdw = nullptr, *bdw = nullptr, *bias = nullptr;   // (bdw must lie behind wdw in the same allocation: the weight blob)
    void *out = nullptr;
    int B = 0, H = 0, W = 0, Cin = 0, Cout = 0, Npad = 0, Cs = 0;  // Cs: channel stride of the output tensor
    int stride = 1, relu_dw = 1, relu = 1, dtype = TDRN_BF16;
};
int dwpw_enabled();                              // TDRN_DWPW (default 1)
int dwpw_supported(const DwPwArgs &a);           // 0 = no, else the tile mode
int launch_dwpw(const DwPwArgs &a, hipStream_t s);
// wide 1x1 convs as a persistent 256 x 256-item GEMM (dwpw.hip pw1x1_kernel); launch_conv hands them over (TDRN_PW1X1=0 / kdisable bit 3: off)
int pw1x1_supported(const ConvArgs &a);
int launch_pw1x1(const ConvArgs &a, hipStream_t s);
// softmax over rows of (R, C) fp32, in place allowed
int launch_softmax_rows(const float *in, float *out, long long R, int C, hipStream_t s);
// 1x1 "offset" convs on the 12-channel ARM loc map: loc fp32 (pixel stride loc_ps, batch stride
// loc_bs, 12 ch) -> off NHWC fp32 [B*H*W][n_out], w fp32 [n_out][12], bias [n_out] (or null)
int launch_offset_conv(const float *loc, long long loc_bs, long long loc_ps, const float *w,
                       const float *bias, float *off, int B, int HW, int n_in, int n_out,
                       hipStream_t s);
// ... of up to four pyramid levels in one launch (n_in = 12); same arithmetic per output
struct OffsetProblem { const float *loc; long long loc_bs, loc_ps; const float *w, *bias; float *off; int B, HW, n_out, blk0; };
struct OffsetMulti { OffsetProblem pr[4]; int n; };
int launch_offset_conv_multi(const OffsetProblem *pr, int n, hipStream_t s);
// layout conversions for the API surfaces that are NCHW fp32
int launch_nchw_to_nhwc(const float *in, void *out, int B, int C, int HW, int Cpad, int dtype,
                        hipStream_t s);   // fp32 NCHW -> DT NHWC (channel-padded with zeros)
int launch_nchw_to_nhwc_grouped(const float *in, void *out, int B, int C, int HW, int G, int cpg_pad, int dtype,
                                hipStream_t s);
int launch_repack_oihw(const float *w, void *out, int Cout, int Npad, int Cin, int taps, int G, int cpg_pad, int dtype,
                       hipStream_t s);
int launch_nhwc_to_nchw_f32(const float *in, long long in_bs, long long in_ps, float *out, int B,
                            int C, int HW, hipStream_t s);   // fp32 "NHWC view" -> fp32 NCHW
// NHWC (channel stride Cpad, dtype DT or fp32) -> fp32 NCHW
int launch_nhwc_any_to_nchw_f32(const void *in, int dtype, int Cpad, float *out, int B, int C, int HW, hipStream_t s);
int launch_preprocess(const unsigned char *in, int B, int H0, int W0, int S, const float *mean_bgr, int to_rgb, float *out,
                      hipStream_t s);
int launch_preprocess_u8(const unsigned char *in, int B, int H0, int W0, int S, int to_rgb, unsigned char *out, hipStream_t s);   // resized uint8 planes
int launch_u8_planes_to_f32(const unsigned char *in, int B, int S, const float *mean, float *out, hipStream_t s);                // (B,3,S,S) u8 -> fp32 - mean[c]
int launch_fill_zero(void *p, size_t bytes, hipStream_t s);

// ---------------------------------------------------------------------------------------------
// Deformable-conv GEMM (deform.hip): up to two branches (3x3 + 5x5) sharing the input and summed.
//   in  : NHWC DT [B][H][W][Cin]
//   off : NHWC fp32 [B][Ho][Wo][off_stride], branch offsets at off + off_ch0 (G*2*k*k channels,
//         channel order of the reference: g, then 2*(i*kw+j)+{0:dh,1:dw})
//   w   : DT [Npad][taps][Cin]
//   out : fp32, element (pixel m, channel c) at out + m*o_ps + c (c < Cout)
// ---------------------------------------------------------------------------------------------
struct DeformBranch {
    const float *off = nullptr;
    int off_stride = 0;
    const void *w = nullptr;
    int kh = 3, kw = 3, pad = 1, stride = 1, dil = 1, G = 1;      // pad / stride / dil: the H axis
    int pad_w = -1, stride_w = -1, dil_w = -1;                     // W axis; < 0 (< 1): same as the H axis
    // key-frame broadcast (TRN clips): `off` holds off_rows pixel rows (the key frames' maps) and output pixel m of the launch reads row
    // (off_row0 + m) % off_rows; 0 = one row per output pixel
    int off_rows = 0, off_row0 = 0;
};
struct DeformArgs {
    const void *in = nullptr, *zero_page = nullptr;
    DeformBranch br[2];
    int n_branches = 1;
    int B = 0, H = 0, W = 0, Cin = 0, Ho = 0, Wo = 0, Cout = 0, Npad = 0;
    // two output segments so that loc and conf heads fused into one GEMM land in their own buffers:
    // channels [0,split) -> out0 (+ b*o0_bs + pix*o0_ps), [split,Cout) -> out1
    float *out0 = nullptr, *out1 = nullptr;
    int split = 0;
    long long o0_bs = 0, o0_ps = 0, o1_bs = 0, o1_ps = 0;
    int dtype = TDRN_BF16;
};
int launch_deform(const DeformArgs &a, hipStream_t s);
// up to 4 independent problems (same dtype / Npad) in one launch
// split_branches = 1: two-branch problems run as two work items that atomicAdd into PRE-ZEROED outputs
int launch_deform_multi(const DeformArgs *args, int n, hipStream_t s, int split_branches);
int deform_n_pad(int cout);
// transform-then-sample path of the 16-bit one-group heads (deform.hip): the caller computes Y = 1x1 GEMM of the input with the
// per-tap weight slabs ([taps][80 columns] per pixel, deform_sample_cols(taps) channels), this launch blends the corners
int deform_sample_supported(const DeformArgs &a);      // 0 = no, else the number of taps of all branches
int deform_sample_cols(int taps);
// column of tap t's 80 outputs in the transform GEMM's output: three taps per 256-column slice (a slice of ygemm_k256 then holds
// whole taps only; columns 240..255 of every slice are zero padding)
constexpr int deform_y_col(int tap) { return (tap / 3) * 256 + (tap % 3) * 80; }
// the largest batch whose Y ([taps][B*H*W][80] tap-major or [B*H*W][ycs]) stays below the kernels' 32-bit byte offsets
int deform_ts_max_batch(int H, int W, int ycs, int taps);
int launch_deform_sample_multi(const DeformArgs *args, const void *const *y, const int *ycs, int n, hipStream_t s, int tap_major = 0);
// Y = X[M][256] * Wt[N][256]^T in the net dtype, weights held in registers (deform.hip); N % 256 == 0
int ygemm_supported(int Cin, int ycols, int dtype);
int launch_ygemm(const void *x, const void *w, void *y, long long M, int N, int ycs, int dtype, hipStream_t s, int taps = 0);   // taps > 0: Y tap-major [taps][M][80]
struct YGemmProblem { const void *x, *w; void *y; long long M; int N, ycs, taps; };
int launch_ygemm_multi(const YGemmProblem *pr, int n, int dtype, hipStream_t s, int kdisable = 0);      // up to 4 problems (pyramid levels) in one launch

// ---------------------------------------------------------------------------------------------
// Detect (detect.hip)
// ---------------------------------------------------------------------------------------------
int launch_decode(const float *loc, const float *priors, int P, float v0, float v1, float *out,
                  hipStream_t s);
int launch_center_size(const float *boxes, int P, float *out, hipStream_t s);
size_t detect_workspace_bytes(int B, int P, int C, int top_k);
int launch_detect(const float *loc, const float *conf, const float *priors, const float *arm_loc,
                  const float *scale4, int scale_on_device, int B, int P, int C, int top_k, float conf_thresh,
                  double nms_thresh, float *out, int32_t *counts, void *ws, size_t ws_bytes,
                  hipStream_t s);
size_t nms_workspace_bytes(int n);
size_t nms_classes_workspace_bytes(int n, int C);
// DetectOTA's association arithmetic (detect.hip): ROI features of kept boxes, similarity of detections to tubelets
int launch_roi_resample(const float *feat, int C, int H, int W, const int32_t *cells, int n, int S, float *out, hipStream_t s);
int launch_ota_similarity(const float *boxes, const float *roi, int n, int F, const float *rows, const int32_t *row_off, int m, float *best,
                          int32_t *arg, hipStream_t s);
int launch_nms_classes(const float *boxes, const float *scores, int n, int C, int first_class, float overlap, float min_score, int top_k,
                       int32_t *keep_out, int32_t *num_out, void *ws, size_t ws_bytes, hipStream_t s);
int launch_nms(const float *dets, int n, double thresh, int strict_gt, int presorted,
               int32_t *keep_out, int32_t *num_out, void *ws, size_t ws_bytes, hipStream_t s,
               int plain_rule = 0, float min_score = -INFINITY, int pre_top_k = 0);

}  // namespace tdrn
