// api.hip -- C ABI entry points of tdrn_hip.h sections (i) and (ii) (section (iii) is in net.hip).
#include <cmath>
#include <cstring>
#include <vector>

#include "kernels.h"

using namespace tdrn;

namespace {

struct DeformPlan {
    int Ho, Wo, taps, ck, cpg_pad, Cin_pad, Npad_total, chunks;
    size_t o_zero, o_in, o_w, o_off, o_out, total;
};

int deform_plan(int N, int Cin, int H, int W, int Cout, int kH, int kW, int dH, int dW, int padH, int padW, int dilH,
                int dilW, int G, int dtype, DeformPlan &p)
{
    // shape_check, utils/deformconv/deform_conv_cuda.c:7-96
    if (kW <= 0 || kH <= 0) return TDRN_E_SHAPE;
    if (dW <= 0 || dH <= 0) return TDRN_E_SHAPE;
    if (dilW <= 0 || dilH <= 0) return TDRN_E_SHAPE;
    if (N <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return TDRN_E_SHAPE;
    if (G <= 0 || Cin % G != 0) return TDRN_E_SHAPE;
    if (dtype < 0 || dtype > 2) return TDRN_E_ARG;
    p.Ho = (H + 2 * padH - (dilH * (kH - 1) + 1)) / dH + 1;
    p.Wo = (W + 2 * padW - (dilW * (kW - 1) + 1)) / dW + 1;
    if (p.Ho < 1 || p.Wo < 1) return TDRN_E_SHAPE;
    if (H < kH || W < kW) return TDRN_E_SHAPE;
    if (padH < 0 || padW < 0) return TDRN_E_SHAPE;
    const int es = dtype_bytes(dtype);
    p.taps = kH * kW;
    p.ck = 128 / es;
    p.cpg_pad = (int)align_up((size_t)(Cin / G), p.ck);
    p.Cin_pad = p.cpg_pad * G;
    p.chunks = cdiv(Cout, 128);
    p.Npad_total = (p.chunks - 1) * 128 + deform_n_pad(Cout - (p.chunks - 1) * 128);
    size_t o = 0;
    p.o_zero = o; o += kZeroPageBytes;
    p.o_in = o;   o += align_up((size_t)N * H * W * p.Cin_pad * es, 256);
    p.o_w = o;    o += align_up((size_t)p.Npad_total * p.taps * p.Cin_pad * es, 256);
    p.o_off = o;  o += align_up((size_t)N * p.Ho * p.Wo * G * 2 * p.taps * 4, 256);
    p.o_out = o;  o += align_up((size_t)N * p.Ho * p.Wo * Cout * 4, 256);
    p.total = o;
    return TDRN_OK;
}

}  // namespace

extern "C" {

const char *tdrn_version(void) { return "tdrn_hip 0.6 (gfx950)"; }

const char *tdrn_error_string(int code)
{
    switch (code) {
        case TDRN_OK: return "ok";
        case TDRN_E_ARG: return "invalid argument";
        case TDRN_E_SHAPE: return "shape check failed";
        case TDRN_E_WORKSPACE: return "workspace too small";
        case TDRN_E_UNSUPPORTED: return "unsupported configuration";
        case TDRN_E_PARAM: return "unknown, missing or mis-shaped parameter";
        case TDRN_E_STATE: return "invalid call order";
        case TDRN_E_VALUE: return "nms_threshold must be non negative.";
        case TDRN_E_DEVICE: return "a device-side hand-off of an earlier forward timed out (its outputs are invalid)";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
    }
}

size_t tdrn_deform_conv_workspace_bytes(int N, int Cin, int H, int W, int Cout, int kH, int kW, int dH, int dW, int padH,
                                        int padW, int dilationH, int dilationW, int deformable_group, tdrn_dtype compute)
{
    DeformPlan p;
    if (deform_plan(N, Cin, H, W, Cout, kH, kW, dH, dW, padH, padW, dilationH, dilationW, deformable_group, compute, p) != TDRN_OK)
        return 0;
    return p.total;
}

int tdrn_deform_conv_forward(const float *input, const float *weight, const float *offset, float *output, int N, int Cin,
                             int H, int W, int Cout, int kW, int kH, int dW, int dH, int padW, int padH, int dilationH,
                             int dilationW, int deformable_group, tdrn_dtype compute, void *workspace,
                             size_t workspace_bytes, void *stream)
{
    if (!input || !weight || !offset || !output) return TDRN_E_ARG;
    DeformPlan p;
    TDRN_TRY(deform_plan(N, Cin, H, W, Cout, kH, kW, dH, dW, padH, padW, dilationH, dilationW, deformable_group, compute, p));
    if (!workspace || workspace_bytes < p.total) return TDRN_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    char *ws = (char *)workspace;
    const int G = deformable_group, es = dtype_bytes(compute);
    TDRN_HIP_TRY(hipMemsetAsync(ws + p.o_zero, 0, kZeroPageBytes, s));
    TDRN_TRY(launch_nchw_to_nhwc_grouped(input, ws + p.o_in, N, Cin, H * W, G, p.cpg_pad, compute, s));
    TDRN_TRY(launch_repack_oihw(weight, ws + p.o_w, Cout, p.Npad_total, Cin, p.taps, G, p.cpg_pad, compute, s));
    const int offC = G * 2 * p.taps;
    TDRN_TRY(launch_nchw_to_nhwc(offset, ws + p.o_off, N, offC, p.Ho * p.Wo, offC, TDRN_F32, s));
    float *out_nhwc = (float *)(ws + p.o_out);
    for (int c = 0; c < p.chunks; ++c) {
        const int c0 = c * 128, cn = (Cout - c0) < 128 ? (Cout - c0) : 128;
        DeformArgs a;
        a.in = ws + p.o_in; a.zero_page = ws + p.o_zero; a.n_branches = 1;
        a.br[0] = DeformBranch{(const float *)(ws + p.o_off), offC, ws + p.o_w + (size_t)c0 * p.taps * p.Cin_pad * es,
                               kH, kW, padH, dH, dilationH, G, padW, dW, dilationW};
        a.B = N; a.H = H; a.W = W; a.Cin = p.Cin_pad; a.Ho = p.Ho; a.Wo = p.Wo; a.Cout = cn; a.Npad = deform_n_pad(cn);
        a.out0 = out_nhwc + c0; a.o0_bs = (long long)p.Ho * p.Wo * Cout; a.o0_ps = Cout; a.split = cn;
        a.dtype = compute;
        TDRN_TRY(launch_deform(a, s));
    }
    return launch_nhwc_to_nchw_f32(out_nhwc, (long long)p.Ho * p.Wo * Cout, Cout, output, N, Cout, p.Ho * p.Wo, s);
}

size_t tdrn_nms_workspace_bytes(int n) { return nms_workspace_bytes(n); }

int tdrn_nms(const float *dets, int n, double thresh, int strict_gt, int32_t *keep_out, int32_t *num_out, void *workspace,
             size_t workspace_bytes, void *stream)
{
    return launch_nms(dets, n, thresh, strict_gt, 0, keep_out, num_out, workspace, workspace_bytes, (hipStream_t)stream);
}

int tdrn_nms_topk(const float *dets, int n, float overlap, float min_score, int top_k, int32_t *keep_out, int32_t *num_out,
                  void *workspace, size_t workspace_bytes, void *stream)
{
    if (top_k < 0) return TDRN_E_ARG;
    return launch_nms(dets, n, (double)overlap, 0, 0, keep_out, num_out, workspace, workspace_bytes, (hipStream_t)stream, 1, min_score,
                      top_k);
}

size_t tdrn_nms_topk_classes_workspace_bytes(int n, int num_classes) { return nms_classes_workspace_bytes(n, num_classes); }

int tdrn_nms_topk_classes(const float *boxes, const float *scores, int n, int num_classes, int first_class, float overlap, float min_score,
                          int top_k, int32_t *keep_out, int32_t *num_out, void *workspace, size_t workspace_bytes, void *stream)
{
    if (top_k < 0) return TDRN_E_ARG;
    return launch_nms_classes(boxes, scores, n, num_classes, first_class, overlap, min_score, top_k, keep_out, num_out, workspace,
                              workspace_bytes, (hipStream_t)stream);
}

int tdrn_roi_resample(const float *feature, int C, int H, int W, const int32_t *cells, int n, int S, float *out, void *stream)
{
    return launch_roi_resample(feature, C, H, W, cells, n, S, out, (hipStream_t)stream);
}

int tdrn_ota_similarity(const float *boxes, const float *roi, int n, int F, const float *rows, const int32_t *row_off, int m, float *best,
                        int32_t *arg, void *stream)
{
    return launch_ota_similarity(boxes, roi, n, F, rows, row_off, m, best, arg, (hipStream_t)stream);
}

int tdrn_gpu_nms_host(int *keep_out, int *num_out, const float *boxes_host, int boxes_num, int boxes_dim,
                      float nms_overlap_thresh, int device_id)
{
    if (!keep_out || !num_out || (!boxes_host && boxes_num > 0) || boxes_num < 0) return TDRN_E_ARG;
    if (boxes_dim != 5) return TDRN_E_SHAPE;
    *num_out = 0;
    if (boxes_num == 0) return TDRN_OK;
    // the reference's _nms switches the device and never switches back (nms_kernel.cu:21-32); a frame-sharded rank
    // must keep ITS device current, so the previous one is restored on every exit
    int prev_dev = -1;
    TDRN_HIP_TRY(hipGetDevice(&prev_dev));
    struct Restore { int d; ~Restore() { if (d >= 0) (void)hipSetDevice(d); } } restore{device_id >= 0 && device_id != prev_dev ? prev_dev : -1};
    if (device_id >= 0 && device_id != prev_dev) TDRN_HIP_TRY(hipSetDevice(device_id));
    const size_t bytes = (size_t)boxes_num * 5 * sizeof(float), wsb = nms_workspace_bytes(boxes_num);
    char *dev = nullptr;
    TDRN_HIP_TRY(hipMalloc((void **)&dev, align_up(bytes, 256) + wsb + align_up((size_t)boxes_num * 4, 256) + 256));
    float *d_boxes = (float *)dev;
    char *d_ws = dev + align_up(bytes, 256);
    int *d_keep = (int *)(d_ws + wsb);
    int *d_num = (int *)((char *)d_keep + align_up((size_t)boxes_num * 4, 256));
    int rc = hip_status(hipMemcpy(d_boxes, boxes_host, bytes, hipMemcpyHostToDevice));
    // the host twin takes a float threshold (gpu_nms.hpp:1-2); nms_kernel.cu:71 compares fp32 > fp32
    if (rc == TDRN_OK) rc = launch_nms(d_boxes, boxes_num, (double)nms_overlap_thresh, 1, 1, d_keep, d_num, d_ws, wsb, 0);
    if (rc == TDRN_OK) rc = hip_status(hipMemcpy(num_out, d_num, sizeof(int), hipMemcpyDeviceToHost));
    if (rc == TDRN_OK && *num_out > 0) rc = hip_status(hipMemcpy(keep_out, d_keep, (size_t)*num_out * sizeof(int), hipMemcpyDeviceToHost));
    (void)hipFree(dev);
    return rc;
}

int tdrn_decode(const float *loc, const float *priors, int P, float var0, float var1, float *boxes_out, void *stream)
{
    if (!loc || !priors || !boxes_out || P < 0) return TDRN_E_ARG;
    return launch_decode(loc, priors, P, var0, var1, boxes_out, (hipStream_t)stream);
}

int tdrn_center_size(const float *boxes, int P, float *out, void *stream)
{
    if (!boxes || !out || P < 0) return TDRN_E_ARG;
    return launch_center_size(boxes, P, out, (hipStream_t)stream);
}

int tdrn_prior_box(int n_maps, const int *feature_maps, double image_size, const double *steps, const double *min_sizes,
                   const double *max_sizes, int n_max_sizes, const int *ar_count, const double *ars, int clip, int flip,
                   float *out)
{
    // layers/functions/prior_box.py:33-64 -- python floats are C doubles; torch.Tensor(list) rounds to fp32
    if (n_maps <= 0 || !feature_maps || !steps || !min_sizes || !ar_count || image_size <= 0) return TDRN_E_ARG;
    if (n_max_sizes > 0 && (!max_sizes || n_max_sizes < n_maps)) return TDRN_E_ARG;
    long long n = 0;
    int ar_base = 0;
    for (int k = 0; k < n_maps; ++k) {
        const int f = feature_maps[k];
        const double f_k = image_size / steps[k];
        const double s_k = min_sizes[k] / image_size;
        for (int i = 0; i < f; ++i)
            for (int j = 0; j < f; ++j) {
                const double cx = (j + 0.5) / f_k, cy = (i + 0.5) / f_k;
                auto emit = [&](double a, double b, double c, double d) {
                    if (out) {
                        float *o = out + n * 4;
                        o[0] = (float)a; o[1] = (float)b; o[2] = (float)c; o[3] = (float)d;
                    }
                    ++n;
                };
                emit(cx, cy, s_k, s_k);
                if (n_max_sizes > 0) {
                    const double sp = std::sqrt(s_k * (max_sizes[k] / image_size));
                    emit(cx, cy, sp, sp);
                }
                for (int a = 0; a < ar_count[k]; ++a) {
                    const double ar = ars[ar_base + a];
                    emit(cx, cy, s_k * std::sqrt(ar), s_k / std::sqrt(ar));
                    if (flip) emit(cx, cy, s_k / std::sqrt(ar), s_k * std::sqrt(ar));
                }
            }
        ar_base += ar_count[k];
    }
    if (out && clip)
        for (long long i = 0; i < n * 4; ++i) out[i] = out[i] > 1.f ? 1.f : (out[i] < 0.f ? 0.f : out[i]);
    return (int)n;
}

int tdrn_preprocess(const uint8_t *frames, int B, int H0, int W0, int S, const float mean_bgr[3], int to_rgb, float *out,
                    void *stream)
{
    if (!frames || !out || !mean_bgr || B <= 0 || H0 <= 0 || W0 <= 0 || S <= 0) return TDRN_E_ARG;
    return launch_preprocess(frames, B, H0, W0, S, mean_bgr, to_rgb, out, (hipStream_t)stream);
}

int tdrn_preprocess_u8(const uint8_t *frames, int B, int H0, int W0, int S, int to_rgb, uint8_t *out, void *stream)
{
    if (!frames || !out || B <= 0 || H0 <= 0 || W0 <= 0 || S <= 0) return TDRN_E_ARG;
    return launch_preprocess_u8(frames, B, H0, W0, S, to_rgb, out, (hipStream_t)stream);
}

size_t tdrn_detect_workspace_bytes(int B, int P, int C, int top_k)
{
    if (B <= 0 || P <= 0 || C <= 0 || top_k <= 0) return 0;
    return detect_workspace_bytes(B, P, C, top_k);
}

int tdrn_detect(const float *loc, const float *conf, const float *priors, const float *arm_loc, const float *scale_host,
                int B, int P, int C, int top_k, float conf_thresh, double nms_thresh, float *out, int32_t *counts_out,
                void *workspace, size_t workspace_bytes, void *stream)
{
    return launch_detect(loc, conf, priors, arm_loc, scale_host, 0, B, P, C, top_k, conf_thresh, nms_thresh, out, counts_out,
                         workspace, workspace_bytes, (hipStream_t)stream);
}

int tdrn_detect_dev_scale(const float *loc, const float *conf, const float *priors, const float *arm_loc, const float *scale_dev,
                          int B, int P, int C, int top_k, float conf_thresh, double nms_thresh, float *out, int32_t *counts_out,
                          void *workspace, size_t workspace_bytes, void *stream)
{
    return launch_detect(loc, conf, priors, arm_loc, scale_dev, 1, B, P, C, top_k, conf_thresh, nms_thresh, out, counts_out,
                         workspace, workspace_bytes, (hipStream_t)stream);
}

}  // extern "C"
