/*
 * tdrn_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the arithmetic on TDRN's dual-refinement inference path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product path (tdrn_amd/, libtdrn_hip.so) never links or calls it.
 *
 * Every function cites the reference file:line (paths relative to the upstream
 * SeanChenxy/TDRN tree) whose algorithm it follows.  Parity pinning: the reference has
 * no golden vectors; this oracle is pinned (tests/test_oracle_pin.py,
 * tests/golden/) against outputs of the reference's own Python run on CPU in the build
 * container (PriorBox, decode, center_size, L2Norm, Detect, full non-deformable and
 * deformable nets) and, for the deformable op whose reference implementation is CUDA-only
 * and un-buildable here (nvcc + THC), against analytic known-answer tests.
 *
 * All arithmetic is IEEE fp32 unless a comment says otherwise; build with
 * -ffp-contract=off so that no multiply-add is fused (the reference's numpy/torch/CUDA
 * fp32 code rounds after every operation).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* ---------------------------------------------------------------------------------------
 * Deformable convolution v1 forward.
 * utils/deformconv/deform_conv_cuda_kernel.cu:15-51  (deformable_im2col_bilinear)
 * utils/deformconv/deform_conv_cuda_kernel.cu:156-208 (deformable_im2col_gpu_kernel)
 * utils/deformconv/deform_conv_cuda.c:98-213          (deform_conv_forward_cuda: per image
 *   zero the output, im2col, then out(Cout x HW) += W(Cout x K) * col(K x HW); no bias)
 * ------------------------------------------------------------------------------------- */

/* .cu:15-51.  `data` points at (c_im, h_in, w_in); (h, w) are coordinates RELATIVE to it;
 * height/width are the remaining extent (H - h_in, W - w_in). */
static float orc_bilinear(const float *data, int data_width, int height, int width, float h,
                          float w)
{
    int h_low = (int)floorf(h);
    int w_low = (int)floorf(w);
    int h_high, w_high;
    if (h_low >= height - 1) {
        h_high = h_low = height - 1;
        h = (float)h_low;
    } else {
        h_high = h_low + 1;
    }
    if (w_low >= width - 1) {
        w_high = w_low = width - 1;
        w = (float)w_low;
    } else {
        w_high = w_low + 1;
    }
    float lh = h - h_low;
    float lw = w - w_low;
    float hh = 1 - lh, hw = 1 - lw;
    float v1 = data[h_low * data_width + w_low];
    float v2 = data[h_low * data_width + w_high];
    float v3 = data[h_high * data_width + w_low];
    float v4 = data[h_high * data_width + w_high];
    float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
    return (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);
}

/* .cu:156-208 for ONE image.  col layout: [(c*kh*kw + i*kw + j)][h_col*W_col + w_col]. */
ORC_API void orc_deform_im2col(const float *im, const float *offset, int C, int H, int W, int kh,
                               int kw, int pad_h, int pad_w, int stride_h, int stride_w,
                               int dil_h, int dil_w, int G, float *col)
{
    const int Hc = (H + 2 * pad_h - (dil_h * (kh - 1) + 1)) / stride_h + 1;
    const int Wc = (W + 2 * pad_w - (dil_w * (kw - 1) + 1)) / stride_w + 1;
    const int cpg = C / G; /* channel_per_deformable_group, .cu:225 */
#pragma omp parallel for schedule(static)
    for (int c_im = 0; c_im < C; ++c_im) {
        const int g = c_im / cpg;
        const float *off_g = offset + (size_t)g * 2 * kh * kw * Hc * Wc;
        for (int h_col = 0; h_col < Hc; ++h_col) {
            for (int w_col = 0; w_col < Wc; ++w_col) {
                const int h_in = h_col * stride_h - pad_h;
                const int w_in = w_col * stride_w - pad_w;
                const float *im_ptr = im + ((ptrdiff_t)c_im * H + h_in) * W + w_in;
                for (int i = 0; i < kh; ++i) {
                    for (int j = 0; j < kw; ++j) {
                        const float offset_h =
                            off_g[((size_t)(2 * (i * kw + j)) * Hc + h_col) * Wc + w_col];
                        const float offset_w =
                            off_g[((size_t)(2 * (i * kw + j) + 1) * Hc + h_col) * Wc + w_col];
                        float val = 0.f;
                        const float h_im = h_in + i * dil_h + offset_h;
                        const float w_im = w_in + j * dil_w + offset_w;
                        if (h_im >= 0 && w_im >= 0 && h_im < H && w_im < W) {
                            const float map_h = i * dil_h + offset_h;
                            const float map_w = j * dil_w + offset_w;
                            val = orc_bilinear(im_ptr, W, H - h_in, W - w_in, map_h, map_w);
                        }
                        col[((size_t)(c_im * kh * kw + i * kw + j) * Hc + h_col) * Wc + w_col] =
                            val;
                    }
                }
            }
        }
    }
}

/* deform_conv_cuda.c:98-213.  input NCHW, offset (N, G*2*kh*kw, Hc, Wc), weight OIHW,
 * output (N, Cout, Hc, Wc).  Returns 0, or a negative code for the shape_check failures
 * (.c:7-96).  `col` scratch is malloc'ed here (the reference resizes a caller tensor). */
ORC_API int orc_deform_conv_forward(const float *input, const float *offset, const float *weight,
                                    float *output, int N, int Cin, int H, int W, int Cout, int kh,
                                    int kw, int stride_h, int stride_w, int pad_h, int pad_w,
                                    int dil_h, int dil_w, int G)
{
    if (kh <= 0 || kw <= 0) return -9;
    if (stride_h <= 0 || stride_w <= 0) return -11;
    if (dil_h <= 0 || dil_w <= 0) return -14;
    if (G <= 0 || Cin % G != 0) return -2;
    const int Hc = (H + 2 * pad_h - (dil_h * (kh - 1) + 1)) / stride_h + 1;
    const int Wc = (W + 2 * pad_w - (dil_w * (kw - 1) + 1)) / stride_w + 1;
    if (Hc < 1 || Wc < 1) return -3;
    if (H < kh || W < kw) return -2;
    const size_t K = (size_t)Cin * kh * kw, HW = (size_t)Hc * Wc;
    float *col = (float *)malloc(K * HW * sizeof(float));
    if (!col) return -100;
    for (int n = 0; n < N; ++n) {
        orc_deform_im2col(input + (size_t)n * Cin * H * W, offset + (size_t)n * G * 2 * kh * kw * HW,
                          Cin, H, W, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w, G,
                          col);
        float *out_n = output + (size_t)n * Cout * HW;
        memset(out_n, 0, (size_t)Cout * HW * sizeof(float)); /* THCudaTensor_zero, .c:175 */
        /* SGEMM 'n','n': out(Cout x HW) += W(Cout x K) * col(K x HW), fp32 (.c:185-192) */
#pragma omp parallel for schedule(static)
        for (int m = 0; m < Cout; ++m) {
            float *o = out_n + (size_t)m * HW;
            const float *wrow = weight + (size_t)m * K;
            for (size_t k = 0; k < K; ++k) {
                const float wv = wrow[k];
                const float *c = col + k * HW;
                for (size_t p = 0; p < HW; ++p) o[p] += wv * c[p];
            }
        }
    }
    free(col);
    return 0;
}

/* ---------------------------------------------------------------------------------------
 * Greedy NMS.  utils/nms/cpu_nms.pyx:17-68 (cpu_nms).  dets (n,5) = x1,y1,x2,y2,score in
 * PIXEL coordinates, "+1" width/height convention, suppression on ovr >= thresh where ovr
 * is fp32 and thresh a C double.  order = scores.argsort()[::-1] (:25) is unspecified among
 * equal scores; this restatement uses a stable descending order (lower index first) and
 * parity is defined on tie-free score vectors (SURVEY.md 8d "NMS tie caveat").
 * `gt_strict` != 0 gives the GPU twin's rule (utils/nms/nms_kernel.cu:71, ovr > thresh).
 * Returns kept indices (into dets) in descending score; *num_out = count.
 * ------------------------------------------------------------------------------------- */
typedef struct {
    float s;
    int32_t i;
} orc_si;

static int orc_cmp_desc(const void *a, const void *b)
{
    const orc_si *x = (const orc_si *)a, *y = (const orc_si *)b;
    if (x->s > y->s) return -1;
    if (x->s < y->s) return 1;
    return (x->i > y->i) - (x->i < y->i);
}

ORC_API void orc_cpu_nms(const float *dets, int n, double thresh, int gt_strict, int32_t *keep_out,
                         int32_t *num_out)
{
    *num_out = 0;
    if (n <= 0) return;
    float *areas = (float *)malloc((size_t)n * sizeof(float));
    orc_si *order = (orc_si *)malloc((size_t)n * sizeof(orc_si));
    uint8_t *suppressed = (uint8_t *)calloc((size_t)n, 1);
    for (int i = 0; i < n; ++i) {
        const float *d = dets + (size_t)i * 5;
        const float w = d[2] - d[0] + 1; /* :24, each op rounded to fp32 (numpy float32) */
        const float h = d[3] - d[1] + 1;
        areas[i] = w * h;
        order[i].s = d[4];
        order[i].i = i;
    }
    qsort(order, (size_t)n, sizeof(orc_si), orc_cmp_desc);
    int nk = 0;
    for (int _i = 0; _i < n; ++_i) {
        const int i = order[_i].i;
        if (suppressed[i]) continue;
        keep_out[nk++] = i;
        const float ix1 = dets[(size_t)i * 5 + 0], iy1 = dets[(size_t)i * 5 + 1];
        const float ix2 = dets[(size_t)i * 5 + 2], iy2 = dets[(size_t)i * 5 + 3];
        const float iarea = areas[i];
        for (int _j = _i + 1; _j < n; ++_j) {
            const int j = order[_j].i;
            if (suppressed[j]) continue;
            const float *d = dets + (size_t)j * 5;
            const float xx1 = ix1 >= d[0] ? ix1 : d[0];
            const float yy1 = iy1 >= d[1] ? iy1 : d[1];
            const float xx2 = ix2 <= d[2] ? ix2 : d[2];
            const float yy2 = iy2 <= d[3] ? iy2 : d[3];
            float w = xx2 - xx1 + 1;
            float h = yy2 - yy1 + 1;
            w = 0.0f >= w ? 0.0f : w; /* max(0.0, w) with the pyx's own max(a,b)= a if a>=b */
            h = 0.0f >= h ? 0.0f : h;
            const float inter = w * h;
            const float ovr = inter / (iarea + areas[j] - inter);
            if (gt_strict ? ((double)ovr > thresh) : ((double)ovr >= thresh)) suppressed[j] = 1;
        }
    }
    *num_out = nk;
    free(areas);
    free(order);
    free(suppressed);
}

/* ---------------------------------------------------------------------------------------
 * Box arithmetic.  layers/box_utils.py:176-195 (decode), :16-25 (center_size).
 * torch fp32 semantics: one rounding per elementwise op, python-float scalars become fp32.
 * ------------------------------------------------------------------------------------- */
ORC_API void orc_decode(const float *loc, const float *priors, int P, float var0, float var1,
                        float *boxes)
{
    for (int p = 0; p < P; ++p) {
        const float *l = loc + (size_t)p * 4, *pr = priors + (size_t)p * 4;
        float *b = boxes + (size_t)p * 4;
        /* priors[:, :2] + loc[:, :2] * variances[0] * priors[:, 2:] */
        const float cx = pr[0] + (l[0] * var0) * pr[2];
        const float cy = pr[1] + (l[1] * var0) * pr[3];
        /* priors[:, 2:] * exp(loc[:, 2:] * variances[1]) */
        const float w = pr[2] * expf(l[2] * var1);
        const float h = pr[3] * expf(l[3] * var1);
        const float x1 = cx - w / 2; /* boxes[:, :2] -= boxes[:, 2:] / 2 */
        const float y1 = cy - h / 2;
        b[0] = x1;
        b[1] = y1;
        b[2] = w + x1; /* boxes[:, 2:] += boxes[:, :2] */
        b[3] = h + y1;
    }
}

ORC_API void orc_center_size(const float *boxes, int P, float *out)
{
    for (int p = 0; p < P; ++p) {
        const float *b = boxes + (size_t)p * 4;
        float *o = out + (size_t)p * 4;
        o[0] = (b[2] + b[0]) / 2;
        o[1] = (b[3] + b[1]) / 2;
        o[2] = b[2] - b[0];
        o[3] = b[3] - b[1];
    }
}

/* ---------------------------------------------------------------------------------------
 * PriorBox.  layers/functions/prior_box.py:33-64.  Python float (= C double) arithmetic,
 * cast to fp32 (torch.Tensor(list)), then clamp to [0,1] when clip.  aspect_ratios is a
 * ragged list flattened as (ar_count[k], then values).  Returns the number of priors.
 * ------------------------------------------------------------------------------------- */
ORC_API int orc_prior_box(int n_maps, const int *feature_maps, double image_size,
                          const double *steps, const double *min_sizes, const double *max_sizes,
                          int n_max_sizes, const int *ar_count, const double *ars, int clip,
                          int flip, float *out)
{
    size_t n = 0;
    int ar_base = 0;
    for (int k = 0; k < n_maps; ++k) {
        const int f = feature_maps[k];
        for (int i = 0; i < f; ++i) {
            for (int j = 0; j < f; ++j) {
                const double f_k = image_size / steps[k];
                const double cx = (j + 0.5) / f_k;
                const double cy = (i + 0.5) / f_k;
                const double s_k = min_sizes[k] / image_size;
#define ORC_EMIT(a, b, c, d)                                                                       \
    do {                                                                                           \
        if (out) {                                                                                 \
            out[n * 4 + 0] = (float)(a);                                                           \
            out[n * 4 + 1] = (float)(b);                                                           \
            out[n * 4 + 2] = (float)(c);                                                           \
            out[n * 4 + 3] = (float)(d);                                                           \
        }                                                                                          \
        ++n;                                                                                       \
    } while (0)
                ORC_EMIT(cx, cy, s_k, s_k);
                if (n_max_sizes > 0) {
                    const double s_k_prime = sqrt(s_k * (max_sizes[k] / image_size));
                    ORC_EMIT(cx, cy, s_k_prime, s_k_prime);
                }
                for (int a = 0; a < ar_count[k]; ++a) {
                    const double ar = ars[ar_base + a];
                    ORC_EMIT(cx, cy, s_k * sqrt(ar), s_k / sqrt(ar));
                    if (flip) ORC_EMIT(cx, cy, s_k / sqrt(ar), s_k * sqrt(ar));
                }
            }
        }
        ar_base += ar_count[k];
    }
    if (out && clip) {
        for (size_t i = 0; i < n * 4; ++i) {
            if (out[i] > 1.f) out[i] = 1.f;
            if (out[i] < 0.f) out[i] = 0.f;
        }
    }
    return (int)n;
}

/* ---------------------------------------------------------------------------------------
 * Detect.  layers/functions/detection.py:25-70.  loc (B,P,4), conf (B*P,C), priors (P,4),
 * arm_loc (B,P,4) or NULL, scale[4]; out (B,C,top_k,5) zero-filled then rows
 * [score, x1,y1,x2,y2] (normalised boxes) for the first top_k NMS survivors per class
 * 1..C-1.  Lines 65-68 of the reference operate on a copy and are a no-op.
 * keep_counts (B*C, optional) receives the number of survivors BEFORE the top_k cut.
 * ------------------------------------------------------------------------------------- */
ORC_API void orc_detect(const float *loc, const float *conf, const float *priors,
                        const float *arm_loc, const float *scale, int B, int P, int C, int top_k,
                        float conf_thresh, double nms_thresh, float *out, int32_t *keep_counts)
{
    memset(out, 0, (size_t)B * C * top_k * 5 * sizeof(float));
    float *dflt = (float *)malloc((size_t)P * 4 * sizeof(float));
    float *tmp = (float *)malloc((size_t)P * 4 * sizeof(float));
    float *boxes = (float *)malloc((size_t)P * 4 * sizeof(float));
    float *dets = (float *)malloc((size_t)P * 5 * sizeof(float));
    int32_t *cidx = (int32_t *)malloc((size_t)P * sizeof(int32_t));
    int32_t *keep = (int32_t *)malloc((size_t)P * sizeof(int32_t));
    for (int b = 0; b < B; ++b) {
        const float *anchors = priors;
        if (arm_loc) { /* :43-45 */
            orc_decode(arm_loc + (size_t)b * P * 4, priors, P, 0.1f, 0.2f, tmp);
            orc_center_size(tmp, P, dflt);
            anchors = dflt;
        }
        orc_decode(loc + (size_t)b * P * 4, anchors, P, 0.1f, 0.2f, boxes); /* :48 */
        for (int cl = 1; cl < C; ++cl) {
            int n = 0;
            for (int p = 0; p < P; ++p) {
                const float s = conf[((size_t)b * P + p) * C + cl];
                if (s > conf_thresh) { /* .gt(), :53 */
                    dets[(size_t)n * 5 + 0] = boxes[(size_t)p * 4 + 0] * scale[0]; /* :59 */
                    dets[(size_t)n * 5 + 1] = boxes[(size_t)p * 4 + 1] * scale[1];
                    dets[(size_t)n * 5 + 2] = boxes[(size_t)p * 4 + 2] * scale[2];
                    dets[(size_t)n * 5 + 3] = boxes[(size_t)p * 4 + 3] * scale[3];
                    dets[(size_t)n * 5 + 4] = s;
                    cidx[n++] = p;
                }
            }
            int32_t nk = 0;
            if (n > 0) orc_cpu_nms(dets, n, nms_thresh, 0, keep, &nk); /* :60 */
            if (keep_counts) keep_counts[(size_t)b * C + cl] = nk;
            const int m = nk < top_k ? nk : top_k;
            for (int r = 0; r < m; ++r) { /* :61-63 */
                const int p = cidx[keep[r]];
                float *o = out + (((size_t)b * C + cl) * top_k + r) * 5;
                o[0] = conf[((size_t)b * P + p) * C + cl];
                o[1] = boxes[(size_t)p * 4 + 0];
                o[2] = boxes[(size_t)p * 4 + 1];
                o[3] = boxes[(size_t)p * 4 + 2];
                o[4] = boxes[(size_t)p * 4 + 3];
            }
        }
        if (keep_counts) keep_counts[(size_t)b * C] = 0;
    }
    free(dflt);
    free(tmp);
    free(boxes);
    free(dets);
    free(cidx);
    free(keep);
}

/* ---------------------------------------------------------------------------------------
 * L2Norm.  layers/modules/l2norm.py:17-21.  x NCHW; norm = sqrt(sum_c x^2) + eps;
 * out = weight[c] * (x / norm).
 * ------------------------------------------------------------------------------------- */
ORC_API void orc_l2norm(const float *x, const float *weight, int N, int C, int HW, float eps,
                        float *out)
{
    for (int n = 0; n < N; ++n) {
        for (int p = 0; p < HW; ++p) {
            float s = 0.f;
            for (int c = 0; c < C; ++c) {
                const float v = x[((size_t)n * C + c) * HW + p];
                s += v * v;
            }
            const float norm = sqrtf(s) + eps;
            for (int c = 0; c < C; ++c) {
                const size_t i = ((size_t)n * C + c) * HW + p;
                out[i] = weight[c] * (x[i] / norm);
            }
        }
    }
}

/* nn.Softmax(dim=1) over rows of (R, C).  model/dualrefinedet_vggbn.py:116-117, 196. */
ORC_API void orc_softmax_rows(const float *x, int R, int C, float *out)
{
    for (int r = 0; r < R; ++r) {
        const float *xr = x + (size_t)r * C;
        float *o = out + (size_t)r * C;
        float m = xr[0];
        for (int c = 1; c < C; ++c) m = xr[c] > m ? xr[c] : m;
        float s = 0.f;
        for (int c = 0; c < C; ++c) {
            o[c] = expf(xr[c] - m);
            s += o[c];
        }
        for (int c = 0; c < C; ++c) o[c] = o[c] / s;
    }
}
