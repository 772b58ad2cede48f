// layers.hip -- the HBM-bound layers of the trunk and heads, NHWC, 16 bytes per lane.
//   first conv (Cin = 3)      model/networks.py:145 (vgg), dualrefinedet_mobilenet.py:20
//   MaxPool2d 2x2 s2          model/networks.py:141-143,151
//   L2Norm                    layers/modules/l2norm.py:17-21
//   depthwise 3x3 + BN + ReLU model/networks.py:736-745
//   Softmax(dim=1)            model/dualrefinedet_vggbn.py:116-117,196
//   offset 1x1 convs          model/dualrefinedet_vggbn.py:53-57,71-76,155-164
#include <atomic>
#include <type_traits>

#include "kernels.h"

namespace tdrn {

// ---------------------------------------------------------------------------------------------
// First conv: NCHW fp32 image -> NHWC DT.  One thread = one output pixel; its 27 inputs sit in
// registers, the folded weights are wave-uniform (scalar loads), outputs leave as 16-B chunks.
// ---------------------------------------------------------------------------------------------
template <typename DT>
__global__ __launch_bounds__(256) void first_conv_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                         const float *__restrict__ bias, char *__restrict__ out,
                                                         int B, int S, int So, int stride, int Cout, int Cpad, int relu)
{
    constexpr int P16 = elem_traits<DT>::per16;
    constexpr int ES = elem_traits<DT>::bytes;
    // 256 pixels x 64 channels per pass are transposed through LDS so that the block writes whole
    // NHWC lines (a lane-per-pixel store touches 64 different 128-B lines per instruction and the
    // memory side sees 2.75x the bytes: measured with WRITE_SIZE).
    __shared__ __attribute__((aligned(16))) char tile[256 * (64 * ES + 16)];
    constexpr int TS = 64 * ES + 16;            // padded LDS row stride
    const long long total = (long long)B * So * So;
    const long long pix0 = (long long)blockIdx.x * 256;
    const long long pix = pix0 + threadIdx.x;
    const bool live = pix < total;
    float v[27];
    if (live) {
        const int b = (int)(pix / (So * So));
        const int rem = (int)(pix - (long long)b * So * So);
        const int ho = rem / So, wo = rem - ho * So;
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const int hi = ho * stride - 1 + r, wi = wo * stride - 1 + q;
                    const bool ok = (unsigned)hi < (unsigned)S && (unsigned)wi < (unsigned)S;
                    v[c * 9 + r * 3 + q] = ok ? x[(((size_t)b * 3 + c) * S + hi) * S + wi] : 0.f;
                }
    } else {
#pragma unroll
        for (int k = 0; k < 27; ++k) v[k] = 0.f;
    }
    const int npix = (int)((total - pix0) < 256 ? (total - pix0) : 256);
    for (int cb = 0; cb < Cpad; cb += 64) {
        for (int c0 = 0; c0 < 64; c0 += P16) {
            float o[P16];
#pragma unroll
            for (int j = 0; j < P16; ++j) {
                const int co = cb + c0 + j;          // wave-uniform
                float acc = 0.f;
                if (co < Cout) {
                    acc = bias[co];
#pragma unroll
                    for (int k = 0; k < 27; ++k) acc = fmaf(w[co * 27 + k], v[k], acc);
                    if (relu) acc = fmaxf(acc, 0.f);
                }
                o[j] = acc;
            }
            *(u32x4 *)(tile + threadIdx.x * TS + c0 * ES) = pack16<DT>(o);
        }
        __syncthreads();
        constexpr int CPR = 64 * ES / 16;           // 16-B chunks per pixel row of this channel block
        for (int i = threadIdx.x; i < npix * CPR; i += 256) {
            const int row = i / CPR, ch = i - row * CPR;
            *(u32x4 *)(out + ((size_t)(pix0 + row) * Cpad + cb) * ES + ch * 16) = *(const u32x4 *)(tile + row * TS + ch * 16);
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// First conv on the matrix cores, exact fp32: D[cout][pixel] = sum_k W[cout][k] X[k][pixel], k = (c, dy, dx)
// (27 -> 28) with v_mfma_f32_32x32x2_f32 -- 14 instructions per 32 couts x 32 pixels, 64 cycles each, so the
// layer is paced by the matrix pipe (~80 us at B = 32, 320x320) instead of the vector ALU (1728 FMAs a
// pixel, ~235 us) and sits beside its 0.42 GB of NHWC output writes.  A workgroup (4 waves) takes tiles
// of 8 output rows x 32 columns: the fp32 halo tile is staged in LDS (zero padding materialised there),
// each wave multiplies two rows, and the result leaves through a per-wave LDS transpose as whole
// 128-byte NHWC lines.  Weights (one value per lane per instruction) stay in registers across tiles.
// Couts beyond Cout (channel padding) are written as zeros.  Requires Cpad == 64 and stride 1 or 2.
// ---------------------------------------------------------------------------------------------
template <typename DT, int STRIDE>
__global__ __launch_bounds__(256) void first_conv_mfma_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                              const float *__restrict__ bias, char *__restrict__ out,
                                                              int B, int S, int So, int Cout, int relu, int tiles_x, int tiles_y)
{
    constexpr int ES = elem_traits<DT>::bytes;
    constexpr int TR = 8, TC = 32;                       // output tile
    constexpr int IH = (TR - 1) * STRIDE + 3, IW = (TC - 1) * STRIDE + 3;   // halo tile
    constexpr int NIN = 3 * IH * IW, NE = (NIN + 255) / 256;
    constexpr int TS = 64 * ES + 16;                     // staging row stride (one pixel = 64 couts)
    __shared__ __attribute__((aligned(16))) float xin[NIN];
    __shared__ __attribute__((aligned(16))) char stage[4 * 32 * TS];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r32 = lane & 31, hh = lane >> 5;

    // fp32 mode: exact fp32 on the matrix cores (v_mfma_f32_32x32x2_f32, 14 steps of K = 2 per 32 couts): lane (r32, hh)
    // holds W[32*ci + r32][2*s + hh] (A) and X[2*s + hh][pixel r32] (B).
    // 16-bit modes: like every other layer, inputs and weights are rounded to the net dtype and accumulated in fp32
    // (v_mfma_f32_32x32x16_*: K = 27 padded to 32 = two steps; lane (r32, hh) holds k = 16*ks + 8*hh + j, j = 0..7).
    // (pixel - mean) of 8-bit frames are integers below 256: exact in bf16 and fp16.)  The fp32 matrix rate is 1/16 of
    // the 16-bit one and made this layer MFMA-bound (28 x 64 cycles per 32 pixels x 64 couts, 81 us at batch 32) beside
    // a 420-MB output; with 4 x 32 cycles it runs at the speed of its stores.
    constexpr bool kHalf = ES == 2;
    constexpr int NS = kHalf ? 16 : 14;                  // LDS reads of my pixel per output row
    float wr[2][kHalf ? 1 : 14];
    u32x4 wq[2][2];                                      // 16-bit modes: packed weights [ci][k-step]
    int koff[NS];                                        // LDS offset of my k, relative to the pixel's halo origin
#pragma unroll
    for (int s2 = 0; s2 < NS; ++s2) {
        const int k = kHalf ? 16 * (s2 >> 3) + 8 * hh + (s2 & 7) : 2 * s2 + hh;
        const int c = k / 9, r = (k - 9 * c) / 3, q = k - 9 * c - 3 * r;
        koff[s2] = k < 27 ? (c * IH + r) * IW + q : 0;   // k >= 27 is the zero pad of K: its weight is 0
        if constexpr (!kHalf) {
#pragma unroll
            for (int ci = 0; ci < 2; ++ci) {
                const int co = ci * 32 + r32;
                wr[ci][s2] = (k < 27 && co < Cout) ? w[co * 27 + k] : 0.f;
            }
        }
    }
    if constexpr (kHalf) {
#pragma unroll
        for (int ci = 0; ci < 2; ++ci)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int co = ci * 32 + r32, k0 = 16 * ks + 8 * hh + 2 * jj;
                    const float a = (k0 < 27 && co < Cout) ? w[co * 27 + k0] : 0.f;
                    const float b2 = (k0 + 1 < 27 && co < Cout) ? w[co * 27 + k0 + 1] : 0.f;
                    wq[ci][ks][jj] = pack2<typename std::conditional<kHalf, DT, bf16_t>::type>(a, b2);
                }
    }
    // bias of the 64 couts in LDS (32 registers otherwise: they cost the fourth wave per SIMD)
    __shared__ __attribute__((aligned(16))) float bias_s[64];
    if (t < 64) bias_s[t] = t < Cout ? bias[t] : 0.f;
    __syncthreads();
    char *stg = stage + wave * 32 * TS;
    constexpr int CPR = 64 * ES / 16;                    // 16-B chunks per pixel
    constexpr int RPI = 64 / CPR;                        // pixels per wave store instruction
    const int my_ch = lane % CPR, my_row = lane / CPR;

    // halo elements e*256 + t of a tile: plane/row/col are tile-independent (compile-time divisors)
    int e_rq[NE], e_off[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int i = e * 256 + t;
        const int c = i / (IH * IW), rem = i - c * (IH * IW);
        const int r = rem / IW, q = rem - r * IW;
        e_rq[e] = i < NIN ? ((r << 16) | q) : -1;
        e_off[e] = (c * S + r) * S + q;
    }
    const int n_tiles = B * tiles_y * tiles_x;
    float pre[NE];
    auto fetch = [&](int tile) {                         // halo of `tile` -> registers (zero padding applied here)
        const int b = tile / (tiles_y * tiles_x), tt = tile - b * tiles_y * tiles_x;
        const int ty = tt / tiles_x, tx = tt - ty * tiles_x;
        const int iy0 = ty * TR * STRIDE - 1, ix0 = tx * TC * STRIDE - 1;
        const float *xb = x + (size_t)b * 3 * S * S + (long long)iy0 * S + ix0;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int yy = iy0 + (e_rq[e] >> 16), xx = ix0 + (e_rq[e] & 0xffff);
            const bool ok = e_rq[e] >= 0 && (unsigned)yy < (unsigned)S && (unsigned)xx < (unsigned)S;
            pre[e] = ok ? xb[e_off[e]] : 0.f;
        }
    };
    if ((int)blockIdx.x < n_tiles) fetch(blockIdx.x);
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int b = tile / (tiles_y * tiles_x), tt = tile - b * tiles_y * tiles_x;
        const int ty = tt / tiles_x, tx = tt - ty * tiles_x;
        const int oy0 = ty * TR, ox0 = tx * TC;
        __syncthreads();                                 // the previous tile's reads of xin are done
#pragma unroll
        for (int e = 0; e < NE; ++e)
            if (e * 256 + t < NIN) xin[e * 256 + t] = pre[e];
        __syncthreads();
        if (tile + (int)gridDim.x < n_tiles) fetch(tile + gridDim.x);       // lands while this tile multiplies
#pragma unroll 1
        for (int rr = 0; rr < 2; ++rr) {
            const int orow = 2 * wave + rr;              // tile-local output row of this pass
            const int oy = oy0 + orow;
            if (oy >= So) break;                         // wave-uniform
            f32x16 acc[2];
#pragma unroll
            for (int ci = 0; ci < 2; ++ci)
#pragma unroll
                for (int g = 0; g < 4; ++g) {            // accumulator rows: cout = 32*ci + 8*g + 4*hh + j
                    const f32x4 bv = *(const f32x4 *)(bias_s + ci * 32 + 8 * g + 4 * hh);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[ci][4 * g + j] = bv[j];
                }
            const int porg = (orow * STRIDE) * IW + r32 * STRIDE;      // halo origin of my pixel
            float xv[NS];
#pragma unroll
            for (int s2 = 0; s2 < NS; ++s2) xv[s2] = xin[porg + koff[s2]];
            if constexpr (kHalf) {
                typedef typename std::conditional<kHalf, DT, bf16_t>::type HT;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    u32x4 xq;
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) xq[jj] = pack2<HT>(xv[8 * ks + 2 * jj], xv[8 * ks + 2 * jj + 1]);
#pragma unroll
                    for (int ci = 0; ci < 2; ++ci) {
                        if constexpr (__is_same(HT, bf16_t))
                            acc[ci] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(i16x8, wq[ci][ks]), __builtin_bit_cast(i16x8, xq), acc[ci], 0, 0, 0);
                        else
                            acc[ci] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wq[ci][ks]), __builtin_bit_cast(f16x8, xq), acc[ci], 0, 0, 0);
                    }
                }
            } else {
#pragma unroll
                for (int s2 = 0; s2 < 14; ++s2)
#pragma unroll
                    for (int ci = 0; ci < 2; ++ci)
                        acc[ci] = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[ci][s2], xv[s2], acc[ci], 0, 0, 0);
            }
            // ---- ReLU, convert, transpose through LDS, whole-line stores ----
#pragma unroll
            for (int ci = 0; ci < 2; ++ci)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float q4[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) q4[j] = relu ? fmaxf(acc[ci][4 * g + j], 0.f) : acc[ci][4 * g + j];
                    char *d = stg + r32 * TS + (ci * 32 + 8 * g + 4 * hh) * ES;
                    if constexpr (ES == 4) {
                        *(f32x4 *)d = f32x4{q4[0], q4[1], q4[2], q4[3]};
                    } else {
                        *(uint2 *)d = make_uint2(pack2<DT>(q4[0], q4[1]), pack2<DT>(q4[2], q4[3]));
                    }
                }
            __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0): my LDS writes are done
            __builtin_amdgcn_wave_barrier();
            const size_t prow = ((size_t)b * So + oy) * So + ox0;
#pragma unroll
            for (int k = 0; k < 32 / RPI; ++k) {
                const int px = my_row + k * RPI;
                if (ox0 + px < So)
                    *(u32x4 *)(out + (prow + px) * 64 * ES + my_ch * 16) = *(const u32x4 *)(stg + px * TS + my_ch * 16);
            }
            __builtin_amdgcn_wave_barrier();             // (a wave's DS instructions execute in order)
        }
    }
}

int launch_first_conv(const float *x, const float *w, const float *bias, void *out, int B, int S, int stride,
                      int Cout, int Cpad, int relu, int dtype, hipStream_t s)
{
    const int So = (S + 2 - 3) / stride + 1;
    static int use_mfma = -1;
    if (use_mfma < 0) { const char *e = getenv("TDRN_FIRST_MFMA"); use_mfma = e ? atoi(e) : 1; }
    if (use_mfma && Cpad == 64 && (stride == 1 || stride == 2)) {
        const int tiles_x = (So + 31) / 32, tiles_y = (So + 7) / 8;
        const long long n_tiles = (long long)B * tiles_x * tiles_y;
#define LM(DT)                                                                                                              \
    do {                                                                                                                    \
        auto k1 = first_conv_mfma_kernel<DT, 1>;                                                                            \
        auto k2 = first_conv_mfma_kernel<DT, 2>;                                                                            \
        static int fit[2] = {0, 0};   /* persistent grid = what fits at once (register-limited: 3-4 workgroups per CU) */   \
        int &per_cu = fit[stride - 1];                                                                                      \
        if (per_cu == 0) TDRN_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, stride == 1 ? k1 : k2, 256, 0)); \
        const long long cap = 256ll * (per_cu > 0 ? per_cu : 1);                                                            \
        dim3 grid((unsigned)(n_tiles < cap ? n_tiles : cap));                                                               \
        if (stride == 1) hipLaunchKernelGGL(k1, grid, dim3(256), 0, s, x, w, bias, (char *)out, B, S, So, Cout, relu, tiles_x, tiles_y); \
        else hipLaunchKernelGGL(k2, grid, dim3(256), 0, s, x, w, bias, (char *)out, B, S, So, Cout, relu, tiles_x, tiles_y);             \
    } while (0)
        if (dtype == TDRN_F32) LM(float); else if (dtype == TDRN_BF16) LM(bf16_t); else LM(f16_t);
#undef LM
        return hip_status(hipGetLastError());
    }
    const long long total = (long long)B * So * So;
    dim3 grid((unsigned)((total + 255) / 256));
#define L(DT) hipLaunchKernelGGL((first_conv_kernel<DT>), grid, dim3(256), 0, s, x, w, bias, (char *)out, B, S, So, stride, Cout, Cpad, relu)
    if (dtype == TDRN_F32) L(float); else if (dtype == TDRN_BF16) L(bf16_t); else L(f16_t);
#undef L
    return hip_status(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// MaxPool 2x2 stride 2 (ceil_mode: windows clipped at the border)
// ---------------------------------------------------------------------------------------------
template <typename DT>
__global__ __launch_bounds__(256) void maxpool2_kernel(const char *__restrict__ in, char *__restrict__ out, int B, int H,
                                                       int W, int Ho, int Wo, int C)
{
    constexpr int P16 = elem_traits<DT>::per16;
    constexpr int ES = elem_traits<DT>::bytes;
    const int cpr = C / P16;
    const long long total = (long long)B * Ho * Wo * cpr;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int ch = (int)(i % cpr);
        long long pix = i / cpr;
        const int wo = (int)(pix % Wo);
        pix /= Wo;
        const int ho = (int)(pix % Ho);
        const int b = (int)(pix / Ho);
        float m[P16];
#pragma unroll
        for (int j = 0; j < P16; ++j) m[j] = -INFINITY;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int hi = 2 * ho + dy, wi = 2 * wo + dx;
                if (hi < H && wi < W) {
                    float v[P16];
                    unpack16<DT>(*(const u32x4 *)(in + ((((size_t)b * H + hi) * W + wi) * C + (size_t)ch * P16) * ES), v);
#pragma unroll
                    for (int j = 0; j < P16; ++j) m[j] = fmaxf(m[j], v[j]);
                }
            }
        *(u32x4 *)(out + ((((size_t)b * Ho + ho) * Wo + wo) * C + (size_t)ch * P16) * ES) = pack16<DT>(m);
    }
}

int launch_maxpool2(const void *in, void *out, int B, int H, int W, int C, int ceil_mode, int dtype, hipStream_t s)
{
    const int Ho = ceil_mode ? (H + 1) / 2 : H / 2, Wo = ceil_mode ? (W + 1) / 2 : W / 2;
    const int per16 = 16 / dtype_bytes(dtype);
    if (C % per16) return TDRN_E_UNSUPPORTED;
    const long long total = (long long)B * Ho * Wo * (C / per16);
    dim3 grid((unsigned)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256));
#define L(DT) hipLaunchKernelGGL((maxpool2_kernel<DT>), grid, dim3(256), 0, s, (const char *)in, (char *)out, B, H, W, Ho, Wo, C)
    if (dtype == TDRN_F32) L(float); else if (dtype == TDRN_BF16) L(bf16_t); else L(f16_t);
#undef L
    return hip_status(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// L2Norm: one wavefront per pixel, 16 B per lane per step, wave-wide butterfly reduction.
// ---------------------------------------------------------------------------------------------
template <typename DT>
__global__ __launch_bounds__(256) void l2norm_kernel(const char *__restrict__ in, const float *__restrict__ w,
                                                     char *__restrict__ out, long long pixels, int C)
{
    constexpr int P16 = elem_traits<DT>::per16;
    constexpr int ES = elem_traits<DT>::bytes;
    constexpr int MAXS = 8;                      // up to 64*8*8 = 4096 channels (16-bit)
    const int lane = threadIdx.x & 63;
    const long long pix = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (pix >= pixels) return;
    const int cpr = C / P16;
    const char *src = in + (size_t)pix * C * ES;
    float v[MAXS][P16];
    float ss = 0.f;
#pragma unroll
    for (int st = 0; st < MAXS; ++st) {
        const int ch = st * 64 + lane;
        if (ch < cpr) {
            unpack16<DT>(*(const u32x4 *)(src + (size_t)ch * 16), v[st]);
#pragma unroll
            for (int j = 0; j < P16; ++j) ss = fmaf(v[st][j], v[st][j], ss);
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) ss += __shfl_xor(ss, o, 64);
    const float norm = sqrtf(ss) + 1e-10f;
    char *dst = out + (size_t)pix * C * ES;
#pragma unroll
    for (int st = 0; st < MAXS; ++st) {
        const int ch = st * 64 + lane;
        if (ch < cpr) {
            float o[P16];
#pragma unroll
            for (int j = 0; j < P16; ++j) o[j] = w[ch * P16 + j] * (v[st][j] / norm);
            *(u32x4 *)(dst + (size_t)ch * 16) = pack16<DT>(o);
        }
    }
}

int launch_l2norm(const void *in, const float *w, void *out, long long pixels, int C, int dtype, hipStream_t s)
{
    const int per16 = 16 / dtype_bytes(dtype);
    if (C % per16 || C / per16 > 64 * 8) return TDRN_E_UNSUPPORTED;
    dim3 grid((unsigned)((pixels + 3) / 4));
#define L(DT) hipLaunchKernelGGL((l2norm_kernel<DT>), grid, dim3(256), 0, s, (const char *)in, w, (char *)out, pixels, C)
    if (dtype == TDRN_F32) L(float); else if (dtype == TDRN_BF16) L(bf16_t); else L(f16_t);
#undef L
    return hip_status(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// Depthwise 3x3 (pad 1) + folded BN + ReLU.  Thread = (pixel, 16-B channel chunk).
// ---------------------------------------------------------------------------------------------
template <typename DT>
__global__ __launch_bounds__(256) void dwconv3_kernel(const char *__restrict__ in, const float *__restrict__ w,
                                                      const float *__restrict__ bias, char *__restrict__ out, int B,
                                                      int H, int W, int Ho, int Wo, int C, int stride, int relu)
{
    constexpr int P16 = elem_traits<DT>::per16;
    constexpr int ES = elem_traits<DT>::bytes;
    const int cpr = C / P16;
    const long long total = (long long)B * Ho * Wo * cpr;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int ch = (int)(i % cpr);
        long long pix = i / cpr;
        const int wo = (int)(pix % Wo);
        pix /= Wo;
        const int ho = (int)(pix % Ho);
        const int b = (int)(pix / Ho);
        float acc[P16];
#pragma unroll
        for (int j = 0; j < P16; ++j) acc[j] = bias[ch * P16 + j];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int hi = ho * stride - 1 + r, wi = wo * stride - 1 + q;
                if ((unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W) {
                    float v[P16];
                    unpack16<DT>(*(const u32x4 *)(in + ((((size_t)b * H + hi) * W + wi) * C + (size_t)ch * P16) * ES), v);
                    const float *wt = w + (size_t)(r * 3 + q) * C + ch * P16;
#pragma unroll
                    for (int j = 0; j < P16; ++j) acc[j] = fmaf(wt[j], v[j], acc[j]);
                }
            }
        if (relu) {
#pragma unroll
            for (int j = 0; j < P16; ++j) acc[j] = fmaxf(acc[j], 0.f);
        }
        *(u32x4 *)(out + ((((size_t)b * Ho + ho) * Wo + wo) * C + (size_t)ch * P16) * ES) = pack16<DT>(acc);
    }
}

// Row-strip version: a thread produces TW = 4 consecutive outputs of one row for one 16-B channel chunk.  The
// three input rows are walked once, every loaded column feeds up to three outputs (18 loads per 4 outputs at
// stride 1 instead of 36; 27 instead of 36 at stride 2), and when the channel chunk of a thread is the same in
// every grid-stride iteration (256 % (C/P16) == 0) its 9 x P16 fp32 weights stay in registers for the launch
// instead of being re-read per output.  Same arithmetic order per output as dwconv3_kernel (taps row-major).
template <typename DT, int STRIDE>
__global__ __launch_bounds__(256) void dwconv3_strip_kernel(const char *__restrict__ in, const float *__restrict__ w,
                                                            const float *__restrict__ bias, char *__restrict__ out, int B,
                                                            int H, int W, int Ho, int Wo, int C, int relu)
{
    constexpr int P16 = elem_traits<DT>::per16;
    constexpr int ES = elem_traits<DT>::bytes;
    constexpr int TW = 4;
    constexpr int NCOL = (TW - 1) * STRIDE + 3;            // input columns under one strip
    const int cpr = C / P16;
    const int wg = (Wo + TW - 1) / TW;                     // strips per output row
    const long long total = (long long)B * Ho * wg * cpr;
    const long long step = (long long)gridDim.x * blockDim.x;
    const bool fixed_ch = (256 % cpr) == 0;                // then i % cpr never changes for this thread
    float wt[9][P16], bs[P16];
    int ch_loaded = -1;
    // (measured and rejected, round 4: an XCD-aware block order -- every XCD a contiguous range of output rows, so that the rows
    // shared by vertically adjacent strips are fetched into ONE L2; FETCH_SIZE says every input row is fetched ~1.8 times -- made
    // the large maps SLOWER (160x160: 120 -> 150 us, 40x40 x 512 ch: 67 -> 73 us) and only the 20x20 maps faster (34 -> 30 us))
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += step) {
        const int ch = fixed_ch ? (int)(threadIdx.x % cpr) : (int)(i % cpr);
        long long r = i / cpr;
        const int sg = (int)(r % wg);
        r /= wg;
        const int ho = (int)(r % Ho);
        const int b = (int)(r / Ho);
        if (ch != ch_loaded) {
            ch_loaded = ch;
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int j = 0; j < P16; ++j) wt[t][j] = w[(size_t)t * C + ch * P16 + j];
#pragma unroll
            for (int j = 0; j < P16; ++j) bs[j] = bias[ch * P16 + j];
        }
        const int wo0 = sg * TW;
        float acc[TW][P16];
#pragma unroll
        for (int o = 0; o < TW; ++o)
#pragma unroll
            for (int j = 0; j < P16; ++j) acc[o][j] = bs[j];
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {
            const int hi = ho * STRIDE - 1 + rr;
            if ((unsigned)hi >= (unsigned)H) continue;     // zero padding row: contributes nothing
            const char *rowp = in + (((size_t)b * H + hi) * W) * C * ES + (size_t)ch * 16;
            u32x4 raw[NCOL];
#pragma unroll
            for (int c = 0; c < NCOL; ++c) {
                const int wi = wo0 * STRIDE - 1 + c;
                raw[c] = (unsigned)wi < (unsigned)W ? *(const u32x4 *)(rowp + (size_t)wi * C * ES) : u32x4{0u, 0u, 0u, 0u};
            }
#pragma unroll
            for (int c = 0; c < NCOL; ++c) {
                float v[P16];
                unpack16<DT>(raw[c], v);
#pragma unroll
                for (int o = 0; o < TW; ++o) {
                    const int q = c - o * STRIDE;          // tap column of output o that reads input column c
                    if (q >= 0 && q < 3) {
#pragma unroll
                        for (int j = 0; j < P16; ++j) acc[o][j] = fmaf(wt[rr * 3 + q][j], v[j], acc[o][j]);
                    }
                }
            }
        }
#pragma unroll
        for (int o = 0; o < TW; ++o) {
            if (wo0 + o >= Wo) break;
            if (relu) {
#pragma unroll
                for (int j = 0; j < P16; ++j) acc[o][j] = fmaxf(acc[o][j], 0.f);
            }
            *(u32x4 *)(out + ((((size_t)b * Ho + ho) * Wo + wo0 + o) * C + (size_t)ch * P16) * ES) = pack16<DT>(acc[o]);
        }
    }
}

// Sliding-window version of the strip kernel: a thread keeps its TW-output strip and walks DOWN a segment of TH output rows
// with the input rows it has already loaded kept in registers, so an input row is loaded once per segment (+ the segment's
// halo rows) instead of once per output row that touches it.  Why: the strip kernel's consecutive output rows land in
// different workgroups = different XCDs, each of which pulls the shared input rows through its own L2 -- FETCH_SIZE said 3.2
// times the input of a stride-1 layer crossed the fabric (profiles/r04_cfg4/traffic_by_kernel.csv: 502 MB per launch for
// 240 MB of tensor, 6.6 TB/s: the launch is bound by exactly that traffic).  The next row's loads are issued before the
// current row's arithmetic.  Per output the taps are accumulated in the strip kernel's order: same bits.
template <typename DT, int STRIDE, int TW>
__global__ __launch_bounds__(256, 2) void dwconv3_slide_kernel(const char *__restrict__ in, const float *__restrict__ w,
                                                            const float *__restrict__ bias, char *__restrict__ out, int B,
                                                            int H, int W, int Ho, int Wo, int C, int relu, int TH, int nseg)
{
    constexpr int P16 = elem_traits<DT>::per16;
    constexpr int ES = elem_traits<DT>::bytes;
    constexpr int NCOL = (TW - 1) * STRIDE + 3;
    const int cpr = C / P16;
    const int wg = (Wo + TW - 1) / TW;
    const long long total = (long long)B * nseg * wg * cpr;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int ch = (int)(i % cpr);
    long long r = i / cpr;
    const int sg = (int)(r % wg);
    r /= wg;
    const int seg = (int)(r % nseg);
    const int b = (int)(r / nseg);
    float wt[9][P16], bs[P16];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < P16; ++j) wt[t][j] = w[(size_t)t * C + ch * P16 + j];
#pragma unroll
    for (int j = 0; j < P16; ++j) bs[j] = bias[ch * P16 + j];
    const int wo0 = sg * TW;
    const int ho0 = seg * TH, ho1 = ho0 + TH < Ho ? ho0 + TH : Ho;
    const char *colp = in + ((size_t)b * H * W) * C * ES + (size_t)ch * 16;
    struct Row { u32x4 v[NCOL]; };
    auto load_row = [&](int hi) -> Row {                   // (a row outside the image is never USED: see `add_row`)
        Row q;
        const bool rok = (unsigned)hi < (unsigned)H;
        const char *rowp = colp + (size_t)(rok ? hi : 0) * W * C * ES;
#pragma unroll
        for (int c = 0; c < NCOL; ++c) {
            const int wi = wo0 * STRIDE - 1 + c;
            q.v[c] = (rok && (unsigned)wi < (unsigned)W) ? *(const u32x4 *)(rowp + (size_t)wi * C * ES) : u32x4{0u, 0u, 0u, 0u};
        }
        return q;
    };
    float acc[TW][P16];
    auto add_row = [&](const Row &q, int rr, int hi) {
        if ((unsigned)hi >= (unsigned)H) return;           // zero padding row: contributes nothing (as the strip kernel)
#pragma unroll
        for (int c = 0; c < NCOL; ++c) {
            float v[P16];
            unpack16<DT>(q.v[c], v);
#pragma unroll
            for (int o = 0; o < TW; ++o) {
                const int t = c - o * STRIDE;
                if (t >= 0 && t < 3) {
#pragma unroll
                    for (int j = 0; j < P16; ++j) acc[o][j] = fmaf(wt[rr * 3 + t][j], v[j], acc[o][j]);
                }
            }
        }
    };
    auto finish = [&](int ho) {
#pragma unroll
        for (int o = 0; o < TW; ++o) {
            if (wo0 + o >= Wo) break;
            if (relu) {
#pragma unroll
                for (int j = 0; j < P16; ++j) acc[o][j] = fmaxf(acc[o][j], 0.f);
            }
            *(u32x4 *)(out + ((((size_t)b * Ho + ho) * Wo + wo0 + o) * C + (size_t)ch * P16) * ES) = pack16<DT>(acc[o]);
        }
    };
    auto start = [&]() {
#pragma unroll
        for (int o = 0; o < TW; ++o)
#pragma unroll
            for (int j = 0; j < P16; ++j) acc[o][j] = bs[j];
    };
    if (STRIDE == 1) {
        Row ra = load_row(ho0 - 1), rb = load_row(ho0), rc = load_row(ho0 + 1);
#pragma unroll 1
        for (int ho = ho0; ho < ho1; ++ho) {
            start();
            add_row(ra, 0, ho - 1);
            ra = load_row(ho + 2 < ho1 + 1 ? ho + 2 : -1);         // (into the registers of the row just used up; in flight under the
            add_row(rb, 1, ho);                                    // other two rows' arithmetic; nothing past the segment)
            add_row(rc, 2, ho + 1);
            finish(ho);
            const Row t = rb;
            rb = rc; rc = ra; ra = t;
        }
    } else {
        Row ra = load_row(2 * ho0 - 1), rb = load_row(2 * ho0), rc = load_row(2 * ho0 + 1);
#pragma unroll 1
        for (int ho = ho0; ho < ho1; ++ho) {
            const bool more = ho + 1 < ho1;
            Row nb = load_row(more ? 2 * ho + 2 : -1), nc = load_row(more ? 2 * ho + 3 : -1);
            start();
            add_row(ra, 0, 2 * ho - 1);
            add_row(rb, 1, 2 * ho);
            add_row(rc, 2, 2 * ho + 1);
            finish(ho);
            ra = rc; rb = nb; rc = nc;
        }
    }
}

int launch_dwconv3(const void *in, const float *w, const float *bias, void *out, int B, int H, int W, int C,
                   int stride, int relu, int dtype, hipStream_t s, int kdisable)
{
    const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
    const int per16 = 16 / dtype_bytes(dtype);
    if (C % per16) return TDRN_E_UNSUPPORTED;
    static int strip = -1, slide = -1;
    if (strip < 0) { const char *e = getenv("TDRN_DW_STRIP"); strip = e ? atoi(e) : 1; }
    if (slide < 0) { const char *e = getenv("TDRN_DW_SLIDE"); slide = e ? atoi(e) : 1; }
    if (slide && !(kdisable & 16) && strip && (stride == 1 || stride == 2) && Wo >= 4) {
        // strips of 4 outputs at stride 1, of 2 at stride 2 (9 input columns per row do not fit the registers of two waves per
        // SIMD there).  Segment height: the tallest of 8 / 4 / 2 rows that still leaves >= 768 workgroups (3 per CU); below
        // that the strip kernel (one row per thread).  Measured on dualrefinedet_mobilenet 320 x 64 (profiles/r04_experiments.md):
        // 8 rows against 4 / 16: equal / 10-25 % slower; 2 waves per SIMD against a forced 3 (spills): 3x faster.
        const int tw = stride == 1 ? 4 : 2;
        const long long per_row = (long long)B * ((Wo + tw - 1) / tw) * (C / per16);
        int th = 0;
        for (int t = 8; t >= 2 && !th; t >>= 1)
            if (per_row * ((Ho + t - 1) / t) >= 768ll * 256) th = t;
        if (slide > 1) th = slide;                          // (experiments: a forced segment height)
        if (kdisable & 32) th = 8;                          // (TDRN_PLAN_DW_SLIDE_ALL)
        if (th) {
            const int nseg = (Ho + th - 1) / th;
            dim3 grid((unsigned)((per_row * nseg + 255) / 256));
#define LD(DT)                                                                                                                          \
    do {                                                                                                                                \
        if (stride == 1) hipLaunchKernelGGL((dwconv3_slide_kernel<DT, 1, 4>), grid, dim3(256), 0, s, (const char *)in, w, bias, (char *)out, B, H, W, Ho, Wo, C, relu, th, nseg); \
        else hipLaunchKernelGGL((dwconv3_slide_kernel<DT, 2, 2>), grid, dim3(256), 0, s, (const char *)in, w, bias, (char *)out, B, H, W, Ho, Wo, C, relu, th, nseg);             \
    } while (0)
            if (dtype == TDRN_F32) LD(float); else if (dtype == TDRN_BF16) LD(bf16_t); else LD(f16_t);
#undef LD
            return hip_status(hipGetLastError());
        }
    }
    if (strip && (stride == 1 || stride == 2) && Wo >= 4) {
        const long long total = (long long)B * Ho * ((Wo + 3) / 4) * (C / per16);
        dim3 grid((unsigned)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256));
#define LS(DT)                                                                                                                         \
    do {                                                                                                                               \
        if (stride == 1) hipLaunchKernelGGL((dwconv3_strip_kernel<DT, 1>), grid, dim3(256), 0, s, (const char *)in, w, bias, (char *)out, B, H, W, Ho, Wo, C, relu); \
        else hipLaunchKernelGGL((dwconv3_strip_kernel<DT, 2>), grid, dim3(256), 0, s, (const char *)in, w, bias, (char *)out, B, H, W, Ho, Wo, C, relu);             \
    } while (0)
        if (dtype == TDRN_F32) LS(float); else if (dtype == TDRN_BF16) LS(bf16_t); else LS(f16_t);
#undef LS
        return hip_status(hipGetLastError());
    }
    const long long total = (long long)B * Ho * Wo * (C / per16);
    dim3 grid((unsigned)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256));
#define L(DT) hipLaunchKernelGGL((dwconv3_kernel<DT>), grid, dim3(256), 0, s, (const char *)in, w, bias, (char *)out, B, H, W, Ho, Wo, C, stride, relu)
    if (dtype == TDRN_F32) L(float); else if (dtype == TDRN_BF16) L(bf16_t); else L(f16_t);
#undef L
    return hip_status(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// Row softmax over C classes: rows staged through LDS so that global traffic is coalesced although a row is C*4 bytes; one thread
// owns one row.  A workgroup takes `rb` rows (256 up to 59 classes -- VOC's 21, VID's 31 --, 128 up to 119 -- COCO's 81 --, 64 beyond):
// rb * (C | 1) floats of LDS stay under 64 KiB.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float *__restrict__ in, float *__restrict__ out,
                                                           long long R, int C, int rb)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const long long r0 = (long long)blockIdx.x * rb;
    const int rows = (int)((R - r0) < rb ? (R - r0) : rb);
    const int n = rows * C;
    const int ld = C | 1;                       // odd stride: conflict-free row-per-thread access
    for (int i = threadIdx.x; i < n; i += 256) sm[(i / C) * ld + (i % C)] = in[r0 * C + i];
    __syncthreads();
    if ((int)threadIdx.x < rows) {
        float *row = sm + threadIdx.x * ld;
        float m = row[0];
        for (int c = 1; c < C; ++c) m = fmaxf(m, row[c]);
        float ssum = 0.f;
        for (int c = 0; c < C; ++c) {
            const float e = expf(row[c] - m);
            row[c] = e;
            ssum += e;
        }
        for (int c = 0; c < C; ++c) row[c] = row[c] / ssum;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 256) out[r0 * C + i] = sm[(i / C) * ld + (i % C)];
}

int launch_softmax_rows(const float *in, float *out, long long R, int C, hipStream_t s)
{
    if (C < 1 || C > 255) return TDRN_E_UNSUPPORTED;
    if (R <= 0) return TDRN_OK;
    const int rb = C <= 59 ? 256 : (C <= 119 ? 128 : 64);
    dim3 grid((unsigned)((R + rb - 1) / rb));
    hipLaunchKernelGGL(softmax_rows_kernel, grid, dim3(256), rb * (size_t)(C | 1) * sizeof(float), s, in, out, R, C, rb);
    return hip_status(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// offset convs: off[m][o] = bias[o] + sum_j w[o][j] * loc[m][j]   (n_in = 12, n_out = 18+50)
// ---------------------------------------------------------------------------------------------
// One workgroup = 64 pixels: their loc rows (n_in <= 16 values each) and the whole weight matrix (n_out <= 128 rows) sit in
// LDS (blockIdx.y = slice of 128 output channels), every thread produces outputs idx = t, t + 256, ... of the 64 x 128 block, so
// stores are contiguous runs and
// no thread divides a 64-bit index (the first version did, per output element: 49 us for level 0 at batch 32, now 6).
__global__ __launch_bounds__(256) void offset_conv_kernel(const float *__restrict__ loc, long long loc_bs,
                                                          long long loc_ps, const float *__restrict__ w,
                                                          const float *__restrict__ bias, float *__restrict__ off,
                                                          int B, int HW, int n_in, int n_out)
{
    __shared__ float s_w[128 * 16], s_b[128], s_l[64 * 16];
    const int t = threadIdx.x;
    const long long M = (long long)B * HW;
    const int oc0 = blockIdx.y * 128;                    // this workgroup's slice of the output channels
    const int nc = n_out - oc0 < 128 ? n_out - oc0 : 128;
    for (int i = t; i < nc * n_in; i += 256) s_w[i] = w[oc0 * n_in + i];
    for (int i = t; i < nc; i += 256) s_b[i] = bias ? bias[oc0 + i] : 0.f;
    for (long long m0 = (long long)blockIdx.x * 64; m0 < M; m0 += (long long)gridDim.x * 64) {
        __syncthreads();
        for (int i = t; i < 64 * n_in; i += 256) {
            const int r = i / n_in, j = i - r * n_in;
            const long long m = m0 + r;
            float v = 0.f;
            if (m < M) {
                const int b = (int)(m / HW), pix = (int)(m - (long long)b * HW);
                v = loc[b * loc_bs + pix * loc_ps + j];
            }
            s_l[i] = v;
        }
        __syncthreads();
        const int rows = M - m0 < 64 ? (int)(M - m0) : 64;
        const int total = rows * nc;
        float *dst = off + m0 * n_out + oc0;
        for (int idx = t; idx < total; idx += 256) {
            const int r = idx / nc, o = idx - r * nc;
            float acc = s_b[o];
            for (int j = 0; j < n_in; ++j) acc = fmaf(s_w[o * n_in + j], s_l[r * n_in + j], acc);
            dst[(long long)r * n_out + o] = acc;
        }
    }
}

int launch_offset_conv(const float *loc, long long loc_bs, long long loc_ps, const float *w, const float *bias,
                       float *off, int B, int HW, int n_in, int n_out, hipStream_t s)
{
    const long long M = (long long)B * HW;
    if (M <= 0 || n_out <= 0) return TDRN_OK;
    if (n_in > 16) return TDRN_E_UNSUPPORTED;
    const long long blocks = (M + 63) / 64;
    dim3 grid((unsigned)(blocks > 4096 ? 4096 : blocks), (unsigned)((n_out + 127) / 128));
    hipLaunchKernelGGL(offset_conv_kernel, grid, dim3(256), 0, s, loc, loc_bs, loc_ps, w, bias, off, B, HW, n_in, n_out);
    return hip_status(hipGetLastError());
}

// The offset convs of all pyramid levels in ONE launch (round 6: four launches of 16-25 us each -- most of it the launch itself --
// sat back to back on the ARM side lane).  blockIdx.x = 64-pixel block of problem p (pr[p].blk0 <= blockIdx.x < pr[p + 1].blk0),
// blockIdx.y = 128-column slice; the body is offset_conv_kernel's, per-output arithmetic and order unchanged (bit-identical).
__global__ __launch_bounds__(256) void offset_conv_multi_kernel(const OffsetMulti mp)
{
    __shared__ float s_w[128 * 16], s_b[128], s_l[64 * 16];
    const int t = threadIdx.x;
    int q = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i)
        if (i < mp.n && (int)blockIdx.x >= mp.pr[i].blk0) q = i;
    const OffsetProblem &pr = mp.pr[q];
    const int n_in = 12, n_out = pr.n_out;
    const int oc0 = blockIdx.y * 128;
    if (oc0 >= n_out) return;
    const int nc = n_out - oc0 < 128 ? n_out - oc0 : 128;
    const long long M = (long long)pr.B * pr.HW;
    const long long m0 = (long long)((int)blockIdx.x - pr.blk0) * 64;
    for (int i = t; i < nc * n_in; i += 256) s_w[i] = pr.w[oc0 * n_in + i];
    for (int i = t; i < nc; i += 256) s_b[i] = pr.bias ? pr.bias[oc0 + i] : 0.f;
    for (int i = t; i < 64 * n_in; i += 256) {
        const int r = i / n_in, j = i - r * n_in;
        const long long m = m0 + r;
        float v = 0.f;
        if (m < M) {
            const int b = (int)(m / pr.HW), pix = (int)(m - (long long)b * pr.HW);
            v = pr.loc[b * pr.loc_bs + pix * pr.loc_ps + j];
        }
        s_l[i] = v;
    }
    __syncthreads();
    const int rows = M - m0 < 64 ? (int)(M - m0) : 64;
    const int total = rows * nc;
    float *dst = pr.off + m0 * n_out + oc0;
    for (int idx = t; idx < total; idx += 256) {
        const int r = idx / nc, o = idx - r * nc;
        float acc = s_b[o];
        for (int j = 0; j < n_in; ++j) acc = fmaf(s_w[o * n_in + j], s_l[r * n_in + j], acc);
        dst[(long long)r * n_out + o] = acc;
    }
}

int launch_offset_conv_multi(const OffsetProblem *pr, int n, hipStream_t s)
{
    if (n <= 0) return TDRN_OK;
    if (n > 4) return TDRN_E_ARG;
    OffsetMulti mp;
    mp.n = 0;
    int blocks = 0, max_out = 0;
    for (int i = 0; i < n; ++i) {
        const long long M = (long long)pr[i].B * pr[i].HW;
        if (M <= 0 || pr[i].n_out <= 0) continue;
        const long long b = (M + 63) / 64;
        if (blocks + b >= (1ll << 30)) return TDRN_E_UNSUPPORTED;
        mp.pr[mp.n] = pr[i];
        mp.pr[mp.n].blk0 = blocks;
        blocks += (int)b;
        max_out = pr[i].n_out > max_out ? pr[i].n_out : max_out;
        ++mp.n;
    }
    if (!mp.n) return TDRN_OK;
    hipLaunchKernelGGL(offset_conv_multi_kernel, dim3((unsigned)blocks, (unsigned)((max_out + 127) / 128)), dim3(256), 0, s, mp);
    return hip_status(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// layout conversions (API surfaces are NCHW fp32)
// ---------------------------------------------------------------------------------------------
// channel c of group g = c / cpg lands at g*cpg_pad + (c - g*cpg); pad channels are zero
template <typename DT>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float *__restrict__ in, char *__restrict__ out, int B,
                                                           int C, int HW, int G, int cpg_pad)
{
    constexpr int ES = elem_traits<DT>::bytes;
    const int Cpad = G * cpg_pad, cpg = C / G;
    const long long total = (long long)B * HW * Cpad;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int cp = (int)(i % Cpad);
        const long long m = i / Cpad;
        const int b = (int)(m / HW), pix = (int)(m - (long long)b * HW);
        const int g = cp / cpg_pad, cl = cp - g * cpg_pad;
        const float v = cl < cpg ? in[((size_t)b * C + g * cpg + cl) * HW + pix] : 0.f;
        *(DT *)(out + (size_t)i * ES) = from_f32<DT>(v);
    }
}

int launch_nchw_to_nhwc_grouped(const float *in, void *out, int B, int C, int HW, int G, int cpg_pad, int dtype,
                                hipStream_t s)
{
    const long long total = (long long)B * HW * G * cpg_pad;
    if (total <= 0) return TDRN_OK;
    if (G < 1 || C % G) return TDRN_E_ARG;
    dim3 grid((unsigned)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256));
#define L(DT) hipLaunchKernelGGL((nchw_to_nhwc_kernel<DT>), grid, dim3(256), 0, s, in, (char *)out, B, C, HW, G, cpg_pad)
    if (dtype == TDRN_F32) L(float); else if (dtype == TDRN_BF16) L(bf16_t); else L(f16_t);
#undef L
    return hip_status(hipGetLastError());
}

int launch_nchw_to_nhwc(const float *in, void *out, int B, int C, int HW, int Cpad, int dtype, hipStream_t s)
{
    return launch_nchw_to_nhwc_grouped(in, out, B, C, HW, 1, Cpad, dtype, s);
}

// OIHW fp32 -> [Npad][tap][G*cpg_pad] DT (rows >= Cout and pad channels zero)
template <typename DT>
__global__ __launch_bounds__(256) void repack_oihw_kernel(const float *__restrict__ w, char *__restrict__ out, int Cout,
                                                          int Npad, int Cin, int taps, int G, int cpg_pad)
{
    constexpr int ES = elem_traits<DT>::bytes;
    const int Cpad = G * cpg_pad, cpg = Cin / G;
    const long long total = (long long)Npad * taps * Cpad;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int cp = (int)(i % Cpad);
        const long long r = i / Cpad;
        const int t = (int)(r % taps), co = (int)(r / taps);
        const int g = cp / cpg_pad, cl = cp - g * cpg_pad;
        const float v = (co < Cout && cl < cpg) ? w[((size_t)co * Cin + g * cpg + cl) * taps + t] : 0.f;
        *(DT *)(out + (size_t)i * ES) = from_f32<DT>(v);
    }
}

int launch_repack_oihw(const float *w, void *out, int Cout, int Npad, int Cin, int taps, int G, int cpg_pad, int dtype,
                       hipStream_t s)
{
    const long long total = (long long)Npad * taps * G * cpg_pad;
    if (total <= 0) return TDRN_OK;
    dim3 grid((unsigned)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256));
#define L(DT) hipLaunchKernelGGL((repack_oihw_kernel<DT>), grid, dim3(256), 0, s, w, (char *)out, Cout, Npad, Cin, taps, G, cpg_pad)
    if (dtype == TDRN_F32) L(float); else if (dtype == TDRN_BF16) L(bf16_t); else L(f16_t);
#undef L
    return hip_status(hipGetLastError());
}

__global__ __launch_bounds__(256) void nhwc_to_nchw_f32_kernel(const float *__restrict__ in, long long in_bs,
                                                               long long in_ps, float *__restrict__ out, int B, int C,
                                                               int HW)
{
    const long long total = (long long)B * C * HW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int pix = (int)(i % HW);
        const long long bc = i / HW;
        const int c = (int)(bc % C), b = (int)(bc / C);
        out[i] = in[b * in_bs + pix * in_ps + c];
    }
}

int launch_nhwc_to_nchw_f32(const float *in, long long in_bs, long long in_ps, float *out, int B, int C, int HW,
                            hipStream_t s)
{
    const long long total = (long long)B * C * HW;
    if (total <= 0) return TDRN_OK;
    dim3 grid((unsigned)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256));
    hipLaunchKernelGGL(nhwc_to_nchw_f32_kernel, grid, dim3(256), 0, s, in, in_bs, in_ps, out, B, C, HW);
    return hip_status(hipGetLastError());
}

template <typename DT>
__global__ __launch_bounds__(256) void nhwc_any_to_nchw_kernel(const char *__restrict__ in, int Cpad, float *__restrict__ out,
                                                               int B, int C, int HW)
{
    constexpr int ES = elem_traits<DT>::bytes;
    const long long total = (long long)B * C * HW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int pix = (int)(i % HW);
        const long long bc = i / HW;
        const int c = (int)(bc % C), b = (int)(bc / C);
        out[i] = to_f32<DT>(*(const DT *)(in + (((size_t)b * HW + pix) * Cpad + c) * ES));
    }
}

int launch_nhwc_any_to_nchw_f32(const void *in, int dtype, int Cpad, float *out, int B, int C, int HW, hipStream_t s)
{
    const long long total = (long long)B * C * HW;
    if (total <= 0) return TDRN_OK;
    dim3 grid((unsigned)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256));
#define L(DT) hipLaunchKernelGGL((nhwc_any_to_nchw_kernel<DT>), grid, dim3(256), 0, s, (const char *)in, Cpad, out, B, C, HW)
    if (dtype == TDRN_F32) L(float); else if (dtype == TDRN_BF16) L(bf16_t); else L(f16_t);
#undef L
    return hip_status(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// Preprocess: uint8 BGR HWC -> cv2-style bilinear resize (fixed point) -> -mean -> (RGB) -> fp32 NCHW.
// OpenCV imgproc/resize.cpp, 8-bit INTER_LINEAR: fx = (dx+0.5)*scale-0.5, 11-bit coefficients
// (saturate_cast<short>(w*2048)), horizontal sums in int, vertical
//   dst = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void cv_coef(int d, double scale, int n_src, int &s0, int &c0, int &c1)
{
    float f = (float)((d + 0.5) * scale - 0.5);
    int sidx = (int)floorf(f);
    f -= (float)sidx;
    if (sidx < 0) { f = 0.f; sidx = 0; }
    if (sidx >= n_src - 1) { f = 0.f; sidx = n_src - 1; }
    s0 = sidx;
    c0 = (int)rintf((1.f - f) * 2048.f);
    c1 = (int)rintf(f * 2048.f);
}

__global__ __launch_bounds__(256) void preprocess_kernel(const unsigned char *__restrict__ in, int B, int H0, int W0, int S,
                                                         float m0, float m1, float m2, int to_rgb, float *__restrict__ out)
{
    const long long total = (long long)B * S * S;
    const double sx = (double)W0 / S, sy = (double)H0 / S;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int dx = (int)(i % S), dy = (int)((i / S) % S), b = (int)(i / ((long long)S * S));
        int x0, a0, a1, y0, b0, b1;
        cv_coef(dx, sx, W0, x0, a0, a1);
        cv_coef(dy, sy, H0, y0, b0, b1);
        const int x1 = x0 + 1 < W0 ? x0 + 1 : x0, y1 = y0 + 1 < H0 ? y0 + 1 : y0;
        const unsigned char *r0 = in + (((size_t)b * H0 + y0) * W0) * 3, *r1 = in + (((size_t)b * H0 + y1) * W0) * 3;
        const float mean[3] = {m0, m1, m2};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int h0 = r0[x0 * 3 + c] * a0 + r0[x1 * 3 + c] * a1;     // horizontal pass, rows y0 / y1
            const int h1 = r1[x0 * 3 + c] * a0 + r1[x1 * 3 + c] * a1;
            const int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
            const float px = (float)(v < 0 ? 0 : (v > 255 ? 255 : v)) - mean[c];
            const int oc = to_rgb ? 2 - c : c;
            out[(((size_t)b * 3 + oc) * S + dy) * S + dx] = px;
        }
    }
}

int launch_preprocess(const unsigned char *in, int B, int H0, int W0, int S, const float *mean_bgr, int to_rgb, float *out,
                      hipStream_t s)
{
    const long long total = (long long)B * S * S;
    if (total <= 0) return TDRN_OK;
    dim3 grid((unsigned)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256));
    hipLaunchKernelGGL(preprocess_kernel, grid, dim3(256), 0, s, in, B, H0, W0, S, mean_bgr[0], mean_bgr[1], mean_bgr[2], to_rgb, out);
    return hip_status(hipGetLastError());
}

// The same resize with the result left as what cv2.resize returns: uint8, planar (B, 3, S, S), channel-swapped on request -- a quarter of
// the fp32 tensor; the mean is subtracted where the frame is read (conv3x3_ws.hip's producers, or u8_planes_to_f32 below).
__global__ __launch_bounds__(256) void preprocess_u8_kernel(const unsigned char *__restrict__ in, int B, int H0, int W0, int S, int to_rgb,
                                                            unsigned char *__restrict__ out)
{
    const long long total = (long long)B * S * S;
    const double sx = (double)W0 / S, sy = (double)H0 / S;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int dx = (int)(i % S), dy = (int)((i / S) % S), b = (int)(i / ((long long)S * S));
        int x0, a0, a1, y0, b0, b1;
        cv_coef(dx, sx, W0, x0, a0, a1);
        cv_coef(dy, sy, H0, y0, b0, b1);
        const int x1 = x0 + 1 < W0 ? x0 + 1 : x0, y1 = y0 + 1 < H0 ? y0 + 1 : y0;
        const unsigned char *r0 = in + (((size_t)b * H0 + y0) * W0) * 3, *r1 = in + (((size_t)b * H0 + y1) * W0) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int h0 = r0[x0 * 3 + c] * a0 + r0[x1 * 3 + c] * a1;
            const int h1 = r1[x0 * 3 + c] * a0 + r1[x1 * 3 + c] * a1;
            const int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
            const int oc = to_rgb ? 2 - c : c;
            out[(((size_t)b * 3 + oc) * S + dy) * S + dx] = (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
    }
}

int launch_preprocess_u8(const unsigned char *in, int B, int H0, int W0, int S, int to_rgb, unsigned char *out, hipStream_t s)
{
    const long long total = (long long)B * S * S;
    if (total <= 0) return TDRN_OK;
    dim3 grid((unsigned)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256));
    hipLaunchKernelGGL(preprocess_u8_kernel, grid, dim3(256), 0, s, in, B, H0, W0, S, to_rgb, out);
    return hip_status(hipGetLastError());
}

// uint8 planes (B, 3, S, S) -> fp32 (B, 3, S, S) minus the per-plane mean: the net input for the plans whose first conv is a launch of
// its own (fp32 mode, stride-2 trunks); float(u8) - mean is exact, so both routes feed the first conv the same values
__global__ __launch_bounds__(256) void u8_planes_to_f32_kernel(const unsigned char *__restrict__ in, long long plane, long long total, float m0,
                                                               float m1, float m2, float *__restrict__ out)
{
    for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i < total; i += (long long)gridDim.x * 256 * 4) {
        const int c = (int)((i / plane) % 3);
        const float m = c == 0 ? m0 : (c == 1 ? m1 : m2);
        const uchar4 v = *(const uchar4 *)(in + i);            // (plane is a multiple of 4: S is a multiple of 64)
        *(float4 *)(out + i) = make_float4((float)v.x - m, (float)v.y - m, (float)v.z - m, (float)v.w - m);
    }
}

int launch_u8_planes_to_f32(const unsigned char *in, int B, int S, const float *mean, float *out, hipStream_t s)
{
    const long long plane = (long long)S * S, total = plane * 3 * B;
    if (total <= 0) return TDRN_OK;
    if (plane % 4) return TDRN_E_UNSUPPORTED;
    const long long th = total / 4;
    dim3 grid((unsigned)((th + 255) / 256 > 8192 ? 8192 : (th + 255) / 256));
    hipLaunchKernelGGL(u8_planes_to_f32_kernel, grid, dim3(256), 0, s, in, plane, total, mean[0], mean[1], mean[2], out);
    return hip_status(hipGetLastError());
}

int launch_fill_zero(void *p, size_t bytes, hipStream_t s)
{
    return hip_status(hipMemsetAsync(p, 0, bytes, s));
}

int allow_big_lds(const void *kernel)
{
    constexpr int kSlots = 32;
    static std::atomic<const void *> fn[kSlots];
    static std::atomic<unsigned long long> done[kSlots];     // bit d: set on device d
    int dev = 0;
    TDRN_HIP_TRY(hipGetDevice(&dev));
    int slot = -1;
    for (int i = 0; i < kSlots && slot < 0; ++i) {
        const void *cur = fn[i].load(std::memory_order_acquire);
        if (cur == nullptr) {
            const void *expect = nullptr;
            if (fn[i].compare_exchange_strong(expect, kernel, std::memory_order_acq_rel)) cur = kernel;
            else cur = expect;
        }
        if (cur == kernel) slot = i;
    }
    const bool memo = slot >= 0 && dev >= 0 && dev < 64;
    if (memo && ((done[slot].load(std::memory_order_acquire) >> dev) & 1ull)) return TDRN_OK;
    TDRN_HIP_TRY(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    if (memo) done[slot].fetch_or(1ull << dev, std::memory_order_release);
    return TDRN_OK;
}

}  // namespace tdrn
