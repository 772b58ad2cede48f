// deform.hip -- deformable convolution v1 forward as ONE fused kernel: bilinear gather of NHWC
// feature rows straight into the LDS operand tile, then MFMA.  Replaces the reference's
// per-image deformable_im2col kernel + cuBLAS SGEMM with its column buffer round trip through
// global memory (utils/deformconv/deform_conv_cuda_kernel.cu:156-238, deform_conv_cuda.c:157-193).
//
// Sampling semantics follow deform_conv_cuda_kernel.cu:15-51 and :189-203 exactly, including the
// asymmetric border rule: any negative coordinate or coordinate >= H (W) gives 0; coordinates in
// (H-1, H) clamp to row H-1 with fraction 0 (no zero-padding decay on the bottom/right edge).
// Offsets and bilinear weights are fp32 in every mode.
//
// Up to two branches (the 3x3 and the 5x5 "multihead" heads, model/dualrefinedet_vggbn.py:181-187)
// and several heads stacked along Cout (loc = 12, conf = 63 channels) share one launch: the sum
// l(ob,f) + l2(ob,f2) is just a longer K loop, and loc/conf are column segments of the same GEMM.
#include <algorithm>

#include "kernels.h"

namespace tdrn {

struct DeformBranchP {
    const float *off;
    const char *w;
    int off_stride, kh, kw, G;
    int pad_h, pad_w, stride_h, stride_w, dil_h, dil_w;      // per axis, like deform_conv_cuda.c:98-103
    int off_rows, off_row0;                                  // DeformBranch: offsets of off_rows key-frame pixel rows, broadcast
};
struct DeformParams {
    const char *in, *zero;
    DeformBranchP br[2];
    int n_branches;
    int M, H, W, Cin, Ho, Wo, Cout, Npad, split;
    float *out0, *out1;
    long long o0_bs, o0_ps, o1_bs, o1_ps;
    int accumulate;   // 1: outputs are pre-zeroed and this problem atomically adds its partial result
    int g_begin, g_end;   // deformable groups [g_begin, g_end) of every tap (all of them unless the launch splits a problem by groups)
};
// several independent problems (the four pyramid levels) in ONE launch: their K loops are
// latency-bound per workgroup, so running them side by side costs the time of the longest.
constexpr int kMaxDeformProblems = 8;
struct DeformMulti {
    DeformParams p[kMaxDeformProblems];
    int block_start[kMaxDeformProblems + 1];
    int n;
    int xcd_order;    // 1: the grid is 8 * ceil(tiles / 8) workgroups and workgroup b works on tile (b % 8) * ceil(tiles / 8) + b / 8 (see the kernel)
};

template <typename DT> struct MmaD;
template <> struct MmaD<bf16_t> {
    __device__ static __forceinline__ void run(const u32x4 &a, const u32x4 &b, f32x16 &c)
    { c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(i16x8, a), __builtin_bit_cast(i16x8, b), c, 0, 0, 0); }
};
template <> struct MmaD<f16_t> {
    __device__ static __forceinline__ void run(const u32x4 &a, const u32x4 &b, f32x16 &c)
    { c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0); }
};
template <> struct MmaD<float> {
    __device__ static __forceinline__ void run(const u32x4 &a, const u32x4 &b, f32x16 &c)
    {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a[j]), __uint_as_float(b[j]), c, 0, 0, 0);
    }
};

// 128 pixels x (NTL*32) couts per workgroup; wave w owns pixels [32w, 32w+32) and all cout tiles.
template <typename DT, int NTL>
__global__ __launch_bounds__(256) void deform_gemm_kernel(const DeformMulti mp)
{
    // XCD-aware tile order: workgroup b runs on XCD b % 8; with `xcd_order` every XCD works on a CONTIGUOUS run of pixel tiles, so the
    // rows that neighbouring tiles gather (the taps reach +-1..2 rows plus the offsets) come through one L2 instead of eight
    int bid = (int)blockIdx.x;
    if (mp.xcd_order) {
        const int total = mp.block_start[mp.n], per = (total + 7) >> 3;
        bid = (bid & 7) * per + (bid >> 3);
        if (bid >= total) return;                       // (whole workgroup; the grid is rounded up to 8 * per)
    }
    int prob = 0;
#pragma unroll
    for (int i = 1; i < kMaxDeformProblems; ++i)
        if (i < mp.n && bid >= mp.block_start[i]) prob = i;
    const DeformParams &p = mp.p[prob];
    constexpr int BM = 128, BN = NTL * 32, NT = 256, RPP = 32, PA = BM / RPP;
    constexpr int ES = elem_traits<DT>::bytes;
    constexpr int P16 = elem_traits<DT>::per16;
    constexpr int CK = 128 / ES;
    constexpr int STAGE = (BM + BN) * 128;
    constexpr int CS = BN * 4 + 16;
    constexpr int LDS = (2 * STAGE > BM * CS) ? 2 * STAGE : BM * CS;
    __shared__ __attribute__((aligned(16))) char smem[LDS];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int lrow = t >> 3, pc = t & 7;
    const int lc16 = (pc ^ ((lrow >> 1) & 7)) << 4;
    const int m0 = (bid - mp.block_start[prob]) * BM;
    const int HoWo = p.Ho * p.Wo;

    int ho_[PA], wo_[PA], ibase[PA];
    long long mrow[PA];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int m = m0 + i * RPP + lrow;
        if (m < p.M) {
            const int b = m / HoWo, rem = m - b * HoWo;
            ho_[i] = rem / p.Wo;
            wo_[i] = rem - ho_[i] * p.Wo;
            ibase[i] = b * p.H * p.W * p.Cin;
            // (the row of the offset maps this output pixel reads: its own, or under the key-frame broadcast of DeformBranch the
            // key frame's -- both branches of a problem share the mapping)
            mrow[i] = p.br[0].off_rows ? (m + p.br[0].off_row0) % p.br[0].off_rows : m;
        } else {
            ho_[i] = 0; wo_[i] = 0; ibase[i] = 0; mrow[i] = -1;
        }
    }

    f32x16 acc[NTL];
#pragma unroll
    for (int ci = 0; ci < NTL; ++ci)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[ci][e] = 0.f;

    // K-loop cursor: branch, tap (ti,tj), group g, channel chunk cc inside the group
    int br = 0, ti = 0, tj = 0, g = p.g_begin, cc = 0, kofs = p.g_begin * (p.Cin / p.br[0].G);
    int coff[PA][4];      // element offsets of the four corners (relative to `in`)
    float cw[PA][4];      // bilinear weights (0 when the tap is rejected)

    // offsets of the NEXT (tap, group) are fetched one tap ahead: a tap lasts cpg/CK K-steps, so they have landed
    // long before tap_params() needs them and the K loop never waits on an offset load
    float noff[PA][2];
    auto fetch_offsets = [&](int br_, int ti_, int tj_, int g_) {
        const DeformBranchP &B = p.br[br_];
        const int tapidx = ti_ * B.kw + tj_;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            noff[i][0] = 0.f;
            noff[i][1] = 0.f;
            if (mrow[i] >= 0) {
                const float *op = B.off + mrow[i] * B.off_stride + (size_t)g_ * 2 * B.kh * B.kw + 2 * tapidx;
                noff[i][0] = op[0];
                noff[i][1] = op[1];
            }
        }
    };
    auto tap_params = [&]() {
        const DeformBranchP &B = p.br[br];
        const int cpg = p.Cin / B.G;
        float offh[PA], offw[PA];
#pragma unroll
        for (int i = 0; i < PA; ++i) { offh[i] = noff[i][0]; offw[i] = noff[i][1]; }
        {   // the tap after this one (same stepping as advance())
            int nb = br, ni = ti, nj = tj, ng = g + 1;
            if (ng == p.g_end) {
                ng = p.g_begin;
                if (++nj == B.kw) { nj = 0; ++ni; }
                if (ni == B.kh) { ni = 0; ++nb; }
            }
            if (nb < p.n_branches) fetch_offsets(nb, ni, nj, ng);
        }
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            float w1 = 0.f, w2 = 0.f, w3 = 0.f, w4 = 0.f;
            int o1 = ibase[i], o2 = ibase[i], o3 = ibase[i], o4 = ibase[i];
            if (mrow[i] >= 0) {
                const float offset_h = offh[i], offset_w = offw[i];
                const int h_in = ho_[i] * B.stride_h - B.pad_h, w_in = wo_[i] * B.stride_w - B.pad_w;
                const float h_im = (float)(h_in + ti * B.dil_h) + offset_h;
                const float w_im = (float)(w_in + tj * B.dil_w) + offset_w;
                if (h_im >= 0.f && w_im >= 0.f && h_im < (float)p.H && w_im < (float)p.W) {
                    float h = (float)(ti * B.dil_h) + offset_h;     // map_h, relative to h_in
                    float w = (float)(tj * B.dil_w) + offset_w;
                    const int height = p.H - h_in, width = p.W - w_in;
                    int h_low = (int)floorf(h), w_low = (int)floorf(w), h_high, w_high;
                    if (h_low >= height - 1) { h_high = h_low = height - 1; h = (float)h_low; } else { h_high = h_low + 1; }
                    if (w_low >= width - 1) { w_high = w_low = width - 1; w = (float)w_low; } else { w_high = w_low + 1; }
                    const float lh = h - (float)h_low, lw = w - (float)w_low;
                    const float hh = 1.f - lh, hw = 1.f - lw;
                    w1 = hh * hw; w2 = hh * lw; w3 = lh * hw; w4 = lh * lw;
                    // absolute coordinates (clamped only for memory safety; a no-op whenever the
                    // reference itself stays in bounds)
                    const int r0 = min(max(h_in + h_low, 0), p.H - 1), r1 = min(max(h_in + h_high, 0), p.H - 1);
                    const int q0 = min(max(w_in + w_low, 0), p.W - 1), q1 = min(max(w_in + w_high, 0), p.W - 1);
                    const int gb = ibase[i] + g * cpg;
                    o1 = gb + (r0 * p.W + q0) * p.Cin;
                    o2 = gb + (r0 * p.W + q1) * p.Cin;
                    o3 = gb + (r1 * p.W + q0) * p.Cin;
                    o4 = gb + (r1 * p.W + q1) * p.Cin;
                }
            }
            coff[i][0] = o1; coff[i][1] = o2; coff[i][2] = o3; coff[i][3] = o4;
            cw[i][0] = w1; cw[i][1] = w2; cw[i][2] = w3; cw[i][3] = w4;
        }
    };

    // one K-step of operands: weights by LDS-DMA; the four bilinear corners of each of this
    // thread's pixel rows go to registers first (issue) and are blended into LDS one MFMA phase
    // later (finish), so their L2 latency hides under the matrix work of the previous K-step.
    u32x4 raw[PA][4];
    auto issue = [&](int buf) {
        char *sb = smem + buf * STAGE;
        const DeformBranchP &B = p.br[br];
        const int Ktot = B.kh * B.kw * p.Cin;
#pragma unroll
        for (int i = 0; i < NTL; ++i) {
            const char *src = B.w + ((size_t)(i * RPP + lrow) * Ktot + kofs) * ES + lc16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(sb + (i * RPP + wave * 8) * 128), 16, 0, 0);
        }
        const int coffs = cc * CK * ES + lc16;
#pragma unroll
        for (int i = 0; i < PA; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k)      // 32-bit byte offset from the scalar base (fill_params bounds the tensor below 4 GiB)
                raw[i][k] = *(const u32x4 *)(p.in + (unsigned)((unsigned)coff[i][k] * (unsigned)ES + (unsigned)coffs));
    };
    float fw[PA][4];      // weights that belong to the registers in flight
    auto latch_weights = [&]() {
#pragma unroll
        for (int i = 0; i < PA; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) fw[i][k] = cw[i][k];
    };
    auto finish = [&](int buf) {
        char *sb = smem + buf * STAGE;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            float v[4][P16], o[P16];
#pragma unroll
            for (int k = 0; k < 4; ++k) unpack16<DT>(raw[i][k], v[k]);
#pragma unroll
            for (int j = 0; j < P16; ++j)
                o[j] = fmaf(fw[i][3], v[3][j], fmaf(fw[i][2], v[2][j], fmaf(fw[i][1], v[1][j], fw[i][0] * v[0][j])));
            *(u32x4 *)(sb + BN * 128 + (i * RPP + lrow) * 128 + pc * 16) = pack16<DT>(o);
        }
    };

    // returns false when the K loop is exhausted; recomputes tap params when (tap, group) changes
    auto advance = [&]() -> bool {
        const DeformBranchP &B = p.br[br];
        const int cpg = p.Cin / B.G;
        kofs += CK;
        if (++cc < cpg / CK) return true;
        cc = 0;
        if (++g == p.g_end) {
            g = p.g_begin;
            if (++tj == B.kw) { tj = 0; ++ti; }
            if (ti == B.kh) {
                ti = 0;
                if (++br == p.n_branches) return false;
            }
            // K offset of (tap, first group of my range) in the weight row [taps][Cin] (the groups outside the range are skipped)
            kofs = (ti * p.br[br].kw + tj) * p.Cin + p.g_begin * (p.Cin / p.br[br].G);
        }
        tap_params();
        return true;
    };

    const int r32 = lane & 31, hh = lane >> 5;
    const int sw = (r32 >> 1) & 7;
    auto compute = [&](int buf) {
        const char *wsb = smem + buf * STAGE;
        const char *psb = wsb + BN * 128;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int ch = ((2 * kk + hh) ^ sw) << 4;
            const u32x4 pf = *(const u32x4 *)(psb + (wave * 32 + r32) * 128 + ch);
#pragma unroll
            for (int ci = 0; ci < NTL; ++ci) {
                const u32x4 wf = *(const u32x4 *)(wsb + (ci * 32 + r32) * 128 + ch);
                MmaD<DT>::run(wf, pf, acc[ci]);
            }
        }
    };

    fetch_offsets(0, 0, 0, p.g_begin);
    tap_params();
    issue(0);
    latch_weights();
    finish(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    bool more = true;
    while (more) {
        more = advance();
        if (more) {
            issue(cur ^ 1);
            latch_weights();
        }
        compute(cur);
        if (more) finish(cur ^ 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue: transpose through LDS, then split the channel range into the two outputs ----
#pragma unroll
    for (int ci = 0; ci < NTL; ++ci)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = acc[ci][4 * q + j];
            *(f32x4 *)(smem + (wave * 32 + r32) * CS + (ci * 32 + 8 * q + 4 * hh) * 4) = v;
        }
    __syncthreads();
    for (int idx = t; idx < BM * p.Cout; idx += NT) {
        const int row = idx / p.Cout, c = idx - row * p.Cout;
        const int m = m0 + row;
        if (m >= p.M) break;
        const int b = m / HoWo, pix = m - b * HoWo;
        const float v = *(const float *)(smem + row * CS + c * 4);
        float *dst = c < p.split ? p.out0 + b * p.o0_bs + pix * p.o0_ps + c : p.out1 + b * p.o1_bs + pix * p.o1_ps + (c - p.split);
        if (p.accumulate) atomicAdd(dst, v);      // exactly two addends land on a zeroed value: order-independent
        else *dst = v;
    }
}

int deform_n_pad(int cout) { return (int)align_up((size_t)cout, 32); }

template <typename DT> static int launch_deform_dt(const DeformMulti &mp, int ntl, hipStream_t s)
{
    dim3 grid((unsigned)(mp.xcd_order ? 8 * ((mp.block_start[mp.n] + 7) / 8) : mp.block_start[mp.n]));
    switch (ntl) {
        case 1: hipLaunchKernelGGL((deform_gemm_kernel<DT, 1>), grid, dim3(256), 0, s, mp); break;
        case 2: hipLaunchKernelGGL((deform_gemm_kernel<DT, 2>), grid, dim3(256), 0, s, mp); break;
        case 3: hipLaunchKernelGGL((deform_gemm_kernel<DT, 3>), grid, dim3(256), 0, s, mp); break;
        case 4: hipLaunchKernelGGL((deform_gemm_kernel<DT, 4>), grid, dim3(256), 0, s, mp); break;
        default: return TDRN_E_UNSUPPORTED;
    }
    return hip_status(hipGetLastError());
}

static int fill_params(const DeformArgs &a, DeformParams &p)
{
    if (!a.in || !a.out0 || a.n_branches < 1 || a.n_branches > 2) return TDRN_E_ARG;
    const int es = dtype_bytes(a.dtype), ck = 128 / es;
    if (a.Npad % 32 || a.Npad > 128 || a.Cout > a.Npad || a.Cout < 1) return TDRN_E_UNSUPPORTED;
    if (a.split < a.Cout && !a.out1) return TDRN_E_ARG;
    if ((long long)a.B * a.H * a.W * a.Cin * es >= (1ll << 32)) return TDRN_E_UNSUPPORTED;     // 32-bit byte offsets in the gather
    p.in = (const char *)a.in;
    p.zero = (const char *)a.zero_page;
    p.n_branches = a.n_branches;
    for (int i = 0; i < a.n_branches; ++i) {
        const DeformBranch &b = a.br[i];
        if (!b.off || !b.w || b.G < 1 || a.Cin % b.G || (a.Cin / b.G) % ck) return TDRN_E_UNSUPPORTED;
        if (b.kh < 1 || b.kw < 1 || b.stride < 1 || b.dil < 1) return TDRN_E_SHAPE;
        p.br[i] = DeformBranchP{b.off, (const char *)b.w, b.off_stride, b.kh, b.kw, b.G,
                                b.pad, b.pad_w < 0 ? b.pad : b.pad_w, b.stride, b.stride_w < 1 ? b.stride : b.stride_w,
                                b.dil, b.dil_w < 1 ? b.dil : b.dil_w, b.off_rows, b.off_row0};
        if (b.off_rows < 0 || b.off_row0 < 0) return TDRN_E_ARG;
    }
    if (a.n_branches == 1) p.br[1] = p.br[0];
    p.M = a.B * a.Ho * a.Wo;
    p.H = a.H; p.W = a.W; p.Cin = a.Cin; p.Ho = a.Ho; p.Wo = a.Wo; p.Cout = a.Cout; p.Npad = a.Npad;
    p.split = a.split > a.Cout ? a.Cout : a.split;
    p.out0 = a.out0; p.out1 = a.out1;
    p.o0_bs = a.o0_bs; p.o0_ps = a.o0_ps; p.o1_bs = a.o1_bs; p.o1_ps = a.o1_ps;
    return TDRN_OK;
}

// all problems must share dtype and Npad (they do: the pyramid levels of one model).
// split_branches: a two-branch problem (3x3 + 5x5 heads) becomes two single-branch work items that
// atomically add into PRE-ZEROED outputs.  All workgroups have the same K-loop length per branch and the
// kernel is latency-bound per workgroup, so 532 two-branch workgroups on 512 resident slots cost two
// full rounds; as 532 long (25-tap) + 532 short (9-tap) items dispatched longest-first they pack tightly.
int launch_deform_multi(const DeformArgs *args, int n, hipStream_t s, int split_branches)
{
    if (!args || n < 1 || n > 4) return TDRN_E_ARG;
    DeformMulti mp;
    mp.n = 0;
    mp.block_start[0] = 0;
    // split_branches = 2: a ONE-branch problem with an even number >= 2 of deformable groups (the TRN temporal heads, df_group = 8)
    // becomes two work items, groups [0, G/2) and [G/2, G), that atomically add into PRE-ZEROED outputs: its K loop (one step per
    // (tap, group): 72 dependent steps with 8 groups, latency-bound per workgroup, ~100 workgroups at config #5's batch) halves and
    // the workgroup count doubles.  Exactly two addends per output element: order-independent, deterministic.
    for (int pass = 0; pass < 2; ++pass) {           // pass 0: the longer branch / first half of every problem, pass 1: the rest
        for (int i = 0; i < n; ++i) {
            if (args[i].dtype != args[0].dtype || args[i].Npad != args[0].Npad) return TDRN_E_UNSUPPORTED;
            DeformArgs a = args[i];
            const bool two = a.n_branches == 2 && split_branches == 1;
            const bool halves = split_branches == 2 && a.n_branches == 1 && a.br[0].G >= 2 && a.br[0].G % 2 == 0;
            if (!two && !halves && pass == 1) continue;
            if (two) {
                const int t0 = a.br[0].kh * a.br[0].kw, t1 = a.br[1].kh * a.br[1].kw;
                const int longer = t1 > t0 ? 1 : 0;
                a.br[0] = args[i].br[pass == 0 ? longer : 1 - longer];
                a.n_branches = 1;
            }
            DeformParams p;
            TDRN_TRY(fill_params(a, p));
            p.accumulate = (two || halves) ? 1 : 0;
            p.g_begin = 0;
            p.g_end = a.br[0].G;
            if (halves) {
                p.g_begin = pass == 0 ? 0 : a.br[0].G / 2;
                p.g_end = pass == 0 ? a.br[0].G / 2 : a.br[0].G;
            }
            if (p.M <= 0) continue;
            if (mp.n >= kMaxDeformProblems) return TDRN_E_UNSUPPORTED;
            mp.p[mp.n] = p;
            mp.block_start[mp.n + 1] = mp.block_start[mp.n] + cdiv(p.M, 128);
            ++mp.n;
        }
    }
    if (mp.n == 0) return TDRN_OK;
    for (int i = mp.n; i < kMaxDeformProblems; ++i) { mp.p[i] = mp.p[0]; mp.block_start[i + 1] = mp.block_start[mp.n]; }
    static int xo = -1;
    if (xo < 0) { const char *e = getenv("TDRN_DEFORM_XCD"); xo = e ? atoi(e) : 1; }
    mp.xcd_order = xo;
    switch (args[0].dtype) {
        case TDRN_F32: return launch_deform_dt<float>(mp, args[0].Npad / 32, s);
        case TDRN_BF16: return launch_deform_dt<bf16_t>(mp, args[0].Npad / 32, s);
        case TDRN_F16: return launch_deform_dt<f16_t>(mp, args[0].Npad / 32, s);
    }
    return TDRN_E_ARG;
}

int launch_deform(const DeformArgs &a, hipStream_t s) { return launch_deform_multi(&a, 1, s, 0); }

// =============================================================================================
// Transform, then sample (16-bit plans, one deformable group, Cout <= 80 -- the ODM heads).
//
// The op is linear in the sampled columns and has FEWER outputs (75: 12 loc + 63 conf) than inputs (256):
//     out[p][co] = sum_tap sum_c W[co][tap][c] * ( sum_k w_k(p,tap) * X[corner_k(p,tap)][c] )
//                = sum_tap sum_k w_k(p,tap) * Y[corner_k(p,tap)][tap][co],      Y[q][tap][co] = sum_c W[co][tap][c] * X[q][c]
// Y is ONE plain 1x1 GEMM per pyramid level (conv_igemm.hip: 256 -> 34 taps x 80 columns, the same FLOPs as the fused
// kernel's, at dense-GEMM rate, fp32 accumulation, rounded once to the net dtype); the bilinear blend then gathers 160-byte
// rows instead of 512-byte ones and blends 75 values per corner instead of 256 -- 3.4x less of exactly the two things that
// bound deform_gemm_kernel above (gathered bytes, vector-ALU blend: profiles/r02_final).  The sampling rule -- which taps
// are rejected, the (H-1, H) clamp, fp32 offsets and weights -- is the same code as tap_params() there
// (deform_conv_cuda_kernel.cu:15-51, 189-203); no atomics, no zero-fill: one wave owns one output pixel and both branches.
// Where the rounding sits differs from the fused kernel (Y is rounded to 16 bits, there the blended columns are); the fp32
// mode and the fp32 C-ABI entry keep the fused kernel.
// =============================================================================================
struct SampleBranchP {
    const float *off;              // offsets of this branch: [pixel][off_stride], 2 floats per tap (dh, dw)
    int off_stride, kh, kw, pad_h, pad_w, dil_h, dil_w, col0;   // col0: first Y column block (tap index) of the branch
    int off_rows, off_row0;        // DeformBranch: key-frame broadcast of the offsets (0: one row per output pixel)
};
struct SampleParams {
    const char *y;                 // [B*H*W][ycs] DT: tap t's 80 columns at t*80
    SampleBranchP br[2];
    int n_branches, n_taps;
    int M, H, W, ycs, Cout, split;
    int tap_major;                 // y is [tap][B*H*W][80] instead
    float *out0, *out1;
    long long o0_bs, o0_ps, o1_bs, o1_ps;
};
struct SampleMulti {
    SampleParams p[4];
    int block_start[5];
    int n;
    int xcd_remap;
};
constexpr int kSampleCols = 80;    // Y columns per tap (75 used)

template <typename DT>
__global__ __launch_bounds__(256) void deform_sample_kernel(const SampleMulti mp)
{
    // Workgroups are dealt round-robin over the 8 XCDs: give every XCD a contiguous run of the launch's blocks (= of output pixels,
    // 4 per block), so that the Y rows shared by neighbouring pixels -- the x+1 / y+1 corners, the overlapping taps -- are fetched
    // into ONE L2 instead of up to eight.
    const int nblk = (int)gridDim.x, xq = nblk >> 3, xr = nblk & 7, xcd = (int)blockIdx.x & 7;
    const int blk = mp.xcd_remap ? (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
    int prob = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i)
        if (i < mp.n && blk >= mp.block_start[i]) prob = i;
    const SampleParams &p = mp.p[prob];
    constexpr int MAXR = 34 * 4;                        // corner rows of one output pixel (9 + 25 taps)
    // per wave and corner row: {byte offset of the row in Y (incl. the tap's columns), bilinear weight (0: tap rejected)} -- one 8-byte LDS
    // read per row (round 5; two arrays before), padded with weight-0 copies of the last row up to whole batches of 36 rows, so that the
    // gather loop below is branch-free
    __shared__ uint2 s_ow[4][MAXR + 8];
    __shared__ float s_red[4][6][kSampleCols];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = (blk - mp.block_start[prob]) * 4 + wave;                 // one wave = one output pixel (stride 1: Ho = H)
    if (m >= p.M) return;
    const int HW = p.H * p.W;
    const int b = m / HW, rem = m - b * HW, ho = rem / p.W, wo = rem - ho * p.W;
    const int nrows = p.n_taps * 4;
    if (lane < p.n_taps) {
        const int brn = (p.n_branches == 2 && lane >= p.br[0].kh * p.br[0].kw) ? 1 : 0;
        const SampleBranchP &B = p.br[brn];
        const int tl = lane - (brn ? p.br[0].kh * p.br[0].kw : 0);
        const int ti = tl / B.kw, tj = tl - ti * B.kw;
        const int orow = B.off_rows ? (m + B.off_row0) % B.off_rows : m;
        const float *op = B.off + (size_t)orow * B.off_stride + 2 * tl;
        const float offset_h = op[0], offset_w = op[1];
        float w1 = 0.f, w2 = 0.f, w3 = 0.f, w4 = 0.f;
        int q1 = 0, q2 = 0, q3 = 0, q4 = 0;
        const int h_in = ho - B.pad_h, w_in = wo - B.pad_w;
        const float h_im = (float)(h_in + ti * B.dil_h) + offset_h;
        const float w_im = (float)(w_in + tj * B.dil_w) + offset_w;
        if (h_im >= 0.f && w_im >= 0.f && h_im < (float)p.H && w_im < (float)p.W) {
            float h = (float)(ti * B.dil_h) + offset_h;     // map_h, relative to h_in
            float w = (float)(tj * B.dil_w) + offset_w;
            const int height = p.H - h_in, width = p.W - w_in;
            int h_low = (int)floorf(h), w_low = (int)floorf(w), h_high, w_high;
            if (h_low >= height - 1) { h_high = h_low = height - 1; h = (float)h_low; } else { h_high = h_low + 1; }
            if (w_low >= width - 1) { w_high = w_low = width - 1; w = (float)w_low; } else { w_high = w_low + 1; }
            const float lh = h - (float)h_low, lw = w - (float)w_low;
            const float hh = 1.f - lh, hw = 1.f - lw;
            w1 = hh * hw; w2 = hh * lw; w3 = lh * hw; w4 = lh * lw;
            const int r0 = min(max(h_in + h_low, 0), p.H - 1), r1 = min(max(h_in + h_high, 0), p.H - 1);
            const int c0 = min(max(w_in + w_low, 0), p.W - 1), c1 = min(max(w_in + w_high, 0), p.W - 1);
            q1 = r0 * p.W + c0; q2 = r0 * p.W + c1; q3 = r1 * p.W + c0; q4 = r1 * p.W + c1;
        }
        const unsigned colb = p.tap_major ? (unsigned)(B.col0 + tl) * (unsigned)p.M * (unsigned)(kSampleCols * 2) : (unsigned)(deform_y_col(B.col0 + tl) * 2);
        const unsigned rowb = p.tap_major ? (unsigned)(kSampleCols * 2) : (unsigned)(p.ycs * 2);
        const unsigned img = (unsigned)(b * HW);
        s_ow[wave][4 * lane + 0] = make_uint2((img + q1) * rowb + colb, __float_as_uint(w1));
        s_ow[wave][4 * lane + 1] = make_uint2((img + q2) * rowb + colb, __float_as_uint(w2));
        s_ow[wave][4 * lane + 2] = make_uint2((img + q3) * rowb + colb, __float_as_uint(w3));
        s_ow[wave][4 * lane + 3] = make_uint2((img + q4) * rowb + colb, __float_as_uint(w4));
        if (lane == p.n_taps - 1)                       // rows past the end re-read the last row with weight 0 (as the loop did before)
            for (int r = nrows; r < 36 * ((nrows + 35) / 36); ++r) s_ow[wave][r] = make_uint2((img + q4) * rowb + colb, 0u);
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    // lanes 10j + c (j = 0..5, c = 0..9): corner row 6i + j, 16-byte column chunk c; lanes 60..63 idle
    const int j = lane / 10, c = lane - 10 * j;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (lane < 60) {
        // branch-free batches of six independent 16-byte loads (the table is padded to whole batches); Y's base in scalar registers,
        // 32-bit offsets (Y < 4 GiB: ygemm_fill)
        const int nb = (nrows + 35) / 36;
        const __attribute__((address_space(1))) char *yb = (const __attribute__((address_space(1))) char *)p.y;
        for (int ib = 0; ib < nb; ++ib) {
            u32x4 raw[6];
            float wgt[6];
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                const uint2 ow = s_ow[wave][6 * (6 * ib + u) + j];
                wgt[u] = __uint_as_float(ow.y);
                raw[u] = *(const __attribute__((address_space(1))) u32x4 *)(yb + (ow.x + (unsigned)(c * 16)));
            }
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                float v[8];
                unpack16<DT>(raw[u], v);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] = fmaf(wgt[u], v[e], acc[e]);
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) s_red[wave][j][c * 8 + e] = acc[e];
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    // columns 0..79: a fixed-order sum over the six row groups (deterministic)
    for (int co = lane; co < p.Cout; co += 64) {
        float v = s_red[wave][0][co];
#pragma unroll
        for (int jj = 1; jj < 6; ++jj) v += s_red[wave][jj][co];
        float *dst = co < p.split ? p.out0 + b * p.o0_bs + rem * p.o0_ps + co : p.out1 + b * p.o1_bs + rem * p.o1_ps + (co - p.split);
        *dst = v;
    }
}

// ---------------------------------------------------------------------------------------------
// The transform: Y[pixel][col] = sum_c X[pixel][c] * Wt[col][c] with K = Cin = 256 and N = taps x 80 columns (2816).
// A short-K GEMM: conv_igemm.hip runs it at 0.4 PFLOP/s (four K steps per tile: prologue, epilogue and the 288-MB result
// dominate; one 256x128 tile per workgroup, nothing overlaps).  Here the WEIGHTS LIVE IN REGISTERS: a wave owns 32 columns,
// i.e. 32 x 256 weights = 16 MFMA A-fragments = 64 registers, loaded once; the workgroup (8 waves = 256 columns) then
// streams 32-pixel activation tiles (16 KiB, LDS-DMA, double-buffered) past them: per tile and wave 16 ds_read_b128 +
// 16 MFMAs and 4 eight-byte stores per lane.  LDS-DMA volume 16 B/clk per CU, no weight traffic at all after the prologue;
// two workgroups per CU cover each other's barriers.  grid = (N / 256 column groups) x (pixel partitions).
// ---------------------------------------------------------------------------------------------
struct YGemmParams {
    const char *x;     // [M][256] DT
    const char *w;     // [N][256] DT
    char *y;           // [M][ycs] DT, or tap-major [taps][M][80] (taps > 0): the layout the sampling launch gathers from
    int M, N, ycs, parts, tiles_per_part, taps;
};

// CT = column tiles (of 32) per wave.  CT = 1: eight waves x 32 columns (round 3, the default).  CT = 2 (round 4, TDRN_YGEMM_CT=2): FOUR waves x 64
// columns -- 128 weight registers per wave, two workgroups of four waves per CU: every wave still reads the whole 16-KiB pixel tile
// from LDS per tile, but there are half as many waves doing it for the same 256 columns, so the LDS reads per output halve (CT = 1:
// 128 KiB of LDS reads per tile = 1024 cycles beside 1024 cycles of MFMA per SIMD: the two pipes were co-critical).  Same K order
// per output element: bit-identical.
// up to four independent problems (the pyramid levels) in ONE launch: the transforms of the small levels fill the tail of the
// large one, and three kernel boundaries (~7 us each in the step, round 4 trace) disappear from the critical path
struct YGemmMulti {
    YGemmParams p[4];
    int block_start[5];            // workgroups of problem i: [block_start[i], block_start[i+1]); inside: cg = local % cgs, part = local / cgs
    int n;
#ifdef TDRN_YG_STAMP
    unsigned *stamps;              // diagnostics build only (make EXTRA=-DTDRN_YG_STAMP): [workgroup][wave][6] cycle sums / tile count
#endif
};
#ifdef TDRN_YG_STAMP
#define YG_STAMP_DECL unsigned long long st_t = __builtin_amdgcn_s_memtime(); unsigned st_acc[6] = {0, 0, 0, 0, 0, 0};
#define YG_STAMP(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[k] += (unsigned)(t_ - st_t); st_t = t_; } while (0)
#define YG_STAMP_FLUSH do { st_acc[5] = (unsigned)nt; if (lane == 0) for (int k_ = 0; k_ < 6; ++k_) mp.stamps[((size_t)blockIdx.x * 8 + wave) * 6 + k_] = st_acc[k_]; } while (0)
#else
#define YG_STAMP_DECL
#define YG_STAMP(k) do { } while (0)
#define YG_STAMP_FLUSH do { } while (0)
#endif

template <typename DT, int CT>
__global__ __launch_bounds__(CT == 1 ? 512 : 256, CT == 1 ? 4 : 2) void ygemm_k256_kernel(const YGemmMulti mp)
{
    int prob = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i)
        if (i < mp.n && (int)blockIdx.x >= mp.block_start[i]) prob = i;
    const YGemmParams &p = mp.p[prob];
    const int local_blk = (int)blockIdx.x - mp.block_start[prob];
    const int ncg = p.N / 256;
    constexpr int NW = 8 / CT;                           // waves per workgroup
    constexpr int NT = NW * 64;
    constexpr int TP = 32;                               // pixels per tile
    constexpr int TBYTES = TP * 512;                     // 16 KiB
    constexpr int SROW = 512 + 16;                       // staging row: 256 columns + a 16-byte pad (bank spread)
    constexpr int PPW = 16 / NW;                         // LDS-DMA pieces per wave and tile
    __shared__ __attribute__((aligned(16))) char smem[2 * TBYTES];
    __shared__ __attribute__((aligned(16))) char sst[TP * SROW];     // the output tile [pixel][256 columns]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r32 = lane & 31, hh = lane >> 5;
    const int part = local_blk / ncg, cg = local_blk - part * ncg;
    const int col0 = cg * 256 + wave * (32 * CT);
    // my 32 CT columns x 256 channels: fragment [ct][kk] = channels [16kk + 8hh, +8) of column col0 + 32 ct + r32
    u32x4 wf[CT][16];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const char *wr = p.w + ((size_t)(col0 + 32 * ct + r32) * 256 + 8 * hh) * 2;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) wf[ct][kk] = *(const u32x4 *)(wr + kk * 32);
    }
    const int t0 = part * p.tiles_per_part;
    int nt = (p.M + TP - 1) / TP - t0;
    nt = nt < p.tiles_per_part ? nt : p.tiles_per_part;
    if (nt <= 0) return;
    YG_STAMP_DECL

    // staging: a tile = 32 rows x 512 B = 16 pieces of 1 KiB (2 rows each); wave w issues pieces PPW w .. PPW w + PPW - 1.
    // LDS image linear; the 16-byte chunk c of row r is stored at chunk position c ^ (r & 31) (swizzle on the SOURCE address)
    // The LDS-DMA goes out as inline asm (round 5): behind the builtin hipcc put an `s_waitcnt vmcnt(0)` in front of the first LDS read
    // of every tile -- it cannot tell the DMA's LDS writes from the tile being read -- which drained the NEXT tile's pieces (issued a few
    // instructions earlier) and every store in flight: the double buffer never overlapped anything (stamps: 2800 of a tile's 5500
    // cycles in the multiply phase, 16 MFMAs).  Completion is waited for by hand below, as before.
    const unsigned smem_lds = (unsigned)(unsigned long long)(__attribute__((address_space(3))) const char *)smem;
    auto stage = [&](int t, int buf) {
        const long long mb = (long long)(t0 + t) * TP;   // (wave-uniform)
        const char *sb = p.x + (size_t)mb * 512;
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            const int piece = PPW * wave + j;
            const int row = 2 * piece + (lane >> 5), cpos = lane & 31;
            int r = row;
            if (mb + r >= p.M) r = (int)(p.M - 1 - mb);  // (rows past the end re-read the last pixel; their results are not stored)
            const unsigned voff = (unsigned)(r * 512 + ((cpos ^ (row & 31)) << 4));
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep)
                         : "v"(voff), "s"(sb), "s"(__builtin_amdgcn_readfirstlane(smem_lds + buf * TBYTES + piece * 1024))
                         : "memory");
        }
    };
    stage(0, 0);
    YG_STAMP(0);                                         // prologue: my weight fragments
    // 16-byte store instructions this wave issues per full tile (wave-uniform)
    constexpr int NSTR = 32 * 32 / NT;                   // row-major layout: 32 rows x 32 chunks / threads
    int nst_prev = NSTR;
    if (p.taps > 0) {
        nst_prev = 0;
        for (int it = wave; it < 18; it += NW) nst_prev += (cg * 3 + it / 6) < p.taps ? 1 : 0;
    }
    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        if (t + 1 < nt) {
            stage(t + 1, buf ^ 1);                       // (its last readers passed the barrier that ended tile t-1)
            // tile t's pieces must have landed; younger than them and free to stay in flight: the previous tile's 16-byte
            // stores of THIS wave (nst_prev of them; always issued in the row-major layout, 0..5 in the tap-major one, where the
            // last slice holds fewer than three taps) and the PPW pieces just issued
            switch ((t == 0 ? 0 : nst_prev) + PPW) {
                case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
                case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
                case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
                case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
                case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
                case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
                case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
                default: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
            }
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        YG_STAMP(1);                                     // landing of tile t (and of the stores before it)
        __builtin_amdgcn_s_barrier();                    // (raw: __syncthreads() would drain the piece just issued)
        asm volatile("" ::: "memory");
        YG_STAMP(2);                                     // barrier
        f32x16 acc[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[ct][e] = 0.f;
        const char *ab = smem + buf * TBYTES + r32 * 512;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const u32x4 a = *(const u32x4 *)(ab + (((2 * kk + hh) ^ r32) << 4));
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) MmaD<DT>::run(wf[ct][kk], a, acc[ct]);
        }
        // lane = pixel r32; register e = column (e & 3) + 8 (e >> 2) + 4 hh of a 32-column tile.  The tile goes through an LDS image
        // [pixel][256 columns] so that the stores are whole 128-byte lines (512 B per pixel row, 16 B per lane): written as
        // 8-byte pieces straight from the registers, a line of Y was assembled from eight partial writes (222 -> 167 us for
        // the four levels).  The staging stores are inline asm: in front of an LDS store it can see, hipcc drains every LDS-DMA
        // piece in flight with a vmcnt(0).
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const unsigned sa = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char *)sst + (unsigned)(r32 * SROW + (wave * 32 * CT + 32 * ct + 4 * hh) * 2);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const uint2 v = make_uint2(pack2<DT>(acc[ct][4 * g], acc[ct][4 * g + 1]), pack2<DT>(acc[ct][4 * g + 2], acc[ct][4 * g + 3]));
                asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(sa), "v"(v), "n"(16 * g) : "memory");
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0xC07F);              // my LDS reads of tile t have returned, my staging writes are done ...
        YG_STAMP(3);                                     // reads + MFMAs + staging writes
        __builtin_amdgcn_s_barrier();                    // ... and everybody's: tile t's buffer may be refilled, the image is whole
        asm volatile("" ::: "memory");
        YG_STAMP(2);                                     // barrier
        if (p.taps > 0) {
            // tap-major Y: [tap][pixel][80 columns].  A tap's 160-byte rows of consecutive pixels are contiguous, so the four bilinear
            // corners of the sampling launch are two runs of 320 bytes and neighbouring output pixels share their cache lines.
            // A slice holds three whole taps (deform_y_col); a wave stores one (tap, 6-pixel group) per trip: lane = 10 * pixel +
            // chunk -> 960 contiguous bytes.
            const int px = lane / 10, k = lane - 10 * px;
            for (int it = wave; it < 18; it += NW) {                 // three taps per slice (deform_y_col) x six 6-pixel groups
                const int sgi = it / 6, pg = it - sgi * 6;
                const int tap = cg * 3 + sgi;
                const int row = pg * 6 + px;
                const long long m = (long long)(t0 + t) * TP + row;
                if (lane < 60 && row < TP && tap < p.taps && m < p.M)
                    *(u32x4 *)(p.y + ((size_t)tap * p.M + (size_t)m) * 160 + k * 16) = *(const u32x4 *)(sst + row * SROW + (sgi * 10 + k) * 16);
            }
        } else {
#pragma unroll
            for (int q = 0; q < NSTR; ++q) {
                const int row = (threadIdx.x >> 5) + (NT / 32) * q, ch = threadIdx.x & 31;      // 32 lanes = one pixel row of 512 B
                const long long m = (long long)(t0 + t) * TP + row;
                if (m < p.M) *(u32x4 *)(p.y + ((size_t)m * p.ycs + cg * 256) * 2 + ch * 16) = *(const u32x4 *)(sst + row * SROW + ch * 16);
            }
        }
        // (the next tile's staging writes come behind the barrier at the top of the next iteration: no wave can still be
        // reading this image then)
        YG_STAMP(4);                                     // issue of the stores
    }
    YG_STAMP_FLUSH;
}

// ---------------------------------------------------------------------------------------------
// ygemm_k256_v2 (round 5; tap-major Y only): the same tiles, fragments, K order and stores as the kernel above -- every Y element has the
// same bits -- in a schedule in which nothing waits for anything it does not need:
//  * ONE barrier per tile.  Behind it: tile t has landed (every wave waited for its own LDS-DMA pieces), the staged image of tile t - 1 is
//    whole, nobody reads tile t - 1's input buffer any more.  The period that follows issues the LDS-DMA of tile t + 1 (a whole period to
//    land), multiplies tile t and -- between its MFMAs -- carries tile t - 1's image from LDS to Y (two images, used alternately).
//  * the pixel fragments are read THREE MFMAs ahead (the compiler's own order was one ahead: a 16-long chain of read -> MFMA at ~110
//    cycles a link); the 16 swizzled read offsets are one XOR with a literal each instead of 16 registers.
// Stamps of the first version (diagnostics build, ticks per tile and wave): landing 830, barriers 870, reads + MFMAs + staging 2800
// (with hipcc's vmcnt(0) in front of the first read, see `stage` above), store issue 1040.
// ---------------------------------------------------------------------------------------------
template <typename DT>
__global__ __launch_bounds__(512, 4) void ygemm_k256_v2_kernel(const YGemmMulti mp)
{
    int prob = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i)
        if (i < mp.n && (int)blockIdx.x >= mp.block_start[i]) prob = i;
    const YGemmParams &p = mp.p[prob];
    const int local_blk = (int)blockIdx.x - mp.block_start[prob];
    const int ncg = p.N / 256;
    constexpr int TP = 32;                               // pixels per tile
    constexpr int TBYTES = TP * 512;                     // 16 KiB
    constexpr int SROW = 512 + 16;                       // staging row: 256 columns + a 16-byte pad (bank spread)
    constexpr int SBYTES = TP * SROW;
    constexpr int PPW = 2;                               // LDS-DMA pieces per wave and tile
    __shared__ __attribute__((aligned(16))) char smem[2 * TBYTES + 2 * SBYTES];
    char *const sst = smem + 2 * TBYTES;                 // two output images [pixel][256 columns]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r32 = lane & 31, hh = lane >> 5;
    const int part = local_blk / ncg, cg = local_blk - part * ncg;
    const int col0 = cg * 256 + wave * 32;
    u32x4 wf[16];                                        // fragment kk = channels [16kk + 8hh, +8) of column col0 + r32
    {
        const char *wr = p.w + ((size_t)(col0 + r32) * 256 + 8 * hh) * 2;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) wf[kk] = *(const u32x4 *)(wr + kk * 32);
    }
    const int t0 = part * p.tiles_per_part;
    int nt = (p.M + TP - 1) / TP - t0;
    nt = nt < p.tiles_per_part ? nt : p.tiles_per_part;
    if (nt <= 0) return;
    YG_STAMP_DECL

    const unsigned smem_lds = (unsigned)(unsigned long long)(__attribute__((address_space(3))) const char *)smem;
    // (as above: 32 rows x 512 B, chunk c of row r at position c ^ (r & 31); `sb` = the tile's first row, `avail` = rows of the tensor
    // from there on -- rows past the end re-read the last one, their results are not stored; kept incrementally: the 64-bit index
    // arithmetic of the first version was ~60 scalar instructions per tile in front of two DMA instructions)
    auto stage = [&](const char *sb, int avail, int buf) {
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            const int piece = PPW * wave + j;
            const int row = 2 * piece + (lane >> 5), cpos = lane & 31;
            const int r = row < avail ? row : avail - 1;
            const unsigned voff = (unsigned)(r * 512 + ((cpos ^ (row & 31)) << 4));
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep)
                         : "v"(voff), "s"(sb), "s"(__builtin_amdgcn_readfirstlane(smem_lds + buf * TBYTES + piece * 1024))
                         : "memory");
        }
    };
    // the store trips of this wave: trip j carries (tap, 6-pixel group) number it = wave + 8 j of the slice's 3 x 6 from the image to Y
    // (lane = 10 * pixel + chunk -> 960 contiguous bytes of Y); three per tile for waves 0 and 1, two for the others
    const int px = lane / 10, kch = lane - 10 * px;
    unsigned t_lds[3], t_off[3];
    int t_on[3], t_pg[3], nst = 0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int it = wave + 8 * j, sgi = it / 6, pg = it - sgi * 6, tap = cg * 3 + sgi, row = pg * 6 + px;
        t_on[j] = it < 18 && tap < p.taps;               // (wave-uniform)
        t_pg[j] = pg;
        const bool ok = lane < 60 && row < TP && t_on[j];
        t_lds[j] = ok ? (unsigned)(row * SROW + (sgi * 10 + kch) * 16) : 0u;
        t_off[j] = ok ? (unsigned)(((size_t)tap * p.M + (size_t)t0 * TP + row) * 160 + kch * 16) : 0u;     // (taps x M x 160 < 4 GiB: ygemm_fill)
        nst += t_on[j] ? 1 : 0;
    }
    auto carry_read = [&](int j, int img) -> u32x4 { return *(const u32x4 *)(sst + img * SBYTES + t_lds[j]); };
    __attribute__((address_space(1))) char *ybase = (__attribute__((address_space(1))) char *)p.y;
    asm volatile("" : "+s"(ybase));                      // (kept in scalar registers: re-loaded from the kernel arguments in front of every store,
                                                         // it came back through lgkmcnt and drained the LDS reads in flight)
    auto carry_store = [&](int j, int rows, const u32x4 &v) {
        if (t_on[j] && rows > 0) {                       // (rows == 0: the first period, nothing to carry yet)
            if (lane < 60 && t_pg[j] * 6 + px < rows) *(__attribute__((address_space(1))) u32x4 *)(ybase + (size_t)t_off[j]) = v;
            t_off[j] += TP * 160;
        }
    };
    // my swizzled read offsets: fragment kk of pixel r32 sits at r32 * 512 + (((2 kk + hh) ^ r32) << 4) = rd0 ^ (kk << 5)
    unsigned rd0 = (unsigned)(r32 * 512 + (((hh ^ r32) & 1) << 4) + ((r32 & 30) << 4));
    const unsigned sw = (unsigned)(r32 * SROW + (wave * 32 + 4 * hh) * 2);
    const char *sb = p.x + (size_t)t0 * TP * 512;        // (wave-uniform) first row of the tile to stage next
    int avail = p.M - t0 * TP;                           // rows from there to the end of the tensor (> 0 for every tile staged)
    stage(sb, avail, 0);
    sb += TBYTES; avail -= TP;
    YG_STAMP(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        if (t + 1 < nt) stage(sb, avail, buf ^ 1);       // (its last readers passed the barrier that ended period t - 1)
        sb += TBYTES; avail -= TP;                       // (avail is now the row count from tile t + 2 on: tile t - 1 had avail + 3 TP)
        YG_STAMP(1);
        const int left = avail + 3 * TP;
        const int rows_prev = t > 0 ? (left < TP ? left : TP) : 0;           // rows of tile t - 1 (0: nothing to carry yet)
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        const char *ab = smem + buf * TBYTES;
        u32x4 a[4], sv;
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) {
            asm volatile("" : "+v"(rd0));
            a[kk] = *(const u32x4 *)(ab + (rd0 ^ (unsigned)(kk << 5)));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            if (kk + 3 < 16) {
                asm volatile("" : "+v"(rd0));
                a[(kk + 3) & 3] = *(const u32x4 *)(ab + (rd0 ^ (unsigned)((kk + 3) << 5)));
            }
            if (kk == 1 || kk == 6 || kk == 11) sv = carry_read((kk - 1) / 5, buf ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            MmaD<DT>::run(wf[kk], a[kk & 3], acc);
            if (kk == 5 || kk == 10 || kk == 15) carry_store((kk - 5) / 5, rows_prev, sv);
            __builtin_amdgcn_sched_barrier(0);
        }
        // lane = pixel r32; register e = column (e & 3) + 8 (e >> 2) + 4 hh of my 32 columns -> the image of tile t
        {
            const unsigned sa = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char *)sst + (unsigned)(buf * SBYTES) + sw;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const uint2 v = make_uint2(pack2<DT>(acc[4 * g], acc[4 * g + 1]), pack2<DT>(acc[4 * g + 2], acc[4 * g + 3]));
                asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(sa), "v"(v), "n"(16 * g) : "memory");
            }
        }
        YG_STAMP(3);
        // my image rows are written, my reads of tile t have returned; tile t + 1's pieces (older than this period's stores) have landed
        if (t + 1 < nt) {
            switch (t > 0 ? nst : 0) {
                case 0: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); break;
                case 1: asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory"); break;
                case 2: asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory"); break;
                default: asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory"); break;
            }
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        YG_STAMP(4);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        YG_STAMP(2);
    }
    {   // the last tile's image
        const int left = p.M - (t0 + nt - 1) * TP;
        const int rows = left < TP ? left : TP;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const u32x4 v = carry_read(j, (nt - 1) & 1);
            carry_store(j, rows, v);
        }
    }
    YG_STAMP_FLUSH;
}

int ygemm_supported(int Cin, int ycols, int dtype) { return dtype != TDRN_F32 && Cin == 256 && ycols % 256 == 0; }

static int ygemm_fill(const YGemmProblem &q, int dtype, YGemmParams &p, int &blocks)
{
    if (!ygemm_supported(256, q.N, dtype) || q.M <= 0 || q.M >= (1ll << 31)) return TDRN_E_UNSUPPORTED;
    p.x = (const char *)q.x; p.w = (const char *)q.w; p.y = (char *)q.y;
    p.M = (int)q.M; p.N = q.N; p.ycs = q.ycs; p.taps = q.taps;
    if (q.taps > 0 && (deform_sample_cols(q.taps) > q.N || (long long)q.taps * q.M * 160 >= (1ll << 32))) return TDRN_E_UNSUPPORTED;
    const int tiles = (int)((q.M + 31) / 32), cgs = q.N / 256;
    // 512 workgroups are resident at a time (125 VGPRs: two 8-wave workgroups per CU).  A problem is cgs x parts workgroups of
    // ceil(tiles / parts) tiles each; a partition is at least 8 tiles (the weight prologue is 8 KiB per wave).  The number of
    // partitions is the candidate that minimises rounds x tiles per workgroup (measured with 256 / 512 / 768 / 1024 assumed slots:
    // the 20x20 level 45.6 / 36.0 / 44.4 / 44.1 us, the 40x40 level 123-126 us whatever the partitioning -- it is not bound by
    // rounds).
    const int kSlots = 512, max_parts = (tiles + 7) / 8 > 1 ? (tiles + 7) / 8 : 1;
    int parts = 1;
    long long best = -1;
    for (int r = 1; r <= 3; ++r) {
        int c = kSlots * r / cgs;
        c = c > max_parts ? max_parts : (c < 1 ? 1 : c);
        const long long rounds = ((long long)cgs * c + kSlots - 1) / kSlots, cost = rounds * ((tiles + c - 1) / c);
        if (best < 0 || cost < best) { best = cost; parts = c; }
    }
    p.parts = parts;
    p.tiles_per_part = (tiles + parts - 1) / parts;
    blocks = cgs * parts;
    return TDRN_OK;
}

int launch_ygemm_multi(const YGemmProblem *pr, int n, int dtype, hipStream_t s, int kdisable)
{
    if (!pr || n < 1 || n > 4) return TDRN_E_ARG;
    YGemmMulti mp;
    mp.n = 0;
    mp.block_start[0] = 0;
    // largest problem first: its workgroups are dispatched first, the small levels fill the tail
    int order[4] = {0, 1, 2, 3};
    for (int i = 0; i < n; ++i)
        for (int j = i + 1; j < n; ++j)
            if (pr[order[j]].M > pr[order[i]].M) { const int t = order[i]; order[i] = order[j]; order[j] = t; }
    for (int i = 0; i < n; ++i) {
        if (pr[order[i]].M <= 0) continue;
        int blocks = 0;
        TDRN_TRY(ygemm_fill(pr[order[i]], dtype, mp.p[mp.n], blocks));
        mp.block_start[mp.n + 1] = mp.block_start[mp.n] + blocks;
        ++mp.n;
    }
    if (mp.n == 0) return TDRN_OK;
    for (int i = mp.n; i < 4; ++i) { mp.p[i] = mp.p[0]; mp.block_start[i + 1] = mp.block_start[mp.n]; }
    dim3 grid((unsigned)mp.block_start[mp.n]);
#ifdef TDRN_YG_STAMP
    static unsigned *stamps = nullptr;
    static int calls = 0;
    const size_t nst = (size_t)grid.x * 8 * 6;
    if (!stamps) TDRN_HIP_TRY(hipMalloc((void **)&stamps, 4096 * 8 * 6 * sizeof(unsigned)));
    TDRN_HIP_TRY(hipMemsetAsync(stamps, 0, nst * sizeof(unsigned), s));
    mp.stamps = stamps;
    struct Report {
        hipStream_t s; size_t n; unsigned *d; YGemmMulti &mp; int *calls;
        ~Report() {
            if (++*calls != 5) return;                   // (one report, of a launch that is not the first)
            (void)hipStreamSynchronize(s);
            std::vector<unsigned> h(n);
            (void)hipMemcpy(h.data(), d, n * sizeof(unsigned), hipMemcpyDeviceToHost);
            for (int pr = 0; pr < mp.n; ++pr) {
                double a[5] = {0, 0, 0, 0, 0}, tiles = 0, waves = 0;
                for (int b = mp.block_start[pr]; b < mp.block_start[pr + 1]; ++b)
                    for (int w = 0; w < 8; ++w) {
                        const unsigned *q = &h[((size_t)b * 8 + w) * 6];
                        if (!q[5]) continue;
                        for (int k = 0; k < 5; ++k) a[k] += q[k];
                        tiles += q[5]; waves += 1;
                    }
                if (waves == 0) continue;
                fprintf(stderr, "yg_stamp problem %d M %d blocks %d tiles/wg %.1f | per wave: prologue %.0f | per tile: landing %.0f barriers %.0f mma+stage %.0f stores %.0f (ticks of s_memtime)\n",
                        pr, mp.p[pr].M, mp.block_start[pr + 1] - mp.block_start[pr], tiles / waves, a[0] / waves, a[1] / tiles, a[2] / tiles, a[3] / tiles, a[4] / tiles);
            }
        }
    } report{s, nst, stamps, mp, &calls};
#endif
    static int ct = -1;
    // CT = 2 (four waves x 64 columns: half the LDS reads per output) measured 337-344 us against 332-334 us for the pair of deform
    // launches (round 4, interleaved): the transform is not bound by its LDS reads; the round-3 shape stays the default
    if (ct < 0) { const char *e = getenv("TDRN_YGEMM_CT"); ct = e ? atoi(e) : 1; }
    static int v2 = -1;
    if (v2 < 0) { const char *e = getenv("TDRN_YGEMM_V2"); v2 = e ? atoi(e) : 1; }
    bool tapmajor = true;
    for (int i = 0; i < mp.n; ++i) tapmajor = tapmajor && mp.p[i].taps > 0;
    if (v2 && !(kdisable & 128) && tapmajor && ct == 1) {
        if (dtype == TDRN_BF16) hipLaunchKernelGGL((ygemm_k256_v2_kernel<bf16_t>), grid, dim3(512), 0, s, mp);
        else hipLaunchKernelGGL((ygemm_k256_v2_kernel<f16_t>), grid, dim3(512), 0, s, mp);
    } else if (ct == 1) {
        if (dtype == TDRN_BF16) hipLaunchKernelGGL((ygemm_k256_kernel<bf16_t, 1>), grid, dim3(512), 0, s, mp);
        else hipLaunchKernelGGL((ygemm_k256_kernel<f16_t, 1>), grid, dim3(512), 0, s, mp);
    } else {
        if (dtype == TDRN_BF16) hipLaunchKernelGGL((ygemm_k256_kernel<bf16_t, 2>), grid, dim3(256), 0, s, mp);
        else hipLaunchKernelGGL((ygemm_k256_kernel<f16_t, 2>), grid, dim3(256), 0, s, mp);
    }
    return hip_status(hipGetLastError());
}

int launch_ygemm(const void *x, const void *w, void *y, long long M, int N, int ycs, int dtype, hipStream_t s, int taps)
{
    const YGemmProblem q{x, w, y, M, N, ycs, taps};
    return launch_ygemm_multi(&q, 1, dtype, s);
}

// the fast path takes: 16-bit, stride 1, one deformable group, at most two branches with at most 34 taps together, Cout <= 80
int deform_sample_supported(const DeformArgs &a)
{
    if (a.dtype == TDRN_F32 || a.n_branches < 1 || a.n_branches > 2 || a.Cout > kSampleCols || a.Ho != a.H || a.Wo != a.W) return 0;
    int taps = 0;
    for (int i = 0; i < a.n_branches; ++i) {
        const DeformBranch &b = a.br[i];
        if (b.G != 1 || b.stride != 1 || (b.stride_w >= 1 && b.stride_w != 1)) return 0;
        taps += b.kh * b.kw;
    }
    return taps <= 34 ? taps : 0;
}
int deform_sample_cols(int taps) { return ((taps + 2) / 3) * 256; }
int deform_ts_max_batch(int H, int W, int ycs, int taps)
{
    // launch_ygemm: taps * M * 160 < 2^32 (tap-major) ; launch_deform_sample_multi: M * ycs * 2 < 2^32 ; M < 2^31
    const long long hw = (long long)H * W;
    const long long per = std::max((long long)taps * 160, (long long)ycs * 2) * hw;
    long long b = ((1ll << 32) - 1) / per;
    const long long bm = ((1ll << 31) - 1) / hw;
    b = b < bm ? b : bm;
    return (int)(b > (1 << 20) ? (1 << 20) : b);
}

// y[i]: the level's Y tensor ([B*H*W][ycs[i]], net dtype), computed by the caller's 1x1 GEMM with the branches' taps in order
int launch_deform_sample_multi(const DeformArgs *args, const void *const *y, const int *ycs, int n, hipStream_t s, int tap_major)
{
    if (!args || n < 1 || n > 4) return TDRN_E_ARG;
    SampleMulti mp;
    mp.n = 0;
    mp.block_start[0] = 0;
    for (int i = 0; i < n; ++i) {
        const DeformArgs &a = args[i];
        const int taps = deform_sample_supported(a);
        if (!taps || a.dtype != args[0].dtype) return TDRN_E_UNSUPPORTED;
        if ((long long)a.B * a.H * a.W * ycs[i] * 2 >= (1ll << 32)) return TDRN_E_UNSUPPORTED;     // 32-bit byte offsets into Y
        SampleParams p;
        p.y = (const char *)y[i];
        p.n_branches = a.n_branches; p.n_taps = taps;
        int col0 = 0;
        for (int k = 0; k < a.n_branches; ++k) {
            const DeformBranch &b = a.br[k];
            p.br[k] = SampleBranchP{b.off, b.off_stride, b.kh, b.kw, b.pad, b.pad_w < 0 ? b.pad : b.pad_w, b.dil, b.dil_w < 1 ? b.dil : b.dil_w, col0, b.off_rows, b.off_row0};
            col0 += b.kh * b.kw;
        }
        if (a.n_branches == 1) p.br[1] = p.br[0];
        p.M = a.B * a.H * a.W; p.H = a.H; p.W = a.W; p.ycs = ycs[i]; p.Cout = a.Cout; p.tap_major = tap_major;
        p.split = a.split > a.Cout ? a.Cout : a.split;
        p.out0 = a.out0; p.out1 = a.out1;
        p.o0_bs = a.o0_bs; p.o0_ps = a.o0_ps; p.o1_bs = a.o1_bs; p.o1_ps = a.o1_ps;
        if (p.M <= 0) continue;
        mp.p[mp.n] = p;
        mp.block_start[mp.n + 1] = mp.block_start[mp.n] + cdiv(p.M, 4);
        ++mp.n;
    }
    if (mp.n == 0) return TDRN_OK;
    for (int i = mp.n; i < 4; ++i) { mp.p[i] = mp.p[0]; mp.block_start[i + 1] = mp.block_start[mp.n]; }
    {
        static int remap = -1;
        if (remap < 0) { const char *e = getenv("TDRN_SAMPLE_XCD"); remap = e ? atoi(e) : 1; }
        mp.xcd_remap = remap;
    }
    dim3 grid((unsigned)mp.block_start[mp.n]);
    if (args[0].dtype == TDRN_BF16) hipLaunchKernelGGL((deform_sample_kernel<bf16_t>), grid, dim3(256), 0, s, mp);
    else hipLaunchKernelGGL((deform_sample_kernel<f16_t>), grid, dim3(256), 0, s, mp);
    return hip_status(hipGetLastError());
}

}  // namespace tdrn
