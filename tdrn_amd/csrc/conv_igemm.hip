// conv_igemm.hip -- dense convolution as an implicit GEMM on the gfx950 matrix cores.
//
// Replaces the cuDNN/ATen convolutions the reference reaches through nn.Conv2d /
// nn.ConvTranspose2d (model/networks.py:136-163, model/dualrefinedet_vggbn.py:30-117) with the
// BatchNorm (eval), bias, residual add and ReLU that follow them fused into the epilogue.
//
// GEMM view:  D[cout][pixel] = sum_k  Wt[cout][k] * X[pixel][k],   k = (tap, cin)
//   * activations are NHWC, so one K-step (one tap, 128 bytes of channels) of a pixel is one
//     contiguous, coalesced 128-byte line; padding taps read a zero page instead of branching.
//   * both operands go HBM/L2 -> LDS with LDS-DMA (global_load_lds_dwordx4, 16 B per lane);
//     the LDS image is lane-linear, so the bank-conflict swizzle is applied to the per-lane
//     SOURCE address and undone on the ds_read_b128 side (chunk ^= (row>>1)&7).
//   * weights are the MFMA "A" operand and pixels the "B" operand, so every lane ends up with
//     4 consecutive output channels of ONE pixel per accumulator quad: the epilogue transposes
//     through LDS with 16-byte writes and leaves the chip as whole NHWC rows (16 B per lane).
//   * fp32 mode uses v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain) on the same 128-byte tiles;
//     it is the reference-precision path the parity tests pin to 1e-3.
#include <cstdlib>

#include "kernels.h"

#ifndef TDRN_IGEMM_PRIO
#define TDRN_IGEMM_PRIO 0
#endif
namespace tdrn {

struct ConvParams {
    const char *in, *w, *res, *zero;
    const float *bias;
    char *out;
    int M, H, W, Cin, Ho, Wo, Cout, Npad;
    int kh, kw, stride, pad, dil, relu, out_f32, out_vec;
    long long o_bs, o_rs, o_cs, o_base, o_pr, o_pc;
    int n_tiles, Ktot;
    int out_linear;   // out element (m, c) at o_base + m*o_cs + c (plain NHWC tensor): no index decode
    int splits;       // split-K: blockIdx.y = K slice; raw fp32 partial tiles go to `partial`
    float *partial;   // [splits][phases][M][Npad] fp32
    int ablate;   // diagnostics only (TDRN_CONV_ABLATE): 1 = skip the K-loop loads, 2 = skip the MFMAs
    int batch_minor, B;   // batch_minor: GEMM row m = (ho*Wo + wo)*B + b instead of (b*Ho + ho)*Wo + wo (see conv_igemm_kernel)
};

template <typename DT> struct Mma;
template <> struct Mma<bf16_t> {
    __device__ static __forceinline__ void run(const u32x4 &a, const u32x4 &b, f32x16 &c)
    {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(i16x8, a), __builtin_bit_cast(i16x8, b), c, 0, 0, 0);
    }
};
template <> struct Mma<f16_t> {
    __device__ static __forceinline__ void run(const u32x4 &a, const u32x4 &b, f32x16 &c)
    {
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    // lane half h holds k = 4*(2kk+h)+j of the 128-byte row; A and B use the same map, so the
    // four x2 MFMAs below cover each k exactly once.
    __device__ static __forceinline__ void run(const u32x4 &a, const u32x4 &b, f32x16 &c)
    {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a[j]), __uint_as_float(b[j]), c, 0, 0, 0);
    }
};

// COH = agent-coherent access (the `sc1` cache-policy bit: served by / written through to the memory side, past the XCD's own
// L2): what conv_chain_kernel uses for every tensor one of its tasks writes and another -- possibly on another XCD -- reads,
// instead of agent-scope fences (a fence writes back / invalidates a whole L2: with ~3000 tasks a launch that was 3x slower
// than the launches it replaces and slowed every concurrent kernel by 3-5x).
template <bool COH = false>
__device__ __forceinline__ void glds16(const char *src, char *lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, COH ? 16 : 0);
}
template <bool COH>
__device__ __forceinline__ void st8(void *p, uint2 v)
{
    if constexpr (COH) __hip_atomic_store((unsigned long long *)p, ((unsigned long long)v.y << 32) | v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *(uint2 *)p = v;
}
template <bool COH>
__device__ __forceinline__ uint2 ld8(const void *p)
{
    if constexpr (COH) {
        const unsigned long long v = __hip_atomic_load((const unsigned long long *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return make_uint2((unsigned)v, (unsigned)(v >> 32));
    } else {
        return *(const uint2 *)p;
    }
}
template <bool COH>
__device__ __forceinline__ void st16(void *p, const u32x4 &v)
{
    if constexpr (COH) {
        st8<true>(p, make_uint2(v[0], v[1]));
        st8<true>((char *)p + 8, make_uint2(v[2], v[3]));
    } else {
        *(u32x4 *)p = v;
    }
}
template <bool COH>
__device__ __forceinline__ u32x4 ld16(const void *p)
{
    if constexpr (COH) {
        const uint2 a = ld8<true>(p), b = ld8<true>((const char *)p + 8);
        return u32x4{a.x, a.y, b.x, b.y};
    } else {
        return *(const u32x4 *)p;
    }
}

// BM = pixel tile, BN = cout tile, waves arranged WGM (pixels) x WGN (couts).
// STAGES = 2: one K-step prefetched, vmcnt(0) + __syncthreads() per step (small problems, 2 blocks/CU).
// STAGES = 3: LDS ring with TWO K-steps of LDS-DMA in flight across a raw s_barrier and a counted
//             s_waitcnt vmcnt(N) -- one workgroup per CU, latency hidden by prefetch depth, not occupancy.
template <int BM, int BN, int STAGES>
constexpr int igemm_lds_bytes()
{
    return (STAGES * (BM + BN) * 128 > BM * (BN * 4 + 16)) ? STAGES * (BM + BN) * 128 : BM * (BN * 4 + 16);
}

// One (pixel tile, cout tile) x K slice x phase of the GEMM: tile index `wg` (pixel-tile major), K slice `ky` of p.splits,
// phase `z` of `nz`.  Called by conv_igemm_kernel (one tile per workgroup) and by conv_chain_kernel (a queue of tiles of
// several dependent layers).  Every thread of the workgroup runs it; it contains workgroup barriers.
template <typename DT, int BM, int BN, int WGM, int WGN, int STAGES, bool COH = false>
__device__ __forceinline__ void igemm_tile(const ConvParams &p, const int wg, const int ky, const int z, const int nz, char *smem)
{
    constexpr int NT = 64 * WGM * WGN;
    constexpr int ES = elem_traits<DT>::bytes;
    constexpr int CK = 128 / ES;             // channels per K-step
    constexpr int RPP = NT / 8;              // tile rows staged per pass (8 lanes per 128-B row)
    constexpr int PA = BM / RPP, PB = BN / RPP;
    constexpr int STAGE = (BM + BN) * 128;
    constexpr int CS = BN * 4 + 16;          // fp32 C-tile row stride (bytes), padded
    constexpr int LDS = igemm_lds_bytes<BM, BN, STAGES>();
    constexpr int WP = BM / WGM / 32, WC = BN / WGN / 32;
    static_assert(LDS <= 160 * 1024, "LDS budget");
    static_assert(BM % RPP == 0 && BN % RPP == 0 && WP >= 1 && WC >= 1, "tile shape");

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int lrow = t >> 3;
    const int lc16 = ((t & 7) ^ ((lrow >> 1) & 7)) << 4;   // swizzled source chunk (bytes)
    const int mt = wg / p.n_tiles, nt = wg - mt * p.n_tiles;
    const int m0 = mt * BM, n0 = nt * BN;
    const char *wbase = p.w + (size_t)z * p.Npad * p.Ktot * ES;
    const long long obase = p.o_base + (z >> 1) * p.o_pr + (z & 1) * p.o_pc;
    const int HoWo = p.Ho * p.Wo;
    // GEMM row -> (image, pixel).  Small maps with padding use the batch-minor order: the 128 / 256 rows of a tile are
    // then the same few pixel positions of many images, so a tap that falls into the padding does so for the WHOLE tile
    // and its K-steps are skipped (loads and MFMAs): fc6 (3x3, dilation 6 on a 10x10 map) has 64 % of its (pixel, tap)
    // pairs in the padding.  A skipped step would have added exact zeros, so results are bit-identical either way.
    auto decode = [&](int m, int &b, int &rem) {
        if (p.batch_minor) { rem = m / p.B; b = m - rem * p.B; }
        else { b = m / HoWo; rem = m - b * HoWo; }
    };

    // per-thread pixel rows of the activation tile
    int hi0[PA], wi0[PA], pbase[PA];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int m = m0 + i * RPP + lrow;
        if (m < p.M) {
            int b, rem;
            decode(m, b, rem);
            const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
            hi0[i] = ho * p.stride - p.pad;
            wi0[i] = wo * p.stride - p.pad;
            pbase[i] = b * p.H * p.W * p.Cin;
        } else {
            hi0[i] = -(1 << 28);
            wi0[i] = 0;
            pbase[i] = 0;
        }
    }

    f32x16 acc[WC][WP];
#pragma unroll
    for (int ci = 0; ci < WC; ++ci)
#pragma unroll
        for (int pi = 0; pi < WP; ++pi)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[ci][pi][e] = 0.f;

    int tr = 0, tq = 0, c0 = 0, kofs = 0;   // current tap (row, col), channel offset, K offset
    auto stage = [&](int buf) {
#ifdef TDRN_DEV_ABLATE
        if (p.ablate & 1) return;
#endif
        char *sb = smem + buf * STAGE;
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const char *src = wbase + ((size_t)(n0 + i * RPP + lrow) * p.Ktot + kofs) * ES + lc16;
            glds16(src, sb + (i * RPP + wave * 8) * 128);
        }
        const int dh = tr * p.dil, dw = tq * p.dil;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int hi = hi0[i] + dh, wi = wi0[i] + dw;
            const bool ok = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
            const size_t off = (size_t)(pbase[i] + (hi * p.W + wi) * p.Cin + c0) * ES + lc16;
            glds16<COH>(ok ? p.in + off : p.zero, sb + BN * 128 + (i * RPP + wave * 8) * 128);
        }
    };
    // taps that touch the image for at least one row of this tile (bit = tr*kw + tq); all ones without padding
    unsigned tapmask = 0xFFFFFFFFu;
    if (p.pad > 0 && p.kh * p.kw <= 32) {
        unsigned mine = 0u;
        for (int a = 0; a < p.kh; ++a)
            for (int c = 0; c < p.kw; ++c) {
                bool any = false;
#pragma unroll
                for (int i = 0; i < PA; ++i)
                    any |= (unsigned)(hi0[i] + a * p.dil) < (unsigned)p.H && (unsigned)(wi0[i] + c * p.dil) < (unsigned)p.W;
                if (any) mine |= 1u << (a * p.kw + c);
            }
        for (int o = 32; o > 0; o >>= 1) mine |= __shfl_xor(mine, o, 64);
        unsigned *mk = (unsigned *)smem;
        if (t == 0) *mk = 0u;
        __syncthreads();
        if (lane == 0) atomicOr(mk, mine);
        __syncthreads();
        tapmask = *mk;
        __syncthreads();                     // (the first stage() overwrites this word)
    }
    auto advance = [&]() {
        c0 += CK;
        kofs += CK;
        if (c0 == p.Cin) {
            c0 = 0;
            do {                             // on to the next tap that is not all padding (the caller knows there is one)
                if (++tq == p.kw) { tq = 0; ++tr; }
                if (p.kh * p.kw > 32 || ((tapmask >> ((tr * p.kw + tq) & 31)) & 1u)) break;
                kofs += p.Cin;
            } while (tr < p.kh);
        }
    };

    const int r32 = lane & 31, hh = lane >> 5;
    const int sw = (r32 >> 1) & 7;
    const int wm = wave % WGM, wn = wave / WGM;
    const int prow0 = wm * (WP * 32) + r32, crow0 = wn * (WC * 32) + r32;
    auto compute = [&](int buf) {
#ifdef TDRN_DEV_ABLATE
        if (p.ablate & 2) return;
#endif
        const char *wsb = smem + buf * STAGE;
        const char *psb = wsb + BN * 128;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int ch = ((2 * kk + hh) ^ sw) << 4;
            u32x4 wf[WC], pf[WP];
#pragma unroll
            for (int ci = 0; ci < WC; ++ci) wf[ci] = *(const u32x4 *)(wsb + (crow0 + ci * 32) * 128 + ch);
#pragma unroll
            for (int pi = 0; pi < WP; ++pi) pf[pi] = *(const u32x4 *)(psb + (prow0 + pi * 32) * 128 + ch);
#pragma unroll
            for (int ci = 0; ci < WC; ++ci)
#pragma unroll
                for (int pi = 0; pi < WP; ++pi) Mma<DT>::run(wf[ci], pf[pi], acc[ci][pi]);
        }
    };

    int nk = p.kh * p.kw * (p.Cin / CK);
    {
        // this workgroup's K slice [ks0, ks1) of the FULL step sequence (split boundaries never depend on what is
        // skipped: a frame's partial sums must not depend on the batch it travels in); inside it only the steps of
        // live taps run.  Position the (tap, channel) cursor at the first live step.
        const int per = p.splits > 1 ? (nk + p.splits - 1) / p.splits : nk;
        const int ks0 = p.splits > 1 ? ky * per : 0, ks1 = min(nk, ks0 + per);
        const int cpt = p.Cin / CK;                  // K-steps per tap
        int live = 0, first = -1;
        for (int tap = ks0 / cpt; tap * cpt < ks1; ++tap) {
            if (p.kh * p.kw <= 32 && !((tapmask >> tap) & 1u)) continue;
            const int a = max(ks0, tap * cpt), e = min(ks1, (tap + 1) * cpt);
            if (first < 0) first = a;
            live += e - a;
        }
        nk = live;
        if (first < 0) first = ks0;
        const int tap0 = first / cpt;
        tr = tap0 / p.kw;
        tq = tap0 - tr * p.kw;
        c0 = (first - tap0 * cpt) * CK;
        kofs = first * CK;
    }
    if constexpr (STAGES == 2) {
        if (nk > 0) stage(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int ks = 0; ks < nk; ++ks) {
            const int cur = ks & 1;
            if (ks + 1 < nk) {
                advance();
                stage(cur ^ 1);
            }
            compute(cur);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    } else {
        // every stage() issues exactly PA+PB LDS-DMA instructions per lane, so "all but the newest
        // PA+PB" == "K-step ks has landed".  Read a buffer only after the wait AND the barrier that
        // follow its loads; refill it only after the barrier that follows its last read.
        constexpr int LPS = PA + PB;
        if (nk > 0) stage(0);                            // (dead-tap skipping can leave a split-K slice with no live K step)
        if (nk > 1) {
            advance();
            stage(1);
        }
        int cur = 0, nxt = 2;
        for (int ks = 0; ks < nk; ++ks) {
            if (ks + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (ks + 2 < nk) {
                advance();
                stage(nxt);
            }
#if TDRN_IGEMM_PRIO
            __builtin_amdgcn_s_setprio(1);
#endif
            compute(cur);
#if TDRN_IGEMM_PRIO
            __builtin_amdgcn_s_setprio(0);
#endif
            cur = cur + 1 == STAGES ? 0 : cur + 1;
            nxt = nxt + 1 == STAGES ? 0 : nxt + 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (no LDS-DMA may still be in flight when the epilogue re-uses smem)
        __syncthreads();
    }

    // ---- epilogue: + bias, transpose through LDS (fp32), residual, ReLU, coalesced store -----
    const bool partial_out = p.splits > 1;
#pragma unroll
    for (int ci = 0; ci < WC; ++ci) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int cl = wn * (WC * 32) + ci * 32 + 8 * g + 4 * hh;   // local cout of reg 4g
            const f32x4 bv = partial_out ? f32x4{0.f, 0.f, 0.f, 0.f} : *(const f32x4 *)(p.bias + n0 + cl);
#pragma unroll
            for (int pi = 0; pi < WP; ++pi) {
                f32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = acc[ci][pi][4 * g + j] + bv[j];
                *(f32x4 *)(smem + (prow0 + pi * 32) * CS + cl * 4) = v;
            }
        }
    }
    __syncthreads();

    if (partial_out) {
        // raw fp32 partial tile -> slab [split][phase][m][Npad]; bias / residual / ReLU happen in the reduce kernel
        constexpr int CPR = BN / 4;
        float *slab = p.partial + ((size_t)ky * nz + z) * (size_t)p.M * p.Npad;
        for (int idx = t; idx < BM * CPR; idx += NT) {
            const int row = idx / CPR, chn = idx - row * CPR;
            const int m = m0 + row;
            if (m >= p.M) continue;
            st16<COH>(slab + (size_t)m * p.Npad + n0 + chn * 4, *(const u32x4 *)(smem + row * CS + chn * 16));
        }
        return;
    }
    if (p.out_f32) {
        // fp32 heads (ARM loc / conf logits): 4 channels per chunk, scalar stores when unaligned
        constexpr int CPR = BN / 4;
        for (int idx = t; idx < BM * CPR; idx += NT) {
            const int row = idx / CPR, chn = idx - row * CPR;
            const int m = m0 + row, c = n0 + chn * 4;
            if (m >= p.M || c >= p.Cout) continue;
            int b, rem;
            decode(m, b, rem);
            const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
            float *dst = (float *)p.out + obase + b * p.o_bs + ho * p.o_rs + wo * p.o_cs + c;
            f32x4 v = *(const f32x4 *)(smem + row * CS + chn * 16);
            if (p.relu) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
            }
            if (p.out_vec && c + 4 <= p.Cout) {
                *(f32x4 *)dst = v;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (c + j < p.Cout) dst[j] = v[j];
            }
        }
    } else {
        constexpr int P16 = elem_traits<DT>::per16;
        constexpr int CPR = BN / P16;              // 16-byte chunks per tile row
        constexpr int RSTEP = NT / CPR;            // tile rows covered per pass
        static_assert(NT % CPR == 0, "store mapping");
        const int chn = t % CPR, c = n0 + chn * P16;
        if (c < p.Cout) {
            for (int row = t / CPR; row < BM; row += RSTEP) {
                const int m = m0 + row;
                if (m >= p.M) break;
                long long eo;
                if (p.out_linear && !p.batch_minor) {
                    eo = obase + (long long)m * p.o_cs + c;
                } else {
                    int b, rem;
                    decode(m, b, rem);
                    const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
                    eo = obase + b * p.o_bs + ho * p.o_rs + wo * p.o_cs + c;
                }
                float v[P16];
#pragma unroll
                for (int q = 0; q < P16 / 4; ++q) {
                    const f32x4 x = *(const f32x4 *)(smem + row * CS + chn * (P16 * 4) + q * 16);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[q * 4 + j] = x[j];
                }
                if (p.res) {
                    float rv[P16];
                    unpack16<DT>(ld16<COH>(p.res + eo * ES), rv);
#pragma unroll
                    for (int j = 0; j < P16; ++j) v[j] += rv[j];
                }
                if (p.relu) {
#pragma unroll
                    for (int j = 0; j < P16; ++j) v[j] = fmaxf(v[j], 0.f);
                }
                st16<COH>(p.out + eo * ES, pack16<DT>(v));
            }
        }
    }
}

template <typename DT, int BM, int BN, int WGM, int WGN, int STAGES>
__global__ __launch_bounds__(64 * WGM * WGN) void conv_igemm_kernel(const ConvParams p)
{
    __shared__ __attribute__((aligned(16))) char smem[igemm_lds_bytes<BM, BN, STAGES>()];
    // XCD-aware remap (bijective): workgroups are dealt round-robin over the 8 XCDs, each with its
    // own L2; give every XCD a contiguous run of tiles so that the cout-tile siblings of a pixel
    // tile and neighbouring pixel tiles (shared 3x3 halo) hit the same L2.
    const int nwg = gridDim.x, xq = nwg >> 3, xr = nwg & 7, xcd = blockIdx.x & 7;
    const int wg = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + ((int)blockIdx.x >> 3);
    igemm_tile<DT, BM, BN, WGM, WGN, STAGES>(p, wg, (int)blockIdx.y, (int)blockIdx.z, (int)gridDim.z, smem);
}

int conv_patch_enabled()
{
    static int use_patch = -1;
    if (use_patch < 0) { const char *e = getenv("TDRN_CONV_PATCH"); use_patch = e ? atoi(e) : 1; }
    return use_patch;
}

int conv_n_pad(int cout) { return cout <= 32 ? 32 : (cout <= 64 ? 64 : (int)align_up((size_t)cout, 128)); }

template <typename DT, int BM, int BN, int WGM, int WGN, int STAGES>
static int launch_cfg(const ConvParams &p, int phases, hipStream_t s)
{
    const int mt = cdiv(p.M, BM);
    ConvParams q = p;
    q.n_tiles = p.Npad / BN;
    dim3 grid((unsigned)(mt * q.n_tiles), (unsigned)q.splits, (unsigned)phases);
    hipLaunchKernelGGL((conv_igemm_kernel<DT, BM, BN, WGM, WGN, STAGES>), grid, dim3(64 * WGM * WGN), 0, s, q);
    return hip_status(hipGetLastError());
}

static int g_conv_variant = -1;   // TDRN_CONV_VARIANT: 0 = always the 2-stage kernels (A/B switch)

template <typename DT> static int launch_dt(const ConvParams &p, int phases, hipStream_t s)
{
    if (g_conv_variant < 0) {
        const char *e = getenv("TDRN_CONV_VARIANT");
        g_conv_variant = e ? atoi(e) : 1;
    }
    // the deep-pipelined 256x128 kernel runs one workgroup per CU: use it when the grid still has
    // at least ~2 waves of workgroups over the 256 CUs
    const long long big_tiles = (long long)cdiv(p.M, 256) * (p.Npad / 128) * phases;
    if (g_conv_variant >= 1 && p.Npad % 128 == 0 && big_tiles >= 512) return launch_cfg<DT, 256, 128, 4, 2, 3>(p, phases, s);
    if (p.Npad % 128 == 0) return launch_cfg<DT, 128, 128, 2, 2, 2>(p, phases, s);
    if (p.Npad % 64 == 0) return launch_cfg<DT, 128, 64, 2, 2, 2>(p, phases, s);
    return launch_cfg<DT, 128, 32, 4, 1, 2>(p, phases, s);
}

// split-K second pass: out = epilogue(sum_s partial[s]) with the same views / flags as the fused epilogue.
// One thread = 4 consecutive channels of one pixel (16-B slab reads, 8/16-B stores).
// elements [i0, i1) of the (phase, pixel, 4-channel group) index space, thread t of a 256-thread workgroup taking i0 + t,
// i0 + t + step, ...
template <typename DT, bool COH = false>
__device__ __forceinline__ void splitk_reduce_range(const ConvParams &p, const int phases, const long long i0, const long long i1,
                                                    const long long step)
{
    constexpr int ES = elem_traits<DT>::bytes;
    const int C4 = (p.Cout + 3) / 4;
    const int HoWo = p.Ho * p.Wo;
    const size_t slab = (size_t)phases * p.M * p.Npad;        // floats per K slice
    for (long long i = i0 + threadIdx.x; i < i1; i += step) {
        const int c = (int)(i % C4) * 4;
        const long long r = i / C4;
        const int m = (int)(r % p.M), z = (int)(r / p.M);
        const float *src = p.partial + ((size_t)z * p.M + m) * p.Npad + c;
        f32x4 v = *(const f32x4 *)(p.bias + c);
        for (int sidx = 0; sidx < p.splits; ++sidx) v += __builtin_bit_cast(f32x4, ld16<COH>(src + sidx * slab));
        long long eo;
        if (p.out_linear && !p.batch_minor) {
            eo = p.o_base + (long long)m * p.o_cs + c;
        } else {
            int b, rem;
            if (p.batch_minor) { rem = m / p.B; b = m - rem * p.B; }
            else { b = m / HoWo; rem = m - b * HoWo; }
            const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
            eo = p.o_base + (z >> 1) * p.o_pr + (z & 1) * p.o_pc + b * p.o_bs + ho * p.o_rs + wo * p.o_cs + c;
        }
        const bool full = c + 4 <= p.Cout;
        if (!p.out_f32 && full) {                     // NHWC tensor in the net dtype: channel counts are multiples of 8
            if (p.res) {
                if constexpr (ES == 4) {
                    const f32x4 rv = __builtin_bit_cast(f32x4, ld16<COH>(p.res + eo * 4));
                    v += rv;
                } else {
                    const uint2 rv = ld8<COH>(p.res + eo * 2);
                    DT t0{(unsigned short)(rv.x & 0xffffu)}, t1{(unsigned short)(rv.x >> 16)}, t2{(unsigned short)(rv.y & 0xffffu)}, t3{(unsigned short)(rv.y >> 16)};
                    v[0] += to_f32<DT>(t0); v[1] += to_f32<DT>(t1); v[2] += to_f32<DT>(t2); v[3] += to_f32<DT>(t3);
                }
            }
            if (p.relu) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
            }
            if constexpr (ES == 4) {
                st16<COH>(p.out + eo * 4, __builtin_bit_cast(u32x4, v));
            } else {
                const unsigned lo = pack2<DT>(v[0], v[1]), hi = pack2<DT>(v[2], v[3]);
                st8<COH>(p.out + eo * 2, make_uint2(lo, hi));
            }
            continue;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (c + j >= p.Cout) break;
            float x = v[j];
            if (p.res) x += to_f32<DT>(*(const DT *)(p.res + (eo + j) * ES));
            if (p.relu) x = fmaxf(x, 0.f);
            if (p.out_f32) ((float *)p.out)[eo + j] = x;
            else *(DT *)(p.out + (eo + j) * ES) = from_f32<DT>(x);
        }
    }
}

template <typename DT>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const ConvParams p, int phases)
{
    const long long total = (long long)phases * p.M * ((p.Cout + 3) / 4);
    splitk_reduce_range<DT>(p, phases, (long long)blockIdx.x * 256, total, (long long)gridDim.x * 256);
}

// number of K slices for a small-M problem: fill the chip (~2 workgroups per CU) but keep >= 4 K-steps each
int conv_splitk_choice(const ConvArgs &a)
{
    static int enabled = -1;
    if (enabled < 0) { const char *e = getenv("TDRN_SPLITK"); enabled = e ? atoi(e) : 1; }
    if (!enabled) return 1;
    const int es = dtype_bytes(a.dtype);
    const int nk = a.kh * a.kw * (a.Cin / (128 / es));
    const int bn = a.Npad % 128 == 0 ? 128 : (a.Npad % 64 == 0 ? 64 : 32);
    const long long blocks = (long long)cdiv(a.B * a.Ho * a.Wo, 128) * (a.Npad / bn) * a.phases;
    if (conv_patch_enabled() && patch_conv_supported(a) && a.H * a.W >= conv_patch_enabled() * 400) return 1;
    if (head3x3_supported(a)) return 1;                  // (head3x3.hip takes the launch whole)
    if (blocks >= 160 || nk < 8) return 1;
    int s = (int)((384 + blocks - 1) / blocks);
    if (s > nk / 4) s = nk / 4;
    if (s > 16) s = 16;
    return s < 2 ? 1 : s;
}
size_t conv_splitk_bytes(const ConvArgs &a, int splits)
{
    return splits > 1 ? (size_t)splits * a.phases * a.B * a.Ho * a.Wo * a.Npad * sizeof(float) : 0;
}

// ConvArgs -> the kernels' parameter block (validation included)
static int make_params(const ConvArgs &a, ConvParams &p)
{
    if (!a.in || !a.w || !a.out || !a.bias || !a.zero_page) return TDRN_E_ARG;
    const int es = dtype_bytes(a.dtype);
    const int ck = 128 / es;
    if (a.Cin % ck != 0 || a.Npad % 32 != 0 || a.Npad < a.Cout || a.Cout <= 0) return TDRN_E_UNSUPPORTED;
    if (a.phases != 1 && a.phases != 4) return TDRN_E_ARG;
    if (!a.out_f32 && (a.Cout % (16 / es) != 0)) return TDRN_E_UNSUPPORTED;
    if ((long long)a.B * a.H * a.W * a.Cin >= (1ll << 31)) return TDRN_E_UNSUPPORTED;
    p.in = (const char *)a.in;
    p.w = (const char *)a.w;
    p.res = (const char *)a.res;
    p.zero = (const char *)a.zero_page;
    p.bias = a.bias;
    p.out = (char *)a.out;
    p.M = a.B * a.Ho * a.Wo;
    p.H = a.H; p.W = a.W; p.Cin = a.Cin; p.Ho = a.Ho; p.Wo = a.Wo; p.Cout = a.Cout; p.Npad = a.Npad;
    p.kh = a.kh; p.kw = a.kw; p.stride = a.stride; p.pad = a.pad; p.dil = a.dil;
    p.relu = a.relu; p.out_f32 = a.out_f32;
    p.o_bs = a.o_bs; p.o_rs = a.o_rs; p.o_cs = a.o_cs; p.o_base = a.o_base; p.o_pr = a.o_pr; p.o_pc = a.o_pc;
    p.out_vec = (a.out_f32 && a.o_bs % 4 == 0 && a.o_rs % 4 == 0 && a.o_cs % 4 == 0 && a.o_base % 4 == 0 &&
                 a.o_pr % 4 == 0 && a.o_pc % 4 == 0 && ((uintptr_t)a.out % 16 == 0)) ? 1 : 0;
    p.Ktot = a.kh * a.kw * a.Cin;
    p.out_linear = (!a.out_f32 && a.phases == 1 && a.o_rs == (long long)a.Wo * a.o_cs &&
                    a.o_bs == (long long)a.Ho * a.Wo * a.o_cs) ? 1 : 0;
    p.n_tiles = 0;
    p.B = a.B;
    {
        static int bm = -1;                  // TDRN_IGEMM_BATCH_MINOR=0: keep the image-major row order (A/B switch)
        if (bm < 0) { const char *e = getenv("TDRN_IGEMM_BATCH_MINOR"); bm = e ? atoi(e) : 1; }
        p.batch_minor = (bm && a.pad > 0 && a.phases == 1 && a.B > 1 && a.Ho * a.Wo <= 1600 && a.kh * a.kw <= 32) ? 1 : 0;
    }
    p.splits = (a.splitk > 1 && a.partial) ? a.splitk : 1;
    p.partial = (float *)a.partial;
    static int ablate = -1;
    if (ablate < 0) ablate = dev_ablate_env("TDRN_CONV_ABLATE");     // (developer builds only: common.h)
    p.ablate = ablate;
    return TDRN_OK;
}

int launch_conv(const ConvArgs &a, hipStream_t s)
{
    ConvParams p;
    TDRN_TRY(make_params(a, p));
    if (p.M <= 0) return TDRN_OK;
    // 3x3/s1/p1 layers with enough tiles go to the warp-specialised patch kernel (TDRN_CONV_PATCH=0: off)
    static int use_patch = -1;
    if (use_patch < 0) { const char *e = getenv("TDRN_CONV_PATCH"); use_patch = e ? atoi(e) : 1; }
    // (chosen by layer geometry only, never by batch size: a frame's result must not depend on
    // what else is in the batch)
    if (p.splits == 1 && use_patch && patch_conv_supported(a) && a.H * a.W >= use_patch * 400)
        return launch_conv3x3_patch(a, nullptr, s);
    if (a.fuse_x) return TDRN_E_UNSUPPORTED;             // only the patch kernel computes the first conv itself
    if (p.splits == 1 && head3x3_supported(a)) return launch_head3x3(a, s);      // the narrow fp32 heads (ARM loc)
    if (p.splits == 1 && pw1x1_supported(a)) {           // wide pointwise layers: dwpw.hip's persistent GEMM (same bits)
        const int rc1 = launch_pw1x1(a, s);
        if (rc1 != TDRN_E_UNSUPPORTED) return rc1;       // (it declines launches too small to fill the chip)
    }
    int rc = TDRN_E_ARG;
    switch (a.dtype) {
        case TDRN_F32: rc = launch_dt<float>(p, a.phases, s); break;
        case TDRN_BF16: rc = launch_dt<bf16_t>(p, a.phases, s); break;
        case TDRN_F16: rc = launch_dt<f16_t>(p, a.phases, s); break;
    }
    if (rc != TDRN_OK || p.splits == 1) return rc;
    const long long total = (long long)a.phases * p.M * ((p.Cout + 3) / 4);
    dim3 grid((unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256));
    switch (a.dtype) {
        case TDRN_F32: hipLaunchKernelGGL((splitk_reduce_kernel<float>), grid, dim3(256), 0, s, p, a.phases); break;
        case TDRN_BF16: hipLaunchKernelGGL((splitk_reduce_kernel<bf16_t>), grid, dim3(256), 0, s, p, a.phases); break;
        case TDRN_F16: hipLaunchKernelGGL((splitk_reduce_kernel<f16_t>), grid, dim3(256), 0, s, p, a.phases); break;
    }
    return hip_status(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// conv_chain: several small, mutually dependent layers in ONE launch.
//
// The top of the pyramid (conv7's successors: extras, the last TCB level, its up-sampling, the 10x10 lateral) is a chain of
// layers with 0.4-15 GFLOP each at batch 32: as launches each costs 18-40 us of launch + split-K reduce + dependency
// latency for ~5 us of arithmetic, and they sit on the step's critical path.  Here every layer is a STAGE whose tiles
// (igemm_tile: the same code and the same K order as the stand-alone launch, so every output bit is the same) and split-K
// reduce ranges are TASKS in one queue, ordered so that a task depends only on earlier tasks.  A workgroup takes the next
// task with an atomic counter, waits until the stages it depends on are complete (per-stage completion counters,
// agent-scope acquire), runs it, and signals (agent-scope release).  Because a task is only ever taken by a RUNNING
// workgroup and waits only for tasks taken before it, the launch cannot deadlock however few of its workgroups are resident
// (persistent convs of other stream lanes may hold most CUs).  Independent stages (the lateral of the 10x10 level beside the
// 5x5 chain) overlap by themselves.
//
// MEASURED, AND NOT THE DEFAULT PLAN (opt-in: TDRN_PLAN_CHAIN / TDRN_CHAIN=1).  Nine layers, ~3200 tasks at batch 32, bit-identical
// to the nine launches (tested): 664 us alone / 717 us in the step with coherent (`sc1`) accesses for everything a task hands
// to another, 806 us with plain accesses + one agent-scope release and acquire per task (which also slowed every concurrent
// kernel 3-5x: each release writes an XCD's L2 back) -- against 170 us alone / 270 us in the step for the launches it
// replaces.  A hand-off between workgroups costs what a kernel boundary costs on this chip (per-XCD L2s are not coherent
// with each other and a CU's L1 is never refreshed by another CU's stores, MI355X_MICROARCH.md "splitk-seam": 5-13 us per seam),
// `sc1` loads pay the memory-side latency on every K step of a 2-stage pipeline, and each task adds two atomic round trips.
// profiles/r03_experiments.md has the timeline.
// ---------------------------------------------------------------------------------------------
constexpr int kChainMax = 12;
constexpr int kChainRedPerTask = 1024;          // (phase, pixel, 4-channel) elements per reduce task
struct ChainStage {
    ConvParams p;
    int phases, tiles, gemm_tasks, red_tasks, task0, dep0, dep1, pad_;
};
struct ChainParams {
    int n, total;
    unsigned *ctr;                              // [0] queue head, [1 + 2s] tiles done of stage s, [2 + 2s] reduce tasks done; zero on entry
    unsigned *status;                           // host-visible status words (or null): [1] <- 1 when a bounded wait runs out (tdrn_net_check)
    ChainStage st[kChainMax];
};
static_assert(sizeof(ChainParams) <= 4096, "kernel argument block");

// One agent-scope atomic add issued by lane 0 only, with NO lane-divergent control flow for the compiler to structurize: the
// exec mask is narrowed inside the asm block.  (A first version used `if (threadIdx.x == 0)` around the queue fetch at the top
// and the completion signal at the bottom of the task loop; hipcc's loop structurizer let lanes 1..63 of wave 0 run on into the
// next iteration's barrier while lane 0 was parked until "the others leave the loop" -- the fetch never happened and every
// workgroup repeated its first task forever.  Everything in the loop below is wave-uniform.)
__device__ __forceinline__ unsigned lane0_atomic_add(unsigned *p, unsigned v)
{
    unsigned old, zero = 0u;
    unsigned long long save;
    asm volatile("s_mov_b64 %1, exec\n\t"
                 "s_mov_b64 exec, 1\n\t"
                 "global_atomic_add %0, %2, %3, %4 sc0\n\t"
                 "s_waitcnt vmcnt(0)\n\t"
                 "s_mov_b64 exec, %1"
                 : "=&v"(old), "=&s"(save)
                 : "v"(zero), "v"(v), "s"(p)
                 : "memory");
    return (unsigned)__builtin_amdgcn_readfirstlane((int)old);
}

template <typename DT>
__global__ __launch_bounds__(256) void conv_chain_kernel(const ChainParams cp)
{
    __shared__ __attribute__((aligned(16))) char smem[igemm_lds_bytes<128, 128, 2>()];
    __shared__ int s_task;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    // wave 0, all lanes (the same address: one request): poll until *c >= n.  Bounded: a wait that runs out -- it never should --
    // is counted in ctr[40], its task kept in ctr[41], and REPORTED through the host-visible status word (tdrn_net_check).
    auto wait_for = [&](const unsigned *c, unsigned n, int task) {
        unsigned spins = 0;
        while ((unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < n) {
            if (++spins >= (1u << 18)) {
                (void)lane0_atomic_add(cp.ctr + 40, 1u);
                (void)lane0_atomic_add(cp.ctr + 41, (unsigned)task);
                if (cp.status) __hip_atomic_store(cp.status + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // (all lanes: the same word, the same value)
                break;
            }
            __builtin_amdgcn_s_sleep(16);
        }
    };
    int done_word = -1;                                         // completion counter of the task just finished
    for (;;) {
        if (wave == 0) {
            if (done_word >= 0) (void)lane0_atomic_add(cp.ctr + done_word, 1u);   // (every wave's coherent stores had left before the barrier)
            const int next = (int)lane0_atomic_add(cp.ctr, 1u);
            s_task = next;                                      // (all lanes store the same value)
        }
        __syncthreads();
        const int task = __builtin_amdgcn_readfirstlane(s_task);
        if (task >= cp.total) return;
        int s = 0;
        while (s + 1 < cp.n && task >= cp.st[s + 1].task0) ++s;
        s = __builtin_amdgcn_readfirstlane(s);
        const ChainStage &st = cp.st[s];
        const int local = task - st.task0;
        const bool red = local >= st.gemm_tasks;
        if (wave == 0) {
            if (red) {
                wait_for(cp.ctr + 1 + 2 * s, (unsigned)st.gemm_tasks, task);
            } else {
                for (int k = 0; k < 2; ++k) {
                    const int d = k ? st.dep1 : st.dep0;
                    if (d < 0) continue;
                    if (cp.st[d].red_tasks) wait_for(cp.ctr + 2 + 2 * d, (unsigned)cp.st[d].red_tasks, task);
                    else wait_for(cp.ctr + 1 + 2 * d, (unsigned)cp.st[d].gemm_tasks, task);
                }
            }
        }
        __syncthreads();
        if (!red) {
            const int per_z = st.tiles * st.p.splits;
            const int z = local / per_z, r = local - z * per_z;
            const int ky = r / st.tiles, wg = r - ky * st.tiles;
            igemm_tile<DT, 128, 128, 2, 2, 2, true>(st.p, wg, ky, z, st.phases, smem);
        } else {
            const long long total = (long long)st.phases * st.p.M * ((st.p.Cout + 3) / 4);
            const long long i0 = (long long)(local - st.gemm_tasks) * kChainRedPerTask;
            const long long i1 = i0 + kChainRedPerTask < total ? i0 + kChainRedPerTask : total;
            splitk_reduce_range<DT, true>(st.p, st.phases, i0, i1, 256);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's stores have left
        __syncthreads();
        done_word = (red ? 2 : 1) + 2 * s;                         // signalled by wave 0 at the top of the next round
    }
}

size_t conv_chain_ctr_bytes() { return 256; }    // 1 + 2 * kChainMax counters, two diagnostic words at [40], [41]
int conv_chain_max_layers() { return kChainMax; }

int conv_chain_supported(const ConvArgs &a)
{
    if (a.out_f32 || a.fuse_x || a.Npad % 128 != 0) return 0;
    if (a.splitk == 1 && conv_patch_enabled() && patch_conv_supported(a) && a.H * a.W >= conv_patch_enabled() * 400) return 0;   // the patch kernels' layer
    return 1;
}

int launch_conv_chain(const ChainLayer *layers, int n, unsigned *ctr, hipStream_t s, unsigned *status)
{
    if (!layers || n <= 0 || n > kChainMax || !ctr) return TDRN_E_ARG;
    ChainParams cp;
    cp.n = n;
    cp.ctr = ctr;
    cp.status = status;
    int task0 = 0;
    const int dtype = layers[0].a.dtype;
    for (int i = 0; i < n; ++i) {
        const ConvArgs &a = layers[i].a;
        if (a.dtype != dtype || !conv_chain_supported(a)) return TDRN_E_UNSUPPORTED;
        ChainStage &st = cp.st[i];
        TDRN_TRY(make_params(a, st.p));
        st.p.n_tiles = a.Npad / 128;
        st.phases = a.phases;
        st.tiles = cdiv(st.p.M, 128) * st.p.n_tiles;
        st.gemm_tasks = st.tiles * st.p.splits * a.phases;
        const long long red_total = (long long)a.phases * st.p.M * ((st.p.Cout + 3) / 4);
        st.red_tasks = st.p.splits > 1 ? (int)((red_total + kChainRedPerTask - 1) / kChainRedPerTask) : 0;
        st.task0 = task0;
        st.dep0 = layers[i].dep[0];
        st.dep1 = layers[i].dep[1];
        if (st.dep0 >= i || st.dep1 >= i) return TDRN_E_ARG;      // a stage may only depend on earlier stages
        st.pad_ = 0;
        task0 += st.gemm_tasks + st.red_tasks;
    }
    cp.total = task0;
    if (cp.total <= 0) return TDRN_OK;
    const int grid = cp.total < 512 ? cp.total : 512;             // two 66-KiB workgroups per CU
    switch (dtype) {
        case TDRN_F32: hipLaunchKernelGGL((conv_chain_kernel<float>), dim3(grid), dim3(256), 0, s, cp); break;
        case TDRN_BF16: hipLaunchKernelGGL((conv_chain_kernel<bf16_t>), dim3(grid), dim3(256), 0, s, cp); break;
        case TDRN_F16: hipLaunchKernelGGL((conv_chain_kernel<f16_t>), dim3(grid), dim3(256), 0, s, cp); break;
        default: return TDRN_E_ARG;
    }
    return hip_status(hipGetLastError());
}

}  // namespace tdrn
