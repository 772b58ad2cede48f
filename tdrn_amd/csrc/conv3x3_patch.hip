// conv3x3_patch.hip -- 3x3 / stride 1 / pad 1 convolution (the 13 VGG trunk convs and the TCB convs:
// 95 % of the network's FLOPs) as a warp-specialised, persistent direct convolution on MFMA.
//
// Why not the generic implicit GEMM (conv_igemm.hip) for these layers: measured there (rocprofv3 PMC +
// ablation builds, profiles/r01_*), the K loop is bound by LDS-DMA issue and L2->LDS traffic, not by
// the matrix pipe: every one of the 9 taps re-stages the same activations, and all waves run the
// same load->MFMA program in lockstep, so loads and MFMAs barely overlap (28 % MFMA utilisation).
//
// Structure (one 512-thread workgroup per CU, persistent over output tiles):
//   * waves 4-7 = LOADERS.  Per 128-byte channel chunk they bring the input PATCH of the tile
//     (tile + 1-pixel halo, <= 352 rows of 128 B) into LDS once -- the 9 taps are then just shifted
//     row addresses into that image -- plus the [BN x 128 B] weight slice of every (chunk, tap) step
//     into a 3-slot ring, two steps ahead.  All by global_load_lds_dwordx4 with the bank swizzle on
//     the source address; out-of-image rows read a zero page.  Waits are counted vmcnt(N).
//   * waves 0-3 = CONSUMERS.  ds_read_b128 + v_mfma only: each owns 64 pixels x BN couts
//     (2 x BN/32 accumulator tiles), weights as the A operand so a lane holds 4 consecutive couts
//     of one pixel.  Epilogue per wave through a private LDS strip -> whole-line NHWC stores, with
//     bias, ReLU and (optionally) the following 2x2 max-pool fused in registers.
//   * one s_barrier per (chunk, tap) step; loaders run ahead across tile boundaries, so the next
//     tile's first operands arrive while the consumers are still in the previous tile's epilogue.
//
// Two tile geometries share the code: "2-D" tiles of 8x32 or 16x16 output pixels of one image (the
// loader zero-fills the halo outside the image), and "flat" tiles of 256 consecutive NHW pixels for
// narrow maps (W <= 43: 40x40, 20x20), where the per-lane tap-validity mask zeroes fragments whose
// tap falls outside the image.
#include <cstdlib>

#include "kernels.h"

namespace tdrn {

struct PatchParams {
    const char *in, *w, *zero;
    const float *bias;
    char *out, *out_pool;          // NHWC [B][H][W][Cs] and optional pooled [B][H/2][W/2][Cs]
    int B, H, W, Cin, Cout, Cs, Ktot;
    int relu;
    int tw;                        // 32 / 16: 2-D tiles of (256/tw) x tw ; 0: flat tiles
    int tiles_x, tiles_per_img;    // 2-D mode
    int m_tiles, n_tiles, items;   // items = m_tiles * n_tiles
    int M;                         // B*H*W
};

namespace {

template <typename DT> struct MmaP;
template <> struct MmaP<bf16_t> {
    __device__ static __forceinline__ void run(const u32x4 &a, const u32x4 &b, f32x16 &c)
    { c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(i16x8, a), __builtin_bit_cast(i16x8, b), c, 0, 0, 0); }
};
template <> struct MmaP<f16_t> {
    __device__ static __forceinline__ void run(const u32x4 &a, const u32x4 &b, f32x16 &c)
    { c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0); }
};
template <> struct MmaP<float> {
    __device__ static __forceinline__ void run(const u32x4 &a, const u32x4 &b, f32x16 &c)
    {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a[j]), __uint_as_float(b[j]), c, 0, 0, 0);
    }
};

__device__ __forceinline__ void glds(const char *src, char *dst)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                     (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
}

__device__ __forceinline__ void wait_vmcnt(int n)   // n is wave-uniform
{
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    }
}

constexpr int kPatchSlots = 44;                 // 8-row LDS-DMA pieces per patch buffer (352 rows)
constexpr int kPatchBytes = kPatchSlots * 1024;
constexpr int kSlotsPerLoader = kPatchSlots / 4;

// position of the persistent workgroup in the (item, chunk, tap) step sequence
struct Cursor {
    int it, item, cc, tap;      // iteration, work item (m_tile, n_tile), channel chunk, tap 0..8
    bool valid;
};

}  // namespace

template <typename DT, int BN>
__global__ __launch_bounds__(512) void conv3x3_patch_kernel(const PatchParams p)
{
    constexpr int ES = elem_traits<DT>::bytes;
    constexpr int P16 = elem_traits<DT>::per16;
    constexpr int CK = 128 / ES;
    constexpr int WC = BN / 32;                         // cout tiles per consumer
    constexpr int WBYTES = BN * 128;                    // one weight slot
    constexpr int WL = BN / 32;                         // weight LDS-DMA pieces per loader wave per step
    constexpr int SROWS = ES == 2 ? 16 : 8;             // pixels per epilogue round (per wave)
    constexpr int SSTRIDE = BN * ES + 16;               // staging row stride (bytes)
    constexpr int OFF_W = 2 * kPatchBytes;
    constexpr int OFF_S = OFF_W + 3 * WBYTES;
    constexpr int LDS = OFF_S + 4 * SROWS * SSTRIDE;
    static_assert(LDS <= 160 * 1024, "LDS budget");
    __shared__ __attribute__((aligned(16))) char smem[LDS];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nchunks = p.Cin / CK;
    const int G = gridDim.x;

    // ---- work distribution: each XCD (blockIdx % 8) owns a contiguous range of items so that cout
    // siblings of a pixel tile and neighbouring tiles share its L2.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int per_xcd = (p.items + 7) >> 3, cus_per_xcd = (G + 7) >> 3;
    auto item_of = [&](int it) -> int {
        const int local = it * cus_per_xcd + slot;
        if (local >= per_xcd) return -1;
        const int item = xcd * per_xcd + local;
        return item < p.items ? item : -1;
    };
    auto first = [&]() {
        Cursor c;
        c.it = 0; c.item = item_of(0); c.cc = 0; c.tap = 0; c.valid = c.item >= 0;
        return c;
    };
    auto step = [&](Cursor &c) {           // advance by one (chunk, tap) step
        if (!c.valid) return;
        if (++c.tap == 9) {
            c.tap = 0;
            if (++c.cc == nchunks) {
                c.cc = 0;
                ++c.it;
                c.item = item_of(c.it);
                c.valid = c.item >= 0;
            }
        }
    };
    auto next_chunk = [&](Cursor &c) {     // advance to the first step of the next chunk
        if (!c.valid) return;
        c.tap = 0;
        if (++c.cc == nchunks) {
            c.cc = 0;
            ++c.it;
            c.item = item_of(c.it);
            c.valid = c.item >= 0;
        }
    };

    const int RS = p.tw ? p.tw + 2 : p.W;              // patch row stride of one image row

    if (wave >= 4) {
        // =========================== LOADER ===========================
        const int lw = wave - 4;
        const int lrow = lane >> 3, pc = lane & 7;
        unsigned poff[kSlotsPerLoader];                 // byte offset of my 16 B in the tensor, or ~0u
        int poff_item = -2;
        auto patch_table = [&](int item) {
            const int mt = item / p.n_tiles;
            if (mt == poff_item) return;
            poff_item = mt;
#pragma unroll
            for (int j = 0; j < kSlotsPerLoader; ++j) {
                const int pr = (lw + 4 * j) * 8 + lrow;
                const int lc16 = (pc ^ ((pr >> 1) & 7)) << 4;
                long long pix = -1;
                if (p.tw) {
                    const int th = 256 / p.tw;
                    const int b = mt / p.tiles_per_img, tt = mt - b * p.tiles_per_img;
                    const int ty = tt / p.tiles_x, tx = tt - ty * p.tiles_x;
                    const int py = pr / RS, px = pr - py * RS;
                    const int y = ty * th - 1 + py, x = tx * p.tw - 1 + px;
                    if (py < th + 2 && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W)
                        pix = ((long long)b * p.H + y) * p.W + x;
                } else {
                    const long long j0 = (long long)mt * 256 - p.W - 1 + pr;
                    if (pr < 256 + 2 * p.W + 2 && j0 >= 0 && j0 < p.M) pix = j0;
                }
                poff[j] = pix < 0 ? 0xFFFFFFFFu : (unsigned)(pix * p.Cin * ES + lc16);
            }
        };
        unsigned woff[WL];                              // weight row part of the source offset
#pragma unroll
        for (int k = 0; k < WL; ++k) {
            const int n = (lw + 4 * k) * 8 + lrow;
            woff[k] = (unsigned)((size_t)n * p.Ktot * ES + ((pc ^ ((n >> 1) & 7)) << 4));
        }
        auto load_weights = [&](const Cursor &c, int sidx) {     // step sidx -> ring slot sidx % 3
            char *dst = smem + OFF_W + (sidx % 3) * WBYTES;
            const int nt = c.item % p.n_tiles;
            const char *src = p.w + ((size_t)nt * BN * p.Ktot + (size_t)c.tap * p.Cin + (size_t)c.cc * CK) * ES;
#pragma unroll
            for (int k = 0; k < WL; ++k) glds(src + woff[k], dst + (lw + 4 * k) * 1024);
        };
        auto load_patch_slot = [&](int cc, int buf, int j) {
            const unsigned o = poff[j];
            glds(o == 0xFFFFFFFFu ? p.zero : p.in + (size_t)o + (size_t)cc * 128, smem + buf * kPatchBytes + (lw + 4 * j) * 1024);
        };

        Cursor wc = first();            // weights cursor (runs 2 steps ahead)
        Cursor pcur = first();          // patch cursor (runs 1 chunk ahead)
        int chunk_no = 0;               // global chunk counter of the patch cursor (buffer = chunk_no & 1)
        if (pcur.valid) {
            patch_table(pcur.item);
#pragma unroll
            for (int j = 0; j < kSlotsPerLoader; ++j) load_patch_slot(pcur.cc, 0, j);
            load_weights(wc, 0);
            step(wc);
            if (wc.valid) load_weights(wc, 1);
            step(wc);
        }
        next_chunk(pcur);
        chunk_no = 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();

        Cursor cur = first();
        for (int s = 0; cur.valid; ++s) {
            // operands of step s are in LDS.  Issue: weights of step s+2, and this tap's share of the
            // next chunk's patch (taps 0-4: two pieces, tap 5: one).
            int issued = 0;
            if (wc.valid) {
                load_weights(wc, s + 2);
                issued += WL;
            }
            step(wc);
            if (pcur.valid) {
                if (cur.tap == 0) patch_table(pcur.item);
                // static register indices (a runtime-indexed poff[] would live in scratch)
#pragma unroll
                for (int j = 0; j < kSlotsPerLoader; ++j)
                    if ((j >> 1) == cur.tap) {
                        load_patch_slot(pcur.cc, chunk_no & 1, j);
                        ++issued;
                    }
            }
            if (cur.tap == 8) {
                next_chunk(pcur);
                ++chunk_no;
            }
            // everything issued BEFORE this step (weights of s+1, older patch pieces) must have landed
            wait_vmcnt(issued);
            __builtin_amdgcn_s_barrier();
            step(cur);
        }
        return;
    }

    // =========================== CONSUMER ===========================
    const int r32 = lane & 31, hh = lane >> 5;
    const int cw = wave;                               // consumer index 0..3: pixels [64*cw, 64*cw+64)
    char *stg = smem + OFF_S + cw * SROWS * SSTRIDE;
    f32x16 acc[WC][2];

    int base_i[2];                                      // patch row of tap (0,0) for my two pixel tiles
    unsigned tapmask[2] = {0x1FFu, 0x1FFu};             // flat mode: bit t = tap t inside the image
    unsigned need_mask = 0;                             // wave-uniform: taps where some lane is masked
    int cur_item = -2;
    auto setup_item = [&](int item) {
        const int mt = item / p.n_tiles;
        if (mt == cur_item) return;
        cur_item = mt;
        need_mask = 0;
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
            const int i = cw * 64 + pt * 32 + r32;      // tile-local pixel
            if (p.tw) {
                const int ty = i / p.tw, tx = i - ty * p.tw;
                base_i[pt] = ty * RS + tx;
            } else {
                base_i[pt] = i;
                const long long m = (long long)mt * 256 + i;
                unsigned mk = 0;
                if (m < p.M) {
                    const int rem = (int)(m % ((long long)p.H * p.W));
                    const int y = rem / p.W, x = rem - y * p.W;
#pragma unroll
                    for (int t = 0; t < 9; ++t) {
                        const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
                        if ((unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W) mk |= 1u << t;
                    }
                }
                tapmask[pt] = mk;
                unsigned bad = ~mk & 0x1FFu;
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) bad |= __shfl_xor(bad, o, 64);
                need_mask |= bad;
            }
        }
        need_mask = __builtin_amdgcn_readfirstlane(need_mask);
    };

    const int wsw = (r32 >> 1) & 7;                     // swizzle of my weight rows (row = 32*ci + r32)

    __builtin_amdgcn_s_barrier();                       // prologue operands landed
    Cursor cur = first();
    int chunk_no = 0;
    for (int s = 0; cur.valid; ++s) {
        if (cur.cc == 0 && cur.tap == 0) {
            setup_item(cur.item);
#pragma unroll
            for (int ci = 0; ci < WC; ++ci)
#pragma unroll
                for (int pt = 0; pt < 2; ++pt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[ci][pt][e] = 0.f;
        }
        const char *wsb = smem + OFF_W + (s % 3) * WBYTES;
        const char *psb = smem + (chunk_no & 1) * kPatchBytes;
        const int tr = cur.tap / 3, tq = cur.tap - tr * 3;
        const int delta = tr * RS + tq;
        int prow[2], psw[2];
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
            prow[pt] = base_i[pt] + delta;
            psw[pt] = (prow[pt] >> 1) & 7;
        }
        const bool masked = (need_mask >> cur.tap) & 1u;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int lc = 2 * kk + hh;
            u32x4 wf[WC], pf[2];
#pragma unroll
            for (int ci = 0; ci < WC; ++ci) wf[ci] = *(const u32x4 *)(wsb + (ci * 32 + r32) * 128 + ((lc ^ wsw) << 4));
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) pf[pt] = *(const u32x4 *)(psb + prow[pt] * 128 + ((lc ^ psw[pt]) << 4));
            if (masked) {
#pragma unroll
                for (int pt = 0; pt < 2; ++pt)
                    if (!((tapmask[pt] >> cur.tap) & 1u)) pf[pt] = u32x4{0u, 0u, 0u, 0u};
            }
#pragma unroll
            for (int ci = 0; ci < WC; ++ci)
#pragma unroll
                for (int pt = 0; pt < 2; ++pt) MmaP<DT>::run(wf[ci], pf[pt], acc[ci][pt]);
        }
        __builtin_amdgcn_s_setprio(0);

        if (cur.tap == 8 && cur.cc == nchunks - 1) {
            // ---------------- epilogue of this item (wave-private) ----------------
            const int mt = cur.item / p.n_tiles, nt = cur.item - mt * p.n_tiles;
            const int n0 = nt * BN;
            // bias (+ReLU) in registers
#pragma unroll
            for (int ci = 0; ci < WC; ++ci)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 bv = *(const f32x4 *)(p.bias + n0 + ci * 32 + 8 * g + 4 * hh);
#pragma unroll
                    for (int pt = 0; pt < 2; ++pt)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            float v = acc[ci][pt][4 * g + j] + bv[j];
                            acc[ci][pt][4 * g + j] = p.relu ? fmaxf(v, 0.f) : v;
                        }
                }
            // global pixel index of tile-local pixel i (or -1)
            auto pixel_of = [&](int i) -> long long {
                if (p.tw) {
                    const int th = 256 / p.tw;
                    const int b = mt / p.tiles_per_img, tt = mt - b * p.tiles_per_img;
                    const int ty = tt / p.tiles_x, tx = tt - ty * p.tiles_x;
                    const int y = ty * th + i / p.tw, x = tx * p.tw + i % p.tw;
                    return ((long long)b * p.H + y) * p.W + x;
                }
                const long long m = (long long)mt * 256 + i;
                return m < p.M ? m : -1;
            };
            constexpr int CPR = BN * ES / 16;           // 16-B chunks per pixel
            if (p.out) {
#pragma unroll
                for (int pt = 0; pt < 2; ++pt) {
#pragma unroll
                    for (int rd = 0; rd < 32 / SROWS; ++rd) {
                        // lanes whose pixel is in this round write their 4-cout quads
                        if (r32 / SROWS == rd) {
#pragma unroll
                            for (int ci = 0; ci < WC; ++ci)
#pragma unroll
                                for (int g = 0; g < 4; ++g) {
                                    float q4[4] = {acc[ci][pt][4 * g], acc[ci][pt][4 * g + 1], acc[ci][pt][4 * g + 2], acc[ci][pt][4 * g + 3]};
                                    char *d = stg + (r32 % SROWS) * SSTRIDE + (ci * 32 + 8 * g + 4 * hh) * ES;
                                    if constexpr (ES == 4) {
                                        *(f32x4 *)d = f32x4{q4[0], q4[1], q4[2], q4[3]};
                                    } else {
                                        unsigned lo = (unsigned)from_f32<DT>(q4[0]).v | ((unsigned)from_f32<DT>(q4[1]).v << 16);
                                        unsigned hi = (unsigned)from_f32<DT>(q4[2]).v | ((unsigned)from_f32<DT>(q4[3]).v << 16);
                                        *(uint2 *)d = make_uint2(lo, hi);
                                    }
                                }
                        }
                        __builtin_amdgcn_s_waitcnt(0xC07F);     // lgkmcnt(0): my LDS writes are done
                        __builtin_amdgcn_wave_barrier();
                        for (int idx = lane; idx < SROWS * CPR; idx += 64) {
                            const int row = idx / CPR, ch = idx - row * CPR;
                            const long long gp = pixel_of(cw * 64 + pt * 32 + rd * SROWS + row);
                            const int c = n0 + ch * P16;
                            if (gp >= 0 && c < p.Cout)
                                *(u32x4 *)(p.out + ((size_t)gp * p.Cs + c) * ES) = *(const u32x4 *)(stg + row * SSTRIDE + ch * 16);
                        }
                        __builtin_amdgcn_s_waitcnt(0xC07F);
                        __builtin_amdgcn_wave_barrier();
                    }
                }
            }
            if (p.out_pool) {
                // fused MaxPool2d(2,2): the partner row is my other pixel tile (tw = 32) or lane^16
                // (tw = 16); the partner column is lane^1.  Even-x lanes of the top row hold the result.
                const int PW = p.W >> 1, PH = p.H >> 1;
#pragma unroll
                for (int pt = 0; pt < 2; ++pt) {
                    if (p.tw == 32 && pt == 1) break;
#pragma unroll
                    for (int ci = 0; ci < WC; ++ci)
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            float v = acc[ci][pt][e];
                            if (p.tw == 32) v = fmaxf(v, acc[ci][1][e]);
                            else v = fmaxf(v, __shfl_xor(v, 16, 64));
                            v = fmaxf(v, __shfl_xor(v, 1, 64));
                            acc[ci][pt][e] = v;
                        }
                    // pooled pixels of this pixel tile: tw=32 -> 16 (one row); tw=16 -> 8 (lanes 0-15, even x)
                    const int npool = p.tw == 32 ? 16 : 8;
                    const bool holder = (r32 & 1) == 0 && (p.tw == 32 || r32 < 16);
                    const int prow_l = p.tw == 32 ? (r32 >> 1) : (r32 >> 1);   // 0..15 / 0..7
                    for (int rd = 0; rd < (npool + SROWS - 1) / SROWS; ++rd) {
                        if (holder && prow_l / SROWS == rd) {
#pragma unroll
                            for (int ci = 0; ci < WC; ++ci)
#pragma unroll
                                for (int g = 0; g < 4; ++g) {
                                    float q4[4] = {acc[ci][pt][4 * g], acc[ci][pt][4 * g + 1], acc[ci][pt][4 * g + 2], acc[ci][pt][4 * g + 3]};
                                    char *d = stg + (prow_l % SROWS) * SSTRIDE + (ci * 32 + 8 * g + 4 * hh) * ES;
                                    if constexpr (ES == 4) {
                                        *(f32x4 *)d = f32x4{q4[0], q4[1], q4[2], q4[3]};
                                    } else {
                                        unsigned lo = (unsigned)from_f32<DT>(q4[0]).v | ((unsigned)from_f32<DT>(q4[1]).v << 16);
                                        unsigned hi = (unsigned)from_f32<DT>(q4[2]).v | ((unsigned)from_f32<DT>(q4[3]).v << 16);
                                        *(uint2 *)d = make_uint2(lo, hi);
                                    }
                                }
                        }
                        __builtin_amdgcn_s_waitcnt(0xC07F);
                        __builtin_amdgcn_wave_barrier();
                        const int rows_here = (npool - rd * SROWS) < SROWS ? (npool - rd * SROWS) : SROWS;
                        for (int idx = lane; idx < rows_here * CPR; idx += 64) {
                            const int row = idx / CPR, ch = idx - row * CPR;
                            const int pl = rd * SROWS + row;                    // pooled pixel within the tile row
                            // tile-local coordinates of the 2x2 window's top-left pixel
                            const int i0 = cw * 64 + pt * 32 + (p.tw == 32 ? 2 * pl : 2 * pl);
                            const int th = 256 / p.tw;
                            const int b = mt / p.tiles_per_img, tt = mt - b * p.tiles_per_img;
                            const int tyt = tt / p.tiles_x, txt = tt - tyt * p.tiles_x;
                            const int y = tyt * th + i0 / p.tw, x = txt * p.tw + i0 % p.tw;
                            const int c = n0 + ch * P16;
                            if (c < p.Cout)
                                *(u32x4 *)(p.out_pool + ((((size_t)b * PH + (y >> 1)) * PW + (x >> 1)) * p.Cs + c) * ES) =
                                    *(const u32x4 *)(stg + row * SSTRIDE + ch * 16);
                        }
                        __builtin_amdgcn_s_waitcnt(0xC07F);
                        __builtin_amdgcn_wave_barrier();
                    }
                }
            }
        }
        if (cur.tap == 8) ++chunk_no;
        __builtin_amdgcn_s_barrier();
        step(cur);
    }
}

// ---------------------------------------------------------------------------------------------
int patch_conv_supported(const ConvArgs &a)
{
    if (a.kh != 3 || a.kw != 3 || a.stride != 1 || a.pad != 1 || a.dil != 1) return 0;
    if (a.phases != 1 || a.out_f32 || a.res) return 0;
    if (a.Ho != a.H || a.Wo != a.W) return 0;
    if (a.Npad % 64) return 0;
    if (a.o_rs != (long long)a.Wo * a.o_cs || a.o_bs != (long long)a.Ho * a.Wo * a.o_cs || a.o_base) return 0;
    if ((long long)a.B * a.H * a.W * a.Cin * dtype_bytes(a.dtype) >= (1ll << 32)) return 0;
    if (a.W % 32 == 0 && a.H % 8 == 0) return 32;
    if (a.W % 16 == 0 && a.H % 16 == 0) return 16;
    if (2 * a.W + 2 + 256 <= kPatchSlots * 8) return -1;       // flat tiles
    return 0;
}

template <typename DT, int BN> static int launch_patch_cfg(const PatchParams &p, hipStream_t s)
{
    // a multiple of 8 workgroups (the item split is per XCD); surplus workgroups find no item and exit
    const int grid = p.items >= 256 ? 256 : ((p.items + 7) / 8) * 8;
    hipLaunchKernelGGL((conv3x3_patch_kernel<DT, BN>), dim3(grid), dim3(512), 0, s, p);
    return hip_status(hipGetLastError());
}

// out_pool: optional fused MaxPool2d(2,2) output; `a.out` may then be null (pooled output only).
int launch_conv3x3_patch(const ConvArgs &a, void *out_pool, hipStream_t s)
{
    const int mode = patch_conv_supported(a);
    if (!mode) return TDRN_E_UNSUPPORTED;
    if (out_pool && (mode < 0 || (a.H & 1) || (a.W & 1))) return TDRN_E_UNSUPPORTED;
    PatchParams p;
    p.in = (const char *)a.in; p.w = (const char *)a.w; p.zero = (const char *)a.zero_page; p.bias = a.bias;
    p.out = (char *)a.out; p.out_pool = (char *)out_pool;
    p.B = a.B; p.H = a.H; p.W = a.W; p.Cin = a.Cin; p.Cout = a.Cout; p.Cs = (int)a.o_cs; p.Ktot = 9 * a.Cin;
    p.relu = a.relu;
    p.tw = mode > 0 ? mode : 0;
    p.M = a.B * a.H * a.W;
    if (p.tw) {
        p.tiles_x = a.W / p.tw;
        p.tiles_per_img = p.tiles_x * (a.H / (256 / p.tw));
        p.m_tiles = a.B * p.tiles_per_img;
    } else {
        p.tiles_x = 0; p.tiles_per_img = 0;
        p.m_tiles = cdiv(p.M, 256);
    }
    const int BN = a.Npad % 128 == 0 ? 128 : 64;
    p.n_tiles = a.Npad / BN;
    p.items = p.m_tiles * p.n_tiles;
    if (p.items <= 0) return TDRN_OK;
#define LP(DT)                                                    \
    return BN == 128 ? launch_patch_cfg<DT, 128>(p, s) : launch_patch_cfg<DT, 64>(p, s)
    switch (a.dtype) {
        case TDRN_F32: LP(float);
        case TDRN_BF16: LP(bf16_t);
        case TDRN_F16: LP(f16_t);
    }
#undef LP
    return TDRN_E_ARG;
}

}  // namespace tdrn
