// conv3x3_patch.hip -- 3x3 / stride 1 / pad 1 convolution (the 13 VGG trunk convs and the TCB convs:
// 95 % of the network's FLOPs) as a warp-specialised, persistent direct convolution on MFMA.
//
// Why not the generic implicit GEMM (conv_igemm.hip) for these layers: measured there (rocprofv3 PMC +
// ablation builds, profiles/r01_*), the K loop is bound by LDS-DMA issue and L2->LDS traffic, not by
// the matrix pipe: every one of the 9 taps re-stages the same activations, and all waves run the
// same load->MFMA program in lockstep, so loads and MFMAs barely overlap (28 % MFMA utilisation).
//
// Structure (one 768-thread workgroup per CU, persistent over output tiles; per SIMD: 2 consumer
// waves + 1 loader wave):
//   * waves 8-11 = LOADERS.  Per 128-byte channel chunk they bring the input PATCH of the tile
//     (tile + 1-pixel halo, <= 352 rows of 128 B) into LDS once -- the 9 taps are then just shifted
//     row addresses into that image -- plus the [BN x 128 B] weight slice of every (chunk, tap) step
//     into a 3-slot ring, two steps ahead.  All by global_load_lds_dwordx4 with the bank swizzle on
//     the source address; out-of-image rows read a zero page.  Waits are counted vmcnt(N).
//   * waves 0-7 = CONSUMERS.  ds_read_b128 + v_mfma only: each owns 64 pixels x BN/2 couts
//     (2 x BN/64 accumulator tiles), operand fragments double-buffered in registers and the last
//     K-slice of a step multiplied AFTER the step's barrier (covers the barrier and the first reads
//     of the next step); weights as the A operand so a lane holds 4 consecutive couts
//     of one pixel.  Epilogue per wave through a private LDS strip -> whole-line NHWC stores, with
//     bias, ReLU and (optionally) the following 2x2 max-pool fused in registers.
//   * one s_barrier per (chunk, tap) step; loaders run ahead across tile boundaries, so the next
//     tile's first operands arrive while the consumers are still in the previous tile's epilogue.
//
// Two tile geometries share the code: "2-D" tiles of 8x32 or 16x16 output pixels of one image (the
// loader zero-fills the halo outside the image), and "flat" tiles of 256 consecutive NHW pixels for
// narrow maps (W <= 43: 40x40, 20x20), where the per-lane tap-validity mask zeroes fragments whose
// tap falls outside the image.
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "kernels.h"

// diagnostics build switch (never set in the product build): 1 = loaders issue nothing, 2 = consumers
// skip ds_read + MFMA, 4 = skip the epilogue stores.  Compile-time on purpose: a runtime flag splits
// the K loop into basic blocks and hipcc then waits lgkmcnt(0/1) where lgkmcnt(4) would do.
#ifndef TDRN_PATCH_PRIO
#define TDRN_PATCH_PRIO 4     // 0: s_setprio 1 around every MFMA group (measured 10 % SLOWER: per-segment flips hurt), 1: none,
                              // 2/3: static consumer priorities (= none), 4: loaders at static priority 2 (another -3 %), 5: 3
#endif
#ifndef TDRN_PATCH_ABLATE
#define TDRN_PATCH_ABLATE 0
#endif

namespace tdrn {

struct PatchParams {
    const char *in, *w, *zero;
    const float *bias;
    char *out, *out_pool;          // NHWC [B][H][W][Cs] and optional pooled [B][H/2][W/2][Cs]
    int B, H, W, Cin, Cout, Cs, Ktot;
    int relu;
    int tw, lgtw;                  // 32 / 16: 2-D tiles of (256/tw) x tw ; 0: flat tiles
    int tiles_x, tiles_per_img;    // 2-D mode
    int m_tiles, n_tiles, items;   // items = m_tiles * n_tiles
    int M;                         // B*H*W
    int max_wgs;                   // > 0: cap on the persistent grid
    int tail_split;                // 16-bit 128-cout kernels: an XCD's last, at most half-filled round of items runs as 64-cout half items
    // FUSE instantiation only: the layer's input is the first conv's output (3 -> 64 channels, 3x3, pad 1, stride 1, BN folded,
    // ReLU), computed here from the raw frames instead of being read back from HBM
    const float *fx, *fw, *fb;     // frames NCHW fp32 [B][3][S][S]; first-conv weights [64][27] (k = c*9 + r*3 + q) and bias [64], fp32
    int fS, fCout;                 // frame size (= H = W of this layer) and the first conv's real channel count
    int ablate;                    // diagnostics (TDRN_CONV_ABLATE): 1 = loaders issue nothing, 2 = consumers skip ds_read+MFMA
#ifdef TDRN_PATCH_STAMP
    unsigned *stamps;              // diagnostics build only: [workgroup][wave][8] cycle sums (s_memtime), see the launcher
#endif
#ifdef TDRN_PATCH_WGTIME
    unsigned long long *wgt;       // diagnostics build only (make EXTRA=-DTDRN_PATCH_WGTIME): [workgroup][2] s_memrealtime at its first / last instruction
#endif
};

// In-kernel cycle stamps of a diagnostics build (make EXTRA=-DTDRN_PATCH_STAMP; never in the product): where do the
// loader and the consumer waves spend a step?  s_memtime returns through lgkmcnt, so the consumers stamp only where
// they drain it anyway (before the step's barrier) and right behind the barrier / the last K slice.
#ifdef TDRN_PATCH_STAMP
#define STAMP_DECL unsigned long long st_t = __builtin_amdgcn_s_memtime(); unsigned st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define STAMP(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[k] += (unsigned)(t_ - st_t); st_t = t_; } while (0)
#define STAMP_FLUSH do { if (lane == 0) for (int k_ = 0; k_ < 8; ++k_) p.stamps[((size_t)blockIdx.x * 12 + wave) * 8 + k_] = st_acc[k_]; } while (0)
#else
#define STAMP_DECL
#define STAMP(k) do { } while (0)
#define STAMP_FLUSH do { } while (0)
#endif

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
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
    }
}

#ifndef TDRN_PATCH_RING
#define TDRN_PATCH_RING 3
#endif
constexpr int kRing = TDRN_PATCH_RING;          // weight ring depth for the 128-cout kernels: weights go kRing-1 steps ahead
constexpr int kPatchSlots = kRing == 4 ? 43 : 44;   // 8-row LDS-DMA pieces per patch buffer (344 / 352 rows)
constexpr int kSlotsPerLoader = (kPatchSlots + 3) / 4;

}  // namespace

// TW = 32 / 16: 2-D tiles of (256/TW) x TW pixels of one image; TW = 0: flat tiles.  Compile-time: with a runtime
// tile width every per-element step of the pooled epilogue carried a branch and the index math its shifts as variables.
// FUSE (16-bit types, BN = 64, TW = 32, one cout tile): the loader waves COMPUTE the patch of the next item -- the first conv
// (layers.hip, first_conv_mfma_kernel: same operand layout, same instruction, bit-identical values) on the raw frame's
// 12 x 36 halo tile -- instead of reading the first conv's 420-MB output back: that tensor never exists.
template <typename DT, int BN, int TW, bool FUSE = false>
__global__ __launch_bounds__(768) void conv3x3_patch_kernel(const PatchParams p)
{
    static_assert(!FUSE || (BN == 64 && TW == 32 && sizeof(DT) == 2), "fused first conv: 16-bit, 64 couts, 8x32 tiles");
    constexpr bool FLAT = TW == 0;
    constexpr int LGTW = TW == 32 ? 5 : 4;              // (2-D tiles only)
    constexpr int TPX = FUSE ? 512 : 256;               // pixels per work item (FUSE: 16 x 32, see below)
    constexpr int TH = TW ? TPX / TW : 0;
    constexpr int PXG = TPX / 64;                       // pixel groups of 64 (one per consumer wave and cout group)
    constexpr int CG = 8 / PXG;                         // cout groups
    constexpr int ES = elem_traits<DT>::bytes;
    constexpr int P16 = elem_traits<DT>::per16;
    constexpr int CK = 128 / ES;
    constexpr int BNH = BN / CG;                        // couts per consumer wave (8 consumers = PXG pixel groups x CG cout groups)
    constexpr int WC = BNH / 32;                        // cout tiles per consumer
    constexpr int WBYTES = BN * 128;                    // one weight slot
    constexpr int WL = BN / 32;                         // weight LDS-DMA pieces per loader wave per step
    constexpr int RING = BN == 128 ? kRing : 3;         // (the 64-cout kernels have LDS to spare but gain nothing)
    constexpr int SROWS = (ES == 2 ? 16 : 8) / ((RING == 4 && BN == 128) ? 2 : 1);   // pixels per epilogue round (per wave)
    constexpr int SSTRIDE = BNH * ES + 16;              // staging row stride (bytes)
    constexpr int NPB = FUSE ? 1 : 2;                   // patch buffers (FUSE: one; the next patch is computed under the epilogue)
    constexpr int PSLOTS = FUSE ? ((TH + 2) * (TW + 2) + 7) / 8 : kPatchSlots;   // 8-row pieces per patch buffer
    constexpr int PBYTES = PSLOTS * 1024;
    constexpr int OFF_W = NPB * PBYTES;
    constexpr int OFF_S = OFF_W + RING * WBYTES;
    constexpr int OFF_B = OFF_S + 8 * SROWS * SSTRIDE;  // bias of the next item (one 1-KiB LDS-DMA piece)
    constexpr int RAWR = TH + 4, RAWC = 36;             // FUSE: fp32 halo tile of the frame: 3 planes x (TH + 4) rows x (TW + 4) columns
    constexpr int RAWN = 3 * RAWR * RAWC;
    constexpr int RAWP = (RAWN + 63) / 64, RAWJ = (RAWP + 3) / 4;      // 256-byte LDS-DMA pieces, and pieces per loader wave
    constexpr int FTILES = ((TH + 2) * (TW + 2) + 31) / 32, FTJ = (FTILES + 3) / 4;   // 32-pixel slices of the patch, per loader wave
    constexpr int RAWB = ((RAWN + 63) / 64) * 256;      // ... in whole 256-byte LDS-DMA pieces
    constexpr int OFF_R = OFF_B + 1024;                 // two raw tiles + the first conv's bias
    constexpr int LDS = FUSE ? OFF_R + 2 * RAWB + 256 : OFF_B + 1024;
    static_assert(LDS <= 160 * 1024, "LDS budget");
    __shared__ __attribute__((aligned(16))) char smem[LDS];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nchunks = p.Cin / CK;
#ifdef TDRN_PATCH_WGTIME
    if (threadIdx.x == 0) p.wgt[2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
#endif

    // ---- work distribution: each XCD (blockIdx % 8) owns a contiguous range of items so that cout
    // siblings of a pixel tile and neighbouring tiles share its L2.  This workgroup runs items
    // item0, item0 + istride, ... (n_it of them); the step sequence is (item, chunk, tap).
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int per_xcd = (p.items + 7) >> 3, istride = ((int)gridDim.x + 7) >> 3;
    int avail = p.items - xcd * per_xcd;
    avail = avail < per_xcd ? avail : per_xcd;
    int n_full = avail > slot ? (avail - slot + istride - 1) / istride : 0;
    const int item0 = xcd * per_xcd + slot;
    // ---- tail split (round 6).  1600 items on 256 workgroups are 6.25 rounds: after six rounds an XCD has 8 items left for its 32
    // workgroups and the launch ends with 24 of every 32 CUs idle for a whole item (per-workgroup stamps, profiles/r05_experiments.md:
    // 7-13 % of the chip time of conv2_1 / 2_2 / 3_1 / 3_3).  When an XCD's last round is at most HALF filled, its items are cut in two
    // along the couts -- 2 * rem HALF ITEMS of 256 pixels x 64 couts, one per workgroup -- and a half item runs on the same eight
    // consumer waves with one 32-cout accumulator tile each (the BN = 64 kernel's wave tile).  Every output element still sees the
    // same MFMA instruction with the same operand rows in the same K order: bit-identical whatever the batch (= the item count) does
    // to the cut (tests/test_gpu_pin16.py, TDRN_PATCH_TAIL=0 keeps whole items).  A half item costs ~0.6 of a whole one (half the MFMAs,
    // the same patch loads and barriers), so the last round shrinks from 1 to ~0.6 item times.
    [[maybe_unused]] int n_tail = 0, tail_enc = 0;       // tail_enc = 2 * item + cout half (one scalar: the kernel is at its SGPR budget)
    constexpr bool TAILOK = BN == 128 && sizeof(DT) == 2 && !FUSE;
    if constexpr (TAILOK) {
        if (p.tail_split && avail > 0) {
            const int full = avail / istride, rem = avail - full * istride;
            if (full > 0 && rem > 0 && 2 * rem <= istride) {
                n_full = full;
                if (slot < 2 * rem) {
                    n_tail = 1;
                    tail_enc = 2 * (xcd * per_xcd + full * istride) + slot;
                }
            }
        }
    }
    const int n_it = n_full + n_tail;
    const int n_steps = n_it * nchunks * 9;
    // item `it` of this workgroup: pixel tile, first cout, whole (BN couts) or half (64 couts) item
    auto item_mt = [&](int it) { return (it < n_full ? item0 + it * istride : tail_enc >> 1) / p.n_tiles; };
    auto item_c0 = [&](int it) {
        const int item = it < n_full ? item0 + it * istride : tail_enc >> 1;
        return (item % p.n_tiles) * BN + (it < n_full ? 0 : (tail_enc & 1) * 64);
    };
    const int RS = TW ? TW + 2 : p.W;                  // patch row stride of one image row

    if (wave >= 8) {
        // =========================== LOADER ===========================
        // the loaders' few instructions per step gate everyone's barrier: let them issue ahead of the consumers
        if (TDRN_PATCH_PRIO == 4) __builtin_amdgcn_s_setprio(2);
        if (TDRN_PATCH_PRIO == 5) __builtin_amdgcn_s_setprio(3);
        const int lw = wave - 8;
        const int lrow = lane >> 3, pc = lane & 7;
        unsigned poff[kSlotsPerLoader];                 // byte offset of my 16 B in the tensor, or ~0u
        int pyx[kSlotsPerLoader];                       // (py << 16 | px) of my patch row, -1 = beyond the patch
        int plc[kSlotsPerLoader];                       // swizzled 16-B chunk offset inside the 128-B row
        {
#pragma unroll
            for (int j = 0; j < kSlotsPerLoader; ++j) {
                const int pr = (lw + 4 * j) * 8 + lrow;
                plc[j] = (pc ^ ((pr >> 1) & 7)) << 4;
                if (TW) {
                    const int py = pr / RS, px = pr - py * RS;
                    pyx[j] = py < TH + 2 ? ((py << 16) | px) : -1;
                } else {
                    pyx[j] = pr < 256 + 2 * p.W + 2 ? pr : -1;
                }
            }
        }
        // ---- FUSE: raw-tile loads and the first conv on the matrix cores ----------------------------------------------
        const int f_r32 = lane & 31, f_hh = lane >> 5;
        [[maybe_unused]] int r_rq[FUSE ? RAWJ : 1], r_off[FUSE ? RAWJ : 1];         // my elements of a raw tile: (row << 16 | col) or -1, offset in the frame
        [[maybe_unused]] int koff1[16];                 // LDS offset of my k = (c, dy, dx), relative to a pixel's halo origin
        [[maybe_unused]] u32x4 wq1[2][2];               // packed first-conv weights [cout tile][k step]
        if constexpr (FUSE) {
#pragma unroll
            for (int j = 0; j < RAWJ; ++j) {
                const int i = (lw + 4 * j) * 64 + lane;
                const int c = i / (RAWR * RAWC), rem = i - c * (RAWR * RAWC);
                const int r = rem / RAWC, q = rem - r * RAWC;
                r_rq[j] = i < RAWN ? ((r << 16) | q) : -1;
                r_off[j] = (c * p.fS + r) * p.fS + q;
            }
#pragma unroll
            for (int s2 = 0; s2 < 16; ++s2) {
                const int k = 16 * (s2 >> 3) + 8 * f_hh + (s2 & 7);
                const int c = k / 9, r = (k - 9 * c) / 3, q = k - 9 * c - 3 * r;
                koff1[s2] = k < 27 ? (c * RAWR + r) * RAWC + q : 0;      // k >= 27 pads K: its weight is 0
            }
#pragma unroll
            for (int ci = 0; ci < 2; ++ci)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const int co = ci * 32 + f_r32, k0 = 16 * ks + 8 * f_hh + 2 * jj;
                        const float a = (k0 < 27 && co < p.fCout) ? p.fw[co * 27 + k0] : 0.f;
                        const float b2 = (k0 + 1 < 27 && co < p.fCout) ? p.fw[co * 27 + k0 + 1] : 0.f;
                        wq1[ci][ks][jj] = pack2<DT>(a, b2);
                    }
            if (lw == 0) ((float *)(smem + OFF_R + 2 * RAWB))[lane] = lane < p.fCout ? p.fb[lane] : 0.f;
        }
        auto tile_origin = [&](int item, int &b, int &y0, int &x0) {
            const int mt = item / p.n_tiles;
            b = mt / p.tiles_per_img;
            const int tt = mt - b * p.tiles_per_img, ty = tt / p.tiles_x, tx = tt - ty * p.tiles_x;
            y0 = ty * TH; x0 = tx * TW;
        };
        auto load_raw = [&](int item, int buf) -> int {      // my pieces of the item's raw halo tile; returns how many were issued
            int n = 0;
            if constexpr (FUSE) {
                int b, y0, x0;
                tile_origin(item, b, y0, x0);
                const int iy0 = y0 - 2, ix0 = x0 - 2;
                const float *xb = p.fx + (size_t)b * 3 * p.fS * p.fS + ((long long)iy0 * p.fS + ix0);
                char *dst = smem + OFF_R + buf * RAWB;
#pragma unroll
                for (int j = 0; j < RAWJ; ++j) {
                    if ((lw + 4 * j) * 64 >= RAWN) continue;                       // (wave-uniform)
                    const int yy = iy0 + (r_rq[j] >> 16), xx = ix0 + (r_rq[j] & 0xffff);
                    const bool ok = r_rq[j] >= 0 && (unsigned)yy < (unsigned)p.fS && (unsigned)xx < (unsigned)p.fS;
                    const char *src = ok ? (const char *)(xb + r_off[j]) : p.zero + (lane & 15) * 4;
                    if constexpr (!(TDRN_PATCH_ABLATE & 1))
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                         (__attribute__((address_space(3))) void *)(dst + (lw + 4 * j) * 256), 4, 0, 0);
                    ++n;
                }
            }
            return n;
        };
        // 32 patch pixels (flat index tt*32 + lane%32 over the (TH+2) x (TW+2) patch) x 64 channels -> patch buffer rows
        [[maybe_unused]] int f_y0 = 0, f_x0 = 0;          // tile origin of the item whose patch is being computed (set once per item)
        auto first_conv_tile = [&](int tt, int rawb, char *dstbuf) {
            if constexpr (FUSE) {
                const int y0 = f_y0, x0 = f_x0;
                const int pq = tt * 32 + f_r32;
                const bool valid = pq < (TH + 2) * RS;
                const int py = pq / RS, px = pq - py * RS;
                // outside the frame the NEXT conv pads with zeros (not with the first conv evaluated out there)
                const bool inimg = valid && (unsigned)(y0 - 1 + py) < (unsigned)p.fS && (unsigned)(x0 - 1 + px) < (unsigned)p.fS;
                const float *raw = (const float *)(smem + OFF_R + rawb * RAWB);
                const float *b1 = (const float *)(smem + OFF_R + 2 * RAWB);
                const int porg = valid ? py * RAWC + px : 0;
                float xv[16];
#pragma unroll
                for (int s2 = 0; s2 < 16; ++s2) xv[s2] = raw[porg + koff1[s2]];
                f32x16 a1[2];
#pragma unroll
                for (int ci = 0; ci < 2; ++ci)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 bv = *(const f32x4 *)(b1 + ci * 32 + 8 * g + 4 * f_hh);
#pragma unroll
                        for (int j = 0; j < 4; ++j) a1[ci][4 * g + j] = bv[j];
                    }
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    u32x4 xq;
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) xq[jj] = pack2<DT>(xv[8 * ks + 2 * jj], xv[8 * ks + 2 * jj + 1]);
#pragma unroll
                    for (int ci = 0; ci < 2; ++ci) MmaP<DT>::run(wq1[ci][ks], xq, a1[ci]);
                }
                if (valid) {
                    char *row = dstbuf + pq * 128 + 8 * f_hh;
                    const int sw = (pq >> 1) & 7;
#pragma unroll
                    for (int ci = 0; ci < 2; ++ci)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            float q4[4];
#pragma unroll
                            for (int j = 0; j < 4; ++j) q4[j] = inimg ? fmaxf(a1[ci][4 * g + j], 0.f) : 0.f;
                            *(uint2 *)(row + (((4 * ci + g) ^ sw) << 4)) = make_uint2(pack2<DT>(q4[0], q4[1]), pack2<DT>(q4[2], q4[3]));
                        }
                }
            }
        };
        int table_mt = -1;
        auto patch_table = [&](int mt) {                // per pixel tile: where my patch rows live
            if (mt == table_mt) return;
            table_mt = mt;
            const long long rowbytes = (long long)p.Cin * ES;
            if (TW) {
                const int b = mt / p.tiles_per_img, tt = mt - b * p.tiles_per_img;
                const int ty = tt / p.tiles_x, tx = tt - ty * p.tiles_x;
                const int y0 = ty * TH - 1, x0 = tx * TW - 1;
#pragma unroll
                for (int j = 0; j < kSlotsPerLoader; ++j) {
                    const int y = y0 + (pyx[j] >> 16), x = x0 + (pyx[j] & 0xffff);
                    const bool ok = pyx[j] >= 0 && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
                    poff[j] = ok ? (unsigned)((((long long)b * p.H + y) * p.W + x) * rowbytes + plc[j]) : 0xFFFFFFFFu;
                }
            } else {
                const long long j0 = (long long)mt * 256 - p.W - 1;
#pragma unroll
                for (int j = 0; j < kSlotsPerLoader; ++j) {
                    const long long pix = j0 + pyx[j];
                    const bool ok = pyx[j] >= 0 && pix >= 0 && pix < p.M;
                    poff[j] = ok ? (unsigned)(pix * rowbytes + plc[j]) : 0xFFFFFFFFu;
                }
            }
        };
        unsigned woff[WL];                              // weight row part of the source offset
#pragma unroll
        for (int k = 0; k < WL; ++k) {
            const int n = (lw + 4 * k) * 8 + lrow;
            woff[k] = (unsigned)((size_t)n * p.Ktot * ES + ((pc ^ ((n >> 1) & 7)) << 4));
        }
        constexpr bool live = !(TDRN_PATCH_ABLATE & 1);
        // weight stream state: step index ws, (w_it, w_cc, w_tap), source = wbase + wk
        int ws = 0, w_it = 0, w_cc = 0, w_tap = 0, wslot = 0;
        const char *wbase = p.w;
        unsigned wk = 0;
        int w_pieces = WL;                              // weight pieces per step of the item being streamed (half items: WL / 2)
        auto weight_item = [&]() {
            wbase = p.w + (size_t)item_c0(w_it) * p.Ktot * ES;
            if constexpr (TAILOK) w_pieces = w_it < n_full ? WL : WL / 2;
        };
        constexpr bool wfixed = (TDRN_PATCH_ABLATE & 16) != 0, pfixed = (TDRN_PATCH_ABLATE & 32) != 0;
        auto load_weights = [&]() -> int {              // issue step ws into ring slot wslot, then advance; returns the pieces issued
            const int np = w_pieces;
            if (live) {
                char *dst = smem + OFF_W + wslot * WBYTES;
#pragma unroll
                for (int k = 0; k < WL; ++k)
                    if (!TAILOK || k < np) glds(wfixed ? p.w + (woff[k] & 0xffffu) : wbase + wk + woff[k], dst + (lw + 4 * k) * 1024);
            }
            ++ws;
            wslot = wslot == RING - 1 ? 0 : wslot + 1;
            wk += (unsigned)(p.Cin * ES);
            if (++w_tap == 9) {
                w_tap = 0;
                if (++w_cc == nchunks) {
                    w_cc = 0;
                    ++w_it;
                    if (w_it < n_it) weight_item();
                }
                wk = (unsigned)(w_cc * 128);
            }
            return np;
        };
        auto load_patch = [&](int j, unsigned ccoff, char *dstbuf) {
            if (!live || lw + 4 * j >= kPatchSlots || FUSE) return;
            const unsigned o = poff[j];
            glds(o == 0xFFFFFFFFu ? p.zero : (pfixed ? p.in + (o & 0xfffffu) : p.in + (size_t)o + ccoff), dstbuf + (lw + 4 * j) * 1024);
        };
        // patch stream state: the chunk being PREFETCHED: (p_it, p_cc), buffer pbuf
        int p_it = 0, p_cc = 0, pbuf = 0;
        auto next_patch_chunk = [&]() {
            if (NPB == 2) pbuf ^= 1;
            if (++p_cc == nchunks) {
                p_cc = 0;
                ++p_it;
            }
        };

        if constexpr (FUSE) {
            if (n_it > 0) load_raw(item0, 0);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();               // every loader wave's share of the raw tiles (and the bias) is in LDS
            if (n_it > 0) {
                int b_;
                tile_origin(item0, b_, f_y0, f_x0);
#pragma unroll
                for (int q = 0; q < FTJ; ++q)
                    if (lw + 4 * q < FTILES) first_conv_tile(lw + 4 * q, 0, smem);
            }
        }
        if (n_it > 0) {
            weight_item();
            if constexpr (!FUSE) {
                patch_table(item_mt(0));
#pragma unroll
                for (int j = 0; j < kSlotsPerLoader; ++j) load_patch(j, 0u, smem);
            }
#pragma unroll
            for (int k = 0; k < RING - 1; ++k)
                if (k < n_steps) load_weights();
            next_patch_chunk();
            if (lw == 0 && live)
                glds(lane < (n_full > 0 ? BN : 64) / 4 ? (const char *)(p.bias + item_c0(0)) + lane * 16 : p.zero, smem + OFF_B);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();

        int tap = 0, c_it = 0, c_cc = 0, carried = 0;
        STAMP_DECL
        const int last_piece = lw + 4 * (kSlotsPerLoader - 1) < kPatchSlots ? 1 : 0;
        for (int g = 0; g < n_steps; ++g) {
            // operands of step g are in LDS.  Issue the weights of step g+2 and this tap's share of the
            // NEXT chunk's patch (taps 0-4: two pieces, tap 5: one); then make sure everything issued
            // before this step has landed, and release the step.
            int issued = 0;
            if (ws < n_steps) issued = load_weights();
            if constexpr (FUSE) {
                // tap 0: the raw tile of the NEXT item (its patch is computed behind this item's last barrier, below)
                if (tap == 0 && c_it + 1 < n_it) issued += load_raw(item0 + (c_it + 1) * istride, (c_it + 1) & 1);
            } else if (p_it < n_it && tap < 6) {
                char *dstbuf = smem + pbuf * PBYTES;
                const unsigned ccoff = (unsigned)(p_cc * 128);
                switch (tap) {
                    case 0:
                        patch_table(item_mt(p_it));
                        load_patch(0, ccoff, dstbuf); load_patch(1, ccoff, dstbuf); issued += 2; break;
                    case 1: load_patch(2, ccoff, dstbuf); load_patch(3, ccoff, dstbuf); issued += 2; break;
                    case 2: load_patch(4, ccoff, dstbuf); load_patch(5, ccoff, dstbuf); issued += 2; break;
                    case 3: load_patch(6, ccoff, dstbuf); load_patch(7, ccoff, dstbuf); issued += 2; break;
                    case 4: load_patch(8, ccoff, dstbuf); load_patch(9, ccoff, dstbuf); issued += 2; break;
                    default: load_patch(10, ccoff, dstbuf); issued += last_piece; break;
                }
            }
            if (lw == 0 && c_cc == nchunks - 1 && tap == 6 && c_it + 1 < n_it) {
                // the NEXT item's bias -> the other LDS bias slot (the consumers initialise their
                // accumulators from it when that item starts; this item reads slot c_it & 1)
                if (live) glds(lane < (c_it + 1 < n_full ? BN : 64) / 4 ? (const char *)(p.bias + item_c0(c_it + 1)) + lane * 16 : p.zero, smem + OFF_B);
                issued += 1;
            }
            // ring of 3: everything issued before this step has landed; ring of 4: before the previous step
            STAMP(0);                                       // issue
            wait_vmcnt(live ? issued + (RING == 4 ? carried : 0) : 0);
            if constexpr (FUSE) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // my patch rows are written
            STAMP(1);                                       // landing of the previous step's pieces
            carried = issued;
            __builtin_amdgcn_s_barrier();
            STAMP(2);                                       // barrier (waiting for the consumers)
            if (++tap == 9) {
                tap = 0;
                next_patch_chunk();
                if (++c_cc == nchunks) {
                    if constexpr (FUSE) {
                        // The item is finished: its patch is dead and the consumers are busy with their epilogue (no MFMA, no
                        // patch reads): compute the NEXT item's patch into the same buffer now, then meet the consumers at the
                        // extra barrier that releases their first reads of it.
                        if (c_it + 1 < n_it) {
                            int b_;
                            tile_origin(item0 + (c_it + 1) * istride, b_, f_y0, f_x0);
#pragma unroll
                            for (int q = 0; q < FTJ; ++q)
                                if (lw + 4 * q < FTILES) first_conv_tile(lw + 4 * q, (c_it + 1) & 1, smem);
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                            __builtin_amdgcn_s_barrier();
                        }
                    }
                    c_cc = 0;
                    ++c_it;
                }
            }
        }
        STAMP_FLUSH;
        return;
    }

    // =========================== CONSUMER ===========================
    const int r32 = lane & 31, hh = lane >> 5;
    const int cw = wave % PXG;                         // pixel group: pixels [64*cw, 64*cw+64)
    const int chalf = wave / PXG;                      // cout group: couts [BNH*chalf, BNH*chalf + BNH) of the tile
    char *stg = smem + OFF_S + wave * SROWS * SSTRIDE;
    f32x16 acc[WC][2];

    int base_i[2];                                      // patch row of tap (0,0) for my two pixel tiles
    unsigned tapmask[2] = {0x1FFu, 0x1FFu};             // flat mode: bit t = tap t inside the image
    unsigned need_mask = 0;                             // wave-uniform: taps where some lane is masked
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
        const int i = cw * 64 + pt * 32 + r32;          // tile-local pixel
        base_i[pt] = TW ? (i >> LGTW) * RS + (i & (TW - 1)) : i;
    }
    const int wsw = (r32 >> 1) & 7;                     // swizzle of my weight rows (row = 32*ci + r32)
    constexpr int CPR = BNH * ES / 16;                  // 16-B chunks per pixel (my cout half)
    constexpr int RPI = 64 / CPR;                       // staging rows copied per wave pass
    const int my_ch = lane % CPR, my_row = lane / CPR;
    constexpr bool compute = !(TDRN_PATCH_ABLATE & 2);

    // ---- per-item state ---------------------------------------------------------------------------
    int cur_mt = -1, n0 = 0;
    long long tile_pix0 = 0;                            // 2-D: global pixel of the tile's (0,0); flat: mt*256
    int tile_row0 = 0, tile_x0 = 0;                     // 2-D: b*H + y and x of the tile's (0,0)
    auto setup_item = [&](int it) {
        const int mt = item_mt(it);
        n0 = item_c0(it);
        if (mt == cur_mt) return;
        cur_mt = mt;
        if (TW) {
            const int b = mt / p.tiles_per_img, tt = mt - b * p.tiles_per_img;
            const int ty = tt / p.tiles_x, tx = tt - ty * p.tiles_x;
            tile_row0 = b * p.H + ty * TH;
            tile_x0 = tx * TW;
            tile_pix0 = (long long)tile_row0 * p.W + tile_x0;
            return;
        }
        tile_pix0 = (long long)mt * 256;
        need_mask = 0;
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
            const long long m = tile_pix0 + cw * 64 + pt * 32 + r32;
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
        need_mask = __builtin_amdgcn_readfirstlane(need_mask);
    };

    // ---- epilogue of one item, 16-bit types: straight from the registers (round 5) ---------------------------------------------------
    // A lane holds, per pixel fragment, WC x 4 groups (ci, g) of 4 couts: group c = 4 ci + g is the 16-byte chunk c of the pixel's span of
    // BNH couts, of which lane r32 (hh = 0) has the first 8 bytes and lane r32 + 32 the second.  v_permlane32_swap on a PAIR of chunks
    // (c, c + 1) gives the lower lane all 16 bytes of chunk c and the upper lane all of chunk c + 1 (cdna_hip_programming.md T21): one
    // dwordx4 store per pair, 32 contiguous bytes per pixel and instruction -- no staging strip, no LDS round trips, no wave barriers.
    // Same values as the staged version (bias is in the accumulators; ReLU is a signed 16-bit maximum with 0 on the PACKED pair, exact for
    // finite values; the conversion is the same v_cvt_pk).  Pooled output: the window's four (TW = 16) or two (TW = 32: the partner row is
    // my other pixel fragment) lanes hold the same maxima after the lane exchanges (DPP quad permute for lane ^ 1, v_permlane16_swap for
    // lane ^ 16) and each stores a different chunk pair of the same pooled pixel.
    typedef short pk_s2 __attribute__((ext_vector_type(2)));
    [[maybe_unused]] const pk_s2 relu_lo = p.relu ? pk_s2{0, 0} : pk_s2{(short)-32768, (short)-32768};
    auto pack_relu16 = [&](float a, float b) -> unsigned {
        if constexpr (sizeof(DT) == 2) return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(pk_s2, pack2<DT>(a, b)), relu_lo));
        else return 0u;
    };
    auto epilogue16 = [&](auto wcn_tag) {
        if constexpr (sizeof(DT) == 2) {
            constexpr int WCN = decltype(wcn_tag)::value;       // accumulator tiles per wave along the couts: WC, or 1 in a half item
            const int cbase = n0 + chalf * (32 * WCN);          // first cout of my span
            constexpr int NPAIR = 2 * WCN;                      // chunk pairs per pixel (2 or 4)
            if (p.out && !(TDRN_PATCH_ABLATE & 4)) {
#pragma unroll
                for (int pt = 0; pt < 2; ++pt) {
                    const int i = cw * 64 + pt * 32 + r32;
                    long long gp;
                    if (TW) gp = tile_pix0 + (long long)(i >> LGTW) * p.W + (i & (TW - 1));
                    else { gp = tile_pix0 + i; if (gp >= p.M) gp = -1; }
                    char *row = p.out + ((size_t)(gp < 0 ? 0 : gp) * p.Cs + cbase) * 2;
#pragma unroll
                    for (int pr = 0; pr < NPAIR; ++pr) {
                        const int ci = pr >> 1, g0 = 2 * (pr & 1);
                        const f32x16 &t = acc[ci][pt];
                        const unsigned ax = pack_relu16(t[4 * g0], t[4 * g0 + 1]), ay = pack_relu16(t[4 * g0 + 2], t[4 * g0 + 3]);
                        const unsigned bx = pack_relu16(t[4 * g0 + 4], t[4 * g0 + 5]), by = pack_relu16(t[4 * g0 + 6], t[4 * g0 + 7]);
                        auto rx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
                        auto ry = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
                        if (gp >= 0 && cbase + (2 * pr + hh) * 8 < p.Cout)
                            *(u32x4 *)(row + (2 * pr + hh) * 16) = u32x4{rx[0], ry[0], rx[1], ry[1]};
                    }
                }
            }
            if constexpr (TW != 0) if (p.out_pool) {
                // fused MaxPool2d(2,2) on the RAW accumulators (max commutes with the monotonic bias + ReLU + rounding applied afterwards)
                const int PW = p.W >> 1;
#pragma unroll
                for (int pt = 0; pt < 2; ++pt) {
                    if (TW == 32 && pt == 1) break;
                    uint2 pk[WCN][4];
#pragma unroll
                    for (int ci = 0; ci < WCN; ++ci)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            float m[4];
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                float v = acc[ci][pt][4 * g + j];
                                if (TW == 32) {
                                    v = fmaxf(v, acc[ci][1][4 * g + j]);
                                } else {
                                    // lane ^ 16: rows of 16 lanes; v_permlane16_swap exchanges the odd rows of its first operand with the even
                                    // rows of its second: from two copies of v it leaves (lower row | it again) and (upper row | it again), whose
                                    // maximum is the row pair's.  Inline asm on two registers of their own (through the builtin, given the same value
                                    // twice, hipcc dropped the second result: the pooled maxima of conv3_3 came out wrong in 29 % of the outputs);
                                    // the two wait states a VALU write needs before the swap reads it are inside the string.
                                    float va = v, vb = v;
                                    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(va), "+v"(vb));
                                    v = fmaxf(va, vb);
                                }
                                v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)));
                                m[j] = v;
                            }
                            pk[ci][g] = make_uint2(pack_relu16(m[0], m[1]), pack_relu16(m[2], m[3]));
                        }
                    // my pixel's window -> pooled pixel; the window's lanes (2 for TW = 32, 4 for TW = 16) take different chunk pairs
                    const int i = cw * 64 + pt * 32 + r32;
                    const long long gpool = (long long)((tile_row0 + (i >> LGTW)) >> 1) * PW + ((tile_x0 + (i & (TW - 1))) >> 1);
                    char *row = p.out_pool + ((size_t)gpool * p.Cs + cbase) * 2;
                    constexpr int NDUP = TW == 32 ? 2 : 4;      // lanes holding the same window
                    const int sel = TW == 32 ? (r32 & 1) : ((r32 & 1) | ((r32 >> 3) & 2));
#pragma unroll
                    for (int rd = 0; rd < (NPAIR + NDUP - 1) / NDUP; ++rd) {
                        // my chunk pair of this round: rd * NDUP + sel (a runtime index into registers: selected with compares)
                        uint2 a = pk[0][0], b = pk[0][1];
#pragma unroll
                        for (int d = 0; d < NDUP; ++d) {
                            const int pr = rd * NDUP + d;
                            if (pr < NPAIR) {
                                const bool mine = sel == d;
                                a.x = mine ? pk[pr >> 1][2 * (pr & 1)].x : a.x; a.y = mine ? pk[pr >> 1][2 * (pr & 1)].y : a.y;
                                b.x = mine ? pk[pr >> 1][2 * (pr & 1) + 1].x : b.x; b.y = mine ? pk[pr >> 1][2 * (pr & 1) + 1].y : b.y;
                            }
                        }
                        const int mypr = rd * NDUP + sel;
                        auto rx = __builtin_amdgcn_permlane32_swap(a.x, b.x, false, false);
                        auto ry = __builtin_amdgcn_permlane32_swap(a.y, b.y, false, false);
                        if (mypr < NPAIR && cbase + (2 * mypr + hh) * 8 < p.Cout)
                            *(u32x4 *)(row + (2 * mypr + hh) * 16) = u32x4{rx[0], ry[0], rx[1], ry[1]};
                    }
                }
            }
        }
    };

    // ---- epilogue of one item, fp32 (wave-private staging strip -> whole-line stores) -------------------
    auto epilogue = [&](auto wcn_tag) {
        if constexpr (sizeof(DT) == 2) { epilogue16(wcn_tag); return; }
        const int my_c = n0 + chalf * BNH + my_ch * P16;
        auto pixel_of = [&](int i) -> long long {       // global pixel of tile-local pixel i (or -1)
            if (TW) return tile_pix0 + (long long)(i >> LGTW) * p.W + (i & (TW - 1));
            const long long m = tile_pix0 + i;
            return m < p.M ? m : -1;
        };
        // one 4-cout quad of one accumulator tile -> (ReLU, convert) -> staging row (bias is already in)
        auto stage_quad = [&](const f32x16 &t, int ci, int g, int srow) {
            float q4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) q4[j] = p.relu ? fmaxf(t[4 * g + j], 0.f) : t[4 * g + j];
            char *d = stg + srow * SSTRIDE + (ci * 32 + 8 * g + 4 * hh) * ES;
            if constexpr (ES == 4) {
                *(f32x4 *)d = f32x4{q4[0], q4[1], q4[2], q4[3]};
            } else if constexpr (sizeof(DT) == 2 && __is_same(DT, bf16_t)) {
                typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
                typedef float fl2 __attribute__((ext_vector_type(2)));
                const bf2 lo = __builtin_convertvector(fl2{q4[0], q4[1]}, bf2), hi = __builtin_convertvector(fl2{q4[2], q4[3]}, bf2);
                *(uint2 *)d = make_uint2(__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi));
            } else {
                typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                typedef float fl2 __attribute__((ext_vector_type(2)));
                const h2 lo = __builtin_convertvector(fl2{q4[0], q4[1]}, h2), hi = __builtin_convertvector(fl2{q4[2], q4[3]}, h2);
                *(uint2 *)d = make_uint2(__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi));
            }
        };
        if (p.out && !(TDRN_PATCH_ABLATE & 4)) {
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) {
#pragma unroll 1
                for (int rd = 0; rd < 32 / SROWS; ++rd) {
                    if (r32 / SROWS == rd) {            // lanes whose pixel is in this round
#pragma unroll
                        for (int ci = 0; ci < WC; ++ci)
#pragma unroll
                            for (int g = 0; g < 4; ++g) stage_quad(acc[ci][pt], ci, g, r32 % SROWS);
                    }
                    __builtin_amdgcn_s_waitcnt(0xC07F);     // lgkmcnt(0): my LDS writes are done
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int k = 0; k < SROWS / RPI; ++k) {
                        const int row = my_row + k * RPI;
                        const long long gp = pixel_of(cw * 64 + pt * 32 + rd * SROWS + row);
                        if (gp >= 0 && my_c < p.Cout)
                            *(u32x4 *)(p.out + ((size_t)gp * p.Cs + my_c) * ES) = *(const u32x4 *)(stg + row * SSTRIDE + my_ch * 16);
                    }
                    __builtin_amdgcn_wave_barrier();     // (a wave's DS instructions execute in order)
                }
            }
        }
        if (p.out_pool) {
            // fused MaxPool2d(2,2) on the RAW accumulators (max commutes with the monotonic bias+ReLU
            // applied at staging): the partner row is my other pixel tile (tw = 32) or lane^16
            // (tw = 16); the partner column is lane^1.  Even-x lanes of the top row hold the result.
            const int PW = p.W >> 1;
            constexpr int npool = TW == 32 ? 16 : 8;    // pooled pixels per pixel tile
            const bool holder = (r32 & 1) == 0 && (TW == 32 || r32 < 16);
            const int prow_l = r32 >> 1;
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) {
                if (TW == 32 && pt == 1) break;
#pragma unroll
                for (int ci = 0; ci < WC; ++ci)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        float v = acc[ci][pt][e];
                        if (TW == 32) v = fmaxf(v, acc[ci][1][e]);
                        else v = fmaxf(v, __shfl_xor(v, 16, 64));
                        v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)));     // lane ^ 1: a DPP quad permute, not an LDS round trip
                        acc[ci][pt][e] = v;
                    }
#pragma unroll 1
                for (int rd = 0; rd < (npool + SROWS - 1) / SROWS; ++rd) {
                    if (holder && prow_l / SROWS == rd) {
#pragma unroll
                        for (int ci = 0; ci < WC; ++ci)
#pragma unroll
                            for (int g = 0; g < 4; ++g) stage_quad(acc[ci][pt], ci, g, prow_l % SROWS);
                    }
                    __builtin_amdgcn_s_waitcnt(0xC07F);
                    __builtin_amdgcn_wave_barrier();
                    const int rows_here = (npool - rd * SROWS) < SROWS ? (npool - rd * SROWS) : SROWS;
#pragma unroll
                    for (int k = 0; k < SROWS / RPI; ++k) {
                        const int row = my_row + k * RPI;
                        if (row < rows_here && my_c < p.Cout) {
                            const int pl = rd * SROWS + row;            // pooled pixel within this pixel tile
                            const int i0 = cw * 64 + pt * 32 + 2 * pl;  // top-left pixel of the 2x2 window
                            // window (b*H + y, x) with y, x even (H is even)  ->  pooled index ((b*H + y)/2)*PW + x/2
                            const long long gpool = (long long)((tile_row0 + (i0 >> LGTW)) >> 1) * PW + ((tile_x0 + (i0 & (TW - 1))) >> 1);
                            *(u32x4 *)(p.out_pool + ((size_t)gpool * p.Cs + my_c) * ES) = *(const u32x4 *)(stg + row * SSTRIDE + my_ch * 16);
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
    };

    // ---- operand fragments: explicit double buffering (A/B sets) so that the reads of K-slice kk+1
    // are in flight during the MFMAs of kk; the step's LAST slice is multiplied after the barrier,
    // covering the barrier and the first reads of the next step.
    u32x4 wfA[WC], pfA[2], wfB[WC], pfB[2];
    const char *wsb = smem + OFF_W;
    const char *psb = smem;
    int prow[2] = {base_i[0], base_i[1]};
    int psw[2] = {(base_i[0] >> 1) & 7, (base_i[1] >> 1) & 7};
    auto load_frags = [&](auto wcn_tag, u32x4 *wf, u32x4 *pf, int kk) {
        constexpr int WCN = decltype(wcn_tag)::value;
        if (!compute) return;
        const int lc = 2 * kk + hh;
#pragma unroll
        for (int ci = 0; ci < WCN; ++ci) wf[ci] = *(const u32x4 *)(wsb + (chalf * (32 * WCN) + ci * 32 + r32) * 128 + ((lc ^ wsw) << 4));
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) pf[pt] = *(const u32x4 *)(psb + prow[pt] * 128 + ((lc ^ psw[pt]) << 4));
    };
    // m0/m1: all-ones, or zero for lanes whose tap falls outside the image (flat tiles only; branch-free)
    auto mma_frags = [&](auto wcn_tag, const u32x4 *wf, u32x4 *pf, unsigned m0, unsigned m1) {
        constexpr int WCN = decltype(wcn_tag)::value;
        if (!compute) return;
        if constexpr (FLAT) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                pf[0][j] &= m0;
                pf[1][j] &= m1;
            }
        }
#pragma unroll
        for (int ci = 0; ci < WCN; ++ci)
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) MmaP<DT>::run(wf[ci], pf[pt], acc[ci][pt]);
    };
    // accumulators start at the bias (staged into LDS by loader wave 0 one item ahead)
    auto init_acc = [&](auto wcn_tag) {
        constexpr int WCN = decltype(wcn_tag)::value;
        const char *bsrc = smem + OFF_B;
#pragma unroll
        for (int ci = 0; ci < WCN; ++ci)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 bv = *(const f32x4 *)(bsrc + (chalf * (32 * WCN) + ci * 32 + 8 * g + 4 * hh) * 4);
#pragma unroll
                for (int pt = 0; pt < 2; ++pt)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[ci][pt][4 * g + j] = bv[j];
            }
    };

    if constexpr (FUSE) __builtin_amdgcn_s_barrier();   // (the loaders' raw tiles)
    __builtin_amdgcn_s_barrier();                       // prologue operands landed
    if (TDRN_PATCH_PRIO == 2 && wave >= 4) __builtin_amdgcn_s_setprio(1);     // static: the younger half
    if (TDRN_PATCH_PRIO == 3) __builtin_amdgcn_s_setprio(1);                  // static: all consumers over the loaders
    typedef std::integral_constant<int, WC> full_t;      // a whole item: WC accumulator tiles per wave along the couts
    typedef std::integral_constant<int, 1> half_t;       // a half item of the tail split: one
    if (n_it > 0) {
        setup_item(0);
        init_acc(full_t{});
        load_frags(full_t{}, wfA, pfA, 0);
    }
    int it = 0, cc = 0, tap = 0, tq = 0, delta = 0, wslot = 0, pbuf = 0;
    STAMP_DECL
    // one (chunk, tap) step of the current item; `wcn` says how many accumulator tiles the item has per wave
    auto step = [&](auto wcn) {
        unsigned m0 = 0xFFFFFFFFu, m1 = 0xFFFFFFFFu;
        if constexpr (FLAT) {
            m0 = ((tapmask[0] >> tap) & 1u) ? 0xFFFFFFFFu : 0u;
            m1 = ((tapmask[1] >> tap) & 1u) ? 0xFFFFFFFFu : 0u;
        }
        if (TDRN_PATCH_PRIO == 0) __builtin_amdgcn_s_setprio(1);
        load_frags(wcn, wfB, pfB, 1);
        __builtin_amdgcn_sched_barrier(0);
        mma_frags(wcn, wfA, pfA, m0, m1);
        __builtin_amdgcn_sched_barrier(0);
        load_frags(wcn, wfA, pfA, 2);
        __builtin_amdgcn_sched_barrier(0);
        mma_frags(wcn, wfB, pfB, m0, m1);
        __builtin_amdgcn_sched_barrier(0);
        load_frags(wcn, wfB, pfB, 3);
        __builtin_amdgcn_sched_barrier(0);
        mma_frags(wcn, wfA, pfA, m0, m1);
        if (TDRN_PATCH_PRIO == 0) __builtin_amdgcn_s_setprio(0);
        // every LDS read of this step has returned -> the step's buffers may be refilled after the barrier
        __builtin_amdgcn_s_waitcnt(0xC07F);
        STAMP(0);                                           // K slices 0-2: reads + 12 MFMAs + drain
        __builtin_amdgcn_s_barrier();
        STAMP(1);                                           // barrier (waiting for the loaders / the other consumers)
        // ---- advance to the next step and start its first reads, THEN finish this step's last K-slice ----
        // (measured and rejected: the cursor and the read addresses computed before the barrier, in the shadow of slice 2's
        // MFMAs, so that only the four reads stand between the barrier and slice 3: -7 % on the family)
        const bool item_done = tap == 8 && cc == nchunks - 1;
        wslot = wslot == RING - 1 ? 0 : wslot + 1;
        ++tap;
        if (++tq == 3) {
            tq = 0;
            delta += RS;
        }
        if (tap == 9) {
            tap = 0; tq = 0; delta = 0;
            pbuf ^= 1;
            if (++cc == nchunks) cc = 0;
        }
        wsb = smem + OFF_W + wslot * WBYTES;
        psb = smem + (NPB == 2 ? pbuf : 0) * PBYTES;
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
            prow[pt] = base_i[pt] + delta + tq;
            psw[pt] = (prow[pt] >> 1) & 7;
        }
        // (FUSE: behind an item's last barrier the single patch buffer is being rewritten: its first reads wait for the
        // extra barrier after the epilogue)
        if (!(FUSE && item_done)) load_frags(wcn, wfA, pfA, 0);    // (past the last step this reads stale but in-bounds LDS; unused)
        __builtin_amdgcn_sched_barrier(0);
        if (TDRN_PATCH_PRIO == 0) __builtin_amdgcn_s_setprio(1);
        mma_frags(wcn, wfB, pfB, m0, m1);
        if (TDRN_PATCH_PRIO == 0) __builtin_amdgcn_s_setprio(0);
        STAMP(2);                                           // K slice 3 under the next step's first reads
        if (item_done) {
            epilogue(wcn);
            ++it;
            if (it < n_it) {
                setup_item(it);
                if (TAILOK && it >= n_full) {
                    // the next item is this workgroup's half item: its bias is 64 floats, its weight rows sit at 32 * chalf of the
                    // slot, and the first fragments fetched above used a whole item's rows: fetch them again
                    init_acc(half_t{});
                    load_frags(half_t{}, wfA, pfA, 0);
                } else {
                    init_acc(full_t{});
                }
                if constexpr (FUSE) {
                    __builtin_amdgcn_s_barrier();           // the loaders have written the next item's patch
                    load_frags(wcn, wfA, pfA, 0);
                }
            }
            STAMP(3);                                       // epilogue + next item's set-up
        }
    };
    const int n_steps_full = n_full * nchunks * 9;
#pragma unroll 1
    for (int g = 0; g < n_steps_full; ++g) step(full_t{});
    if constexpr (TAILOK) {
#pragma unroll 1
        for (int g = n_steps_full; g < n_steps; ++g) step(half_t{});
    }
    STAMP_FLUSH;
#ifdef TDRN_PATCH_WGTIME
    if (threadIdx.x == 0) p.wgt[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
#endif
}

// ---------------------------------------------------------------------------------------------
int patch_conv_supported(const ConvArgs &a)
{
    if (a.kdisable & 4) return 0;
    if (a.kh != 3 || a.kw != 3 || a.stride != 1 || a.pad != 1 || a.dil != 1) return 0;
    if (a.phases != 1 || a.out_f32 || a.res) return 0;
    if (a.Ho != a.H || a.Wo != a.W) return 0;
    if (a.Npad % 64) return 0;
    if (a.o_rs != (long long)a.Wo * a.o_cs || a.o_bs != (long long)a.Ho * a.Wo * a.o_cs || a.o_base) return 0;
    // (32-bit byte offsets into the input tensor; the fused-first-conv instantiation never reads it -- its input is the raw frame)
    if (!a.fuse_x && (long long)a.B * a.H * a.W * a.Cin * dtype_bytes(a.dtype) >= (1ll << 32)) return 0;
    if (a.W % 32 == 0 && a.H % 8 == 0) return 32;
    if (a.W % 16 == 0 && a.H % 16 == 0) return 16;
    if (2 * a.W + 2 + 256 <= kPatchSlots * 8) return -1;       // flat tiles
    return 0;
}

template <typename DT, int BN> static int launch_patch_cfg(const PatchParams &p_in, hipStream_t s)
{
    PatchParams p = p_in;
    // a multiple of 8 workgroups (the item split is per XCD); surplus workgroups find no item and exit
    int grid = p.items >= 256 ? 256 : ((p.items + 7) / 8) * 8;
    if (p.max_wgs > 0 && grid > p.max_wgs) grid = p.max_wgs;
#ifdef TDRN_PATCH_WGTIME
    // diagnostics build: when did every workgroup of launch number TDRN_WGTIME_CALL (and the 17 after it) start and end?  (s_memrealtime,
    // 100 MHz; never inside a stream capture: the report synchronises)
    static unsigned long long *wgt = nullptr;
    static long calls = 0, want = -1;
    if (!wgt) TDRN_HIP_TRY(hipMalloc((void **)&wgt, 256 * 2 * sizeof(unsigned long long)));
    if (want < 0) { const char *e = getenv("TDRN_WGTIME_CALL"); want = e ? atol(e) : 2000; }
    p.wgt = wgt;
    const bool report = calls >= want && calls < want + 18;
    ++calls;
    if (report) TDRN_HIP_TRY(hipMemsetAsync(wgt, 0, 256 * 2 * sizeof(unsigned long long), s));
    struct Report {
        bool on; hipStream_t s; int grid; const PatchParams &p; unsigned long long *d;
        ~Report() {
            if (!on) return;
            (void)hipStreamSynchronize(s);
            unsigned long long h[512];
            (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
            unsigned long long t0 = ~0ull, t1 = 0;
            int n = 0;
            for (int i = 0; i < grid; ++i) if (h[2 * i]) { t0 = h[2 * i] < t0 ? h[2 * i] : t0; t1 = h[2 * i + 1] > t1 ? h[2 * i + 1] : t1; ++n; }
            double late = 0, busy = 0, worst = 0, e_min = 1e18; int n5 = 0, n20 = 0;
            for (int i = 0; i < grid; ++i) if (h[2 * i]) {
                const double d0 = (h[2 * i] - t0) * 0.01, b = (h[2 * i + 1] - h[2 * i]) * 0.01, e = (h[2 * i + 1] - t0) * 0.01;
                late += d0; busy += b; worst = d0 > worst ? d0 : worst; e_min = e < e_min ? e : e_min; n5 += d0 > 5; n20 += d0 > 20;
            }
            fprintf(stderr, "wgtime %dx%d Cin %d Cout %d items %d grid %d: span %.1f us, workgroup busy mean %.1f us, first end at %.1f us; start later than the first: mean %.1f us, max %.1f us, > 5 us: %d, > 20 us: %d of %d\n",
                    p.H, p.W, p.Cin, p.Cout, p.items, grid, (t1 - t0) * 0.01, busy / n, e_min, late / n, worst, n5, n20, n);
        }
    } rep{report, s, grid, p, wgt};
#endif
    if constexpr (BN == 64 && sizeof(DT) == 2) {
        if (p.fx) {
            if (p.tw != 32 || p.n_tiles != 1 || p.H % 16) return TDRN_E_UNSUPPORTED;
            hipLaunchKernelGGL((conv3x3_patch_kernel<DT, 64, 32, true>), dim3(grid), dim3(768), 0, s, p);
            return hip_status(hipGetLastError());
        }
    }
    if (p.fx) return TDRN_E_UNSUPPORTED;
    if (p.tw == 0) hipLaunchKernelGGL((conv3x3_patch_kernel<DT, BN, 0>), dim3(grid), dim3(768), 0, s, p);
    else if (p.tw == 32) hipLaunchKernelGGL((conv3x3_patch_kernel<DT, BN, 32>), dim3(grid), dim3(768), 0, s, p);
    else hipLaunchKernelGGL((conv3x3_patch_kernel<DT, BN, 16>), dim3(grid), dim3(768), 0, s, p);
    return hip_status(hipGetLastError());
}

// out_pool: optional fused MaxPool2d(2,2) output; `a.out` may then be null (pooled output only).
int launch_conv3x3_patch(const ConvArgs &a, void *out_pool, hipStream_t s)
{
    const int mode = patch_conv_supported(a);
    if (!mode) return TDRN_E_UNSUPPORTED;
    if (out_pool && (mode < 0 || (a.H & 1) || (a.W & 1))) return TDRN_E_UNSUPPORTED;
    if (ws_conv_supported(a)) {                           // conv3x3_ws.hip (Cin == 64): same arithmetic, same bits
        const int rc = launch_conv3x3_ws(a, out_pool, s);
        if (rc != TDRN_E_UNSUPPORTED) return rc;         // (it declines launches too small to fill the chip and fused launches it has no LDS for)
    }
    // (pooled layers stay here: conv3x3_pp.hip's POOL instantiation no longer spills with the register-only epilogue and is bit-identical --
    // tests/test_gpu_pin16.py ran green on it -- but conv3_3 measured 210 us there against 204-210 us here: no gain; TDRN_PP_POOL=1 selects it)
    static int pp_pool = -1;
    if (pp_pool < 0) { const char *e = getenv("TDRN_PP_POOL"); pp_pool = e ? atoi(e) : 0; }
    if ((!out_pool || pp_pool) && pp_conv_supported(a)) {    // conv3x3_pp.hip: same arithmetic, same bits
        const int rc = launch_conv3x3_pp(a, out_pool, s);
        if (rc != TDRN_E_UNSUPPORTED) return rc;         // (it declines launches too small to fill the chip)
    }
    PatchParams p;
    p.in = (const char *)a.in; p.w = (const char *)a.w; p.zero = (const char *)a.zero_page; p.bias = a.bias;
    p.out = (char *)a.out; p.out_pool = (char *)out_pool;
    p.B = a.B; p.H = a.H; p.W = a.W; p.Cin = a.Cin; p.Cout = a.Cout; p.Cs = (int)a.o_cs; p.Ktot = 9 * a.Cin;
    p.relu = a.relu;
    p.tw = mode > 0 ? mode : 0;
    p.lgtw = mode == 32 ? 5 : (mode == 16 ? 4 : 0);
    p.M = a.B * a.H * a.W;
    if (p.tw) {
        p.tiles_x = a.W / p.tw;
        p.tiles_per_img = p.tiles_x * (a.H / ((a.fuse_x ? 512 : 256) / p.tw));      // the fused first-conv variant works on 16 x 32 tiles
        p.m_tiles = a.B * p.tiles_per_img;
    } else {
        p.tiles_x = 0; p.tiles_per_img = 0;
        p.m_tiles = cdiv(p.M, 256);
    }
    // 128-cout items unless they would leave most CUs idle (20x20 maps at small batch): 64-cout items double
    // the item count for the same per-output arithmetic (K order unchanged, results identical)
    static int small_bn = -1;
    if (small_bn < 0) { const char *e = getenv("TDRN_PATCH_SMALL_BN"); small_bn = e ? atoi(e) : 1; }
    int BN = a.Npad % 128 == 0 ? 128 : 64;
    if (small_bn && BN == 128 && p.m_tiles * (a.Npad / 128) < 160) BN = 64;
    p.n_tiles = a.Npad / BN;
    p.items = p.m_tiles * p.n_tiles;
    static int ablate = -1;
    if (ablate < 0) ablate = dev_ablate_env("TDRN_CONV_ABLATE");     // (developer builds only: common.h)
    p.ablate = ablate;
    p.fx = a.fuse_x; p.fw = a.fuse_w; p.fb = a.fuse_b; p.fS = a.H; p.fCout = a.fuse_cout;
    p.max_wgs = a.max_wgs > 0 ? (a.max_wgs / 8) * 8 : 0;
    static int tail = -1;
    if (tail < 0) { const char *e = getenv("TDRN_PATCH_TAIL"); tail = e ? atoi(e) : 1; }
    p.tail_split = tail && !(a.kdisable & 1024);     // (TDRN_PLAN_NO_PATCH_TAIL)
    if (p.items <= 0) return TDRN_OK;
#ifdef TDRN_PATCH_STAMP
    // diagnostics build: synchronise after every launch and print the mean cycles per wave in each state
    static unsigned *stamps = nullptr;
    if (!stamps) TDRN_HIP_TRY(hipMalloc((void **)&stamps, 256 * 12 * 8 * sizeof(unsigned)));
    TDRN_HIP_TRY(hipMemsetAsync(stamps, 0, 256 * 12 * 8 * sizeof(unsigned), s));
    p.stamps = stamps;
    struct Report {
        const PatchParams &p; hipStream_t s; int BN, nchunks;
        ~Report()
        {
            static unsigned host[256 * 12 * 8];
            (void)hipStreamSynchronize(s);
            (void)hipMemcpy(host, p.stamps, sizeof(host), hipMemcpyDeviceToHost);
            double c[4] = {0, 0, 0, 0}, l[3] = {0, 0, 0};
            int nc = 0, nl = 0;
            for (int b = 0; b < 256; ++b)
                for (int w = 0; w < 12; ++w) {
                    const unsigned *v = host + (b * 12 + w) * 8;
                    if (v[0] + v[1] + v[2] == 0) continue;
                    if (w < 8) { for (int k = 0; k < 4; ++k) c[k] += v[k]; ++nc; }
                    else { for (int k = 0; k < 3; ++k) l[k] += v[k]; ++nl; }
                }
            if (!nc || !nl) return;
            const double steps = (double)((p.items + 255) / 256) * nchunks * 9;
            fprintf(stderr, "patch_stamp H%d W%d Cin%d Cout%d BN%d tw%d items%d steps/CU~%.0f | consumer cyc/wave: slices0-2 %.0f barrier %.0f slice3 %.0f epilogue %.0f"
                            " | loader cyc/wave: issue %.0f vmcnt %.0f barrier %.0f\n",
                    p.H, p.W, p.Cin, p.Cout, BN, p.tw, p.items, steps, c[0] / nc, c[1] / nc, c[2] / nc, c[3] / nc, l[0] / nl, l[1] / nl, l[2] / nl);
        }
    } report{p, s, BN, (int)(p.Cin * dtype_bytes(a.dtype) / 128)};
#endif
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
