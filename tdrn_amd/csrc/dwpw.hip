// dwpw.hip -- the conv_dw block of the MobileNet trunks (model/networks.py:736-745: depthwise 3x3 + BN + ReLU, pointwise 1x1 + BN
// + ReLU) as ONE launch for the 16-bit plans: the depthwise output never leaves the chip.
//
// Unfused (round 3) a block is dwconv3_strip (reads the map, writes it again: 210 MB at 512 ch x 40x40 x 64 frames) followed by a
// 1x1 GEMM on conv_igemm that reads it back -- 75 + 105 us for 54 GFLOP, neither roof in sight (profiles/r04_cfg4).  Here a
// persistent workgroup of eight waves owns items of 256 pixels x 256 couts (the wave tile of conv3x3_pp.hip: 64 px x 128 couts,
// 128 accumulator registers) and walks the input channels in 64-channel chunks:
//     patch   the 256 pixels + 1-pixel halo of the chunk, staged by LDS-DMA exactly as in conv3x3_pp.hip (2 buffers, swizzled
//             128-byte rows, zero padding materialised in LDS for 2-D tiles, per-pixel tap masks for flat tiles), one chunk ahead;
//     dw      all 512 threads compute the depthwise conv of the chunk from the patch -- thread = (8-channel column, 4 adjacent
//             pixels), fp32 FMAs in the reference kernel's order (bias first, taps row-major), ReLU, ONE rounding to the 16-bit type
//             -- and write the [256 px][64 ch] operand tile into LDS;
//     pw      32 MFMAs per wave on that tile and the chunk's [256 couts][64 ch] weight slice (LDS-DMA, issued before the dw phase);
//     out     accumulators + bias, ReLU, rounding, whole-line NHWC stores through a wave-private LDS strip.
// The arithmetic of every output element is that of the two-launch plan -- same dw FMA order, same rounding of the intermediate,
// same K order of the MFMAs (chunks ascending, four 16-channel slices each), bias added after the accumulation as conv_igemm.hip
// does -- so the fused plan is BIT-IDENTICAL to the unfused one (tests/test_gpu_net.py::test_fused_dwpw_equals_two_launches) and the
// exact-input stage tests of the unfused plan (tests/test_gpu_pin16.py) pin it.  A pixel tile's depthwise conv is recomputed once
// per 256-cout tile (Cout = 512: twice): its operand tile for ALL chunks (256 px x 512 ch) does not fit LDS beside the
// accumulators of a second cout tile.
// Scope: stride 1, Cin % 64 == 0, Cout % 256 == 0, the tile shapes of patch_conv_supported (80x80 and up as 2-D tiles, 40x40 /
// 20x20 as flat tiles): 8 of the 13 trunk blocks at 320 px (blocks 4, 5, 7-11, 13: 85 % of the trunk's FLOPs); stride-2 blocks and
// the narrow first three stay on the two-launch path.
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "kernels.h"

namespace tdrn {

struct DwPwParams {
    const char *in, *w;            // NHWC input [B][H][W][Cin]; pointwise weights [Npad][Cin] (DT)
    const char *wdw;               // depthwise weights fp32 [9][Cin]; its bias fp32 [Cin] sits bdw_off bytes behind
    unsigned bdw_off;
    const float *bias;             // pointwise bias fp32 [Npad]
    char *out;                     // NHWC [B][H][W][Cs]
    int B, H, W, Cin, Cout, Cs;
    int relu_dw, relu;
    int tiles_x, tiles_per_img, m_tiles, n_tiles, items, M;
};

namespace {

template <typename DT> struct MmaDP;
template <> struct MmaDP<bf16_t> {
    __device__ static __forceinline__ void run(const u32x4 &a, const u32x4 &b, f32x16 &c)
    { c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(i16x8, a), __builtin_bit_cast(i16x8, b), c, 0, 0, 0); }
};
template <> struct MmaDP<f16_t> {
    __device__ static __forceinline__ void run(const u32x4 &a, const u32x4 &b, f32x16 &c)
    { c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0); }
};

// (see conv3x3_pp.hip: inline asm so that hipcc's waitcnt pass neither sees nor drains the LDS-DMA queue)
__device__ __forceinline__ void dp_glds16(const char *sbase, unsigned voff, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory");
}
__device__ __forceinline__ unsigned dp_lds_addr(const void *p)
{
    return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const char *)p;
}
#define DP_BAR()                                  \
    do {                                          \
        __builtin_amdgcn_sched_barrier(0);        \
        __builtin_amdgcn_s_barrier();             \
        __builtin_amdgcn_sched_barrier(0);        \
    } while (0)
#define DP_LGKM0() __builtin_amdgcn_s_waitcnt(0xC07F)

constexpr int kDPSlots = 44;                      // 8-row LDS-DMA pieces per patch buffer (352 rows)
constexpr int kDPPieces = (kDPSlots + 7) / 8;     // patch pieces per wave and chunk

typedef float f32x2 __attribute__((ext_vector_type(2)));

}  // namespace

// TW = 32 / 16: 2-D tiles of (256/TW) x TW pixels of one image; TW = 0: flat tiles of 256 consecutive NHW pixels.
template <typename DT, int TW>
__global__ __launch_bounds__(512, 2) void dwpw_kernel(const DwPwParams p)
{
    static_assert(sizeof(DT) == 2, "16-bit element types only");
    constexpr bool FLAT = TW == 0;
    constexpr int LGTW = TW == 32 ? 5 : 4;
    constexpr int TH = TW ? 256 / TW : 0;
    constexpr int ES = 2, P16 = 8;
    constexpr int BN = 256, BNH = 128, WC = 4;
    constexpr int PBYTES = kDPSlots * 1024;             // 44 KiB
    constexpr int OFF_A = 2 * PBYTES;                   // the depthwise output of the chunk: [256 px][128 B], swizzled like a patch row
    constexpr int OFF_W = OFF_A + 256 * 128;            // the chunk's pointwise weights: [256 couts][128 B]
    constexpr int OFF_DW = OFF_W + BN * 128;            // 2 x 3 KiB: depthwise weights of a chunk, rows of 64 fp32: taps 0..8, bias, (2 unused)
    constexpr int OFF_Z = OFF_DW + 2 * 3072;
    constexpr int LDS = OFF_Z + 128;
    constexpr int SROWS = 16;
    constexpr int SSTRIDE = BNH * ES + 16;
    constexpr int STRIP = SROWS * SSTRIDE;
    static_assert(8 * STRIP <= PBYTES, "staging strips fit the dead patch buffer");
    static_assert(LDS <= 160 * 1024, "LDS budget");
    __shared__ __attribute__((aligned(16))) char smem[LDS];

    const int lane = threadIdx.x & 63;
    auto opaque_lane = [&]() -> int {                   // (see conv3x3_pp.hip: keeps hipcc from hoisting -- and spilling -- lane constants)
        int ln = lane;
        asm volatile("" : "+v"(ln));
        return ln;
    };
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wave >> 2;                          // cout half
    const int cw = wave & 3;                            // pixel quarter: pixels [64*cw, 64*cw + 64)
    const int nchunks = p.Cin / 64;

    // ---- work distribution: every XCD label (blockIdx % 8) owns a contiguous range of items; cout-tile-major numbering
    // (item = nt * m_tiles + mt): an XCD's range needs one cout tile's weights
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int per_xcd = (p.items + 7) >> 3, istride = ((int)gridDim.x + 7) >> 3;
    int avail = p.items - xcd * per_xcd;
    avail = avail < per_xcd ? avail : per_xcd;
    avail = avail < 0 ? 0 : avail;
    const int n_items = avail > slot ? (avail - slot + istride - 1) / istride : 0;
    if (n_items == 0) return;                           // (whole workgroup)
    const int item0 = xcd * per_xcd + slot;
    const int RS = TW ? TW + 2 : p.W;                   // patch row stride of one image row

    // =========================== patch staging (as conv3x3_pp.hip) ===========================
    int pt_b = 0, pt_y0 = 0, pt_x0 = 0, pt_j0 = 0;
    auto mt_of = [&](int item) -> int { return item % p.m_tiles; };
    auto nt_of = [&](int item) -> int { return item / p.m_tiles; };
    auto patch_tile = [&](int item) {
        const int mt = mt_of(item);
        if (TW) {
            const int b = mt / p.tiles_per_img, tt = mt - b * p.tiles_per_img;
            const int ty = tt / p.tiles_x, tx = tt - ty * p.tiles_x;
            pt_b = __builtin_amdgcn_readfirstlane(b); pt_y0 = __builtin_amdgcn_readfirstlane(ty * TH - 1); pt_x0 = __builtin_amdgcn_readfirstlane(tx * TW - 1);
        } else {
            pt_j0 = __builtin_amdgcn_readfirstlane(mt * 256 - p.W - 1);
        }
    };
    const unsigned smem_lds = __builtin_amdgcn_readfirstlane(dp_lds_addr(smem));
    int pp_yx = 0;
    // one 8-row piece of the patch: rows outside the image are zeroed by an LDS store of the lanes concerned, the DMA runs with
    // those lanes off; returns 1 when a DMA was issued (the caller counts them for its vmcnt)
    auto patch_piece = [&](int j, unsigned ccoff, int dstbuf_off) -> int {
        const int q = wave + 8 * j;
        if (q >= kDPSlots) return 0;                    // (wave-uniform)
        const int ln = opaque_lane();
        const int lrow = ln >> 3, pc = ln & 7;
        const unsigned lc = (unsigned)((pc ^ ((4 * wave + (lrow >> 1)) & 7)) << 4) + ccoff;
        const unsigned rowbytes = (unsigned)(p.Cin * ES);
        unsigned off;
        bool ok;
        if (TW) {
            if (j == 0) {
                const int pr = wave * 8 + lrow;
                const int py0 = pr / RS;
                pp_yx = (py0 << 8) | (pr - py0 * RS);
            } else {
                pp_yx += ((64 / RS) << 8) | (64 % RS);
                if ((pp_yx & 0xff) >= RS) pp_yx += 256 - RS;
            }
            const int pp_py = pp_yx >> 8, pp_px = pp_yx & 0xff;
            const int y = pt_y0 + pp_py, x = pt_x0 + pp_px;
            ok = pp_py < TH + 2 && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
            off = (unsigned)((pt_b * p.H + y) * p.W + x) * rowbytes;
        } else {
            const int pr = q * 8 + lrow;
            const int pix = pt_j0 + pr;
            ok = pr < 256 + 2 * p.W + 2 && pix >= 0 && pix < p.M;
            off = (unsigned)pix * rowbytes;
        }
        const int piece = dstbuf_off + q * 1024;
        if (!ok) {
            unsigned z;
            asm volatile("v_mov_b32 %0, 0" : "=v"(z));
            *(u32x4 *)(smem + piece + ln * 16) = u32x4{z, z, z, z};
        }
        const bool any = __builtin_amdgcn_ballot_w64(ok) != 0ull;
        if (any) {
            if (ok) dp_glds16(p.in, off + lc, __builtin_amdgcn_readfirstlane(smem_lds + piece));
            return 1;
        }
        return 0;
    };
    // my group's half of the chunk's pointwise weights: rows grp*128 + (cw + 4k)*8 + lrow, k = 0..3
    const unsigned wo = (unsigned)((grp * 128 + cw * 8 + (lane >> 3)) * p.Cin * ES) + (unsigned)((((lane & 7) ^ ((4 * cw + (lane >> 4)) & 7)) << 4));
    const unsigned wstep = (unsigned)(32u * p.Cin * ES);
    auto weight_pieces = [&](int item, int c) {
        const unsigned off = (unsigned)__builtin_amdgcn_readfirstlane(nt_of(item) * BN * p.Cin * ES + c * 128);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            unsigned w = wo;
            asm volatile("" : "+v"(w));
            dp_glds16(p.w + off, w + k * wstep, __builtin_amdgcn_readfirstlane(smem_lds + OFF_W + grp * (BN * 64) + (cw + 4 * k) * 1024));
        }
    };
    // the chunk's depthwise weights: three 1-KiB pieces = rows of 64 fp32 (taps 0..8, the bias, two rows of padding), by wave 1
    auto dw_pieces = [&](int c, int par) {
        const int ln = opaque_lane();
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int t = 4 * q + (ln >> 4);
            const unsigned src = (t < 9 ? (unsigned)(t * p.Cin * 4) : p.bdw_off) + (unsigned)(c * 256 + (ln & 15) * 16);
            if (t < 10) dp_glds16(p.wdw, src, __builtin_amdgcn_readfirstlane(smem_lds + OFF_DW + par * 3072 + q * 1024));
        }
    };

    // =========================== compute state ===========================
    f32x16 acc[WC][2];
    int cur_mt = -1, n0 = 0;
    long long tile_pix0 = 0;
    int tile_row0 = 0, tile_x0 = 0;
    unsigned dwmask = 0xFFFFFFFFu;                      // flat tiles: 4 x (3 row bits | 3 column bits << 3) of my four depthwise pixels
    auto setup_item = [&](int item) {
        const int mt = mt_of(item);
        n0 = nt_of(item) * BN;
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
        const int pg = (int)(threadIdx.x >> 3);
        unsigned mk = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long long m = tile_pix0 + 4 * pg + k;
            unsigned bits = 0;
            if (m < p.M) {
                const int rem = (int)(m % ((long long)p.H * p.W));
                const int y = rem / p.W, x = rem - y * p.W;
                bits = (y > 0 ? 1u : 0u) | 2u | (y < p.H - 1 ? 4u : 0u) | (x > 0 ? 8u : 0u) | 16u | (x < p.W - 1 ? 32u : 0u);
            }
            mk |= bits << (8 * k);
        }
        dwmask = mk;
    };

    // ---- the depthwise conv of one chunk: patch buffer -> operand tile.  Thread = (16-byte channel column cg, pixels 4pg..4pg+3),
    // two pixels at a time (the 128 accumulators of the pointwise GEMM are live: ~50 registers is all this phase may use; the
    // loops over pixel pairs and tap rows are NOT unrolled for that reason).
    auto dw_phase = [&](int pbuf, int par) {
        const int tid = (int)threadIdx.x;
        int cg = tid & 7, pg = tid >> 3;
        asm volatile("" : "+v"(cg), "+v"(pg));
        const char *pb = smem + pbuf * PBYTES;
        const char *dwb = smem + OFF_DW + par * 3072 + cg * 32;
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            const int i0 = 4 * pg + 2 * half;           // tile-local pixel of my first output of this pair
            const int row00 = TW ? (i0 >> LGTW) * RS + (i0 & (TW - 1)) : i0;      // patch row of tap (0,0) of that pixel
            float a[2][P16];
            {
                const f32x4 b0 = *(const f32x4 *)(dwb + 9 * 256), b1 = *(const f32x4 *)(dwb + 9 * 256 + 16);
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int j = 0; j < 4; ++j) { a[k][j] = b0[j]; a[k][4 + j] = b1[j]; }
            }
#pragma unroll 1
            for (int ti = 0; ti < 3; ++ti) {
                float wt[3][P16];
#pragma unroll
                for (int tj = 0; tj < 3; ++tj) {
                    const f32x4 w0 = *(const f32x4 *)(dwb + (3 * ti + tj) * 256), w1 = *(const f32x4 *)(dwb + (3 * ti + tj) * 256 + 16);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { wt[tj][j] = w0[j]; wt[tj][4 + j] = w1[j]; }
                }
                const int rbase = row00 + ti * RS;
                unsigned mrow = 0;
                if constexpr (FLAT) mrow = dwmask >> (16 * half);   // my pair's two mask bytes
#pragma unroll
                for (int col = 0; col < 4; ++col) {     // patch rows rbase + col: input columns x-1 .. x+2 of my first pixel
                    const int r = rbase + col;
                    const u32x4 raw = *(const u32x4 *)(pb + r * 128 + ((cg ^ ((r >> 1) & 7)) << 4));
                    float v[P16];
                    unpack16<DT>(raw, v);
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const int tj = col - k;         // tap column of output k that reads this input column
                        if (tj < 0 || tj > 2) continue;
                        if constexpr (FLAT) {
                            const bool okk = ((mrow >> (8 * k + ti)) & 1u) && ((mrow >> (8 * k + 3 + tj)) & 1u);
#pragma unroll
                            for (int j = 0; j < P16; ++j) a[k][j] = fmaf(wt[tj][j], okk ? v[j] : 0.f, a[k][j]);
                        } else {
#pragma unroll
                            for (int j = 0; j < P16; ++j) a[k][j] = fmaf(wt[tj][j], v[j], a[k][j]);
                        }
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (p.relu_dw) {
#pragma unroll
                    for (int j = 0; j < P16; ++j) a[k][j] = fmaxf(a[k][j], 0.f);
                }
                const int i = i0 + k;
                *(u32x4 *)(smem + OFF_A + i * 128 + ((cg ^ ((i >> 1) & 7)) << 4)) = pack16<DT>(a[k]);
            }
        }
    };

    // ---- epilogue of one item (wave-private staging strip in the dead patch buffer -> whole-line stores) ----------
    auto epilogue = [&](char *stg) {
        const int ln = opaque_lane();
        const int r32 = ln & 31, hh = ln >> 5;
        constexpr int CPR = BNH * ES / 16, RPI = 64 / CPR;
        const int my_ch = ln % CPR, my_row = ln / CPR;
        const int my_c = n0 + grp * BNH + my_ch * P16;
        auto pixel_of = [&](int i) -> long long {
            if (TW) return tile_pix0 + (long long)(i >> LGTW) * p.W + (i & (TW - 1));
            const long long m = tile_pix0 + i;
            return m < p.M ? m : -1;
        };
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
#pragma unroll 1
            for (int rd = 0; rd < 32 / SROWS; ++rd) {
                if (r32 / SROWS == rd) {
#pragma unroll
                    for (int ci = 0; ci < WC; ++ci)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            // (+ bias AFTER the accumulation, like conv_igemm.hip's epilogue: the two-launch plan's bits)
                            const f32x4 bv = *(const f32x4 *)(p.bias + n0 + grp * BNH + ci * 32 + 8 * g + 4 * hh);
                            float q4[4];
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const float t = acc[ci][pt][4 * g + j] + bv[j];
                                q4[j] = p.relu ? fmaxf(t, 0.f) : t;
                            }
                            char *d = stg + (r32 % SROWS) * SSTRIDE + (ci * 32 + 8 * g + 4 * hh) * ES;
                            *(uint2 *)d = make_uint2(pack2<DT>(q4[0], q4[1]), pack2<DT>(q4[2], q4[3]));
                        }
                }
                DP_LGKM0();
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int k = 0; k < SROWS / RPI; ++k) {
                    const int row = my_row + k * RPI;
                    const long long gp = pixel_of(cw * 64 + pt * 32 + rd * SROWS + row);
                    if (gp >= 0 && my_c < p.Cout)
                        *(u32x4 *)(p.out + ((size_t)gp * p.Cs + my_c) * ES) = *(const u32x4 *)(stg + row * SSTRIDE + my_ch * 16);
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
    };

    auto zero_acc = [&]() {
#pragma unroll
        for (int ci = 0; ci < WC; ++ci)
#pragma unroll
            for (int pt = 0; pt < 2; ++pt)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[ci][pt][e] = 0.f;
    };

    // the 32 MFMAs of one chunk: K slices 0..3 in order, operand fragments straight from LDS (8 weight + 4 pixel reads per half)
    auto mma_phase = [&]() {
        const int ln = opaque_lane();
        const int r32 = ln & 31, hh = ln >> 5;
        const unsigned wa = (unsigned)(OFF_W + grp * (BN * 64) + r32 * 128 + ((hh ^ ((r32 >> 1) & 7)) << 4));
        unsigned pa[2];
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
            const int i = cw * 64 + pt * 32 + r32;
            pa[pt] = (unsigned)(OFF_A + i * 128 + ((hh ^ ((i >> 1) & 7)) << 4));
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            u32x4 wf[WC][2], pf[2][2];
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const unsigned kx = (unsigned)((2 * half + k2) << 5);
#pragma unroll
                for (int ci = 0; ci < WC; ++ci) wf[ci][k2] = *(const u32x4 *)(smem + ((wa ^ kx) + ci * 4096));
#pragma unroll
                for (int pt = 0; pt < 2; ++pt) pf[pt][k2] = *(const u32x4 *)(smem + (pa[pt] ^ kx));
            }
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                for (int ci = 0; ci < WC; ++ci)
#pragma unroll
                    for (int pt = 0; pt < 2; ++pt) MmaDP<DT>::run(wf[ci][k2], pf[pt][k2], acc[ci][pt]);
        }
    };

    // =========================== prologue ===========================
    int k_item = 0, cur_item = item0, cc = 0;           // the unit being computed: item index in my list, item, chunk
    patch_tile(cur_item);
#pragma unroll
    for (int j = 0; j < kDPPieces; ++j) (void)patch_piece(j, 0u, 0);
    if (wave == 1) dw_pieces(0, 0);
    if (wave == 2 && lane < 8) *(u32x4 *)(smem + OFF_Z + lane * 16) = u32x4{0u, 0u, 0u, 0u};
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    DP_BAR();
    setup_item(cur_item);
    zero_acc();
    int pbuf = 0, par = 0;

#pragma unroll 1
    for (;;) {
        // ---- the chunk's pointwise weights first (the buffer is free: everybody passed the barrier behind the last MFMA phase)
        weight_pieces(cur_item, cc);
        // ---- then, one unit ahead: the next chunk's patch and depthwise weights
        int n_item = cur_item, n_cc = cc + 1, n_k = k_item;
        if (n_cc == nchunks) { n_cc = 0; n_k = k_item + 1; n_item = item0 + n_k * istride; }
        const bool has_next = n_k < n_items;
        int fly = 0;
        if (has_next) {
            if (n_cc == 0) patch_tile(n_item);
#pragma unroll
            for (int j = 0; j < kDPPieces; ++j) fly += patch_piece(j, (unsigned)(n_cc * 128), (pbuf ^ 1) * PBYTES);
            if (wave == 1) { dw_pieces(n_cc, par ^ 1); fly += 3; }
        }
        // ---- depthwise conv of this chunk -> operand tile
        dw_phase(pbuf, par);
        DP_LGKM0();
        // in-order retirement: everything but the `fly` youngest pieces has landed = this chunk's pointwise weights
        switch (fly) {
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
            case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
            case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
            case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        }
        DP_BAR();
        // ---- pointwise: 32 MFMAs per wave
        __builtin_amdgcn_s_setprio(1);
        mma_phase();
        __builtin_amdgcn_s_setprio(0);
        if (cc + 1 == nchunks) {
            // the item is complete; the patch buffer of this chunk is dead (the next unit's patch goes to the other one)
            epilogue(smem + pbuf * PBYTES + wave * STRIP);
            if (has_next) {
                setup_item(n_item);
                zero_acc();
            }
        }
        if (!has_next) break;
        k_item = n_k; cur_item = n_item; cc = n_cc;
        pbuf ^= 1; par ^= 1;
        // everything issued for the next unit has landed, my LDS reads of this chunk are done -> (barrier) everybody's
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        DP_BAR();
    }
}

// =============================================================================================
// pw1x1_kernel -- the pointwise half alone: a 1x1 conv = GEMM  out[M][Cout] = relu(in[M][Cin] * W^T + bias)  with M = B*H*W
// rows of 128-byte channel chunks, on the same 256 x 256 items and wave tiles, both operands double-buffered by LDS-DMA one
// chunk ahead, ONE workgroup barrier per chunk.  Replaces conv_igemm.hip on the wide pointwise layers of the MobileNet
// trunks (K = Cin <= 1024: eight K steps per 128 x 128 tile there, prologue and epilogue dominate: 0.20 of the MFMA peak,
// profiles/r04_cfg4).  Same K order, bias after the accumulation: bit-identical to conv_igemm's output.
// =============================================================================================
struct Pw1x1Params {
    const char *in, *w;
    const float *bias;
    char *out;
    int M, Cin, Cout, Cs, relu;
    int m_tiles, n_tiles, items;
    int ablate;                    // diagnostics (TDRN_PW_ABLATE): 1 no pixel DMA, 2 no weight DMA, 4 no LDS reads / MFMA, 8 no stores
    int n_major;                   // item = nt * m_tiles + mt instead of mt * n_tiles + nt (see the kernel)
    int stagger;                   // group 1 issues a unit's LDS-DMA pieces BETWEEN its two MFMA halves, group 0 in front of them (TDRN_PW_STAGGER=0: both in front)
    int tail_split;                // an XCD's last, sparsely filled round of items runs as 64- or 128-cout sub-items (see the kernel)
};

// (developer builds only: the ablation switches are compiled out of the product -- a RUNTIME branch around the MFMA block of a K loop made
// hipcc keep the accumulators in a second register set and copy all of them around every step in conv_igemm.hip, round 6)
__device__ __forceinline__ bool pw_store_on(const Pw1x1Params &p)
{
#ifdef TDRN_DEV_ABLATE
    return !(p.ablate & 8);
#else
    (void)p;
    return true;
#endif
}

template <typename DT>
__global__ __launch_bounds__(512, 2) void pw1x1_kernel(const Pw1x1Params p)
{
    constexpr int ES = 2, P16 = 8;
    constexpr int BN = 256, WC = 4;
    // LDS: THREE pixel-operand buffers (the activations stream from HBM / the Infinity Cache: ~2 us under load, more than one
    // chunk's 32 MFMAs -- with one chunk of prefetch the kernel ran at conv_igemm's speed) + two weight buffers (L2 hits) = 160 KiB;
    // the epilogue's staging strips live in the pixel buffer that died with the item's last chunk
    constexpr int ABYTES = 256 * 128, WBYTES = BN * 128;
    constexpr int OFF_W = 3 * ABYTES;
    constexpr int LDS = OFF_W + 2 * WBYTES;
    static_assert(LDS <= 160 * 1024, "LDS budget");
    __shared__ __attribute__((aligned(16))) char smem[LDS];

    const int lane = threadIdx.x & 63;
    auto opaque_lane = [&]() -> int {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        return ln;
    };
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wave >> 2, cw = wave & 3;
    const int nchunks = p.Cin / 64;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int per_xcd = (p.items + 7) >> 3, istride = ((int)gridDim.x + 7) >> 3;
    int avail = p.items - xcd * per_xcd;
    avail = avail < per_xcd ? avail : per_xcd;
    avail = avail < 0 ? 0 : avail;
    int n_full = avail > slot ? (avail - slot + istride - 1) / istride : 0;
    const int item0 = xcd * per_xcd + slot;
    // ---- tail split (round 6; conv3x3_patch.hip has the same idea).  800 items on 256 workgroups are 3.125 rounds: after three rounds an
    // XCD has 4 items left for its 32 workgroups and the launch's last quarter runs on an eighth of the chip (the five 512 -> 512 layers of
    // config 4, 256 -> 512, ...).  When the XCD's last round is at most a quarter (half) filled, its items are cut ALONG THE COUTS into
    // four 64-cout (two 128-cout) sub-items, one per workgroup: the same eight waves with one (two) 32-cout accumulator tiles each, a
    // quarter (half) of the weight rows, bias and stores, the same pixel rows.  Every output element sees the same MFMA rows in the same
    // K order: bit-identical whatever the batch does to the cut (TDRN_PLAN_NO_PATCH_TAIL / TDRN_PATCH_TAIL=0 keep whole items).
    int n_tail = 0, tail_item = 0, tail_c0 = 0, tail_wcn = WC;
    if (p.tail_split && avail > 0) {
        const int full = avail / istride, rem = avail - full * istride;
        const int f = (full > 0 && rem > 0) ? (4 * rem <= istride ? 4 : (2 * rem <= istride ? 2 : 0)) : 0;
        if (f) {
            n_full = full;
            if (slot < f * rem) {
                n_tail = 1;
                tail_item = xcd * per_xcd + full * istride + slot / f;
                tail_wcn = WC / f;
                tail_c0 = (slot % f) * (BN / f);
            }
        }
    }
    const int n_items = n_full + n_tail;
    if (n_items == 0) return;
    // element i of this workgroup's sequence: item, first cout inside its 256-cout tile, 32-cout accumulator tiles per wave
    auto seq_item = [&](int i) -> int { return i < n_full ? item0 + i * istride : tail_item; };
    auto seq_c0 = [&](int i) -> int { return i < n_full ? 0 : tail_c0; };
    auto seq_wcn = [&](int i) -> int { return i < n_full ? WC : tail_wcn; };
    const unsigned smem_lds = __builtin_amdgcn_readfirstlane(dp_lds_addr(smem));
    // item numbering: pixel-tile major by default -- the cout tiles of one pixel tile are neighbouring items of ONE XCD, dealt to
    // neighbouring workgroups at the same time, so the pixel rows (the operand that streams from HBM) cross the fabric once; the
    // whole weight matrix (<= 2 MB) stays in every XCD's L2.  (cout-tile major, conv3x3_pp.hip's choice for its 9x bigger weight
    // matrices, fetched the activations once per cout tile: 185 MB for a 105-MB tensor, profiles/r04_cfg4.)
    auto mt_of = [&](int item) -> int { return p.n_major ? item % p.m_tiles : item / p.n_tiles; };
    auto nt_of = [&](int item) -> int { return p.n_major ? item / p.m_tiles : item % p.n_tiles; };

    // one chunk of an operand: wave w stages pixel rows [8(w + 8k), +8) / its group's weight rows, k = 0..3 (4 pieces each);
    // rows past M re-read the last row (their outputs are never stored)
    const unsigned rowb = (unsigned)(p.Cin * ES);
    auto stage_a = [&](int i, int c, int abuf) {
#ifdef TDRN_DEV_ABLATE
        if (p.ablate & 1) return;
#endif
        const int ln = opaque_lane();
        const int lrow = ln >> 3, pc = ln & 7;
        const int mt = mt_of(seq_item(i));
        const unsigned lc = (unsigned)((pc ^ ((4 * wave + (lrow >> 1)) & 7)) << 4) + (unsigned)(c * 128);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int m = mt * 256 + (wave + 8 * k) * 8 + lrow;
            m = m < p.M ? m : p.M - 1;
            dp_glds16(p.in, (unsigned)m * rowb + lc, __builtin_amdgcn_readfirstlane(smem_lds + abuf * ABYTES + (wave + 8 * k) * 1024));
        }
    };
    auto stage_w = [&](int i, int c, int wbuf) {
#ifdef TDRN_DEV_ABLATE
        if (p.ablate & 2) return;
#endif
        const int ln = opaque_lane();
        const int lrow = ln >> 3, pc = ln & 7;
        const int wcn = seq_wcn(i);                     // my group's rows: 32 wcn of them, behind the (sub-)item's first cout
        const unsigned woff = (unsigned)__builtin_amdgcn_readfirstlane((nt_of(seq_item(i)) * BN + seq_c0(i)) * p.Cin * ES + c * 128);
        const unsigned wo = (unsigned)((grp * 32 * wcn + cw * 8 + lrow) * p.Cin * ES) + (unsigned)(((pc ^ ((4 * cw + (lrow >> 1)) & 7)) << 4));
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < wcn)
                dp_glds16(p.w + woff, wo + (unsigned)k * (32u * rowb), __builtin_amdgcn_readfirstlane(smem_lds + OFF_W + wbuf * WBYTES + grp * (WBYTES / 2) + (cw + 4 * k) * 1024));
    };
    const int n_units = n_items * nchunks;
    // (item, chunk) of unit u + d, advanced incrementally (no division in the loop)
    struct Cur { int item, c; };                        // (item = index into this workgroup's sequence)
    auto next_of = [&](Cur q) -> Cur {
        Cur r = q;
        if (++r.c == nchunks) { r.c = 0; r.item += 1; }
        return r;
    };

    f32x16 acc[WC][2];
    auto zero_acc = [&]() {
#pragma unroll
        for (int ci = 0; ci < WC; ++ci)
#pragma unroll
            for (int pt = 0; pt < 2; ++pt)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[ci][pt][e] = 0.f;
    };
    auto mma_phase = [&](auto wcn_tag, int abuf, int wbuf, int h0 = 0, int h1 = 2) {
        constexpr int WCN = decltype(wcn_tag)::value;   // accumulator tiles per wave along the couts: 4 (whole item), 2, 1 (tail sub-items)
#ifdef TDRN_DEV_ABLATE
        if (p.ablate & 4) return;
#endif
        const int ln = opaque_lane();
        const int r32 = ln & 31, hh = ln >> 5;
        const unsigned wa = (unsigned)(OFF_W + wbuf * WBYTES + grp * (WBYTES / 2) + r32 * 128 + ((hh ^ ((r32 >> 1) & 7)) << 4));
        unsigned pa[2];
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
            const int i = cw * 64 + pt * 32 + r32;
            pa[pt] = (unsigned)(abuf * ABYTES + i * 128 + ((hh ^ ((i >> 1) & 7)) << 4));
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (half < h0 || half >= h1) continue;
            u32x4 wf[WCN][2], pf[2][2];
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const unsigned kx = (unsigned)((2 * half + k2) << 5);
#pragma unroll
                for (int ci = 0; ci < WCN; ++ci) wf[ci][k2] = *(const u32x4 *)(smem + ((wa ^ kx) + ci * 4096));
#pragma unroll
                for (int pt = 0; pt < 2; ++pt) pf[pt][k2] = *(const u32x4 *)(smem + (pa[pt] ^ kx));
            }
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                for (int ci = 0; ci < WCN; ++ci)
#pragma unroll
                    for (int pt = 0; pt < 2; ++pt) MmaDP<DT>::run(wf[ci][k2], pf[pt][k2], acc[ci][pt]);
        }
    };
    // my group's 128 biases: one f32x4 per lane (lanes 0..31), fetched BEFORE the item's last MFMA phase (the latency hides under
    // it), parked in a wave-private LDS copy behind the staging strips for the epilogue's rounds.  (Read from global inside the
    // rounds -- eight rounds, sixteen loads each -- the epilogue cost 30 us of a 105-us launch.)
    f32x4 bias4 = f32x4{0.f, 0.f, 0.f, 0.f};
    auto fetch_bias = [&](int i) {
        const int ln = opaque_lane();
        const int wcn = seq_wcn(i);
        if (ln < 8 * wcn) bias4 = *(const f32x4 *)(p.bias + nt_of(seq_item(i)) * BN + seq_c0(i) + grp * 32 * wcn + 4 * ln);
    };
    // Epilogue of one item.  Rounds are (pixel tile, 64-cout half): EVERY lane stages its pixel row's 64 couts (8 quads) into a
    // wave-private strip of 32 rows x 128 B, then the wave copies the strip out as whole 128-byte lines.  (A first version staged
    // 8 pixel rows x 128 couts per round -- a quarter of the lanes active -- and read each bias quad from LDS right where it was
    // used: 128 dependent LDS round trips, 27k cycles per item = half of the launch.)  Strips of waves 0-3 live in the pixel
    // buffer, of waves 4-7 in the weight buffer, that died with the item's last chunk.
    constexpr int SST = 64 * ES + 16;                   // strip row stride: 64 couts + 16 B against bank conflicts
    constexpr int SBYTES = 32 * SST;                    // 4.5 KiB per wave
    static_assert(4 * (SBYTES + 512) <= ABYTES && 4 * (SBYTES + 512) <= WBYTES, "strips + bias copies fit the dead buffers");
    auto epilogue = [&](auto wcn_tag, int i, int dead_abuf, int dead_wbuf) {
        constexpr int WCN = decltype(wcn_tag)::value;
        const int item = seq_item(i);
        char *base = (wave < 4 ? smem + dead_abuf * ABYTES : smem + OFF_W + dead_wbuf * WBYTES) + (wave & 3) * (SBYTES + 512);
        char *stg = base, *sbias = base + SBYTES;
        const int ln = opaque_lane();
        if (ln < 32) *(f32x4 *)(sbias + ln * 16) = bias4;
        DP_LGKM0();
        __builtin_amdgcn_wave_barrier();
        const int r32 = ln & 31, hh = ln >> 5;
        const int n0 = nt_of(item) * BN + seq_c0(i);
        const long long pix0 = (long long)mt_of(item) * 256 + cw * 64;
        const int orow = ln >> 3, och = ln & 7;         // copy-out: 8 lanes x 16 B = one 128-byte line of a pixel row, 8 rows per pass
        constexpr int NCH = (WCN + 1) / 2;              // rounds of 64 couts (a 32-cout wave tile fills half a strip row)
        constexpr int NC2 = WCN >= 2 ? 2 : 1;
#pragma unroll
        for (int chalf = 0; chalf < NCH; ++chalf) {
            f32x4 bv[NC2][4];
#pragma unroll
            for (int c2 = 0; c2 < NC2; ++c2)
#pragma unroll
                for (int g = 0; g < 4; ++g) bv[c2][g] = *(const f32x4 *)(sbias + ((2 * chalf + c2) * 32 + 8 * g + 4 * hh) * 4);
            const int my_c = n0 + grp * (32 * WCN) + chalf * 64 + och * P16;
            const bool och_ok = och * P16 < 32 * NC2;   // (a 32-cout tile: the row's first 64 bytes only)
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) {
#pragma unroll
                for (int c2 = 0; c2 < NC2; ++c2)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float q4[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float t = acc[2 * chalf + c2][pt][4 * g + j] + bv[c2][g][j];      // (+ bias AFTER the accumulation: conv_igemm's bits)
                            q4[j] = p.relu ? fmaxf(t, 0.f) : t;
                        }
                        *(uint2 *)(stg + r32 * SST + (c2 * 32 + 8 * g + 4 * hh) * ES) = make_uint2(pack2<DT>(q4[0], q4[1]), pack2<DT>(q4[2], q4[3]));
                    }
                DP_LGKM0();
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int row = orow + 8 * k;
                    const long long gp = pix0 + pt * 32 + row;
                    if (gp < p.M && my_c < p.Cout && och_ok && pw_store_on(p))
                        *(u32x4 *)(p.out + ((size_t)gp * p.Cs + my_c) * ES) = *(const u32x4 *)(stg + row * SST + och * 16);
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
    };

    // ---- pipeline: in iteration u the weights of unit u+1 and the pixels of unit u+2 are issued, in that order: the in-order
    // counter then lets the four youngest pieces (pixels, needed one iteration later) stay in flight
    Cur q0{0, 0};
    Cur q1 = next_of(q0), q2 = next_of(q1);
    stage_w(q0.item, q0.c, 0);
    stage_a(q0.item, q0.c, 0);
    if (n_units > 1) stage_a(q1.item, q1.c, 1);
    zero_acc();
    int abuf = 0, wbuf = 0;
    // One unit = one 64-channel chunk of one (sub-)item.  The unit body is a generic lambda instantiated once per accumulator-tile count
    // and run by SEPARATE loops (whole items first, then this workgroup's tail sub-item): with the three MFMA bodies as alternatives inside
    // one loop hipcc gave each its own accumulator registers and copied all 128 of them around every unit (first build of the tail split:
    // every pointwise layer 3-4x slower, tail split on or off).
    typedef std::integral_constant<int, 4> w4_t;
    typedef std::integral_constant<int, 2> w2_t;
    typedef std::integral_constant<int, 1> w1_t;
    auto unit = [&](auto wtag, int u) {
        // unit u's operands have landed: its weights were issued BEFORE the (younger) pixel pieces of unit u+1
        if (u + 1 < n_units) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        DP_BAR();                                       // ... everybody's; and every wave has left the MFMAs of unit u-1
        const bool late = p.stagger && grp == 1;        // (wave-uniform: group 1 issues its pieces between its two MFMA halves)
        if (!late) {
            if (u + 1 < n_units) stage_w(q1.item, q1.c, wbuf ^ 1);
            if (u + 2 < n_units) stage_a(q2.item, q2.c, abuf == 0 ? 2 : abuf - 1);     // (= the buffer of unit u-1)
        }
        const bool last = q0.c + 1 == nchunks;
        if (last) fetch_bias(q0.item);
        __builtin_amdgcn_s_setprio(1);
        mma_phase(wtag, abuf, wbuf, 0, 1);
        if (late) {
            if (u + 1 < n_units) stage_w(q1.item, q1.c, wbuf ^ 1);
            if (u + 2 < n_units) stage_a(q2.item, q2.c, abuf == 0 ? 2 : abuf - 1);
        }
        mma_phase(wtag, abuf, wbuf, 1, 2);
        __builtin_amdgcn_s_setprio(0);
        if (last) {
            DP_LGKM0();
            DP_BAR();                                   // every wave has read unit u's pixels: that buffer is the staging area now
            epilogue(wtag, q0.item, abuf, wbuf);
            zero_acc();
            DP_LGKM0();
        }
        q0 = q1; q1 = q2; q2 = next_of(q2);
        abuf = abuf == 2 ? 0 : abuf + 1;
        wbuf ^= 1;
    };
    const int n_main = n_full * nchunks;
    int u = 0;
#pragma unroll 1
    for (; u < n_main; ++u) unit(w4_t{}, u);
    if (n_tail) {
        if (tail_wcn == 2) {
#pragma unroll 1
            for (; u < n_units; ++u) unit(w2_t{}, u);
        } else {
#pragma unroll 1
            for (; u < n_units; ++u) unit(w1_t{}, u);
        }
    }
}

// ---------------------------------------------------------------------------------------------
int dwpw_enabled()
{
    static int e = -1;
    if (e < 0) { const char *s = getenv("TDRN_DWPW"); e = s ? atoi(s) : 1; }      // 0: never; 1: where the plan asks (TDRN_PLAN_DWPW); 2: every plan
    return e;
}

// 0 = no; else the tile mode (32 / 16 = 2-D tiles, -1 = flat tiles)
int dwpw_supported(const DwPwArgs &a)
{
    if (!dwpw_enabled() || a.dtype == TDRN_F32) return 0;
    if (a.stride != 1 || a.Cin % 64 || a.Cin < 64 || a.Npad % 256 || a.Cout > a.Npad || a.Cs < a.Cout) return 0;
    if ((long long)a.B * a.H * a.W * a.Cin * 2 >= (1ll << 32)) return 0;       // 32-bit byte offsets into the input
    if ((long long)a.Npad * a.Cin * 2 >= (1ll << 31)) return 0;
    if (a.W % 32 == 0 && a.H % 8 == 0) return 32;
    if (a.W % 16 == 0 && a.H % 16 == 0) return 16;
    if (2 * a.W + 2 + 256 <= kDPSlots * 8 && a.W >= 3) return -1;
    return 0;
}

int launch_dwpw(const DwPwArgs &a, hipStream_t s)
{
    const int mode = dwpw_supported(a);
    if (!mode) return TDRN_E_UNSUPPORTED;
    if (!a.in || !a.w || !a.wdw || !a.bdw || !a.bias || !a.out) return TDRN_E_ARG;
    if ((const char *)a.bdw < (const char *)a.wdw || (const char *)a.bdw - (const char *)a.wdw >= (1ll << 31)) return TDRN_E_ARG;
    DwPwParams p;
    p.in = (const char *)a.in; p.w = (const char *)a.w; p.wdw = (const char *)a.wdw;
    p.bdw_off = (unsigned)((const char *)a.bdw - (const char *)a.wdw);
    p.bias = a.bias; p.out = (char *)a.out;
    p.B = a.B; p.H = a.H; p.W = a.W; p.Cin = a.Cin; p.Cout = a.Cout; p.Cs = a.Cs;
    p.relu_dw = a.relu_dw; p.relu = a.relu;
    p.M = a.B * a.H * a.W;
    const int tw = mode > 0 ? mode : 0;
    if (tw) {
        p.tiles_x = a.W / tw;
        p.tiles_per_img = p.tiles_x * (a.H / (256 / tw));
        p.m_tiles = a.B * p.tiles_per_img;
    } else {
        p.tiles_x = 0; p.tiles_per_img = 0;
        p.m_tiles = cdiv(p.M, 256);
    }
    p.n_tiles = a.Npad / 256;
    p.items = p.m_tiles * p.n_tiles;
    if (p.items <= 0) return TDRN_OK;
    const int grid = p.items >= 256 ? 256 : ((p.items + 7) / 8) * 8;
#define DP_LAUNCH(DT)                                                                                        \
    do {                                                                                                     \
        if (tw == 0) hipLaunchKernelGGL((dwpw_kernel<DT, 0>), dim3(grid), dim3(512), 0, s, p);              \
        else if (tw == 32) hipLaunchKernelGGL((dwpw_kernel<DT, 32>), dim3(grid), dim3(512), 0, s, p);      \
        else hipLaunchKernelGGL((dwpw_kernel<DT, 16>), dim3(grid), dim3(512), 0, s, p);                     \
    } while (0)
    if (a.dtype == TDRN_BF16) DP_LAUNCH(bf16_t);
    else DP_LAUNCH(f16_t);
#undef DP_LAUNCH
    return hip_status(hipGetLastError());
}


int pw1x1_enabled()
{
    static int e = -1;
    if (e < 0) { const char *s = getenv("TDRN_PW1X1"); e = s ? atoi(s) : 1; }
    return e;
}

// the 1x1 / stride 1 / unpadded convs this kernel takes over from conv_igemm.hip: 16-bit, whole 64-channel chunks, couts in whole
// 256-groups, a plain NHWC output tensor, no residual, no split-K, and enough items to fill the chip
int pw1x1_supported(const ConvArgs &a)
{
    if (!pw1x1_enabled() || (a.kdisable & 8) || a.dtype == TDRN_F32) return 0;
    if (a.kh != 1 || a.kw != 1 || a.stride != 1 || a.pad != 0 || a.phases != 1 || a.res || a.out_f32 || a.splitk > 1 || a.fuse_x) return 0;
    if (a.Ho != a.H || a.Wo != a.W || a.Cin % 64 || a.Npad % 256 || a.Cout > a.Npad) return 0;
    if (a.o_rs != (long long)a.Wo * a.o_cs || a.o_bs != (long long)a.Ho * a.Wo * a.o_cs || a.o_base) return 0;
    const long long M = (long long)a.B * a.H * a.W;
    if (M * a.Cin * 2 >= (1ll << 32) || (long long)a.Npad * a.Cin * 2 >= (1ll << 31)) return 0;
    return 1;
}

int launch_pw1x1(const ConvArgs &a, hipStream_t s)
{
    if (!pw1x1_supported(a)) return TDRN_E_UNSUPPORTED;
    Pw1x1Params p;
    p.in = (const char *)a.in; p.w = (const char *)a.w; p.bias = a.bias; p.out = (char *)a.out;
    p.M = a.B * a.H * a.W; p.Cin = a.Cin; p.Cout = a.Cout; p.Cs = (int)a.o_cs; p.relu = a.relu;
    p.m_tiles = cdiv(p.M, 256); p.n_tiles = a.Npad / 256; p.items = p.m_tiles * p.n_tiles;
    if (p.items <= 0) return TDRN_OK;
    // (below ~3/4 of a full grid conv_igemm's 128 x 128 tiles fill more CUs; the two kernels produce the same bits, so the choice
    // may depend on the batch)
    if (p.items < 192) return TDRN_E_UNSUPPORTED;
    static int ablate = -1;
    if (ablate < 0) ablate = dev_ablate_env("TDRN_PW_ABLATE");     // (developer builds only: common.h)
    p.ablate = ablate;
    static int nmajor = -1;
    if (nmajor < 0) { const char *e = getenv("TDRN_PW_NMAJOR"); nmajor = e ? atoi(e) : 0; }
    p.n_major = nmajor;
    static int stag = -1;
    // (round 6, default on: wave group 1 issues a unit's eight LDS-DMA pieces between its two MFMA halves, group 0 in front of them, so
    // the two waves of a SIMD no longer issue and multiply in lockstep: -2 % on the 8-chunk layers, -9...-10 % on the 16-chunk ones
    // (512 -> 1024, 1024 -> 1024 at 20 x 20); same arithmetic, same bits.  Round 4 had moved only the pixel pieces behind ALL the MFMAs: nothing.)
    if (stag < 0) { const char *e = getenv("TDRN_PW_STAGGER"); stag = e ? atoi(e) : 1; }
    p.stagger = stag;
    static int tail = -1;
    if (tail < 0) { const char *e = getenv("TDRN_PATCH_TAIL"); tail = e ? atoi(e) : 1; }
    p.tail_split = tail && !(a.kdisable & 1024);     // (TDRN_PLAN_NO_PATCH_TAIL; pixel-tile-major items only: the sub-items of a tile are neighbours)
    const int grid = p.items >= 256 ? 256 : ((p.items + 7) / 8) * 8;
    if (a.dtype == TDRN_BF16) hipLaunchKernelGGL((pw1x1_kernel<bf16_t>), dim3(grid), dim3(512), 0, s, p);
    else hipLaunchKernelGGL((pw1x1_kernel<f16_t>), dim3(grid), dim3(512), 0, s, p);
    return hip_status(hipGetLastError());
}

}  // namespace tdrn
