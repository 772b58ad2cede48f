// conv3x3_ws.hip -- WEIGHT-STATIONARY 3x3 / stride 1 / pad 1 convolution for the Cin = 64 layers of the 16-bit plans
// (conv1_2 with the first conv fused in, conv2_1: 374 of a step's 2307 GFLOP, and the two worst members of the family on
// conv3x3_patch.hip: 0.38 of the MFMA peak).
//
// Why a third kernel.  With one 64-channel chunk an item of conv3x3_patch.hip is nine (tap) steps: the stamps of round 4 show a
// quarter to a third of such an item lost BETWEEN items (the loaders refill the weight ring, the epilogue, and -- fused
// conv1_2 -- the next patch is computed while nobody multiplies) and 1400-1580 cycles per step against the 1024 of its MFMAs
// (one workgroup barrier per step).  But at Cin = 64 the whole weight matrix of a 64-cout tile is 9 taps x 64 couts x 128 B =
// 72 KiB: it FITS LDS next to the activations.  So:
//   * the weights are loaded ONCE per workgroup (again only when its cout tile changes) and never move: there is no weight
//     stream, no ring, no per-step barrier -- a 256-pixel tile is 144 MFMAs per wave between two barriers;
//   * the activations live in a RING of 18 image rows x 34 pixels x 128 B (76.5 KiB): a workgroup walks DOWN a 32-pixel
//     column strip of one image in tiles of 8 rows; tile t reads ring rows 8t .. 8t+9 while the 8 rows tile t+1 adds are
//     being produced into the other 8 slots -- every activation row is staged (conv1_2: COMPUTED) once per strip instead of
//     once per tile plus halo, and production always runs one tile ahead of consumption;
//   * four CONSUMER waves (one per SIMD, 64 pixels x 64 couts each: 4 ds_read_b128 per 4 MFMAs, fragments prefetched two
//     K-slices ahead in registers) and four PRODUCER waves (one per SIMD) meet at ONE barrier per tile.  Producers: plain
//     layers -- LDS-DMA of the next 8 rows; FUSE -- the first conv (3 -> 64 channels, BN folded, ReLU) of those rows on the matrix
//     cores from an fp32 halo tile of the raw frame that was LDS-DMA'd one tile earlier (same operand layout, same
//     instruction, same rounding as first_conv_mfma_kernel / the FUSE loaders of conv3x3_patch.hip: bit-identical).
// Same K order per output element as the other two kernels (bias first, then tap-major, 4 K-slices per tap, weights as the
// A operand): the output bits do not depend on which kernel ran (tests/test_gpu_pin16.py, dev/conv_check.hip).
//
// Work unit = (cout tile, image, column strip, row segment of TSEG tiles); a workgroup owns a CONTIGUOUS range of units of
// its XCD's share (vertical neighbours share two halo rows through L2; the cout tile changes at most once).  Per unit of T
// tiles the producers make T + 1 batches of rows -- 8 rows (into the slots the previous unit's last tile does not use),
// 2 rows, then 8 rows per further tile -- and the consumers multiply tile k while batch k + 2 is produced; the ring's slot
// counter runs on across units.
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "kernels.h"

namespace tdrn {

struct WsParams {
    const char *in, *w, *zero;
    const float *bias;
    char *out, *out_pool;          // NHWC [B][H][W][Cs] and / or pooled [B][H/2][W/2][Cs]
    int B, H, W, Cout, Cs, Ktot;   // Cin == 64 (one 128-byte chunk); Ktot = 9 * 64
    int relu;
    int SX, TY, TSEG, NSEG, NT;    // strips per row (W / 32), tiles per column (H / 8), tiles per segment, segments, cout tiles
    int units;                     // NT * B * SX * NSEG; unit = ((nt * B + b) * SX + sx) * NSEG + seg
    // FUSE: the layer's input is the first conv's output, computed here from the raw frames
    const float *fx, *fw, *fb;     // frames NCHW fp32 [B][3][S][S]; first-conv weights [64][27] (k = c*9 + r*3 + q) and bias [64], fp32
    int fS, fCout;
    // U8: the frames as uint8 planes [B][3][S][S] instead of fx; the net input is float(byte) - fmean[plane] (the
    // frame is read, and its mean subtracted, inside the first conv's loader)
    const unsigned char *fx8;
    float fmean[3];
#ifdef TDRN_WS_STAMP
    unsigned *stamps;              // diagnostics build only: [workgroup][wave][4] cycle sums (s_memtime), see the launcher
#endif
};

// In-kernel cycle stamps of a diagnostics build (make EXTRA=-DTDRN_WS_STAMP; never in the product): where do the consumer and the
// producer waves spend a period?  s_memtime returns through lgkmcnt, so the stamps sit where that counter is drained anyway.
#ifdef TDRN_WS_STAMP
#define WS_STAMP_DECL unsigned long long st_t = __builtin_amdgcn_s_memtime(); unsigned st_acc[4] = {0, 0, 0, 0};
#define WS_STAMP(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[k] += (unsigned)(t_ - st_t); st_t = t_; } while (0)
#define WS_STAMP_FLUSH do { if (lane == 0) for (int k_ = 0; k_ < 4; ++k_) p.stamps[((size_t)blockIdx.x * 8 + wave) * 4 + k_] = st_acc[k_]; } while (0)
#else
#define WS_STAMP_DECL
#define WS_STAMP(k) do { } while (0)
#define WS_STAMP_FLUSH do { } while (0)
#endif
#ifndef TDRN_WS_PRIO
#define TDRN_WS_PRIO 2            // 0: equal priorities, 1: producers at priority 2, 2: consumers at priority 2
#endif
#ifndef TDRN_WS_ABLATE
#define TDRN_WS_ABLATE 0          // diagnostics: 1 = producers produce nothing, 2 = consumers skip reads + MFMAs, 4 = no epilogue
#endif

namespace {

template <typename DT> struct MmaW;
template <> struct MmaW<bf16_t> {
    __device__ static __forceinline__ void run(const u32x4 &a, const u32x4 &b, f32x16 &c)
    { c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(i16x8, a), __builtin_bit_cast(i16x8, b), c, 0, 0, 0); }
};
template <> struct MmaW<f16_t> {
    __device__ static __forceinline__ void run(const u32x4 &a, const u32x4 &b, f32x16 &c)
    { c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0); }
};

// One LDS-DMA piece from a wave-uniform base plus a 32-bit per-lane byte offset: 64 lanes x 16 B -> 1 KiB (or x 4 B -> 256 B) of LDS at
// lds_dst + lane * size.  Inline asm as in conv3x3_pp.hip: hipcc's waitcnt pass does not see it, so it cannot put an
// `s_waitcnt vmcnt(0)` in front of the LDS accesses that follow (the producers prefetch a raw tile a whole tile time ahead and write ring
// rows meanwhile); completion is waited for by hand, once per period.  Lanes switched off by the caller's branch write nothing.
__device__ __forceinline__ void ws_glds16(const char *sbase, unsigned voff, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory");
}
__device__ __forceinline__ void ws_glds4(const char *sbase, unsigned voff, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory");
}
__device__ __forceinline__ unsigned ws_lds_addr(const void *p)
{
    return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const char *)p;
}

constexpr int kRingRows = 18, kRingCols = 34;
constexpr int kWBytes = 9 * 64 * 128;                   // 73 728: [tap][cout][128 B], 16-byte chunk c of row n at position c ^ ((n >> 1) & 7)
constexpr int kRingBytes = kRingRows * kRingCols * 128; // 78 336: pixel q = slot * 34 + col, chunk c at position c ^ ((q >> 1) & 7)
constexpr int kRawRows = 10, kRawCols = 36, kRawN = 3 * kRawRows * kRawCols;      // fp32 halo tile of a batch of <= 8 patch rows
constexpr int kRawPieces = (kRawN + 63) / 64;           // 17 LDS-DMA pieces of 256 B
constexpr int kRawBytes = kRawPieces * 256;             // 4352 per buffer

}  // namespace

// MODE: 1 = full-resolution output, 2 = fused MaxPool2d(2,2) output (compile-time: the epilogue's ops sit inside the MFMA loop)
// U8 (FUSE only): the raw frames are uint8 planes
template <typename DT, bool FUSE, int MODE, bool U8 = false>
__global__ __launch_bounds__(512, 2) void conv3x3_ws_kernel(const WsParams p)
{
    static_assert(sizeof(DT) == 2, "16-bit element types only");
    constexpr int OFF_W = 0;
    constexpr int OFF_RING = OFF_W + kWBytes;
    constexpr int OFF_BIAS = OFF_RING + kRingBytes;     // 64 floats
    constexpr int OFF_FB = OFF_BIAS + 256;              // FUSE: the first conv's bias, 64 floats
    constexpr int OFF_RAW = OFF_FB + 256;               // FUSE: two raw halo tiles
    constexpr int LDS = OFF_RAW + (FUSE ? 2 * kRawBytes : 0);
    static_assert(LDS <= 160 * 1024, "LDS budget");
    __shared__ __attribute__((aligned(16))) char smem[LDS];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r32 = lane & 31, hh = lane >> 5;

    // ---- my contiguous range of units ---------------------------------------------------------------------------------
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = ((int)gridDim.x + 7) >> 3;
    const int per_xcd = (p.units + 7) >> 3;
    int avail = p.units - xcd * per_xcd;
    avail = avail < per_xcd ? avail : per_xcd;
    avail = avail < 0 ? 0 : avail;
    const int u0 = xcd * per_xcd + (int)((long long)avail * slot / nslots), u1 = xcd * per_xcd + (int)((long long)avail * (slot + 1) / nslots);
    if (u1 <= u0) return;                               // (whole workgroup)

    struct Unit { int nt, b, x0, y0, T; };
    auto decode = [&](int u) -> Unit {
        Unit r;
        const int seg = u % p.NSEG;
        int t = u / p.NSEG;
        const int sx = t % p.SX;
        t /= p.SX;
        r.b = t % p.B;
        r.nt = t / p.B;
        r.x0 = sx * 32;
        r.y0 = seg * p.TSEG * 8;
        const int left = p.TY - seg * p.TSEG;
        r.T = left < p.TSEG ? left : p.TSEG;
        return r;
    };

    const unsigned smem_lds = __builtin_amdgcn_readfirstlane(ws_lds_addr(smem));
    // ---- weights (and bias) of a cout tile -> LDS: 72 pieces of 8 rows, nine per wave; piece (tap t, rows 8 wave .. 8 wave + 7) ----
    auto load_weights = [&](int nt) {
        const int lrow = lane >> 3, pc = lane & 7;
        const int n = 8 * wave + lrow;
        const char *base = p.w + (size_t)nt * 64 * p.Ktot * 2;
        const unsigned voff = (unsigned)(n * p.Ktot * 2 + ((pc ^ ((n >> 1) & 7)) << 4));
#pragma unroll
        for (int t = 0; t < 9; ++t) ws_glds16(base, voff + t * 128, __builtin_amdgcn_readfirstlane(smem_lds + OFF_W + (t * 64 + 8 * wave) * 128));
        if (wave == 0) ws_glds4((const char *)(p.bias + nt * 64), (unsigned)lane * 4u, __builtin_amdgcn_readfirstlane(smem_lds + OFF_BIAS));
    };
    auto write_first_bias = [&]() {
        if constexpr (FUSE) {
            if (wave == 1) ((float *)(smem + OFF_FB))[lane] = lane < p.fCout ? p.fb[lane] : 0.f;
        }
    };

    if (wave >= 4) {
        // ======================================= PRODUCERS =======================================
        const int lw = wave - 4;
        // ---- FUSE: lane constants of the raw-tile loads and of the first conv ---------------------------------------------
        [[maybe_unused]] int r_rc[FUSE ? 5 : 1], r_off[FUSE ? 5 : 1];
        [[maybe_unused]] float r_mean[U8 ? 5 : 1];               // U8: the mean of my element's plane
        [[maybe_unused]] unsigned r_byte[U8 ? 5 : 1];            // U8: the bytes of the NEXT batch's raw tile, loaded a period ahead (0x100: outside the frame)
        [[maybe_unused]] int koff1[16];
        [[maybe_unused]] u32x4 wq1[2][2];
        if constexpr (FUSE) {
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int i = (lw + 4 * j) * 64 + lane;
                const int c = i / (kRawRows * kRawCols), rem = i - c * (kRawRows * kRawCols);
                const int r = rem / kRawCols, q = rem - r * kRawCols;
                r_rc[j] = i < kRawN ? ((r << 8) | q) : -1;
                r_off[j] = (c * p.fS + r) * p.fS + q;
                if constexpr (U8) { r_mean[j] = c == 0 ? p.fmean[0] : (c == 1 ? p.fmean[1] : p.fmean[2]); r_byte[j] = 0x100u; }
            }
#pragma unroll
            for (int s2 = 0; s2 < 16; ++s2) {
                const int k = 16 * (s2 >> 3) + 8 * hh + (s2 & 7);
                const int c = k / 9, r = (k - 9 * c) / 3, q = k - 9 * c - 3 * r;
                koff1[s2] = k < 27 ? (c * kRawRows + r) * kRawCols + q : 0;      // k >= 27 pads K: its weight is 0
            }
#pragma unroll
            for (int ci = 0; ci < 2; ++ci)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const int co = ci * 32 + r32, k0 = 16 * ks + 8 * hh + 2 * jj;
                        const float a = (k0 < 27 && co < p.fCout) ? p.fw[co * 27 + k0] : 0.f;
                        const float b2 = (k0 + 1 < 27 && co < p.fCout) ? p.fw[co * 27 + k0 + 1] : 0.f;
                        wq1[ci][ks][jj] = pack2<DT>(a, b2);
                    }
        }
        // a batch: `nrows` patch rows starting at image row yf (may lie outside the image) of strip x0 of image b -> ring slots s0 ..
        struct Batch { int b, x0, yf, nrows, s0; };
        // raw halo tile of a batch -> raw buffer `buf` (my pieces)
        // U8: the raw tile's bytes are fetched into registers when issue_raw is called (ordinary byte loads, one period ahead of their use)
        // and turned into the fp32 tile by land_raw at the END of the period -- float(byte) - mean, 0 outside the frame: the same values the
        // fp32 route finds in its tile
        auto land_raw = [&](int buf) {
            if constexpr (U8) {
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    if ((lw + 4 * j) >= kRawPieces) continue;
                    *(float *)(smem + OFF_RAW + buf * kRawBytes + (lw + 4 * j) * 256 + lane * 4) = r_byte[j] < 0x100u ? (float)r_byte[j] - r_mean[j] : 0.f;
                }
            }
        };
        auto issue_raw = [&](const Batch &bt, int buf) {
            if constexpr (U8) {
                const unsigned char *xb = p.fx8 + (size_t)bt.b * 3 * p.fS * p.fS + ((long long)(bt.yf - 1) * p.fS + (bt.x0 - 2));
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    if ((lw + 4 * j) >= kRawPieces) continue;
                    const int rr = r_rc[j] >> 8, cc = r_rc[j] & 0xff;
                    const int yy = bt.yf - 1 + rr, xx = bt.x0 - 2 + cc;
                    const bool ok = r_rc[j] >= 0 && rr < bt.nrows + 2 && (unsigned)yy < (unsigned)p.fS && (unsigned)xx < (unsigned)p.fS;
                    r_byte[j] = ok ? (unsigned)xb[r_off[j]] : 0x100u;
                }
                (void)buf;
            } else if constexpr (FUSE) {
                const char *xb = (const char *)(p.fx + (size_t)bt.b * 3 * p.fS * p.fS + ((long long)(bt.yf - 1) * p.fS + (bt.x0 - 2)));
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    if ((lw + 4 * j) >= kRawPieces) continue;                        // (wave-uniform)
                    const int rr = r_rc[j] >> 8, cc = r_rc[j] & 0xff;
                    const int yy = bt.yf - 1 + rr, xx = bt.x0 - 2 + cc;
                    const bool ok = r_rc[j] >= 0 && rr < bt.nrows + 2 && (unsigned)yy < (unsigned)p.fS && (unsigned)xx < (unsigned)p.fS;
                    const int dst = OFF_RAW + buf * kRawBytes + (lw + 4 * j) * 256;
                    // (elements outside the frame -- the first conv's zero padding -- and beyond the tile are ZEROED by an ordinary LDS
                    // store of the lanes concerned; the LDS-DMA runs with those lanes switched off)
                    if (!ok) *(float *)(smem + dst + lane * 4) = 0.f;
                    if (ok) ws_glds4(xb, (unsigned)(r_off[j] * 4), __builtin_amdgcn_readfirstlane(smem_lds + dst));
                }
            }
        };
        // FUSE: the first conv of NS slices of 32 patch pixels (slices lw, lw + 4, ... of the batch, row-major over nrows x 34) -> ring
        // rows.  The slices of a wave are processed TOGETHER, phase by phase (all raw reads, all MFMAs, all conversions and ring
        // writes): a producer wave is alone on its SIMD beside a consumer that multiplies, so nothing else hides its LDS and
        // matrix-pipe latencies (one slice at a time: ~2950 cycles per slice, the producers were the critical path of the launch).
        // The accumulators start at the bias, held in registers for the whole launch.
        // (ReLU on the packed pair as a signed 16-bit maximum with 0, see the consumers' pack_relu: exact for finite values)
        typedef short ws_s2p __attribute__((ext_vector_type(2)));
        auto fc_relu = [&](float a, float b) -> unsigned {
            return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(ws_s2p, pack2<DT>(a, b)), ws_s2p{0, 0}));
        };
        [[maybe_unused]] f32x16 fbias[2];
        auto load_first_bias = [&]() {
            if constexpr (FUSE) {
                const float *b1 = (const float *)(smem + OFF_FB);
#pragma unroll
                for (int ci = 0; ci < 2; ++ci)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 bv = *(const f32x4 *)(b1 + ci * 32 + 8 * g + 4 * hh);
#pragma unroll
                        for (int j = 0; j < 4; ++j) fbias[ci][4 * g + j] = bv[j];
                    }
            }
        };
        auto first_conv_slices = [&](const Batch &bt, auto nsc, int buf) {
            if constexpr (FUSE) {
                constexpr int NS = decltype(nsc)::value;
                const float *raw = (const float *)(smem + OFF_RAW + buf * kRawBytes);
                float xv[NS][16];
                int pqs[NS];
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                    const int pq = (lw + 4 * i) * 32 + r32;
                    pqs[i] = pq;
                    const bool valid = pq < bt.nrows * kRingCols;
                    const int prow = pq / kRingCols, pcol = pq - prow * kRingCols;
                    const int porg = valid ? prow * kRawCols + pcol : 0;
#pragma unroll
                    for (int s2 = 0; s2 < 16; ++s2) xv[i][s2] = raw[porg + koff1[s2]];
                }
                f32x16 a1[NS][2];
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int i = 0; i < NS; ++i) {
                        u32x4 xq;
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) xq[jj] = pack2<DT>(xv[i][8 * ks + 2 * jj], xv[i][8 * ks + 2 * jj + 1]);
#pragma unroll
                        for (int ci = 0; ci < 2; ++ci) {
                            if (ks == 0) a1[i][ci] = fbias[ci];
                            MmaW<DT>::run(wq1[ci][ks], xq, a1[i][ci]);
                        }
                    }
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                    const int pq = pqs[i];
                    const bool valid = pq < bt.nrows * kRingCols;
                    const int prow = pq / kRingCols, pcol = pq - prow * kRingCols;
                    // outside the frame the NEXT conv pads with zeros (not with the first conv evaluated out there)
                    const bool inimg = valid && (unsigned)(bt.yf + prow) < (unsigned)p.fS && (unsigned)(bt.x0 - 1 + pcol) < (unsigned)p.fS;
                    int sl_ = bt.s0 + prow;
                    sl_ = sl_ >= kRingRows ? sl_ - kRingRows : sl_;
                    const int q = sl_ * kRingCols + pcol;
                    char *row = smem + OFF_RING + q * 128 + 8 * hh;
                    const int sw = (q >> 1) & 7;
                    const unsigned keep = inimg ? 0xFFFFFFFFu : 0u;      // (one AND per packed pair; no second code path for border slices)
                    if (valid) {
#pragma unroll
                        for (int ci = 0; ci < 2; ++ci)
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                const unsigned w0 = fc_relu(a1[i][ci][4 * g], a1[i][ci][4 * g + 1]) & keep;
                                const unsigned w1 = fc_relu(a1[i][ci][4 * g + 2], a1[i][ci][4 * g + 3]) & keep;
                                *(uint2 *)(row + (((4 * ci + g) ^ sw) << 4)) = make_uint2(w0, w1);
                            }
                    }
                }
            }
        };
        // plain layers: the batch's rows by LDS-DMA: five pieces per row (4 x 8 pixels + 1 x 2), pieces of different rows may land in
        // non-adjacent slots (the ring wraps), pixels outside the image read the zero page
        auto dma_rows = [&](const Batch &bt) {
            if constexpr (!FUSE) {
                const int lrow = lane >> 3, pc = lane & 7;
                const int npieces = bt.nrows * 5;
                const char *base = p.in + (size_t)bt.b * p.H * p.W * 128;               // (one image: offsets below stay far under 2^32)
                for (int pi = lw; pi < npieces; pi += 4) {
                    const int row = pi / 5, k = pi - row * 5;
                    const int pcol = k * 8 + lrow;
                    int sl_ = bt.s0 + row;
                    sl_ = sl_ >= kRingRows ? sl_ - kRingRows : sl_;
                    const int q = sl_ * kRingCols + pcol;
                    const int y = bt.yf + row, x = bt.x0 - 1 + pcol;
                    const bool inrow = pcol < kRingCols;
                    const bool ok = inrow && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
                    const int dst = OFF_RING + (sl_ * kRingCols + k * 8) * 128;
                    if (inrow && !ok) *(u32x4 *)(smem + dst + lane * 16) = u32x4{0u, 0u, 0u, 0u};     // the conv's zero padding
                    if (ok) ws_glds16(base, (unsigned)((y * p.W + x) * 128 + ((pc ^ ((q >> 1) & 7)) << 4)), __builtin_amdgcn_readfirstlane(smem_lds + dst));
                }
            }
        };
        // batch j of a unit whose first ring slot is us0
        auto batch_of = [&](const Unit &un, int j, int us0) -> Batch {
            Batch bt;
            bt.b = un.b; bt.x0 = un.x0;
            const int rel = j == 0 ? 0 : (j == 1 ? 8 : 8 * j - 6);
            bt.nrows = j == 1 ? 2 : 8;
            bt.yf = un.y0 - 1 + rel;
            bt.s0 = (us0 + rel) % kRingRows;
            return bt;
        };

        Unit un = decode(u0);
        int us0 = 0;                                    // ring slot of the unit's first row (runs on across units)
        load_weights(un.nt);
        if constexpr (FUSE) issue_raw(batch_of(un, 0, us0), 0);
        land_raw(0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                   // (a) weights, bias and the first raw tile landed
        write_first_bias();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                   // (b) the first conv's bias is in place
        load_first_bias();
        if (TDRN_WS_PRIO == 1) __builtin_amdgcn_s_setprio(2);
        int n = 0, prev_nt = un.nt;
        WS_STAMP_DECL
        for (int u = u0; u < u1; ++u) {
            const Unit nx = u + 1 < u1 ? decode(u + 1) : un;
            const int nx_s0 = (us0 + 8 * un.T + 2) % kRingRows;
            for (int j = 0; j <= un.T; ++j) {
                const Batch bt = batch_of(un, j, us0);
                if constexpr (FUSE) {
                    if (j < un.T) issue_raw(batch_of(un, j + 1, us0), (n + 1) & 1);
                    else if (u + 1 < u1) issue_raw(batch_of(nx, 0, nx_s0), (n + 1) & 1);
                }
                if (j == 1 && un.nt != prev_nt) {
                    // the cout tile changes: nobody reads the weights in this period (the previous unit's last tile finished its reads
                    // before the barrier that ended the period of batch 0).  FUSE layers have ONE cout tile: no raw tile is in flight here.
                    load_weights(un.nt);
                    prev_nt = un.nt;
                }
                if constexpr (!(TDRN_WS_ABLATE & 1)) {
                    if constexpr (FUSE) {
                        // 8 rows: 272 pixels = 9 slices (wave 0 three, the others two); 2 rows: 68 pixels = 3 slices (waves 0 - 2 one each)
                        if (bt.nrows == 8) {
                            if (lw == 0) first_conv_slices(bt, std::integral_constant<int, 3>{}, n & 1);
                            else first_conv_slices(bt, std::integral_constant<int, 2>{}, n & 1);
                        } else if (lw < 3) {
                            first_conv_slices(bt, std::integral_constant<int, 1>{}, n & 1);
                        }
                    } else {
                        dma_rows(bt);
                    }
                }
                land_raw((n + 1) & 1);                  // (U8: the next batch's tile; nobody has read that buffer since the barrier before last)
                WS_STAMP(0);                            // production (issue / first conv)
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                WS_STAMP(1);                            // landing of the DMA pieces
                __builtin_amdgcn_s_barrier();
                WS_STAMP(2);                            // barrier (waiting for the consumers)
                ++n;
            }
            us0 = nx_s0;
            un = nx;
        }
        __builtin_amdgcn_s_barrier();                   // the consumers' last tile
        WS_STAMP_FLUSH;
        return;
    }

    // ======================================= CONSUMERS =======================================
    const int cw = wave;                                // tile rows 2 cw, 2 cw + 1 (pixel fragment pt = row 2 cw + pt, lane r32 = x)
    f32x16 acc[2][2];                                   // [cout tile ci][pixel fragment pt]
    // weight fragment bases (tile-invariant): row ci*32 + r32 of a tap, 16-byte chunk (2 kk + hh) ^ ((r32 >> 1) & 7)
    unsigned wa[2];
#pragma unroll
    for (int ci = 0; ci < 2; ++ci) wa[ci] = (unsigned)(OFF_W + (ci * 32 + r32) * 128 + ((hh ^ ((r32 >> 1) & 7)) << 4));
    // ReLU on the PACKED 16-bit pair as a signed 16-bit maximum with 0 (v_pk_max_i16: a negative value -- and -0 -- has its sign bit set,
    // i.e. is a negative integer; finite values come out exactly as max(v, 0) before the conversion does); no ReLU: maximum with -32768
    typedef short ws_s2 __attribute__((ext_vector_type(2)));
    const ws_s2 relu_lo = p.relu ? ws_s2{0, 0} : ws_s2{(short)-32768, (short)-32768};
    auto pack_relu = [&](float a, float b) -> unsigned {
        return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(ws_s2, pack2<DT>(a, b)), relu_lo));
    };

    // ---- epilogue in two halves ------------------------------------------------------------------------------------------------
    // A lane holds, per pixel fragment, 8 groups (ci, g) of 4 couts: group c = 4 ci + g is the 16-byte chunk c of the pixel's 128-byte
    // row, of which lane r32 (hh = 0) has the first 8 bytes and lane r32 + 32 the second.  (1) Right behind a tile's last MFMAs the
    // accumulators are CONVERTED into packed 16-bit registers (full-resolution rows: 2 x 8 groups; pooled: the fused MaxPool2d(2,2) on
    // the raw accumulators first -- max commutes with the monotonic bias + ReLU; the partner row is my other pixel fragment, the partner
    // column lane ^ 1, a DPP quad permute -- then 8 groups), and the accumulators are free for the next tile.  (2) The STORES run as
    // independent ops between the MFMA groups of the next tile's first steps: v_permlane32_swap on a PAIR of chunks (c, c + 1) gives the
    // lower lane all 16 bytes of chunk c and the upper lane all of chunk c + 1 (cdna_hip_programming.md T21), one dwordx4 store per pair,
    // 32 contiguous bytes per pixel and instruction, no LDS round trip.  Pooled rows: both lanes of a column pair hold the same maxima and
    // store the same bytes to the same pooled pixel.  (A consumer wave is alone on its SIMD beside a producer: the first version's epilogue
    // -- LDS staging rounds, ds_bpermute -- ran on its own for 4150 of a tile's 9700 cycles.)
    uint2 pk_out[MODE & 1 ? 2 : 1][8], pk_pool[8];     // packed groups of the finished tile: [pixel fragment][chunk c]
    char *pd_out[2] = {nullptr, nullptr}, *pd_pool = nullptr;      // ... and where its rows go (null: nothing pending)
    auto convert = [&]() {
        if constexpr (TDRN_WS_ABLATE & 4) return;
        if constexpr (MODE & 1) {
#pragma unroll
            for (int pt = 0; pt < 2; ++pt)
#pragma unroll
                for (int ci = 0; ci < 2; ++ci)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        pk_out[pt][4 * ci + g] = make_uint2(pack_relu(acc[ci][pt][4 * g], acc[ci][pt][4 * g + 1]), pack_relu(acc[ci][pt][4 * g + 2], acc[ci][pt][4 * g + 3]));
        }
        if constexpr (MODE & 2) {
            // MaxPool2d(2,2) + ReLU on the PACKED values, as signed 16-bit maxima: rounding to 16 bits is monotonic, so it commutes with the
            // maximum; after max(., 0) every candidate is a non-negative 16-bit float, whose bit patterns order like integers -- and
            // max(max(a, b), 0) = max(max(a, 0), max(b, 0)).  (Pooled layers always carry a ReLU: the launcher declines otherwise.)
            // Only compiler-visible instructions here: the accumulators come straight out of the MFMAs, and hipcc pads the MFMA -> VALU
            // and VALU -> DPP wait states for its own instructions only, not around inline asm.
#pragma unroll
            for (int ci = 0; ci < 2; ++ci)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    unsigned w2[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const ws_s2 zero = {0, 0};
                        const ws_s2 u0 = __builtin_bit_cast(ws_s2, pack2<DT>(acc[ci][0][4 * g + 2 * h], acc[ci][0][4 * g + 2 * h + 1]));
                        const ws_s2 u1 = __builtin_bit_cast(ws_s2, pack2<DT>(acc[ci][1][4 * g + 2 * h], acc[ci][1][4 * g + 2 * h + 1]));
                        const ws_s2 m = __builtin_elementwise_max(__builtin_elementwise_max(u0, u1), zero);
                        const ws_s2 nb = __builtin_bit_cast(ws_s2, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0xB1, 0xF, 0xF, true));
                        w2[h] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(m, nb));
                    }
                    pk_pool[4 * ci + g] = make_uint2(w2[0], w2[1]);
                }
        }
    };
    // op 0-7: full-resolution rows (pixel fragment op / 4, chunk pair op % 4); ops 8-9: pooled rows -- both lanes of a column pair hold the
    // same maxima, so the even lane stores pair 2 (op - 8) and the odd lane pair 2 (op - 8) + 1 of the same pooled pixel: TWO stores per
    // tile and wave instead of four with the odd lanes idle (a store costs the issuing wave ~150 cycles: 8 % of the launch with four)
    auto store_op = [&](auto opc) {
        constexpr int OP = decltype(opc)::value;
        if constexpr (TDRN_WS_ABLATE & 4) return;
        uint2 a, b;
        char *dst;
        if constexpr (OP < 8) {
            constexpr int pr = OP & 3, pt = (MODE & 1) ? (OP >> 2) : 0;
            a = pk_out[pt][2 * pr]; b = pk_out[pt][2 * pr + 1];
            dst = pd_out[pt] + pr * 32 + hh * 16;
        } else {
            constexpr int j = OP - 8;
            const bool odd = r32 & 1;
            a.x = odd ? pk_pool[4 * j + 2].x : pk_pool[4 * j].x;     a.y = odd ? pk_pool[4 * j + 2].y : pk_pool[4 * j].y;
            b.x = odd ? pk_pool[4 * j + 3].x : pk_pool[4 * j + 1].x; b.y = odd ? pk_pool[4 * j + 3].y : pk_pool[4 * j + 1].y;
            dst = pd_pool + (2 * j + (odd ? 1 : 0)) * 32 + hh * 16;
        }
        auto rx = __builtin_amdgcn_permlane32_swap(a.x, b.x, false, false);
        auto ry = __builtin_amdgcn_permlane32_swap(a.y, b.y, false, false);
        if constexpr (TDRN_WS_ABLATE & 8) {             // (diagnostics: everything but the store instruction itself; p.B < 0 never holds)
            if (p.B < 0) *(u32x4 *)dst = u32x4{rx[0], ry[0], rx[1], ry[1]};
        } else {
            *(u32x4 *)dst = u32x4{rx[0], ry[0], rx[1], ry[1]};
        }
    };
    constexpr int NOPS = (MODE & 1 ? 8 : 0) + (MODE & 2 ? 2 : 0);
    auto op_of = [](int i) constexpr -> int { return (MODE & 1) ? i : 8 + i; };      // i-th op of this MODE
    auto flush = [&]() {
        if (!pd_pool && !pd_out[0]) return;
        store_op(std::integral_constant<int, op_of(0)>{}); store_op(std::integral_constant<int, op_of(1)>{});
        if constexpr (NOPS > 2) {
            store_op(std::integral_constant<int, op_of(2)>{}); store_op(std::integral_constant<int, op_of(3)>{});
            store_op(std::integral_constant<int, op_of(4)>{}); store_op(std::integral_constant<int, op_of(5)>{});
            store_op(std::integral_constant<int, op_of(6)>{}); store_op(std::integral_constant<int, op_of(7)>{});
        }
    };

    WS_STAMP_DECL
    // the bias as the C operand of a tile's first MFMAs (two accumulator tiles' worth of registers, reloaded with the weights)
    f32x16 cbias[2];
    auto load_cbias = [&]() {
#pragma unroll
        for (int ci = 0; ci < 2; ++ci)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 bv = *(const f32x4 *)(smem + OFF_BIAS + (ci * 32 + 8 * g + 4 * hh) * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) cbias[ci][4 * g + j] = bv[j];
            }
    };
    // fragment base addresses of a tile whose first row (of mine) sits in ring slot s: (row i of 4, column shift kx)
    unsigned pa[4][3];
    auto tile_addresses = [&](int s) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int q = s * kRingCols + r32 + kx;
                pa[i][kx] = (unsigned)(OFF_RING + q * 128 + ((hh ^ ((q >> 1) & 7)) << 4));
            }
            s = s + 1 == kRingRows ? 0 : s + 1;
        }
    };
    u32x4 wf[3][2], pf[3][2];                           // operand fragments, two steps ahead of their MFMAs
    auto LD = [&](int s, int set) {
        const int tap = s >> 2, kk = s & 3, ky = tap / 3, kx = tap - 3 * ky;
        const unsigned kxor = (unsigned)(kk << 5);
#pragma unroll
        for (int ci = 0; ci < 2; ++ci) wf[set][ci] = *(const u32x4 *)(smem + ((wa[ci] ^ kxor) + tap * 8192));
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
            // (the XOR is recomputed at every use on purpose: hipcc otherwise keeps all 48 (row, shift, slice) addresses of a tile live --
            // they recur across taps -- and spills into the MFMA loop, with an `s_waitcnt vmcnt(0)` in front of every reload)
            unsigned a = pa[pt + ky][kx] ^ kxor;
            asm volatile("" : "+v"(a));
            pf[set][pt] = *(const u32x4 *)(smem + a);
        }
    };
    bool prefetched = false;                            // the first two steps' fragments of the coming tile are already on their way
    // one tile; HAVE_OLD: the previous tile's packed rows are pending, their store ops run after MFMA groups 1, 3, 5, ...
    // next_slot >= 0: the NEXT period is a tile too, its first row of mine in that ring slot -- its addresses and first fragments are
    // fetched right behind this tile's barrier, under the last step's MFMAs and the conversion
    auto tile = [&](auto oldc, const Unit &un, int k, int us0, int next_slot) {
        constexpr bool HAVE_OLD = decltype(oldc)::value;
        WS_STAMP(3);                                    // between tiles (periods without a tile, unit bookkeeping)
        constexpr bool compute = !(TDRN_WS_ABLATE & 2);
        if (!prefetched) {
            tile_addresses((us0 + 8 * k + 2 * cw) % kRingRows);
            if (compute) { LD(0, 0); LD(1, 1); }
        }
        auto step = [&](auto sc) {
            constexpr int s = decltype(sc)::value;
            if (compute && s + 2 < 36) LD(s + 2, (s + 2) % 3);
            __builtin_amdgcn_sched_barrier(0);
            if (s == 35) {
                // every ring / weight read of this tile has returned (the last fragments arrived for step 34's wait at the
                // latest): the producers may overwrite this tile's first rows; the last step multiplies behind the barrier
                __builtin_amdgcn_s_waitcnt(0xC07F);
                WS_STAMP(0);                            // steps 0..34
                __builtin_amdgcn_s_barrier();
                WS_STAMP(1);                            // barrier (waiting for the producers)
                if (next_slot >= 0) {                   // (wave-uniform)
                    tile_addresses(next_slot);
                    if (compute) { LD(0, 0); LD(1, 1); }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (compute) {
                if constexpr (s == 0) {                 // the accumulators start at the bias: it is the first MFMAs' C operand
#pragma unroll
                    for (int ci = 0; ci < 2; ++ci)
#pragma unroll
                        for (int pt = 0; pt < 2; ++pt) {
                            acc[ci][pt] = cbias[ci];
                            MmaW<DT>::run(wf[0][ci], pf[0][pt], acc[ci][pt]);
                        }
                } else {
#pragma unroll
                    for (int ci = 0; ci < 2; ++ci)
#pragma unroll
                        for (int pt = 0; pt < 2; ++pt) MmaW<DT>::run(wf[s % 3][ci], pf[s % 3][pt], acc[ci][pt]);
                }
            } else if constexpr (s == 0) {
#pragma unroll
                for (int ci = 0; ci < 2; ++ci)
#pragma unroll
                    for (int pt = 0; pt < 2; ++pt) acc[ci][pt] = cbias[ci];
            }
            __builtin_amdgcn_sched_barrier(0);
            // the previous tile's store op (s - 1) / 2 in the shadow of this step's MFMAs
            if constexpr (HAVE_OLD && (s & 1) && (s >> 1) < NOPS) {
                store_op(std::integral_constant<int, op_of(s >> 1)>{});
                __builtin_amdgcn_sched_barrier(0);
            }
        };
#define WS_S4(b) step(std::integral_constant<int, b>{}); step(std::integral_constant<int, b + 1>{}); step(std::integral_constant<int, b + 2>{}); step(std::integral_constant<int, b + 3>{});
        WS_S4(0) WS_S4(4) WS_S4(8) WS_S4(12) WS_S4(16) WS_S4(20) WS_S4(24) WS_S4(28) WS_S4(32)
#undef WS_S4
        prefetched = next_slot >= 0;
        // this tile is now the pending one: its packed rows, and where they go
        convert();
        const int n0 = un.nt * 64;
        const int ty = un.y0 + 8 * k + 2 * cw;          // image row of my first pixel fragment
        if constexpr (MODE & 1) {
#pragma unroll
            for (int pt = 0; pt < 2; ++pt)
                pd_out[pt] = p.out + ((((size_t)un.b * p.H + ty + pt) * p.W + un.x0 + r32) * p.Cs + n0) * 2;
        }
        if constexpr (MODE & 2)
            pd_pool = p.out_pool + (((size_t)(((size_t)un.b * p.H + ty) >> 1) * (p.W >> 1) + ((un.x0 + r32) >> 1)) * p.Cs + n0) * 2;
        WS_STAMP(2);                                    // step 35 + conversion
    };
    int ntile = 0;
    auto run_tile = [&](const Unit &tu, int tk, int ts0) {
        // the next period is a tile as well iff this one is not its unit's last (the period behind a unit's last tile produces the next
        // unit's rows 8 and 9; the last unit's last tile has no successor)
        const int next_slot = tk + 1 < tu.T ? (ts0 + 8 * (tk + 1) + 2 * cw) % kRingRows : -1;
        if (ntile == 0) tile(std::false_type{}, tu, tk, ts0, next_slot);
        else tile(std::true_type{}, tu, tk, ts0, next_slot);
        ++ntile;
    };

    Unit un = decode(u0), pv = un;
    int us0 = 0, pv_s0 = 0, prev_nt = un.nt;
    // the consumers are the critical path (the producers wait a third of every period at the barrier): their instructions go first
    if (TDRN_WS_PRIO == 2) __builtin_amdgcn_s_setprio(2);
    load_weights(un.nt);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                       // (a)
    write_first_bias();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                       // (b)
    load_cbias();
    // periods: unit u, batch j = 0 .. T (u == u1: the one period of the last unit's last tile)
    for (int u = u0; u <= u1; ++u) {
        const int nper = u < u1 ? un.T + 1 : 1;
        for (int j = 0; j < nper; ++j) {
            const bool prev_tile = j == 0 && u > u0, own_tile = j >= 2;
            if (prev_tile || own_tile) {
                run_tile(prev_tile ? pv : un, prev_tile ? pv.T - 1 : j - 2, prev_tile ? pv_s0 : us0);
            } else {
                const bool reload = j == 1 && un.nt != prev_nt;
                if (reload) {
                    load_weights(un.nt);
                    prev_nt = un.nt;
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_s_barrier();
                if (reload) load_cbias();               // (the bias piece landed before wave 0 reached the barrier)
            }
        }
        if (u == u1) break;
        pv = un;
        pv_s0 = us0;
        us0 = (us0 + 8 * un.T + 2) % kRingRows;
        if (u + 1 < u1) un = decode(u + 1);
    }
    flush();                                            // the last tile's rows
    WS_STAMP_FLUSH;
}

// ---------------------------------------------------------------------------------------------
static int g_ws_override = -1;                          // dev harness: -1 = environment, 0 / 1 = forced, 2 = forced also below the size where it pays
void conv_ws_force(int v) { g_ws_override = v; }
int conv_ws_enabled()
{
    if (g_ws_override >= 0) return g_ws_override != 0;
    static int e = -1;
    if (e < 0) { const char *s = getenv("TDRN_CONV_WS"); e = s ? atoi(s) : 1; }
    return e;
}

// the layers this kernel takes over from conv3x3_patch.hip: 16-bit, ONE 64-channel chunk, 2-D geometry in whole 8 x 32 tiles
int ws_conv_supported(const ConvArgs &a)
{
    if (!conv_ws_enabled() || (a.kdisable & 64)) return 0;
    if (a.dtype == TDRN_F32 || a.Cin != 64 || a.Npad % 64 || a.Cout % 64) return 0;       // (Cout = the output tensor's padded channel count)
    if (a.W % 32 || a.H % 8) return 0;
    return patch_conv_supported(a) != 0;
}

int launch_conv3x3_ws(const ConvArgs &a, void *out_pool, hipStream_t s)
{
    if (!ws_conv_supported(a)) return TDRN_E_UNSUPPORTED;
    if (out_pool && ((a.H & 1) || (a.W & 1) || !a.relu)) return TDRN_E_UNSUPPORTED;      // (the pooled epilogue's integer maxima assume the ReLU)
    if (!a.out && !out_pool) return TDRN_E_ARG;
    WsParams p;
    p.in = (const char *)a.in; p.w = (const char *)a.w; p.zero = (const char *)a.zero_page; p.bias = a.bias;
    p.out = (char *)a.out; p.out_pool = (char *)out_pool;
    p.B = a.B; p.H = a.H; p.W = a.W; p.Cout = a.Cout; p.Cs = (int)a.o_cs; p.Ktot = 9 * a.Cin;
    p.relu = a.relu;
    p.SX = a.W / 32; p.TY = a.H / 8; p.NT = a.Cout / 64;      // (cout tiles of all-padding rows beyond Cout are not computed)
    p.fx = a.fuse_x; p.fw = a.fuse_w; p.fb = a.fuse_b; p.fS = a.H; p.fCout = a.fuse_cout;
    p.fx8 = a.fuse_x8; p.fmean[0] = a.fuse_mean[0]; p.fmean[1] = a.fuse_mean[1]; p.fmean[2] = a.fuse_mean[2];
    if (a.fuse_x && a.fuse_x8) return TDRN_E_ARG;
    if (a.fuse_x || a.fuse_x8) {
        // the fused variant keeps LDS for the raw tiles instead of a full staging strip: pooled output only, one cout tile
        if (a.out || !out_pool || p.NT != 1 || a.H != a.W || a.fuse_cout > 64) return TDRN_E_UNSUPPORTED;
    }
    int grid = 256;
    if (a.max_wgs > 0 && grid > (a.max_wgs / 8) * 8) grid = (a.max_wgs / 8) * 8;
    if (grid < 8) return TDRN_E_UNSUPPORTED;
    // Rows per unit: a unit of T tiles costs ~T + 0.6 tile times (its first 10 rows are produced before its first tile can start, 8 of
    // them under the previous unit's last tile); the launch takes ceil(units / grid) units per workgroup.  Depends on the geometry
    // and the batch only through the unit count -- and the choice changes no output bit.
    {
        double best = 1e30;
        int best_t = p.TY;
        for (int t = p.TY; t >= 2; --t) {
            const int nseg = cdiv(p.TY, t);
            const long long units = (long long)p.NT * a.B * p.SX * nseg;
            const double cost = (double)((units + grid - 1) / grid) * (t + 0.6);
            if (cost < best - 1e-9) { best = cost; best_t = t; }
        }
        if (p.TY < 2) best_t = p.TY;
        p.TSEG = best_t;
        p.NSEG = cdiv(p.TY, p.TSEG);
    }
    const long long units = (long long)p.NT * a.B * p.SX * p.NSEG;
    if (units <= 0) return TDRN_OK;
    if (units >= (1ll << 31)) return TDRN_E_UNSUPPORTED;
    p.units = (int)units;
    if (p.units < grid) grid = ((p.units + 7) / 8) * 8;
    // below ~3/4 of the chip conv3x3_patch.hip's independent 256-pixel items spread better (small batches)
    if (p.units < 192 && g_ws_override != 2) return TDRN_E_UNSUPPORTED;     // (2: the dev harness runs small cases through it)
#ifdef TDRN_WS_STAMP
    static unsigned *stamps = nullptr;
    if (!stamps) TDRN_HIP_TRY(hipMalloc((void **)&stamps, 256 * 8 * 4 * sizeof(unsigned)));
    TDRN_HIP_TRY(hipMemsetAsync(stamps, 0, 256 * 8 * 4 * sizeof(unsigned), s));
    p.stamps = stamps;
    struct Report {
        const WsParams &p; hipStream_t s; int grid;
        ~Report()
        {
            static unsigned host[256 * 8 * 4];
            (void)hipStreamSynchronize(s);
            (void)hipMemcpy(host, p.stamps, sizeof(host), hipMemcpyDeviceToHost);
            double c[4] = {0, 0, 0, 0}, l[4] = {0, 0, 0, 0};
            int nc = 0, nl = 0;
            for (int b = 0; b < grid; ++b)
                for (int w = 0; w < 8; ++w) {
                    const unsigned *v = host + (b * 8 + w) * 4;
                    if (v[0] + v[1] + v[2] + v[3] == 0) continue;
                    if (w < 4) { for (int k = 0; k < 4; ++k) c[k] += v[k]; ++nc; }
                    else { for (int k = 0; k < 4; ++k) l[k] += v[k]; ++nl; }
                }
            if (!nc || !nl) return;
            const double tiles = (double)p.units * p.TSEG / grid;
            fprintf(stderr, "ws_stamp H%d W%d Cout%d fuse%d TSEG%d units%d tiles/CU~%.1f | consumer cyc/wave: steps0-34 %.0f barrier %.0f step35+epilogue %.0f between %.0f"
                            " | producer cyc/wave: produce %.0f landing %.0f barrier %.0f\n",
                    p.H, p.W, p.Cout, (p.fx || p.fx8) ? 1 : 0, p.TSEG, p.units, tiles, c[0] / nc, c[1] / nc, c[2] / nc, c[3] / nc, l[0] / nl, l[1] / nl, l[2] / nl);
        }
    } report{p, s, grid};
#endif
#define WS_LAUNCH(DT)                                                                                              \
    do {                                                                                                           \
        if (a.fuse_x8) hipLaunchKernelGGL((conv3x3_ws_kernel<DT, true, 2, true>), dim3(grid), dim3(512), 0, s, p); \
        else if (a.fuse_x) hipLaunchKernelGGL((conv3x3_ws_kernel<DT, true, 2>), dim3(grid), dim3(512), 0, s, p);  \
        else if (mode == 1) hipLaunchKernelGGL((conv3x3_ws_kernel<DT, false, 1>), dim3(grid), dim3(512), 0, s, p); \
        else hipLaunchKernelGGL((conv3x3_ws_kernel<DT, false, 2>), dim3(grid), dim3(512), 0, s, p);               \
    } while (0)
    const int mode = (a.out ? 1 : 0) | (out_pool ? 2 : 0);
    if (mode == 3) return TDRN_E_UNSUPPORTED;            // (both outputs at once: no plan asks for it; conv3x3_patch.hip takes such a launch)
    // Full-resolution outputs (conv2_1) are eight 1-KiB stores per tile and consumer wave, ~150 cycles of the wave's time each, against two
    // for a pooled tile: measured in the net 127-128 us against 123-126 us on conv3x3_patch.hip (whose two consumer waves per SIMD hide each
    // other's stores) -- so this kernel takes such a layer only when asked to (TDRN_CONV_WS=2, the dev harness).  Same bits either way.
    {
        static int e = -1;
        if (e < 0) { const char *v = getenv("TDRN_CONV_WS"); e = v ? atoi(v) : 1; }
        if (mode == 1 && g_ws_override != 2 && e < 2) return TDRN_E_UNSUPPORTED;
    }
    if (a.dtype == TDRN_BF16) WS_LAUNCH(bf16_t);
    else WS_LAUNCH(f16_t);
#undef WS_LAUNCH
    return hip_status(hipGetLastError());
}

}  // namespace tdrn
