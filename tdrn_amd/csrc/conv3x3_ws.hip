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

template <typename DT, bool FUSE>
__global__ __launch_bounds__(512, 2) void conv3x3_ws_kernel(const WsParams p)
{
    static_assert(sizeof(DT) == 2, "16-bit element types only");
    constexpr int OFF_W = 0;
    constexpr int OFF_RING = OFF_W + kWBytes;
    constexpr int OFF_BIAS = OFF_RING + kRingBytes;     // 64 floats
    constexpr int OFF_FB = OFF_BIAS + 256;              // FUSE: the first conv's bias, 64 floats
    constexpr int OFF_RAW = OFF_FB + 256;               // FUSE: two raw halo tiles
    constexpr int OFF_STG = OFF_RAW + (FUSE ? 2 * kRawBytes : 0);
    constexpr int SROWS = FUSE ? 4 : 16;                // staging rows (pixels) per epilogue round and consumer wave
    constexpr int SSTRIDE = 128 + 16;
    constexpr int LDS = OFF_STG + 4 * SROWS * SSTRIDE;
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
        auto issue_raw = [&](const Batch &bt, int buf) {
            if constexpr (FUSE) {
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
        // FUSE: the first conv of 32 patch pixels (slice sl of the batch, row-major over nrows x 34) -> ring rows
        auto first_conv_slice = [&](const Batch &bt, int sl, int buf) {
            if constexpr (FUSE) {
                const int pq = sl * 32 + r32;
                const bool valid = pq < bt.nrows * kRingCols;
                const int prow = pq / kRingCols, pcol = pq - prow * kRingCols;
                // outside the frame the NEXT conv pads with zeros (not with the first conv evaluated out there)
                const bool inimg = valid && (unsigned)(bt.yf + prow) < (unsigned)p.fS && (unsigned)(bt.x0 - 1 + pcol) < (unsigned)p.fS;
                const float *raw = (const float *)(smem + OFF_RAW + buf * kRawBytes);
                const float *b1 = (const float *)(smem + OFF_FB);
                const int porg = valid ? prow * kRawCols + pcol : 0;
                float xv[16];
#pragma unroll
                for (int s2 = 0; s2 < 16; ++s2) xv[s2] = raw[porg + koff1[s2]];
                f32x16 a1[2];
#pragma unroll
                for (int ci = 0; ci < 2; ++ci)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 bv = *(const f32x4 *)(b1 + ci * 32 + 8 * g + 4 * hh);
#pragma unroll
                        for (int j = 0; j < 4; ++j) a1[ci][4 * g + j] = bv[j];
                    }
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    u32x4 xq;
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) xq[jj] = pack2<DT>(xv[8 * ks + 2 * jj], xv[8 * ks + 2 * jj + 1]);
#pragma unroll
                    for (int ci = 0; ci < 2; ++ci) MmaW<DT>::run(wq1[ci][ks], xq, a1[ci]);
                }
                if (valid) {
                    int sl_ = bt.s0 + prow;
                    sl_ = sl_ >= kRingRows ? sl_ - kRingRows : sl_;
                    const int q = sl_ * kRingCols + pcol;
                    char *row = smem + OFF_RING + q * 128 + 8 * hh;
                    const int sw = (q >> 1) & 7;
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
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                   // (a) weights, bias and the first raw tile landed
        write_first_bias();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                   // (b) the first conv's bias is in place
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
                        const int nsl = (bt.nrows * kRingCols + 31) / 32;
                        for (int sl = lw; sl < nsl; sl += 4) first_conv_slice(bt, sl, n & 1);
                    } else {
                        dma_rows(bt);
                    }
                }
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
    char *stg = smem + OFF_STG + cw * SROWS * SSTRIDE;
    f32x16 acc[2][2];                                   // [cout tile ci][pixel fragment pt]
    // weight fragment bases (tile-invariant): row ci*32 + r32 of a tap, 16-byte chunk (2 kk + hh) ^ ((r32 >> 1) & 7)
    unsigned wa[2];
#pragma unroll
    for (int ci = 0; ci < 2; ++ci) wa[ci] = (unsigned)(OFF_W + (ci * 32 + r32) * 128 + ((hh ^ ((r32 >> 1) & 7)) << 4));

    WS_STAMP_DECL
    auto tile = [&](const Unit &un, int k, int us0) {
        WS_STAMP(3);                                    // between tiles (periods without a tile, unit bookkeeping)
        // ring slots of tile rows 2 cw .. 2 cw + 3 (rel rows 8k + 2cw + i), and the fragment base address of (row i, column shift kx)
        unsigned pa[4][3];
        {
            int s = (us0 + 8 * k + 2 * cw) % kRingRows;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int q = s * kRingCols + r32 + kx;
                    pa[i][kx] = (unsigned)(OFF_RING + q * 128 + ((hh ^ ((q >> 1) & 7)) << 4));
                }
                s = s + 1 == kRingRows ? 0 : s + 1;
            }
        }
        // accumulators start at the bias
#pragma unroll
        for (int ci = 0; ci < 2; ++ci)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 bv = *(const f32x4 *)(smem + OFF_BIAS + (ci * 32 + 8 * g + 4 * hh) * 4);
#pragma unroll
                for (int pt = 0; pt < 2; ++pt)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[ci][pt][4 * g + j] = bv[j];
            }
        // 36 steps (tap-major, 4 K-slices per tap) of 4 reads + 4 MFMAs; fragments two steps ahead in registers
        u32x4 wf[3][2], pf[3][2];
        auto LD = [&](int s, int set) {
            const int tap = s >> 2, kk = s & 3, ky = tap / 3, kx = tap - 3 * ky;
            const unsigned kxor = (unsigned)(kk << 5);
#pragma unroll
            for (int ci = 0; ci < 2; ++ci) wf[set][ci] = *(const u32x4 *)(smem + ((wa[ci] ^ kxor) + tap * 8192));
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) pf[set][pt] = *(const u32x4 *)(smem + (pa[pt + ky][kx] ^ kxor));
        };
        auto MMA = [&](int set) {
#pragma unroll
            for (int ci = 0; ci < 2; ++ci)
#pragma unroll
                for (int pt = 0; pt < 2; ++pt) MmaW<DT>::run(wf[set][ci], pf[set][pt], acc[ci][pt]);
        };
        constexpr bool compute = !(TDRN_WS_ABLATE & 2);
        if (compute) { LD(0, 0); LD(1, 1); }
#pragma unroll
        for (int s = 0; s < 36; ++s) {
            if (compute && s + 2 < 36) LD(s + 2, (s + 2) % 3);
            __builtin_amdgcn_sched_barrier(0);
            if (s == 35) {
                // every ring / weight read of this tile has returned (the last fragments arrived for step 34's wait at the
                // latest): the producers may overwrite this tile's first rows; the last step multiplies behind the barrier
                __builtin_amdgcn_s_waitcnt(0xC07F);
                WS_STAMP(0);                            // steps 0..34
                __builtin_amdgcn_s_barrier();
                WS_STAMP(1);                            // barrier (waiting for the producers)
            }
            if (compute) MMA(s % 3);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- epilogue: ReLU, convert, whole-line stores through my staging strip -------------------------------------------
        const int n0 = un.nt * 64;
        const int ty = un.y0 + 8 * k + 2 * cw;          // image row of my first pixel fragment
        auto stage_quad = [&](const f32x16 &t, int ci, int g, int srow) {
            float q4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) q4[j] = p.relu ? fmaxf(t[4 * g + j], 0.f) : t[4 * g + j];
            *(uint2 *)(stg + srow * SSTRIDE + (ci * 32 + 8 * g + 4 * hh) * 2) = make_uint2(pack2<DT>(q4[0], q4[1]), pack2<DT>(q4[2], q4[3]));
        };
        if (p.out && !(TDRN_WS_ABLATE & 4)) {
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) {
#pragma unroll 1
                for (int rd = 0; rd < 32 / SROWS; ++rd) {
                    if (r32 / SROWS == rd) {
#pragma unroll
                        for (int ci = 0; ci < 2; ++ci)
#pragma unroll
                            for (int g = 0; g < 4; ++g) stage_quad(acc[ci][pt], ci, g, r32 % SROWS);
                    }
                    __builtin_amdgcn_s_waitcnt(0xC07F);
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int k2 = 0; k2 < (SROWS * 8 + 63) / 64; ++k2) {
                        const int idx = lane + 64 * k2, row = idx >> 3, ch = idx & 7;
                        if (row < SROWS && n0 + ch * 8 < p.Cout) {
                            const size_t gp = ((size_t)un.b * p.H + ty + pt) * p.W + un.x0 + rd * SROWS + row;
                            *(u32x4 *)(p.out + (gp * p.Cs + n0 + ch * 8) * 2) = *(const u32x4 *)(stg + row * SSTRIDE + ch * 16);
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
        if (p.out_pool && !(TDRN_WS_ABLATE & 4)) {
            // fused MaxPool2d(2,2) on the RAW accumulators (max commutes with the monotonic bias + ReLU applied at staging): the
            // partner row is my other pixel fragment, the partner column lane ^ 1; even-x lanes hold the 16 pooled pixels
            const int PW = p.W >> 1;
#pragma unroll
            for (int ci = 0; ci < 2; ++ci)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float v = fmaxf(acc[ci][0][e], acc[ci][1][e]);
                    v = fmaxf(v, __shfl_xor(v, 1, 64));
                    acc[ci][0][e] = v;
                }
            const bool holder = (r32 & 1) == 0;
            const int prow_l = r32 >> 1;
#pragma unroll 1
            for (int rd = 0; rd < 16 / SROWS; ++rd) {
                if (holder && prow_l / SROWS == rd) {
#pragma unroll
                    for (int ci = 0; ci < 2; ++ci)
#pragma unroll
                        for (int g = 0; g < 4; ++g) stage_quad(acc[ci][0], ci, g, prow_l % SROWS);
                }
                __builtin_amdgcn_s_waitcnt(0xC07F);
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int k2 = 0; k2 < (SROWS * 8 + 63) / 64; ++k2) {
                    const int idx = lane + 64 * k2, row = idx >> 3, ch = idx & 7;
                    if (row < SROWS && n0 + ch * 8 < p.Cout) {
                        const int pl = rd * SROWS + row;                  // pooled pixel of my row pair
                        const size_t gpool = (size_t)(((size_t)un.b * p.H + ty) >> 1) * PW + ((un.x0 >> 1) + pl);
                        *(u32x4 *)(p.out_pool + (gpool * p.Cs + n0 + ch * 8) * 2) = *(const u32x4 *)(stg + row * SSTRIDE + ch * 16);
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        WS_STAMP(2);                                    // step 35 + epilogue
    };

    Unit un = decode(u0), pv = un;
    int us0 = 0, pv_s0 = 0, prev_nt = un.nt;
    load_weights(un.nt);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                       // (a)
    write_first_bias();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                       // (b)
    // periods: unit u, batch j = 0 .. T (u == u1: the one period of the last unit's last tile)
    for (int u = u0; u <= u1; ++u) {
        const int nper = u < u1 ? un.T + 1 : 1;
        for (int j = 0; j < nper; ++j) {
            const bool prev_tile = j == 0 && u > u0, own_tile = j >= 2;
            if (prev_tile || own_tile) {
                tile(prev_tile ? pv : un, prev_tile ? pv.T - 1 : j - 2, prev_tile ? pv_s0 : us0);
            } else {
                if (j == 1 && un.nt != prev_nt) {
                    load_weights(un.nt);
                    prev_nt = un.nt;
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_s_barrier();
            }
        }
        if (u == u1) break;
        pv = un;
        pv_s0 = us0;
        us0 = (us0 + 8 * un.T + 2) % kRingRows;
        if (u + 1 < u1) un = decode(u + 1);
    }
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
    if (a.dtype == TDRN_F32 || a.Cin != 64 || a.Npad % 64) return 0;
    if (a.W % 32 || a.H % 8) return 0;
    return patch_conv_supported(a) != 0;
}

int launch_conv3x3_ws(const ConvArgs &a, void *out_pool, hipStream_t s)
{
    if (!ws_conv_supported(a)) return TDRN_E_UNSUPPORTED;
    if (out_pool && ((a.H & 1) || (a.W & 1))) return TDRN_E_UNSUPPORTED;
    if (!a.out && !out_pool) return TDRN_E_ARG;
    WsParams p;
    p.in = (const char *)a.in; p.w = (const char *)a.w; p.zero = (const char *)a.zero_page; p.bias = a.bias;
    p.out = (char *)a.out; p.out_pool = (char *)out_pool;
    p.B = a.B; p.H = a.H; p.W = a.W; p.Cout = a.Cout; p.Cs = (int)a.o_cs; p.Ktot = 9 * a.Cin;
    p.relu = a.relu;
    p.SX = a.W / 32; p.TY = a.H / 8; p.NT = a.Npad / 64;
    p.fx = a.fuse_x; p.fw = a.fuse_w; p.fb = a.fuse_b; p.fS = a.H; p.fCout = a.fuse_cout;
    if (a.fuse_x) {
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
                    p.H, p.W, p.Cout, p.fx ? 1 : 0, p.TSEG, p.units, tiles, c[0] / nc, c[1] / nc, c[2] / nc, c[3] / nc, l[0] / nl, l[1] / nl, l[2] / nl);
        }
    } report{p, s, grid};
#endif
#define WS_LAUNCH(DT)                                                                                       \
    do {                                                                                                    \
        if (a.fuse_x) hipLaunchKernelGGL((conv3x3_ws_kernel<DT, true>), dim3(grid), dim3(512), 0, s, p);   \
        else hipLaunchKernelGGL((conv3x3_ws_kernel<DT, false>), dim3(grid), dim3(512), 0, s, p);           \
    } while (0)
    if (a.dtype == TDRN_BF16) WS_LAUNCH(bf16_t);
    else WS_LAUNCH(f16_t);
#undef WS_LAUNCH
    return hip_status(hipGetLastError());
}

}  // namespace tdrn
