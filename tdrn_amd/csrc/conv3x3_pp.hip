// conv3x3_pp.hip -- 3x3 / stride 1 / pad 1 convolution for the Cin >= 128, Cout % 256 == 0 layers of the 16-bit plans
// (conv3_x, conv4_x, the 40x40 TCB convs: ~2/3 of the network's FLOPs) as an ALL-WAVES-COMPUTE direct convolution:
// eight waves per workgroup, one workgroup per CU, every wave owns a 64-pixel x 128-cout accumulator tile (128
// registers), item = 256 pixels x 256 couts.  The two waves of a SIMD alternate ("ping-pong"): while the waves of
// group 0 (waves 0-3, couts 0-127) multiply, the waves of group 1 (waves 4-7, couts 128-255) read their next operand
// fragments from LDS and issue their share of the LDS-DMA stream, and vice versa -- every interval between two
// s_barriers has one wave per SIMD in a bare 16-MFMA segment (512 cycles) and its partner in a load segment.
//
// Why a second kernel next to conv3x3_patch.hip (4 loader + 8 consumer waves, 64 x 64 per consumer): that structure
// tops out at 0.49 of the MFMA peak with NO loads at all (both consumer waves of a SIMD run the same read->multiply
// program in lockstep, profiles/r02_final), and its 3-waves-per-SIMD register budget cannot hold a bigger tile.
// With two waves per SIMD a wave holds 128 accumulators, so a (chunk, tap) step is 32 MFMAs per wave for 24
// ds_read_b128 (1.33 MFMA per read against 1.0) and a step's LDS-DMA volume per MFMA cycle is 18 B/clk against
// 20.5 -- at twice the work between the waits of one step.
//
// What is kept from the patch kernel (and makes this bit-identical to it: same K order per output element):
//   * the activation PATCH of the tile (256 pixels + 1-pixel halo, <= 352 rows of 128 B) is staged once per
//     64-channel chunk and the nine taps are shifted row addresses into it (2 buffers);
//   * the [256 x 128 B] weight slice of every (chunk, tap) step goes through a 2-slot ring, one step ahead; each
//     group stages and reads only its own half of a slot, so the two halves are private rings;
//   * bank swizzle on the SOURCE address of every LDS-DMA piece, raw s_barrier, counted s_waitcnt vmcnt(N) once per
//     step, accumulators start at the bias, epilogue through a wave-private LDS strip -> whole-line NHWC stores with
//     ReLU and the optional MaxPool2d(2,2) fused.  The strip lives in the patch buffer that died with the item's
//     last step (there is no LDS left for a dedicated one).
//
// Interval timeline (g = step, one interval = the span between two consecutive barriers; L = load segment,
// M = 16 MFMAs):
//     interval 4g+0 : group 0  L(g, K-slices 0-1) + weight pieces 0,1,2 of step g+1     group 1  M(g-1, slices 2-3)
//     interval 4g+1 : group 0  M(g, 0-1), bare                                          group 1  L(g, 0-1) + its pieces
//     interval 4g+2 : group 0  L(g, 2-3) + weight piece 3 of step g+1 + the patch       group 1  M(g, 0-1)
//                               piece of this tap (next chunk's patch)
//     interval 4g+3 : group 0  M(g, 2-3), then vmcnt([patch] [+bias]):                  group 1  L(g, 2-3) + its pieces
//                               step g+1's weights have landed
// An LDS-DMA piece costs the issuing wave ~100 cycles of issue, during which it issues nothing else: inside a multiply segment
// that is 100 cycles of idle matrix pipe per piece (measured: 3 pieces between the MFMA groups of one segment = +20 % time),
// inside a load segment it is free as long as the segment stays under the partner's 512 cycles.  So the load segments are kept
// to the 12 reads + the pieces (3 and 2) and almost no address arithmetic, and the multiply segments are bare.
// Ordering rules (cdna_hip_programming.md "Read a staged buffer one phase AFTER the wait that retires it"): a piece is
// read only behind the issuing wave's counted vmcnt AND a later barrier; a buffer is re-filled only behind a barrier
// that every reader passed after an lgkmcnt(0).
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "kernels.h"

#ifndef TDRN_PP_ABLATE
#define TDRN_PP_ABLATE 0      // diagnostics: 1 = no LDS-DMA, 2 = no ds_read/MFMA, 4 = no epilogue stores, 8 = no patch pieces, 16 = no weight pieces
#endif

namespace tdrn {

struct PPParams {
    const char *in, *w, *zero;
    const float *bias;
    char *out, *out_pool;          // NHWC [B][H][W][Cs] and optional pooled [B][H/2][W/2][Cs]
    int B, H, W, Cin, Cout, Cs, Ktot;
    int relu;
    int tiles_x, tiles_per_img;    // 2-D tiles
    int m_tiles, n_tiles, items;   // items = m_tiles * n_tiles
    int n_major;                   // item = nt * m_tiles + mt (an XCD's contiguous item range then needs ONE cout tile's weights) instead of mt * n_tiles + nt
    int M;                         // B*H*W
    // chained split ("stream-K", see the kernel): fp32 accumulator slabs [workgroup][8 waves][32][64 lanes][4] and one flag
    // word per workgroup; null = whole items only
    float *sk_slab;
    unsigned *sk_flag;
    unsigned *status;              // host-visible status word (or null): <- 1 when a poll of the chained split runs out
    unsigned poll_max;             // bound of that poll (iterations of load + s_sleep 8)
    int fault;                     // fault injection: the producer never raises its flag
};

namespace {

template <typename DT> struct MmaPP;
template <> struct MmaPP<bf16_t> {
    __device__ static __forceinline__ void run(const u32x4 &a, const u32x4 &b, f32x16 &c)
    { c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(i16x8, a), __builtin_bit_cast(i16x8, b), c, 0, 0, 0); }
};
template <> struct MmaPP<f16_t> {
    __device__ static __forceinline__ void run(const u32x4 &a, const u32x4 &b, f32x16 &c)
    { c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0); }
};

// One LDS-DMA piece (64 lanes x 16 B -> 1 KiB of LDS at lds_dst + 16*lane) from a wave-uniform base plus a 32-bit per-lane
// byte offset.  Inline asm on purpose: (1) no 64-bit per-lane address arithmetic (the accumulators need the registers),
// (2) hipcc's waitcnt pass does not see it, so it cannot put an `s_waitcnt vmcnt(0)` in front of the ds_reads that follow
// (cdna_hip_programming.md 5.7 item 1: the completion is counted by hand -- the vmcnt(N) of every step below).  M0 is
// written in the statement that uses it and restored.
__device__ __forceinline__ void glds16(const char *sbase, unsigned voff, unsigned lds_dst)
{
    if constexpr (!(TDRN_PP_ABLATE & 1)) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(voff), "s"(sbase), "s"(lds_dst)
                     : "memory");
    }
}
__device__ __forceinline__ unsigned lds_addr(const void *p)
{
    return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const char *)p;
}

// (all waves of a workgroup take part in every barrier; nothing may move across it)
#define PP_BAR()                                  \
    do {                                          \
        __builtin_amdgcn_sched_barrier(0);        \
        __builtin_amdgcn_s_barrier();             \
        __builtin_amdgcn_sched_barrier(0);        \
    } while (0)
#define PP_LGKM0() __builtin_amdgcn_s_waitcnt(0xC07F)
#define PP_VM0() __builtin_amdgcn_s_waitcnt(0x0F70)   // vmcnt(0), through the builtin: the compiler's waitcnt pass sees it

constexpr int kPPSlots = 44;                      // 8-row LDS-DMA pieces per patch buffer (352 rows)
constexpr int kPPPieces = (kPPSlots + 7) / 8;     // patch pieces per wave and chunk (6: one per tap 0..5)

}  // namespace

// TW = 32 / 16: 2-D tiles of (256/TW) x TW pixels of one image; TW = 0: flat tiles of 256 consecutive NHW pixels.
// POOL: the instantiation that also (or only) writes the fused MaxPool2d(2,2) output (two layers of a VGG trunk); kept apart so
// that the other layers' register allocation does not carry the pooled epilogue.
template <typename DT, int TW, bool POOL>
__global__ __launch_bounds__(512, 2) void conv3x3_pp_kernel(const PPParams p)
{
    static_assert(sizeof(DT) == 2, "16-bit element types only");
    constexpr bool FLAT = TW == 0;
    constexpr int LGTW = TW == 32 ? 5 : 4;
    constexpr int TH = TW ? 256 / TW : 0;
    constexpr int ES = 2;
    constexpr int BN = 256, BNH = 128, WC = 4;
    constexpr int PBYTES = kPPSlots * 1024;
    constexpr int WBYTES = BN * 128;                    // one weight slot: 32 KiB (two group halves of 16 KiB)
    constexpr int OFF_W = 2 * PBYTES;
    constexpr int OFF_B = OFF_W + 2 * WBYTES;
    constexpr int OFF_Z = OFF_B + 1024;                 // 128 zero bytes: where a flat tile's out-of-image taps read
    constexpr int LDS = OFF_Z + 128;
    constexpr int SROWS = 16;                           // pixels per epilogue round (per wave)
    constexpr int SSTRIDE = BNH * ES + 16;              // staging row stride (bytes)
    constexpr int STRIP = SROWS * SSTRIDE;              // a wave's staging strip inside the dead patch buffer
    static_assert(8 * STRIP <= PBYTES, "staging strips fit the dead patch buffer");
    static_assert(LDS <= 160 * 1024, "LDS budget");
    __shared__ __attribute__((aligned(16))) char smem[LDS];

    const int lane = threadIdx.x & 63;
    // Register discipline: the 128 accumulators + 48 fragment registers leave ~70 for everything else, and hipcc hoists every
    // loop-invariant lane expression of the (rarely executed) staging / epilogue code out of the step loop and then spills
    // them -- re-loaded behind an `s_waitcnt vmcnt(0)` that would drain the LDS-DMA pipeline.  So those code paths derive
    // their lane constants from an OPAQUE copy of the lane id (a few vector instructions where they are used).
    auto opaque_lane = [&]() -> int {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        return ln;
    };
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wave >> 2;                          // cout half / ping-pong group
    const int cw = wave & 3;                            // pixel group: pixels [64*cw, 64*cw + 64)
    const int nchunks = p.Cin / 64;

    // ---- work distribution.  Each XCD label (blockIdx % 8) owns a contiguous range of items (cout siblings and
    // neighbouring tiles share its L2).  A workgroup's work is a list of SEGMENTS (item, chunks [c0, c1)):
    //   * whole-item mode: items item0, item0 + istride, ... (as conv3x3_patch.hip);
    //   * chained split mode (sk_slab != null; "stream-K"): the XCD's (item, chunk) units are cut into EQUAL contiguous
    //     ranges, one per workgroup, so a range may begin and end inside an item.  The workgroup that owns the first part of a
    //     split item writes its fp32 accumulators to a slab; the owner of the rest STARTS from them -- the K order of every
    //     output element is exactly that of an unsplit item, so the result does not depend on where the cuts fall (i.e. on
    //     the batch size): bit-identical to whole-item mode.  800 items on 256 CUs take 3.125 item times instead of 4; and
    //     the workgroups reach their epilogues at different times instead of all bursting their stores at once.
    //     A workgroup runs its unfinished LAST item first (its successor needs that slab), whole items next, and the item
    //     whose first part belongs to its predecessor last (the slab has long been written by then).
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int per_xcd = (p.items + 7) >> 3, istride = ((int)gridDim.x + 7) >> 3;
    int avail = p.items - xcd * per_xcd;
    avail = avail < per_xcd ? avail : per_xcd;
    avail = avail < 0 ? 0 : avail;
    const bool sk = p.sk_slab != nullptr;
    int n_seg, seg_item0, c0_first = 0, c1_last = nchunks, tail_first = 0, head_last = 0;
    if (sk) {
        const int units = avail * nchunks;
        // (units * 32 < 2^31 for any tensor below the 4-GiB limit of patch_conv_supported; 32-bit scalar arithmetic)
        const int u0 = __builtin_amdgcn_readfirstlane(units * slot / istride), u1 = __builtin_amdgcn_readfirstlane(units * (slot + 1) / istride);
        if (u1 <= u0) return;                           // (whole workgroup)
        const int fi = u0 / nchunks, li = (u1 - 1) / nchunks;
        n_seg = li - fi + 1;
        seg_item0 = xcd * per_xcd + fi;
        c0_first = u0 - fi * nchunks;
        c1_last = u1 - li * nchunks;
        tail_first = (c1_last != nchunks && n_seg > 1) ? 1 : 0;
        head_last = (c0_first != 0 && n_seg > 1) ? 1 : 0;
    } else {
        n_seg = avail > slot ? (avail - slot + istride - 1) / istride : 0;
        if (n_seg == 0) return;                         // (whole workgroup)
        seg_item0 = xcd * per_xcd + slot;
    }
    // segment k of the execution order -> item and chunk range (wave-uniform scalars)
    auto seg = [&](int k, int &item, int &c0, int &c1) {
        if (!sk) {
            item = seg_item0 + k * istride; c0 = 0; c1 = nchunks;
            return;
        }
        int j = k - tail_first + head_last;
        if (tail_first && k == 0) j = n_seg - 1;
        else if (head_last && k == n_seg - 1) j = 0;
        item = __builtin_amdgcn_readfirstlane(seg_item0 + j);
        c0 = __builtin_amdgcn_readfirstlane(j == 0 ? c0_first : 0);
        c1 = __builtin_amdgcn_readfirstlane(j == n_seg - 1 ? c1_last : nchunks);
    };
    const int my_wg = xcd * istride + slot;             // slab / flag index (the producer's); the consumer reads my_wg - 1
    const int RS = TW ? TW + 2 : p.W;                   // patch row stride of one image row

    // =========================== staging (LDS-DMA) state ===========================
    // Nothing per piece is kept in registers (the accumulators and fragments need them): a patch piece's source
    // address is recomputed in the load segment that issues it (~25 vector instructions; those segments have slack).
    int pt_b = 0, pt_y0 = 0, pt_x0 = 0;                 // 2-D: image and origin (incl. halo) of the tile being prefetched
    int pt_j0 = 0;                                      // flat: NHW pixel of patch row 0 (M = B*H*W < 2^31)
    auto mt_of = [&](int item) -> int { return p.n_major ? item % p.m_tiles : item / p.n_tiles; };
    auto nt_of = [&](int item) -> int { return p.n_major ? item / p.m_tiles : item % p.n_tiles; };
    auto patch_tile = [&](int item) {                   // (wave-uniform scalars; once per prefetched chunk)
        const int mt = mt_of(item);
        if (TW) {
            const int b = mt / p.tiles_per_img, tt = mt - b * p.tiles_per_img;
            const int ty = tt / p.tiles_x, tx = tt - ty * p.tiles_x;
            pt_b = __builtin_amdgcn_readfirstlane(b); pt_y0 = __builtin_amdgcn_readfirstlane(ty * TH - 1); pt_x0 = __builtin_amdgcn_readfirstlane(tx * TW - 1);
        } else {
            pt_j0 = __builtin_amdgcn_readfirstlane(mt * 256 - p.W - 1);
        }
    };
    const unsigned smem_lds = __builtin_amdgcn_readfirstlane(lds_addr(smem));
    // A patch piece is PREPARED in a load segment (source offset, which lanes are inside the image; lanes outside get their
    // zeros by an ordinary LDS store right there) and ISSUED in the following multiply segment, between two MFMA groups.
    unsigned pp_voff = 0, pp_dst = 0;                   // prepared piece: per-lane source offset, LDS destination
    bool pp_ok = false, pp_any = false;                 // per-lane: inside the image; wave-uniform: any lane is
    // my row of piece j is patch row pr = (wave + 8j)*8 + lane/8: (py, px) = (pr / RS, pr % RS) advance by 64 rows per piece
    // (kept incrementally: two registers instead of a division per step); the swizzle term of the row does not depend on j
    int pp_yx = 0;                                      // (py << 8) | px, one register
    auto prep_patch = [&](int j, unsigned ccoff, int dstbuf_off) {   // dstbuf_off: byte offset of the patch buffer in smem
        const int q = wave + 8 * j;                     // piece = patch rows [8q, 8q + 8)
        pp_any = false;
        if (q >= kPPSlots) return;                      // (wave-uniform)
        const int ln = opaque_lane();
        const int lrow = ln >> 3, pc = ln & 7;
        const unsigned lc = (unsigned)((pc ^ ((4 * wave + (lrow >> 1)) & 7)) << 4) + ccoff;
        const unsigned rowbytes = (unsigned)(p.Cin * ES);
        unsigned off;                                   // (the tensor is < 4 GiB: patch_conv_supported)
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
        // rows outside the image (the conv's zero padding) and beyond the patch are ZEROED by an ordinary LDS store of the
        // lanes concerned; the LDS-DMA runs with those lanes switched off (an inactive lane writes nothing)
        const int piece = dstbuf_off + q * 1024;
        if (!ok) {
            unsigned z;
            asm volatile("v_mov_b32 %0, 0" : "=v"(z));  // (materialised HERE: a hoisted zero is one more register to spill)
            *(u32x4 *)(smem + piece + ln * 16) = u32x4{z, z, z, z};
        }
        pp_voff = off + lc;
        pp_ok = ok;
        pp_any = __builtin_amdgcn_ballot_w64(ok) != 0ull;   // a piece wholly outside the image issues NO DMA
        pp_dst = __builtin_amdgcn_readfirstlane(smem_lds + piece);
    };
    auto issue_patch = [&]() {
        if (pp_any) {
            if (pp_ok && !(TDRN_PP_ABLATE & 8)) glds16(p.in, pp_voff, __builtin_amdgcn_readfirstlane(pp_dst));
        }
    };
    // my group's half of a weight slot: rows grp*128 + (cw + 4k)*8 + lrow, k = 0..3 (the swizzle term is the same for all k)
    const unsigned wo = (unsigned)((grp * 128 + cw * 8 + (lane >> 3)) * p.Ktot * ES) + (unsigned)((((lane & 7) ^ ((4 * cw + (lane >> 4)) & 7)) << 4));
    const unsigned wstep = (unsigned)(32u * p.Ktot * ES);
    // byte offset into p.w of the weight slice of (item, chunk c, tap 0); a tap adds Cin*ES  (32-bit scalars: a 64-bit pointer
    // that hipcc cannot prove wave-uniform lands in a VGPR pair -- and is spilled)
    auto weight_off = [&](int item, int c) -> unsigned {
        return (unsigned)__builtin_amdgcn_readfirstlane(nt_of(item) * BN * p.Ktot * ES + c * 128);
    };
    // piece k (0..3) of the slice at byte offset `off` -> my group's half of weight slot `slot`
    auto weight_piece = [&](int k, unsigned off, int slot) {
        unsigned w = wo;
        asm volatile("" : "+v"(w));                     // (or hipcc keeps wo + k*wstep, k = 1..3, live across the loop -- and spills them)
        if (!(TDRN_PP_ABLATE & 16))
            glds16(p.w + off, w + k * wstep, __builtin_amdgcn_readfirstlane(smem_lds + OFF_W + slot * WBYTES + grp * (WBYTES / 2) + (cw + 4 * k) * 1024));
    };
    auto load_bias = [&](int item) {                    // wave 0: the item's 256 biases = one 1-KiB piece
        glds16((const char *)(p.bias + nt_of(item) * BN), (unsigned)opaque_lane() * 16u, smem_lds + OFF_B);
    };

    // =========================== compute state ===========================
    f32x16 acc[WC][2];
    unsigned tapmask[2] = {0x1FFu, 0x1FFu};             // flat mode: bit t = tap t inside the image
    constexpr bool compute = !(TDRN_PP_ABLATE & 2);

    int cur_mt = -1, n0 = 0;
    long long tile_pix0 = 0;                            // 2-D: global pixel of the tile's (0,0); flat: mt*256
    int tile_row0 = 0, tile_x0 = 0;                     // 2-D: b*H + y and x of the tile's (0,0)
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
        const int r32 = opaque_lane() & 31;
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
        }
    };

    // ---- epilogue of one item: straight from the registers (round 5; rounds 3-4 staged every tile through a strip in the dead patch buffer) ----
    auto epilogue = [&](char *) {
        // lane-derived constants are re-derived here from an opaque copy of the lane id: hoisted out of the step loop they
        // would be live (or spilled and re-loaded behind a vmcnt(0)) across every multiply segment
        const int ln = opaque_lane();
        const int r32 = ln & 31, hh = ln >> 5;
        auto pixel_of = [&](int i) -> long long {       // global pixel of tile-local pixel i (or -1)
            if (TW) return tile_pix0 + (long long)(i >> LGTW) * p.W + (i & (TW - 1));
            const long long m = tile_pix0 + i;
            return m < p.M ? m : -1;
        };
        if (p.out && !(TDRN_PP_ABLATE & 4)) {
            // Round 5: straight from the registers.  Group c = 4 ci + g is the 16-byte chunk c of my pixel's span of 128 couts; lane r32 has its
            // first 8 bytes, lane r32 + 32 the second: v_permlane32_swap on a chunk PAIR gives each of the two lanes one whole chunk
            // (cdna_hip_programming.md T21) -> eight dwordx4 stores per pixel fragment, no staging strip, no LDS round trips, no wave barriers.
            // ReLU as a signed 16-bit maximum with 0 on the packed pair (exact for finite values); same v_cvt_pk as before: same bits.
            typedef short pk_s2 __attribute__((ext_vector_type(2)));
            const pk_s2 relu_lo = p.relu ? pk_s2{0, 0} : pk_s2{(short)-32768, (short)-32768};
            auto pk = [&](float a, float b) -> unsigned {
                return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(pk_s2, pack2<DT>(a, b)), relu_lo));
            };
            const int cbase = n0 + grp * BNH;
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) {
                const long long gp = pixel_of(cw * 64 + pt * 32 + r32);
                char *row = p.out + ((size_t)(gp < 0 ? 0 : gp) * p.Cs + cbase) * ES;
#pragma unroll
                for (int pr = 0; pr < 2 * WC; ++pr) {
                    const int ci = pr >> 1, g0 = 2 * (pr & 1);
                    const f32x16 &t = acc[ci][pt];
                    const unsigned ax = pk(t[4 * g0], t[4 * g0 + 1]), ay = pk(t[4 * g0 + 2], t[4 * g0 + 3]);
                    const unsigned bx = pk(t[4 * g0 + 4], t[4 * g0 + 5]), by = pk(t[4 * g0 + 6], t[4 * g0 + 7]);
                    auto rx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
                    auto ry = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
                    if (gp >= 0 && cbase + (2 * pr + hh) * 8 < p.Cout)
                        *(u32x4 *)(row + (2 * pr + hh) * 16) = u32x4{rx[0], ry[0], rx[1], ry[1]};
                }
            }
        }
        if constexpr (POOL && TW != 0) if (p.out_pool) {
            // fused MaxPool2d(2,2) on the RAW accumulators (max commutes with the monotonic bias + ReLU + rounding applied afterwards), straight
            // from the registers as in conv3x3_patch.hip (round 5): the partner row is my other pixel fragment (TW = 32) or lane ^ 16
            // (v_permlane16_swap), the partner column lane ^ 1 (DPP); the window's 2 / 4 lanes then hold the same maxima and store different
            // chunk pairs of the same pooled pixel (v_permlane32_swap pairs -> dwordx4).
            typedef short pk_s2 __attribute__((ext_vector_type(2)));
            const pk_s2 relu_lo = p.relu ? pk_s2{0, 0} : pk_s2{(short)-32768, (short)-32768};
            auto pkr = [&](float a, float b) -> unsigned {
                return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(pk_s2, pack2<DT>(a, b)), relu_lo));
            };
            const int PW = p.W >> 1;
            const int cbase = n0 + grp * BNH;
            constexpr int NPAIR = 2 * WC;                       // 8 chunk pairs per pixel
            constexpr int NDUP = TW == 32 ? 2 : 4;              // lanes holding the same window
            const int sel = TW == 32 ? (r32 & 1) : ((r32 & 1) | ((r32 >> 3) & 2));
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) {
                if (TW == 32 && pt == 1) break;
                const int i = cw * 64 + pt * 32 + r32;
                const long long gpool = (long long)((tile_row0 + (i >> LGTW)) >> 1) * PW + ((tile_x0 + (i & (TW - 1))) >> 1);
                char *row = p.out_pool + ((size_t)gpool * p.Cs + cbase) * ES;
                // round rd: chunk pairs rd * NDUP .. rd * NDUP + NDUP - 1, one per lane of the window (only their maxima are computed now:
                // 16 packed registers live at a time instead of 64)
#pragma unroll
                for (int rd = 0; rd < NPAIR / NDUP; ++rd) {
                    uint2 a = make_uint2(0u, 0u), b = make_uint2(0u, 0u);
#pragma unroll
                    for (int d = 0; d < NDUP; ++d) {
                        const int pr = rd * NDUP + d, ci = pr >> 1, g0 = 2 * (pr & 1);
                        float m[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            float v = acc[ci][pt][4 * g0 + j];
                            if (TW == 32) {
                                v = fmaxf(v, acc[ci][1][4 * g0 + j]);
                            } else {
                                float va = v, vb = v;
                                asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(va), "+v"(vb));
                                v = fmaxf(va, vb);
                            }
                            m[j] = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)));
                        }
                        const bool mine = sel == d;
                        const unsigned ax = pkr(m[0], m[1]), ay = pkr(m[2], m[3]), bx = pkr(m[4], m[5]), by = pkr(m[6], m[7]);
                        a.x = mine ? ax : a.x; a.y = mine ? ay : a.y; b.x = mine ? bx : b.x; b.y = mine ? by : b.y;
                    }
                    const int mypr = rd * NDUP + sel;
                    auto rx = __builtin_amdgcn_permlane32_swap(a.x, b.x, false, false);
                    auto ry = __builtin_amdgcn_permlane32_swap(a.y, b.y, false, false);
                    if (cbase + (2 * mypr + hh) * 8 < p.Cout)
                        *(u32x4 *)(row + (2 * mypr + hh) * 16) = u32x4{rx[0], ry[0], rx[1], ry[1]};
                }
            }
        }
    };

    // accumulators start at the bias (one 1-KiB LDS slot, re-staged by wave 0 late in the previous item)
    auto init_acc = [&]() {
        const int hh = opaque_lane() >> 5;
        const char *bsrc = smem + OFF_B;
#pragma unroll
        for (int ci = 0; ci < WC; ++ci)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 bv = *(const f32x4 *)(bsrc + (grp * BNH + ci * 32 + 8 * g + 4 * hh) * 4);
#pragma unroll
                for (int pt = 0; pt < 2; ++pt)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[ci][pt][4 * g + j] = bv[j];
            }
    };

    // ---- chained split: a segment that starts inside an item continues from its predecessor's partial sums; one that ends
    // inside an item hands its own on.  Slab image = the accumulator registers as they stand: [wave][quad 0..31][lane] x 16 B,
    // every access a contiguous 1 KiB per wave.  Hand-off = cdna_hip_programming.md Guideline 16: plain stores, every storing
    // wave's vmcnt(0), workgroup barrier, ONE agent-scope release + vmcnt(0), relaxed agent flag store; the consumer polls
    // relaxed (bounded), ONE agent-scope acquire + vmcnt(0), workgroup barrier, plain loads.  A flag has one reader, which
    // resets it: a launch that finds the flag words zero leaves them zero (ConvArgs::sk_flags_zero: net.hip zeroes them once
    // per forward on a side lane; otherwise a memset node in front of the launch).  Each of the two routines contains ONE workgroup barrier.
    auto slab_of = [&](int wg) -> f32x4 * { return (f32x4 *)(p.sk_slab + ((size_t)wg * 8 + wave) * (32 * 64 * 4)) + opaque_lane(); };
    auto begin_acc = [&](int c0) {
        if (c0 == 0) {
            init_acc();
            return;
        }
        if (wave == 0) {
            if (lane == 0) {
                // Bounded poll.  Progress argument: the producer (workgroup my_wg - 1) writes this slab FIRST (its unfinished last
                // item runs first) and I read it LAST (behind all my other items), and all 256 workgroups of the launch are
                // resident at once (grid == 256 == one per CU; the dispatcher starts lower block indices first when a concurrent
                // side-lane kernel holds CUs, so a producer is never started after its consumer).  Should the poll run out all
                // the same, the forward is REPORTED failed through the host-visible status word (tdrn_net_check; the item is then
                // finished from whatever the slab holds, so the launch still ends) -- and the flag is NOT reset: a late producer
                // would leave it raised for good; the next forward's memset clears it.
                unsigned spins = 0;
                bool seen;
                while (!(seen = __hip_atomic_load(p.sk_flag + (my_wg - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) && ++spins < p.poll_max)
                    __builtin_amdgcn_s_sleep(8);
                if (seen) __hip_atomic_store(p.sk_flag + (my_wg - 1), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // (its only reader)
                else if (p.status) __hip_atomic_store(p.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        PP_BAR();
        const f32x4 *src = slab_of(my_wg - 1);
#pragma unroll
        for (int ci = 0; ci < WC; ++ci)
#pragma unroll
            for (int pt = 0; pt < 2; ++pt)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = src[((ci * 2 + pt) * 4 + q) * 64];
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[ci][pt][4 * q + j] = v[j];
                }
        PP_VM0();                                       // (through the builtin: hipcc's scoreboard is empty when the step loop resumes)
    };
    auto write_partial = [&]() {
        f32x4 *dst = slab_of(my_wg);
#pragma unroll
        for (int ci = 0; ci < WC; ++ci)
#pragma unroll
            for (int pt = 0; pt < 2; ++pt)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    dst[((ci * 2 + pt) * 4 + q) * 64] = f32x4{acc[ci][pt][4 * q], acc[ci][pt][4 * q + 1], acc[ci][pt][4 * q + 2], acc[ci][pt][4 * q + 3]};
        PP_VM0();                                       // my stores have left (through the builtin: hipcc sees this wait too)
        PP_BAR();
        if (wave == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0 && !p.fault) __hip_atomic_store(p.sk_flag + my_wg, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };

    // ---- operand fragments of one half step (two K slices): 8 weight + 4 patch ds_read_b128 ----------------------
    // Address of K slice kk (16-B chunk 2kk + hh of a 128-B row, XOR-swizzled by sw = (row >> 1) & 7):
    //     row*128 + (((2kk + hh) ^ sw) << 4)  =  [row*128 + ((hh ^ sw) << 4)]  ^  (kk << 5)
    // so a step needs ONE base per operand row (wa: my weight row, pa[pt]: my two patch rows of this tap) and an XOR per
    // slice.  Flat tiles: a lane whose tap falls outside the image reads the 128-B ZERO ROW instead (one select per
    // pixel tile and step, not an AND per fragment register).
    u32x4 wf[WC][2], pf[2][2];
    unsigned wa = 0, pa[2] = {0, 0};                    // LDS byte addresses (relative to smem)
    auto step_addresses = [&](int wslot, int pbuf_, int rowdelta, unsigned tapbits) {
        const int ln = opaque_lane();
        const int r32 = ln & 31, hh = ln >> 5;
        wa = (unsigned)(OFF_W + wslot * WBYTES + grp * (WBYTES / 2) + r32 * 128 + ((hh ^ ((r32 >> 1) & 7)) << 4));
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
            // (the tap-(0,0) patch row of my pixel is re-derived here: kept in a register it would be live across the epilogue,
            // where hipcc spills it -- and re-loads it in every step behind a vmcnt(0))
            const int i = cw * 64 + pt * 32 + r32;      // tile-local pixel
            const int row = (TW ? (i >> LGTW) * RS + (i & (TW - 1)) : i) + rowdelta;
            const unsigned a = (unsigned)(pbuf_ * PBYTES + row * 128 + ((hh ^ ((row >> 1) & 7)) << 4));
            if constexpr (FLAT) pa[pt] = ((tapmask[pt] >> tapbits) & 1u) ? a : (unsigned)OFF_Z;
            else pa[pt] = a;
        }
    };
    auto read_frags = [&](int half) {
        if (!compute) return;
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            const unsigned kx = (unsigned)((2 * half + k2) << 5);
#pragma unroll
            for (int ci = 0; ci < WC; ++ci) wf[ci][k2] = *(const u32x4 *)(smem + ((wa ^ kx) + ci * 4096));
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) pf[pt][k2] = *(const u32x4 *)(smem + (pa[pt] ^ kx));
        }
    };
    // four MFMAs: K slice k2, cout tiles 2*h and 2*h+1, both pixel tiles
    auto mma_quad = [&](int k2, int h) {
        if (!compute) return;
#pragma unroll
        for (int ci = 2 * h; ci < 2 * h + 2; ++ci)
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) MmaPP<DT>::run(wf[ci][k2], pf[pt][k2], acc[ci][pt]);
    };
#define PP_SB() __builtin_amdgcn_sched_barrier(0)

    // =========================== prologue ===========================
    // The loop runs chunk by chunk with the nine taps of a chunk UNROLLED: which tap issues which patch piece, where the bias
    // goes out, when a chunk / a segment ends are compile-time facts of a step; the cursor logic runs once per chunk.  (As a
    // generic per-step state machine it was ~490 instructions, ~30 branches and ~30 SGPR-spill reads per step -- twice the
    // issue time of the 32 MFMAs they surround: matrix pipe 44 % busy.)
    int kseg = 0, cur_item, cc, c1cur;                  // the chunk being computed: segment, item, chunk, end chunk of the segment
    seg(0, cur_item, cc, c1cur);
    int n_kseg = 0, n_item = 0, n_cc = 0, n_c1 = 0;     // the chunk after it (n_kseg == n_seg: none)
    auto next_chunk = [&]() {
        if (cc + 1 < c1cur) {
            n_kseg = kseg; n_item = cur_item; n_cc = cc + 1; n_c1 = c1cur;
        } else {
            n_kseg = kseg + 1;
            if (n_kseg < n_seg) seg(n_kseg, n_item, n_cc, n_c1);
        }
    };
    next_chunk();
    unsigned w_cur = weight_off(cur_item, cc), w_nxt = n_kseg < n_seg ? weight_off(n_item, n_cc) : 0u;
    int pbuf = 0, cpar = 0;                             // patch buffer of the chunk being computed; parity of its first weight slot
    patch_tile(cur_item);
#pragma unroll
    for (int j = 0; j < kPPPieces; ++j) {
        prep_patch(j, (unsigned)(cc * 128), 0);
        issue_patch();
    }
    pp_any = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) weight_piece(k, w_cur, 0);     // step 0 -> slot 0
    if (wave == 0 && cc == 0) load_bias(cur_item);
    if (wave == 1 && lane < 8) *(u32x4 *)(smem + OFF_Z + lane * 16) = u32x4{0u, 0u, 0u, 0u};
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    PP_VM0();                                           // (see below: hipcc's own scoreboard must be empty when the step loop starts)
    PP_BAR();
    setup_item(cur_item);
    begin_acc(cc);
    step_addresses(0, 0, 0, 0u);
    PP_LGKM0();
    PP_VM0();
    if (grp == 1) PP_BAR();                             // the stagger: group 1 runs one interval behind group 0

    const unsigned tapbytes = (unsigned)(p.Cin * ES);
    // one step: tap T of the current chunk.  `more`: a step follows (its weights are staged here).
    auto step = [&](auto tapc) {
        constexpr int T = decltype(tapc)::value;
        const bool nvalid = n_kseg < n_seg;
        const bool more = T < 8 || nvalid;
        const unsigned w_next = T < 8 ? w_cur + (unsigned)(T + 1) * tapbytes : w_nxt;   // the weight slice of the NEXT step
        const int slot_next = (cpar + T + 1) & 1;
        // ---------------- L(slices 0-1): reads; the four weight pieces of the next step (into the slot the previous step left) ----
        read_frags(0);
        PP_SB();
        if (more) {
            weight_piece(0, w_next, slot_next);
            weight_piece(1, w_next, slot_next);
        }
        PP_LGKM0();
        PP_BAR();
        // ---------------- M(slices 0-1): the 16 MFMAs + the other two weight pieces between the groups ----------------
        __builtin_amdgcn_s_setprio(1);
        mma_quad(0, 0);
        PP_SB();
        if (more) weight_piece(2, w_next, slot_next);
        PP_SB();
        mma_quad(0, 1);
        mma_quad(1, 0);
        PP_SB();
        if (more) weight_piece(3, w_next, slot_next);
        PP_SB();
        mma_quad(1, 1);
        __builtin_amdgcn_s_setprio(0);
        PP_BAR();
        // ---------------- L(slices 2-3): reads; this tap's piece of the NEXT chunk's patch (prepared here); the bias ----------
        read_frags(1);
        PP_SB();
        pp_any = false;
        if constexpr (T < kPPPieces) {
            if (nvalid) {
                if constexpr (T == 0) patch_tile(n_item);
                prep_patch(T, (unsigned)(n_cc * 128), (pbuf ^ 1) * PBYTES);
            }
        }
        const int fly_patch = pp_any ? 1 : 0;
        if constexpr (T < kPPPieces) issue_patch();
        // the NEXT segment's bias (if it starts an item) -> the LDS bias slot; this segment read the slot at its start
        bool bias_now = false;
        if constexpr (T == 6) {
            bias_now = wave == 0 && nvalid && n_kseg != kseg && n_cc == 0;
            if (bias_now) load_bias(n_item);
        }
        PP_LGKM0();
        PP_BAR();
        // ---------------- M(slices 2-3) + (between the MFMA groups) the read addresses of the next step ----------------
        __builtin_amdgcn_s_setprio(1);
        mma_quad(0, 0);
        mma_quad(0, 1);
        PP_SB();
        if constexpr (T < 8) step_addresses(slot_next, pbuf, ((T + 1) / 3) * RS + (T + 1) % 3, (unsigned)(T + 1));
        else if (n_kseg == kseg) step_addresses(slot_next, pbuf ^ 1, 0, 0u);     // (a new segment sets them behind its epilogue)
        PP_SB();
        mma_quad(1, 0);
        mma_quad(1, 1);
        __builtin_amdgcn_s_setprio(0);
        // In-order retirement: `vmcnt(N)` waits for all but the N youngest pieces.  Needed now: the next step's four weight
        // pieces; issued behind them and free to stay in flight for another step: the patch piece (served from beyond L2) and
        // the bias.
        {
            const int fly = (TDRN_PP_ABLATE & (1 | 8 | 16)) ? 0 : fly_patch + (bias_now ? 1 : 0);
            if (fly == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (fly == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        }
        PP_BAR();
    };
#pragma unroll 1
    for (;;) {
        step(std::integral_constant<int, 0>{});
        step(std::integral_constant<int, 1>{});
        step(std::integral_constant<int, 2>{});
        step(std::integral_constant<int, 3>{});
        step(std::integral_constant<int, 4>{});
        step(std::integral_constant<int, 5>{});
        step(std::integral_constant<int, 6>{});
        step(std::integral_constant<int, 7>{});
        step(std::integral_constant<int, 8>{});
        // ---------------- the chunk is done ----------------
        const bool seg_done = n_kseg != kseg;
        const int dead_buf = pbuf;
        const int c1_done = c1cur;
        pbuf ^= 1;
        cpar ^= 1;
        kseg = n_kseg; cur_item = n_item; cc = n_cc; c1cur = n_c1;
        if (kseg < n_seg) {
            w_cur = w_nxt;
            next_chunk();
            w_nxt = n_kseg < n_seg ? weight_off(n_item, n_cc) : 0u;
        }
        if (seg_done) {
            // group 0 waits for group 1's last multiply (one interval), then both groups run their epilogues at the
            // same time out of the patch buffer that died with this step; group 1 re-creates the stagger behind it
            if (grp == 0) PP_BAR();
            if (c1_done == nchunks) epilogue(smem + dead_buf * PBYTES + wave * STRIP);
            else write_partial();                       // (chained split: the item is finished by my successor)
            if (kseg < n_seg) {
                setup_item(cur_item);
                begin_acc(cc);
                step_addresses(cpar, pbuf, 0, 0u);     // (a flat tile's tap masks are the new item's)
            }
            PP_LGKM0();
            // (No vmcnt wait here: the epilogue's stores drain under the next segment's first step -- waiting for them cost
            // ~5 us per item, 14 % of a 36-step item.  hipcc's own scoreboard holds only STORES at this point, which it never
            // waits for; the one path with loads, begin_acc's slab read, ends in a builtin vmcnt(0) of its own.  Why that
            // matters: hipcc does not see the LDS-DMA pieces (inline asm); with a LOAD of its own -- a spill re-load, the slab
            // read -- still on its scoreboard at the loop's back edge it protects the register with an `s_waitcnt vmcnt(0)`
            // inside the step loop, which in hardware drains every piece in flight, every step: +40 % time when it happened.)
            if (grp == 1) PP_BAR();
            if (kseg >= n_seg) break;
        }
    }
    if (grp == 0) PP_BAR();                             // (matches group 1's re-stagger barrier of the last item)

}

// ---------------------------------------------------------------------------------------------
static int g_pp_override = -1;                          // dev harness / tests: -1 = environment, 0 / 1 = forced
void conv_pp_force(int v) { g_pp_override = v; }
int conv_pp_enabled()
{
    if (g_pp_override >= 0) return g_pp_override;
    static int e = -1;
    if (e < 0) { const char *s = getenv("TDRN_CONV_PP"); e = s ? atoi(s) : 1; }
    return e;
}

static int g_pp_sk_override = -1;
void conv_pp_sk_force(int v) { g_pp_sk_override = v; }
int conv_pp_sk_enabled()
{
    if (g_pp_sk_override >= 0) return g_pp_sk_override;
    static int e = -1;
    if (e < 0) { const char *s = getenv("TDRN_CONV_PP_SK"); e = s ? atoi(s) : 1; }
    return e;
}
// scratch of the chained split: 256 flag words + one 256-KiB accumulator slab per workgroup
size_t conv_pp_sk_bytes() { return 1024 + (size_t)256 * 8 * 32 * 64 * 16; }

// the layers this kernel takes over from conv3x3_patch.hip: 16-bit, >= 2 channel chunks, couts in whole 256-groups
int pp_conv_supported(const ConvArgs &a)
{
    if (!conv_pp_enabled() || (a.kdisable & 1)) return 0;
    if (a.dtype == TDRN_F32 || a.fuse_x) return 0;
    // (Cin = 128 -- two chunks, 18 steps per item -- stays with conv3x3_patch.hip: the per-item cost of this kernel, drain +
    // epilogue + re-stagger, weighs 10 % there: 125 vs 116 us on conv3_1 in the net)
    if (a.Cin < 256 || a.Cin % 64 || a.Npad % 256) return 0;
    return patch_conv_supported(a);
}

int launch_conv3x3_pp(const ConvArgs &a, void *out_pool, hipStream_t s)
{
    const int mode = pp_conv_supported(a);
    if (!mode) return TDRN_E_UNSUPPORTED;
    // pooled layers (conv3_3: the pooled output only): the POOL instantiation with the register-only pooled epilogue (round 5; with the staged
    // epilogue its 256-register budget spilled inside the step loop).  Flat tiles have no pooled variant.
    if (out_pool && (mode < 0 || (a.H & 1) || (a.W & 1))) return TDRN_E_UNSUPPORTED;
    PPParams p;
    p.in = (const char *)a.in; p.w = (const char *)a.w; p.zero = (const char *)a.zero_page; p.bias = a.bias;
    p.out = (char *)a.out; p.out_pool = (char *)out_pool;
    p.B = a.B; p.H = a.H; p.W = a.W; p.Cin = a.Cin; p.Cout = a.Cout; p.Cs = (int)a.o_cs; p.Ktot = 9 * a.Cin;
    p.relu = a.relu;
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
    {
        // With two or more cout tiles an XCD whose items are (pixel tile, cout tile) pairs keeps the WHOLE weight matrix live (4.7 MB at
        // 512 x 512, more than its 4-MB L2, and under the chained split its workgroups are at different channel chunks at the same
        // time): cout-tile-major numbering gives an XCD one cout tile's weights (2.4 MB) at the price of reading every patch on two XCDs.
        static int nm = -1;
        if (nm < 0) { const char *e = getenv("TDRN_PP_NMAJOR"); nm = e ? atoi(e) : 1; }
        p.n_major = (nm && p.n_tiles > 1) ? 1 : 0;
    }
    if (p.items <= 0) return TDRN_OK;
    // below ~3/4 of a full grid the loader/consumer kernel's smaller (128- / 64-cout) items fill more CUs: measured 2x faster
    // at 50-100 items, equal at 200 (the two kernels produce the same bits, so the choice may depend on the batch)
    if (p.items < 192) return TDRN_E_UNSUPPORTED;
    int grid = p.items >= 256 ? 256 : ((p.items + 7) / 8) * 8;
    const int cap = a.max_wgs > 0 ? (a.max_wgs / 8) * 8 : 0;
    if (cap > 0 && grid > cap) grid = cap;
    // chained split when it shortens the launch: a full grid, more than one item per workgroup, and an item count that does
    // not divide evenly (otherwise whole items are already balanced); the choice changes no output bit
    p.sk_slab = nullptr; p.sk_flag = nullptr;
    p.status = a.status; p.fault = a.fault_handoff; p.poll_max = a.fault_handoff ? (1u << 10) : (1u << 20);
    // (grid == 256: every workgroup of the launch is resident at once, see the poll in begin_acc)
    if (a.sk_ws && conv_pp_sk_enabled() && !(a.kdisable & 2) && (grid == 256 || (cap > 0 && grid == cap && cap >= 192)) && p.items > grid && p.items % grid != 0) {
        p.sk_flag = (unsigned *)a.sk_ws;
        p.sk_slab = (float *)((char *)a.sk_ws + 1024);
        if (!a.sk_flags_zero) TDRN_HIP_TRY(hipMemsetAsync(p.sk_flag, 0, 1024, s));
    }
#define PP_LAUNCH(DT)                                                                                                  \
    do {                                                                                                               \
        if (out_pool) {                                                                                                \
            if (tw == 32) hipLaunchKernelGGL((conv3x3_pp_kernel<DT, 32, true>), dim3(grid), dim3(512), 0, s, p);     \
            else hipLaunchKernelGGL((conv3x3_pp_kernel<DT, 16, true>), dim3(grid), dim3(512), 0, s, p);               \
        } else if (tw == 0) hipLaunchKernelGGL((conv3x3_pp_kernel<DT, 0, false>), dim3(grid), dim3(512), 0, s, p);    \
        else if (tw == 32) hipLaunchKernelGGL((conv3x3_pp_kernel<DT, 32, false>), dim3(grid), dim3(512), 0, s, p);   \
        else hipLaunchKernelGGL((conv3x3_pp_kernel<DT, 16, false>), dim3(grid), dim3(512), 0, s, p);                  \
    } while (0)
    if (a.dtype == TDRN_BF16) PP_LAUNCH(bf16_t);
    else PP_LAUNCH(f16_t);
#undef PP_LAUNCH
    return hip_status(hipGetLastError());
}

}  // namespace tdrn
