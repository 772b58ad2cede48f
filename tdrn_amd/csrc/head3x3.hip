// head3x3.hip -- the narrow 3x3 heads (ARM loc: 12 fp32 columns per pixel straight into the (B, P, 4) array) as a kernel of
// their own (round 5).
//
// reference: model/dualrefinedet_vggbn.py:155-164 (arm_loc[k] = nn.Conv2d(C, 3 * 4, kernel_size=3, padding=1) on each of the four
// pyramid sources, output permuted to (B, H, W, 12) and concatenated), networks.py:300-330 for the MobileNet variant.
//
// On the generic implicit GEMM (conv_igemm.hip <128, 32>) these launches re-read the source map once per tap -- 51 200 pixels x
// 512 channels x 9 taps = 472 MB through L2 for 5.7 GFLOP at batch 32: 75 us alone, 159 us at MobileNet's batch 64; per step the
// four of them were 6 % (VGG) / 9 % (MobileNet) of the chip's time.  Here a workgroup owns 256 consecutive pixels of one image:
//   * per 64-channel chunk it stages the pixels [p0 - W - 1, p0 + 256 + W + 1) ONCE (LDS-DMA, 128-byte rows, 16-byte chunks
//     XOR-swizzled by the row) together with the chunk's [tap][16 columns] weight rows, double-buffered, one barrier per chunk;
//   * a wave multiplies 32 pixels: v_mfma_f32_16x16x32 with the WEIGHTS as the A operand (M = 16 output columns) and 16 pixels as
//     B, so a lane ends up with four consecutive columns of one pixel = one 16-byte store into the (B, P, 4) array;
//   * a tap is a shift of the LDS read address by dy * W + dx rows; taps that leave the row read a zeroed LDS row instead (rows above /
//     below the image were staged from the zero page).
// K order: (chunk, tap, channel) -- not conv_igemm's (tap, chunk, channel): the fp32 sums differ in their last bits (the 16-bit
// stage tests bound it: tests/test_gpu_pin16.py); TDRN_PLAN_NO_HEAD3X3 keeps the launches on conv_igemm.hip.
#include <hip/hip_runtime.h>

#include "common.h"
#include "kernels.h"

namespace tdrn {

struct Head3Params {
    const char *in, *w;            // NHWC [B][H][W][Cin] DT; [Npad][9][Cin] DT
    const float *bias;             // [Npad]
    float *out;                    // element (b, y, x, co) at out + b * o_bs + y * o_rs + x * o_cs + co
    long long o_bs, o_rs, o_cs;
    int B, H, W, Cin, Cout, relu, tiles_per_img;
    int vec16;                     // out is 16-byte aligned: a pixel's four columns go out as one dwordx4
};

constexpr int kH3MaxW = 64;
constexpr int kH3PatchRows = 256 + 2 * kH3MaxW + 2 + 6;          // (+6: whole 8-row pieces)
constexpr int kH3PatchBytes = ((kH3PatchRows + 7) / 8) * 1024;
constexpr int kH3WBytes = 9 * 16 * 128;

__device__ __forceinline__ void h3_glds16(const char *sbase, unsigned voff, unsigned lds_dst)
{
    // (inline asm: hipcc neither sees nor drains the LDS-DMA queue -- conv3x3_pp.hip)
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory");
}

template <typename DT> struct Mma16;
template <> struct Mma16<bf16_t> {
    __device__ static __forceinline__ f32x4 run(const u32x4 &a, const u32x4 &b, const f32x4 &c)
    { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(i16x8, a), __builtin_bit_cast(i16x8, b), c, 0, 0, 0); }
};
template <> struct Mma16<f16_t> {
    __device__ static __forceinline__ f32x4 run(const u32x4 &a, const u32x4 &b, const f32x4 &c)
    { return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0); }
};

template <typename DT>
__global__ __launch_bounds__(512, 2) void head3x3_kernel(const Head3Params p)
{
    // LDS: two patch buffers, two weight buffers, one zeroed row
    __shared__ __attribute__((aligned(16))) char smem[2 * kH3PatchBytes + 2 * kH3WBytes + 128];
    const unsigned smem_lds = (unsigned)(unsigned long long)(__attribute__((address_space(3))) const char *)smem;
    constexpr int OFF_W = 2 * kH3PatchBytes, OFF_Z = OFF_W + 2 * kH3WBytes;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = (int)blockIdx.x / p.tiles_per_img, tile = (int)blockIdx.x - b * p.tiles_per_img;
    const int HW = p.H * p.W, W = p.W;
    const int p0 = tile * 256;                            // first pixel of my tile (row-major inside image b)
    const int g0 = p0 - W - 1;                            // image pixel of patch row 0
    const int nrows = 256 + 2 * W + 2;                    // patch rows that are read
    const int npieces = (nrows + 7) >> 3;
    const int nchunks = p.Cin >> 6;
    if (threadIdx.x < 32) ((unsigned *)(smem + OFF_Z))[threadIdx.x] = 0u;
    // ---- staging of channel chunk cc into buffer `buf` (every wave: pieces wave, wave + 8, ...) ----
    const char *img = p.in + (size_t)b * HW * p.Cin * 2;
    auto stage = [&](int cc, int buf) {
        const int r8 = lane >> 3, cpos = lane & 7;
        for (int j = wave; j < npieces; j += 8) {
            const int r = 8 * j + r8, g = g0 + r;
            const bool ok = g >= 0 && g < HW;
            // rows above / below the image: an ordinary LDS store of zeros by the lanes concerned (the LDS-DMA runs with them switched off)
            if (ok) h3_glds16(img, (unsigned)(((size_t)g * p.Cin + cc * 64) * 2 + ((cpos ^ (r & 7)) << 4)),
                              __builtin_amdgcn_readfirstlane(smem_lds + buf * kH3PatchBytes + j * 1024));
            else *(u32x4 *)(smem + buf * kH3PatchBytes + j * 1024 + lane * 16) = u32x4{0u, 0u, 0u, 0u};
        }
        for (int j = wave; j < 18; j += 8) {              // weight rows (tap, column): 144 rows of 128 B = 18 pieces
            const int row = 8 * j + r8, tap = row >> 4, co = row & 15;
            const unsigned voff = (unsigned)(((size_t)(co * 9 + tap) * p.Cin + cc * 64) * 2 + ((cpos ^ (row & 7)) << 4));
            h3_glds16(p.w, voff, __builtin_amdgcn_readfirstlane(smem_lds + OFF_W + buf * kH3WBytes + j * 1024));
        }
    };
    // ---- my lanes' roles ----
    // A operand (weights): lane = column (lane & 15), channels 8 (lane >> 4) .. +8 of the 32-channel K step
    // B operand (pixels):  lane = pixel (lane & 15) of a 16-pixel block, the same channels
    // D: lane = pixel (lane & 15), registers = columns 4 (lane >> 4) .. +4
    const int l16 = lane & 15, kg = lane >> 4;
    int prow[2], px_[2];                                  // patch row of my pixel (tap 0,0 at +W+1), its x
    bool left_ok[2], right_ok[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int pl = wave * 32 + nb * 16 + l16;         // pixel inside the tile
        const int pi = p0 + pl;                           // ... inside the image
        const int x = pi % W;
        prow[nb] = pl + W + 1;
        px_[nb] = x;
        left_ok[nb] = x > 0;
        right_ok[nb] = x < W - 1;
    }
    f32x4 acc[2];
    {
        const f32x4 bv = *(const f32x4 *)(p.bias + 4 * kg);
        acc[0] = bv; acc[1] = bv;
    }
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int cc = 0; cc < nchunks; ++cc) {
        const int buf = cc & 1;
        if (cc + 1 < nchunks) stage(cc + 1, buf ^ 1);     // (its last readers passed the barrier that ended chunk cc - 1)
        const char *pb = smem + buf * kH3PatchBytes, *wb = smem + OFF_W + buf * kH3WBytes;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3 - 1, dx = tap % 3 - 1;
            const int shift = dy * W + dx;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int chunk = ks * 4 + kg;            // 16-byte chunk of the 128-byte row
                const int wrow = tap * 16 + l16;
                const u32x4 wf = *(const u32x4 *)(wb + wrow * 128 + ((chunk ^ (wrow & 7)) << 4));
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    const int r = prow[nb] + shift;
                    const bool ok = dx == 0 || (dx < 0 ? left_ok[nb] : right_ok[nb]);
                    const char *src = ok ? pb + r * 128 + ((chunk ^ (r & 7)) << 4) : smem + OFF_Z + (kg << 4);
                    const u32x4 pf = *(const u32x4 *)src;
                    acc[nb] = Mma16<DT>::run(wf, pf, acc[nb]);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the next chunk's pieces (issued a chunk's arithmetic ago) have landed
        __syncthreads();
    }
    // ---- epilogue: lane = pixel, four consecutive columns ----
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int pi = p0 + wave * 32 + nb * 16 + l16;
        if (pi >= HW) continue;
        const int y = pi / W, x = pi - y * W;
        float *dst = p.out + (size_t)b * p.o_bs + (size_t)y * p.o_rs + (size_t)x * p.o_cs + 4 * kg;
        f32x4 v = acc[nb];
        if (p.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
        const int left = p.Cout - 4 * kg;
        if (left >= 4 && p.vec16) *(f32x4 *)dst = v;
        else
            for (int i = 0; i < left; ++i) dst[i] = v[i];
    }
}

int head3x3_supported(const ConvArgs &a)
{
    if (a.kdisable & 256) return 0;
    if (a.dtype == TDRN_F32 || a.kh != 3 || a.kw != 3 || a.stride != 1 || a.pad != 1 || a.dil != 1) return 0;
    if (a.phases != 1 || !a.out_f32 || a.res || a.fuse_x || a.fuse_x8) return 0;
    if (a.Ho != a.H || a.Wo != a.W || a.W > kH3MaxW || a.W < 2) return 0;
    // a level has to give every image at least two tiles' worth of pixels: the 10x10 / 5x5 levels (one 39 % / 10 % filled tile per image,
    // 32 workgroups at batch 32) measured 32 / 18 us here against 25 / 19 us on conv_igemm.hip with its split-K (geometry only, never
    // the batch: a frame's result must not depend on what else is in the batch)
    if (a.H * a.W < 400) return 0;
    if (a.Cout < 1 || a.Cout > 16 || a.Npad < 16 || a.Cin % 64) return 0;
    if ((long long)a.H * a.W * a.Cin * 2 >= (1ll << 32)) return 0;                 // 32-bit byte offsets inside an image
    if ((long long)a.Npad * 9 * a.Cin * 2 >= (1ll << 32)) return 0;
    // 16-byte stores: four consecutive columns of a pixel.  Geometry only -- the caller's POINTER alignment must not choose the
    // kernel (head3x3 and conv_igemm differ in fp32 K order, and a frame's arithmetic depends on geometry alone): a base that is
    // not 16-byte aligned (a sliced `out=` tensor) keeps this kernel and stores the four columns one by one (p.vec16 = 0).
    if ((a.o_base | a.o_bs | a.o_rs | a.o_cs) & 3) return 0;
    return 1;
}

int launch_head3x3(const ConvArgs &a, hipStream_t s)
{
    if (!head3x3_supported(a)) return TDRN_E_UNSUPPORTED;
    if (!a.in || !a.w || !a.out || !a.bias) return TDRN_E_ARG;
    Head3Params p;
    p.in = (const char *)a.in; p.w = (const char *)a.w; p.bias = a.bias;
    p.out = (float *)a.out + a.o_base;
    p.o_bs = a.o_bs; p.o_rs = a.o_rs; p.o_cs = a.o_cs;
    p.vec16 = (((size_t)p.out) & 15) == 0;
    p.B = a.B; p.H = a.H; p.W = a.W; p.Cin = a.Cin; p.Cout = a.Cout; p.relu = a.relu;
    p.tiles_per_img = (a.H * a.W + 255) / 256;
    const long long blocks = (long long)a.B * p.tiles_per_img;
    if (blocks <= 0) return TDRN_OK;
    if (blocks >= (1ll << 31)) return TDRN_E_UNSUPPORTED;
    if (a.dtype == TDRN_BF16) hipLaunchKernelGGL((head3x3_kernel<bf16_t>), dim3((unsigned)blocks), dim3(512), 0, s, p);
    else hipLaunchKernelGGL((head3x3_kernel<f16_t>), dim3((unsigned)blocks), dim3(512), 0, s, p);
    return hip_status(hipGetLastError());
}

}  // namespace tdrn
