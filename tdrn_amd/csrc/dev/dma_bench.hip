// dev/dma_bench.hip -- developer micro-benchmark (never shipped): what an LDS-DMA stream of 1-KiB pieces sustains per CU on
// gfx950, by issue form (compiler builtin / inline asm with and without the M0 restore), source shape (contiguous 1 KiB /
// 8 rows x 128 B at a row stride) and source locality (a table every CU shares = L2-served, or private streams).
//   hipcc -O3 --offload-arch=gfx950 dev/dma_bench.hip -o _build/dma_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

struct P { const char *src; unsigned row_stride; unsigned table_bytes; unsigned cu_stride; int iters; };

template <int FORM> __device__ __forceinline__ void piece(const char *sbase, unsigned voff, unsigned lds_dst, char *lds_ptr)
{
    if constexpr (FORM == 0) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(sbase + voff),
                                         (__attribute__((address_space(3))) void *)lds_ptr, 16, 0, 0);
    } else if constexpr (FORM == 1) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
    } else {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
    }
}

// FORM: 0 builtin, 1 asm + M0 restore, 2 asm; SHAPE: 0 contiguous 1 KiB, 1 = 8 rows x 128 B at row_stride; K = pieces left in flight
template <int FORM, int SHAPE, int K> __global__ __launch_bounds__(512, 2) void dma_kernel(const P p)
{
    __shared__ __attribute__((aligned(16))) char smem[128 * 1024];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char *)smem;
    const unsigned lane_off = SHAPE == 0 ? (unsigned)lane * 16u : (unsigned)(lane >> 3) * p.row_stride + (unsigned)(lane & 7) * 16u;
    const unsigned piece_bytes = SHAPE == 0 ? 1024u : 8u * p.row_stride;
    const char *base = p.src + (size_t)blockIdx.x * p.cu_stride;
    unsigned cur = (unsigned)wave * 4u * piece_bytes;
    for (int i = 0; i < p.iters; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int slot = (wave * 16 + ((i & 3) * 4 + j)) * 1024;
            piece<FORM>(base, cur + lane_off, __builtin_amdgcn_readfirstlane(lds0 + slot), smem + slot);
            cur += piece_bytes;
            if (SHAPE == 1 && (j & 1)) cur += 0;   // (rows of one step are 8-row groups, back to back)
        }
        if (cur + 4u * piece_bytes + piece_bytes > p.table_bytes) cur = (unsigned)wave * 4u * piece_bytes;
        if constexpr (K == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if constexpr (K == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (smem[threadIdx.x] == 123 && p.iters < 0) ((char *)p.src)[0] = 1;   // keep LDS alive
}

// the conv3x3_pp skeleton: two wave groups alternate between barriers; per step a wave issues 3 pieces in one interval and 2 in
// another, then waits vmcnt(2).  MFMA = 1: 16 bare MFMAs in the issuing intervals (4 between pieces), as the kernel does.
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) short i16x8;
template <int MFMA, int NBAR> __global__ __launch_bounds__(512, 2) void pp_skeleton(const P p, float *sink)
{
    __shared__ __attribute__((aligned(16))) char smem[128 * 1024];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wave >> 2;
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char *)smem;
    const unsigned lane_off = (unsigned)(lane >> 3) * p.row_stride + (unsigned)(lane & 7) * 16u;
    const unsigned piece_bytes = 8u * p.row_stride;
    const char *base = p.src;
    unsigned cur = (unsigned)wave * 5u * piece_bytes;
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    i16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (short)(lane * 3 + j); b[j] = (short)(lane + 7 * j); }
#define BAR() do { __builtin_amdgcn_sched_barrier(0); if (NBAR) __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define MM4(o) do { if (MFMA) { for (int q = 0; q < 4; ++q) acc[(o) + q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[(o) + q], 0, 0, 0); } __builtin_amdgcn_sched_barrier(0); } while (0)
#define PIECE(j) do { const int slot = (wave * 16 + (j)) * 1024; piece<2>(base, cur + lane_off, __builtin_amdgcn_readfirstlane(lds0 + slot), smem + slot); cur += piece_bytes; __builtin_amdgcn_sched_barrier(0); } while (0)
    if (grp == 1) BAR();
    for (int i = 0; i < p.iters; ++i) {
        BAR();                                          // (load segment: nothing here)
        __builtin_amdgcn_s_setprio(1);
        MM4(0); PIECE(0); MM4(4); PIECE(1); MM4(0); PIECE(2); MM4(4);
        __builtin_amdgcn_s_setprio(0);
        BAR();
        BAR();
        __builtin_amdgcn_s_setprio(1);
        MM4(0); PIECE(3); MM4(4); PIECE(4); MM4(0); MM4(4);
        __builtin_amdgcn_s_setprio(0);
        if (cur + 6u * piece_bytes > p.table_bytes) cur = (unsigned)wave * 5u * piece_bytes;
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        BAR();
    }
    if (grp == 0) BAR();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float t = 0;
    for (int i = 0; i < 8; ++i) t += acc[i][lane & 15];
    if (t == 12345.f) sink[threadIdx.x] = t + smem[threadIdx.x];
}
// schedule B of conv3x3_pp: load segments = 12 ds_read_b128 (+ 3 / 2 pieces), multiply segments = 16 bare MFMAs
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
template <int DMA, int READS> __global__ __launch_bounds__(512, 2) void pp_skeleton2(const P p, float *sink)
{
    __shared__ __attribute__((aligned(16))) char smem[150 * 1024];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wave >> 2;
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char *)smem;
    const unsigned lane_off = (unsigned)(lane >> 3) * p.row_stride + (unsigned)(lane & 7) * 16u;
    const unsigned piece_bytes = 8u * p.row_stride;
    const char *base = p.src;
    unsigned cur = (unsigned)wave * 5u * piece_bytes;
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    u32x4 fr[12];
    const int r32 = lane & 31, hh = lane >> 5;
    const unsigned ra = (unsigned)(88 * 1024 + grp * 16384 + r32 * 128 + ((hh ^ ((r32 >> 1) & 7)) << 4));   // "weights"
    const unsigned rp = (unsigned)((wave & 3) * 64 * 128 + r32 * 128 + ((hh ^ ((r32 >> 1) & 7)) << 4));      // "patch"
#define BAR2() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define RD(h) do { if (READS) { for (int k2 = 0; k2 < 2; ++k2) { const unsigned kx = (unsigned)((2 * (h) + k2) << 5); \
        for (int ci = 0; ci < 4; ++ci) fr[k2 * 6 + ci] = *(const u32x4 *)(smem + ((ra ^ kx) + ci * 4096)); \
        fr[k2 * 6 + 4] = *(const u32x4 *)(smem + (rp ^ kx)); fr[k2 * 6 + 5] = *(const u32x4 *)(smem + ((rp + 4096) ^ kx)); } } __builtin_amdgcn_sched_barrier(0); } while (0)
#define PC(j) do { if (DMA) { const int slot = 88 * 1024 + ((i & 1) * 32768) + grp * 16384 + ((wave & 3) + 4 * ((j) & 3)) * 1024; piece<2>(base, cur + lane_off, __builtin_amdgcn_readfirstlane(lds0 + slot), smem + slot); cur += piece_bytes; } __builtin_amdgcn_sched_barrier(0); } while (0)
#define MM16() do { __builtin_amdgcn_s_setprio(1); for (int k2 = 0; k2 < 2; ++k2) for (int q = 0; q < 8; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(i16x8, fr[k2 * 6 + (q >> 1)]), __builtin_bit_cast(i16x8, fr[k2 * 6 + 4 + (q & 1)]), acc[q], 0, 0, 0); __builtin_amdgcn_s_setprio(0); __builtin_amdgcn_sched_barrier(0); } while (0)
    for (int j = 0; j < 12; ++j) fr[j] = u32x4{(unsigned)lane * 0x01010101u + j, 0x3c003c00u, 0x3c003c00u + j, 0x40004000u};
    if (grp == 1) BAR2();
    for (int i = 0; i < p.iters; ++i) {
        RD(0); PC(0); PC(1); PC(2);
        __builtin_amdgcn_s_waitcnt(0xC07F);
        BAR2();
        MM16();
        BAR2();
        RD(1); PC(3); PC(0);
        __builtin_amdgcn_s_waitcnt(0xC07F);
        BAR2();
        MM16();
        if (cur + 6u * piece_bytes > p.table_bytes) cur = (unsigned)wave * 5u * piece_bytes;
        if (DMA) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        BAR2();
    }
    if (grp == 0) BAR2();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float t = 0;
    for (int i = 0; i < 8; ++i) t += acc[i][lane & 15];
    if (t == 12345.f) sink[threadIdx.x] = t + smem[threadIdx.x];
}
template <int DMA, int READS> static void run_skel2(const char *name, const char *buf, unsigned row_stride, unsigned table_bytes, float *sink)
{
    P p{buf, row_stride, table_bytes, 0, 2000};
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((pp_skeleton2<DMA, READS>), dim3(256), dim3(512), 0, 0, p, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((pp_skeleton2<DMA, READS>), dim3(256), dim3(512), 0, 0, p, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-64s %8.1f us  %6.0f ns/step (MFMA floor at 2.4 GHz: 853)\n", name, ms * 1e3, ms * 1e6 / p.iters);
    fflush(stdout);
}

template <int MFMA, int NBAR> static void run_skel(const char *name, const char *buf, unsigned row_stride, unsigned table_bytes, float *sink)
{
    P p{buf, row_stride, table_bytes, 0, 2000};
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((pp_skeleton<MFMA, NBAR>), dim3(256), dim3(512), 0, 0, p, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((pp_skeleton<MFMA, NBAR>), dim3(256), dim3(512), 0, 0, p, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = 256.0 * 8 * p.iters * 5 * 1024;
    printf("%-64s %8.1f us  %7.2f TB/s  %6.1f GB/s per CU  %6.0f ns/step (MFMA floor at 2.4 GHz: 853)\n", name, ms * 1e3, bytes / (ms * 1e-3) / 1e12,
           bytes / 256 / (ms * 1e-3) / 1e9, ms * 1e6 / p.iters);
    fflush(stdout);
}

template <int FORM, int SHAPE, int K> static void run(const char *name, const char *buf, unsigned row_stride, unsigned table_bytes, unsigned cu_stride)
{
    P p{buf, row_stride, table_bytes, cu_stride, 2000};
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((dma_kernel<FORM, SHAPE, K>), dim3(256), dim3(512), 0, 0, p);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((dma_kernel<FORM, SHAPE, K>), dim3(256), dim3(512), 0, 0, p);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = 256.0 * 8 * p.iters * 4 * 1024;
    printf("%-64s %8.1f us  %7.2f TB/s  %6.1f GB/s per CU\n", name, ms * 1e3, bytes / (ms * 1e-3) / 1e12, bytes / 256 / (ms * 1e-3) / 1e9);
    fflush(stdout);
}

int main()
{
    char *buf;
    const size_t big = (size_t)1 << 30;
    CK(hipMalloc((void **)&buf, big));
    CK(hipMemset(buf, 1, big));
    // shared table (every CU reads the same 1.2 MB: L2-served), contiguous pieces
    run<0, 0, 4>("builtin  contiguous shared-1.2MB  vmcnt(4)", buf, 0, 1200000, 0);
    run<1, 0, 4>("asm+m0restore contiguous shared-1.2MB vmcnt(4)", buf, 0, 1200000, 0);
    run<2, 0, 4>("asm      contiguous shared-1.2MB  vmcnt(4)", buf, 0, 1200000, 0);
    run<2, 0, 0>("asm      contiguous shared-1.2MB  vmcnt(0)", buf, 0, 1200000, 0);
    run<2, 0, 8>("asm      contiguous shared-1.2MB  vmcnt(8)", buf, 0, 1200000, 0);
    // the weight slice shape: 8 rows x 128 B at 4608 / 9216 B (conv3 / conv4 weight rows), shared table
    run<0, 1, 4>("builtin  8x128B stride 4608 shared vmcnt(4)", buf, 4608, 1179648, 0);
    run<2, 1, 4>("asm      8x128B stride 4608 shared vmcnt(4)", buf, 4608, 1179648, 0);
    run<2, 1, 8>("asm      8x128B stride 4608 shared vmcnt(8)", buf, 4608, 1179648, 0);
    run<2, 1, 4>("asm      8x128B stride 9216 shared vmcnt(4)", buf, 9216, 4718592, 0);
    run<2, 1, 4>("asm      8x128B stride 512 (patch rows) shared vmcnt(4)", buf, 512, 1179648, 0);
    run<2, 1, 4>("asm      8x128B stride 4608+128 shared vmcnt(4)", buf, 4736, 1212416, 0);
    // private streams (each CU its own 4 MB region of a 1 GB buffer: beyond L2)
    run<2, 0, 4>("asm      contiguous private-4MB   vmcnt(4)", buf, 0, 4000000, 4u << 20);
    run<2, 0, 8>("asm      contiguous private-4MB   vmcnt(8)", buf, 0, 4000000, 4u << 20);
    run<2, 1, 8>("asm      8x128B stride 512 private-4MB vmcnt(8)", buf, 512, 4000000, 4u << 20);
    float *sink;
    CK(hipMalloc((void **)&sink, 4096));
    run_skel2<0, 0>("skeleton2: MFMA + barriers only", buf, 4608, 1179648, sink);
    run_skel2<1, 0>("skeleton2: + 5 DMA pieces in the load segments", buf, 4608, 1179648, sink);
    run_skel2<0, 1>("skeleton2: + 24 ds_read_b128 per wave and step", buf, 4608, 1179648, sink);
    run_skel2<1, 1>("skeleton2: + both", buf, 4608, 1179648, sink);
    run_skel<0, 1>("skeleton: barriers, no MFMA, stride 4608 shared", buf, 4608, 1179648, sink);
    run_skel<0, 0>("skeleton: NO barriers, no MFMA, stride 4608 shared", buf, 4608, 1179648, sink);
    run_skel<1, 1>("skeleton: barriers + 32 MFMA/step/wave, stride 4608 shared", buf, 4608, 1179648, sink);
    run_skel<1, 0>("skeleton: NO barriers + 32 MFMA/step/wave", buf, 4608, 1179648, sink);
    return 0;
}
