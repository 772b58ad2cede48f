// detect.hip -- Detect.forward on the device, batched over (image, class).
//
// Replaces layers/functions/detection.py:25-70 (python loops over images and classes, two PCIe
// hops and a serial O(n^2) Cython NMS per class) and its helpers layers/box_utils.py:176-195
// (decode), :16-25 (center_size), utils/nms/cpu_nms.pyx:17-68 (cpu_nms; suppression on
// IoU >= thresh, "+1" pixel convention) and utils/nms/nms_kernel.cu:24-144 (strict > twin).
//
// Bit-exactness: every fp32 operation is issued un-fused and in the reference's order
// (__fmul_rn / __fadd_rn / __fsub_rn / __fdiv_rn), the IoU-vs-threshold test compares the fp32
// IoU against the DOUBLE threshold (pre-rounded on the host to the equivalent fp32 bound), and
// candidates are ordered by (score desc, prior index asc).  Given identical fp32 boxes/scores
// the keep lists equal cpu_nms's on tie-free scores.
//
// Pipeline (3 launches + 1 memset, no host sync):
//   detect_decode_kernel   : two-stage decode, normalised boxes + boxes*scale
//   detect_nms_kernel      : per (image,class): score > conf_thresh compaction into LDS keys,
//                            bitonic sort, wave-64 greedy NMS against an LDS-resident keep list,
//                            early exit at top_k, pack output rows
#include "kernels.h"

namespace tdrn {

__device__ __forceinline__ void decode_one(const float *l, const float *pr, float v0, float v1, float *b)
{
    const float cx = __fadd_rn(pr[0], __fmul_rn(__fmul_rn(l[0], v0), pr[2]));
    const float cy = __fadd_rn(pr[1], __fmul_rn(__fmul_rn(l[1], v0), pr[3]));
    const float w = __fmul_rn(pr[2], expf(__fmul_rn(l[2], v1)));
    const float h = __fmul_rn(pr[3], expf(__fmul_rn(l[3], v1)));
    const float x1 = __fsub_rn(cx, __fdiv_rn(w, 2.f));
    const float y1 = __fsub_rn(cy, __fdiv_rn(h, 2.f));
    b[0] = x1;
    b[1] = y1;
    b[2] = __fadd_rn(w, x1);
    b[3] = __fadd_rn(h, y1);
}
__device__ __forceinline__ void center_size_one(const float *b, float *o)
{
    o[0] = __fdiv_rn(__fadd_rn(b[2], b[0]), 2.f);
    o[1] = __fdiv_rn(__fadd_rn(b[3], b[1]), 2.f);
    o[2] = __fsub_rn(b[2], b[0]);
    o[3] = __fsub_rn(b[3], b[1]);
}

__global__ __launch_bounds__(256) void decode_kernel(const float *__restrict__ loc, const float *__restrict__ priors,
                                                     int P, float v0, float v1, float *__restrict__ out)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const f32x4 l = *(const f32x4 *)(loc + (size_t)p * 4), pr = *(const f32x4 *)(priors + (size_t)p * 4);
    float lf[4] = {l[0], l[1], l[2], l[3]}, pf[4] = {pr[0], pr[1], pr[2], pr[3]}, b[4];
    decode_one(lf, pf, v0, v1, b);
    *(f32x4 *)(out + (size_t)p * 4) = f32x4{b[0], b[1], b[2], b[3]};
}
__global__ __launch_bounds__(256) void center_size_kernel(const float *__restrict__ boxes, int P, float *__restrict__ out)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const f32x4 b = *(const f32x4 *)(boxes + (size_t)p * 4);
    float bf[4] = {b[0], b[1], b[2], b[3]}, o[4];
    center_size_one(bf, o);
    *(f32x4 *)(out + (size_t)p * 4) = f32x4{o[0], o[1], o[2], o[3]};
}
int launch_decode(const float *loc, const float *priors, int P, float v0, float v1, float *out, hipStream_t s)
{
    if (P <= 0) return TDRN_OK;
    hipLaunchKernelGGL(decode_kernel, dim3(cdiv(P, 256)), dim3(256), 0, s, loc, priors, P, v0, v1, out);
    return hip_status(hipGetLastError());
}
int launch_center_size(const float *boxes, int P, float *out, hipStream_t s)
{
    if (P <= 0) return TDRN_OK;
    hipLaunchKernelGGL(center_size_kernel, dim3(cdiv(P, 256)), dim3(256), 0, s, boxes, P, out);
    return hip_status(hipGetLastError());
}

// detection.py:43-48 (+ :59 boxes*scale)
__global__ __launch_bounds__(256) void detect_decode_kernel(const float *__restrict__ loc, const float *__restrict__ arm,
                                                            const float *__restrict__ priors, int B, int P, f32x4 scale,
                                                            float *__restrict__ boxes, float *__restrict__ sboxes)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)B * P) return;
    const int p = (int)(i % P);
    const f32x4 pr = *(const f32x4 *)(priors + (size_t)p * 4);
    float anchor[4] = {pr[0], pr[1], pr[2], pr[3]};
    if (arm) {
        const f32x4 a = *(const f32x4 *)(arm + (size_t)i * 4);
        float af[4] = {a[0], a[1], a[2], a[3]}, t[4];
        decode_one(af, anchor, 0.1f, 0.2f, t);
        center_size_one(t, anchor);
    }
    const f32x4 l = *(const f32x4 *)(loc + (size_t)i * 4);
    float lf[4] = {l[0], l[1], l[2], l[3]}, b[4];
    decode_one(lf, anchor, 0.1f, 0.2f, b);
    *(f32x4 *)(boxes + (size_t)i * 4) = f32x4{b[0], b[1], b[2], b[3]};
    *(f32x4 *)(sboxes + (size_t)i * 4) =
        f32x4{__fmul_rn(b[0], scale[0]), __fmul_rn(b[1], scale[1]), __fmul_rn(b[2], scale[2]), __fmul_rn(b[3], scale[3])};
}

// ---- shared device pieces -------------------------------------------------------------------
__device__ __forceinline__ void bitonic_sort_desc(unsigned long long *k, int N, int tid, int nthreads)
{
    for (int kk = 2; kk <= N; kk <<= 1) {
        for (int j = kk >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < (N >> 1); i += nthreads) {
                const int a = 2 * i - (i & (j - 1)), bidx = a + j;
                const unsigned long long x = k[a], y = k[bidx];
                const bool desc = (a & kk) == 0;
                if ((x < y) == desc) { k[a] = y; k[bidx] = x; }
            }
            __syncthreads();
        }
    }
}

struct Box { float x1, y1, x2, y2, area; };
__device__ __forceinline__ float box_area(float x1, float y1, float x2, float y2)
{   // cpu_nms.pyx:24  (x2 - x1 + 1) * (y2 - y1 + 1)
    return __fmul_rn(__fadd_rn(__fsub_rn(x2, x1), 1.f), __fadd_rn(__fsub_rn(y2, y1), 1.f));
}
// cpu_nms.pyx:55-65, i = the kept (higher-score) box, j = the candidate
__device__ __forceinline__ float iou_plus1(const Box &i, const Box &j)
{
    const float xx1 = i.x1 >= j.x1 ? i.x1 : j.x1;
    const float yy1 = i.y1 >= j.y1 ? i.y1 : j.y1;
    const float xx2 = i.x2 <= j.x2 ? i.x2 : j.x2;
    const float yy2 = i.y2 <= j.y2 ? i.y2 : j.y2;
    float w = __fadd_rn(__fsub_rn(xx2, xx1), 1.f);
    float h = __fadd_rn(__fsub_rn(yy2, yy1), 1.f);
    w = 0.f >= w ? 0.f : w;
    h = 0.f >= h ? 0.f : h;
    const float inter = __fmul_rn(w, h);
    return __fdiv_rn(inter, __fsub_rn(__fadd_rn(i.area, j.area), inter));
}
__device__ __forceinline__ bool over(float ovr, float bound, int strict) { return strict ? (ovr > bound) : (ovr >= bound); }

// One wavefront: greedy NMS over `n` candidates given in descending order.  get(pos, box) loads
// candidate pos.  kept[] (capacity cap) holds the boxes kept so far; emit(slot, pos) records a
// survivor.  Stops when cap survivors exist (cap = n reproduces the full cpu_nms list).
template <typename Get, typename Emit>
__device__ __forceinline__ int wave_greedy_nms(int n, int cap, float bound, int strict, Box *kept, Get get, Emit emit)
{
    const int lane = threadIdx.x & 63;
    int nk = 0;
    for (int c0 = 0; c0 < n && nk < cap; c0 += 64) {
        const int pos = c0 + lane;
        const bool valid = pos < n;
        Box me = {0.f, 0.f, 0.f, 0.f, 1.f};
        if (valid) get(pos, me);
        bool alive = valid;
        for (int i = 0; i < nk && alive; ++i)
            if (over(iou_plus1(kept[i], me), bound, strict)) alive = false;
        // pairwise suppression inside the chunk: bit i of sup = "candidate i (i < lane) suppresses me"
        unsigned long long sup = 0ull;
        for (int i = 0; i < 63; ++i) {
            Box o;
            o.x1 = __shfl(me.x1, i, 64); o.y1 = __shfl(me.y1, i, 64);
            o.x2 = __shfl(me.x2, i, 64); o.y2 = __shfl(me.y2, i, 64); o.area = __shfl(me.area, i, 64);
            if (i < lane && over(iou_plus1(o, me), bound, strict)) sup |= 1ull << i;
        }
        const unsigned long long amask = __ballot(alive);
        unsigned long long km = 0ull;
        for (int i = 0; i < 64; ++i) {
            const unsigned lo = __builtin_amdgcn_readlane((unsigned)sup, i);
            const unsigned hi = __builtin_amdgcn_readlane((unsigned)(sup >> 32), i);
            const unsigned long long s_i = ((unsigned long long)hi << 32) | lo;
            if (((amask >> i) & 1ull) && (s_i & km) == 0ull) km |= 1ull << i;
        }
        const bool keepme = (km >> lane) & 1ull;
        const int slot = nk + __popcll(km & ((1ull << lane) - 1ull));
        if (keepme && slot < cap) {
            kept[slot] = me;
            emit(slot, pos);
        }
        nk += __popcll(km);
        __threadfence_block();
        __builtin_amdgcn_wave_barrier();
    }
    return nk < cap ? nk : cap;
}

// Workgroup version for Detect (256 threads): candidates are taken 256 at a time; every thread first
// tests its candidate against the keep list as it stood at the start of the round (the long scan, now
// spread over 4 waves), then wave 0 resolves the round 64 candidates at a time against only the boxes
// kept DURING this round plus the pairwise bits.  Same decisions as the sequential algorithm.
template <typename Get, typename Emit>
__device__ __forceinline__ int block_greedy_nms(int n, int cap, float bound, Box *kept, unsigned char *alive_s, int *nk_s,
                                                Get get, Emit emit)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) *nk_s = 0;
    __syncthreads();
    for (int c0 = 0; c0 < n; c0 += 256) {
        const int nk0 = *nk_s;
        if (nk0 >= cap) break;
        const int pos = c0 + tid;
        const bool valid = pos < n;
        Box me = {0.f, 0.f, 0.f, 0.f, 1.f};
        if (valid) get(pos, me);
        bool alive = valid;
        for (int i = 0; i < nk0 && alive; ++i)
            if (over(iou_plus1(kept[i], me), bound, 0)) alive = false;
        alive_s[tid] = alive ? 1 : 0;
        __syncthreads();
        if (wave == 0) {
            int nk = nk0;
            for (int sub = 0; sub < 4 && nk < cap && c0 + sub * 64 < n; ++sub) {
                const int p2 = c0 + sub * 64 + lane;
                const bool v2 = p2 < n;
                Box b2 = {0.f, 0.f, 0.f, 0.f, 1.f};
                if (v2) get(p2, b2);
                bool al = v2 && alive_s[sub * 64 + lane];
                for (int i = nk0; i < nk && al; ++i)        // boxes kept earlier in this round
                    if (over(iou_plus1(kept[i], b2), bound, 0)) al = false;
                unsigned long long sup = 0ull;
                for (int i = 0; i < 63; ++i) {
                    Box o;
                    o.x1 = __shfl(b2.x1, i, 64); o.y1 = __shfl(b2.y1, i, 64);
                    o.x2 = __shfl(b2.x2, i, 64); o.y2 = __shfl(b2.y2, i, 64); o.area = __shfl(b2.area, i, 64);
                    if (i < lane && over(iou_plus1(o, b2), bound, 0)) sup |= 1ull << i;
                }
                const unsigned long long amask = __ballot(al);
                unsigned long long km = 0ull;
                for (int i = 0; i < 64; ++i) {
                    const unsigned lo = __builtin_amdgcn_readlane((unsigned)sup, i);
                    const unsigned hi = __builtin_amdgcn_readlane((unsigned)(sup >> 32), i);
                    const unsigned long long s_i = ((unsigned long long)hi << 32) | lo;
                    if (((amask >> i) & 1ull) && (s_i & km) == 0ull) km |= 1ull << i;
                }
                const bool keepme = (km >> lane) & 1ull;
                const int slot = nk + __popcll(km & ((1ull << lane) - 1ull));
                if (keepme && slot < cap) {
                    kept[slot] = b2;
                    emit(slot, p2);
                }
                nk += __popcll(km);
                __threadfence_block();
                __builtin_amdgcn_wave_barrier();
            }
            if (lane == 0) *nk_s = nk;
        }
        __syncthreads();
    }
    const int nk = *nk_s;
    return nk < cap ? nk : cap;
}

// ---- Detect: one workgroup per (image, class) segment -------------------------------------
// detection.py:52-63.  The workgroup scans its class column of conf (score > conf_thresh, :53),
// compacts the candidates into LDS as 64-bit keys (score bits : ~prior index), sorts them and runs the
// greedy NMS against an LDS-resident keep list.  Two launches: the FAST one holds at most `kcap`
// (2048) candidates in 16 KB of LDS so that every segment of a batch is resident at once; segments with
// more candidates raise overflow[seg] and are redone by the second launch with a P-sized key buffer
// (whose other workgroups exit at once).
__global__ __launch_bounds__(256) void detect_nms_kernel(const float *__restrict__ boxes, const float *__restrict__ sboxes,
                                                         const float *__restrict__ conf, int P, int C, int top_k,
                                                         float conf_thresh, float bound, int kcap, int pass,
                                                         int *__restrict__ overflow, float *__restrict__ out,
                                                         int *__restrict__ counts_out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long dsm[];
    unsigned long long *sk = dsm;
    Box *kept = (Box *)(dsm + kcap);
    int *cnt = (int *)(kept + top_k);
    unsigned char *alive_s = (unsigned char *)(cnt + 4);
    const int seg = blockIdx.x;          // b*C + cl
    const int cl = seg % C, b = seg / C;
    if (cl == 0) {
        if (pass == 0 && counts_out && threadIdx.x == 0) counts_out[seg] = 0;
        return;
    }
    if (pass == 1 && !overflow[seg]) return;
    if (threadIdx.x == 0) *cnt = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const float *col = conf + (size_t)b * P * C + cl;
    for (int p0 = 0; p0 < P; p0 += 256) {
        const int p = p0 + threadIdx.x;
        const float sc = p < P ? col[(size_t)p * C] : 0.f;
        const bool pass_thr = p < P && sc > conf_thresh;
        const unsigned long long m = __ballot(pass_thr);
        int base = 0;
        if (lane == 0 && m) base = atomicAdd(cnt, __popcll(m));
        base = __shfl(base, 0, 64);
        const int idx = base + __popcll(m & ((1ull << lane) - 1ull));
        if (pass_thr && idx < kcap)
            sk[idx] = ((unsigned long long)__float_as_uint(sc) << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)p);
    }
    __syncthreads();
    const int n = *cnt;
    if (pass == 0 && threadIdx.x == 0) overflow[seg] = n > kcap ? 1 : 0;
    if (n == 0 || n > kcap) {
        if (n == 0 && counts_out && threadIdx.x == 0) counts_out[seg] = 0;
        return;
    }
    int N = 64;
    while (N < n) N <<= 1;
    for (int i = n + threadIdx.x; i < N; i += 256) sk[i] = 0ull;
    __syncthreads();
    bitonic_sort_desc(sk, N, threadIdx.x, 256);
    const float *sb = sboxes + (size_t)b * P * 4;
    const float *nb = boxes + (size_t)b * P * 4;
    float *orow = out + (size_t)seg * top_k * 5;
    const int nk = block_greedy_nms(
        n, top_k, bound, kept, alive_s, cnt + 1,
        [&](int pos, Box &bx) {
            const unsigned p = 0xFFFFFFFFu - (unsigned)(sk[pos] & 0xFFFFFFFFull);
            const f32x4 v = *(const f32x4 *)(sb + (size_t)p * 4);
            bx.x1 = v[0]; bx.y1 = v[1]; bx.x2 = v[2]; bx.y2 = v[3];
            bx.area = box_area(v[0], v[1], v[2], v[3]);
        },
        [&](int slot, int pos) {
            const unsigned long long key = sk[pos];
            const unsigned p = 0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull);
            const f32x4 v = *(const f32x4 *)(nb + (size_t)p * 4);
            float *o = orow + (size_t)slot * 5;
            o[0] = __uint_as_float((unsigned)(key >> 32));
            o[1] = v[0]; o[2] = v[1]; o[3] = v[2]; o[4] = v[3];
        });
    if (counts_out && threadIdx.x == 0) counts_out[seg] = nk;
}

static int next_pow2(int v) { int n = 64; while (n < v) n <<= 1; return n; }

size_t detect_workspace_bytes(int B, int P, int C, int top_k)
{
    (void)top_k;
    return align_up((size_t)B * P * 4 * sizeof(float), 256) * 2 + align_up((size_t)B * C * sizeof(int), 256);   // boxes, boxes*scale, overflow flags
}

int launch_detect(const float *loc, const float *conf, const float *priors, const float *arm_loc, const float *scale4,
                  int B, int P, int C, int top_k, float conf_thresh, double nms_thresh, float *out, int32_t *counts_out,
                  void *ws, size_t ws_bytes, hipStream_t s)
{
    if (!loc || !conf || !priors || !scale4 || !out || !ws) return TDRN_E_ARG;
    if (nms_thresh <= 0) return TDRN_E_VALUE;
    if (B <= 0 || P <= 0 || C < 2 || top_k <= 0) return TDRN_E_ARG;
    if (ws_bytes < detect_workspace_bytes(B, P, C, top_k)) return TDRN_E_WORKSPACE;
    const int kcap_big = next_pow2(P);
    const int kcap_fast = kcap_big < 2048 ? kcap_big : 2048;
    const size_t tail = (size_t)top_k * sizeof(Box) + 32 + 256;      // keep list, counters, alive flags
    const size_t lds_big = (size_t)kcap_big * 8 + tail, lds_fast = (size_t)kcap_fast * 8 + tail;
    if (lds_big > 160 * 1024) return TDRN_E_UNSUPPORTED;
    char *w = (char *)ws;
    float *boxes = (float *)w;  w += align_up((size_t)B * P * 4 * sizeof(float), 256);
    float *sboxes = (float *)w; w += align_up((size_t)B * P * 4 * sizeof(float), 256);
    int *overflow = (int *)w;
    // (double)ovr >= thresh  <=>  ovr >= bound, bound = smallest fp32 whose double value >= thresh
    float bound = (float)nms_thresh;
    if ((double)bound < nms_thresh) bound = nextafterf(bound, INFINITY);
    TDRN_HIP_TRY(hipMemsetAsync(out, 0, (size_t)B * C * top_k * 5 * sizeof(float), s));
    const long long bp = (long long)B * P;
    hipLaunchKernelGGL(detect_decode_kernel, dim3((unsigned)((bp + 255) / 256)), dim3(256), 0, s, loc, arm_loc, priors, B, P,
                       f32x4{scale4[0], scale4[1], scale4[2], scale4[3]}, boxes, sboxes);
    static bool attr_set = false;
    if (!attr_set) {
        TDRN_HIP_TRY(hipFuncSetAttribute((const void *)detect_nms_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL(detect_nms_kernel, dim3((unsigned)(B * C)), dim3(256), lds_fast, s, boxes, sboxes, conf, P, C, top_k,
                       conf_thresh, bound, kcap_fast, 0, overflow, out, counts_out);
    if (kcap_big > kcap_fast)
        hipLaunchKernelGGL(detect_nms_kernel, dim3((unsigned)(B * C)), dim3(256), lds_big, s, boxes, sboxes, conf, P, C, top_k,
                           conf_thresh, bound, kcap_big, 1, overflow, out, counts_out);
    return hip_status(hipGetLastError());
}

// ---- stand-alone NMS (cpu_nms / gpu_nms twins): one workgroup, keep list in global memory ------
__global__ __launch_bounds__(256) void nms_plain_kernel(const float *__restrict__ dets, int n, float bound, int strict,
                                                        int presorted, int kcap, Box *__restrict__ kept,
                                                        int *__restrict__ keep_out, int *__restrict__ num_out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long dsm[];
    unsigned long long *sk = dsm;
    int N = 64;
    while (N < n) N <<= 1;
    for (int i = threadIdx.x; i < N; i += 256) {
        unsigned long long k = 0ull;
        if (i < n) {
            // presorted: keep the caller's order (gpu_nms.pyx:25-28 sorts on the host)
            const unsigned hi = presorted ? (unsigned)(n - i) : __float_as_uint(dets[(size_t)i * 5 + 4]);
            k = ((unsigned long long)hi << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)i);
        }
        sk[i] = k;
    }
    __syncthreads();
    if (!presorted) bitonic_sort_desc(sk, N, threadIdx.x, 256);
    if (threadIdx.x >= 64) return;
    const int nk = wave_greedy_nms(
        n, n, bound, strict, kept,
        [&](int pos, Box &bx) {
            const unsigned p = 0xFFFFFFFFu - (unsigned)(sk[pos] & 0xFFFFFFFFull);
            const float *d = dets + (size_t)p * 5;
            bx.x1 = d[0]; bx.y1 = d[1]; bx.x2 = d[2]; bx.y2 = d[3];
            bx.area = box_area(d[0], d[1], d[2], d[3]);
        },
        [&](int slot, int pos) { keep_out[slot] = (int)(0xFFFFFFFFu - (unsigned)(sk[pos] & 0xFFFFFFFFull)); });
    if (threadIdx.x == 0) *num_out = nk;
    (void)kcap;
}

size_t nms_workspace_bytes(int n) { return align_up((size_t)(n > 0 ? n : 1) * sizeof(Box), 256); }

int launch_nms(const float *dets, int n, double thresh, int strict_gt, int presorted, int32_t *keep_out, int32_t *num_out,
               void *ws, size_t ws_bytes, hipStream_t s)
{
    if (!num_out || n < 0) return TDRN_E_ARG;
    if (n == 0) return hip_status(hipMemsetAsync(num_out, 0, sizeof(int), s));
    if (!dets || !keep_out || !ws) return TDRN_E_ARG;
    if (ws_bytes < nms_workspace_bytes(n)) return TDRN_E_WORKSPACE;
    const int kcap = next_pow2(n);
    if ((size_t)kcap * 8 > 128 * 1024) return TDRN_E_UNSUPPORTED;   // n <= 16384
    float bound = (float)thresh;
    if (strict_gt) { if ((double)bound > thresh) bound = nextafterf(bound, -INFINITY); }
    else { if ((double)bound < thresh) bound = nextafterf(bound, INFINITY); }
    static bool attr_set = false;
    if (!attr_set) {
        TDRN_HIP_TRY(hipFuncSetAttribute((const void *)nms_plain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL(nms_plain_kernel, dim3(1), dim3(256), (size_t)kcap * 8, s, dets, n, bound, strict_gt, presorted, kcap,
                       (Box *)ws, keep_out, num_out);
    return hip_status(hipGetLastError());
}

}  // namespace tdrn
