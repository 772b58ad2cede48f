// detect.hip -- Detect.forward on the device, batched over (image, class).
//
// Replaces layers/functions/detection.py:25-70 (python loops over images and classes, two PCIe
// hops and a serial O(n^2) Cython NMS per class) and its helpers layers/box_utils.py:176-195
// (decode), :16-25 (center_size), utils/nms/cpu_nms.pyx:17-68 (cpu_nms; suppression on
// IoU >= thresh, "+1" pixel convention) and utils/nms/nms_kernel.cu:24-144 (strict > twin).
//
// Bit-exactness: every fp32 operation is issued un-fused and in the reference's order
// (__fmul_rn / __fadd_rn / __fsub_rn / __fdiv_rn), the IoU-vs-threshold test compares the fp32
// IoU against the DOUBLE threshold (pre-rounded on the host to the equivalent fp32 bound, NmsRule), and
// candidates are ordered by (score desc, prior index asc).  Given identical fp32 boxes/scores the keep
// lists equal cpu_nms's on tie-free scores.
//
// Pipeline (2 launches, no memset, no host sync, any number of priors):
//   detect_decode_kernel     : two-stage decode, normalised boxes + boxes*scale; scores -> class-major rows
//   detect_select_nms_kernel : per (image,class): class row -> score keys (in LDS while 4*P bytes fit beside the
//                              rest, else in place over the class row in global memory); then, up to 1024
//                              candidates at a time in descending score (radix-select of a score threshold,
//                              compaction, bitonic sort), greedy NMS against an LDS-resident keep list until
//                              top_k boxes are kept or the candidates run out; pack + zero-fill the output rows.
//                              More than 1024 candidates with ONE score are taken 1024 at a time in ascending
//                              prior index (that is their place in the order), so LDS use never depends on P.
#include <cmath>
#include <cstring>

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

// detection.py:43-48 (+ :59 boxes*scale).  The same launch transposes the (B*P, C) score rows into
// class-major rows scoresT[b][cl][p] (through an LDS tile, both sides coalesced) so that the per-class
// workgroups of the NMS launch read 4*P contiguous bytes instead of a C-strided column.
__global__ __launch_bounds__(256) void detect_decode_kernel(const float *__restrict__ loc, const float *__restrict__ arm,
                                                            const float *__restrict__ priors, const float *__restrict__ conf,
                                                            int B, int P, int C, f32x4 scale, const float *__restrict__ scale_dev,
                                                            float *__restrict__ boxes,
                                                            float *__restrict__ sboxes, float *__restrict__ scoresT)
{
    if (scale_dev) scale = *(const f32x4 *)scale_dev;       // the caller's [w,h,w,h] lives on the device (evaluate.py:461)
    extern __shared__ __attribute__((aligned(16))) float tile[];       // 256 rows x (C|1) floats
    const long long i0 = (long long)blockIdx.x * 256, total = (long long)B * P;
    const long long i = i0 + threadIdx.x;
    const int Cs = C | 1;
    const long long rows = total - i0 < 256 ? total - i0 : 256;
    const float *src = conf + i0 * C;
    for (int k = threadIdx.x; k < (int)rows * C; k += 256) tile[(k / C) * Cs + k % C] = src[k];
    __syncthreads();
    if (i >= total) return;
    const int p = (int)(i % P), b = (int)(i / P);
    for (int cl = 0; cl < C; ++cl) scoresT[((size_t)b * C + cl) * P + p] = tile[threadIdx.x * Cs + cl];
    const f32x4 pr = *(const f32x4 *)(priors + (size_t)p * 4);
    float anchor[4] = {pr[0], pr[1], pr[2], pr[3]};
    if (arm) {
        const f32x4 a = *(const f32x4 *)(arm + (size_t)i * 4);
        float af[4] = {a[0], a[1], a[2], a[3]}, t[4];
        decode_one(af, anchor, 0.1f, 0.2f, t);
        center_size_one(t, anchor);
    }
    const f32x4 l = *(const f32x4 *)(loc + (size_t)i * 4);
    float lf[4] = {l[0], l[1], l[2], l[3]}, bx[4];
    decode_one(lf, anchor, 0.1f, 0.2f, bx);
    *(f32x4 *)(boxes + (size_t)i * 4) = f32x4{bx[0], bx[1], bx[2], bx[3]};
    *(f32x4 *)(sboxes + (size_t)i * 4) =
        f32x4{__fmul_rn(bx[0], scale[0]), __fmul_rn(bx[1], scale[1]), __fmul_rn(bx[2], scale[2]), __fmul_rn(bx[3], scale[3])};
}

// ---- shared device pieces -------------------------------------------------------------------
// Bitonic sort (descending) of N = 2^k >= 128 keys in LDS by 256 threads.  Compare-exchange i touches
// a = 2i - (i & (j-1)) and a + j: for j <= 64 the 64 pairs of one wavefront's iteration stay inside one
// 128-key span, so those stages need no workgroup barrier (LDS operations of one wave complete in order);
// only the j >= 128 stages and the hand-over between the two kinds synchronise the workgroup.  The loads
// of a stage are issued together (four pairs at a time) so that one LDS round trip covers them.
__device__ __forceinline__ void bitonic_stage(unsigned long long *k, int N, int kk, int j, int tid, int nthreads, int base = 0)
{
    for (int i0 = tid; i0 < (N >> 1); i0 += 4 * nthreads) {
        unsigned long long x[4], y[4];
        int a[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = i0 + r * nthreads;
            a[r] = 2 * i - (i & (j - 1));
            if (i < (N >> 1)) { x[r] = k[a[r]]; y[r] = k[a[r] + j]; }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = i0 + r * nthreads;
            if (i < (N >> 1)) {
                const bool desc = ((base + a[r]) & kk) == 0;      // base: where k[0] sits in a larger (global) sort
                if ((x[r] < y[r]) == desc) { k[a[r]] = y[r]; k[a[r] + j] = x[r]; }
            }
        }
    }
}
__device__ __forceinline__ void bitonic_sort_desc(unsigned long long *k, int N, int tid, int nthreads)
{
    const bool wave_local_ok = (nthreads & 63) == 0;
    for (int kk = 2; kk <= N; kk <<= 1) {
        for (int j = kk >> 1; j > 0; j >>= 1) {
            bitonic_stage(k, N, kk, j, tid, nthreads);
            if (j > 64 || j == 1 || !wave_local_ok) __syncthreads();
            else __builtin_amdgcn_wave_barrier();
        }
    }
}

struct Box { float x1, y1, x2, y2, area; };
// Keep list of up to `cap` boxes: corners as 16-byte vectors, areas apart (20 bytes a box, both parts
// naturally aligned so that a box is one 16-byte and one 4-byte load, broadcast when every lane reads the same i).
struct KeepList {
    f32x4 *c;
    float *area;
    __device__ __forceinline__ KeepList(void *base, int cap) : c((f32x4 *)base), area((float *)((f32x4 *)base + cap)) {}
    __device__ __forceinline__ Box get(int i) const { const f32x4 v = c[i]; return Box{v[0], v[1], v[2], v[3], area[i]}; }
    __device__ __forceinline__ void set(int i, const Box &b) const { c[i] = f32x4{b.x1, b.y1, b.x2, b.y2}; area[i] = b.area; }
};
__device__ __forceinline__ float box_area(float x1, float y1, float x2, float y2, int plain = 0)
{   // cpu_nms.pyx:24  (x2 - x1 + 1) * (y2 - y1 + 1);  plain (layers/box_utils.py:247): (x2 - x1) * (y2 - y1)
    if (plain) return __fmul_rn(__fsub_rn(x2, x1), __fsub_rn(y2, y1));
    return __fmul_rn(__fadd_rn(__fsub_rn(x2, x1), 1.f), __fadd_rn(__fsub_rn(y2, y1), 1.f));
}

// The suppression test of cpu_nms.pyx:55-66 is  ovr = inter / (area_i + area_j - inter)  in fp32, then
// ovr >= thresh (thresh a double; nms_kernel.cu:71 has ovr > thresh).  Both reduce to ovr >= b for one
// fp32 b (make_rule).  (A division-free form -- inter vs m*uni in double, m the rounding midpoint below b --
// is exact too but measured 15-20 % slower than v_div on gfx950: fp64 converts and compares are not full rate.)
// plain = 1: the torch NMS of layers/box_utils.py:229-293 (DetectOTA): no "+1", union = (area_j - inter) + area_i,
// a candidate survives iff IoU <= overlap in fp32 (so a NaN IoU -- two zero-area boxes -- suppresses, like idx[IoU.le()]).
struct NmsRule { float b; int plain; };
static NmsRule make_rule(double thresh, int strict_gt)
{
    float b = (float)thresh;
    if (strict_gt) {                      // ovr > thresh  <=>  ovr >= succ(largest fp32 <= thresh)
        if ((double)b > thresh) b = nextafterf(b, -INFINITY);
        b = nextafterf(b, INFINITY);
    } else if ((double)b < thresh) {      // (double)ovr >= thresh  <=>  ovr >= smallest fp32 >= thresh
        b = nextafterf(b, INFINITY);
    }
    return NmsRule{b, 0};
}
// i = the kept (higher-score) box, j = the candidate
__device__ __forceinline__ bool suppresses(const Box &i, const Box &j, const NmsRule &r)
{
    const float xx1 = fmaxf(i.x1, j.x1), yy1 = fmaxf(i.y1, j.y1);
    const float xx2 = fminf(i.x2, j.x2), yy2 = fminf(i.y2, j.y2);
    if (r.plain) {
        const float w = fmaxf(__fsub_rn(xx2, xx1), 0.f), h = fmaxf(__fsub_rn(yy2, yy1), 0.f);
        const float inter = __fmul_rn(w, h);
        return !(__fdiv_rn(inter, __fadd_rn(__fsub_rn(j.area, inter), i.area)) <= r.b);
    }
    const float w = fmaxf(0.f, __fadd_rn(__fsub_rn(xx2, xx1), 1.f));
    const float h = fmaxf(0.f, __fadd_rn(__fsub_rn(yy2, yy1), 1.f));
    const float inter = __fmul_rn(w, h);
    return __fdiv_rn(inter, __fsub_rn(__fadd_rn(i.area, j.area), inter)) >= r.b;
}

// Box of lane `src` (wave-uniform) broadcast to every lane through v_readlane (no LDS round trip).
__device__ __forceinline__ Box box_of_lane(const Box &me, int src)
{
    Box o;
    o.x1 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(me.x1), src));
    o.y1 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(me.y1), src));
    o.x2 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(me.x2), src));
    o.y2 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(me.y2), src));
    o.area = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(me.area), src));
    return o;
}
// Is the candidate suppressed by any of kept[lo, hi)?  Four boxes per trip so that the (broadcast) LDS
// reads of a trip overlap; the result equals the one-at-a-time scan's.
__device__ __forceinline__ bool suppressed_by(const KeepList &kept, int lo, int hi, const Box &me, const NmsRule &r)
{
    bool dead = false;
    int i = lo;
    for (; i + 4 <= hi && !dead; i += 4) {
        const Box k0 = kept.get(i), k1 = kept.get(i + 1), k2 = kept.get(i + 2), k3 = kept.get(i + 3);
        dead = (int)suppresses(k0, me, r) | (int)suppresses(k1, me, r) | (int)suppresses(k2, me, r) | (int)suppresses(k3, me, r);
    }
    for (; i < hi && !dead; ++i) dead = suppresses(kept.get(i), me, r);
    return dead;
}
// One wavefront settles 64 candidates (lane = candidate, descending score) of which `alive` survived the
// boxes kept so far: candidate j is kept iff no kept candidate i < j of the same 64 suppresses it.
// Returns the keep mask.  Only candidates that are still alive can suppress, so only those are broadcast.
__device__ __forceinline__ unsigned long long settle64(const Box &me, bool alive, const NmsRule &r)
{
    const int lane = threadIdx.x & 63;
    const unsigned long long amask = __ballot(alive);
    unsigned long long sup = 0ull;                    // bit i: candidate i (i < lane, alive) overlaps me
    for (unsigned long long m = amask; m; m &= m - 1ull) {
        const int i = __builtin_ctzll(m);
        const Box o = box_of_lane(me, i);
        if (i < lane && suppresses(o, me, r)) sup |= 1ull << i;
    }
    unsigned long long km = 0ull;
    for (unsigned long long m = amask; m; m &= m - 1ull) {
        const int i = __builtin_ctzll(m);
        const unsigned lo = __builtin_amdgcn_readlane((unsigned)sup, i);
        const unsigned hi = __builtin_amdgcn_readlane((unsigned)(sup >> 32), i);
        const unsigned long long s_i = ((unsigned long long)hi << 32) | lo;
        if ((s_i & km) == 0ull) km |= 1ull << i;
    }
    return km;
}

// One wavefront: greedy NMS over `n` candidates given in descending order.  get(pos, box) loads
// candidate pos.  kept (capacity cap) holds the boxes kept so far; emit(slot, pos) records a
// survivor.  Stops when cap survivors exist (cap = n reproduces the full cpu_nms list).
template <typename Get, typename Emit>
__device__ __forceinline__ int wave_greedy_nms(int n, int cap, const NmsRule &r, const KeepList &kept, Get get, Emit emit)
{
    const int lane = threadIdx.x & 63;
    int nk = 0;
    for (int c0 = 0; c0 < n && nk < cap; c0 += 64) {
        const int pos = c0 + lane;
        const bool valid = pos < n;
        Box me = {0.f, 0.f, 0.f, 0.f, 1.f};
        if (valid) get(pos, me);
        const bool alive = valid && !suppressed_by(kept, 0, nk, me, r);
        const unsigned long long km = settle64(me, alive, r);
        const bool keepme = (km >> lane) & 1ull;
        const int slot = nk + __popcll(km & ((1ull << lane) - 1ull));
        if (keepme && slot < cap) {
            kept.set(slot, me);
            emit(slot, pos);
        }
        nk += __popcll(km);
        __threadfence_block();
        __builtin_amdgcn_wave_barrier();
    }
    return nk < cap ? nk : cap;
}

// Workgroup version for Detect (256 threads), resumable: *nk_s boxes are already in `kept` on entry (the
// caller zeroes it once) and the n candidates given continue the descending order.  Candidates are taken
// 256 at a time; every thread first tests its candidate against the keep list as it stood at the start of
// the round (the long scan, spread over 4 waves), then against the earlier candidates of the same round (pairwise
// bits, again on all 4 waves, the group's boxes handed round by v_readlane), and the greedy decisions are then pure
// mask arithmetic, one 64-candidate group after the other.  Same decisions as the sequential algorithm.  (The bench
// regime consumes ~300 candidates per class for 200 survivors: two rounds; the pair tests are VALU-throughput-bound
// with 2.5 workgroups per CU.)  Returns min(*nk_s, cap).
template <typename Get, typename Emit>
__device__ __forceinline__ int block_greedy_nms(int n, int cap, const NmsRule &r, const KeepList &kept, unsigned char *alive_s,
                                                int *nk_s, Box *rb, unsigned *kmw, unsigned char *part_s, Get get, Emit emit)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __syncthreads();
    for (int c0 = 0; c0 < n;) {
        const int nk0 = *nk_s;
        if (nk0 >= cap) break;
        // groups of 64 candidates in this round: all four while many survivors are still wanted, fewer near the cap (1.5
        // candidates per wanted survivor), so that the last round does not pay 256^2/2 pair tests for a handful of boxes
        int G = ((cap - nk0) * 3 / 2 + 63) >> 6;
        G = G > 4 ? 4 : G;
        const int pos = c0 + tid;
        const bool valid = pos < n && wave < G;
        Box me = {0.f, 0.f, 0.f, 0.f, 1.f};
        bool alive;
        if (G >= 3) {
            if (valid) get(pos, me);
            alive = valid && !suppressed_by(kept, 0, nk0, me, r);           // the long scan, spread over the waves
        } else {
            // Near the cap a round takes one or two groups only: the idle waves take SLICES of the keep list for the same
            // candidates (G = 1: four waves x a quarter of the list each; G = 2: two halves), the owner ORs the verdicts.  (Round 4:
            // with ~170 boxes kept the one-group rounds spent ~11 us of a single wave in this scan.)  Same decisions.
            const int S = G == 1 ? 4 : 2;
            const int grp = wave % G, slice = wave / G;
            const int pos_s = c0 + grp * 64 + lane;
            Box cand = {0.f, 0.f, 0.f, 0.f, 1.f};
            const bool valid_s = pos_s < n;
            if (valid_s) get(pos_s, cand);
            const int lo = nk0 * slice / S, hi = nk0 * (slice + 1) / S;
            part_s[tid] = (valid_s && suppressed_by(kept, lo, hi, cand, r)) ? 1 : 0;
            __syncthreads();
            if (wave < G) me = cand;                                         // (slice 0 of group `wave`: my own candidate)
            bool dead = false;
            for (int q = 0; q < S; ++q) dead = dead || part_s[(wave + q * G) * 64 + lane] != 0;
            alive = valid && !dead;
        }
        alive_s[tid] = alive ? 1 : 0;
        rb[tid] = me;
        __syncthreads();
        // Pairwise bits inside the round, all 256 threads at once: bit i of sup[t] = candidate 64 t + i comes before me
        // in the order, survived the long scan and overlaps me.  (Only a candidate that is alive can ever suppress.)
        unsigned long long sup[4] = {0ull, 0ull, 0ull, 0ull};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (t > wave || wave >= G) break;              // (wave-uniform)
            // group t's boxes: one per lane, handed round through v_readlane (no LDS round trip per pair)
            const Box g = rb[t * 64 + lane];
            const unsigned long long gmask = __ballot(alive_s[t * 64 + lane] != 0);
            const int before = t < wave ? 64 : lane;                         // group members that precede me
            unsigned long long bits = 0ull;
            for (unsigned long long m = gmask; m; m &= m - 1ull) {
                const int i = __builtin_ctzll(m);
                const Box o = box_of_lane(g, i);
                if (alive && i < before && suppresses(o, me, r)) bits |= 1ull << i;
            }
            sup[t] = bits;
        }
        // The greedy decisions, one 64-candidate group after the other (group t on wave t): a candidate is kept iff none
        // of the candidates KEPT before it in this round overlaps it.  kmw[2t], kmw[2t+1] = keep mask of group t.
        for (int t = 0; t < G; ++t) {
            if (wave == t) {
                int nk = *nk_s;
                unsigned long long km = 0ull;
                if (nk < cap && c0 + t * 64 < n) {
                    bool al = alive;
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (u < t) al = al && (sup[u] & (((unsigned long long)kmw[2 * u + 1] << 32) | kmw[2 * u])) == 0ull;
                    const unsigned long long mine = sup[t];
                    for (unsigned long long m = __ballot(al); m; m &= m - 1ull) {
                        const int i = __builtin_ctzll(m);
                        const unsigned lo = __builtin_amdgcn_readlane((unsigned)mine, i);
                        const unsigned hi = __builtin_amdgcn_readlane((unsigned)(mine >> 32), i);
                        if (((((unsigned long long)hi << 32) | lo) & km) == 0ull) km |= 1ull << i;
                    }
                    const bool keepme = (km >> lane) & 1ull;
                    const int slot = nk + __popcll(km & ((1ull << lane) - 1ull));
                    if (keepme && slot < cap) {
                        kept.set(slot, me);
                        emit(slot, pos);
                    }
                    nk += __popcll(km);
                }
                if (lane == 0) {
                    kmw[2 * t] = (unsigned)km;
                    kmw[2 * t + 1] = (unsigned)(km >> 32);
                    *nk_s = nk;
                }
            }
            __syncthreads();
        }
        c0 += G * 64;
    }
    const int nk = *nk_s;
    return nk < cap ? nk : cap;
}

// ---- Detect: one workgroup per (image, class) segment -------------------------------------
// detection.py:52-63.  Scores are order-preserving 32-bit keys (float bits with the sign trick), 0 = "not a
// candidate"; a sort key is (score key : ~prior index), so equal scores order by ascending prior index.
__device__ __forceinline__ unsigned score_key(float s)
{
    const unsigned u = __float_as_uint(s);
    return u ^ ((unsigned)((int)u >> 31) | 0x80000000u);
}
__device__ __forceinline__ float key_score(unsigned k)
{
    return __uint_as_float(k ^ (((k >> 31) - 1u) | 0x80000000u));
}

// Optional phase stamps (builds with -DTDRN_DETECT_TIMING only; read back by scripts/detect_phases.py from
// the tail of the workspace): 100-MHz wall clock at scan/select/compact/sort/nms/pack boundaries.
#ifdef TDRN_DETECT_TIMING
#define DT_STAMP(buf, k)                                          \
    do {                                                          \
        __syncthreads();                                          \
        if (threadIdx.x == 0) (buf)[k] = (long long)wall_clock64(); \
    } while (0)
#else
#define DT_STAMP(buf, k) do { } while (0)
#endif

// Shared by both Detect kernels: sort the n keys in sk[] and continue the greedy NMS with them; survivors
// are packed as [score, x1, y1, x2, y2] rows (normalised boxes, detection.py:59-62) as they are kept.
__device__ __forceinline__ int sort_and_nms(unsigned long long *sk, int n, int top_k, const NmsRule &r, const KeepList &kept,
                                            unsigned char *alive_s, int *nk_s, Box *rb, unsigned *kmw, unsigned char *part_s,
                                            const float *__restrict__ sb, const float *__restrict__ nb, float *__restrict__ orow, long long *stamps)
{
    (void)stamps;
    int N = 64;
    while (N < n) N <<= 1;
    for (int i = n + threadIdx.x; i < N; i += 256) sk[i] = 0ull;
    __syncthreads();
    bitonic_sort_desc(sk, N, threadIdx.x, 256);
    DT_STAMP(stamps, 4);
    return block_greedy_nms(
        n, top_k, r, kept, alive_s, nk_s, rb, kmw, part_s,
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
            o[0] = key_score((unsigned)(key >> 32));
            o[1] = v[0]; o[2] = v[1]; o[3] = v[2]; o[4] = v[3];
        });
}

// LDS layout: sk[kcap] | kept (20 B x top_k) | ctl[16] | alive[256 B] | wsum[8] | hist[256] | round boxes (20 B x 256) | keep masks[8] | (LDS keys: sc[P])
__device__ __forceinline__ void detect_lds(unsigned long long *dsm, int kcap, int top_k, unsigned long long *&sk, void *&kept,
                                           int *&ctl, unsigned char *&alive_s, int *&wsum, unsigned *&extra)
{
    sk = dsm;
    kept = (void *)(dsm + kcap);
    ctl = (int *)((char *)kept + (size_t)top_k * sizeof(Box));
    alive_s = (unsigned char *)(ctl + 16);
    wsum = (int *)(alive_s + 256);
    extra = (unsigned *)(wsum + 8);
}
constexpr int kRoundWords = 256 * (int)sizeof(Box) / 4 + 8;      // the NMS round's boxes and keep masks, in 4-byte words
static size_t detect_lds_bytes(int kcap, int top_k) { return (size_t)kcap * 8 + (size_t)top_k * sizeof(Box) + 64 + 256 + 32 + 256 * 4 + kRoundWords * 4; }

// The workgroup turns its class row (score > conf_thresh, detection.py:53) into score keys.  Greedy NMS walks
// candidates in descending score (cpu_nms.pyx:31) and Detect stops at top_k survivors (detection.py:63), so the
// row is consumed in CHUNKS of at most kcap (1024) candidates: a radix select (8 bits a level, most significant
// first, stopping as soon as a level leaves kcap/2..kcap candidates above a bin boundary) finds a score
// threshold, the candidates above it and not yet consumed are compacted, sorted, and handed to the resumable
// NMS.  Every chunk is a contiguous piece of the descending order, so the decisions are the sequential
// algorithm's.  If kcap or more candidates share ONE score no threshold separates them; their order is
// ascending prior index, so that tie group is taken kcap at a time in index order ("tie mode").
// GK = false: keys live in LDS (4*P bytes).  GK = true (P too large for that, e.g. the 1216-pixel scale of
// multi_eval.py:24 with P = 92055): keys replace the scores of the class row in global memory, in place.
template <bool GK>
__global__ __launch_bounds__(256) void detect_select_nms_kernel(const float *__restrict__ boxes, const float *__restrict__ sboxes,
                                                                float *scoresT, int P, int C, int top_k,
                                                                float conf_thresh, NmsRule rule, int kcap,
                                                                float *__restrict__ out,
                                                                int *__restrict__ counts_out, long long *__restrict__ dbg)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long dsm[];
    long long *stamps = dbg ? dbg + (size_t)blockIdx.x * 8 : nullptr;
    (void)stamps;
    DT_STAMP(stamps, 0);
    unsigned long long *sk; void *kept_mem; int *ctl; unsigned char *alive_s; int *wsum; unsigned *hist;
    detect_lds(dsm, kcap, top_k, sk, kept_mem, ctl, alive_s, wsum, hist);
    const KeepList kept(kept_mem, top_k);
    const int seg = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;          // seg = b*C + cl
    const int cl = seg % C, b = seg / C;
    float *row = scoresT + (size_t)seg * P;
    Box *rb = (Box *)(hist + 256);
    unsigned *kmw = hist + 256 + kRoundWords - 8;
    unsigned *sc = GK ? (unsigned *)row : hist + 256 + kRoundWords;
    float *orow = out + (size_t)seg * top_k * 5;
    if (cl == 0) {                                                         // background: detection.py:51 skips it
        for (int i = tid; i < top_k * 5; i += 256) orow[i] = 0.f;
        if (counts_out && tid == 0) counts_out[seg] = 0;
        return;
    }
    if (tid < 16) ctl[tid] = tid == 3 ? (int)0xFFFFFFFF : 0;              // [0] n, [1] nk, [2] OR, [3] AND, [4] bin, [5] remaining, [6] next tie_lo
    __syncthreads();
    // ---- scan: class row -> score keys (0 = below conf_thresh); loads issued five at a time --------
    int mycnt = 0;
    unsigned vor = 0u, vand = 0xFFFFFFFFu;
    for (int p0 = tid; p0 < P; p0 += 5 * 256) {
        float sv[5];
#pragma unroll
        for (int u = 0; u < 5; ++u) sv[u] = p0 + u * 256 < P ? row[p0 + u * 256] : 0.f;
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            const int p = p0 + u * 256;
            if (p < P) {
                const unsigned v = sv[u] > conf_thresh ? score_key(sv[u]) : 0u;
                sc[p] = v;
                if (v) { ++mycnt; vor |= v; vand &= v; }
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        mycnt += __shfl_xor(mycnt, o, 64);
        vor |= __shfl_xor(vor, o, 64);
        vand &= __shfl_xor(vand, o, 64);
    }
    if (lane == 0) { atomicAdd(&ctl[0], mycnt); atomicOr((unsigned *)&ctl[2], vor); atomicAnd((unsigned *)&ctl[3], vand); }
    __syncthreads();
    DT_STAMP(stamps, 1);
    const int n = ctl[0];
    if (n == 0) {
        for (int i = tid; i < top_k * 5; i += 256) orow[i] = 0.f;
        if (counts_out && tid == 0) counts_out[seg] = 0;
        return;
    }
    // bytes above the highest bit in which two candidates differ are common to all: the select starts below them
    const unsigned diff = (unsigned)ctl[2] ^ (unsigned)ctl[3];
    const int shift0 = diff ? ((31 - __clz(diff)) & ~7) : 0;
    const unsigned mask0 = shift0 < 24 ? 0xFFFFFFFFu << (shift0 + 8) : 0u, prefix0 = (unsigned)ctl[3] & mask0;
    // compaction ranges.  LDS keys: L consecutive scores per thread (L odd: conflict-free).  Global keys: a
    // contiguous quarter of the row per wave, read 64 keys (256 B) at a time.
    const int L = ((P + 255) / 256) | 1;
    const int p_lo = tid * L < P ? tid * L : P, p_hi = p_lo + L < P ? p_lo + L : P;
    const int Q = (((P + 3) / 4) + 63) & ~63;
    const int q_lo = wave * Q < P ? wave * Q : P, q_hi = q_lo + Q < P ? q_lo + Q : P;
    const float *sb = sboxes + (size_t)b * P * 4, *nb = boxes + (size_t)b * P * 4;
    unsigned hi = 0xFFFFFFFFu;                           // candidates not yet consumed: 0 < key <= hi
    bool tie = false;                                    // tie mode: the candidates with key == tie_key and prior >= tie_lo
    unsigned tie_key = 0u;
    int tie_lo = 0;
    int taken = 0, nk = 0;
    for (;;) {
        // ---- threshold: this chunk = unconsumed candidates with (key & mask) > prefix (mask = 0: all of them) ----
        unsigned prefix = 0u, mask = 0u;
        if (!tie && n - taken > kcap) {
            prefix = prefix0;
            mask = mask0;
            int remaining = kcap;
            for (int shift = shift0; shift >= 0; shift -= 8) {
                hist[tid] = 0u;
                __syncthreads();
                for (int p = tid; p < P; p += 256) {
                    const unsigned v = sc[p];
                    if (v && v <= hi && (v & mask) == prefix) atomicAdd(&hist[(v >> shift) & 255u], 1u);
                }
                __syncthreads();
                unsigned above = 0u;
                for (int j = tid + 1; j < 256; ++j) above += hist[j];
                if ((int)above < remaining && (int)(above + hist[tid]) >= remaining) { ctl[4] = tid; ctl[5] = remaining - (int)above; }
                __syncthreads();
                prefix |= (unsigned)ctl[4] << shift;
                mask |= 0xFFu << shift;
                remaining = ctl[5];
                __syncthreads();
                if (kcap - remaining >= kcap / 2) break;
            }
            if (remaining == kcap) {                     // nothing lies strictly above a score that >= kcap candidates share
                tie = true;
                tie_key = prefix;
                tie_lo = 0;
            }
        }
        DT_STAMP(stamps, 2);
        auto selected = [&](unsigned v, int p) -> bool {
            if (tie) return v == tie_key && p >= tie_lo;
            return v && v <= hi && (mask ? (v & mask) > prefix : true);
        };
        // ---- compaction in ascending prior order: per-thread counts, workgroup scan, scatter of the first kcap ----
        int c = 0;
        if constexpr (GK) {
            for (int p0 = q_lo; p0 < q_hi; p0 += 256) {
                unsigned v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int p = p0 + u * 64 + lane; v[u] = p < q_hi ? sc[p] : 0u; }
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int p = p0 + u * 64 + lane; c += (p < q_hi && selected(v[u], p)) ? 1 : 0; }
            }
            for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);     // every lane: this wave's count
            if (lane == 0) wsum[wave] = c;
        } else {
            for (int p = p_lo; p < p_hi; ++p) c += selected(sc[p], p) ? 1 : 0;
            int incl = c;
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(incl, o, 64);
                if (lane >= o) incl += t;
            }
            if (lane == 63) wsum[wave] = incl;
            c = incl - c;                                                  // exclusive prefix inside the wave
        }
        __syncthreads();
        int idx = GK ? 0 : c;
        for (int w = 0; w < wave; ++w) idx += wsum[w];
        const int total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        const int nsel = total < kcap ? total : kcap;                       // total > kcap only in tie mode
        auto place = [&](unsigned v, int p, int at) {
            if (at < kcap) sk[at] = ((unsigned long long)v << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)p);
            else if (at == kcap) ctl[6] = p;                                // where the next tie chunk starts
        };
        if constexpr (GK) {
            for (int p0 = q_lo; p0 < q_hi; p0 += 64) {
                const int p = p0 + lane;
                const unsigned v = p < q_hi ? sc[p] : 0u;
                const bool s = p < q_hi && selected(v, p);
                const unsigned long long m = __ballot(s);
                if (s) place(v, p, idx + __popcll(m & ((1ull << lane) - 1ull)));
                idx += __popcll(m);
            }
        } else {
            for (int p = p_lo; p < p_hi; ++p) {
                const unsigned v = sc[p];
                if (selected(v, p)) place(v, p, idx++);
            }
        }
        __syncthreads();
        DT_STAMP(stamps, 3);
        nk = sort_and_nms(sk, nsel, top_k, rule, kept, alive_s, ctl + 1, rb, kmw, (unsigned char *)hist, sb, nb, orow, stamps);      // (the histogram is idle during the NMS)
        DT_STAMP(stamps, 5);
        taken += nsel;
        if (nk >= top_k || taken >= n) break;
        if (tie) {
            if (total > kcap) {
                tie_lo = ctl[6];
            } else {                                                       // the tie group is used up
                tie = false;
                hi = tie_key - 1u;
            }
        } else {
            hi = prefix | ~mask;                                           // what is left lies at or below this chunk's bin
        }
        __syncthreads();
    }
    for (int i = nk * 5 + tid; i < top_k * 5; i += 256) orow[i] = 0.f;      // detection.py:39 zero-initialised output
    if (counts_out && tid == 0) counts_out[seg] = nk;
    DT_STAMP(stamps, 6);
}

static int next_pow2(int v) { int n = 64; while (n < v) n <<= 1; return n; }

size_t detect_workspace_bytes(int B, int P, int C, int top_k)
{
    (void)top_k;   // boxes, boxes*scale, class-major scores (overwritten by the score keys when P is large)
    size_t n = align_up((size_t)B * P * 4 * sizeof(float), 256) * 2 + align_up((size_t)B * C * P * sizeof(float), 256);
#ifdef TDRN_DETECT_TIMING
    n += (size_t)B * C * 8 * sizeof(long long);
#endif
    return n;
}

int launch_detect(const float *loc, const float *conf, const float *priors, const float *arm_loc, const float *scale4,
                  int scale_on_device, int B, int P, int C, int top_k, float conf_thresh, double nms_thresh, float *out,
                  int32_t *counts_out, void *ws, size_t ws_bytes, hipStream_t s)
{
    if (!loc || !conf || !priors || !scale4 || !out || !ws) return TDRN_E_ARG;
    if (scale_on_device && ((uintptr_t)scale4 & 15)) return TDRN_E_ARG;
    if (nms_thresh <= 0) return TDRN_E_VALUE;
    if (B <= 0 || P <= 0 || C < 2 || top_k <= 0) return TDRN_E_ARG;
    if (ws_bytes < detect_workspace_bytes(B, P, C, top_k)) return TDRN_E_WORKSPACE;
#ifndef TDRN_DET_KCAP
#define TDRN_DET_KCAP 1024      // candidates per chunk: 1024 measured 6 % faster than 2048 (half the sort), 512 slower (second chunks)
#endif
    const int kcap_big = next_pow2(P);
    const int kcap = kcap_big < TDRN_DET_KCAP ? kcap_big : TDRN_DET_KCAP;
    const size_t lds_fixed = detect_lds_bytes(kcap, top_k);
    const size_t lds_dec = (size_t)256 * (C | 1) * sizeof(float);
    constexpr size_t kLdsMax = 160 * 1024;
    if (lds_fixed > kLdsMax || lds_dec > kLdsMax) return TDRN_E_UNSUPPORTED;       // top_k / class count beyond any caller's
    const bool global_keys = lds_fixed + (size_t)P * 4 > kLdsMax;                  // P > ~37000
    const size_t lds_sel = global_keys ? lds_fixed : lds_fixed + (size_t)P * 4;
    char *w = (char *)ws;
    float *boxes = (float *)w;   w += align_up((size_t)B * P * 4 * sizeof(float), 256);
    float *sboxes = (float *)w;  w += align_up((size_t)B * P * 4 * sizeof(float), 256);
    float *scoresT = (float *)w; w += align_up((size_t)B * C * P * sizeof(float), 256);
    long long *dbg = nullptr;
#ifdef TDRN_DETECT_TIMING
    dbg = (long long *)w;
#endif
    const NmsRule rule = make_rule(nms_thresh, 0);
    if (global_keys) TDRN_TRY(allow_big_lds((const void *)detect_select_nms_kernel<true>));
    else TDRN_TRY(allow_big_lds((const void *)detect_select_nms_kernel<false>));
    TDRN_TRY(allow_big_lds((const void *)detect_decode_kernel));
    const long long bp = (long long)B * P;
    hipLaunchKernelGGL(detect_decode_kernel, dim3((unsigned)((bp + 255) / 256)), dim3(256), lds_dec, s, loc, arm_loc, priors, conf,
                       B, P, C, scale_on_device ? f32x4{0.f, 0.f, 0.f, 0.f} : f32x4{scale4[0], scale4[1], scale4[2], scale4[3]},
                       scale_on_device ? scale4 : (const float *)nullptr, boxes, sboxes, scoresT);
    if (global_keys)
        hipLaunchKernelGGL(detect_select_nms_kernel<true>, dim3((unsigned)(B * C)), dim3(256), lds_sel, s, boxes, sboxes, scoresT, P, C,
                           top_k, conf_thresh, rule, kcap, out, counts_out, dbg);
    else
        hipLaunchKernelGGL(detect_select_nms_kernel<false>, dim3((unsigned)(B * C)), dim3(256), lds_sel, s, boxes, sboxes, scoresT, P, C,
                           top_k, conf_thresh, rule, kcap, out, counts_out, dbg);
    return hip_status(hipGetLastError());
}

// ---- stand-alone NMS (cpu_nms / gpu_nms twins): one workgroup, keep list in global memory ------
// sort key of box i: (score key | presorted: descending rank) : ~index
// where a launch finds its boxes and scores: (n,5) rows [x1,y1,x2,y2,score] (box = dets, stride 5; score = dets + 4, stride 5), or
// a shared (n,4) box table with one score COLUMN per workgroup (DetectOTA: box stride 4; score = conf + class, stride C)
struct NmsSrc {
    const float *box; int bstride;
    const float *score; int sstride;
    int score_wg_step;             // floats added to `score` per blockIdx.x (0: one problem)
    long long kept_wg_bytes;       // bytes added to the keep-list scratch / keep_out (x n ints) / num_out (x 1) per blockIdx.x
};
__device__ __forceinline__ unsigned long long nms_key(const float *__restrict__ score, int sstride, int n, int presorted, int i, float min_score = -INFINITY)
{
    if (i >= n) return 0ull;
    if (!presorted && !(score[(size_t)i * sstride] > min_score)) return 0ull;       // not a candidate: sorts behind every candidate
    const unsigned hi = presorted ? (unsigned)(n - i) : score_key(score[(size_t)i * sstride]);
    return ((unsigned long long)hi << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)i);
}

// GK = false: the n <= 16384 keys are built and sorted in LDS.  GK = true: `gkeys` holds them, sorted, in global memory.
template <bool GK>
__global__ __launch_bounds__(256) void nms_plain_kernel(NmsSrc src, int n, NmsRule rule, int presorted,
                                                        const unsigned long long *__restrict__ gkeys,
                                                        void *__restrict__ kept_mem, int *__restrict__ keep_out,
                                                        int *__restrict__ num_out, float min_score, int pre_top_k)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long dsm[];     // (no static LDS beside it: cdna guide G17)
    // one workgroup = one problem; blockIdx.x > 0: the next score column over the same boxes (all classes of a frame in ONE launch)
    const float *score = src.score + (size_t)blockIdx.x * src.score_wg_step;
    kept_mem = (char *)kept_mem + (size_t)blockIdx.x * src.kept_wg_bytes;
    keep_out += (size_t)blockIdx.x * n;
    num_out += blockIdx.x;
    const unsigned long long *sk = GK ? gkeys : dsm;
    if constexpr (!GK) {
        int N = 64;
        while (N < n) N <<= 1;
        for (int i = threadIdx.x; i < N; i += 256) dsm[i] = nms_key(score, src.sstride, n, presorted, i, min_score);
        __syncthreads();
        if (!presorted) bitonic_sort_desc(dsm, N, threadIdx.x, 256);
    }
    __syncthreads();
    if (threadIdx.x >= 64) return;
    int n_cand = 0;     // candidates = the non-zero keys (they sort first); at most pre_top_k of them enter the NMS (box_utils.py:251)
    for (int i = threadIdx.x; i < n; i += 64) n_cand += sk[i] != 0ull ? 1 : 0;
    for (int o = 32; o > 0; o >>= 1) n_cand += __shfl_xor(n_cand, o, 64);
    const int nn = pre_top_k > 0 && n_cand > pre_top_k ? pre_top_k : n_cand;
    const KeepList kept(kept_mem, nn > 0 ? nn : 1);
    const int nk = wave_greedy_nms(
        nn, nn, rule, kept,
        [&](int pos, Box &bx) {
            const unsigned p = 0xFFFFFFFFu - (unsigned)(sk[pos] & 0xFFFFFFFFull);
            const float *d = src.box + (size_t)p * src.bstride;
            bx.x1 = d[0]; bx.y1 = d[1]; bx.x2 = d[2]; bx.y2 = d[3];
            bx.area = box_area(d[0], d[1], d[2], d[3], rule.plain);
        },
        [&](int slot, int pos) { keep_out[slot] = (int)(0xFFFFFFFFu - (unsigned)(sk[pos] & 0xFFFFFFFFull)); });
    if (threadIdx.x == 0) *num_out = nk;
}

// n > 16384 (the P = 92055 priors of multi_eval.py's 1216-pixel scale): the keys no longer fit LDS, so the bitonic
// network runs over global memory: tiles of 4096 keys are sorted in LDS (directions by GLOBAL index, so that the
// tiles come out as bitonic pairs), then per merge size kk the strides >= 4096 are one launch each and the strides
// below that are finished tile by tile in LDS again.
constexpr int kSortTile = 4096;
__global__ __launch_bounds__(256) void nms_tile_sort_kernel(const float *__restrict__ dets, int n, int presorted, int do_sort,
                                                            unsigned long long *__restrict__ keys, float min_score)
{
    __shared__ unsigned long long t[kSortTile];
    const int base = blockIdx.x * kSortTile;
    for (int i = threadIdx.x; i < kSortTile; i += 256) t[i] = nms_key(dets + 4, 5, n, presorted, base + i, min_score);
    __syncthreads();
    if (do_sort)
        for (int kk = 2; kk <= kSortTile; kk <<= 1)
            for (int j = kk >> 1; j > 0; j >>= 1) {
                bitonic_stage(t, kSortTile, kk, j, threadIdx.x, 256, base);
                __syncthreads();
            }
    for (int i = threadIdx.x; i < kSortTile; i += 256) keys[base + i] = t[i];
}
__global__ __launch_bounds__(256) void nms_global_stage_kernel(unsigned long long *__restrict__ keys, int N, int kk, int j)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= (N >> 1)) return;
    const int a = 2 * i - (i & (j - 1));
    const unsigned long long x = keys[a], y = keys[a + j];
    const bool desc = (a & kk) == 0;
    if ((x < y) == desc) { keys[a] = y; keys[a + j] = x; }
}
__global__ __launch_bounds__(256) void nms_tile_merge_kernel(unsigned long long *__restrict__ keys, int kk)
{
    __shared__ unsigned long long t[kSortTile];
    const int base = blockIdx.x * kSortTile;
    for (int i = threadIdx.x; i < kSortTile; i += 256) t[i] = keys[base + i];
    __syncthreads();
    for (int j = kSortTile >> 1; j > 0; j >>= 1) {
        bitonic_stage(t, kSortTile, kk, j, threadIdx.x, 256, base);
        __syncthreads();
    }
    for (int i = threadIdx.x; i < kSortTile; i += 256) keys[base + i] = t[i];
}

constexpr int kNmsLdsKeys = 16384;      // 128 KiB of LDS sort keys
static size_t nms_kept_bytes(int n) { return align_up((size_t)(n > 0 ? n : 1) * sizeof(Box), 256); }
size_t nms_workspace_bytes(int n)
{
    size_t b = nms_kept_bytes(n);
    if (n > kNmsLdsKeys) b += (size_t)next_pow2(n) * sizeof(unsigned long long);
    return b;
}

// plain_rule = 1: the torch NMS of layers/box_utils.py:229-293 (fp32 `thresh`, IoU <= thresh survives, no "+1"); min_score:
// boxes with score <= min_score are no candidates; pre_top_k > 0: only the pre_top_k best candidates enter (box_utils.py:251)
int launch_nms(const float *dets, int n, double thresh, int strict_gt, int presorted, int32_t *keep_out, int32_t *num_out,
               void *ws, size_t ws_bytes, hipStream_t s, int plain_rule, float min_score, int pre_top_k)
{
    if (!num_out || n < 0) return TDRN_E_ARG;
    if (n == 0) return hip_status(hipMemsetAsync(num_out, 0, sizeof(int), s));
    if (!dets || !keep_out || !ws) return TDRN_E_ARG;
    if (ws_bytes < nms_workspace_bytes(n)) return TDRN_E_WORKSPACE;
    const int N = next_pow2(n);
    const NmsRule rule = plain_rule ? NmsRule{(float)thresh, 1} : make_rule(thresh, strict_gt);
    const NmsSrc src{dets, 5, dets + 4, 5, 0, 0};
    if (n <= kNmsLdsKeys) {
        TDRN_TRY(allow_big_lds((const void *)nms_plain_kernel<false>));
        hipLaunchKernelGGL(nms_plain_kernel<false>, dim3(1), dim3(256), (size_t)N * 8, s, src, n, rule, presorted,
                           (const unsigned long long *)nullptr, ws, keep_out, num_out, min_score, pre_top_k);
        return hip_status(hipGetLastError());
    }
    unsigned long long *keys = (unsigned long long *)((char *)ws + nms_kept_bytes(n));
    const int tiles = N / kSortTile;
    hipLaunchKernelGGL(nms_tile_sort_kernel, dim3(tiles), dim3(256), 0, s, dets, n, presorted, presorted ? 0 : 1, keys, min_score);
    if (!presorted)
        for (int kk = 2 * kSortTile; kk <= N; kk <<= 1) {
            for (int j = kk >> 1; j >= kSortTile; j >>= 1)
                hipLaunchKernelGGL(nms_global_stage_kernel, dim3(N / 512), dim3(256), 0, s, keys, N, kk, j);
            hipLaunchKernelGGL(nms_tile_merge_kernel, dim3(tiles), dim3(256), 0, s, keys, kk);
        }
    hipLaunchKernelGGL(nms_plain_kernel<true>, dim3(1), dim3(256), 0, s, src, n, rule, presorted, (const unsigned long long *)keys, ws,
                       keep_out, num_out, min_score, pre_top_k);
    return hip_status(hipGetLastError());
}

// All classes of one frame in ONE launch (DetectOTA, layers/functions/detection_ota.py:61-79): boxes (n,4) shared, scores (n,C)
// row-major, one workgroup per class c in [first_class, C): box_utils.nms's rule on (boxes, scores[:, c]).
// keep_out (C, n) / num_out (C): rows of the classes below first_class are left untouched.  ws: nms_classes_workspace_bytes.
size_t nms_classes_workspace_bytes(int n, int C) { return (size_t)(C > 0 ? C : 1) * align_up(nms_workspace_bytes(n), 256); }
int launch_nms_classes(const float *boxes, const float *scores, int n, int C, int first_class, float overlap, float min_score, int top_k,
                       int32_t *keep_out, int32_t *num_out, void *ws, size_t ws_bytes, hipStream_t s)
{
    if (!boxes || !scores || !keep_out || !num_out || !ws || n < 1 || C < 1 || first_class < 0 || first_class >= C) return TDRN_E_ARG;
    if (ws_bytes < nms_classes_workspace_bytes(n, C)) return TDRN_E_WORKSPACE;
    const size_t per = align_up(nms_workspace_bytes(n), 256);
    if (n > kNmsLdsKeys) {              // beyond the LDS key limit: class by class through the global sort (strided rows are packed first)
        return TDRN_E_UNSUPPORTED;
    }
    const NmsRule rule{overlap, 1};
    const NmsSrc src{boxes, 4, scores + first_class, C, 1, (long long)per};
    TDRN_TRY(allow_big_lds((const void *)nms_plain_kernel<false>));
    hipLaunchKernelGGL(nms_plain_kernel<false>, dim3((unsigned)(C - first_class)), dim3(256), (size_t)next_pow2(n) * 8, s, src, n, rule, 0,
                       (const unsigned long long *)nullptr, ws, keep_out + (size_t)first_class * n, num_out + first_class, min_score, top_k);
    return hip_status(hipGetLastError());
}

// ---- DetectOTA's association arithmetic (layers/functions/detection_ota.py:86-99, layers/box_utils.py:295-367) ------------
// ROI feature: the box's cell range [x0,x1) x [y0,y1) of the (C,H,W) feature map resampled to S x S with
// F.upsample(mode='bilinear', align_corners=True)'s arithmetic (source index = dst * (in - 1) / (out - 1) in fp32, the
// upper neighbour clamped to the crop, weights 1 - l and l), flattened as (C, S, S).  One thread per output element.
__global__ __launch_bounds__(256) void roi_resample_kernel(const float *__restrict__ feat, int C, int H, int W, const int *__restrict__ cells,
                                                           int n, int S, float *__restrict__ out)
{
    const int per = C * S * S;
    const long long total = (long long)n * per;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int b = (int)(i / per), r = (int)(i - (long long)b * per);
        const int c = r / (S * S), q = r - c * S * S, oy = q / S, ox = q - oy * S;
        const int x0 = cells[4 * b], y0 = cells[4 * b + 1], x1 = cells[4 * b + 2], y1 = cells[4 * b + 3];
        const int ih = y1 - y0, iw = x1 - x0;
        float v = 0.f;
        if (ih > 0 && iw > 0) {
            const float sh = S > 1 ? (float)(ih - 1) / (float)(S - 1) : 0.f, sw = S > 1 ? (float)(iw - 1) / (float)(S - 1) : 0.f;
            const float hr = sh * oy, wr = sw * ox;
            const int h1 = (int)hr, w1 = (int)wr;
            const int hp = h1 < ih - 1 ? 1 : 0, wp = w1 < iw - 1 ? 1 : 0;
            const float hl1 = hr - h1, hl0 = 1.f - hl1, wl1 = wr - w1, wl0 = 1.f - wl1;
            const float *f = feat + ((size_t)c * H + (y0 + h1)) * W + (x0 + w1);
            v = hl0 * (wl0 * f[0] + wl1 * f[wp]) + hl1 * (wl0 * f[(size_t)hp * W] + wl1 * f[(size_t)hp * W + wp]);
        }
        out[i] = v;
    }
}

int launch_roi_resample(const float *feat, int C, int H, int W, const int32_t *cells, int n, int S, float *out, hipStream_t s)
{
    if (!feat || !cells || !out || C < 1 || H < 1 || W < 1 || S < 1) return TDRN_E_ARG;
    if (n <= 0) return TDRN_OK;
    const long long total = (long long)n * C * S * S;
    dim3 grid((unsigned)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256));
    hipLaunchKernelGGL(roi_resample_kernel, grid, dim3(256), 0, s, feat, C, H, W, cells, n, S, out);
    return hip_status(hipGetLastError());
}

// similarity of detection i to tubelet j = exp(IoU(box_i, head_j)) * mean over the tubelet's stored rows r of cos(roi_i, row_r)
// (box_utils.IoU: normalised boxes, no "+1"; box_utils.cos_similarity), then the row maximum and its (first) argument
// (detection_ota.py:99).  rows (R, 5 + F): [score, box, feature] of every tubelet back to back, newest row (the head) first;
// row_off (m + 1).  One workgroup per detection.
__global__ __launch_bounds__(256) void ota_similarity_kernel(const float *__restrict__ boxes, const float *__restrict__ roi, int F,
                                                             const float *__restrict__ rows, const int *__restrict__ row_off, int m,
                                                             float *__restrict__ best, int *__restrict__ arg)
{
    __shared__ float red[3][4];
    const int i = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const float *ri = roi + (size_t)i * F;
    auto block_sum = [&](float a, float b, float &sa, float &sb) {
        for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
        __syncthreads();
        if (lane == 0) { red[0][wave] = a; red[1][wave] = b; }
        __syncthreads();
        sa = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        sb = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    };
    float rn2 = 0.f, dummy = 0.f;
    for (int k = t; k < F; k += 256) rn2 = fmaf(ri[k], ri[k], rn2);
    block_sum(rn2, 0.f, rn2, dummy);
    const float rn = sqrtf(rn2);
    const float bx0 = boxes[4 * i], by0 = boxes[4 * i + 1], bx1 = boxes[4 * i + 2], by1 = boxes[4 * i + 3];
    const float area = (bx1 - bx0) * (by1 - by0);
    float bestv = -INFINITY;
    int besti = 0;
    for (int j = 0; j < m; ++j) {
        const int r0 = row_off[j], r1 = row_off[j + 1];
        float cs = 0.f;
        for (int r = r0; r < r1; ++r) {
            const float *tf = rows + (size_t)r * (5 + F) + 5;
            float d = 0.f, n2 = 0.f;
            for (int k = t; k < F; k += 256) { const float v = tf[k]; d = fmaf(ri[k], v, d); n2 = fmaf(v, v, n2); }
            block_sum(d, n2, d, n2);
            cs += d / (rn * sqrtf(n2));
        }
        cs /= (float)(r1 - r0);
        const float *h = rows + (size_t)r0 * (5 + F);              // head row: [score, x1, y1, x2, y2]
        const float ix0 = fmaxf(bx0, h[1]), iy0 = fmaxf(by0, h[2]), ix1 = fminf(bx1, h[3]), iy1 = fminf(by1, h[4]);
        const float inter = fmaxf(ix1 - ix0, 0.f) * fmaxf(iy1 - iy0, 0.f);
        const float iou = inter / ((area - inter) + (h[3] - h[1]) * (h[4] - h[2]));
        const float sim = expf(iou) * cs;
        if (sim > bestv) { bestv = sim; besti = j; }
    }
    if (t == 0) { best[i] = bestv; arg[i] = besti; }
}

int launch_ota_similarity(const float *boxes, const float *roi, int n, int F, const float *rows, const int32_t *row_off, int m, float *best,
                          int32_t *arg, hipStream_t s)
{
    if (!boxes || !roi || !rows || !row_off || !best || !arg || F < 1 || m < 1) return TDRN_E_ARG;
    if (n <= 0) return TDRN_OK;
    hipLaunchKernelGGL(ota_similarity_kernel, dim3((unsigned)n), dim3(256), 0, s, boxes, roi, F, rows, row_off, m, best, arg);
    return hip_status(hipGetLastError());
}

}  // namespace tdrn
