// dev/conv_check.hip -- developer harness (not part of libtdrn_hip.so, never shipped): runs one 3x3/s1/p1 layer through the
// loader/consumer kernel (conv3x3_patch.hip) and through the all-waves-compute kernel (conv3x3_pp.hip) on the same random
// operands, requires the outputs to be BIT-IDENTICAL (same K order per output element), and times both with hipEvents.
//   make -C tdrn_amd/csrc dev      ->  tdrn_amd/csrc/_build/conv_check
//   conv_check [B H W Cin Cout pool relu dtype(1=bf16,2=f16) iters]   (no arguments: the layer list of the 320 / 512 nets)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../kernels.h"

using namespace tdrn;

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(2);                                                                       \
        }                                                                                  \
    } while (0)

static unsigned lcg(unsigned &s) { s = s * 1664525u + 1013904223u; return s; }
static float frand(unsigned &s) { return ((lcg(s) >> 8) & 0xffff) / 32768.0f - 1.0f; }

static int run_case(int B, int H, int W, int Cin, int Cout, int pool, int relu, int dtype, int iters)
{
    const int Npad = conv_n_pad(Cout), es = 2;
    const size_t n_in = (size_t)B * H * W * Cin, n_w = (size_t)Npad * 9 * Cin, n_out = (size_t)B * H * W * Npad;
    const size_t n_pool = n_out / 4;
    std::vector<unsigned short> hin(n_in), hw(n_w);
    std::vector<float> hb(Npad);
    unsigned seed = 12345u + (unsigned)(H * 131 + Cin * 7 + Cout);
    for (auto &v : hin) v = dtype == TDRN_BF16 ? host_f32_to_bf16(frand(seed)) : host_f32_to_f16(frand(seed));
    const float ws = 1.0f / 48.0f;
    for (size_t i = 0; i < n_w; ++i) {
        const float f = (i / ((size_t)9 * Cin)) < (size_t)Cout ? frand(seed) * ws : 0.f;
        hw[i] = dtype == TDRN_BF16 ? host_f32_to_bf16(f) : host_f32_to_f16(f);
    }
    for (int i = 0; i < Npad; ++i) hb[i] = i < Cout ? frand(seed) * 0.5f : 0.f;
    char *din, *dw, *dzero, *dout[3], *dpool[3];
    float *db;
    CK(hipMalloc((void **)&din, n_in * es));
    CK(hipMalloc((void **)&dw, n_w * es));
    CK(hipMalloc((void **)&dzero, 256));
    CK(hipMalloc((void **)&db, Npad * 4));
    for (int k = 0; k < 3; ++k) {
        CK(hipMalloc((void **)&dout[k], n_out * es));
        CK(hipMalloc((void **)&dpool[k], n_pool * es));
        CK(hipMemset(dout[k], 0xAB, n_out * es));
        CK(hipMemset(dpool[k], 0xCD, n_pool * es));
    }
    CK(hipMemset(dzero, 0, 256));
    CK(hipMemcpy(din, hin.data(), n_in * es, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw, hw.data(), n_w * es, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), Npad * 4, hipMemcpyHostToDevice));
    ConvArgs a;
    a.in = din; a.w = dw; a.bias = db; a.zero_page = dzero;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Ho = H; a.Wo = W; a.Cout = Cout; a.Npad = Npad;
    a.kh = a.kw = 3; a.stride = 1; a.pad = 1; a.dil = 1; a.relu = relu; a.dtype = dtype;
    a.o_cs = Npad; a.o_rs = (long long)W * Npad; a.o_bs = (long long)H * W * Npad;
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    double us[3] = {0, 0, 0};
    int rcs[3] = {0, 0, 0};
    void *sk_ws = nullptr;
    CK(hipMalloc(&sk_ws, conv_pp_sk_bytes()));
    CK(hipMemset(sk_ws, 0xFF, conv_pp_sk_bytes()));      // (poisoned: the launcher must zero its flags itself)
    for (int k = 0; k < 3; ++k) {                        // 0: conv3x3_patch, 1: conv3x3_pp whole items, 2: conv3x3_pp chained split
        conv_pp_force(k ? 1 : 0);
        conv_pp_sk_force(k == 2 ? 1 : 0);
        a.sk_ws = k == 2 ? sk_ws : nullptr;
        a.sk_flags_zero = false;                          // first launch: the launcher's own memset node
        a.out = (pool == 2) ? nullptr : dout[k];          // pool == 2: pooled output only (as the trunk does)
        rcs[k] = launch_conv3x3_patch(a, pool ? dpool[k] : nullptr, s);
        if (rcs[k] != TDRN_OK) break;
        CK(hipStreamSynchronize(s));
        a.sk_flags_zero = true;                           // from here on every launch must leave the flag words zero itself
        for (int i = 0; i < 3; ++i) launch_conv3x3_patch(a, pool ? dpool[k] : nullptr, s);
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < iters; ++i) launch_conv3x3_patch(a, pool ? dpool[k] : nullptr, s);
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        us[k] = ms * 1000.0 / iters;
    }
    int bad = 0;
    {
        unsigned flags[256];
        CK(hipMemcpy(flags, sk_ws, sizeof(flags), hipMemcpyDeviceToHost));
        for (int i = 0; i < 256; ++i)
            if (flags[i] && flags[i] != 0xFFFFFFFFu) { bad = 1; fprintf(stderr, "  chained-split flag %d left at %u\n", i, flags[i]); break; }
    }
    if (rcs[0] == TDRN_OK && rcs[1] == TDRN_OK && rcs[2] == TDRN_OK) {
        std::vector<unsigned short> o0(n_out), o1(n_out);
        for (int k = 1; k < 3; ++k) {
            if (pool != 2) {
                CK(hipMemcpy(o0.data(), dout[0], n_out * es, hipMemcpyDeviceToHost));
                CK(hipMemcpy(o1.data(), dout[k], n_out * es, hipMemcpyDeviceToHost));
                size_t nd = 0, first = 0;
                for (size_t i = 0; i < n_out; ++i)
                    if (o0[i] != o1[i]) { if (!nd) first = i; ++nd; }
                if (nd) { bad = 1; fprintf(stderr, "  arm %d out: %zu of %zu differ, first at %zu (pixel %zu, c %zu): %04x vs %04x\n", k, nd, n_out, first, first / Npad, first % Npad, o0[first], o1[first]); }
            }
            if (pool) {
                CK(hipMemcpy(o0.data(), dpool[0], n_pool * es, hipMemcpyDeviceToHost));
                CK(hipMemcpy(o1.data(), dpool[k], n_pool * es, hipMemcpyDeviceToHost));
                size_t nd = 0, first = 0;
                for (size_t i = 0; i < n_pool; ++i)
                    if (o0[i] != o1[i]) { if (!nd) first = i; ++nd; }
                if (nd) { bad = 1; fprintf(stderr, "  arm %d pool: %zu of %zu differ, first at %zu: %04x vs %04x\n", k, nd, n_pool, first, o0[first], o1[first]); }
            }
        }
    }
    const double gflop = 2.0 * B * H * W * 9.0 * Cin * Cout * 1e-9;
    printf("B%-3d %4dx%-4d %4d->%-4d pool%d relu%d %s | patch rc %d %7.1f us %6.1f TF | pp rc %d %7.1f us %6.1f TF | pp+sk rc %d %7.1f us %6.1f TF | %s\n", B, H, W, Cin, Cout, pool, relu,
           dtype == TDRN_BF16 ? "bf16" : "f16 ", rcs[0], us[0], us[0] > 0 ? gflop / us[0] * 1e3 : 0.0, rcs[1], us[1],
           us[1] > 0 ? gflop / us[1] * 1e3 : 0.0, rcs[2], us[2], us[2] > 0 ? gflop / us[2] * 1e3 : 0.0,
           (rcs[0] || rcs[1] || rcs[2]) ? "LAUNCH-ERROR" : (bad ? "MISMATCH" : "bit-identical"));
    fflush(stdout);
    CK(hipFree(din)); CK(hipFree(dw)); CK(hipFree(dzero)); CK(hipFree(db));
    for (int k = 0; k < 3; ++k) { CK(hipFree(dout[k])); CK(hipFree(dpool[k])); }
    CK(hipFree(sk_ws));
    CK(hipStreamDestroy(s));
    return bad || rcs[0] || rcs[1] || rcs[2];
}

// conv3x3_patch.hip vs the weight-stationary conv3x3_ws.hip (Cin == 64 layers), optionally with the first conv fused in (fuse = 1:
// the layer's input is computed from random fp32 frames by both kernels' producers): bit-compared, timed
static int run_case_ws(int B, int H, int W, int Cout, int pool, int fuse, int relu, int dtype, int iters)
{
    const int Cin = 64, Npad = conv_n_pad(Cout), es = 2;
    const size_t n_in = (size_t)B * H * W * Cin, n_w = (size_t)Npad * 9 * Cin, n_out = (size_t)B * H * W * Npad, n_pool = n_out / 4;
    std::vector<unsigned short> hin(n_in), hw(n_w);
    std::vector<float> hb(Npad), hx((size_t)B * 3 * H * W), hfw(64 * 27), hfb(64);
    unsigned seed = 4242u + (unsigned)(H * 131 + Cout * 7 + fuse);
    for (auto &v : hin) v = dtype == TDRN_BF16 ? host_f32_to_bf16(frand(seed)) : host_f32_to_f16(frand(seed));
    for (size_t i = 0; i < n_w; ++i) {
        const float f = (i / ((size_t)9 * Cin)) < (size_t)Cout ? frand(seed) / 24.0f : 0.f;
        hw[i] = dtype == TDRN_BF16 ? host_f32_to_bf16(f) : host_f32_to_f16(f);
    }
    for (int i = 0; i < Npad; ++i) hb[i] = i < Cout ? frand(seed) * 0.5f : 0.f;
    for (auto &v : hx) v = frand(seed) * 128.f;
    for (auto &v : hfw) v = frand(seed) / 300.f;
    for (auto &v : hfb) v = frand(seed) * 0.3f;
    char *din, *dw, *dzero, *dout[2], *dpool[2];
    float *db, *dx, *dfw, *dfb;
    CK(hipMalloc((void **)&din, n_in * es)); CK(hipMalloc((void **)&dw, n_w * es)); CK(hipMalloc((void **)&dzero, 256));
    CK(hipMalloc((void **)&db, Npad * 4)); CK(hipMalloc((void **)&dx, hx.size() * 4)); CK(hipMalloc((void **)&dfw, 64 * 27 * 4)); CK(hipMalloc((void **)&dfb, 64 * 4));
    for (int k = 0; k < 2; ++k) {
        CK(hipMalloc((void **)&dout[k], n_out * es)); CK(hipMalloc((void **)&dpool[k], n_pool * es));
        CK(hipMemset(dout[k], 0xAB, n_out * es)); CK(hipMemset(dpool[k], 0xCD, n_pool * es));
    }
    CK(hipMemset(dzero, 0, 256));
    CK(hipMemcpy(din, hin.data(), n_in * es, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, hw.data(), n_w * es, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), Npad * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dfw, hfw.data(), 64 * 27 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dfb, hfb.data(), 64 * 4, hipMemcpyHostToDevice));
    ConvArgs a;
    a.in = din; a.w = dw; a.bias = db; a.zero_page = dzero;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Ho = H; a.Wo = W; a.Cout = Cout; a.Npad = Npad;
    a.kh = a.kw = 3; a.stride = 1; a.pad = 1; a.dil = 1; a.relu = relu; a.dtype = dtype;
    a.o_cs = Npad; a.o_rs = (long long)W * Npad; a.o_bs = (long long)H * W * Npad;
    if (fuse) { a.fuse_x = dx; a.fuse_w = dfw; a.fuse_b = dfb; a.fuse_cout = 64; }
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double us[2] = {0, 0};
    int rcs[2] = {0, 0};
    conv_pp_force(0);
    for (int k = 0; k < 2; ++k) {                        // 0: conv3x3_patch, 1: conv3x3_ws
        conv_ws_force(k ? 2 : 0);
        a.out = (pool == 2) ? nullptr : dout[k];
        rcs[k] = k ? launch_conv3x3_ws(a, pool ? dpool[k] : nullptr, s) : launch_conv3x3_patch(a, pool ? dpool[k] : nullptr, s);
        if (rcs[k] != TDRN_OK) break;
        CK(hipStreamSynchronize(s));
        for (int i = 0; i < 3; ++i) launch_conv3x3_patch(a, pool ? dpool[k] : nullptr, s);
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < iters; ++i) launch_conv3x3_patch(a, pool ? dpool[k] : nullptr, s);
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        us[k] = ms * 1000.0 / iters;
    }
    conv_ws_force(-1);
    conv_pp_force(-1);
    int bad = 0;
    if (rcs[0] == TDRN_OK && rcs[1] == TDRN_OK) {
        std::vector<unsigned short> o0(n_out), o1(n_out);
        if (pool != 2) {
            CK(hipMemcpy(o0.data(), dout[0], n_out * es, hipMemcpyDeviceToHost)); CK(hipMemcpy(o1.data(), dout[1], n_out * es, hipMemcpyDeviceToHost));
            size_t nd = 0, first = 0;
            for (size_t i = 0; i < n_out; ++i)
                if (o0[i] != o1[i]) { if (!nd) first = i; ++nd; }
            if (nd) { bad = 1; fprintf(stderr, "  ws out: %zu of %zu differ, first at %zu (pixel %zu = b %zu y %zu x %zu, c %zu): %04x vs %04x\n", nd, n_out, first, first / Npad,
                                       first / Npad / ((size_t)H * W), first / Npad / W % H, first / Npad % W, first % Npad, o0[first], o1[first]); }
        }
        if (pool) {
            CK(hipMemcpy(o0.data(), dpool[0], n_pool * es, hipMemcpyDeviceToHost)); CK(hipMemcpy(o1.data(), dpool[1], n_pool * es, hipMemcpyDeviceToHost));
            size_t nd = 0, first = 0;
            for (size_t i = 0; i < n_pool; ++i)
                if (o0[i] != o1[i]) { if (!nd) first = i; ++nd; }
            if (nd) { bad = 1; fprintf(stderr, "  ws pool: %zu of %zu differ, first at %zu (pooled pixel %zu = b %zu y %zu x %zu, c %zu): %04x vs %04x\n", nd, n_pool, first, first / Npad,
                                       first / Npad / ((size_t)H * W / 4), first / Npad / (W / 2) % (H / 2), first / Npad % (W / 2), first % Npad, o0[first], o1[first]); }
        }
    }
    const double gflop = 2.0 * B * H * W * 9.0 * Cin * Cout * 1e-9 + (fuse ? 2.0 * B * H * W * 27.0 * 64 * 1e-9 : 0.0);
    printf("WS B%-3d %4dx%-4d   64->%-4d pool%d fuse%d relu%d %s | patch rc %d %7.1f us %6.1f TF | ws rc %d %7.1f us %6.1f TF | %s\n", B, H, W, Cout, pool, fuse, relu,
           dtype == TDRN_BF16 ? "bf16" : "f16 ", rcs[0], us[0], us[0] > 0 ? gflop / us[0] * 1e3 : 0.0, rcs[1], us[1], us[1] > 0 ? gflop / us[1] * 1e3 : 0.0,
           (rcs[0] || rcs[1]) ? "LAUNCH-ERROR" : (bad ? "MISMATCH" : "bit-identical"));
    fflush(stdout);
    CK(hipFree(din)); CK(hipFree(dw)); CK(hipFree(dzero)); CK(hipFree(db)); CK(hipFree(dx)); CK(hipFree(dfw)); CK(hipFree(dfb));
    for (int k = 0; k < 2; ++k) { CK(hipFree(dout[k])); CK(hipFree(dpool[k])); }
    CK(hipStreamDestroy(s));
    return bad || rcs[0] || rcs[1];
}

static int run_ws_suite()
{
    int fails = 0;
    struct C { int B, H, W, Cout, pool, fuse, relu, dt; };
    const C cases[] = {
        // corners: one strip, ragged unit counts, both epilogues, two cout tiles, T = 1 segments, small batches (units >= 192 needed)
        {2, 64, 64, 64, 0, 0, 1, TDRN_BF16}, {3, 64, 96, 128, 0, 0, 1, TDRN_BF16}, {2, 64, 64, 64, 2, 0, 1, TDRN_F16}, {2, 64, 64, 64, 2, 1, 1, TDRN_BF16},
        {5, 128, 128, 128, 0, 0, 0, TDRN_BF16}, {3, 192, 192, 64, 2, 1, 1, TDRN_F16}, {24, 8, 256, 64, 2, 0, 1, TDRN_BF16}, {4, 160, 160, 192, 0, 0, 1, TDRN_BF16},
        {1, 320, 320, 64, 2, 1, 1, TDRN_BF16}, {1, 160, 160, 128, 0, 0, 1, TDRN_BF16}, {9, 72, 96, 64, 2, 0, 1, TDRN_BF16},
        // the 320 net at batch 32 (config 2): conv1_2 fused (+pool), conv1_2 from a materialised input (+pool), conv2_1
        {32, 320, 320, 64, 2, 1, 1, TDRN_BF16}, {32, 320, 320, 64, 2, 0, 1, TDRN_BF16}, {32, 160, 160, 128, 0, 0, 1, TDRN_BF16},
        // the 512 net at batch 16 (config 3)
        {16, 512, 512, 64, 2, 1, 1, TDRN_F16}, {16, 256, 256, 128, 0, 0, 1, TDRN_F16},
    };
    for (const C &c : cases) fails += run_case_ws(c.B, c.H, c.W, c.Cout, c.pool, c.fuse, c.relu, c.dt, 20);
    printf("%s\n", fails ? "WS FAILED" : "WS ALL BIT-IDENTICAL");
    return fails ? 1 : 0;
}

int main(int argc, char **argv)
{
    if (argc >= 2 && !strcmp(argv[1], "ws")) {
        if (argc >= 8) return run_case_ws(atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), atoi(argv[6]), atoi(argv[7]), 1, argc > 8 ? atoi(argv[8]) : TDRN_BF16, argc > 9 ? atoi(argv[9]) : 20);
        return run_ws_suite();
    }
    if (argc >= 6) {
        const int B = atoi(argv[1]), H = atoi(argv[2]), W = atoi(argv[3]), Cin = atoi(argv[4]), Cout = atoi(argv[5]);
        const int pool = argc > 6 ? atoi(argv[6]) : 0, relu = argc > 7 ? atoi(argv[7]) : 1, dt = argc > 8 ? atoi(argv[8]) : TDRN_BF16;
        const int iters = argc > 9 ? atoi(argv[9]) : 20;
        return run_case(B, H, W, Cin, Cout, pool, relu, dt, iters);
    }
    int fails = 0;
    struct C { int B, H, W, Cin, Cout, pool, relu, dt; };
    const C cases[] = {
        // correctness corners first (small batches: ragged item counts, fewer items than CUs, workgroups without work)
        {1, 80, 80, 128, 256, 0, 1, TDRN_BF16}, {3, 80, 80, 256, 256, 1, 1, TDRN_BF16}, {2, 40, 40, 256, 512, 0, 1, TDRN_F16},
        {1, 40, 40, 512, 512, 0, 0, TDRN_BF16}, {5, 20, 20, 512, 512, 0, 1, TDRN_BF16}, {2, 64, 64, 512, 512, 2, 1, TDRN_F16},
        {1, 128, 128, 128, 256, 0, 1, TDRN_BF16}, {7, 40, 40, 256, 256, 0, 1, TDRN_BF16}, {2, 48, 48, 256, 256, 0, 1, TDRN_BF16},
        // chained split with ragged cuts: 257 / 300 / 511 / 650 items, 2 and 8 chunks, both cout-tile counts
        {11, 80, 80, 128, 256, 0, 1, TDRN_BF16}, {12, 80, 80, 256, 256, 0, 1, TDRN_F16}, {41, 40, 40, 512, 512, 0, 1, TDRN_BF16}, {26, 80, 80, 128, 256, 0, 0, TDRN_BF16}, {5, 128, 128, 256, 512, 0, 1, TDRN_BF16},
        // the 320 net at batch 32 (config 2): conv3_1..3_3 (+pool), conv4_1..4_3, TCB 40x40
        {32, 80, 80, 128, 256, 0, 1, TDRN_BF16}, {32, 80, 80, 256, 256, 0, 1, TDRN_BF16}, {32, 80, 80, 256, 256, 2, 1, TDRN_BF16},
        {32, 40, 40, 256, 512, 0, 1, TDRN_BF16}, {32, 40, 40, 512, 512, 0, 1, TDRN_BF16}, {32, 40, 40, 512, 256, 0, 1, TDRN_BF16},
        {32, 40, 40, 256, 256, 0, 1, TDRN_BF16}, {32, 20, 20, 512, 512, 0, 1, TDRN_BF16},
        // the 512 net at batch 16 (config 3)
        {16, 128, 128, 256, 256, 0, 1, TDRN_F16}, {16, 64, 64, 512, 512, 0, 1, TDRN_F16},
    };
    for (const C &c : cases) fails += run_case(c.B, c.H, c.W, c.Cin, c.Cout, c.pool, c.relu, c.dt, 20);
    printf("%s\n", fails ? "FAILED" : "ALL BIT-IDENTICAL");
    return fails ? 1 : 0;
}
