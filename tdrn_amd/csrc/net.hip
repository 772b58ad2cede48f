// net.hip -- host-side layer plan, weight packing (BatchNorm folding, OIHW -> [Cout][tap][Cin])
// and the forward executor for the reference's model families.  No device allocation: the weight
// blob and the activation workspace are caller-owned (tdrn_hip.h).
//
//   model/dualrefinedet_vggbn.py:10-206      build_drn_vgg()
//   model/dualrefinedet_mobilenet.py:8-199   build_drn_mobilenet()
//   model/ssd4scale_mobile.py:9-140          build_ssd4scale_mobile()
//   model/refinedet_vgg.py:27-219            build_refinedet_vgg()
//   model/ssd4scale_vgg.py                   build_ssd4scale_vgg()
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <mutex>
#include <vector>

#include "kernels.h"

namespace tdrn {

unsigned short host_f32_to_bf16(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);   // NaN
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
unsigned short host_f32_to_f16(float f)
{
    _Float16 h = (_Float16)f;
    unsigned short r;
    memcpy(&r, &h, 2);
    return r;
}

namespace {

enum OpKind { OP_FIRST, OP_CONV, OP_POOL, OP_L2NORM, OP_DW, OP_OFFSET, OP_DEFORM, OP_SOFTMAX, OP_OFF_OUT, OP_LOC_OUT,
              OP_REFLOC_IN };
enum OutKind { OUT_TENSOR = 0, OUT_ARM_LOC = 1, OUT_ODM_LOC = 2, OUT_CONF = 3 };

struct Tensor { int C, Cpad, H, W; bool f32; size_t off; /* bytes per sample from workspace start */ std::string label; };
struct ParamSpec { std::string name; std::vector<int64_t> shape; };

struct Op {
    OpKind kind;
    int in = -1, out = -1, res = -1;
    int Cin = 0, Cout = 0, Npad = 0, k = 1, stride = 1, pad = 0, dil = 1, relu = 0, phases = 1, ceil = 0;
    int out_kind = OUT_TENSOR, scale = 0, hw = 0;        // head ops: which pyramid level
    int G = 1, n_branches = 1, k2 = 0, pad2 = 0;          // deform
    int off_t = -1, off_c0[2] = {0, 0}, off_n = 0;        // offset tensor (fp32 NHWC), channel starts
    int loc_src = OUT_ARM_LOC;                            // OP_OFFSET input: ARM loc view or a ref_loc tensor
    std::string w, b, bn, w2, b2;                          // parameter names (w2/b2: second source)
    size_t w_off = 0, b_off = 0, w2_off = 0;               // blob offsets
    int y_t = -1, y_cols = 0;                              // deform, transform-then-sample plans: the Y tensor (per-tap partial outputs)
    int y_groups = 1;                                      // ... output-column groups of <= 80 (12 + 3 * classes > 80: VID's 31 classes = 2, COCO's 81 = 4): one Y region,
                                                           // one transform and one sampling launch per group (region g of the tensor: y_cols * H * W * B elements each)
    size_t wt_off = 0, bt_off = 0;                         // ... its 1x1 GEMM weights [y_groups][y_cols][Cin] and zero bias
    double flops = 0, bytes = 0;                           // algorithmic, per sample
    int stat = 0;
    int lane = 0;                                          // HIP stream lane (0 = the caller's stream)
    int pool_t = -1;                                       // conv: fused MaxPool2d(2,2) output tensor (patch kernel)
    int splitk = 1;                                        // conv: K slices, fixed per layer at plan time
    bool chain_tag = false;                                // conv: candidate for the one-launch chain of small top-of-pyramid layers
    int chain = -1;                                        // ... its stage index in that launch (conv_igemm.hip conv_chain_kernel), or -1
    int fused_dw = 0;                                      // depthwise op: its launch also computes the next op, the pointwise conv (dwpw.hip); that conv: 1 = computed there
    size_t chain_partial = 0;                              // ... its split-K slab inside the chain's slab region (bytes per sample)
};

// Side-lane streams and no-timing events are POOLED per process instead of destroyed with their net: a hipGraph captured for
// a net created after another net's streams / events had been destroyed crashed inside hipGraphLaunch (ROCm 7.2; reproduced
// with bench.py's three engines: a 16-bit engine destroyed, then the fp32 engine's step captured and replayed).  A net takes
// them from the pool and hands them back in tdrn_net_destroy; nothing in the pool is ever in use by two nets at a time.
namespace pool {
// (keyed by device: a net created while device 1 is current must not inherit device 0's handles)
std::mutex mu;
std::map<int, std::vector<hipStream_t>> streams;
std::map<int, std::vector<hipEvent_t>> events;
std::map<int, std::vector<hipEvent_t>> timing_events;
std::vector<unsigned *> status_words;      // pinned, device-mapped host memory (visible to every device): 64 bytes each
int cur_dev()
{
    int d = 0;
    (void)hipGetDevice(&d);
    return d;
}
int get_stream(int dev, hipStream_t *s)
{
    {
        std::lock_guard<std::mutex> g(mu);
        auto &v = streams[dev];
        if (!v.empty()) { *s = v.back(); v.pop_back(); return TDRN_OK; }
    }
    return hip_status(hipStreamCreateWithFlags(s, hipStreamNonBlocking));
}
int get_event(int dev, hipEvent_t *e)
{
    {
        std::lock_guard<std::mutex> g(mu);
        auto &v = events[dev];
        if (!v.empty()) { *e = v.back(); v.pop_back(); return TDRN_OK; }
    }
    return hip_status(hipEventCreateWithFlags(e, hipEventDisableTiming));
}
int get_timing_event(int dev, hipEvent_t *e)
{
    {
        std::lock_guard<std::mutex> g(mu);
        auto &v = timing_events[dev];
        if (!v.empty()) { *e = v.back(); v.pop_back(); return TDRN_OK; }
    }
    return hip_status(hipEventCreate(e));
}
int get_status(unsigned **w)
{
    {
        std::lock_guard<std::mutex> g(mu);
        if (!status_words.empty()) { *w = status_words.back(); status_words.pop_back(); memset(*w, 0, 64); return TDRN_OK; }
    }
    void *p = nullptr;
    TDRN_HIP_TRY(hipHostMalloc(&p, 64, hipHostMallocMapped | hipHostMallocPortable));
    memset(p, 0, 64);
    *w = (unsigned *)p;
    return TDRN_OK;
}
void put_timing_event(int dev, hipEvent_t e) { if (e) { std::lock_guard<std::mutex> g(mu); timing_events[dev].push_back(e); } }
void put_stream(int dev, hipStream_t s) { if (s) { std::lock_guard<std::mutex> g(mu); streams[dev].push_back(s); } }
void put_event(int dev, hipEvent_t e) { if (e) { std::lock_guard<std::mutex> g(mu); events[dev].push_back(e); } }
void put_status(unsigned *w) { if (w) { std::lock_guard<std::mutex> g(mu); status_words.push_back(w); } }
}  // namespace pool

const char *kStatNames[] = {"conv_igemm_mfma", "first_conv", "maxpool2x2", "l2norm", "dwconv3x3", "offset_conv1x1",
                            "deform_gemm_mfma", "softmax21", "layout", "conv3x3_patch_mfma", "dwpw_mfma"};
enum { ST_CONV, ST_FIRST, ST_POOL, ST_L2, ST_DW, ST_OFFSET, ST_DEFORM, ST_SOFTMAX, ST_LAYOUT, ST_CONV3, ST_DWPW, ST_COUNT };

}  // namespace
}  // namespace tdrn

using namespace tdrn;

struct tdrn_net {
    tdrn_net_config cfg{};
    int es = 2;
    std::vector<Tensor> tensors;
    std::vector<ParamSpec> params;
    std::map<std::string, size_t> param_index;
    std::map<std::string, std::vector<float>> staged;
    std::vector<Op> ops;
    size_t ws_per_sample = 0, blob_bytes = kZeroPageBytes;
    size_t ws_fixed = 0;                 // batch-independent tail of the workspace: scratch of conv3x3_pp.hip's chained split (main lane)
    int P = 0, fm[4] = {0, 0, 0, 0}, scale_off[5] = {0, 0, 0, 0, 0};
    bool weights_ready = false;
    int profile = 0;                   // 0 off; 1 = events around every launch, single stream; 2 = the same with the side lanes on
    std::vector<hipEvent_t> ev;
    std::vector<int> ev_stat;
    std::vector<int> ev_op;
    tdrn_kernel_stat stats[ST_COUNT];
    int last_batch = 0;
    // independent branches of the tail (TCB laterals, ARM heads) run on side streams; dependencies
    // between lanes are hipEvents on the producing tensor.  Created lazily at the first forward.
    static constexpr int kLanes = 4;
    static constexpr size_t kTailCtl = 256;  // bytes of chain counters in front of the chained split's scratch (workspace tail)
    size_t splitk_off[kLanes] = {0, 0, 0, 0};   // per-lane split-K slab region (bytes per sample from workspace start)
    int cur_lane = 0;
    // Side-lane convs (TCB laterals, ARM heads, offset convs) are held back until conv5_3 has been computed: released on their
    // true inputs (L2Norm of conv4_3) they share the CUs with conv5_1..5_3 and stretch the trunk, the critical path, by
    // 0.24 ms; held back, they run beside conv6/conv7 and the small top-down layers instead (+1.4 % frames/s; held until fc7
    // or capped to 128..224 workgroups: no further gain).  TDRN_LATE_SIDE 0: off, 2: until fc7; TDRN_SIDE_GRID: the cap.
    int t_late = -1, t_late2 = -1;
    std::vector<int> chain_ops;          // the chain launch's member ops in stage order (empty: no chain)
    size_t chain_partial_off = 0;        // the chain's split-K slab region (bytes per sample from workspace start)
    bool pp_sk_planned = false;          // some main-lane conv may use conv3x3_pp.hip's chained split
    int fuse_first = -1;                 // index of the conv whose patch loader computes the first conv itself (16-bit modes), or -1
    int x_t = -1;                        // fp32 (3, S, S) workspace tensor: the net input when the caller hands uint8 planes to a plan whose first conv reads fp32
    int late_side = 1, side_grid = 0, main_grid = 0;
    bool use_lanes = true, lanes_ready = false, deform_split = true;
    int plan_error = TDRN_OK;
    int splitk_ref_batch = 32;          // split-K factors are planned for this batch (the benchmark's) and used for every batch (TDRN_SPLITK_REF)
    const void *offs_ws = nullptr;      // ssd4scale deform: the workspace / batch whose offset tensors the last forward filled
    int offs_batch = 0, offs_key_batch = 0;
    int dev = -1;                       // the device the pooled handles below belong to (the one current at the first forward)
    unsigned *status = nullptr;         // host-visible status words (pinned; tdrn_net_check): [0] chained split, [1] chain launch
    int kdisable = 0, fault_handoff = 0;
    hipStream_t side[kLanes - 1] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_zero = nullptr, ev_skz = nullptr, ev_join[kLanes - 1] = {nullptr, nullptr, nullptr};
    std::vector<hipEvent_t> tensor_ev;
    std::vector<int> tensor_lane;
    std::vector<char> tensor_shared;

    // ---- plan building --------------------------------------------------------------------
    int T(int C, int H, int W, bool f32 = false)
    {
        Tensor t;
        t.C = C; t.H = H; t.W = W; t.f32 = f32;
        t.Cpad = f32 ? C : (int)align_up((size_t)C, kChanPad);
        t.off = ws_per_sample;
        ws_per_sample += align_up((size_t)t.Cpad * H * W * (f32 ? 4 : es), 256);
        tensors.push_back(t);
        return (int)tensors.size() - 1;
    }
    void P_(const std::string &name, std::vector<int64_t> shape)
    {
        param_index[name] = params.size();
        params.push_back(ParamSpec{name, std::move(shape)});
    }
    void bn_params(const std::string &bn, int C)
    {
        P_(bn + ".weight", {C}); P_(bn + ".bias", {C}); P_(bn + ".running_mean", {C}); P_(bn + ".running_var", {C});
    }
    void push(Op &o) { o.lane = cur_lane; ops.push_back(o); }
    void label(int t, const std::string &l) { if (t >= 0) tensors[t].label = l; }
    size_t blob(size_t bytes)
    {
        const size_t o = blob_bytes;
        blob_bytes += align_up(bytes, 256);
        return o;
    }

    // first conv (Cin = 3), BN folded
    int first_conv(const std::string &w, bool bias, const std::string &bn, int Cout, int stride, int S)
    {
        const int So = (S + 2 - 3) / stride + 1;
        Op o; o.kind = OP_FIRST; o.stat = ST_FIRST;
        o.Cin = 3; o.Cout = Cout; o.stride = stride; o.relu = 1; o.hw = S;
        o.w = w; o.bn = bn;
        P_(w + ".weight", {Cout, 3, 3, 3});
        if (bias) { o.b = w; P_(w + ".bias", {Cout}); }
        if (!bn.empty()) bn_params(bn, Cout);
        o.out = T(Cout, So, So);
        o.w_off = blob((size_t)Cout * 27 * 4);
        o.b_off = blob((size_t)tensors[o.out].Cpad * 4);
        o.flops = 2.0 * So * So * Cout * 27;
        o.bytes = 3.0 * S * S * 4 + (double)So * So * tensors[o.out].Cpad * es;
        label(o.out, w);
        push(o);
        return o.out;
    }

    // dense conv -> NHWC tensor (out_kind == OUT_TENSOR) or fp32 head output view
    int conv(int in, const std::string &w, bool bias, const std::string &bn, int Cout, int k, int stride, int pad, int dil,
             int relu, int res = -1, int out_kind = OUT_TENSOR, int scale = 0, const std::string &w2 = "", int k2 = 0)
    {
        const Tensor ti = tensors[in];
        Op o; o.kind = OP_CONV; o.stat = ST_CONV;
        o.in = in; o.res = res; o.Cin = ti.Cpad; o.Cout = Cout; o.k = k; o.stride = stride; o.pad = pad; o.dil = dil;
        o.relu = relu; o.out_kind = out_kind; o.scale = scale; o.w = w; o.bn = bn; o.w2 = w2; o.k2 = k2;
        const int Ho = (ti.H + 2 * pad - (dil * (k - 1) + 1)) / stride + 1;
        const int Wo = (ti.W + 2 * pad - (dil * (k - 1) + 1)) / stride + 1;
        P_(w + ".weight", {Cout, ti.C, k, k});
        if (bias) { o.b = w; P_(w + ".bias", {Cout}); }
        if (!bn.empty()) bn_params(bn, Cout);
        if (!w2.empty()) {   // a second, smaller conv merged into the same taps (refinedet multihead)
            P_(w2 + ".weight", {Cout, ti.C, k2, k2});
            if (bias) { o.b2 = w2; P_(w2 + ".bias", {Cout}); }
        }
        if (out_kind == OUT_TENSOR) {
            o.out = T(Cout, Ho, Wo);
            o.Cout = tensors[o.out].Cpad;            // pad channels are written as zeros
        }
        o.Npad = conv_n_pad(o.Cout);
        o.hw = Ho * 65536 + Wo;
        o.w_off = blob((size_t)o.Npad * k * k * o.Cin * es);
        o.b_off = blob((size_t)o.Npad * 4);
        o.flops = 2.0 * Ho * Wo * Cout * (double)k * k * ti.C;
        o.bytes = (double)ti.H * ti.W * ti.Cpad * es + (double)Ho * Wo * o.Cout * (out_kind == OUT_TENSOR ? es : 4) +
                  (res >= 0 ? (double)Ho * Wo * o.Cout * es : 0.0);
        label(o.out, w);
        push(o);
        return o.out;
    }

    // ConvTranspose2d(k=2, s=2) + residual + ReLU as four phase GEMMs
    int conv_transpose2(int in, const std::string &w, bool bias, int Cout, int res, int relu)
    {
        const Tensor ti = tensors[in];
        Op o; o.kind = OP_CONV; o.stat = ST_CONV;
        o.in = in; o.res = res; o.Cin = ti.Cpad; o.k = 1; o.relu = relu; o.phases = 4; o.w = w;
        P_(w + ".weight", {ti.C, Cout, 2, 2});
        if (bias) { o.b = w; P_(w + ".bias", {Cout}); }
        o.out = T(Cout, ti.H * 2, ti.W * 2);
        o.Cout = tensors[o.out].Cpad;
        o.Npad = conv_n_pad(o.Cout);
        o.hw = ti.H * 65536 + ti.W;
        o.w_off = blob((size_t)4 * o.Npad * o.Cin * es);
        o.b_off = blob((size_t)o.Npad * 4);
        o.flops = 2.0 * 4 * ti.H * ti.W * (double)Cout * ti.C;
        o.bytes = (double)ti.H * ti.W * ti.Cpad * es + 2.0 * 4 * ti.H * ti.W * o.Cout * es;
        label(o.out, w);
        push(o);
        return o.out;
    }

    // MaxPool2d(2,2) right after a 3x3 conv whose full-resolution output nobody else reads: fused into
    // the conv's epilogue when the warp-specialised kernel takes the layer with 2-D tiles.
    bool can_fuse_pool(int in) const
    {
        if (ops.empty()) return false;
        const Op &o = ops.back();
        if (o.kind != OP_CONV || o.out != in || o.out_kind != OUT_TENSOR || o.k != 3 || o.stride != 1 || o.pad != 1 ||
            o.dil != 1 || o.phases != 1 || o.res >= 0 || o.pool_t >= 0) return false;
        const Tensor &t = tensors[in];
        if ((t.H & 1) || (t.W & 1)) return false;
        return (t.W % 32 == 0 && t.H % 8 == 0) || (t.W % 16 == 0 && t.H % 16 == 0);
    }
    int pool(int in, int ceil_mode, bool in_needed_elsewhere = false)
    {
        if (!in_needed_elsewhere && can_fuse_pool(in)) {
            const Tensor ti = tensors[in];
            const int out = T(ti.C, ti.H / 2, ti.W / 2);
            label(out, "pool:" + ti.label);
            ops.back().pool_t = out;
            tensors[in].label = "";                  // not materialised on the fused path
            ops.back().bytes += (double)(ti.H / 2) * (ti.W / 2) * ti.Cpad * es - (double)ti.H * ti.W * ti.Cpad * es;
            return out;
        }
        const Tensor ti = tensors[in];
        Op o; o.kind = OP_POOL; o.stat = ST_POOL; o.in = in; o.ceil = ceil_mode;
        const int Ho = ceil_mode ? (ti.H + 1) / 2 : ti.H / 2, Wo = ceil_mode ? (ti.W + 1) / 2 : ti.W / 2;
        o.out = T(ti.C, Ho, Wo);
        o.bytes = ((double)ti.H * ti.W + (double)Ho * Wo) * ti.Cpad * es;
        label(o.out, "pool:" + ti.label);
        push(o);
        return o.out;
    }

    int l2norm(int in, const std::string &name)
    {
        const Tensor ti = tensors[in];
        Op o; o.kind = OP_L2NORM; o.stat = ST_L2; o.in = in; o.w = name;
        P_(name + ".weight", {ti.C});
        o.out = T(ti.C, ti.H, ti.W);
        o.w_off = blob((size_t)ti.Cpad * 4);
        o.bytes = 2.0 * ti.H * ti.W * ti.Cpad * es;
        label(o.out, name);
        push(o);
        return o.out;
    }

    int dwconv(int in, const std::string &w, const std::string &bn, int stride)
    {
        const Tensor ti = tensors[in];
        Op o; o.kind = OP_DW; o.stat = ST_DW; o.in = in; o.stride = stride; o.relu = 1; o.w = w; o.bn = bn;
        P_(w + ".weight", {ti.C, 1, 3, 3});
        bn_params(bn, ti.C);
        const int Ho = (ti.H + 2 - 3) / stride + 1, Wo = (ti.W + 2 - 3) / stride + 1;
        o.out = T(ti.C, Ho, Wo);
        o.w_off = blob((size_t)9 * ti.Cpad * 4);
        o.b_off = blob((size_t)ti.Cpad * 4);
        o.flops = 2.0 * Ho * Wo * ti.C * 9;
        o.bytes = ((double)ti.H * ti.W + (double)Ho * Wo) * ti.Cpad * es;
        label(o.out, w);
        push(o);
        return o.out;
    }
    // conv_dw block, model/networks.py:736-745
    int conv_dw(int in, const std::string &name, int Cout, int stride)
    {
        const int d = dwconv(in, name + ".0", name + ".1", stride);
        return conv(d, name + ".3", false, name + ".4", Cout, 1, 1, 0, 1, 1);
    }

    // 1x1 offset convs on the 12-channel loc map of pyramid level `scale`
    int offset_conv(int scale, int H, int W, const std::string &w1, const std::string &w2, bool bias, int n1, int n2,
                    int loc_src, int ref_tensor = -1)
    {
        Op o; o.kind = OP_OFFSET; o.stat = ST_OFFSET; o.scale = scale; o.hw = H * W; o.w = w1; o.w2 = w2;
        o.loc_src = loc_src; o.in = ref_tensor;
        P_(w1 + ".weight", {n1, 12, 1, 1});
        if (bias) { o.b = w1; P_(w1 + ".bias", {n1}); }
        if (!w2.empty()) {
            P_(w2 + ".weight", {n2, 12, 1, 1});
            if (bias) { o.b2 = w2; P_(w2 + ".bias", {n2}); }
        } else {
            n2 = 0;
        }
        o.off_n = n1 + n2; o.off_c0[0] = 0; o.off_c0[1] = n1;
        o.out = T(o.off_n, H, W, true);
        o.w_off = blob((size_t)o.off_n * 12 * 4);
        o.b_off = blob((size_t)o.off_n * 4);
        o.flops = 2.0 * H * W * o.off_n * 12;
        o.bytes = (double)H * W * (12 + o.off_n) * 4;
        label(o.out, w1);
        push(o);
        return o.out;
    }

    // fused deformable heads of one pyramid level: [loc ; conf] rows, 1 or 2 branches
    void deform_heads(int in, int off_t, int scale, int G, const std::string &loc1, const std::string &conf1,
                      const std::string &loc2, const std::string &conf2, int off_c1, int out_loc_kind)
    {
        const Tensor ti = tensors[in];
        const int nc3 = 3 * cfg.num_classes;
        // shape_check, deform_conv_cuda.c:75-76 "input image is smaller than kernel": the reference throws when a
        // 5x5 multihead branch meets the 3x3 map of a 192-pixel frame
        const int kmax = loc2.empty() ? 3 : 5;
        if (ti.H < kmax || ti.W < kmax) plan_error = TDRN_E_SHAPE;
        Op o; o.kind = OP_DEFORM; o.stat = ST_DEFORM; o.in = in; o.off_t = off_t; o.scale = scale; o.G = G;
        o.Cin = ti.Cpad; o.Cout = 12 + nc3; o.Npad = deform_n_pad(o.Cout);
        o.k = 3; o.pad = 1; o.w = loc1; o.b = conf1; o.out_kind = out_loc_kind;
        P_(loc1 + ".weight", {12, ti.C, 3, 3});
        P_(conf1 + ".weight", {nc3, ti.C, 3, 3});
        o.w_off = blob((size_t)o.Npad * 9 * o.Cin * es);
        o.off_c0[0] = 0;
        double taps = 9;
        if (!loc2.empty()) {
            o.n_branches = 2; o.k2 = 5; o.pad2 = 2; o.w2 = loc2; o.b2 = conf2; o.off_c0[1] = off_c1;
            P_(loc2 + ".weight", {12, ti.C, 5, 5});
            P_(conf2 + ".weight", {nc3, ti.C, 5, 5});
            o.w2_off = blob((size_t)o.Npad * 25 * o.Cin * es);
            taps += 25;
        }
        o.hw = ti.H * 65536 + ti.W;
        o.flops = 2.0 * ti.H * ti.W * o.Cout * taps * ti.C;
        o.bytes = (double)ti.H * ti.W * (ti.Cpad * es + (o.Cout + 2 * taps * G) * 4);
        // 16-bit plans, one deformable group: transform (1x1 GEMM into per-tap partial outputs), then sample (deform.hip);
        // TDRN_DEFORM_TS=0 keeps the fused gather kernel
        {
            const char *e = getenv("TDRN_DEFORM_TS");
            const bool ts_on = e ? atoi(e) != 0 : !(cfg.plan_flags & TDRN_PLAN_NO_DEFORM_TS);
            if (ts_on && cfg.dtype != TDRN_F32 && G == 1 && (int)taps <= 34) {
                o.y_groups = (o.Cout + 79) / 80;
                o.y_cols = deform_sample_cols((int)taps);
                o.y_t = T(o.y_cols * o.y_groups, ti.H, ti.W);
                o.wt_off = blob((size_t)o.y_groups * o.y_cols * o.Cin * es);
                o.bt_off = blob((size_t)o.y_groups * o.y_cols * 4);
            }
        }
        push(o);
    }

    void softmax_op()
    {
        Op o; o.kind = OP_SOFTMAX; o.stat = ST_SOFTMAX;
        o.bytes = 2.0 * P * cfg.num_classes * 4;
        push(o);
    }
    void offsets_out(int scale, int off_t, int n) { Op o; o.kind = OP_OFF_OUT; o.stat = ST_LAYOUT; o.scale = scale; o.in = off_t; o.Cout = n; push(o); }
    void loc_maps_out(int scale) { Op o; o.kind = OP_LOC_OUT; o.stat = ST_LAYOUT; o.scale = scale; push(o); }
    int ref_loc_in(int scale, int H, int W)
    {
        Op o; o.kind = OP_REFLOC_IN; o.stat = ST_LAYOUT; o.scale = scale; o.hw = H * W;
        o.out = T(12, H, W, true);
        push(o);
        return o.out;
    }

    void set_pyramid(int s0)
    {
        fm[0] = s0; fm[1] = s0 / 2; fm[2] = s0 / 4; fm[3] = s0 / 8;
        scale_off[0] = 0;
        for (int i = 0; i < 4; ++i) scale_off[i + 1] = scale_off[i] + fm[i] * fm[i] * 3;
        P = scale_off[4];
    }

    // ---- model families ---------------------------------------------------------------------
    // VGG16 trunk (model/networks.py:136-163).  Returns conv4_3, conv5_3, fc7 tensors (post-ReLU).
    void vgg_trunk(int S, bool bn, int c7, int &c43, int &c53, int &fc7)
    {
        static const int cfgv[] = {64, 64, -1, 128, 128, -1, 256, 256, 256, -2, 512, 512, 512, -1, 512, 512, 512};
        int idx = 0, x = -1, nconv = 0;
        for (int v : cfgv) {
            if (v < 0) {
                x = pool(x, v == -2, nconv == 10 || nconv == 13);
                idx += 1;
                continue;
            }
            const std::string name = "backbone." + std::to_string(idx);
            const std::string bnn = bn ? "backbone." + std::to_string(idx + 1) : "";
            if (x < 0) x = first_conv(name, true, bnn, v, 1, S);
            else x = conv(x, name, true, bnn, v, 3, 1, 1, 1, 1);
            idx += bn ? 3 : 2;
            ++nconv;
            if (nconv == 10) c43 = x;
            if (nconv == 13) { c53 = x; t_late = x; }
        }
        x = pool(x, 0, true);   // pool5_ds (conv5_3 also feeds L2Norm_5_3)
        idx += 1;
        x = conv(x, "backbone." + std::to_string(idx), true, bn ? "backbone." + std::to_string(idx + 1) : "", 1024, 3, 1, 6, 6, 1);
        idx += bn ? 3 : 2;
        fc7 = conv(x, "backbone." + std::to_string(idx), true, bn ? "backbone." + std::to_string(idx + 1) : "", c7, 1, 1, 0, 1, 1);
        t_late2 = fc7;
    }

    // TCB / FPN (dualrefinedet_vggbn.py:30-34,97-114,166-178).  Returns the 4 ODM sources.
    void tcb(const int src[4], bool bias, int odm[4])
    {
        int x = conv(src[3], "last_layer_trans.0", bias, "", 256, 3, 1, 1, 1, 1);
        ops.back().chain_tag = true;
        x = conv(x, "last_layer_trans.2", bias, "", 256, 3, 1, 1, 1, 0);
        ops.back().chain_tag = true;
        x = conv(x, "last_layer_trans.3", bias, "", 256, 3, 1, 1, 1, 0);
        ops.back().chain_tag = true;
        odm[3] = x;
        int t[3];
        for (int s = 0; s < 3; ++s) {
            cur_lane = s == 0 ? 1 : 2;      // lateral branches are independent of the top-down chain
            const std::string n = "trans_layers." + std::to_string(s);
            const int a = conv(src[s], n + ".0", bias, "", 256, 3, 1, 1, 1, 1);
            ops.back().chain_tag = s == 2;
            t[s] = conv(a, n + ".2", bias, "", 256, 3, 1, 1, 1, 0);
            ops.back().chain_tag = s == 2;
        }
        cur_lane = 0;
        for (int i = 0; i < 3; ++i) {
            const int lvl = 2 - i;
            const int u = conv_transpose2(x, "up_layers." + std::to_string(i), bias, 256, t[lvl], 1);
            ops.back().chain_tag = i == 0;
            x = conv(u, "latent_layers." + std::to_string(i), bias, "", 256, 3, 1, 1, 1, 1);
            ops.back().chain_tag = i == 0;
            odm[lvl] = x;
        }
    }

    void drn_heads(const int src[4], const int odm[4], bool bias)
    {
        int off_t[4];
        cur_lane = 3;                       // ARM heads + offset convs: off the critical path
        // the four ARM loc heads first, then the four offset convs back to back: consecutive OP_OFFSET ops of a lane run as ONE
        // launch (round 6: each was a 16-25 us launch of its own between two heads; same arithmetic per output)
        for (int s = 0; s < 4; ++s) conv(src[s], "arm_loc." + std::to_string(s), bias, "", 12, 3, 1, 1, 1, 0, -1, OUT_ARM_LOC, s);
        for (int s = 0; s < 4; ++s) {
            const std::string ss = std::to_string(s);
            off_t[s] = offset_conv(s, fm[s], fm[s], "offset." + ss, cfg.multihead ? "offset2." + ss : "", bias,
                                   cfg.def_groups * 18, cfg.def_groups * 50, OUT_ARM_LOC);
        }
        for (int s = 0; s < 4; ++s) offsets_out(s, off_t[s], cfg.def_groups * 18);
        cur_lane = 0;
        for (int s = 0; s < 4; ++s) {
            const std::string ss = std::to_string(s);
            deform_heads(odm[s], off_t[s], s, cfg.def_groups, "odm_loc." + ss, "odm_conf." + ss,
                         cfg.multihead ? "odm_loc_2." + ss : "", cfg.multihead ? "odm_conf_2." + ss : "",
                         cfg.def_groups * 18, OUT_ODM_LOC);
        }
        if (cfg.test_phase) softmax_op();
    }

    int build_drn_vgg()
    {
        const int S = cfg.size;
        set_pyramid(S / 8);
        int c43, c53, fc7;
        vgg_trunk(S, cfg.bn != 0, cfg.c7_channel, c43, c53, fc7);
        int src[4], odm[4];
        src[0] = l2norm(c43, "L2Norm_4_3");
        src[1] = l2norm(c53, "L2Norm_5_3");
        src[2] = fc7;
        src[3] = vgg_extras(fc7);
        tcb(src, true, odm);
        drn_heads(src, odm, true);
        return TDRN_OK;
    }

    // extras of the VGG variants (dualrefinedet_vggbn.py:36-45, refinedet_vgg.py:47-56, ssd4scale_vgg.py:25-34)
    int vgg_extras(int fc7)
    {
        const int e = cfg.bn ? conv(fc7, "extras.0", true, "extras.1", 256, 1, 1, 0, 1, 1) : conv(fc7, "extras.0", true, "", 256, 1, 1, 0, 1, 1);
        ops.back().chain_tag = true;
        const int x = cfg.bn ? conv(e, "extras.3", true, "extras.4", 512, 3, 2, 1, 1, 1) : conv(e, "extras.2", true, "", 512, 3, 2, 1, 1, 1);
        ops.back().chain_tag = true;
        return x;
    }

    // RefineDet-VGG: same trunk / TCB, plain (non-deformable) ODM heads (model/refinedet_vgg.py:27-219).
    // multihead sums a 3x3 and a 5x5 conv of the same input (:179-182): packed as ONE 5x5 conv whose
    // centre taps carry the 3x3 weights (biases added).
    int build_refinedet_vgg()
    {
        const int S = cfg.size;
        set_pyramid(S / 8);
        int c43, c53, fc7;
        vgg_trunk(S, cfg.bn != 0, cfg.c7_channel, c43, c53, fc7);
        int src[4], odm[4];
        src[0] = l2norm(c43, "L2Norm_4_3");
        src[1] = l2norm(c53, "L2Norm_5_3");
        src[2] = fc7;
        src[3] = vgg_extras(fc7);
        if (cfg.use_refine) {
            cur_lane = 3;
            for (int s = 0; s < 4; ++s) conv(src[s], "arm_loc." + std::to_string(s), true, "", 12, 3, 1, 1, 1, 0, -1, OUT_ARM_LOC, s);
            cur_lane = 0;
        }
        tcb(src, true, odm);
        const int nc3 = 3 * cfg.num_classes;
        for (int s = 0; s < 4; ++s) {
            const std::string ss = std::to_string(s);
            if (cfg.multihead) {
                conv(odm[s], "odm_loc_2." + ss, true, "", 12, 5, 1, 2, 1, 0, -1, OUT_ODM_LOC, s, "odm_loc." + ss, 3);
                conv(odm[s], "odm_conf_2." + ss, true, "", nc3, 5, 1, 2, 1, 0, -1, OUT_CONF, s, "odm_conf." + ss, 3);
            } else {
                conv(odm[s], "odm_loc." + ss, true, "", 12, 3, 1, 1, 1, 0, -1, OUT_ODM_LOC, s);
                conv(odm[s], "odm_conf." + ss, true, "", nc3, 3, 1, 1, 1, 0, -1, OUT_CONF, s);
            }
        }
        if (cfg.test_phase) softmax_op();
        return TDRN_OK;
    }

    // MobileNet-v1 trunk shared by dualrefinedet_mobilenet.py:19-48 and ssd4scale_mobile.py:20-50
    void mobilenet_trunk(int S, int c7, bool extras_bias, int src_raw[4])
    {
        static const int couts[] = {64, 128, 128, 256, 256, 512, 512, 512, 512, 512, 512, 1024, 0};
        static const int strides[] = {1, 2, 1, 1, 1, 2, 1, 1, 1, 1, 1, 2, 1};
        int x = first_conv("backbone.0.0", false, "backbone.0.1", 32, 2, S);
        for (int i = 0; i < 13; ++i) {
            x = conv_dw(x, "backbone." + std::to_string(i + 1), i == 12 ? c7 : couts[i], strides[i]);
            if (i + 1 == 11) src_raw[0] = x;
        }
        src_raw[1] = x;
        for (int k = 0; k < 2; ++k) {
            const std::string n = "extras." + std::to_string(k);
            x = conv(x, n + ".0", extras_bias, n + ".1", 256, 1, 1, 0, 1, 1);
            x = conv_dw(x, n + ".3", 512, 2);
            src_raw[2 + k] = x;
        }
    }

    int build_drn_mobilenet()
    {
        set_pyramid(cfg.size / 8);
        int raw[4], src[4], odm[4];
        mobilenet_trunk(cfg.size, 1024, true, raw);
        src[0] = l2norm(raw[0], "L2Norm_4_3");
        src[1] = l2norm(raw[1], "L2Norm_5_3");
        src[2] = raw[2];
        src[3] = raw[3];
        tcb(src, false, odm);
        drn_heads(src, odm, false);
        return TDRN_OK;
    }

    int build_ssd4scale(bool mobile)
    {
        set_pyramid(cfg.size / 8);
        int src[4];
        if (mobile) {
            int raw[4];
            mobilenet_trunk(cfg.size, cfg.c7_channel, true, raw);
            src[0] = l2norm(raw[0], "L2Norm_4_3");
            src[1] = l2norm(raw[1], "L2Norm_5_3");
            src[2] = raw[2];
            src[3] = raw[3];
        } else {
            int c43, c53, fc7;
            vgg_trunk(cfg.size, cfg.bn != 0, cfg.c7_channel, c43, c53, fc7);
            src[0] = l2norm(c43, "L2Norm_4_3");
            src[1] = l2norm(c53, "L2Norm_5_3");
            src[2] = fc7;
            src[3] = vgg_extras(fc7);
        }
        const int nc3 = 3 * cfg.num_classes;
        if (cfg.deform) {
            // all four levels' offsets first, then the four deformable heads back to back: consecutive OP_DEFORM ops run as ONE
            // launch (the gather kernel is latency-bound per workgroup -- 72 dependent K steps with 8 groups -- so four launches
            // cost four times the one: 4 x 230-330 us -> 330 us at config #5's batch, profiles/r04_cfg5)
            int ot[4], rl[4];
            for (int s = 0; s < 4; ++s) rl[s] = ref_loc_in(s, fm[s], fm[s]);
            for (int s = 0; s < 4; ++s) ot[s] = offset_conv(s, fm[s], fm[s], "offset." + std::to_string(s), "", true, 8 * 18, 0, -1, rl[s]);     // (one launch)
            for (int s = 0; s < 4; ++s) offsets_out(s, ot[s], 8 * 18);
            for (int s = 0; s < 4; ++s) {
                const std::string ss = std::to_string(s);
                deform_heads(src[s], ot[s], s, 8, "arm_loc." + ss, "arm_conf." + ss, "", "", 0, OUT_ARM_LOC);
            }
        }
        for (int s = 0; s < 4 && !cfg.deform; ++s) {
            const std::string ss = std::to_string(s);
            conv(src[s], "arm_loc." + ss, true, "", 12, 3, 1, 1, 1, 0, -1, OUT_ARM_LOC, s);
            conv(src[s], "arm_conf." + ss, true, "", nc3, 3, 1, 1, 1, 0, -1, OUT_CONF, s);
            loc_maps_out(s);
        }
        if (cfg.test_phase) softmax_op();
        return TDRN_OK;
    }

    int build()
    {
        es = dtype_bytes(cfg.dtype);
        kdisable = ((cfg.plan_flags & TDRN_PLAN_NO_CONV_PP) ? 1 : 0) | ((cfg.plan_flags & TDRN_PLAN_NO_PP_SK) ? 2 : 0) |
                   ((cfg.plan_flags & TDRN_PLAN_NO_CONV_PATCH) ? 4 : 0) | ((cfg.plan_flags & TDRN_PLAN_NO_PW1X1) ? 8 : 0) |
                   ((cfg.plan_flags & TDRN_PLAN_NO_DW_SLIDE) ? 16 : 0) | ((cfg.plan_flags & TDRN_PLAN_DW_SLIDE_ALL) ? 32 : 0) |
                   ((cfg.plan_flags & TDRN_PLAN_NO_CONV_WS) ? 64 : 0) | ((cfg.plan_flags & TDRN_PLAN_NO_YGEMM_V2) ? 128 : 0) |
                   ((cfg.plan_flags & TDRN_PLAN_NO_HEAD3X3) ? 256 : 0) | ((cfg.plan_flags & TDRN_PLAN_TS_ONE_RANGE) ? 512 : 0) |
                   ((cfg.plan_flags & TDRN_PLAN_NO_PATCH_TAIL) ? 1024 : 0);
        fault_handoff = (cfg.plan_flags & TDRN_PLAN_FAULT_HANDOFF) ? 1 : 0;
        // build_net() only constructs 320 / 512 nets, but they are fully convolutional and multi_eval.py runs them at
        // 192 ... 1216 (every tested size is a multiple of 64, so all four pyramid levels are exact)
        if (cfg.size < 128 || cfg.size > 1280 || cfg.size % 64 != 0) return TDRN_E_ARG;
        if (cfg.num_classes < 2 || cfg.num_classes > 21 * 4) return TDRN_E_ARG;
        if (cfg.dtype < 0 || cfg.dtype > 2) return TDRN_E_ARG;
        if (cfg.def_groups < 1) return TDRN_E_ARG;
        int rc;
        switch (cfg.model) {
            case TDRN_DRN_VGGBN: rc = build_drn_vgg(); break;
            case TDRN_DRN_MOBILENET: rc = build_drn_mobilenet(); break;
            case TDRN_SSD4SCALE_MOBILE: rc = build_ssd4scale(true); break;
            case TDRN_SSD4SCALE_VGG: rc = build_ssd4scale(false); break;
            case TDRN_REFINEDET_VGG: rc = build_refinedet_vgg(); break;
            default: return TDRN_E_UNSUPPORTED;
        }
        if (rc != TDRN_OK) return rc;
        if (plan_error != TDRN_OK) return plan_error;
        // The fp32 (3, S, S) copy of uint8 frames (plans whose first conv reads fp32: every one but the conv3x3_ws route) costs no
        // workspace and no tensor index of a layer: it is appended LAST and ALIASED with the first layer output behind the second op
        // that is large enough -- that tensor is dead while ops 0 / 1, the only readers of the copy, run (the engine's forwards are
        // stream-ordered, a second step in flight has its own workspace).  Round-5 advisor finding: 1.2-3 MB per frame and engine
        // clone were allocated in front of every other tensor for a fallback most callers never take.
        if (!ops.empty() && ops[0].kind == OP_FIRST) {
            const size_t need = align_up((size_t)3 * cfg.size * cfg.size * 4, 256);
            auto touched_early = [&](int t) {
                for (size_t i = 0; i < 2 && i < ops.size(); ++i)
                    if (ops[i].in == t || ops[i].out == t || ops[i].res == t || ops[i].pool_t == t || ops[i].off_t == t) return true;
                return false;
            };
            int victim = -1;
            for (size_t i = 2; i < ops.size() && victim < 0; ++i) {
                // (a main-lane layer of the trunk: it runs behind ops 0 / 1 in stream order; an op that does not depend on the trunk --
                // the TRN nets' ref_loc conversions on a side lane -- could otherwise write its output while the copy is still being read)
                if (ops[i].lane != 0 || !(ops[i].kind == OP_CONV || ops[i].kind == OP_DW || ops[i].kind == OP_POOL) || ops[i].in < 0) continue;
                for (int t : {ops[i].out, ops[i].pool_t}) {
                    if (t < 0 || victim >= 0 || touched_early(t)) continue;
                    const Tensor &v = tensors[t];
                    if (align_up((size_t)v.Cpad * v.H * v.W * (v.f32 ? 4 : es), 256) >= need) victim = t;
                }
            }
            if (victim >= 0) {
                Tensor t;
                t.C = 3; t.H = cfg.size; t.W = cfg.size; t.f32 = true; t.Cpad = 3; t.off = tensors[victim].off;
                tensors.push_back(t);
                x_t = (int)tensors.size() - 1;
            } else {
                x_t = T(3, cfg.size, cfg.size, true);
            }
        }
        // L2Norm of conv4_3 / conv5_3 right behind its producer and on a side lane: it is HBM-bound and needs no
        // LDS, so it runs under the next (LDS-filling) conv of the trunk, and the lateral TCB convs and ARM heads
        // that read it can start while conv5 / fc6 / fc7 -- which leave CUs idle -- are still running, instead of
        // queueing behind fc7 on the main lane.
        {
            const char *le = getenv("TDRN_L2_EARLY");
            int side = 1;
            for (size_t i = 0; i < ops.size() && !(le && atoi(le) == 0); ++i) {
                if (ops[i].kind != OP_L2NORM) continue;
                size_t prod = i;
                for (size_t j = 0; j < i; ++j)
                    if (ops[j].out == ops[i].in || ops[j].pool_t == ops[i].in) prod = j;
                if (prod == i) continue;
                Op o = ops[i];
                o.lane = side;
                side = side == 1 ? 2 : 1;
                ops.erase(ops.begin() + (long)i);
                ops.insert(ops.begin() + (long)prod + 1, o);
            }
        }
        // first conv fused into the loader of the conv behind it (conv3x3_patch.hip FUSE): 16-bit modes, stride 1, 64 channels,
        // 8x32 tiles, and nobody else reads the first conv's output (TDRN_FUSE_FIRST=0 keeps the two launches)
        {
            const char *fe = getenv("TDRN_FUSE_FIRST");
            const bool fuse_on = fe ? atoi(fe) != 0 : !(cfg.plan_flags & TDRN_PLAN_NO_FUSE_FIRST);
            fuse_first = -1;
            if (fuse_on && cfg.dtype != TDRN_F32 && conv_patch_enabled() && !(kdisable & 4) && ops.size() > 1 && ops[0].kind == OP_FIRST &&
                ops[1].kind == OP_CONV && ops[0].stride == 1 && tensors[ops[0].out].Cpad == 64) {
                const Op &c = ops[1];
                const Tensor &ti = tensors[ops[0].out];
                int readers = 0;
                for (const Op &o : ops) readers += (o.in == ops[0].out) + (o.res == ops[0].out);
                if (c.in == ops[0].out && readers == 1 && c.k == 3 && c.stride == 1 && c.pad == 1 && c.dil == 1 && c.phases == 1 && c.res < 0 &&
                    c.out_kind == OUT_TENSOR && c.Npad == 64 && c.Cin == 64 && ti.W % 32 == 0 && ti.H % 16 == 0 && ti.H == ti.W && c.lane == 0)
                    fuse_first = 1;
                if (fuse_first >= 0) {                   // the fused launch carries both layers' algorithmic work
                    ops[1].flops += ops[0].flops;
                    ops[1].bytes += 3.0 * ops[0].hw * ops[0].hw * 4 - (double)ti.H * ti.W * ti.Cpad * es;
                    ops[0].flops = 0; ops[0].bytes = 0;
                }
            }
        }
        if (cfg.plan_flags & TDRN_PLAN_NO_LATE_SIDE) late_side = 0;
        if (const char *e = getenv("TDRN_LATE_SIDE")) late_side = atoi(e);
        if (const char *e = getenv("TDRN_SIDE_GRID")) side_grid = atoi(e);
        if (const char *e = getenv("TDRN_MAIN_GRID")) main_grid = atoi(e);      // experiment: cap the persistent grids of the main lane (CUs left to the other step in flight)
        if (const char *rb = getenv("TDRN_SPLITK_REF")) splitk_ref_batch = atoi(rb) > 0 ? atoi(rb) : 32;
        // split-K per layer from its geometry only (at the reference batch, 32 unless TDRN_SPLITK_REF says otherwise), so that a frame's arithmetic never
        // depends on the batch it travels in; the partial slabs live in a per-lane region of the workspace
        {
            size_t lane_bytes[kLanes] = {0, 0, 0, 0};
            for (Op &o : ops) {
                if (o.kind == OP_CONV && o.pool_t >= 0 && conv_patch_enabled()) o.stat = ST_CONV3;
                if (o.kind != OP_CONV || o.pool_t >= 0) continue;
                const Tensor &ti = tensors[o.in];
                ConvArgs a;
                a.B = splitk_ref_batch; a.H = ti.H; a.W = ti.W; a.Cin = o.Cin; a.Ho = o.hw >> 16; a.Wo = o.hw & 0xffff;
                a.Cout = o.Cout; a.Npad = o.Npad; a.kh = a.kw = o.k; a.stride = o.stride; a.pad = o.pad; a.dil = o.dil;
                a.phases = o.phases; a.dtype = cfg.dtype; a.out_f32 = o.out_kind != OUT_TENSOR;
                a.kdisable = kdisable;
                o.splitk = conv_splitk_choice(a);
                a.o_cs = o.out_kind == OUT_TENSOR ? tensors[o.out].Cpad : 0;
                a.o_rs = (long long)a.Wo * a.o_cs; a.o_bs = (long long)a.Ho * a.Wo * a.o_cs;
                a.res = o.res >= 0 ? (const void *)1 : nullptr;
                if (o.splitk == 1 && conv_patch_enabled() && patch_conv_supported(a) && a.H * a.W >= conv_patch_enabled() * 400)
                    o.stat = ST_CONV3;
                if (o.chain_tag && !(o.out_kind == OUT_TENSOR && conv_chain_supported(a))) o.chain_tag = false;
            }
            // The chain launch: tagged layers whose inputs are chain members or exist before the first member starts (a layer
            // the patch kernels take at this frame size, and everything behind it, stays an ordinary launch).  Members move to
            // the main lane and get their own split-K slabs (stages overlap inside the launch).
            chain_ops.clear();
            {
                const char *ce = getenv("TDRN_CHAIN");
                const bool chain_on = ce ? atoi(ce) != 0 : (cfg.plan_flags & TDRN_PLAN_CHAIN) != 0;   // opt-in: it lost (conv_igemm.hip)
                int first = -1;
                for (size_t i = 0; i < ops.size() && chain_on; ++i) {
                    Op &o = ops[i];
                    if (o.kind != OP_CONV || !o.chain_tag || (int)chain_ops.size() == conv_chain_max_layers()) continue;
                    bool ok = true;
                    for (int t : {o.in, o.res}) {
                        if (t < 0) continue;
                        int prod = -1;
                        for (size_t j = 0; j < ops.size(); ++j)
                            if (ops[j].out == t || ops[j].pool_t == t) prod = (int)j;
                        const bool member = prod >= 0 && ops[prod].chain >= 0;
                        if (!member && first >= 0 && prod > first) ok = false;
                    }
                    if (!ok) continue;
                    if (first < 0) first = (int)i;
                    o.chain = (int)chain_ops.size();
                    chain_ops.push_back((int)i);
                }
                if (chain_ops.size() < 3) {              // not worth a queue
                    for (int i : chain_ops) ops[i].chain = -1;
                    chain_ops.clear();
                }
                // Queue order = dependency level (a stage's tasks wait only for EARLIER stages), ties in plan order: the
                // independent lateral convs of the level below then sit between the stages of the serial chain and fill the
                // workgroups that would otherwise spin on the chain's next dependency.
                std::vector<int> level(chain_ops.size(), 0);
                for (size_t k = 0; k < chain_ops.size(); ++k)
                    for (int t : {ops[chain_ops[k]].in, ops[chain_ops[k]].res})
                        for (size_t j = 0; j < k; ++j)
                            if (t >= 0 && ops[chain_ops[j]].out == t && level[j] + 1 > level[k]) level[k] = level[j] + 1;
                std::vector<int> order(chain_ops.size());
                for (size_t k = 0; k < order.size(); ++k) order[k] = (int)k;
                std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return level[a] < level[b]; });
                std::vector<int> sorted;
                for (int k : order) sorted.push_back(chain_ops[k]);
                chain_ops = sorted;
                for (size_t k = 0; k < chain_ops.size(); ++k) ops[chain_ops[k]].chain = (int)k;
            }
            size_t chain_bytes = 0;
            for (Op &o : ops) {
                if (o.kind != OP_CONV || o.pool_t >= 0 || o.splitk <= 1) continue;
                const size_t per_sample = align_up((size_t)o.splitk * o.phases * (o.hw >> 16) * (o.hw & 0xffff) * o.Npad * sizeof(float), 256);
                if (o.chain >= 0) {
                    o.chain_partial = chain_bytes;
                    chain_bytes += per_sample;
                } else if (per_sample > lane_bytes[o.lane]) {
                    lane_bytes[o.lane] = per_sample;
                }
            }
            for (int i : chain_ops) ops[i].lane = 0;
            chain_partial_off = ws_per_sample;
            ws_per_sample += chain_bytes;
            if (const char *pd = getenv("TDRN_PLAN_DUMP")) {
                if (atoi(pd))
                    for (const Op &o : ops)
                        if (o.kind == OP_CONV)
                            fprintf(stderr, "plan: %-28s lane %d  %dx%d k%d s%d d%d  Cin %4d Cout %4d  splitk %d  %s  chain %d\n", o.w.c_str(), o.lane,
                                    o.hw >> 16, o.hw & 0xffff, o.k, o.stride, o.dil, o.Cin, o.Cout, o.splitk, o.stat == ST_CONV3 ? "patch" : "igemm", o.chain);
            }
            for (int l = 0; l < kLanes; ++l) {
                splitk_off[l] = ws_per_sample;
                ws_per_sample += lane_bytes[l];
            }
        }
        // OPT-IN (TDRN_PLAN_DWPW; it measured slower than the two launches, dwpw.hip): conv_dw blocks as ONE launch:
        // a depthwise op directly followed by its pointwise conv, which is the only
        // reader of the depthwise output; decided from the geometry (the batch-dependent 4-GiB limit is re-checked per forward,
        // which then falls back to the two launches: the depthwise tensor keeps its place in the workspace)
        if (cfg.dtype != TDRN_F32 && ((cfg.plan_flags & TDRN_PLAN_DWPW) || dwpw_enabled() > 1))
            for (size_t i = 0; i + 1 < ops.size(); ++i) {
                Op &d = ops[i];
                Op &c = ops[i + 1];
                if (d.kind != OP_DW || c.kind != OP_CONV || c.in != d.out || c.k != 1 || c.stride != 1 || c.pad != 0 || c.phases != 1 || c.res >= 0 ||
                    c.out_kind != OUT_TENSOR || c.splitk != 1 || c.lane != d.lane || c.pool_t >= 0 || c.chain >= 0) continue;
                int readers = 0;
                for (const Op &o : ops) readers += (o.in == d.out) + (o.res == d.out);
                if (readers != 1) continue;
                const Tensor &ti = tensors[d.in];
                DwPwArgs a;
                a.B = 1; a.H = ti.H; a.W = ti.W; a.Cin = c.Cin; a.Cout = c.Cout; a.Npad = c.Npad; a.Cs = tensors[c.out].Cpad;
                a.stride = d.stride; a.dtype = cfg.dtype;
                if (ti.Cpad != c.Cin || !dwpw_supported(a)) continue;
                d.fused_dw = 1; c.fused_dw = 1;
                d.stat = ST_DWPW;
                d.flops += c.flops;
                d.bytes = (double)ti.H * ti.W * ti.Cpad * es + (double)ti.H * ti.W * tensors[c.out].Cpad * es;
                c.flops = 0; c.bytes = 0;
            }
        // conv3x3_pp.hip's chained split needs a slab per workgroup; only launches on the main lane use it (one at a time)
        // Batch-independent tail of the workspace: [256 B: the chain launch's counters][1 KiB: the chained split's flag words]
        // [its slabs]; the first 1280 bytes are zeroed once per forward.
        ws_fixed = 0;
        pp_sk_planned = false;
        if (cfg.dtype != TDRN_F32)
            for (const Op &o : ops)
                if (o.kind == OP_CONV && o.stat == ST_CONV3 && o.lane == 0 && o.Cin >= 256 && o.Npad % 256 == 0) pp_sk_planned = true;
        if (pp_sk_planned || !chain_ops.empty()) ws_fixed = kTailCtl + (pp_sk_planned ? align_up(conv_pp_sk_bytes(), 256) : 1024);
        const char *e = getenv("TDRN_STREAMS");
        if (cfg.plan_flags & TDRN_PLAN_ONE_STREAM) use_lanes = false;
        if (e) use_lanes = atoi(e) > 1;
        const char *ds = getenv("TDRN_DEFORM_SPLIT");
        if (ds && atoi(ds) == 0) deform_split = false;
        tensor_lane.assign(tensors.size(), 0);
        tensor_shared.assign(tensors.size(), 0);
        for (const Op &o : ops) {
            if (o.out >= 0) tensor_lane[o.out] = o.lane;
            if (o.pool_t >= 0) tensor_lane[o.pool_t] = o.lane;
        }
        for (const Op &o : ops)
            for (int t : {o.in, o.res, o.off_t})
                if (t >= 0 && tensor_lane[t] != o.lane) tensor_shared[t] = 1;
        return TDRN_OK;
    }

    int init_lanes()
    {
        if (lanes_ready) return TDRN_OK;
        for (int i = 0; i < kLanes - 1; ++i) {
            TDRN_TRY(pool::get_stream(dev, &side[i]));
            TDRN_TRY(pool::get_event(dev, &ev_join[i]));
        }
        TDRN_TRY(pool::get_event(dev, &ev_fork));
        TDRN_TRY(pool::get_event(dev, &ev_zero));
        TDRN_TRY(pool::get_event(dev, &ev_skz));
        tensor_ev.assign(tensors.size(), nullptr);
        for (size_t t = 0; t < tensors.size(); ++t)
            if (tensor_shared[t]) TDRN_TRY(pool::get_event(dev, &tensor_ev[t]));
        lanes_ready = true;
        return TDRN_OK;
    }

    // ---- weight packing -----------------------------------------------------------------------
    const std::vector<float> *get(const std::string &name) const
    {
        auto it = staged.find(name);
        return it == staged.end() ? nullptr : &it->second;
    }
    void put_elem(char *dst, size_t idx, float v) const
    {
        if (cfg.dtype == TDRN_F32) ((float *)dst)[idx] = v;
        else if (cfg.dtype == TDRN_BF16) ((unsigned short *)dst)[idx] = host_f32_to_bf16(v);
        else ((unsigned short *)dst)[idx] = host_f32_to_f16(v);
    }
    // y = scale*conv + shift  with BatchNorm (eps 1e-5, running stats) folded in double
    int fold(const Op &o, int Cout, std::vector<double> &scale, std::vector<double> &shift) const
    {
        scale.assign(Cout, 1.0);
        shift.assign(Cout, 0.0);
        if (!o.b.empty() && o.kind != OP_DEFORM) {
            const auto *b = get(o.b + ".bias");
            if (!b) return TDRN_E_PARAM;
            for (int c = 0; c < Cout; ++c) shift[c] = (*b)[c];
        }
        if (!o.bn.empty()) {
            const auto *g = get(o.bn + ".weight"), *be = get(o.bn + ".bias"), *mu = get(o.bn + ".running_mean"),
                       *var = get(o.bn + ".running_var");
            if (!g || !be || !mu || !var) return TDRN_E_PARAM;
            for (int c = 0; c < Cout; ++c) {
                const double s = (double)(*g)[c] / std::sqrt((double)(*var)[c] + 1e-5);
                scale[c] = s;
                shift[c] = (shift[c] - (double)(*mu)[c]) * s + (double)(*be)[c];
            }
        }
        return TDRN_OK;
    }

    int pack(std::vector<char> &host) const
    {
        host.assign(blob_bytes, 0);
        for (const Op &o : ops) {
            std::vector<double> sc, sh;
            switch (o.kind) {
                case OP_FIRST: {
                    const auto *w = get(o.w + ".weight");
                    if (!w) return TDRN_E_PARAM;
                    TDRN_TRY(fold(o, o.Cout, sc, sh));
                    float *dw = (float *)(host.data() + o.w_off), *db = (float *)(host.data() + o.b_off);
                    for (int c = 0; c < o.Cout; ++c) {
                        for (int k = 0; k < 27; ++k) dw[c * 27 + k] = (float)((double)(*w)[(size_t)c * 27 + k] * sc[c]);
                        db[c] = (float)sh[c];
                    }
                    break;
                }
                case OP_CONV: {
                    const auto *w = get(o.w + ".weight");
                    if (!w) return TDRN_E_PARAM;
                    const Tensor &ti = tensors[o.in];
                    const int Creal = ti.C, Cin = o.Cin, k = o.k, taps = k * k;
                    char *dw = host.data() + o.w_off;
                    float *db = (float *)(host.data() + o.b_off);
                    if (o.phases == 4) {
                        // ConvTranspose2d weight (Cin, Cout, 2, 2): slab(i,j)[co][ci] = W[ci][co][i][j]
                        const int Cout = (int)params[param_index.at(o.w + ".weight")].shape[1];
                        for (int ph = 0; ph < 4; ++ph)
                            for (int co = 0; co < Cout; ++co)
                                for (int ci = 0; ci < Creal; ++ci)
                                    put_elem(dw, ((size_t)ph * o.Npad + co) * Cin + ci,
                                             (*w)[(((size_t)ci * Cout + co) * 2 + (ph >> 1)) * 2 + (ph & 1)]);
                        if (!o.b.empty()) {
                            const auto *b = get(o.b + ".bias");
                            if (!b) return TDRN_E_PARAM;
                            for (int co = 0; co < Cout; ++co) db[co] = (*b)[co];
                        }
                        break;
                    }
                    const int Cout = (int)params[param_index.at(o.w + ".weight")].shape[0];
                    TDRN_TRY(fold(o, Cout, sc, sh));
                    for (int co = 0; co < Cout; ++co) {
                        for (int t = 0; t < taps; ++t)
                            for (int ci = 0; ci < Creal; ++ci)
                                put_elem(dw, ((size_t)co * taps + t) * Cin + ci,
                                         (float)((double)(*w)[((size_t)co * Creal + ci) * taps + t] * sc[co]));
                        db[co] = (float)sh[co];
                    }
                    if (!o.w2.empty()) {   // merge a centred k2 x k2 conv (same stride/dilation) into the k x k taps
                        const auto *w2 = get(o.w2 + ".weight");
                        if (!w2) return TDRN_E_PARAM;
                        const int k2 = o.k2, d = (k - k2) / 2;
                        std::vector<float> merged((size_t)Cout * taps * Creal, 0.f);
                        for (int co = 0; co < Cout; ++co)
                            for (int ci = 0; ci < Creal; ++ci) {
                                for (int t = 0; t < taps; ++t)
                                    merged[((size_t)co * taps + t) * Creal + ci] = (*w)[((size_t)co * Creal + ci) * taps + t];
                                for (int r = 0; r < k2; ++r)
                                    for (int q = 0; q < k2; ++q)
                                        merged[((size_t)co * taps + (r + d) * k + (q + d)) * Creal + ci] +=
                                            (*w2)[((size_t)co * Creal + ci) * k2 * k2 + r * k2 + q];
                            }
                        for (int co = 0; co < Cout; ++co)
                            for (int t = 0; t < taps; ++t)
                                for (int ci = 0; ci < Creal; ++ci)
                                    put_elem(dw, ((size_t)co * taps + t) * Cin + ci, merged[((size_t)co * taps + t) * Creal + ci]);
                        if (!o.b2.empty()) {
                            const auto *b2 = get(o.b2 + ".bias");
                            if (!b2) return TDRN_E_PARAM;
                            for (int co = 0; co < Cout; ++co) db[co] += (*b2)[co];
                        }
                    }
                    break;
                }
                case OP_L2NORM: {
                    const auto *w = get(o.w + ".weight");
                    if (!w) return TDRN_E_PARAM;
                    memcpy(host.data() + o.w_off, w->data(), w->size() * 4);
                    break;
                }
                case OP_DW: {
                    const auto *w = get(o.w + ".weight");
                    if (!w) return TDRN_E_PARAM;
                    const Tensor &ti = tensors[o.in];
                    TDRN_TRY(fold(o, ti.C, sc, sh));
                    float *dw = (float *)(host.data() + o.w_off), *db = (float *)(host.data() + o.b_off);
                    for (int c = 0; c < ti.C; ++c) {
                        for (int t = 0; t < 9; ++t) dw[(size_t)t * ti.Cpad + c] = (float)((double)(*w)[(size_t)c * 9 + t] * sc[c]);
                        db[c] = (float)sh[c];
                    }
                    break;
                }
                case OP_OFFSET: {
                    float *dw = (float *)(host.data() + o.w_off), *db = (float *)(host.data() + o.b_off);
                    const std::string *names[2] = {&o.w, &o.w2};
                    const std::string *bnames[2] = {&o.b, &o.b2};
                    int row = 0;
                    for (int i = 0; i < 2; ++i) {
                        if (names[i]->empty()) continue;
                        const auto *w = get(*names[i] + ".weight");
                        if (!w) return TDRN_E_PARAM;
                        const int n = (int)(w->size() / 12);
                        memcpy(dw + (size_t)row * 12, w->data(), w->size() * 4);
                        if (!bnames[i]->empty()) {
                            const auto *b = get(*bnames[i] + ".bias");
                            if (!b) return TDRN_E_PARAM;
                            memcpy(db + row, b->data(), b->size() * 4);
                        }
                        row += n;
                    }
                    break;
                }
                case OP_DEFORM: {
                    const Tensor &ti = tensors[o.in];
                    const int nc3 = 3 * cfg.num_classes;
                    if (o.y_t >= 0) {        // rows (tap, column) of the 1x1 GEMM: tap-major over the branches, 80 columns per tap, three taps per 256-row slice (deform_y_col)
                        char *dt = host.data() + o.wt_off;
                        int tap0 = 0;
                        for (int br = 0; br < o.n_branches; ++br) {
                            const std::string &ln = br ? o.w2 : o.w, &cn = br ? o.b2 : o.b;
                            const auto *wl = get(ln + ".weight"), *wc = get(cn + ".weight");
                            if (!wl || !wc) return TDRN_E_PARAM;
                            const int k = br ? o.k2 : o.k, taps = k * k;
                            for (int t = 0; t < taps; ++t)
                                for (int co = 0; co < 12 + nc3; ++co) {
                                    const std::vector<float> &src = co < 12 ? *wl : *wc;
                                    const int cs = co < 12 ? co : co - 12;
                                    for (int ci = 0; ci < ti.C; ++ci)
                                        put_elem(dt, ((size_t)(co / 80) * o.y_cols + deform_y_col(tap0 + t) + co % 80) * o.Cin + ci, src[((size_t)cs * ti.C + ci) * taps + t]);
                                }
                            tap0 += taps;
                        }
                    }
                    for (int br = 0; br < o.n_branches; ++br) {
                        const std::string &ln = br ? o.w2 : o.w, &cn = br ? o.b2 : o.b;
                        const auto *wl = get(ln + ".weight"), *wc = get(cn + ".weight");
                        if (!wl || !wc) return TDRN_E_PARAM;
                        const int k = br ? o.k2 : o.k, taps = k * k;
                        char *dw = host.data() + (br ? o.w2_off : o.w_off);
                        for (int co = 0; co < 12 + nc3; ++co) {
                            const std::vector<float> &src = co < 12 ? *wl : *wc;
                            const int cs = co < 12 ? co : co - 12;
                            for (int t = 0; t < taps; ++t)
                                for (int ci = 0; ci < ti.C; ++ci)
                                    put_elem(dw, ((size_t)co * taps + t) * o.Cin + ci, src[((size_t)cs * ti.C + ci) * taps + t]);
                        }
                    }
                    break;
                }
                default: break;
            }
        }
        return TDRN_OK;
    }

    // ---- forward ------------------------------------------------------------------------------
    char *tptr(void *ws, int id, int B) const { return (char *)ws + tensors[id].off * (size_t)B; }

    int first_op = 0;                    // tdrn_net_forward_from: ops below this index are assumed done (analysis only)
    int forward(const void *blob, void *ws, size_t ws_bytes, const tdrn_net_io *io, hipStream_t s0)
    {
        if (!weights_ready) return TDRN_E_STATE;
        if (!blob || !ws || !io || io->batch <= 0) return TDRN_E_ARG;
        // the batch: fp32 (B,3,S,S) in io->x, or uint8 planes + per-plane mean (tdrn_net_io.reserved[3]); `xin` = the fp32 tensor the first
        // conv reads, null until the uint8 planes have been converted (which only happens when no kernel reads them directly)
        const tdrn_u8_frames *u8 = (const tdrn_u8_frames *)io->reserved[3];
        if (u8 ? !u8->planes : !io->x) return TDRN_E_ARG;
        const float *xin = u8 ? nullptr : io->x;
        const int B = io->batch;
        if (ws_bytes < ws_per_sample * (size_t)B + ws_fixed) return TDRN_E_WORKSPACE;
        if (!io->conf) return TDRN_E_ARG;
        // a net's pooled streams / events belong to ONE device: the one current at its first forward
        {
            const int d = pool::cur_dev();
            if (dev < 0) dev = d;
            else if (dev != d) return TDRN_E_STATE;
        }
        // "never continue after an error": a forward whose device-side hand-off timed out makes the NEXT call fail
        // (no synchronisation here: the word is host memory the kernels store to)
        if (ws_fixed) {
            if (!status) TDRN_TRY(pool::get_status(&status));
            TDRN_TRY(check_status(nullptr));
        }
        const bool is_drn = cfg.model == TDRN_DRN_VGGBN || cfg.model == TDRN_DRN_MOBILENET || cfg.model == TDRN_REFINEDET_VGG;
        const bool has_arm = cfg.model != TDRN_REFINEDET_VGG || cfg.use_refine;
        if (is_drn && !io->odm_loc) return TDRN_E_ARG;
        if (has_arm && !io->arm_loc) return TDRN_E_ARG;
        const char *wb = (const char *)blob;
        const int C = cfg.num_classes;
        last_batch = B;
        size_t evi = 0;
        if (profile && ev.size() < 2 * ops.size()) {
            const size_t old = ev.size();
            ev.resize(2 * ops.size());
            for (size_t i = old; i < ev.size(); ++i) TDRN_TRY(pool::get_timing_event(dev, &ev[i]));
        }
        ev_stat.clear();
        ev_op.clear();
        // profile 1 runs single-stream (per-kernel durations without overlap); profile 2 keeps the production lanes, so a
        // launch's duration includes what the concurrent side-lane kernels take from it
        const bool lanes = use_lanes && profile != 1;
        bool lane_used[kLanes] = {true, false, false, false};
        if (lanes) {
            TDRN_TRY(init_lanes());
            TDRN_HIP_TRY(hipEventRecord(ev_fork, s0));
        }
        // Whatever way this function is left -- also on a mid-plan error -- the caller's stream is ordered after
        // everything already queued on the side lanes (they write the workspace and the outputs).
        struct Join {
            tdrn_net *n; hipStream_t s0; bool *used; bool on;
            int run()
            {
                if (!on) return TDRN_OK;
                on = false;
                int rc = TDRN_OK;
                for (int l = 1; l < kLanes; ++l)
                    if (used[l]) {
                        hipError_t e = hipEventRecord(n->ev_join[l - 1], n->side[l - 1]);
                        if (e == hipSuccess) e = hipStreamWaitEvent(s0, n->ev_join[l - 1], 0);
                        if (e != hipSuccess && rc == TDRN_OK) rc = (int)e;
                    }
                return rc;
            }
            ~Join() { (void)run(); }
        } join{this, s0, lane_used, lanes};
        // split two-branch deformable heads accumulate into zeroed outputs: zero them on a side stream at the very
        // start (under the first conv) instead of in front of the deform launch on the critical path
        bool zeroed_early = false;
        if (lanes && deform_split) {
            const Op *dsplit = nullptr;
            int n_deform_groups = 0;
            for (size_t k = 0; k < ops.size(); ++k)
                if (ops[k].kind == OP_DEFORM) {
                    // (the transform-then-sample path stores, no atomics; the gather kernel splits two-branch problems by branch and
                    // one-branch problems with an even number of deformable groups -- the TRN temporal heads -- by group halves)
                    if ((ops[k].n_branches == 2 || (ops[k].G >= 2 && ops[k].G % 2 == 0)) && ops[k].y_t < 0 && !dsplit) dsplit = &ops[k];
                    if (k == 0 || ops[k - 1].kind != OP_DEFORM) ++n_deform_groups;
                }
            if (dsplit && n_deform_groups == 1) {        // (one merged launch writes these outputs; nothing else does)
                float *locbase0 = dsplit->out_kind == OUT_ARM_LOC ? io->arm_loc : io->odm_loc;
                TDRN_HIP_TRY(hipStreamWaitEvent(side[0], ev_fork, 0));
                lane_used[1] = true;
                TDRN_HIP_TRY(hipMemsetAsync(locbase0, 0, (size_t)B * P * 4 * sizeof(float), side[0]));
                TDRN_HIP_TRY(hipMemsetAsync(io->conf, 0, (size_t)B * P * C * sizeof(float), side[0]));
                TDRN_HIP_TRY(hipEventRecord(ev_zero, side[0]));
                zeroed_early = true;
            }
        }
        // the chained split's flag words (conv3x3_pp.hip) are zeroed ONCE per forward, off the critical path; every launch
        // leaves them zero (the consumer of a flag resets it)
        bool skz_pending = false;
        char *const tail = (char *)ws + ws_per_sample * (size_t)B;       // [kTailCtl: chain counters][chained split: 1 KiB flags, slabs]
        if (ws_fixed) {
            if (lanes) {
                if (!lane_used[1]) {
                    TDRN_HIP_TRY(hipStreamWaitEvent(side[0], ev_fork, 0));
                    lane_used[1] = true;
                }
                TDRN_HIP_TRY(hipMemsetAsync(tail, 0, kTailCtl + 1024, side[0]));
                TDRN_HIP_TRY(hipEventRecord(ev_skz, side[0]));
                skz_pending = true;
            } else {
                TDRN_HIP_TRY(hipMemsetAsync(tail, 0, kTailCtl + 1024, s0));
            }
        }
        // ConvArgs of a conv op whose output is a workspace tensor or a head view
        auto conv_args = [&](const Op &o, ConvArgs &a) {
            const Tensor &ti = tensors[o.in];
            a.in = tptr(ws, o.in, B); a.w = wb + o.w_off; a.bias = (const float *)(wb + o.b_off); a.zero_page = wb;
            a.B = B; a.H = ti.H; a.W = ti.W; a.Cin = o.Cin; a.Ho = o.hw >> 16; a.Wo = o.hw & 0xffff;
            a.Cout = o.Cout; a.Npad = o.Npad; a.kh = a.kw = o.k; a.stride = o.stride; a.pad = o.pad; a.dil = o.dil;
            a.relu = o.relu; a.phases = o.phases; a.dtype = cfg.dtype;
            a.kdisable = kdisable; a.status = status; a.fault_handoff = fault_handoff;
            if (o.out_kind == OUT_TENSOR) {
                const Tensor &to = tensors[o.out];
                a.out = tptr(ws, o.out, B);
                if (o.res >= 0) a.res = tptr(ws, o.res, B);
                a.o_cs = to.Cpad;
                if (o.phases == 4) {
                    a.o_bs = (long long)to.H * to.W * to.Cpad; a.o_rs = 2ll * to.W * to.Cpad; a.o_cs = 2ll * to.Cpad;
                    a.o_pr = (long long)to.W * to.Cpad; a.o_pc = to.Cpad;
                } else {
                    a.o_bs = (long long)to.H * to.W * to.Cpad; a.o_rs = (long long)to.W * to.Cpad;
                }
            } else {
                const int per = o.out_kind == OUT_CONF ? 3 * C : 12;     // channels per pixel
                const int per_prior = o.out_kind == OUT_CONF ? C : 4;
                float *base = o.out_kind == OUT_ARM_LOC ? io->arm_loc : (o.out_kind == OUT_ODM_LOC ? io->odm_loc : io->conf);
                a.out = base; a.out_f32 = 1;
                a.o_base = (long long)scale_off[o.scale] * per_prior;
                a.o_bs = (long long)P * per_prior; a.o_rs = (long long)a.Wo * per; a.o_cs = per;
            }
            if (o.splitk > 1) {
                a.splitk = o.splitk;
                a.partial = o.chain >= 0 ? (char *)ws + (chain_partial_off + o.chain_partial) * (size_t)B : (char *)ws + splitk_off[o.lane] * (size_t)B;
            }
        };
        DeformArgs dargs[4];
        const void *ts_y[4] = {nullptr, nullptr, nullptr, nullptr};
        // Y layout of the transform-then-sample heads: tap-major [tap][pixel][80] when every level's transform runs on ygemm_k256
        // (which writes it), else the plain [pixel][columns] matrix of the generic GEMM
        static int ts_tap_env = -1;
        if (ts_tap_env < 0) { const char *e = getenv("TDRN_Y_TAP_MAJOR"); ts_tap_env = e ? atoi(e) : 1; }
        int ts_tap_major = ts_tap_env ? 1 : 0;
        for (const Op &d : ops)
            if (d.kind == OP_DEFORM && d.y_t >= 0 && !ygemm_supported(d.Cin, d.y_cols, cfg.dtype)) ts_tap_major = 0;
        int ts_cs[4] = {0, 0, 0, 0}, ts_op[4] = {-1, -1, -1, -1};
        int n_dargs = 0;
        OffsetProblem oq[4];                                // consecutive offset convs of one lane: one launch (layers.hip)
        int oq_op[4], n_oq = 0;
        bool dwpw_done = false;
        const bool reuse_offsets = cfg.deform && io->reserved[0] != nullptr;
        if (reuse_offsets && (offs_ws != ws || offs_batch != B)) return TDRN_E_STATE;
        // key-frame broadcast (tdrn_net_io.reserved[1]): ref_loc / the offset tensors hold Bk samples, sample b reads those of b % Bk
        int Bk = B;
        if (io->reserved[1]) {
            const long long kb = (long long)(intptr_t)io->reserved[1];
            if (!cfg.deform || kb < 1 || kb > B || B % kb) return TDRN_E_ARG;
            Bk = (int)kb;
        }
        if (reuse_offsets && offs_key_batch != Bk) return TDRN_E_STATE;
        // The reuse state is valid only once the offset launches of THIS forward have been enqueued and the forward returned OK
        // (an early error return, or a tdrn_net_forward_from that starts behind the offset ops, leaves it invalid: a later
        // reserved[0] call then gets TDRN_E_STATE instead of sampling stale or uninitialised offsets)
        int offset_ops_enqueued = 0, offset_ops_planned = 0;
        if (cfg.deform && !reuse_offsets) { offs_ws = nullptr; offs_batch = 0; offs_key_batch = 0; }
        for (const Op &d : ops) offset_ops_planned += d.kind == OP_OFFSET;
        for (size_t oi = 0; oi < ops.size(); ++oi) {
            const Op &o = ops[oi];
            bool skip = false;
            if (o.kind == OP_OFF_OUT && !io->offsets[o.scale]) skip = true;
            if (o.kind == OP_LOC_OUT && !io->loc_maps[o.scale]) skip = true;
            if (reuse_offsets && (o.kind == OP_REFLOC_IN || o.kind == OP_OFFSET)) skip = true;   // (their tensors still hold the key frame's)
            if (o.kind == OP_FIRST && fuse_first >= 0) skip = true;            // computed inside the next conv's patch loader
            if (o.kind == OP_CONV && o.chain > 0) skip = true;                 // computed by the chain launch at its first member's place
            if (o.kind == OP_CONV && o.fused_dw && dwpw_done) { skip = true; dwpw_done = false; }   // computed by the depthwise op's launch
            if ((int)oi < first_op) skip = true;
            if (skip) continue;
            const int lane = lanes ? o.lane : 0;
            hipStream_t s = lane == 0 ? s0 : side[lane - 1];
            if (lanes) {
                if (!lane_used[lane]) {
                    TDRN_HIP_TRY(hipStreamWaitEvent(s, ev_fork, 0));
                    lane_used[lane] = true;
                }
                for (int t : {o.in, o.res, o.off_t})
                    if (t >= 0 && tensor_lane[t] != lane) TDRN_HIP_TRY(hipStreamWaitEvent(s, tensor_ev[t], 0));
                if (late_side && lane != 0 && (o.kind == OP_CONV || o.kind == OP_OFFSET)) {
                    const int tl = late_side == 2 ? t_late2 : t_late;
                    if (tl >= 0 && tensor_shared[tl]) TDRN_HIP_TRY(hipStreamWaitEvent(s, tensor_ev[tl], 0));
                }
            }
            const bool deform_batched = o.kind == OP_DEFORM && oi + 1 < ops.size() && ops[oi + 1].kind == OP_DEFORM && n_dargs < 3;
            const bool offset_batched = o.kind == OP_OFFSET && oi + 1 < ops.size() && ops[oi + 1].kind == OP_OFFSET && n_oq < 3 &&
                                        (lanes ? ops[oi + 1].lane : 0) == lane;
            if (profile && !(o.kind == OP_DEFORM && n_dargs > 0) && !(o.kind == OP_OFFSET && n_oq > 0)) { TDRN_HIP_TRY(hipEventRecord(ev[evi], s)); }
            int rc = TDRN_OK;
            switch (o.kind) {
                case OP_FIRST:
                    if (!xin) {
                        rc = launch_u8_planes_to_f32(u8->planes, B, cfg.size, u8->mean, (float *)tptr(ws, x_t, B), s);
                        if (rc != TDRN_OK) break;
                        xin = (const float *)tptr(ws, x_t, B);
                    }
                    rc = launch_first_conv(xin, (const float *)(wb + o.w_off), (const float *)(wb + o.b_off), tptr(ws, o.out, B),
                                           B, o.hw, o.stride, o.Cout, tensors[o.out].Cpad, o.relu, cfg.dtype, s);
                    break;
                case OP_CONV: {
                    if (o.chain >= 0) {
                        // the whole chain as ONE launch at its first member's place; the other members are skipped below
                        ChainLayer cl[16];
                        const int n = (int)chain_ops.size();
                        for (int k = 0; k < n; ++k) {
                            const Op &m = ops[chain_ops[k]];
                            conv_args(m, cl[k].a);
                            int nd = 0;
                            for (int t : {m.in, m.res}) {
                                if (t < 0) continue;
                                int dep = -1;
                                for (int j = 0; j < k; ++j)
                                    if (ops[chain_ops[j]].out == t) dep = j;
                                if (dep >= 0) cl[k].dep[nd++] = dep;
                                else if (lanes && tensor_lane[t] != 0) TDRN_HIP_TRY(hipStreamWaitEvent(s, tensor_ev[t], 0));
                            }
                        }
                        if (skz_pending) {
                            TDRN_HIP_TRY(hipStreamWaitEvent(s0, ev_skz, 0));
                            skz_pending = false;
                        }
                        rc = launch_conv_chain(cl, n, (unsigned *)tail, s, status);
                        if (rc == TDRN_OK && lanes)
                            for (int k = 1; k < n; ++k) {
                                const int t = ops[chain_ops[k]].out;
                                if (t >= 0 && tensor_shared[t]) TDRN_HIP_TRY(hipEventRecord(tensor_ev[t], s));
                            }
                        break;
                    }
                    ConvArgs a;
                    conv_args(o, a);
                    if (lane != 0) a.max_wgs = side_grid;
                    else if (main_grid > 0) a.max_wgs = main_grid;
                    if (o.lane == 0 && pp_sk_planned) {
                        a.sk_ws = tail + kTailCtl;
                        a.sk_flags_zero = true;
                        if (skz_pending && pp_conv_supported(a)) {
                            TDRN_HIP_TRY(hipStreamWaitEvent(s0, ev_skz, 0));
                            skz_pending = false;
                        }
                    }
                    if ((int)oi == fuse_first) {
                        a.fuse_w = (const float *)(wb + ops[0].w_off); a.fuse_b = (const float *)(wb + ops[0].b_off);
                        a.fuse_cout = ops[0].Cout;
                        if (!xin && o.pool_t >= 0) {
                            // uint8 frames: conv3x3_ws.hip's producers read the planes themselves (the frame never exists in fp32)
                            a.fuse_x8 = u8->planes; a.fuse_mean[0] = u8->mean[0]; a.fuse_mean[1] = u8->mean[1]; a.fuse_mean[2] = u8->mean[2];
                            a.out = nullptr;
                            rc = ws_conv_supported(a) ? launch_conv3x3_ws(a, tptr(ws, o.pool_t, B), s) : TDRN_E_UNSUPPORTED;
                            if (rc != TDRN_E_UNSUPPORTED) break;          // (done, or a real error)
                            a.fuse_x8 = nullptr;
                            a.out = tptr(ws, o.out, B);
                        }
                        if (!xin) {                                       // it declined (a small batch): the fp32 route from here on
                            rc = launch_u8_planes_to_f32(u8->planes, B, cfg.size, u8->mean, (float *)tptr(ws, x_t, B), s);
                            if (rc != TDRN_OK) break;
                            xin = (const float *)tptr(ws, x_t, B);
                        }
                        a.fuse_x = xin;
                    }
                    if (a.fuse_x && !(conv_patch_enabled() && patch_conv_supported(a) > 0)) {
                        // the fusion was planned from the layer geometry; should the patch kernel decline THIS launch (a limit
                        // that depends on the batch), run the two layers as two launches: the first conv's tensor keeps its place
                        // in the workspace
                        rc = launch_first_conv(xin, a.fuse_w, a.fuse_b, tptr(ws, ops[0].out, B), B, ops[0].hw, ops[0].stride, ops[0].Cout,
                                               tensors[ops[0].out].Cpad, ops[0].relu, cfg.dtype, s);
                        if (rc != TDRN_OK) break;
                        a.fuse_x = nullptr; a.fuse_w = nullptr; a.fuse_b = nullptr; a.fuse_cout = 0;
                    }
                    if (o.pool_t >= 0) {
                        const Tensor &tp = tensors[o.pool_t];
                        if (conv_patch_enabled() && patch_conv_supported(a) > 0) {
                            a.out = nullptr;                 // only the pooled map leaves the chip
                            rc = launch_conv3x3_patch(a, tptr(ws, o.pool_t, B), s);
                        } else {
                            rc = launch_conv(a, s);
                            if (rc == TDRN_OK)
                                rc = launch_maxpool2(a.out, tptr(ws, o.pool_t, B), B, a.Ho, a.Wo, tp.Cpad, 0, cfg.dtype, s);
                        }
                        break;
                    }
                    rc = launch_conv(a, s);
                    break;
                }
                case OP_POOL: {
                    const Tensor &ti = tensors[o.in];
                    rc = launch_maxpool2(tptr(ws, o.in, B), tptr(ws, o.out, B), B, ti.H, ti.W, ti.Cpad, o.ceil, cfg.dtype, s);
                    break;
                }
                case OP_L2NORM: {
                    const Tensor &ti = tensors[o.in];
                    rc = launch_l2norm(tptr(ws, o.in, B), (const float *)(wb + o.w_off), tptr(ws, o.out, B),
                                       (long long)B * ti.H * ti.W, ti.Cpad, cfg.dtype, s);
                    break;
                }
                case OP_DW: {
                    const Tensor &ti = tensors[o.in];
                    if (o.fused_dw) {
                        const Op &c = ops[oi + 1];
                        DwPwArgs a;
                        a.in = tptr(ws, o.in, B); a.w = wb + c.w_off; a.wdw = (const float *)(wb + o.w_off); a.bdw = (const float *)(wb + o.b_off);
                        a.bias = (const float *)(wb + c.b_off); a.out = tptr(ws, c.out, B);
                        a.B = B; a.H = ti.H; a.W = ti.W; a.Cin = c.Cin; a.Cout = c.Cout; a.Npad = c.Npad; a.Cs = tensors[c.out].Cpad;
                        a.stride = o.stride; a.relu_dw = o.relu; a.relu = c.relu; a.dtype = cfg.dtype;
                        if (dwpw_supported(a)) {
                            rc = launch_dwpw(a, s);
                            dwpw_done = true;
                            // (the pointwise op's output tensor is produced HERE: its cross-lane event is recorded below through `o2`)
                            if (rc == TDRN_OK && lanes && c.out >= 0 && tensor_shared[c.out]) TDRN_HIP_TRY(hipEventRecord(tensor_ev[c.out], s));
                            break;
                        }
                    }
                    rc = launch_dwconv3(tptr(ws, o.in, B), (const float *)(wb + o.w_off), (const float *)(wb + o.b_off),
                                        tptr(ws, o.out, B), B, ti.H, ti.W, ti.Cpad, o.stride, o.relu, cfg.dtype, s, kdisable);
                    break;
                }
                case OP_REFLOC_IN:
                    if (!io->ref_loc[o.scale]) return TDRN_E_ARG;
                    // (tdrn_net_io.reserved[2]: the loc maps are still being produced on another stream -- wait for its event HERE, not
                    // at the start of the forward: the trunk above does not depend on them)
                    if (io->reserved[2]) TDRN_HIP_TRY(hipStreamWaitEvent(s, (hipEvent_t)io->reserved[2], 0));
                    rc = launch_nchw_to_nhwc(io->ref_loc[o.scale], tptr(ws, o.out, B), Bk, 12, o.hw, 12, TDRN_F32, s);
                    break;
                case OP_OFFSET: {
                    const float *loc;
                    long long bs, ps;
                    if (o.in >= 0) { loc = (const float *)tptr(ws, o.in, B); bs = (long long)o.hw * 12; ps = 12; }
                    else { loc = io->arm_loc + (size_t)scale_off[o.scale] * 4; bs = (long long)P * 4; ps = 12; }
                    // (offsets from ref_loc maps exist for the Bk key frames only; from the net's own ARM loc for every sample)
                    oq[n_oq] = OffsetProblem{loc, bs, ps, (const float *)(wb + o.w_off), (const float *)(wb + o.b_off),
                                             (float *)tptr(ws, o.out, B), o.in >= 0 ? Bk : B, o.hw, o.off_n, 0};
                    oq_op[n_oq++] = (int)oi;
                    if (!offset_batched) {
                        rc = launch_offset_conv_multi(oq, n_oq, s);
                        if (rc != TDRN_OK) break;
                        // the outputs of the launch's earlier members become visible HERE, not where their ops stood
                        for (int i = 0; i + 1 < n_oq; ++i) {
                            const int t_ = ops[oq_op[i]].out;
                            if (lanes && t_ >= 0 && tensor_shared[t_]) TDRN_HIP_TRY(hipEventRecord(tensor_ev[t_], s));
                        }
                        n_oq = 0;
                    }
                    break;
                }
                case OP_DEFORM: {
                    const Tensor &ti = tensors[o.in];
                    const Tensor &tf = tensors[o.off_t];
                    DeformArgs a;
                    a.in = tptr(ws, o.in, B); a.zero_page = wb; a.n_branches = o.n_branches;
                    const float *off = (const float *)tptr(ws, o.off_t, B);
                    a.br[0] = DeformBranch{off + o.off_c0[0], tf.C, wb + o.w_off, 3, 3, 1, 1, 1, o.G};
                    if (o.n_branches == 2) a.br[1] = DeformBranch{off + o.off_c0[1], tf.C, wb + o.w2_off, 5, 5, 2, 1, 1, o.G};
                    if (Bk < B)
                        for (int k = 0; k < o.n_branches; ++k) a.br[k].off_rows = Bk * ti.H * ti.W;
                    a.B = B; a.H = ti.H; a.W = ti.W; a.Cin = o.Cin; a.Ho = ti.H; a.Wo = ti.W; a.Cout = o.Cout; a.Npad = o.Npad;
                    float *locbase = o.out_kind == OUT_ARM_LOC ? io->arm_loc : io->odm_loc;
                    a.out0 = locbase + (size_t)scale_off[o.scale] * 4; a.o0_bs = (long long)P * 4; a.o0_ps = 12;
                    a.out1 = io->conf + (size_t)scale_off[o.scale] * C; a.o1_bs = (long long)P * C; a.o1_ps = 3 * C;
                    a.split = 12; a.dtype = cfg.dtype;
                    dargs[n_dargs++] = a;
                    if (o.y_t >= 0) ts_op[n_dargs - 1] = (int)oi;
                    if (!deform_batched && o.y_t >= 0) {
                        // transform: Y = X * W_taps (1x1 GEMM, net dtype out) per level, then sample: all pyramid levels in one
                        // launch.  Y is addressed with 32-bit byte offsets (deform.hip), so a batch whose Y would pass 4 GiB at
                        // some level runs as several batch RANGES through the same Y buffers, one (transforms, sample) group per
                        // range on this stream -- per-frame arithmetic untouched (DRN at 512 px: 171 frames and up; at 320 px: 437).
                        int Bc = B;
                        for (int i = 0; i < n_dargs; ++i) {
                            const Op &d = ops[ts_op[i]];
                            int taps = 0;
                            for (int k = 0; k < d.n_branches; ++k) taps += dargs[i].br[k].kh * dargs[i].br[k].kw;
                            const int fit = deform_ts_max_batch(dargs[i].H, dargs[i].W, d.y_cols, taps);
                            Bc = fit < Bc ? fit : Bc;
                        }
                        if (Bc < 1) { rc = TDRN_E_UNSUPPORTED; break; }
                        {   // ... and (round 5) a range's Y is kept below 192 MiB, so that it is still in the 256-MiB memory-side cache when
                            // the sampling launch gathers it: the pair of launches 277-285 -> 254-255 us alone at batch 32 (two ranges of
                            // 16 frames; ranges of 8 / 4 frames lose it again to the extra launches), 520 -> 488 us at MobileNet's batch 64.
                            // Per-frame arithmetic untouched.  TDRN_TS_RANGE_MB=0 switches it off, another value is another bound.
                            static long long cap_mb = -1;
                            if (cap_mb < 0) { const char *e = getenv("TDRN_TS_RANGE_MB"); cap_mb = e ? atoll(e) : 192; }
                            size_t per_frame = 0;
                            for (int i = 0; i < n_dargs; ++i)
                                per_frame += (size_t)dargs[i].H * dargs[i].W * ops[ts_op[i]].y_cols * es;       // (one column group's Y: a group's two launches are adjacent)
                            if (cap_mb > 0 && per_frame > 0 && !(kdisable & 512)) {
                                long long fit = (long long)((size_t)cap_mb << 20) / (long long)per_frame;
                                fit = fit < 1 ? 1 : fit;
                                if (fit < Bc) Bc = (int)fit;
                            }
                        }
                        // output columns in groups of 80 (deform.hip: a Y row is 80 columns): group g = columns [80 g, 80 g + 80) of
                        // [12 loc ; 3 * classes conf], its own weight rows, Y region, transform and sampling launch
                        const int n_groups = ops[ts_op[0]].y_groups;
                        for (int b0 = 0; b0 < B && rc == TDRN_OK; b0 += Bc)
                          for (int yg = 0; yg < n_groups && rc == TDRN_OK; ++yg) {
                            const int nb = B - b0 < Bc ? B - b0 : Bc;
                            DeformArgs ca[4];
                            YGemmProblem yq[4];
                            int n_yq = 0;
                            bool all_ygemm = true;
                            for (int i = 0; i < n_dargs; ++i) all_ygemm = all_ygemm && ygemm_supported(ops[ts_op[i]].Cin, ops[ts_op[i]].y_cols, cfg.dtype);
                            static int ymulti = -1;
                            if (ymulti < 0) { const char *e = getenv("TDRN_YGEMM_MULTI"); ymulti = e ? atoi(e) : 1; }
                            for (int i = 0; i < n_dargs && rc == TDRN_OK; ++i) {
                                const Op &d = ops[ts_op[i]];
                                DeformArgs &c = ca[i];
                                c = dargs[i];
                                const size_t px0 = (size_t)b0 * c.H * c.W;
                                c.B = nb;
                                c.in = (const char *)c.in + px0 * c.Cin * es;
                                for (int k = 0; k < c.n_branches; ++k) {
                                    if (c.br[k].off_rows) c.br[k].off_row0 = (int)(px0 % (size_t)c.br[k].off_rows);
                                    else c.br[k].off += px0 * c.br[k].off_stride;
                                }
                                c.out0 += (size_t)b0 * c.o0_bs;
                                c.out1 += (size_t)b0 * c.o1_bs;
                                int taps = 0;
                                for (int k = 0; k < c.n_branches; ++k) taps += c.br[k].kh * c.br[k].kw;
                                if (d.y_groups != n_groups) { rc = TDRN_E_STATE; break; }
                                void *ybuf = tptr(ws, d.y_t, B) + (size_t)yg * d.y_cols * c.H * c.W * es * B;
                                const char *wty = wb + d.wt_off + (size_t)yg * d.y_cols * d.Cin * es;
                                c.Cout = d.Cout - 80 * yg < 80 ? d.Cout - 80 * yg : 80;
                                if (yg > 0) { c.split = 0; c.out1 += 80 * yg - 12; }     // (columns 12.. are conf columns: group g starts at conf column 80 g - 12)
                                if (all_ygemm && ymulti) {       // all levels' transforms in ONE launch (below)
                                    yq[n_yq++] = YGemmProblem{c.in, wty, ybuf, (long long)nb * c.H * c.W, d.y_cols, d.y_cols, ts_tap_major ? taps : 0};
                                } else if (ygemm_supported(d.Cin, d.y_cols, cfg.dtype)) {
                                    rc = launch_ygemm(c.in, wty, ybuf, (long long)nb * c.H * c.W, d.y_cols, d.y_cols, cfg.dtype, s, ts_tap_major ? taps : 0);
                                } else {
                                    ConvArgs g;
                                    g.in = c.in; g.w = wty; g.bias = (const float *)(wb + d.bt_off); g.zero_page = wb;
                                    g.B = nb; g.H = c.H; g.W = c.W; g.Cin = d.Cin; g.Ho = c.H; g.Wo = c.W; g.Cout = d.y_cols; g.Npad = d.y_cols;
                                    g.kh = g.kw = 1; g.stride = 1; g.pad = 0; g.dil = 1; g.relu = 0; g.phases = 1; g.dtype = cfg.dtype;
                                    g.out = ybuf; g.kdisable = kdisable;
                                    g.o_cs = d.y_cols; g.o_rs = (long long)c.W * d.y_cols; g.o_bs = (long long)c.H * c.W * d.y_cols;
                                    rc = launch_conv(g, s);
                                }
                                ts_y[i] = ybuf; ts_cs[i] = d.y_cols;
                            }
                            if (rc == TDRN_OK && n_yq > 0) rc = launch_ygemm_multi(yq, n_yq, cfg.dtype, s, kdisable);
                            if (rc == TDRN_OK) rc = launch_deform_sample_multi(ca, ts_y, ts_cs, n_dargs, s, ts_tap_major);
                          }
                        n_dargs = 0;
                        break;
                    }
                    if (!deform_batched) {      // all pyramid levels in one launch
                        const int split = !deform_split ? 0 : (o.n_branches == 2 ? 1 : ((o.G >= 2 && o.G % 2 == 0) ? 2 : 0));
                        if (split) {             // the two branches / the two halves of the groups accumulate into zeroed outputs
                            if (zeroed_early) {
                                TDRN_HIP_TRY(hipStreamWaitEvent(s, ev_zero, 0));
                            } else {
                                float *locbase0 = o.out_kind == OUT_ARM_LOC ? io->arm_loc : io->odm_loc;
                                TDRN_HIP_TRY(hipMemsetAsync(locbase0, 0, (size_t)B * P * 4 * sizeof(float), s));
                                TDRN_HIP_TRY(hipMemsetAsync(io->conf, 0, (size_t)B * P * C * sizeof(float), s));
                            }
                        }
                        if (dargs[0].Npad <= 128) {
                            rc = launch_deform_multi(dargs, n_dargs, s, split);
                        } else {
                            // the gather kernel holds at most 128 output columns per workgroup (deform.hip): COCO's 12 + 243 columns run as
                            // column ranges of 128, each with its own weight rows and output columns (round 5; the C-ABI op does the same)
                            for (int c0 = 0; c0 < dargs[0].Cout && rc == TDRN_OK; c0 += 128) {
                                DeformArgs ga[4];
                                for (int i = 0; i < n_dargs; ++i) {
                                    ga[i] = dargs[i];
                                    const int cols = dargs[i].Cout - c0 < 128 ? dargs[i].Cout - c0 : 128;
                                    ga[i].Cout = cols; ga[i].Npad = deform_n_pad(cols);
                                    for (int k = 0; k < ga[i].n_branches; ++k)
                                        ga[i].br[k].w = (const char *)dargs[i].br[k].w + (size_t)c0 * dargs[i].br[k].kh * dargs[i].br[k].kw * dargs[i].Cin * es;
                                    if (c0 > 0) { ga[i].out1 += c0 - dargs[i].split; ga[i].split = 0; }
                                }
                                rc = launch_deform_multi(ga, n_dargs, s, split);
                            }
                        }
                        n_dargs = 0;
                    }
                    break;
                }
                case OP_SOFTMAX:
                    rc = launch_softmax_rows(io->conf, io->conf, (long long)B * P, C, s);
                    break;
                case OP_OFF_OUT: {
                    const Tensor &tf = tensors[o.in];
                    rc = launch_nhwc_to_nchw_f32((const float *)tptr(ws, o.in, B), (long long)tf.H * tf.W * tf.C, tf.C,
                                                 io->offsets[o.scale], Bk, o.Cout, tf.H * tf.W, s);
                    break;
                }
                case OP_LOC_OUT:
                    rc = launch_nhwc_to_nchw_f32(io->arm_loc + (size_t)scale_off[o.scale] * 4, (long long)P * 4, 12,
                                                 io->loc_maps[o.scale], B, 12, fm[o.scale] * fm[o.scale], s);
                    break;
            }
            if (rc != TDRN_OK) return rc;
            offset_ops_enqueued += o.kind == OP_OFFSET;
            if (lanes && o.out >= 0 && tensor_shared[o.out] && !(o.kind == OP_OFFSET && n_oq > 0)) TDRN_HIP_TRY(hipEventRecord(tensor_ev[o.out], s));
            if (lanes && o.pool_t >= 0 && tensor_shared[o.pool_t]) TDRN_HIP_TRY(hipEventRecord(tensor_ev[o.pool_t], s));
            if (profile && !(o.kind == OP_DEFORM && n_dargs > 0) && !(o.kind == OP_OFFSET && n_oq > 0)) {
                TDRN_HIP_TRY(hipEventRecord(ev[evi + 1], s));
                ev_stat.push_back(o.stat);
                ev_op.push_back((int)oi);
                evi += 2;
            }
        }
        const int jrc = join.run();
        if (jrc == TDRN_OK && cfg.deform && !reuse_offsets && offset_ops_planned > 0 && offset_ops_enqueued == offset_ops_planned) {
            offs_ws = ws; offs_batch = B; offs_key_batch = Bk;
        }
        return jrc;
    }

    int check_status(unsigned *detail)
    {
        unsigned d = 0;
        if (status) {
            volatile unsigned *w = status;
            d = (w[0] ? 1u : 0u) | (w[1] ? 2u : 0u);
            if (d) { w[0] = 0; w[1] = 0; }
        }
        if (detail) *detail = d;
        return d ? TDRN_E_DEVICE : TDRN_OK;
    }

    int collect_stats(tdrn_kernel_stat *out, int max_entries)
    {
        for (int i = 0; i < ST_COUNT; ++i) {
            memset(&stats[i], 0, sizeof(stats[i]));
            strncpy(stats[i].name, kStatNames[i], sizeof(stats[i].name) - 1);
        }
        bool prev_deform = false, prev_offset = false;
        for (const Op &o : ops) {
            if (!(o.kind == OP_DEFORM && prev_deform) && !(o.kind == OP_OFFSET && prev_offset) && !(o.kind == OP_CONV && o.chain > 0) &&
                !(o.kind == OP_CONV && o.fused_dw)) stats[o.stat].launches += 1;
            prev_deform = o.kind == OP_DEFORM;
            prev_offset = o.kind == OP_OFFSET;
            stats[o.stat].flops += o.flops * last_batch;
            stats[o.stat].bytes += o.bytes * last_batch;
        }
        for (size_t i = 0; i < ev_stat.size(); ++i) {
            float ms = 0.f;
            TDRN_HIP_TRY(hipEventSynchronize(ev[2 * i + 1]));
            TDRN_HIP_TRY(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]));
            stats[ev_stat[i]].ms += ms;
        }
        int n = 0;
        for (int i = 0; i < ST_COUNT && n < max_entries; ++i)
            if (stats[i].launches) out[n++] = stats[i];
        return n;
    }
};

// ---- C ABI (tdrn_hip.h section iii) ------------------------------------------------------------
extern "C" {

int tdrn_net_create(const tdrn_net_config *cfg, tdrn_net **out)
{
    if (!cfg || !out) return TDRN_E_ARG;
    tdrn_net *n = new tdrn_net();
    n->cfg = *cfg;
    const int rc = n->build();
    if (rc != TDRN_OK) {
        delete n;
        return rc;
    }
    *out = n;
    return TDRN_OK;
}

void tdrn_net_destroy(tdrn_net *net)
{
    if (!net) return;
    const int d = net->dev;
    for (hipEvent_t e : net->ev) pool::put_timing_event(d, e);         // (timing events of the profiling passes: never captured; pooled like the rest)
    for (hipEvent_t e : net->tensor_ev) pool::put_event(d, e);
    for (int i = 0; i < tdrn_net::kLanes - 1; ++i) {
        pool::put_event(d, net->ev_join[i]);
        pool::put_stream(d, net->side[i]);
    }
    pool::put_event(d, net->ev_fork);
    pool::put_event(d, net->ev_zero);
    pool::put_event(d, net->ev_skz);
    pool::put_status(net->status);
    delete net;
}

int tdrn_net_param_count(const tdrn_net *net) { return net ? (int)net->params.size() : TDRN_E_ARG; }

int tdrn_net_param_info(const tdrn_net *net, int index, const char **name, int64_t shape[4], int *ndim)
{
    if (!net || index < 0 || index >= (int)net->params.size()) return TDRN_E_ARG;
    const ParamSpec &p = net->params[index];
    if (name) *name = p.name.c_str();
    if (ndim) *ndim = (int)p.shape.size();
    if (shape)
        for (size_t i = 0; i < 4; ++i) shape[i] = i < p.shape.size() ? p.shape[i] : 1;
    return TDRN_OK;
}

int tdrn_net_set_param(tdrn_net *net, const char *name, const float *data_host, int64_t numel)
{
    if (!net || !name || !data_host) return TDRN_E_ARG;
    auto it = net->param_index.find(name);
    if (it == net->param_index.end()) return TDRN_E_PARAM;
    int64_t want = 1;
    for (int64_t d : net->params[it->second].shape) want *= d;
    if (want != numel) return TDRN_E_PARAM;
    net->staged[name].assign(data_host, data_host + numel);
    net->weights_ready = false;
    return TDRN_OK;
}

size_t tdrn_net_weight_bytes(const tdrn_net *net) { return net ? net->blob_bytes : 0; }
size_t tdrn_net_workspace_bytes(const tdrn_net *net, int batch) { return net && batch > 0 ? net->ws_per_sample * (size_t)batch + net->ws_fixed : 0; }
int tdrn_net_num_priors(const tdrn_net *net) { return net ? net->P : TDRN_E_ARG; }

int tdrn_net_pack_weights(tdrn_net *net, void *weights_dev, size_t weights_bytes, void *stream)
{
    if (!net || !weights_dev) return TDRN_E_ARG;
    if (weights_bytes < net->blob_bytes) return TDRN_E_WORKSPACE;
    for (const ParamSpec &p : net->params)
        if (!net->staged.count(p.name)) return TDRN_E_PARAM;
    std::vector<char> host;
    TDRN_TRY(net->pack(host));
    hipStream_t s = (hipStream_t)stream;
    TDRN_HIP_TRY(hipMemcpyAsync(weights_dev, host.data(), host.size(), hipMemcpyHostToDevice, s));
    TDRN_HIP_TRY(hipStreamSynchronize(s));
    net->weights_ready = true;
    net->staged.clear();   // the fp32 staging copy is no longer needed
    return TDRN_OK;
}

int tdrn_net_adopt_weights(tdrn_net *net)
{
    if (!net) return TDRN_E_ARG;
    net->weights_ready = true;
    return TDRN_OK;
}

int tdrn_net_forward(tdrn_net *net, const void *weights_dev, void *workspace, size_t workspace_bytes, const tdrn_net_io *io,
                     void *stream)
{
    if (!net) return TDRN_E_ARG;
    return net->forward(weights_dev, workspace, workspace_bytes, io, (hipStream_t)stream);
}

int tdrn_net_check(tdrn_net *net, unsigned *detail)
{
    if (!net) return TDRN_E_ARG;
    return net->check_status(detail);
}

int tdrn_net_op_count(const tdrn_net *net) { return net ? (int)net->ops.size() : TDRN_E_ARG; }

int tdrn_net_op_info(const tdrn_net *net, int index, tdrn_op_info *out)
{
    if (!net || !out || index < 0 || index >= (int)net->ops.size()) return TDRN_E_ARG;
    const Op &o = net->ops[index];
    memset(out, 0, sizeof(*out));
    switch (o.kind) {
        case OP_FIRST: out->kind = 0; break;
        case OP_CONV: out->kind = o.phases == 4 ? 2 : 1; break;
        case OP_DW: out->kind = 3; break;
        case OP_POOL: out->kind = 4; break;
        case OP_L2NORM: out->kind = 5; break;
        case OP_OFFSET: out->kind = 6; break;
        case OP_DEFORM: out->kind = 7; break;
        default: out->kind = 8; break;
    }
    out->in = o.in; out->out = o.out; out->res = o.res; out->pool = o.pool_t; out->off = o.off_t; out->y = o.y_t;
    out->k = o.k; out->stride = o.stride; out->pad = o.pad; out->dil = o.dil; out->relu = o.relu; out->ceil_mode = o.ceil;
    out->splitk = o.splitk; out->groups = o.G; out->out_kind = o.out_kind; out->level = o.scale;
    out->n_branches = o.n_branches; out->k2 = o.k2; out->pad2 = o.pad2; out->off_c0[0] = o.off_c0[0]; out->off_c0[1] = o.off_c0[1];
    if (o.kind == OP_DEFORM && o.y_t >= 0) {
        static int tm = -1;
        if (tm < 0) { const char *e = getenv("TDRN_Y_TAP_MAJOR"); tm = e ? atoi(e) : 1; }
        bool all = tm != 0;
        for (const Op &d : net->ops)
            if (d.kind == OP_DEFORM && d.y_t >= 0 && !ygemm_supported(d.Cin, d.y_cols, net->cfg.dtype)) all = false;
        out->y_tap_major = all ? 1 : 0;
        out->y_groups = o.y_groups;
    }
    out->fused_first = (index == net->fuse_first) ? 1 : 0;
    out->fused_dw = o.fused_dw;
    auto cp = [](char *d, const std::string &v) { strncpy(d, v.c_str(), 47); };
    cp(out->w, o.w); cp(out->b, o.b); cp(out->bn, o.bn); cp(out->w2, o.w2); cp(out->b2, o.b2);
    return TDRN_OK;
}

int tdrn_net_tensor_count(const tdrn_net *net) { return net ? (int)net->tensors.size() : TDRN_E_ARG; }

int tdrn_net_tensor_info(const tdrn_net *net, int index, const char **label, int *C, int *H, int *W)
{
    if (!net || index < 0 || index >= (int)net->tensors.size()) return TDRN_E_ARG;
    const Tensor &t = net->tensors[index];
    if (label) *label = t.label.c_str();
    if (C) *C = t.C;
    if (H) *H = t.H;
    if (W) *W = t.W;
    return TDRN_OK;
}

int tdrn_net_read_tensor(const tdrn_net *net, const void *workspace, int batch, int index, float *out_dev, void *stream)
{
    if (!net || !workspace || !out_dev || batch <= 0 || index < 0 || index >= (int)net->tensors.size()) return TDRN_E_ARG;
    const Tensor &t = net->tensors[index];
    return launch_nhwc_any_to_nchw_f32((const char *)workspace + t.off * (size_t)batch, t.f32 ? TDRN_F32 : net->cfg.dtype,
                                       t.Cpad, out_dev, batch, t.C, t.H * t.W, (hipStream_t)stream);
}

int tdrn_net_write_tensor(const tdrn_net *net, void *workspace, int batch, int index, const float *in_dev, void *stream)
{
    if (!net || !workspace || !in_dev || batch <= 0 || index < 0 || index >= (int)net->tensors.size()) return TDRN_E_ARG;
    const Tensor &t = net->tensors[index];
    return launch_nchw_to_nhwc(in_dev, (char *)workspace + t.off * (size_t)batch, batch, t.C, t.H * t.W, t.Cpad,
                               t.f32 ? TDRN_F32 : net->cfg.dtype, (hipStream_t)stream);
}

int tdrn_net_forward_from(tdrn_net *net, const void *weights_dev, void *workspace, size_t workspace_bytes, const tdrn_net_io *io,
                          int first_op, void *stream)
{
    if (!net || first_op < 0 || first_op > (int)net->ops.size()) return TDRN_E_ARG;
    if (net->use_lanes) return TDRN_E_STATE;
    net->first_op = first_op;
    const int rc = net->forward(weights_dev, workspace, workspace_bytes, io, (hipStream_t)stream);
    net->first_op = 0;
    return rc;
}

int tdrn_net_profile(tdrn_net *net, int enable)
{
    if (!net) return TDRN_E_ARG;
    net->profile = enable < 0 ? 0 : (enable > 2 ? 2 : enable);
    return TDRN_OK;
}

int tdrn_net_op_stats(tdrn_net *net, tdrn_kernel_stat *out, int max_entries)
{
    if (!net || !out || max_entries <= 0) return TDRN_E_ARG;
    int n = 0;
    for (size_t i = 0; i < net->ev_op.size() && n < max_entries; ++i) {
        const Op &o = net->ops[net->ev_op[i]];
        float ms = 0.f;
        TDRN_HIP_TRY(hipEventSynchronize(net->ev[2 * i + 1]));
        TDRN_HIP_TRY(hipEventElapsedTime(&ms, net->ev[2 * i], net->ev[2 * i + 1]));
        tdrn_kernel_stat &k = out[n++];
        memset(&k, 0, sizeof(k));
        std::string name = std::string(kStatNames[o.stat]) + ":" + (o.w.empty() ? (o.in >= 0 ? net->tensors[o.in].label : "") : o.w);
        if (o.kind == OP_CONV && o.chain == 0) name = "conv_chain:" + o.w + "+" + std::to_string(net->chain_ops.size() - 1);
        strncpy(k.name, name.c_str(), sizeof(k.name) - 1);
        k.launches = 1;
        k.flops = o.flops * net->last_batch;
        k.bytes = o.bytes * net->last_batch;
        if (o.kind == OP_CONV && o.chain == 0)      // one launch covers all members
            for (size_t j = 1; j < net->chain_ops.size(); ++j) {
                k.flops += net->ops[net->chain_ops[j]].flops * net->last_batch;
                k.bytes += net->ops[net->chain_ops[j]].bytes * net->last_batch;
            }
        if (o.kind == OP_DEFORM)   // one launch covers the preceding deform ops of the other pyramid levels
            for (int j = net->ev_op[i] - 1; j >= 0 && net->ops[j].kind == OP_DEFORM; --j) {
                k.flops += net->ops[j].flops * net->last_batch;
                k.bytes += net->ops[j].bytes * net->last_batch;
            }
        k.ms = ms;
    }
    return n;
}

int tdrn_net_op_timeline(tdrn_net *net, float *start_ms, float *end_ms, int *lane, int max_entries)
{
    if (!net || !start_ms || !end_ms || max_entries <= 0) return TDRN_E_ARG;
    int n = 0;
    for (size_t i = 0; i < net->ev_op.size() && n < max_entries; ++i, ++n) {
        TDRN_HIP_TRY(hipEventSynchronize(net->ev[2 * i + 1]));
        TDRN_HIP_TRY(hipEventElapsedTime(&start_ms[n], net->ev[0], net->ev[2 * i]));
        TDRN_HIP_TRY(hipEventElapsedTime(&end_ms[n], net->ev[0], net->ev[2 * i + 1]));
        if (lane) lane[n] = net->profile == 2 && net->use_lanes ? net->ops[net->ev_op[i]].lane : 0;
    }
    return n;
}

int tdrn_net_kernel_stats(tdrn_net *net, tdrn_kernel_stat *out, int max_entries)
{
    if (!net || !out || max_entries <= 0) return TDRN_E_ARG;
    return net->collect_stats(out, max_entries);
}

}  // extern "C"
