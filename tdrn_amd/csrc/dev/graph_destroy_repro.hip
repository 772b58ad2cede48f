// graph_destroy_repro.hip -- stand-alone probe of the crash that made libtdrn_hip POOL its side-lane streams and events instead of
// destroying them with their net (net.hip `pool`; bench.py KEEP_ALIVE).  Seen on ROCm 7.2 / gfx950 through torch.cuda.CUDAGraph:
// a step captured for a net created AFTER another net's side streams / events (which had taken part in an earlier capture) were
// destroyed crashed inside hipGraphLaunch (SIGSEGV).  This file replays that life cycle with nothing but the HIP runtime:
//     round r: create 3 side streams + 8 events, capture a fork/join step (main stream -> side streams -> main) into a graph,
//              instantiate, launch it 20 times, then destroy (argv[1] = 1, default) or keep (argv[1] = 0) exec, graph, streams, events.
//   hipcc --offload-arch=gfx950 -O2 -o graph_destroy_repro graph_destroy_repro.hip && ./graph_destroy_repro [destroy=1] [rounds=6] [mode=0]
// mode 0: capture mode global (torch's default), 1: thread-local, 2: relaxed.  Prints one line per round; exit code 0 = no crash.
// Result on the round-4 boxes: see profiles/r04_experiments.md ("hipGraphLaunch after destruction").
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(2); } } while (0)
__global__ void bump(float *p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.f; }
int main(int argc, char **argv)
{
    const int destroy = argc > 1 ? atoi(argv[1]) : 1, rounds = argc > 2 ? atoi(argv[2]) : 6, mode = argc > 3 ? atoi(argv[3]) : 0;
    const hipStreamCaptureMode cm = mode == 0 ? hipStreamCaptureModeGlobal : (mode == 1 ? hipStreamCaptureModeThreadLocal : hipStreamCaptureModeRelaxed);
    float *buf; const int n = 1 << 20;
    CK(hipMalloc(&buf, 4 * n * sizeof(float))); CK(hipMemset(buf, 0, 4 * n * sizeof(float)));
    hipStream_t main_s; CK(hipStreamCreateWithFlags(&main_s, hipStreamNonBlocking));
    for (int r = 0; r < rounds; ++r) {
        std::vector<hipStream_t> side(3); std::vector<hipEvent_t> ev(8);
        for (auto &s : side) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (int w = 0; w < 2; ++w) {                            // warm-up outside capture, as GraphedCall does
            bump<<<n / 256, 256, 0, main_s>>>(buf, n);
            for (int l = 0; l < 3; ++l) bump<<<n / 256, 256, 0, side[l]>>>(buf + (l + 1) * n, n);
        }
        CK(hipDeviceSynchronize());
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(main_s, cm));
        CK(hipEventRecord(ev[0], main_s));                        // fork
        bump<<<n / 256, 256, 0, main_s>>>(buf, n);
        for (int l = 0; l < 3; ++l) {
            CK(hipStreamWaitEvent(side[l], ev[0], 0));
            CK(hipMemsetAsync(buf + (l + 1) * n, 0, 1024, side[l]));      // (net.hip zeroes flag words on a side lane)
            bump<<<n / 256, 256, 0, side[l]>>>(buf + (l + 1) * n, n);
            CK(hipEventRecord(ev[1 + l], side[l]));
            CK(hipStreamWaitEvent(main_s, ev[1 + l], 0));         // join
        }
        bump<<<n / 256, 256, 0, main_s>>>(buf, n);
        CK(hipStreamEndCapture(main_s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int k = 0; k < 20; ++k) CK(hipGraphLaunch(ge, main_s));
        CK(hipStreamSynchronize(main_s));
        printf("round %d: captured, 20 launches ok%s\n", r, destroy ? ", destroying exec / graph / side streams / events" : "");
        fflush(stdout);
        if (destroy) {
            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
            for (auto &s : side) CK(hipStreamDestroy(s));
            for (auto &e : ev) CK(hipEventDestroy(e));
        }
    }
    printf("no crash (destroy=%d rounds=%d mode=%d)\n", destroy, rounds, mode);
    return 0;
}
