// dev/permlane_probe.hip -- what do v_permlane16_swap / v_permlane32_swap do to a wave?  (developer probe, never shipped)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *o)
{
    const unsigned l = threadIdx.x;
    auto r16 = __builtin_amdgcn_permlane16_swap(l, l + 100u, false, false);
    auto r32 = __builtin_amdgcn_permlane32_swap(l, l + 100u, false, false);
    o[l] = r16[0]; o[64 + l] = r16[1]; o[128 + l] = r32[0]; o[192 + l] = r32[1];
}
int main()
{
    unsigned *d, h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char *names[4] = {"permlane16_swap [0] (new vdst; vdst = lane)", "permlane16_swap [1] (new src; src = lane + 100)", "permlane32_swap [0]", "permlane32_swap [1]"};
    for (int a = 0; a < 4; ++a) {
        printf("%s:\n", names[a]);
        for (int i = 0; i < 64; ++i) printf("%4u%s", h[a * 64 + i], i % 16 == 15 ? "\n" : "");
    }
    return 0;
}
