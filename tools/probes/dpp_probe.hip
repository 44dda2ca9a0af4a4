// Which way does DPP wave_shl:1 move data on gfx950?  Prints lane 0..3 and 62..63 of out[i] = dpp(lane id).
// Build: hipcc --offload-arch=gfx950 -O2 -o dpp_probe dpp_probe.hip ; expected (used by from_next_lane in dbg_sk2.h):
// lane i receives lane i + 1, lane 63 keeps `old`.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int *out) {
    const int v = threadIdx.x;
    out[threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x130, 0xF, 0xF, false);
}
int main() {
    int *d, h[64];
    if (hipMalloc(&d, 256) != hipSuccess) return 2;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    if (hipMemcpy(h, d, 256, hipMemcpyDeviceToHost) != hipSuccess) return 2;
    printf("wave_shl:1 -> lane0=%d lane1=%d lane2=%d lane31=%d lane32=%d lane62=%d lane63=%d\n", h[0], h[1], h[2], h[31], h[32], h[62], h[63]);
    bool ok = h[63] == -1;
    for (int i = 0; i < 63; ++i) ok = ok && h[i] == i + 1;
    printf(ok ? "OK: lane i receives lane i + 1\n" : "UNEXPECTED\n");
    return ok ? 0 : 1;
}
