// Which XCD does workgroup b of a 1-D launch run on?  Reads HW_REG_XCC_ID (hwreg 20, bits 3:0) per workgroup and prints the map
// for a few grid sizes.   hipcc --offload-arch=gfx950 -O2 tools/probes/xcc_probe.hip -o tools/probes/xcc_probe && tools/probes/xcc_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(int *out, int spin) {
    if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15;
    for (volatile int i = 0; i < spin; ++i) {}          // keep the workgroup alive so that the whole grid is in flight
}
int main() {
    for (int grid : {64, 1024, 6104, 31252}) {
        int *d; hipMalloc(&d, grid * sizeof(int));
        hipLaunchKernelGGL(probe, dim3(grid), dim3(256), 0, 0, d, 2000);
        std::vector<int> h(grid); hipMemcpy(h.data(), d, grid * sizeof(int), hipMemcpyDeviceToHost);
        int match = 0, cnt[16] = {0};
        for (int b = 0; b < grid; ++b) { match += (h[b] == b % 8); cnt[h[b] & 15]++; }
        printf("grid %d: xcc == b %% 8 for %d of %d workgroups; first 24:", grid, match, grid);
        for (int b = 0; b < 24 && b < grid; ++b) printf(" %d", h[b]);
        printf("; per-XCC counts:"); for (int x = 0; x < 8; ++x) printf(" %d", cnt[x]);
        printf("\n");
        hipFree(d);
    }
    return 0;
}
