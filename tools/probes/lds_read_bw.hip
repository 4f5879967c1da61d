// Probe: sustained LDS read bandwidth per CU for the fragment-read pattern of score_mask_topk (16 waves per workgroup, every wave reads the
// whole staged tile with ds_read_b128 / ds_read_b64) -- build: hipcc -O3 --offload-arch=gfx950 tools/probes/lds_read_bw.hip -o /tmp/lds_read_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
constexpr int RH = 144, HALF = 64 * RH, STAGEB = 2 * HALF, RING = 4;
template <int WIDTH, int NW>
__global__ __launch_bounds__(64 * NW) void probe(int iters, float *out, long long *cyc) {
    extern __shared__ unsigned char smem[];
    for (int i = threadIdx.x; i < RING * STAGEB / 4; i += blockDim.x) reinterpret_cast<unsigned *>(smem)[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        const unsigned char *buf = smem + (it & (RING - 1)) * STAGEB;
#pragma unroll
        for (int s0 = 0; s0 < 4; s0 += 2) {
            if (WIDTH == 16) {
                f16x8 v[2][2][2];
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks)
                            v[sub][pl][ks] = *reinterpret_cast<const f16x8 *>(buf + (g & 1) * HALF + ((s0 + sub) * 16 + c) * RH + ((pl * 2 + (g >> 1)) * 2 + ks) * 16);
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("" ::"v"(v[0][0][0]), "v"(v[0][0][1]), "v"(v[0][1][0]), "v"(v[0][1][1]), "v"(v[1][0][0]), "v"(v[1][0][1]), "v"(v[1][1][0]), "v"(v[1][1][1]));
            } else {
                f16x4 v[2][2][2][2];
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                            for (int h = 0; h < 2; ++h)
                                v[sub][pl][ks][h] = *reinterpret_cast<const f16x4 *>(buf + ((s0 + sub) * 16 + c) * 264 + ((pl * 2 + ks) * 8 + g * 2 + h) * 8);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl)
                        asm volatile("" ::"v"(v[sub][pl][0][0]), "v"(v[sub][pl][0][1]), "v"(v[sub][pl][1][0]), "v"(v[sub][pl][1][1]));
            }
        }
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    if (iters < 0) out[0] = 1.f;
}
template <int WIDTH, int NW>
void run(const char *name, int iters) {
    float *out; long long *cyc;
    hipMalloc(&out, 4); hipMalloc(&cyc, 8 * 256);
    const int shm = WIDTH == 16 ? RING * STAGEB : 64 * 1024 + 4096;
    hipFuncSetAttribute((const void *)probe<WIDTH, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, shm);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    probe<WIDTH, NW><<<256, 64 * NW, shm>>>(10, out, cyc);
    hipEventRecord(a);
    probe<WIDTH, NW><<<256, 64 * NW, shm>>>(iters, out, cyc);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    const double bytes_per_cu = (double)iters * NW * 16384.0;
    printf("%s: %d waves/CU, %.3f ms, %.1f GB/s per CU, %.1f B per shader clock (clock64 ticks %lld -> %.1f B/tick)\n", name, NW, ms, bytes_per_cu / ms / 1e6,
           bytes_per_cu / (ms * 1e-3 * 2.4e9), h[0], bytes_per_cu / (double)h[0]);
}
int main() {
    run<16, 16>("ds_read_b128, topk image", 20000);
    run<16, 8>("ds_read_b128, topk image", 20000);
    run<16, 4>("ds_read_b128, topk image", 20000);
    run<8, 16>("ds_read_b64, rows of 264 B", 20000);
    run<8, 4>("ds_read_b64, rows of 264 B", 20000);
    return 0;
}
