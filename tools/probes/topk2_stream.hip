// Probe for a re-designed score_mask_topk stream (round 4): how fast can 1 M users x 100 K items (d = 64, fp16 high pieces) be scored and
// pre-filtered when a wave owns 32 users (v_mfma_f32_32x32x16_f16: items = rows, users = columns, so a lane's 16 accumulators all belong to ONE
// user and compare against one threshold register), stages are 128 items behind one workgroup barrier (double-buffered LDS tiles) instead of a
// counter ring, and the per-score bookkeeping is a subtract + v_alignbit into a per-lane bit mask?  Counts survivors only (no queues, no lists).
//   build: hipcc -O3 --offload-arch=gfx950 tools/probes/topk2_stream.hip -o tools/probes/topk2_stream.bin      run: ./topk2_stream.bin [U] [I]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <random>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int D = 64, RS = 144;                  // LDS row stride: 128 B of high pieces + 16 B (conflict-free ds_read_b128 over 16 rows)
template <int MST, int NW, int MODE>
__global__ __launch_bounds__(64 * NW) void stream_kernel(const float *__restrict__ Pu, const _Float16 *__restrict__ image, int U, int I,
                                                         const float *__restrict__ thr, int *__restrict__ counts) {
    extern __shared__ unsigned char smem[];
    constexpr int TB = MST * RS;                 // bytes per staged tile
    constexpr int NT = 64 * NW;
    constexpr int PER = MST * 8 / NT;            // 16-byte pieces per thread and stage (8 per item row)
    static_assert(MST * 8 % NT == 0, "");
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = lane & 31, h = lane >> 5;
    const int u = blockIdx.x * (32 * NW) + wv * 32 + n;
    f16x8 bu[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int t = 0; t < 8; ++t) bu[ks][t] = (_Float16)(u < U ? Pu[(size_t)u * D + 16 * ks + 8 * h + t] : 0.f);
    const float th = u < U ? thr[u] : INFINITY;
    const int nst = (I + MST - 1) / MST;
    float4 st[PER];
    auto gload = [&](int s) {
#pragma unroll
        for (int p = 0; p < PER; ++p) {
            const int f = tid + p * NT;
            const int item = min(s * MST + f / 8, I - 1);
            st[p] = *reinterpret_cast<const float4 *>(reinterpret_cast<const unsigned char *>(image) + (size_t)item * 256 + (f % 8) * 16);
        }
    };
    auto lwrite = [&](unsigned char *buf) {
#pragma unroll
        for (int p = 0; p < PER; ++p) {
            const int f = tid + p * NT;
            *reinterpret_cast<float4 *>(buf + (f / 8) * RS + (f % 8) * 16) = st[p];
        }
    };
    gload(0);
    lwrite(smem);
    if (nst > 1) gload(1);
    __syncthreads();
    int cnt = 0;
    for (int s = 0; s < nst; ++s) {
        unsigned char *buf = smem + (s & 1) * TB;
        if (s + 1 < nst) lwrite(smem + ((s + 1) & 1) * TB);
        if (s + 2 < nst) gload(s + 2);
        if (MODE != 1) {
#pragma unroll
            for (int t = 0; t < MST / 32; ++t) {
                f16x8 a[4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) a[ks] = *reinterpret_cast<const f16x8 *>(buf + (t * 32 + n) * RS + ks * 32 + h * 16);
                f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks], bu[ks], acc, 0, 0, 0);
                if (MODE == 0) {
                    unsigned fails = 0u;
#pragma unroll
                    for (int i = 15; i >= 0; --i) fails = __builtin_amdgcn_alignbit(fails, __float_as_uint(acc[i] - th), 31);
                    const unsigned pm = ~fails & 0xffffu;
                    // rows past I (copies of item I - 1) would be masked here in the real kernel
                    cnt += __popc(pm);
                } else {
                    float m = acc[0];
#pragma unroll
                    for (int i = 1; i < 16; ++i) m = fmaxf(m, acc[i]);
                    cnt += m >= th;
                }
            }
        }
        __syncthreads();
    }
    cnt += __shfl_xor(cnt, 32);
    if (h == 0 && u < U) counts[u] = cnt;
}

template <int MST, int NW, int MODE>
void run(const char *name, const float *Pu, const _Float16 *img, int U, int I, const float *thr, int *counts, int extra_lds) {
    const int shm = 2 * MST * RS + extra_lds;
    CK(hipFuncSetAttribute((const void *)stream_kernel<MST, NW, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, shm));
    const int grid = (U + 32 * NW - 1) / (32 * NW);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((stream_kernel<MST, NW, MODE>), dim3(grid), dim3(64 * NW), shm, 0, Pu, img, U, I, thr, counts);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) best = fminf(best, ms);
    }
    printf("%-44s MST %3d waves %2d lds %6d B: %7.2f ms  (%.0f TFLOP/s of 2 U I d)\n", name, MST, NW, shm, best, 2.0 * U * I * D / best * 1e-9);
}

int main(int argc, char **argv) {
    const int U = argc > 1 ? atoi(argv[1]) : 1000000, I = argc > 2 ? atoi(argv[2]) : 100000;
    std::mt19937 rng(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> hu((size_t)U * D);
    std::vector<_Float16> himg((size_t)I * 2 * D);
    for (auto &x : hu) x = nd(rng);
    for (int i = 0; i < I; ++i)
        for (int k = 0; k < D; ++k) {
            const float x = nd(rng);
            const _Float16 hh = (_Float16)x;
            himg[(size_t)i * 2 * D + k] = hh; himg[(size_t)i * 2 * D + D + k] = (_Float16)(x - (float)hh);
        }
    std::vector<float> hthr(U, 23.0f);            // ~0.998 quantile of N(0, 64): about 2 survivors per 1 024 scores, as the product's stream sees
    float *Pu, *thr; _Float16 *img; int *counts;
    CK(hipMalloc(&Pu, hu.size() * 4)); CK(hipMalloc(&thr, U * 4)); CK(hipMalloc(&img, himg.size() * 2)); CK(hipMalloc(&counts, U * 4));
    CK(hipMemcpy(Pu, hu.data(), hu.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(thr, hthr.data(), U * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(img, himg.data(), himg.size() * 2, hipMemcpyHostToDevice));
    run<128, 8, 0>("sub + alignbit masks", Pu, img, U, I, thr, counts, 0);
    // check a few users on the host (fp16 high pieces, fp32 accumulation in another order: counts may differ by a score on the edge)
    std::vector<int> hc(U);
    CK(hipMemcpy(hc.data(), counts, U * 4, hipMemcpyDeviceToHost));
    long long tot = 0; for (int x : hc) tot += x;
    for (int uu : {0, 1, 31, 32, 255, 256, U - 1}) {
        int c = 0;
        for (int i = 0; i < I; ++i) {
            float s = 0.f;
            for (int k = 0; k < D; ++k) s += (float)(_Float16)hu[(size_t)uu * D + k] * (float)himg[(size_t)i * 2 * D + k];
            c += s >= 23.0f;
        }
        printf("  user %7d: device %4d host %4d\n", uu, hc[uu], c);
    }
    printf("  survivors per 1024 scores: %.3f\n", (double)tot / ((double)U * I) * 1024);
    run<128, 8, 0>("... one workgroup per CU (LDS padded)", Pu, img, U, I, thr, counts, 90 * 1024 - 2 * 128 * RS);
    run<128, 8, 2>("max tree + one compare per tile", Pu, img, U, I, thr, counts, 0);
    run<128, 8, 1>("staging + barriers only", Pu, img, U, I, thr, counts, 0);
    run<256, 8, 0>("sub + alignbit masks", Pu, img, U, I, thr, counts, 0);
    run<128, 16, 0>("sub + alignbit masks, 512 users / WG", Pu, img, U, I, thr, counts, 0);
    run<128, 4, 0>("sub + alignbit masks, 128 users / WG", Pu, img, U, I, thr, counts, 0);
    run<64, 8, 0>("sub + alignbit masks", Pu, img, U, I, thr, counts, 0);
    return 0;
}
