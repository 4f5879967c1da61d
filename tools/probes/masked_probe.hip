// Where does the flag-masked hop spend its 0.5 ms?  Stand-alone model of its access chain on a cfg2-shaped CSR (1 M rows of 32 edges +
// 100 K rows of 320 edges, random columns), one 16-lane group per row, 4 rows per wave, variants that add one dependent stage at a time:
//   0: rowptr pair + 256-B output row store      1: + the row's column loads      2: + one bitmap word per edge (137 KB bitmap)
//   3: as 2, but two rows per group back to back with both rows' loads issued before either is consumed
// hipcc --offload-arch=gfx950 -O3 tools/probes/masked_probe.hip -o tools/probes/masked_probe && tools/probes/masked_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>
template <int V>
__global__ __launch_bounds__(256) void k(int n_rows, const int *__restrict__ rowptr, const int *__restrict__ col, const unsigned *__restrict__ bits,
                                         float *__restrict__ Y) {
    const int lane = threadIdx.x & 63, g = lane >> 4, q = lane & 15;
    const long long grp = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + g;
    constexpr int RPG = (V == 3) ? 2 : 1;
    int b[RPG], e[RPG], row[RPG];
#pragma unroll
    for (int r = 0; r < RPG; ++r) {
        const long long rr = grp * RPG + r;
        row[r] = rr < n_rows ? (int)rr : -1;
        b[r] = e[r] = 0;
        if (row[r] >= 0) { b[r] = rowptr[row[r]]; e[r] = rowptr[row[r] + 1]; }
    }
    unsigned acc = 0;
    if (V >= 1) {
        int mx = 0;
#pragma unroll
        for (int r = 0; r < RPG; ++r) mx = max(mx, e[r] - b[r]);
        mx = max(mx, __shfl_xor(mx, 16)); mx = max(mx, __shfl_xor(mx, 32));
        for (int off = 0; off < mx; off += 32) {
            int c[RPG][2];
#pragma unroll
            for (int r = 0; r < RPG; ++r) {
                const int e0 = b[r] + off + q, e1 = e0 + 16;
                c[r][0] = e0 < e[r] ? col[e0] : -1; c[r][1] = e1 < e[r] ? col[e1] : -1;
            }
#pragma unroll
            for (int r = 0; r < RPG; ++r)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if (V >= 2) { if (c[r][h] >= 0) acc += (bits[c[r][h] >> 5] >> (c[r][h] & 31)) & 1u; }
                    else acc += (unsigned)c[r][h];
                }
        }
    }
#pragma unroll
    for (int r = 0; r < RPG; ++r)
        if (row[r] >= 0) *reinterpret_cast<float4 *>(Y + (size_t)row[r] * 64 + q * 4) = make_float4((float)acc, 0.f, 0.f, 0.f);
}
template <int V> float run(int n_rows, const int *rp, const int *col, const unsigned *bits, float *Y) {
    const int rpg = (V == 3) ? 2 : 1;
    const long long groups = (n_rows + rpg - 1) / rpg;
    const unsigned grid = (unsigned)((groups + 15) / 16);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<V>, dim3(grid), dim3(256), 0, 0, n_rows, rp, col, bits, Y);
    hipEventRecord(a);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k<V>, dim3(grid), dim3(256), 0, 0, n_rows, rp, col, bits, Y);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / 10;
}
int main() {
    const int U = 1000000, I = 100000, N = U + I;
    std::vector<int> rp(N + 1); rp[0] = 0;
    for (int r = 0; r < N; ++r) rp[r + 1] = rp[r] + (r < U ? 32 : 320);
    const long long nnz = rp[N];
    std::vector<int> col(nnz);
    uint64_t s = 88172645463325252ull;
    for (long long e = 0; e < nnz; ++e) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; const bool urow = e < 32ll * U; col[e] = urow ? U + (int)(s % I) : (int)(s % U); }
    int *drp, *dcol; unsigned *dbits; float *Y;
    hipMalloc(&drp, (N + 1) * 4); hipMalloc(&dcol, nnz * 4); hipMalloc(&dbits, ((N + 31) / 32) * 4); hipMalloc(&Y, (size_t)N * 64 * 4);
    hipMemcpy(drp, rp.data(), (N + 1) * 4, hipMemcpyHostToDevice); hipMemcpy(dcol, col.data(), nnz * 4, hipMemcpyHostToDevice);
    hipMemset(dbits, 0, ((N + 31) / 32) * 4);
    printf("rows %d, edges %lld\n", N, nnz);
    printf("0 rowptr + output store            : %.3f ms\n", run<0>(N, drp, dcol, dbits, Y));
    printf("1 + column loads                   : %.3f ms\n", run<1>(N, drp, dcol, dbits, Y));
    printf("2 + bitmap word per edge           : %.3f ms\n", run<2>(N, drp, dcol, dbits, Y));
    printf("3 as 2, two rows per group in flight: %.3f ms\n", run<3>(N, drp, dcol, dbits, Y));
    return 0;
}
