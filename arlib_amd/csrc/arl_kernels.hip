// arl_kernels.hip -- gfx950 (MI355X / CDNA4) kernels behind the C ABI in include/arlib_amd.h.
//
// Everything here is HBM/L2-bound gather/scatter/streaming work on fp32 rows (no MFMA: these are sparse
// gathers, not dense contractions).  Design rules used throughout (cdna_hip_programming.md):
//   * wavefront = 64 lanes; one embedding row of d floats is owned by LPR = d/4 lanes holding a float4
//     each, so a wave moves 64/LPR rows per 1-KiB dwordx4 load instruction (d = 64 -> 4 rows/instr);
//   * no LDS round trips for the SpMM: edge (col,val) pairs are loaded coalesced 64 at a time and
//     broadcast with ds_bpermute (__shfl), partial rows are combined with wave shuffles;
//   * several independent gathers in flight per lane (UNROLL) to cover Infinity-Cache/HBM latency;
//   * >> 256 workgroups per launch, 256 threads (4 waves) each, no inter-workgroup communication.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "arlib_amd.h"
#include <type_traits>

#define ARL_LAUNCH_CHECK()                                  \
    do {                                                    \
        hipError_t e__ = hipGetLastError();                 \
        if (e__ != hipSuccess) return (int)e__;             \
    } while (0)

namespace {

constexpr int kWave = 64;
constexpr int kBlock = 256;
constexpr int kWavesPerBlock = kBlock / kWave;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

__device__ __forceinline__ void fma4(float4 &a, float s, const float4 &x) {
    a.x = fmaf(s, x.x, a.x); a.y = fmaf(s, x.y, a.y); a.z = fmaf(s, x.z, a.z); a.w = fmaf(s, x.w, a.w);
}
__device__ __forceinline__ float4 add4(const float4 &a, const float4 &b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 shfl_xor4(const float4 &a, int off) {
    return make_float4(__shfl_xor(a.x, off), __shfl_xor(a.y, off), __shfl_xor(a.z, off), __shfl_xor(a.w, off));
}

// ================================================================================================
// SpMM  (recommender/LightGCN.py:234 torch.sparse.mm and its backward; A symmetric)
// ================================================================================================
struct CsrDev {
    int n_rows;
    const int32_t *rowptr, *col;
    const float *val;
    int chunk, n_chunks;
    const int32_t *chunk_row, *chunk_begin, *chunk_end;
    int n_long;
    const int32_t *long_row, *long_first, *long_count;
    float *partial;
    const int4 *row_tasks;        // optional (row, begin, end, 0) per row task, in the order the masked hop takes them
};

enum { EPI_AXPBY = 0, EPI_LAYERSUM = 1, EPI_ADAM = 2 };

struct Epi {
    float alpha, beta;
    const float *Z;          // AXPBY / ADAM
    const uint8_t *zflags;   // optional: Z[row] is read only where zflags[row] != 0 (Z is zero elsewhere: sparse batch gradient)
    float *Y;                // AXPBY / LAYERSUM
    const float *S_in;       // LAYERSUM
    float *S;
    float *P, *M, *V;        // ADAM
    float step_size, bc2_sqrt, w1, b2, w2, eps;   // ADAM: lr/(1-b1^t), sqrt(1-b2^t), 1-b1, b2, 1-b2 as torch rounds them (adam_scalars)
    const float *rscale;     // AXPBY, optional: y = alpha * rscale[row] * (A x)[row] + beta * z (a diagonal factor applied to the product)
    int ld;                  // split-row combine only: row stride (floats) of the tables when it is not the kernel's width (0 = dense)
};

// Accumulate sum_e val[e] * X[col[e], 4q..4q+3] over edges [begin,end) for this lane's column quad.
// Lanes are split in G = 64/LPR groups; group g takes edges g, g+G, ... of each 64-edge window.
// Returns the per-lane partial (still split over the G groups).
template <int LPR, int UNROLL, bool NT = false>
__device__ __forceinline__ float4 spmm_gather(const int32_t *__restrict__ col, const float *__restrict__ val, int begin,
                                              int end, const float *__restrict__ X, int d, int lane) {
    constexpr int G = kWave / LPR;
    const int g = lane / LPR, q = lane % LPR;
    const bool qact = (q * 4 < d);
    const float *xq = X + q * 4;
    float4 acc[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int base = begin; base < end; base += kWave) {
        const int e = base + lane;
        int c = 0;
        float v = 0.f;
        if (e < end) {
            if (NT) { c = __builtin_nontemporal_load(col + e); v = __builtin_nontemporal_load(val + e); }
            else { c = col[e]; v = val[e]; }
        }
        const int cnt = min(kWave, end - base);
        for (int j = 0; j < cnt; j += G * UNROLL) {
            int cj[UNROLL];
            float vj[UNROLL];
            float4 x[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                // the shuffles run with every lane active (a bpermute reads 0 from a disabled source lane)
                const int s = j + u * G + g;
                const int cs = __shfl(c, s & (kWave - 1));
                const float vs = __shfl(v, s & (kWave - 1));
                const bool ok = s < cnt;
                cj[u] = ok ? cs : -1;
                vj[u] = ok ? vs : 0.f;
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                x[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (qact && cj[u] >= 0) x[u] = *reinterpret_cast<const float4 *>(xq + (size_t)cj[u] * d);
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) fma4(acc[u], vj[u], x[u]);
        }
    }
#pragma unroll
    for (int u = 1; u < UNROLL; ++u) acc[0] = add4(acc[0], acc[u]);
    return acc[0];
}

template <int LPR>
__device__ __forceinline__ float4 group_reduce(float4 a) {
#pragma unroll
    for (int off = LPR; off < kWave; off <<= 1) a = add4(a, shfl_xor4(a, off));
    return a;
}

template <int MODE>
__device__ __forceinline__ void spmm_epilogue(const Epi &ep, int row, int d, int q, float4 a) {
    const size_t o = (size_t)row * d + q * 4;
    if (MODE == EPI_AXPBY) {
        const float al = ep.rscale ? ep.alpha * ep.rscale[row] : ep.alpha;
        float4 y = make_float4(al * a.x, al * a.y, al * a.z, al * a.w);
        if (ep.Z && (!ep.zflags || ep.zflags[row])) {
            const float4 z = *reinterpret_cast<const float4 *>(ep.Z + o);
            y.x = fmaf(ep.beta, z.x, y.x); y.y = fmaf(ep.beta, z.y, y.y); y.z = fmaf(ep.beta, z.z, y.z); y.w = fmaf(ep.beta, z.w, y.w);
        }
        *reinterpret_cast<float4 *>(ep.Y + o) = y;
    } else if (MODE == EPI_LAYERSUM) {
        const float4 s = *reinterpret_cast<const float4 *>(ep.S_in + o);
        if (ep.Y) *reinterpret_cast<float4 *>(ep.Y + o) = a;
        *reinterpret_cast<float4 *>(ep.S + o) = add4(s, a);
    } else {  // EPI_ADAM: torch/optim/adam.py _single_tensor_adam
        float g[4] = {ep.alpha * a.x, ep.alpha * a.y, ep.alpha * a.z, ep.alpha * a.w};
        if (ep.Z && (!ep.zflags || ep.zflags[row])) {
            const float4 z = *reinterpret_cast<const float4 *>(ep.Z + o);
            g[0] = fmaf(ep.beta, z.x, g[0]); g[1] = fmaf(ep.beta, z.y, g[1]); g[2] = fmaf(ep.beta, z.z, g[2]); g[3] = fmaf(ep.beta, z.w, g[3]);
        }
        float4 p4 = *reinterpret_cast<const float4 *>(ep.P + o);
        float4 m4 = *reinterpret_cast<const float4 *>(ep.M + o);
        float4 v4 = *reinterpret_cast<const float4 *>(ep.V + o);
        float p[4] = {p4.x, p4.y, p4.z, p4.w}, m[4] = {m4.x, m4.y, m4.z, m4.w}, v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            m[k] = m[k] + (g[k] - m[k]) * ep.w1;
            v[k] = v[k] * ep.b2 + ep.w2 * g[k] * g[k];
            const float denom = sqrtf(v[k]) / ep.bc2_sqrt + ep.eps;
            p[k] = p[k] - ep.step_size * (m[k] / denom);
        }
        *reinterpret_cast<float4 *>(ep.P + o) = make_float4(p[0], p[1], p[2], p[3]);
        *reinterpret_cast<float4 *>(ep.M + o) = make_float4(m[0], m[1], m[2], m[3]);
        *reinterpret_cast<float4 *>(ep.V + o) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// One wavefront per task.  Tasks [0, n_chunks) are slices of long rows (heavy; dispatched first) whose
// partial sums go to A.partial; tasks [n_chunks, n_chunks + n_rows) are whole rows (long rows skip).
template <int LPR, int MODE>
__global__ __launch_bounds__(kBlock) void spmm_rows_kernel(CsrDev A, const float *__restrict__ X, int d, Epi ep) {
    const int lane = threadIdx.x & (kWave - 1);
    const long long task = (long long)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const int q = lane % LPR;
    if (task < A.n_chunks) {
        const int t = (int)task;
        float4 a = spmm_gather<LPR, 4>(A.col, A.val, A.chunk_begin[t], A.chunk_end[t], X, d, lane);
        a = group_reduce<LPR>(a);
        if (lane < LPR && q * 4 < d) *reinterpret_cast<float4 *>(A.partial + (size_t)t * d + q * 4) = a;
        return;
    }
    const long long r = task - A.n_chunks;
    if (r >= A.n_rows) return;
    const int begin = A.rowptr[r], end = A.rowptr[r + 1];
    if (A.n_chunks > 0 && end - begin > A.chunk) return;       // long row: summed by spmm_long_rows_kernel
    float4 a = spmm_gather<LPR, 4>(A.col, A.val, begin, end, X, d, lane);
    a = group_reduce<LPR>(a);
    if (lane < LPR && q * 4 < d) spmm_epilogue<MODE>(ep, (int)r, d, q, a);
}

// One wavefront per long row: add its chunk partials in slot order (deterministic) and run the epilogue.
template <int LPR, int MODE>
__global__ __launch_bounds__(kBlock) void spmm_long_rows_kernel(CsrDev A, int d, Epi ep) {
    constexpr int G = kWave / LPR;
    const int lane = threadIdx.x & (kWave - 1);
    const int t = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (t >= A.n_long) return;
    const int g = lane / LPR, q = lane % LPR;
    const int first = A.long_first[t], cnt = A.long_count[t];
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q * 4 < d)
        for (int s = g; s < cnt; s += G) a = add4(a, *reinterpret_cast<const float4 *>(A.partial + (size_t)(first + s) * d + q * 4));
    a = group_reduce<LPR>(a);
    if (lane < LPR && q * 4 < d) spmm_epilogue<MODE>(ep, A.long_row[t], ep.ld ? ep.ld : d, q, a);      // ep.ld: the tables' row stride when it is not d
}


// ---- masked hop: the operand X is zero except on rows whose bit is set in `xbits` (the batch gradient G: <= 3B rows).
// The kernel is bound by its dependent-load chain (rowptr -> col -> flag bit -> G row) times the number of waves, not by
// bandwidth, so every LPR-lane group of a wave owns its own row (4 rows per wave at d = 64): 4x fewer waves than wave-per-row,
// no cross-group reduction; each group walks its edge list LPR edges at a time, tests one bit per edge and only gathers (and
// only loads `val` for) the flagged ones.
template <int LPR, int MODE>
__global__ __launch_bounds__(kBlock) void spmm_rows_masked_gpr_kernel(CsrDev A, const float *__restrict__ X, int d, Epi ep,
                                                                        const uint32_t *__restrict__ xbits) {
    constexpr int G = kWave / LPR;
    const int lane = threadIdx.x & (kWave - 1);
    const int g = lane / LPR, q = lane % LPR;
    const long long task = ((long long)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6)) * G + g;
    const long long total = (long long)A.n_chunks + A.n_rows;
    int begin = 0, end = 0, row = -1, slot = -1;
    if (task < A.n_chunks) {
        slot = (int)task; begin = A.chunk_begin[slot]; end = A.chunk_end[slot];
    } else if (task < total) {
        if (A.row_tasks) {                               // rows sorted by length: the wave's four rows are equally long, one 16-B record each
            const int4 rt = A.row_tasks[task - A.n_chunks];
            row = rt.x; begin = rt.y; end = rt.z;
        } else {
            row = (int)(task - A.n_chunks);
            begin = A.rowptr[row]; end = A.rowptr[row + 1];
        }
        if (A.n_chunks > 0 && end - begin > A.chunk) { end = begin; row = -1; }     // long row: summed by spmm_long_rows_kernel
    }
    const bool qact = (q * 4 < d);
    const float *xq = X + q * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    // two LPR-edge windows per trip, their column loads and bit tests issued back to back: a 32-edge user row is one dependent
    // load -> load -> ballot chain instead of two
    for (int base = begin; __any(base < end); base += 2 * LPR) {
        const int e0 = base + q, e1 = base + LPR + q;
        int c0 = 0, c1 = 0;
        if (e0 < end) c0 = A.col[e0];
        if (e1 < end) c1 = A.col[e1];
        bool f0 = false, f1 = false;
        if (e0 < end) f0 = (xbits[c0 >> 5] >> (c0 & 31)) & 1u;                     // 1 bit per node: the bitmap (137 KB at cfg2) stays in L1/L2
        if (e1 < end) f1 = (xbits[c1 >> 5] >> (c1 & 31)) & 1u;
        const unsigned long long wm0 = __ballot(f0), wm1 = __ballot(f1);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const unsigned long long wm = half ? wm1 : wm0;
            const int c = half ? c1 : c0;
            unsigned long long mask = (wm >> (g * LPR)) & ((LPR == 64) ? ~0ull : ((1ull << (LPR & 63)) - 1ull));      // 64 bits: LPR = 64 at d = 256
            while (mask) {
                const int j = __ffsll((long long)mask) - 1;
                mask &= mask - 1ull;
                const int cj = __shfl(c, g * LPR + j);
                const float vj = A.val[base + half * LPR + j];
                if (qact) fma4(acc, vj, *reinterpret_cast<const float4 *>(xq + (size_t)cj * d));
            }
        }
    }
    if (!qact) return;
    if (slot >= 0) *reinterpret_cast<float4 *>(A.partial + (size_t)slot * d + q * 4) = acc;
    else if (row >= 0) spmm_epilogue<MODE>(ep, row, d, q, acc);
}

// ---- row-subset SpMM: only the listed rows are produced (the last forward hop is consumed on the <= 3B batch rows only).
// Each listed row is cut into `nsplit` equal edge ranges (a 100k-edge popular item must not serialise one wave); partials are
// combined in split order by subset_finish_kernel, which also adds the earlier layers' rows: out_c[t] = alpha * (sum_k layer_k[r] + (A X)[r]).
struct LayerPtrs { const float *p[8]; int n; };

template <int LPR>
__global__ __launch_bounds__(kBlock) void spmm_subset_kernel(CsrDev A, const float *__restrict__ X, int d, const int32_t *__restrict__ rows, int n,
                                                              int nsplit, float *__restrict__ partial) {
    const int lane = threadIdx.x & (kWave - 1);
    const long long task = (long long)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (task >= (long long)n * nsplit) return;
    const int t = (int)(task / nsplit), sp = (int)(task % nsplit);
    const int r = rows[t];
    const int begin = A.rowptr[r], end = A.rowptr[r + 1];
    const int per = (end - begin + nsplit - 1) / nsplit;
    const int b = begin + sp * per, e = min(end, b + per);
    const int q = lane % LPR;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (b < e) a = spmm_gather<LPR, 4>(A.col, A.val, b, e, X, d, lane);
    a = group_reduce<LPR>(a);
    if (lane < LPR && q * 4 < d) *reinterpret_cast<float4 *>(partial + (size_t)task * d + q * 4) = a;
}

template <int LPR>
__global__ __launch_bounds__(kBlock) void subset_finish_kernel(const float *__restrict__ partial, int n, int nsplit, int d, const int32_t *__restrict__ rows,
                                                                LayerPtrs L, float alpha, float *__restrict__ out_c, const float *__restrict__ row_weight) {
    constexpr int G = kWave / LPR;
    const int lane = threadIdx.x & (kWave - 1);
    const int g = lane / LPR, q = lane % LPR;
    const int t = (blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6)) * G + g;
    if (t >= n || q * 4 >= d) return;
    const int r = rows[t];
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < L.n; ++k) a = add4(a, *reinterpret_cast<const float4 *>(L.p[k] + (size_t)r * d + q * 4));
    for (int s = 0; s < nsplit; ++s) a = add4(a, *reinterpret_cast<const float4 *>(partial + ((size_t)t * nsplit + s) * d + q * 4));
    const float w = row_weight ? alpha * row_weight[t] : alpha;             // optional per-listed-row factor (user-sharded batch: 1 on the owner, 0 elsewhere)
    *reinterpret_cast<float4 *>(out_c + (size_t)t * d + q * 4) = make_float4(w * a.x, w * a.y, w * a.z, w * a.w);
}

// out = alpha * sum_k layers[k]  (element-wise over up to 8 equally shaped tables; the LightGCN mean over layers, LightGCN.py:236-240,
// in ONE pass instead of a clone + L adds + a division)
__global__ __launch_bounds__(kBlock) void tables_sum_kernel(LayerPtrs L, long long n4, float alpha, float4 *__restrict__ out) {
    for (long long t = (long long)blockIdx.x * kBlock + threadIdx.x; t < n4; t += (long long)gridDim.x * kBlock) {
        float4 a = reinterpret_cast<const float4 *>(L.p[0])[t];
        for (int k = 1; k < L.n; ++k) a = add4(a, reinterpret_cast<const float4 *>(L.p[k])[t]);
        out[t] = make_float4(alpha * a.x, alpha * a.y, alpha * a.z, alpha * a.w);
    }
}

__global__ __launch_bounds__(kBlock) void mark_rows_kernel(uint8_t *__restrict__ flags, const int32_t *__restrict__ idx, int n, int value) {
    const int t = blockIdx.x * kBlock + threadIdx.x;
    if (t < n) flags[idx[t]] = (uint8_t)value;
}
__global__ __launch_bounds__(kBlock) void mark_bits_kernel(uint32_t *__restrict__ bits, const int32_t *__restrict__ idx, int n, int set) {
    const int t = blockIdx.x * kBlock + threadIdx.x;
    if (t >= n) return;
    const int i = idx[t];
    if (set) atomicOr(bits + (i >> 5), 1u << (i & 31));
    else atomicAnd(bits + (i >> 5), ~(1u << (i & 31)));
}
__global__ __launch_bounds__(kBlock) void zero_rows_kernel(float *__restrict__ dst, const int32_t *__restrict__ idx, int n, int d) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (t >= n) return;
    float *o = dst + (size_t)idx[t] * d;
    for (int k = lane; k < d; k += kWave) o[k] = 0.f;
}

// ---- ordered accumulation (no float atomics): dst[idx[t]] += scale * src[t], duplicates summed in ascending t.
// One wave per contribution t of the window [c0, c1).  The wave scans the window's index list 256 entries per trip (coalesced; the list is a
// few KB and stays in L1): if an EARLIER entry names the same row the wave exits -- the first entry of a row is its owner -- otherwise it
// adds the row's contributions one after the other in index order, starting from the value already in dst.  That is the association of CPU
// index_put_(accumulate=True), i.e. of the reference's gathers under autograd (recommender/LightGCN.py:51-56, util/loss.py:5-9), and it
// is bit-identical from run to run.  Work is O(n^2 / 64) wave-trips, so a launch covers at most kOrderedWindow entries; longer lists run
// as successive launches (stream order keeps the result deterministic).
constexpr int kOrderedWindow = 16384;

// Scan of the window by one wave: `on_hit(tt)` runs, in ascending tt, for every entry tt whose row (`rowof(tt)`) equals `row`.  Returns false --
// the wave must exit -- as soon as an entry BEFORE t names the row (that entry's wave owns it).
template <class RowOf, class OnHit>
__device__ __forceinline__ bool ordered_scan(int c0, int c1, int t, long long row, int lane, RowOf rowof, OnHit on_hit) {
    for (int base = c0; base < c1; base += 256) {
        long long v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int j = base + 64 * i + lane;
            v[i] = j < c1 ? (long long)rowof(j) : -1ll;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned long long m = __ballot(v[i] == row);
            const int b0 = base + 64 * i;
            if (b0 < t) {
                const int lim = t - b0;
                if (lim >= 64 ? m != 0ull : (m & ((1ull << lim) - 1ull)) != 0ull) return false;
            }
            while (m) {
                const int tt = b0 + __ffsll((long long)m) - 1;
                m &= m - 1ull;
                on_hit(tt);
            }
        }
    }
    return true;
}

// The same scan over a copy of the window's rows in LDS (staged once per workgroup by stage_rows): one dependent global-memory latency per
// 256 entries becomes an LDS read -- the scan of a 6 144-entry batch list drops from ~10 us to ~1 us per wave.
template <class OnHit>
__device__ __forceinline__ bool ordered_scan_lds(const int32_t *srows, int c0, int c1, int t, int row, int lane, OnHit on_hit) {
    const int n = c1 - c0, tl = t - c0;
    for (int base = 0; base < n; base += 512) {
        int v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int j = base + 64 * i + lane;
            v[i] = j < n ? srows[j] : -1;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            unsigned long long m = __ballot(v[i] == row);
            const int b0 = base + 64 * i;
            if (b0 < tl) {
                const int lim = tl - b0;
                if (lim >= 64 ? m != 0ull : (m & ((1ull << lim) - 1ull)) != 0ull) return false;
            }
            while (m) {
                const int tt = c0 + b0 + __ffsll((long long)m) - 1;
                m &= m - 1ull;
                on_hit(tt);
            }
        }
    }
    return true;
}
// group form: `on_group(m, g0)` receives the 64-bit hit mask of entries g0 .. g0 + 63 (absolute indices), groups in ascending order
template <class OnGroup>
__device__ __forceinline__ bool ordered_scan_lds_groups(const int32_t *srows, int c0, int c1, int t, int row, int lane, OnGroup on_group) {
    const int n = c1 - c0, tl = t - c0;
    for (int base = 0; base < n; base += 512) {
        int v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int j = base + 64 * i + lane;
            v[i] = j < n ? srows[j] : -1;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const unsigned long long m = __ballot(v[i] == row);
            const int b0 = base + 64 * i;
            if (b0 < tl) {
                const int lim = tl - b0;
                if (lim >= 64 ? m != 0ull : (m & ((1ull << lim) - 1ull)) != 0ull) return false;
            }
            if (m) on_group(m, c0 + b0);
        }
    }
    return true;
}
template <class RowOf>
__device__ __forceinline__ void stage_rows(int32_t *srows, int c0, int c1, RowOf rowof) {
    for (int j = threadIdx.x; j < c1 - c0; j += kBlock) srows[j] = (int32_t)rowof(c0 + j);
    __syncthreads();
}

// Which rows does the list name more than once?  One thread per entry sets the row's bit in `bits`; an entry that finds it already set
// sets the row's bit in `dup`.  WHICH entry arrives second is a race, the resulting SET of duplicated rows is not.  With it the accumulation
// kernel below scans the list only for the duplicated rows (a few hundred of a 6 144-row batch) -- 44 -> ~10 us per call at cfg2.
__global__ __launch_bounds__(kBlock) void rows_mark_dups_kernel(uint32_t *__restrict__ bits, uint32_t *__restrict__ dup, const int32_t *__restrict__ idx, int n,
                                                                 const float *__restrict__ row_scale) {
    const int t = blockIdx.x * kBlock + threadIdx.x;
    if (t >= n) return;
    if (row_scale && row_scale[t] == 0.f) return;                     // an entry with factor 0 is ABSENT (see rows_add_ordered_kernel)
    const int row = idx[t];
    const uint32_t m = 1u << (row & 31);
    if (atomicOr(bits + (row >> 5), m) & m) atomicOr(dup + (row >> 5), m);
}

// MARK: also set the row's byte flag and bitmap bit (the sparse-batch step touches three node-sized arrays with one index list: once to
// add the batch gradients and mark the rows, once to clear everything again; one launch each)
template <bool MARK>
__global__ __launch_bounds__(kBlock) void rows_add_ordered_kernel(float *__restrict__ dst, uint8_t *__restrict__ flags, uint32_t *__restrict__ bits,
                                                                   const int32_t *__restrict__ idx, int c0, int c1, int d,
                                                                   const float *__restrict__ src, float scale, const float *__restrict__ row_scale,
                                                                   const uint32_t *__restrict__ dup) {
    extern __shared__ int32_t srows[];
    const int lane = threadIdx.x & 63;
    const int t = c0 + blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    // An entry whose per-contribution factor is exactly 0 is ABSENT: it adds nothing, marks nothing and never takes part in a scan.  The user-sharded
    // step keeps static shapes by giving every foreign sample factor 0 on a clamped local row (shard_batch_prep_kernel): at 8 ranks ~7/8 of the
    // batch's user entries would otherwise pile up on two rows -- hundreds of dependent add trips of zeros on one wave, and two marked rows more
    const bool valid = t < c1 && !(row_scale && row_scale[t] == 0.f);
    const int row = valid ? idx[t] : 0;
    float *o = dst + (size_t)row * d;
    const bool fast = valid && dup && !((dup[row >> 5] >> (row & 31)) & 1u);
    if (fast) {                                                       // the list names this row once (rows_mark_dups_kernel): no scan, no order to keep
        const float sc = row_scale ? scale * row_scale[t] : scale;
        const float *s = src + (size_t)t * d;
        for (int k = lane; k < d; k += kWave) {
#pragma clang fp contract(off)
            const float c = sc * s[k];
            o[k] = o[k] + c;
        }
        if (MARK && lane == 0) flags[row] = 1;
    }
    if (!__syncthreads_or(valid && !fast)) return;                    // nothing to scan in this workgroup
    stage_rows(srows, c0, c1, [&](int j) { return (row_scale && row_scale[j] == 0.f) ? -1 : idx[j]; });      // absent entries match no row
    if (!valid || fast) return;
    for (int k0 = 0; k0 < d; k0 += 256) {
        float acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { const int k = k0 + 64 * i + lane; acc[i] = k < d ? o[k] : 0.f; }
        // hits of one 64-entry group arrive together (`m`): their rows are loaded four at a time and added in index order -- a row the batch
        // names 60 times (a popular item) is otherwise 60 dependent load -> add round trips on one wave, the kernel's tail
        const bool owner = ordered_scan_lds_groups(srows, c0, c1, t, row, lane, [&](unsigned long long m, int g0) {
#pragma clang fp contract(off)                       // index_put_ adds the ROUNDED product scale * src: no fused multiply-add here
            while (m) {
                int tt[4]; float x[4][4], sc[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    tt[q] = -1;
                    if (m) { tt[q] = g0 + __ffsll((long long)m) - 1; m &= m - 1ull; }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (tt[q] < 0) continue;
                    const float *s = src + (size_t)tt[q] * d;
                    sc[q] = row_scale ? scale * row_scale[tt[q]] : scale;     // optional per-contribution factor (sharded batch: 1 for own samples, 0 for foreign ones)
#pragma unroll
                    for (int i = 0; i < 4; ++i) { const int k = k0 + 64 * i + lane; x[q][i] = k < d ? s[k] : 0.f; }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (tt[q] < 0) continue;
#pragma unroll
                    for (int i = 0; i < 4; ++i) { const float c = sc[q] * x[q][i]; acc[i] = acc[i] + c; }
                }
            }
        });
        if (!owner) return;
#pragma unroll
        for (int i = 0; i < 4; ++i) { const int k = k0 + 64 * i + lane; if (k < d) o[k] = acc[i]; }
    }
    if (MARK && lane == 0) {
        flags[row] = 1;
        atomicOr(bits + (row >> 5), 1u << (row & 31));
    }
}
__global__ __launch_bounds__(kBlock) void batch_rows_clear_kernel(float *__restrict__ G, uint8_t *__restrict__ flags, uint32_t *__restrict__ bits,
                                                                   const int32_t *__restrict__ idx, int n, int d, uint32_t *__restrict__ dup) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (t >= n) return;
    const int row = idx[t];
    float *o = G + (size_t)row * d;
    for (int k = lane; k < d; k += kWave) o[k] = 0.f;
    if (lane == 0) {
        flags[row] = 0;
        atomicAnd(bits + (row >> 5), ~(1u << (row & 31)));
        if (dup) atomicAnd(dup + (row >> 5), ~(1u << (row & 31)));
    }
}

// ================================================================================================
// L2-blocked ("tiled") SpMM.  The row-per-wave kernel above moves one 256-B row per edge through the Infinity-Cache fabric
// (~8 TB/s).  Gathers that hit the XCD's 4 MB L2 run 2.2x faster (tools/l2_gather_bench.py: 15.8 TB/s from a 2 MB table), so
// this form makes the gathered rows L2-resident:
//   * columns are cut into blocks of Tc rows (Tc*4d bytes ~ 2 MB);
//   * every workgroup (one per CU, persistent over `n_sweeps` sweeps) owns a BIN of up to `cap` output rows whose fp32
//     accumulators live in LDS for a whole sweep, and walks the column blocks in ascending order -- all CUs of an XCD are on
//     the same column block at about the same time (bins carry equal numbers of edges), so each gathered row is fetched from
//     the fabric once per XCD and then served from L2 to every bin that needs it;
//   * a bin's edges are stored sorted by (column block, local row, col); a 16-lane group accumulates a run of edges of one
//     row in registers and flushes it to the LDS accumulator with ds_add_f32 (runs from different waves commute);
//   * the epilogue (alpha*AX + beta*Z, or the fused Adam update) runs when a sweep's accumulators are written out.
// No inter-workgroup communication: XCD co-scheduling only affects speed, never correctness.
struct TiledDev {
    int n_sweeps, n_slots, cap, n_cb, n_groups;   // bins = n_sweeps * n_slots; n_cb column blocks; n_groups = 16 waves * (64/LPR)
    const int32_t *bin_rows;                   // [bins][cap] global row id or -1
    const int32_t *seg_ptr;                    // [bins * n_groups + 1] edge offsets of the (bin, owner group) lists
    const int32_t *e_col;                      // [nnz] column (global row of X)
    const float *e_val;                        // [nnz]
    const uint16_t *e_row;                     // [nnz] local row inside the bin
};

constexpr int kTiledThreads = 1024;
constexpr int kTiledWaves = kTiledThreads / kWave;

// Every LPR-lane group OWNS a fixed subset of the bin's local rows for the whole sweep and only ever touches the edges of
// its own rows: the bin's edges are sorted by (owner group, column block, local row, column), so a group streams ONE
// contiguous edge list per sweep (prefetched a round ahead) whose order walks the column blocks in ascending order.  Groups
// hold equal numbers of edges (snake deal by degree), so all groups of all CUs cross the column blocks at about the same
// time -- that is what keeps the gathered rows L2-resident -- without any barrier or per-block pointer.  LDS accumulators
// are updated with plain read-add-write (exclusive ownership): no atomics, deterministic.
template <int LPR, int MODE>
__global__ __launch_bounds__(kTiledThreads) void spmm_tiled_kernel(TiledDev T, const float *__restrict__ X, int d, Epi ep) {
    constexpr int G = kWave / LPR;
    extern __shared__ float acc_lds[];                        // [cap][ld]
    const int ld = d + 4;                                     // row stride: consecutive rows start 4 banks apart
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int g = lane / LPR, q = lane % LPR;
    const int og = wv * G + g;                                // owner-group id inside the workgroup
    const bool qact = q * 4 < d;
    const float *xq = X + q * 4;
    for (int sweep = 0; sweep < T.n_sweeps; ++sweep) {
        const int bin = sweep * T.n_slots + blockIdx.x;
        for (int t = tid; t < T.cap * ld; t += kTiledThreads) acc_lds[t] = 0.f;
        const int lo = T.seg_ptr[(size_t)bin * T.n_groups + og], hi = T.seg_ptr[(size_t)bin * T.n_groups + og + 1];
        __syncthreads();
        // prefetch round 0
        int c_n = 0, r_n = -1;
        float v_n = 0.f;
        if (lo + q < hi) { c_n = T.e_col[lo + q]; v_n = T.e_val[lo + q]; r_n = T.e_row[lo + q]; }
        for (int base = lo; __any(base < hi); base += LPR) {
            const int c = c_n, rloc = r_n;
            const float v = v_n;
            const int e2 = base + LPR + q;                     // next round's edge of this lane, in flight during this round
            c_n = 0; v_n = 0.f; r_n = -1;
            if (e2 < hi) { c_n = T.e_col[e2]; v_n = T.e_val[e2]; r_n = T.e_row[e2]; }
            const int n = min(LPR, hi - base);                 // this group's edges in this round (<= 0: none)
            float4 x[LPR];
#pragma unroll
            for (int j = 0; j < LPR; ++j) {                    // all gathers of the round in flight together
                const int cj = __shfl(c, g * LPR + j);
                x[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (qact && j < n) x[j] = *reinterpret_cast<const float4 *>(xq + (size_t)cj * d);
            }
            float4 run = make_float4(0.f, 0.f, 0.f, 0.f);
            int cur = -1;
#pragma unroll
            for (int j = 0; j < LPR; ++j) {
                const float vj = __shfl(v, g * LPR + j);
                const int rj = __shfl(rloc, g * LPR + j);
                if (j < n) {
                    if (rj != cur) {
                        if (cur >= 0 && qact) { float4 *a = reinterpret_cast<float4 *>(acc_lds + cur * ld + q * 4); *a = add4(*a, run); }
                        cur = rj;
                        run = make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                    fma4(run, vj, x[j]);
                }
            }
            if (cur >= 0 && qact) { float4 *a = reinterpret_cast<float4 *>(acc_lds + cur * ld + q * 4); *a = add4(*a, run); }
        }
        __syncthreads();
        // write-out with the epilogue: one LPR-lane group per row of the bin
        const int32_t *rows = T.bin_rows + (size_t)bin * T.cap;
        for (int i = og; i < T.cap; i += kTiledWaves * G) {
            const int r = rows[i];
            if (r >= 0 && qact) {
                const float4 a = *reinterpret_cast<const float4 *>(acc_lds + i * ld + q * 4);
                spmm_epilogue<MODE>(ep, r, d, q, a);
            }
        }
        __syncthreads();
    }
}

template <int MODE>
int launch_spmm_tiled(const arl_tiled *T, const float *X, int64_t d, const Epi &ep, hipStream_t st) {
    if (!T || !X || !T->bin_rows || !T->seg_ptr || !T->e_col || !T->e_val || !T->e_row) return ARL_E_NULL;
    if (d <= 0 || d > 256 || (d & 3)) return ARL_E_DIM;
    if (T->n_sweeps < 0 || T->n_slots <= 0 || T->cap <= 0 || T->cap > 65535 || T->n_cb <= 0) return ARL_E_ARG;
    {
        const int64_t lpr = d <= 16 ? 4 : d <= 32 ? 8 : d <= 64 ? 16 : d <= 128 ? 32 : 64;
        if (T->n_groups != kTiledWaves * (kWave / lpr)) return ARL_E_ARG;      // the plan is built for one embedding width class
    }
    if (T->n_sweeps == 0) return ARL_OK;
    TiledDev D;
    D.n_sweeps = (int)T->n_sweeps; D.n_slots = (int)T->n_slots; D.cap = (int)T->cap; D.n_cb = (int)T->n_cb; D.n_groups = (int)T->n_groups;
    D.bin_rows = T->bin_rows; D.seg_ptr = T->seg_ptr; D.e_col = T->e_col; D.e_val = T->e_val; D.e_row = T->e_row;
    const size_t shm = sizeof(float) * (size_t)D.cap * (size_t)(d + 4);
    if (shm > 160 * 1024) return ARL_E_ARG;
    const int di = (int)d;
#define ARL_TILED_CASE(LPRV)                                                                                                   \
    do {                                                                                                                       \
        hipError_t e1 = hipFuncSetAttribute((const void *)spmm_tiled_kernel<LPRV, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); \
        if (e1 != hipSuccess) return (int)e1;                                                                                  \
        hipLaunchKernelGGL((spmm_tiled_kernel<LPRV, MODE>), dim3((unsigned)D.n_slots), dim3(kTiledThreads), shm, st, D, X, di, ep); \
    } while (0)
    if (d <= 16) ARL_TILED_CASE(4);
    else if (d <= 32) ARL_TILED_CASE(8);
    else if (d <= 64) ARL_TILED_CASE(16);
    else if (d <= 128) ARL_TILED_CASE(32);
    else ARL_TILED_CASE(64);
#undef ARL_TILED_CASE
    ARL_LAUNCH_CHECK();
    return ARL_OK;
}

template <int MODE>
int launch_spmm(const arl_csr *A, const float *X, int64_t d, const Epi &ep, hipStream_t st, const uint32_t *xflags = nullptr) {
    if (!A || !X || (A->n_rows > 0 && !A->rowptr) || (A->nnz > 0 && (!A->col || !A->val))) return ARL_E_NULL;
    if (d <= 0 || d > 256 || (d & 3)) return ARL_E_DIM;
    if (A->n_rows < 0 || A->n_rows > 0x7fffffffll || A->nnz > 0x7fffffffll || A->n_chunks > 0x7fffffffll) return ARL_E_RANGE;
    if (A->n_chunks > 0 && (!A->chunk_row || !A->chunk_begin || !A->chunk_end || !A->partial || A->chunk <= 0)) return ARL_E_NULL;
    if (A->n_long > 0 && (!A->long_row || !A->long_first || !A->long_count)) return ARL_E_NULL;
    if ((A->n_chunks > 0) != (A->n_long > 0)) return ARL_E_ARG;
    if (A->n_rows == 0 && A->n_chunks == 0) return ARL_OK;       // n_rows == 0 with a chunk plan = "long rows only" (hub pass of the tiled SpMM)
    CsrDev D;
    D.n_rows = (int)A->n_rows; D.rowptr = A->rowptr; D.col = A->col; D.val = A->val;
    D.chunk = A->chunk; D.n_chunks = (int)A->n_chunks; D.chunk_row = A->chunk_row; D.chunk_begin = A->chunk_begin;
    D.chunk_end = A->chunk_end; D.n_long = (int)A->n_long; D.long_row = A->long_row; D.long_first = A->long_first;
    D.long_count = A->long_count; D.partial = A->partial;
    D.row_tasks = xflags ? reinterpret_cast<const int4 *>(A->row_tasks) : nullptr;
    const long long tasks = (long long)D.n_rows + D.n_chunks;
    const unsigned grid = (unsigned)((tasks + kWavesPerBlock - 1) / kWavesPerBlock);
    const unsigned grid_long = (unsigned)((D.n_long + kWavesPerBlock - 1) / kWavesPerBlock);
    const int di = (int)d;
#define ARL_SPMM_CASE(LPRV)                                                                              \
    do {                                                                                                 \
        if (xflags) hipLaunchKernelGGL((spmm_rows_masked_gpr_kernel<LPRV, MODE>), dim3((unsigned)((tasks + kWavesPerBlock * (kWave / LPRV) - 1) / (kWavesPerBlock * (kWave / LPRV)))), dim3(kBlock), 0, st, D, X, di, ep, xflags); \
        else hipLaunchKernelGGL((spmm_rows_kernel<LPRV, MODE>), dim3(grid), dim3(kBlock), 0, st, D, X, di, ep); \
        ARL_LAUNCH_CHECK();                                                                              \
        if (D.n_long > 0) {                                                                              \
            hipLaunchKernelGGL((spmm_long_rows_kernel<LPRV, MODE>), dim3(grid_long), dim3(kBlock), 0, st, D, di, ep); \
            ARL_LAUNCH_CHECK();                                                                          \
        }                                                                                                \
    } while (0)
    if (d <= 16) ARL_SPMM_CASE(4);
    else if (d <= 32) ARL_SPMM_CASE(8);
    else if (d <= 64) ARL_SPMM_CASE(16);
    else if (d <= 128) ARL_SPMM_CASE(32);
    else ARL_SPMM_CASE(64);
#undef ARL_SPMM_CASE
    return ARL_OK;
}

// ================================================================================================
// Register-blocked SpMM (d = 64): a wave owns up to RPW output rows for the whole kernel, their accumulators in registers
// (lane = column), and consumes one flat record stream sorted by (column block, row slot).  All waves of a launch are resident
// together, carry the same number of edges (the plan deals rows longest-first to the least loaded wave) and sweep the column
// blocks in the same order, so the operand rows a wave gathers were usually just brought into the L2 by its neighbours.
// A record = (col | slot << 24, val), fetched 64 at a time by one coalesced load and broadcast with v_readlane; the accumulator
// is selected with the wave-uniform slot (M0-relative register addressing): two vector instructions per edge, no LDS, no
// atomics, deterministic.  Rows longer than the plan's hub threshold are not in the plan (chunked CSR kernel).
// ================================================================================================
typedef float f32x32v __attribute__((ext_vector_type(32)));
#ifndef ARL_SPMM_EXP_NOVAL
#define ARL_SPMM_EXP_NOVAL 0                     // 1 = developer timing probe of a value-free record stream (`make variant`); never the product library
#endif

struct BlockedDev {
    int n_waves;
    const int32_t *wave_ptr;      // [n_waves + 1] record offsets (multiples of 64)
    const int32_t *wave_rows;     // [n_waves][RPW] output row or -1
    const int32_t *rec_col;       // col | slot << 24
    const float *rec_val;
    float *partial;               // [n_pieces][d] raw sums of split-row pieces (wave_rows entry -(2 + t))
};

// spmm_epilogue for CPL adjacent columns per lane (d = 64 * CPL); same arithmetic (a wave writes one 256-B / 512-B row at a time)
template <int MODE, int CPL, int LD>
__device__ __forceinline__ void spmm_epilogue1(const Epi &ep, int row, int lane, const float *a) {
    const size_t o = (size_t)row * LD + lane * CPL;
    const bool zr = ep.Z && (!ep.zflags || ep.zflags[row]);
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
        if (MODE == EPI_AXPBY) {
            float y = (ep.rscale ? ep.alpha * ep.rscale[row] : ep.alpha) * a[c];
            if (zr) y = fmaf(ep.beta, ep.Z[o + c], y);
            __builtin_nontemporal_store(y, ep.Y + o + c);
        } else if (MODE == EPI_LAYERSUM) {
            const float sv = ep.S_in[o + c];
            if (ep.Y) ep.Y[o + c] = a[c];
            ep.S[o + c] = sv + a[c];
        } else {
            float g = ep.alpha * a[c];
            if (zr) g = fmaf(ep.beta, ep.Z[o + c], g);
            float p = ep.P[o + c], m = ep.M[o + c], v = ep.V[o + c];
            m = m + (g - m) * ep.w1;
            v = v * ep.b2 + ep.w2 * g * g;
            const float denom = sqrtf(v) / ep.bc2_sqrt + ep.eps;
            p = p - ep.step_size * (m / denom);
            ep.P[o + c] = p; ep.M[o + c] = m; ep.V[o + c] = v;
        }
    }
}

template <int RPW, int MODE, int UNR, int CPL, int LD = 64 * CPL>       // LD: row stride (floats) of the operand and of every table the epilogue touches
__global__ __launch_bounds__(kBlock) void spmm_blocked64_kernel(BlockedDev P, const float *__restrict__ X, Epi ep) {
    static_assert(RPW == 16 || RPW == 32, "accumulators are 32-register vectors");
    static_assert(CPL == 1 || CPL == 2, "d = 64 (one column per lane) or d = 128 (two adjacent columns per lane)");
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (w >= P.n_waves) return;
    f32x32v acc[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c)
#pragma unroll
        for (int r = 0; r < 32; ++r) acc[c][r] = 0.f;
    const int begin = P.wave_ptr[w], end = P.wave_ptr[w + 1];
    const float *xl = X + lane * CPL;
    int rc = 0; float rv = 0.f;
    if (begin < end) { rc = __builtin_nontemporal_load(P.rec_col + begin + lane); if (!ARL_SPMM_EXP_NOVAL) rv = __builtin_nontemporal_load(P.rec_val + begin + lane); }
    for (int base = begin; base < end; base += 64) {
        const int c_cur = rc; const float v_cur = rv;
        if (base + 64 < end) { rc = __builtin_nontemporal_load(P.rec_col + base + 64 + lane); if (!ARL_SPMM_EXP_NOVAL) rv = __builtin_nontemporal_load(P.rec_val + base + 64 + lane); }       // next batch in flight
#pragma unroll
        for (int j = 0; j < 64; j += UNR) {
            float x[UNR][CPL]; int cs[UNR];
#pragma unroll
            for (int t = 0; t < UNR; ++t) {
                cs[t] = __builtin_amdgcn_readlane(c_cur, j + t);
                const float *src = xl + (size_t)(cs[t] & 0xffffff) * LD;
                if (CPL == 2) { const float2 v2 = *reinterpret_cast<const float2 *>(src); x[t][0] = v2.x; x[t][CPL - 1] = v2.y; }
                else x[t][0] = src[0];
            }
#pragma unroll
            for (int t = 0; t < UNR; ++t) {
                const int slot = ((unsigned)cs[t] >> 24) & 31;
                if (ARL_SPMM_EXP_NOVAL) {                          // timing probe only (profiles/r04_experiments.md): no value stream, plain adds -- results are wrong
#pragma unroll
                    for (int c = 0; c < CPL; ++c) acc[c][slot] += x[t][c];
                    continue;
                }
                const float v = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(v_cur), j + t));
#pragma unroll
                for (int c = 0; c < CPL; ++c) acc[c][slot] = fmaf(v, x[t][c], acc[c][slot]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int row = P.wave_rows[w * RPW + r];
        float a[CPL];
#pragma unroll
        for (int c = 0; c < CPL; ++c) a[c] = acc[c][r];
        if (row >= 0) spmm_epilogue1<MODE, CPL, LD>(ep, row, lane, a);
        else if (row < -1) {                                     // piece of a split (hub) row: raw sum, combined by spmm_long_rows_kernel
            float *dst = P.partial + ((size_t)(-row - 2) * 64 + lane) * CPL;
#pragma unroll
            for (int c = 0; c < CPL; ++c) dst[c] = a[c];
        }
    }
}

#ifndef ARL_D128_HALF_HOPS
#define ARL_D128_HALF_HOPS 1
#endif
template <int MODE>
int launch_spmm_blocked(const arl_blocked *P, const float *X, int64_t d, const Epi &ep, hipStream_t st) {
    if (!P || !X) return ARL_E_NULL;
    if (d != 64 && d != 128) return ARL_E_DIM;
    if (P->n_waves < 0 || P->n_waves > 0x7fffffffll / 64) return ARL_E_RANGE;
    if (P->rows_per_wave != 16 && P->rows_per_wave != 32) return ARL_E_ARG;
    if (P->n_waves == 0) return ARL_OK;
    if (!P->wave_ptr || !P->wave_rows || !P->rec_col || !P->rec_val) return ARL_E_NULL;
    if (P->n_split < 0 || P->n_split > 0x7fffffffll) return ARL_E_RANGE;
    if (P->n_split > 0 && (!P->split_row || !P->split_first || !P->split_count || !P->partial)) return ARL_E_NULL;
    BlockedDev D = {(int)P->n_waves, P->wave_ptr, P->wave_rows, P->rec_col, P->rec_val, P->partial};
    const int64_t wpg = P->waves_per_group ? P->waves_per_group : kWavesPerBlock;
    if (wpg != 1 && wpg != 2 && wpg != 4) return ARL_E_ARG;
    const dim3 grid((unsigned)((P->n_waves + wpg - 1) / wpg)), block((unsigned)(wpg * kWave));
    if (P->loads_in_flight != 16 && P->loads_in_flight != 32) return ARL_E_ARG;
#if ARL_D128_HALF_HOPS
    if (d == 128) {
        // d = 128 as two d = 64 passes over the column halves of the 128-wide tables (row stride 128): the one-column-per-lane kernel moves
        // 12.2 TB/s of gathered rows, the two-column one 10.3 (fewer rows in flight at 118 registers) -- the second pass over the 8-B
        // record stream costs less than that buys.  The split rows' pieces of a half use the front of `partial` and are combined before
        // the next half overwrites them (stream order).
        for (int half = 0; half < 2; ++half) {
            Epi e2 = ep;
            e2.ld = 128;
            const int off = 64 * half;
            if (e2.Z) e2.Z += off;
            if (e2.Y) e2.Y += off;
            if (e2.S_in) e2.S_in += off;
            if (e2.S) e2.S += off;
            if (e2.P) e2.P += off;
            if (e2.M) e2.M += off;
            if (e2.V) e2.V += off;
            if (P->rows_per_wave == 16) hipLaunchKernelGGL((spmm_blocked64_kernel<16, MODE, 16, 1, 128>), grid, block, 0, st, D, X + off, e2);
            else if (P->loads_in_flight == 16) hipLaunchKernelGGL((spmm_blocked64_kernel<32, MODE, 16, 1, 128>), grid, block, 0, st, D, X + off, e2);
            else hipLaunchKernelGGL((spmm_blocked64_kernel<32, MODE, 32, 1, 128>), grid, block, 0, st, D, X + off, e2);
            ARL_LAUNCH_CHECK();
            if (P->n_split > 0) {
                CsrDev C = {};
                C.n_long = (int)P->n_split; C.long_row = P->split_row; C.long_first = P->split_first; C.long_count = P->split_count; C.partial = P->partial;
                const unsigned grid_long = (unsigned)((C.n_long + kWavesPerBlock - 1) / kWavesPerBlock);
                hipLaunchKernelGGL((spmm_long_rows_kernel<16, MODE>), dim3(grid_long), dim3(kBlock), 0, st, C, 64, e2);
                ARL_LAUNCH_CHECK();
            }
        }
        return ARL_OK;
    }
#endif
    if (d == 128) {                                      // two columns per lane: 64 accumulator registers, 16 rows in flight
        if (P->rows_per_wave == 16) hipLaunchKernelGGL((spmm_blocked64_kernel<16, MODE, 16, 2>), grid, block, 0, st, D, X, ep);
        else hipLaunchKernelGGL((spmm_blocked64_kernel<32, MODE, 16, 2>), grid, block, 0, st, D, X, ep);
    } else if (P->rows_per_wave == 16) hipLaunchKernelGGL((spmm_blocked64_kernel<16, MODE, 16, 1>), grid, block, 0, st, D, X, ep);
    else if (P->loads_in_flight == 16) hipLaunchKernelGGL((spmm_blocked64_kernel<32, MODE, 16, 1>), grid, block, 0, st, D, X, ep);
    else hipLaunchKernelGGL((spmm_blocked64_kernel<32, MODE, 32, 1>), grid, block, 0, st, D, X, ep);
    ARL_LAUNCH_CHECK();
    if (P->n_split > 0) {                                // split rows: add the pieces in piece order and run the epilogue
        CsrDev C = {};
        C.n_long = (int)P->n_split; C.long_row = P->split_row; C.long_first = P->split_first; C.long_count = P->split_count; C.partial = P->partial;
        const unsigned grid_long = (unsigned)((C.n_long + kWavesPerBlock - 1) / kWavesPerBlock);
        if (d == 128) hipLaunchKernelGGL((spmm_long_rows_kernel<32, MODE>), dim3(grid_long), dim3(kBlock), 0, st, C, 128, ep);
        else hipLaunchKernelGGL((spmm_long_rows_kernel<16, MODE>), dim3(grid_long), dim3(kBlock), 0, st, C, 64, ep);
        ARL_LAUNCH_CHECK();
    }
    return ARL_OK;
}

// ================================================================================================
// Degree normalisation (util/DataLoader.py:73-87, recommender/LightGCN.py:212-215)
// ================================================================================================
__global__ __launch_bounds__(kBlock) void row_dinv_kernel(int n_rows, const int32_t *__restrict__ rowptr,
                                                           const float *__restrict__ w, float *__restrict__ dinv) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (r >= n_rows) return;
    float s = 0.f;
    for (int e = rowptr[r] + lane; e < rowptr[r + 1]; e += kWave) s += w[e];
    s = wave_sum(s);
    if (lane == 0) dinv[r] = s > 0.f ? 1.0f / sqrtf(s) : 0.f;
}

__global__ __launch_bounds__(kBlock) void norm_vals_kernel(int n_rows, const int32_t *__restrict__ rowptr,
                                                            const int32_t *__restrict__ col, const float *__restrict__ w,
                                                            const float *__restrict__ dinv, float *__restrict__ val) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (r >= n_rows) return;
    const float dr = dinv[r];
    for (int e = rowptr[r] + lane; e < rowptr[r + 1]; e += kWave) val[e] = (dr * w[e]) * dinv[col[e]];
}

// Edge-parallel form of norm_vals_kernel for callers that keep the row id of every edge (one 4-B stream more, no per-row waves:
// 1.1 M rows of ~32 edges cost a wave launch each in the row form).  Same expression, same result.
__global__ __launch_bounds__(kBlock) void norm_vals_coo_kernel(long long nnz, const int32_t *__restrict__ erow, const int32_t *__restrict__ col,
                                                                const float *__restrict__ w, const float *__restrict__ dinv, float *__restrict__ val) {
    const long long e0 = ((long long)blockIdx.x * kBlock + threadIdx.x) * 4;
    if (e0 + 3 < nnz) {
        const int4 r = *reinterpret_cast<const int4 *>(erow + e0), c = *reinterpret_cast<const int4 *>(col + e0);
        const float4 ww = *reinterpret_cast<const float4 *>(w + e0);
        float4 o;
        o.x = (dinv[r.x] * ww.x) * dinv[c.x]; o.y = (dinv[r.y] * ww.y) * dinv[c.y];
        o.z = (dinv[r.z] * ww.z) * dinv[c.z]; o.w = (dinv[r.w] * ww.w) * dinv[c.w];
        *reinterpret_cast<float4 *>(val + e0) = o;
    } else {
        for (long long e = e0; e < nnz; ++e) val[e] = (dinv[erow[e]] * w[e]) * dinv[col[e]];
    }
}

// ================================================================================================
// BPR + L2 (util/loss.py:5-9,25-29; gathers recommender/LightGCN.py:51-52)
// workspace layout (floats): coef[B] | bpr_terms[B] | uu[B] | pp[B]
// ================================================================================================
__global__ __launch_bounds__(kBlock) void bpr_fwd_kernel(const float *__restrict__ emb, int d, long long item_off,
                                                          const int32_t *__restrict__ ui, const int32_t *__restrict__ pi,
                                                          const int32_t *__restrict__ ni, int B, float b_norm, float *__restrict__ ws) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (b >= B) return;
    const float *u = emb + (size_t)ui[b] * d, *p = emb + (size_t)(item_off + pi[b]) * d, *n = emb + (size_t)(item_off + ni[b]) * d;
    float ps = 0.f, ns = 0.f, uu = 0.f, pp = 0.f;
    for (int k = lane; k < d; k += kWave) {
        const float uk = u[k], pk = p[k], nk = n[k];
        ps = fmaf(uk, pk, ps); ns = fmaf(uk, nk, ns); uu = fmaf(uk, uk, uu); pp = fmaf(pk, pk, pp);
    }
    ps = wave_sum(ps); ns = wave_sum(ns); uu = wave_sum(uu); pp = wave_sum(pp);
    if (lane == 0) {
        const float x = ps - ns;
        const float s = 1.0f / (1.0f + expf(-x));
        ws[b] = -(s * (1.0f - s)) / ((1e-7f + s) * b_norm);        // d(mean loss)/dx_b; b_norm = global batch size
        ws[B + b] = -logf(1e-7f + s);
        ws[2 * B + b] = uu;
        ws[3 * B + b] = pp;
    }
}

// single block: fixed-order tree reduction -> bitwise reproducible loss and norms
// raw_sums != 0 (user-sharded batch): out[0..2] = sum of loss terms, sum |u|^2, sum |p|^2 over the LOCAL samples, to be
// all-reduced by the caller before the backward kernel.
__global__ __launch_bounds__(kBlock) void bpr_finalize_kernel(int B, float reg, const float *__restrict__ ws, float *__restrict__ out, int raw_sums) {
    __shared__ float sh[3][kBlock];
    float a = 0.f, b = 0.f, c = 0.f;
    for (int i = threadIdx.x; i < B; i += kBlock) { a += ws[B + i]; b += ws[2 * B + i]; c += ws[3 * B + i]; }
    sh[0][threadIdx.x] = a; sh[1][threadIdx.x] = b; sh[2][threadIdx.x] = c;
    __syncthreads();
    for (int s = kBlock / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            sh[0][threadIdx.x] += sh[0][threadIdx.x + s]; sh[1][threadIdx.x] += sh[1][threadIdx.x + s]; sh[2][threadIdx.x] += sh[2][threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0 && raw_sums) {
        out[0] = sh[0][0]; out[1] = sh[1][0]; out[2] = sh[2][0];
    } else if (threadIdx.x == 0) {
        const float nu = sqrtf(sh[1][0]), np_ = sqrtf(sh[2][0]);
        out[0] = sh[0][0] / (float)B;
        out[1] = reg * (nu + np_);
        out[2] = nu;
        out[3] = np_;
    }
}

// dst[r] += alpha * src[r] ONCE for every distinct row r of the list (duplicates in the list are ignored): adds the listed rows of a table that
// is zero elsewhere (the sparse batch gradient G) into a dense one without a pass over the whole table.
__global__ __launch_bounds__(kBlock) void rows_axpy_unique_kernel(float *__restrict__ dst, const float *__restrict__ src, const int32_t *__restrict__ idx,
                                                                   int c0, int c1, int d, float alpha, const uint32_t *__restrict__ dup, int lds_n) {
    extern __shared__ int32_t srows[];
    const int lane = threadIdx.x & 63;
    const int t = c0 + blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const bool valid = t < c1;
    const long long row = valid ? idx[t] : 0;
    // first occurrence in the WHOLE list (windows are independent launches, so the scan always starts at 0); rows whose bit is clear in
    // `dup` (a bitmap of the rows some list CONTAINING this one names twice, optional) are named once: no scan
    const bool scan = valid && (!dup || ((dup[row >> 5] >> (row & 31)) & 1u));
    if (lds_n > 0) {                                                  // the whole list fits one window: scan a staged copy
        if (__syncthreads_or(scan)) stage_rows(srows, 0, lds_n, [&](int j) { return idx[j]; });
        if (!valid) return;
        if (scan && !ordered_scan_lds(srows, 0, t, t, (int)row, lane, [&](int) {})) return;
    } else {
        if (!valid) return;
        if (scan && !ordered_scan(0, t, t, row, lane, [&](int j) { return idx[j]; }, [&](int) {})) return;
    }
    float *o = dst + (size_t)row * d;
    const float *x = src + (size_t)row * d;
    for (int k = lane; k < d; k += kWave) o[k] = fmaf(alpha, x[k], o[k]);
}

// user-sharded batch: local row ids, ownership weights and the packed row lists of one global batch in ONE launch (replaces ~10 element-wise
// ATen launches per step).  Samples of other ranks' users keep a clamped local row id and weight 0 (static shapes, no host sync).
__global__ __launch_bounds__(kBlock) void shard_batch_prep_kernel(const int32_t *__restrict__ u, const int32_t *__restrict__ p, const int32_t *__restrict__ n,
                                                                   int B, int u0, int u1, int32_t *__restrict__ lu, float *__restrict__ own,
                                                                   int32_t *__restrict__ item_rows, int32_t *__restrict__ rows_l) {
    const int b = blockIdx.x * kBlock + threadIdx.x;
    if (b >= B) return;
    const int Ul = u1 - u0;
    const int uu = u[b];
    const int l = min(max(uu - u0, 0), max(Ul - 1, 0));
    lu[b] = l; own[b] = (uu >= u0 && uu < u1) ? 1.f : 0.f; own[B + b] = 1.f; own[2 * B + b] = 1.f;      // [3B]: per-contribution factors of [u | p | n]
    const int pp = p[b], nn = n[b];
    item_rows[b] = pp; item_rows[B + b] = nn;
    rows_l[b] = l; rows_l[B + b] = Ul + pp; rows_l[2 * B + b] = Ul + nn;
}

// Backward of BPR + L2 into the rows of G, ordered (no float atomics; see rows_add_ordered_kernel).  The 3B contributions are numbered
// t = b (user row of sample b), B + b (its positive item), 2B + b (its negative item): a user row receives its samples' terms in sample
// order, an item row its positive-role terms in sample order and then its negative-role ones -- the order in which autograd's three
// index_put_(accumulate) calls of the reference add them (recommender/LightGCN.py:51-52).
__global__ __launch_bounds__(kBlock) void bpr_bwd_kernel(const float *__restrict__ emb, int d, long long item_off,
                                                          const int32_t *__restrict__ ui, const int32_t *__restrict__ pi,
                                                          const int32_t *__restrict__ ni, int B, float reg, float upstream,
                                                          const float *__restrict__ ws, const float *__restrict__ out,
                                                          float *__restrict__ G, int c0, int c1, int distinct) {
    extern __shared__ int32_t srows[];
    const int lane = threadIdx.x & 63;
    const int t = c0 + blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
#define ARL_ROWOF_BPR(j) ((j) < B ? (long long)ui[j] : ((j) < 2 * B ? item_off + pi[(j) - B] : item_off + ni[(j) - 2 * B]))
    if (!distinct) stage_rows(srows, c0, c1, [&](int j) { return ARL_ROWOF_BPR(j); });      // the window's rows, once per workgroup
    if (t >= c1) return;
    const int row = (int)ARL_ROWOF_BPR(t);
    const float cu = out[2] > 0.f ? upstream * reg / out[2] : 0.f, cp = out[3] > 0.f ? upstream * reg / out[3] : 0.f;
    float *o = G + (size_t)row * d;
    const float *own = emb + (size_t)row * d;
    for (int k0 = 0; k0 < d; k0 += 256) {
        float acc[4], mine[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { const int k = k0 + 64 * i + lane; acc[i] = k < d ? o[k] : 0.f; mine[i] = k < d ? own[k] : 0.f; }
        auto term = [&](int tt) {
            const int kind = tt / B, b = tt - kind * B;
            const float g = ws[b] * upstream;
            const float *x = emb + (size_t)(kind == 0 ? item_off + pi[b] : (long long)ui[b]) * d;   // the other operand: positive row (user term) / user row (item terms)
            const float *y = emb + (size_t)(item_off + ni[b]) * d;                                // negative row (user term only)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = k0 + 64 * i + lane;
                if (k < d) {
                    float c;
                    if (kind == 0) c = fmaf(g, x[k] - y[k], cu * mine[i]);
                    else if (kind == 1) c = fmaf(g, x[k], cp * mine[i]);
                    else c = -g * x[k];
                    acc[i] = __fadd_rn(acc[i], c);
                }
            }
        };
        // distinct: the caller guarantees 3B distinct rows (compact batch form) -- every entry owns its row, nothing to scan
        if (distinct) term(t);
        else if (!ordered_scan_lds(srows, c0, c1, t, row, lane, term)) return;
#pragma unroll
        for (int i = 0; i < 4; ++i) { const int k = k0 + 64 * i + lane; if (k < d) o[k] = acc[i]; }
    }
#undef ARL_ROWOF_BPR
}

// ================================================================================================
// Dense optimisers (torch.optim.Adam / SGD)
// ================================================================================================
__global__ __launch_bounds__(kBlock) void adam_kernel(float4 *__restrict__ p, const float4 *__restrict__ g, float4 *__restrict__ m,
                                                       float4 *__restrict__ v, long long n4, float step_size, float bc2_sqrt,
                                                       float w1, float b2, float w2, float eps) {
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (long long)gridDim.x * kBlock) {
        const float4 g4 = g[i];
        float4 p4 = p[i], m4 = m[i], v4 = v[i];
        float gg[4] = {g4.x, g4.y, g4.z, g4.w}, pp[4] = {p4.x, p4.y, p4.z, p4.w}, mm[4] = {m4.x, m4.y, m4.z, m4.w}, vv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            mm[k] = mm[k] + (gg[k] - mm[k]) * w1;
            vv[k] = vv[k] * b2 + w2 * gg[k] * gg[k];
            pp[k] = pp[k] - step_size * (mm[k] / (sqrtf(vv[k]) / bc2_sqrt + eps));
        }
        p[i] = make_float4(pp[0], pp[1], pp[2], pp[3]); m[i] = make_float4(mm[0], mm[1], mm[2], mm[3]); v[i] = make_float4(vv[0], vv[1], vv[2], vv[3]);
    }
}
// scalar form for the <= 3 trailing elements (or a whole table whose base is not 16-byte aligned)
__global__ __launch_bounds__(kBlock) void adam_scalar_kernel(float *p, const float *g, float *m, float *v, long long from, long long n,
                                                              float step_size, float bc2_sqrt, float w1, float b2, float w2, float eps) {
    for (long long i = from + (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) {
        const float gi = g[i];
        const float mi = m[i] + (gi - m[i]) * w1;
        const float vi = v[i] * b2 + w2 * gi * gi;
        m[i] = mi; v[i] = vi;
        p[i] = p[i] - step_size * (mi / (sqrtf(vi) / bc2_sqrt + eps));
    }
}
__global__ __launch_bounds__(kBlock) void sgd_kernel(float *__restrict__ p, const float *__restrict__ g, long long n, float lr) {
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) p[i] = p[i] - lr * g[i];
}

// torch.optim.Adam holds lr, betas and eps as Python doubles and rounds DERIVED quantities to fp32 (torch/optim/adam.py _single_tensor_adam:
// lerp weight 1 - beta1, addcmul value 1 - beta2, step_size = lr / (1 - beta1^t), sqrt(1 - beta2^t)): (float)(1 - 0.999) = 0.001f, whereas
// 1.0f - 0.999f = 0.00100005 (4.7e-5 off in every second-moment update).  The C ABI carries floats, so the double the caller typed is recovered
// as the shortest decimal that rounds to the float (0.999f -> 0.999); a float that is no short decimal is taken as it is.
inline double typed_double(float x) {
    char buf[32];
    snprintf(buf, sizeof buf, "%.7g", (double)x);
    const double d = strtod(buf, nullptr);
    return (float)d == x ? d : (double)x;
}
struct AdamScalars { float step_size, bc2_sqrt, w1, b2, w2; };
inline AdamScalars adam_scalars(float lr, float b1, float b2, int64_t step) {
    const double B1 = typed_double(b1), B2 = typed_double(b2), LR = typed_double(lr);
    const double bc1 = 1.0 - pow(B1, (double)step), bc2 = 1.0 - pow(B2, (double)step);
    return {(float)(LR / bc1), (float)sqrt(bc2), (float)(1.0 - B1), (float)B2, (float)(1.0 - B2)};
}

// ================================================================================================
// gather / scatter-add rows
// ================================================================================================
__global__ __launch_bounds__(kBlock) void gather_rows_kernel(const float *__restrict__ src, const int32_t *__restrict__ idx, int n, int d,
                                                              float *__restrict__ dst) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (t >= n) return;
    const float *s = src + (size_t)idx[t] * d;
    for (int k = lane; k < d; k += kWave) dst[(size_t)t * d + k] = s[k];
}
// ================================================================================================
// InfoNCE (util/loss.py:42-49).  workspace (floats): a[n*d] | b[n*d] | n1[n] | n2[n] | ttl[n] | rowloss[n] | da[n*d] | db[n*d]
// ================================================================================================
__global__ __launch_bounds__(kBlock) void nce_normalize_kernel(const float *__restrict__ v, int n, int d, float *__restrict__ y, float *__restrict__ nrm) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (i >= n) return;
    float s = 0.f;
    for (int k = lane; k < d; k += kWave) { const float x = v[(size_t)i * d + k]; s = fmaf(x, x, s); }
    s = wave_sum(s);
    const float r = fmaxf(sqrtf(s), 1e-12f);          // F.normalize eps
    for (int k = lane; k < d; k += kWave) y[(size_t)i * d + k] = v[(size_t)i * d + k] / r;
    if (lane == 0) nrm[i] = r;
}

// One block per row i of `a`: s_ij = <a_i, b_j> for all j; ttl_i = sum_j exp(s_ij/tau); rowloss_i = -(s_ii/tau - log ttl_i)
__global__ __launch_bounds__(kBlock) void nce_rowsum_kernel(const float *__restrict__ a, const float *__restrict__ b, int n, int d, float inv_tau,
                                                             float *__restrict__ ttl, float *__restrict__ rowloss) {
    __shared__ float ai[256];
    __shared__ float red[kBlock];
    __shared__ float sii_sh;
    const int i = blockIdx.x;
    for (int k = threadIdx.x; k < d; k += kBlock) ai[k] = a[(size_t)i * d + k];
    __syncthreads();
    float acc = 0.f;
    for (int j = threadIdx.x; j < n; j += kBlock) {
        const float *bj = b + (size_t)j * d;
        float s = 0.f;
        for (int k = 0; k < d; k += 4) {
            const float4 x = *reinterpret_cast<const float4 *>(bj + k);
            s = fmaf(ai[k], x.x, s); s = fmaf(ai[k + 1], x.y, s); s = fmaf(ai[k + 2], x.z, s); s = fmaf(ai[k + 3], x.w, s);
        }
        if (j == i) sii_sh = s;
        acc += expf(s * inv_tau);
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = kBlock / 2; s > 0; s >>= 1) { if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
    if (threadIdx.x == 0) { ttl[i] = red[0]; rowloss[i] = -(sii_sh * inv_tau - logf(red[0])); }
}

__global__ __launch_bounds__(kBlock) void nce_loss_kernel(const float *__restrict__ rowloss, int n, float *__restrict__ out) {
    __shared__ float red[kBlock];
    float a = 0.f;
    for (int i = threadIdx.x; i < n; i += kBlock) a += rowloss[i];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int s = kBlock / 2; s > 0; s >>= 1) { if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
    if (threadIdx.x == 0) out[0] = red[0] / (float)n;
}

// One block per row r of the side being differentiated.
//   ROWSIDE = true : r = i, w_j = (exp(s_ij/tau)/ttl_i - [i==j]) * scale ; dy = sum_j w_j other_j   (other = b, self = a)
//   ROWSIDE = false: r = j, w_i = (exp(s_ij/tau)/ttl_i - [i==j]) * scale ; dy = sum_i w_i other_i   (other = a, self = b)
// then back through the row normalisation: dx = (dy - y <y,dy>) / nrm.
template <bool ROWSIDE>
__global__ __launch_bounds__(kBlock) void nce_grad_kernel(const float *__restrict__ self, const float *__restrict__ other, const float *__restrict__ ttl,
                                                           const float *__restrict__ nrm, int n, int d, float inv_tau, float scale,
                                                           float *__restrict__ dx) {
    extern __shared__ float shm[];           // y[d] | w[n] | part[kBlock]
    float *y = shm, *w = shm + 256, *part = w + n;
    const int r = blockIdx.x;
    for (int k = threadIdx.x; k < d; k += kBlock) y[k] = self[(size_t)r * d + k];
    __syncthreads();
    const float ttl_r = ROWSIDE ? ttl[r] : 0.f;
    for (int j = threadIdx.x; j < n; j += kBlock) {
        const float *oj = other + (size_t)j * d;
        float s = 0.f;
        for (int k = 0; k < d; k += 4) {
            const float4 x = *reinterpret_cast<const float4 *>(oj + k);
            s = fmaf(y[k], x.x, s); s = fmaf(y[k + 1], x.y, s); s = fmaf(y[k + 2], x.z, s); s = fmaf(y[k + 3], x.w, s);
        }
        const float t = ROWSIDE ? ttl_r : ttl[j];
        w[j] = (expf(s * inv_tau) / t - (j == r ? 1.f : 0.f)) * scale;
    }
    __syncthreads();
    // dy[k] = sum_j w[j] * other[j][k]; threads split j into kBlock/dpad slices per column k
    const int dpad = d <= 64 ? 64 : (d <= 128 ? 128 : 256);
    const int slices = kBlock / dpad;
    const int k = threadIdx.x % dpad, sl = threadIdx.x / dpad;
    float acc = 0.f;
    if (k < d)
        for (int j = sl; j < n; j += slices) acc = fmaf(w[j], other[(size_t)j * d + k], acc);
    part[threadIdx.x] = acc;
    __syncthreads();
    float dy = 0.f;
    if ((int)threadIdx.x < dpad) for (int s = 0; s < slices; ++s) dy += part[s * dpad + threadIdx.x];
    __syncthreads();
    // <y, dy>
    part[threadIdx.x] = ((int)threadIdx.x < d) ? y[threadIdx.x] * dy : 0.f;
    __syncthreads();
    for (int s = kBlock / 2; s > 0; s >>= 1) { if ((int)threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s]; __syncthreads(); }
    const float dot = part[0];
    if ((int)threadIdx.x < d) dx[(size_t)r * d + threadIdx.x] = (dy - y[threadIdx.x] * dot) / nrm[r];
}

// ================================================================================================
// All-rows InfoNCE (recommender/NCL.py:96-115 ssl_layer_loss; attack/White/InfoAttack.py:214-230): for normalised rows A [nA, d] (the batch)
// and a normalised table V [nV, d] (ALL users or items),
//     lse_b = log sum_j exp(<a_b, v_j> / T),   dA_b = sum_j P_bj v_j,   dV_j = sum_b P_bj a_b,   P_bj = exp(<a_b, v_j>/T - lse_b)
// with nA x nV logits (2 048 x 10^6 at cfg2) that are never stored.  The reference materialises them; the round-2 form walked the table in
// panels with two library GEMMs per panel and pass.  Here one kernel, three uses, exact fp32 on v_mfma_f32_16x16x4_f32:
//   * a wave keeps 16 rows of the RESIDENT side in registers (B operand; lane (c, g) holds row c, columns [16g, 16g + 16) -- a permutation
//     of the contraction index both operands share) and its output for them (a sum of exponentials, or a 16 x d gradient tile);
//   * the STREAMED side goes through LDS 64 rows at a time (double-buffered, one block barrier per stage);
//   * scores come out as C[t = 4g + reg][r = c]; exp() of them is already in A-operand layout for the second product
//     out[r][:] += sum_t P[t][r] X_t[t][:]  (MFMA j contracts t = 4g + j), so no transposition through LDS;
//   * |<a, v>| <= 1 for normalised rows: exp((s - 1)/T) cannot overflow, so the log-sum-exp needs no running maximum.
// Uses: lse and dA with A resident and V streamed in `gridDim.y` splits (partials summed in split order: deterministic), dV with V resident and
// A streamed whole.  LSE_ON_R: the log-sum-exp subtracted from a score belongs to the resident row (A resident) or to the streamed row.
// FUSED (with GRAD, LSE_ON_R): no log-sum-exp comes in; the tile accumulates the UNNORMALISED sum_t exp((s - 1)/T) x_t and the row sums go to
// `sums` as in the GRAD = false form -- the fold divides (one pass over the table less than log-sum-exp first, gradient second).
template <int D, bool GRAD, bool LSE_ON_R, bool FUSED = false>
__global__ __launch_bounds__(kBlock) void nce_allrows_kernel(const float *__restrict__ Xr, int nR, const float *__restrict__ Xt, int nT, int split_len,
                                                              float inv_tau, const float *__restrict__ lse, float *__restrict__ out, float *__restrict__ sums) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    constexpr int Q = D / 4, LD = D + 4, NT = D / 16;
    __shared__ float tile[2][64 * LD];
    __shared__ float tlse[2][64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, c = lane & 15, g = lane >> 4;
    const int r0 = (blockIdx.x * kWavesPerBlock + wv) * 16;
    const int t_begin = blockIdx.y * split_len, t_end = min(nT, t_begin + split_len);
    float br[Q];
    {
        const int r = r0 + c;
#pragma unroll
        for (int i = 0; i < Q; i += 4) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < nR) v = *reinterpret_cast<const float4 *>(Xr + (size_t)r * D + Q * g + i);
            br[i] = v.x; br[i + 1] = v.y; br[i + 2] = v.z; br[i + 3] = v.w;
        }
    }
    const float lse_r = (GRAD && LSE_ON_R && !FUSED && r0 + c < nR) ? lse[r0 + c] : 0.f;
    v4f o[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) o[n] = v4f{0.f, 0.f, 0.f, 0.f};
    float sum = 0.f;
    const int nst = (t_end - t_begin + 63) / 64;
    // staging: a stage is 64 x D floats = 16 D float4s over 256 threads; the NEXT stage is fetched into registers before the current one
    // is consumed and goes to the other LDS buffer afterwards (its last readers passed the barrier of the stage before)
    // (macros, not lambdas: an array captured by reference stays in scratch memory and every load is waited for at once)
    constexpr int PF = 64 * (D / 4) / kBlock;
    v4f pre[PF];
    float pre_lse = 0.f;
    const int frow = tid / (D / 4), fq = (tid % (D / 4)) * 4;               // this thread's row (+ i * kBlock / (D/4)) and column of a stage
    constexpr int FSTEP = kBlock / (D / 4);
#define ARL_NCE_FETCH(ST)                                                                                                          \
    do {                                                                                                                           \
        _Pragma("unroll") for (int i = 0; i < PF; ++i) {                                                                           \
            const int t = min(t_begin + (ST) * 64 + frow + i * FSTEP, nT - 1);     /* clamped; rows past t_end are masked below */ \
            pre[i] = *reinterpret_cast<const v4f *>(Xt + (size_t)t * D + fq);                                                      \
        }                                                                                                                          \
        if (GRAD && !LSE_ON_R && tid < 64) pre_lse = lse[min(t_begin + (ST) * 64 + tid, nT - 1)];                                  \
    } while (0)
#define ARL_NCE_STASH(BUF)                                                                                                         \
    do {                                                                                                                           \
        _Pragma("unroll") for (int i = 0; i < PF; ++i) *reinterpret_cast<v4f *>(&tile[BUF][(frow + i * FSTEP) * LD + fq]) = pre[i]; \
        if (GRAD && !LSE_ON_R && tid < 64) tlse[BUF][tid] = pre_lse;                                                               \
    } while (0)
    if (nst > 0) { ARL_NCE_FETCH(0); ARL_NCE_STASH(0); }
    __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0): the resident rows have landed on every path into the loop (else the in-loop wait is vmcnt(0) too)
    __syncthreads();
    for (int st = 0; st < nst; ++st) {
        const int buf = st & 1;
        if (st + 1 < nst) ARL_NCE_FETCH(st + 1);
        const float *T = tile[buf];
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
            v4f sc = v4f{0.f, 0.f, 0.f, 0.f};
            const float *arow = T + (sub * 16 + c) * LD + Q * g;
#pragma unroll
            for (int i = 0; i < Q; i += 4) {
                const float4 a = *reinterpret_cast<const float4 *>(arow + i);
                sc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, br[i], sc, 0, 0, 0);
                sc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, br[i + 1], sc, 0, 0, 0);
                sc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, br[i + 2], sc, 0, 0, 0);
                sc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, br[i + 3], sc, 0, 0, 0);
            }
            const int tb = t_begin + st * 64 + sub * 16 + 4 * g;        // streamed row of register 0
            float pj[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool ok = tb + j < t_end;
                if (!GRAD || FUSED) pj[j] = ok ? __expf((sc[j] - 1.f) * inv_tau) : 0.f;
                else pj[j] = ok ? __expf(sc[j] * inv_tau - (LSE_ON_R ? lse_r : tlse[buf][sub * 16 + 4 * g + j])) : 0.f;
            }
            if (!GRAD || FUSED) sum += (pj[0] + pj[1]) + (pj[2] + pj[3]);
            if (GRAD) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float *brow = T + (sub * 16 + 4 * g + j) * LD + c;
#pragma unroll
                    for (int n = 0; n < NT; ++n) o[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(pj[j], brow[16 * n], o[n], 0, 0, 0);
                }
            }
        }
        if (st + 1 < nst) ARL_NCE_STASH(buf ^ 1);
        __syncthreads();
    }
#undef ARL_NCE_FETCH
#undef ARL_NCE_STASH
    if (!GRAD || FUSED) {
        sum += __shfl_xor(sum, 16); sum += __shfl_xor(sum, 32);         // over the four lane groups (fixed order)
        if (g == 0 && r0 + c < nR) (GRAD ? sums : out)[(size_t)blockIdx.y * nR + r0 + c] = sum;
    }
    if (GRAD) {
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int r = r0 + 4 * g + reg;
                if (r < nR) out[((size_t)blockIdx.y * nR + r) * D + 16 * n + c] = o[n][reg];
            }
    }
}

// lse[b] = 1/T + log(sum over splits of the partial sums), splits added in order
__global__ __launch_bounds__(kBlock) void nce_allrows_lse_finish_kernel(const float *__restrict__ part, int n_splits, int nA, float inv_tau, float *__restrict__ lse) {
    const int b = blockIdx.x * kBlock + threadIdx.x;
    if (b >= nA) return;
    float s = 0.f;
    for (int k = 0; k < n_splits; ++k) s += part[(size_t)k * nA + b];
    lse[b] = inv_tau + logf(s);
}
// fused form: lse[b] = 1/T + log Z_b, dA[b][:] = (sum over splits of the unnormalised tiles) / Z_b, Z_b = sum over splits of the row sums
__global__ __launch_bounds__(kBlock) void nce_allrows_fold_norm_kernel(const float4 *__restrict__ part, const float *__restrict__ sums, int n_splits, int nA, int d4,
                                                                        float inv_tau, float *__restrict__ lse, float4 *__restrict__ out) {
    const long long n4 = (long long)nA * d4;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (long long)gridDim.x * kBlock) {
        const int b = (int)(i / d4);
        float z = 0.f;
        for (int k = 0; k < n_splits; ++k) z += sums[(size_t)k * nA + b];
        float4 a = part[i];
        for (int k = 1; k < n_splits; ++k) a = add4(a, part[(size_t)k * n4 + i]);
        out[i] = make_float4(a.x / z, a.y / z, a.z / z, a.w / z);
        if (i - (long long)b * d4 == 0) lse[b] = inv_tau + logf(z);
    }
}
// out[i] = sum over splits of part[k][i], in split order
__global__ __launch_bounds__(kBlock) void nce_allrows_fold_kernel(const float4 *__restrict__ part, int n_splits, long long n4, float4 *__restrict__ out) {
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (long long)gridDim.x * kBlock) {
        float4 a = part[i];
        for (int k = 1; k < n_splits; ++k) a = add4(a, part[(size_t)k * n4 + i]);
        out[i] = a;
    }
}

// F.normalize(x, dim=1) and its autograd on whole tables (recommender/NCL.py:98-99, 110-111: the four normalisations around each all-rows
// InfoNCE), one pass each: LPR = d/4 rounded up to a power of two lanes hold a row as float4s, 64 / LPR rows per wave.
//   forward : y = x / max(||x||, 1e-12), nrm = max(||x||, 1e-12)
//   backward: dx = scale * (dy - y <y, dy>) / nrm      (dx = scale * dy / nrm for a row at the eps clamp: torch's clamp_min passes no gradient there)
template <bool BWD>
__global__ __launch_bounds__(kBlock) void rows_normalize_kernel(const float *__restrict__ X, const float *__restrict__ dY, float *__restrict__ nrm, long long n, int d,
                                                                int lpr, float scale, const float *__restrict__ scale_dev, float *__restrict__ out) {
    if (BWD && scale_dev) scale *= *scale_dev;
    const int lane = threadIdx.x & 63, sub = lane % lpr, rows_per_wave = 64 / lpr;
    const long long wave0 = (long long)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    for (long long w = wave0; w * rows_per_wave < n; w += (long long)gridDim.x * kWavesPerBlock) {
        const long long r = w * rows_per_wave + lane / lpr;
        const bool live = r < n && sub * 4 < d;
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f), g = x;
        if (live) x = *reinterpret_cast<const float4 *>(X + r * d + sub * 4);
        if (BWD && live) g = *reinterpret_cast<const float4 *>(dY + r * d + sub * 4);
        float s = BWD ? (x.x * g.x + x.y * g.y) + (x.z * g.z + x.w * g.w) : (x.x * x.x + x.y * x.y) + (x.z * x.z + x.w * x.w);
        for (int o = 1; o < lpr; o <<= 1) s += __shfl_xor(s, o);
        if (!BWD) {
            const float m = fmaxf(sqrtf(s), 1e-12f);
            if (live) *reinterpret_cast<float4 *>(out + r * d + sub * 4) = make_float4(x.x / m, x.y / m, x.z / m, x.w / m);
            if (r < n && sub == 0) nrm[r] = m;
        } else if (live) {
            const float m = nrm[r];
            const float dot = m > 1e-12f ? s : 0.f;
            *reinterpret_cast<float4 *>(out + r * d + sub * 4) = make_float4(scale * (g.x - x.x * dot) / m, scale * (g.y - x.y * dot) / m,
                                                                              scale * (g.z - x.z * dot) / m, scale * (g.w - x.w * dot) / m);
        }
    }
}

// Gradient of a scalar loss with respect to the STORED entries of the adjacency, given dL/d(A X) = dY (the autograd of torch.sparse.mm with
// respect to its sparse argument, which recommender/LightGCN.py:41-43,58-59 accumulates when requires_adjgrad is set):
//     gval[e] += alpha * <dY[row(e)], X[col[e]]>      for every stored entry e, CSR order.
// One wave per row; LPR = d/4 (rounded up to a power of two) lanes hold an edge's operand row as float4s, 64 / LPR edges per round.
// Each entry has one writer: no atomics.
__global__ __launch_bounds__(kBlock) void sddmm_csr_kernel(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col, long long n_rows, int d, int lpr,
                                                           const float *__restrict__ dY, const float *__restrict__ X, float alpha, float *__restrict__ gval) {
    const int lane = threadIdx.x & 63, sub = lane % lpr, slot = lane / lpr, per_round = 64 / lpr;
    const bool qact = sub * 4 < d;
    for (long long r = (long long)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6); r < n_rows; r += (long long)gridDim.x * kWavesPerBlock) {
        const int e0 = rowptr[r], e1 = rowptr[r + 1];
        if (e0 == e1) continue;
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        if (qact) g = *reinterpret_cast<const float4 *>(dY + r * d + sub * 4);
        for (int e = e0 + slot; __any(e < e1); e += per_round) {
            float s = 0.f;
            if (e < e1 && qact) {
                const float4 x = *reinterpret_cast<const float4 *>(X + (long long)col[e] * d + sub * 4);
                s = (g.x * x.x + g.y * x.y) + (g.z * x.z + g.w * x.w);
            }
            for (int o = 1; o < lpr; o <<= 1) s += __shfl_xor(s, o);
            if (e < e1 && sub == 0) gval[e] = fmaf(alpha, s, gval[e]);
        }
    }
}

// ================================================================================================
// NGCF layer glue (recommender/NGCF.py:200-208): E' = leaky_relu((P + E) W1 + (P * E) W2), P = A E.
// The two d x d products run as ONE rocBLAS GEMM on [S | T] (N x 2d) by [W1; W2]; these kernels are the element-wise passes
// around it, each touching every operand once (the ATen expression graph makes ~18 passes forward, ~30 backward).
// ================================================================================================
__global__ __launch_bounds__(kBlock) void ngcf_combine_kernel(const float4 *__restrict__ P, const float4 *__restrict__ E, float4 *__restrict__ ST,
                                                              long long n4, int d4) {
    for (long long t = (long long)blockIdx.x * kBlock + threadIdx.x; t < n4; t += (long long)gridDim.x * kBlock) {
        const long long row = t / d4;
        const int c = (int)(t - row * d4);
        const float4 p = P[t], e = E[t];
        ST[row * 2 * d4 + c] = make_float4(p.x + e.x, p.y + e.y, p.z + e.z, p.w + e.w);
        ST[row * 2 * d4 + d4 + c] = make_float4(p.x * e.x, p.y * e.y, p.z * e.z, p.w * e.w);
    }
}

// Z <- leaky_relu(Z) in place; acc += Z (the running layer sum), when acc != NULL
__global__ __launch_bounds__(kBlock) void ngcf_act_kernel(float4 *__restrict__ Z, float4 *__restrict__ acc, long long n4, float slope) {
    for (long long t = (long long)blockIdx.x * kBlock + threadIdx.x; t < n4; t += (long long)gridDim.x * kBlock) {
        float4 z = Z[t];
        z.x = z.x > 0.f ? z.x : z.x * slope; z.y = z.y > 0.f ? z.y : z.y * slope;
        z.z = z.z > 0.f ? z.z : z.z * slope; z.w = z.w > 0.f ? z.w : z.w * slope;
        Z[t] = z;
        if (acc) { float4 a = acc[t]; a.x += z.x; a.y += z.y; a.z += z.z; a.w += z.w; acc[t] = a; }
    }
}

// gZ = gOut * (Out > 0 ? 1 : slope)   (sign(Out) == sign(Z) for slope > 0; at 0 torch uses the negative-side slope)
__global__ __launch_bounds__(kBlock) void ngcf_act_bwd_kernel(const float4 *__restrict__ gOut, const float4 *__restrict__ Out, float4 *__restrict__ gZ,
                                                              long long n4, float slope) {
    for (long long t = (long long)blockIdx.x * kBlock + threadIdx.x; t < n4; t += (long long)gridDim.x * kBlock) {
        const float4 g = gOut[t], o = Out[t];
        gZ[t] = make_float4(g.x * (o.x > 0.f ? 1.f : slope), g.y * (o.y > 0.f ? 1.f : slope), g.z * (o.z > 0.f ? 1.f : slope), g.w * (o.w > 0.f ? 1.f : slope));
    }
}

// gP = gS + gT * E;  gE = gS + gT * P   from gST = [gS | gT]
__global__ __launch_bounds__(kBlock) void ngcf_combine_bwd_kernel(const float4 *__restrict__ gST, const float4 *__restrict__ P, const float4 *__restrict__ E,
                                                                  float4 *__restrict__ gP, float4 *__restrict__ gE, long long n4, int d4) {
    for (long long t = (long long)blockIdx.x * kBlock + threadIdx.x; t < n4; t += (long long)gridDim.x * kBlock) {
        const long long row = t / d4;
        const int c = (int)(t - row * d4);
        const float4 gs = gST[row * 2 * d4 + c], gt = gST[row * 2 * d4 + d4 + c], p = P[t], e = E[t];
        gP[t] = make_float4(fmaf(gt.x, e.x, gs.x), fmaf(gt.y, e.y, gs.y), fmaf(gt.z, e.z, gs.z), fmaf(gt.w, e.w, gs.w));
        gE[t] = make_float4(fmaf(gt.x, p.x, gs.x), fmaf(gt.y, p.y, gs.y), fmaf(gt.z, p.z, gs.z), fmaf(gt.w, p.w, gs.w));
    }
}


// ================================================================================================
// NGCF dense layer on the matrix cores (recommender/NGCF.py:200-208), fp32 in / fp32 out with v_mfma_f32_16x16x4_f32 (exact fp32
// products, fp32 accumulation): the layer's  Z = (P + E) W1 + (P * E) W2  is a [N, 2d] x [2d, d] product whose left operand is never
// materialised -- a wave owns 16 rows, forms S = P + E and T = P * E in registers from one float4 per lane and operand, and streams the
// d x d weights from LDS (row stride d + 4 floats: the four k-groups of a wave read four disjoint bank quarters).  The K order inside a
// group of 16 is permuted so that lane (row, q) feeds k = 16 g + 4 q + j in step j: its float4 needs no shuffle; B is read to match.
//   forward  : out = leaky_relu(Z)                                   reads P, E once, writes out once (was: 3 passes + a library GEMM)
//   dgrad    : gZ = gOut * act'(out);  [gS | gT] = gZ [W1; W2]^T;  gP = gS + gT * E;  gE = gS + gT * P     (gZ is kept for wgrad)
//   wgrad    : [gW1; gW2] = [S | T]^T gZ   per workgroup over its row tiles (4 waves split the 2d/16 row tiles of gW), partials summed
//              in a fixed order by a second kernel (deterministic)
// ================================================================================================
typedef float f32x4v __attribute__((ext_vector_type(4)));
constexpr int kNgcfBlock = 256;

// workgroup size of the fwd / dgrad kernels: the LDS image of the weights (135 KB at d = 128) allows one workgroup per CU there, so that one is
// 16 waves wide (4 per SIMD); smaller widths fit several 4-wave workgroups per CU
template <int D> struct NgcfBlk { static constexpr int value = D >= 128 ? 1024 : 256; };

template <int D>
__device__ __forceinline__ void ngcf_stage_weights(float *Wl, const float *__restrict__ W, int rows) {
    constexpr int S = D + 4;
    for (int t = threadIdx.x; t < rows * D; t += NgcfBlk<D>::value) Wl[(t / D) * S + (t % D)] = W[t];
    __syncthreads();
}

template <int D>
__global__ __launch_bounds__(NgcfBlk<D>::value) void ngcf_dense_fwd_kernel(const float *__restrict__ P, const float *__restrict__ E, const float *__restrict__ W,
                                                                     long long n_rows, float slope, float *__restrict__ out) {
    constexpr int S = D + 4, NT = D / 16, KG = D / 16;
    extern __shared__ float Wl[];                           // [2 D][S]: W1 rows then W2 rows
    ngcf_stage_weights<D>(Wl, W, 2 * D);
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    const long long n_tiles = (n_rows + 15) / 16;
    for (long long tile = (long long)blockIdx.x * (NgcfBlk<D>::value / 64) + (threadIdx.x >> 6); tile < n_tiles; tile += (long long)gridDim.x * (NgcfBlk<D>::value / 64)) {
        const long long r = tile * 16 + c;
        const bool rv = r < n_rows;
        f32x4v acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int g = 0; g < KG; ++g) {
            float4 p = make_float4(0.f, 0.f, 0.f, 0.f), e = p;
            if (rv) { p = *reinterpret_cast<const float4 *>(P + r * D + 16 * g + 4 * q); e = *reinterpret_cast<const float4 *>(E + r * D + 16 * g + 4 * q); }
            const float sv[4] = {p.x + e.x, p.y + e.y, p.z + e.z, p.w + e.w}, tv[4] = {p.x * e.x, p.y * e.y, p.z * e.z, p.w * e.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float *w1 = Wl + (16 * g + 4 * q + j) * S + c, *w2 = w1 + D * S;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(sv[j], w1[16 * nt], acc[nt], 0, 0, 0);
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(tv[j], w2[16 * nt], acc[nt], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const long long row = tile * 16 + 4 * q + i;
                const float z = acc[nt][i];
                if (row < n_rows) out[row * D + 16 * nt + c] = z > 0.f ? z : z * slope;
            }
    }
}

// Wt = [W1; W2]^T, i.e. [D][2 D] row-major: the B operand of gZ [W1; W2]^T read exactly like the forward's
template <int D>
__global__ __launch_bounds__(NgcfBlk<D>::value) void ngcf_dense_dgrad_kernel(const float *__restrict__ gOut, const float *__restrict__ Out, const float *__restrict__ P,
                                                                       const float *__restrict__ E, const float *__restrict__ Wt, long long n_rows, float slope,
                                                                       float *__restrict__ gZ, float *__restrict__ gP, float *__restrict__ gE) {
    constexpr int D2 = 2 * D, S = D2 + 4, NT = D2 / 16, KG = D / 16;
    extern __shared__ float Wl[];                           // [D][S]
    for (int t = threadIdx.x; t < D * D2; t += NgcfBlk<D>::value) Wl[(t / D2) * S + (t % D2)] = Wt[t];
    __syncthreads();
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    const long long n_tiles = (n_rows + 15) / 16;
    for (long long tile = (long long)blockIdx.x * (NgcfBlk<D>::value / 64) + (threadIdx.x >> 6); tile < n_tiles; tile += (long long)gridDim.x * (NgcfBlk<D>::value / 64)) {
        const long long r = tile * 16 + c;
        const bool rv = r < n_rows;
        f32x4v acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int g = 0; g < KG; ++g) {
            float4 go = make_float4(0.f, 0.f, 0.f, 0.f), o = go;
            if (rv) { go = *reinterpret_cast<const float4 *>(gOut + r * D + 16 * g + 4 * q); o = *reinterpret_cast<const float4 *>(Out + r * D + 16 * g + 4 * q); }
            const float4 gz = make_float4(go.x * (o.x > 0.f ? 1.f : slope), go.y * (o.y > 0.f ? 1.f : slope), go.z * (o.z > 0.f ? 1.f : slope), go.w * (o.w > 0.f ? 1.f : slope));
            if (rv) *reinterpret_cast<float4 *>(gZ + r * D + 16 * g + 4 * q) = gz;
            const float zv[4] = {gz.x, gz.y, gz.z, gz.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float *w = Wl + (16 * g + 4 * q + j) * S + c;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(zv[j], w[16 * nt], acc[nt], 0, 0, 0);
            }
        }
        // tiles [0, D/16) hold gS, tiles [D/16, 2D/16) hold gT, both in the C layout: rows 4 q + i, column 16 nt + c
#pragma unroll
        for (int nt = 0; nt < D / 16; ++nt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const long long row = tile * 16 + 4 * q + i;
                if (row < n_rows) {
                    const long long o = row * D + 16 * nt + c;
                    const float gs = acc[nt][i], gt = acc[nt + D / 16][i];
                    gP[o] = fmaf(gt, E[o], gs);
                    gE[o] = fmaf(gt, P[o], gs);
                }
            }
    }
}

template <int CP>
__device__ __forceinline__ void ngcf_load_cp(const float *__restrict__ p, float *o) {
    if constexpr (CP % 4 == 0) {
#pragma unroll
        for (int k = 0; k < CP / 4; ++k) { const float4 v = *reinterpret_cast<const float4 *>(p + 4 * k); o[4 * k] = v.x; o[4 * k + 1] = v.y; o[4 * k + 2] = v.z; o[4 * k + 3] = v.w; }
    } else if constexpr (CP == 2) { const float2 v = *reinterpret_cast<const float2 *>(p); o[0] = v.x; o[1] = v.y; }
    else o[0] = p[0];
}

// [gW1; gW2] partial of one workgroup.  A = [S | T]^T tile (M = 16 rows of gW, K = 4 data rows), B = gZ tile (K = 4 data rows, N = 16 columns of
// gW).  M and N indices are PERMUTED so that one 16-B load per lane serves CP = D/16 tiles: lane c holds columns [c CP, (c + 1) CP) of its data row
// (a full 256-B row per 16 lanes, coalesced), and tile t covers the columns {c CP + t}.  Every wave forms S, T for all 2 CP row tiles of gW
// and owns the column tiles t = wave, wave + 4, ... (one gZ scalar per K step each).
template <int D>
__global__ __launch_bounds__(kNgcfBlock) void ngcf_dense_wgrad_kernel(const float *__restrict__ P, const float *__restrict__ E, const float *__restrict__ gZ,
                                                                       long long n_rows, float *__restrict__ partial) {
    constexpr int CP = D / 16, NPW = (CP + 3) / 4;
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4, wave = threadIdx.x >> 6;
    f32x4v acc[2 * CP][NPW];
#pragma unroll
    for (int a = 0; a < 2 * CP; ++a)
#pragma unroll
        for (int n = 0; n < NPW; ++n) acc[a][n] = (f32x4v){0.f, 0.f, 0.f, 0.f};
    const long long n_tiles = (n_rows + 15) / 16;
    for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
#pragma unroll
        for (int st = 0; st < 4; ++st) {                                   // K step: data rows tile * 16 + 4 st + q
            const long long row = tile * 16 + 4 * st + q;
            const bool rv = row < n_rows;
            float sv[CP], tv[CP], bv[NPW], pv[CP], ev[CP];
#pragma unroll
            for (int k = 0; k < CP; ++k) pv[k] = ev[k] = 0.f;
            if (rv) { ngcf_load_cp<CP>(P + row * D + c * CP, pv); ngcf_load_cp<CP>(E + row * D + c * CP, ev); }      // 16-B loads: a full row per 16 lanes
#pragma unroll
            for (int k = 0; k < CP; ++k) { sv[k] = pv[k] + ev[k]; tv[k] = pv[k] * ev[k]; }
#pragma unroll
            for (int n = 0; n < NPW; ++n) { const int t = wave + 4 * n; bv[n] = (rv && t < CP) ? gZ[row * D + c * CP + t] : 0.f; }
#pragma unroll
            for (int n = 0; n < NPW; ++n)
                if (wave + 4 * n < CP) {
#pragma unroll
                    for (int k = 0; k < CP; ++k) {
                        acc[k][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(sv[k], bv[n], acc[k][n], 0, 0, 0);
                        acc[CP + k][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(tv[k], bv[n], acc[CP + k][n], 0, 0, 0);
                    }
                }
        }
    }
    // C layout: acc[a][n][i] = gW[row m, column nn] with m = (4 q + i) CP + (a mod CP) (+ D for the T half), nn = c CP + (wave + 4 n)
    float *dst = partial + (size_t)blockIdx.x * 2 * D * D;
#pragma unroll
    for (int n = 0; n < NPW; ++n) {
        const int t = wave + 4 * n;
        if (t < CP)
#pragma unroll
            for (int a = 0; a < 2 * CP; ++a)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int m = (a < CP ? 0 : D) + (4 * q + i) * CP + (a % CP);
                    dst[m * D + c * CP + t] = acc[a][n][i];
                }
    }
}

// gW[e] = sum over the partials, in a fixed order: 16 elements x 16 partial lanes per workgroup, then an LDS tree
__global__ __launch_bounds__(kBlock) void ngcf_wgrad_fold_kernel(const float *__restrict__ partial, int n_part, int n_elem, float *__restrict__ gW) {
    __shared__ float red[16][17];
    const int e = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int t = blockIdx.x * 16 + e;
    float s = 0.f;
    if (t < n_elem)
        for (int b = pl; b < n_part; b += 16) s += partial[(size_t)b * n_elem + t];
    red[pl][e] = s;
    __syncthreads();
    if (pl == 0 && t < n_elem) {
        float r = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) r += red[k][e];
        gW[t] = r;
    }
}
// ================================================================================================
// SimGCL perturbation (recommender/SimGCL.py:203-205)
// ================================================================================================
__global__ __launch_bounds__(kBlock) void simgcl_perturb_kernel(float *__restrict__ E, const float *__restrict__ noise, int n, int d, float eps) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (r >= n) return;
    float s = 0.f;
    for (int k = lane; k < d; k += kWave) { const float x = noise[(size_t)r * d + k]; s = fmaf(x, x, s); }
    s = wave_sum(s);
    const float nr = fmaxf(sqrtf(s), 1e-12f);
    for (int k = lane; k < d; k += kWave) {
        const float x = E[(size_t)r * d + k];
        const float sg = x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f);
        E[(size_t)r * d + k] = x + sg * (noise[(size_t)r * d + k] / nr) * eps;
    }
}

// The same perturbation with the noise drawn inside the kernel: u(row, k) = uniform [0,1) from a counter-based hash of
// (seed, stream, row_id * d + k) -- what torch.rand_like(ego) supplies in recommender/SimGCL.py:203-205, without materialising an
// [N, d] noise table, a clone of the operand, or a second pass.  dst may alias src.  row_ids (optional): the operand holds the listed
// rows of a larger table and gets exactly the noise those rows would get in a full-table call with the same (seed, stream).
__device__ __forceinline__ unsigned long long arl_splitmix64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__global__ __launch_bounds__(kBlock) void simgcl_perturb_rng_kernel(const float *__restrict__ src, float *__restrict__ dst, int n, int d,
                                                                     const int32_t *__restrict__ row_ids, float eps, unsigned long long key) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (r >= n) return;
    const unsigned long long base = (unsigned long long)(row_ids ? row_ids[r] : r) * (unsigned long long)d;
    float u[4];                                                    // d <= 256: at most four columns per lane
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int k = lane + q * kWave;
        u[q] = 0.f;
        if (k < d) {
            u[q] = (float)(arl_splitmix64(key ^ (base + k)) >> 40) * (1.0f / 16777216.0f);      // 24 random bits -> [0, 1)
            s = fmaf(u[q], u[q], s);
        }
    }
    s = wave_sum(s);
    const float sc = eps / fmaxf(sqrtf(s), 1e-12f);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int k = lane + q * kWave;
        if (k < d) {
            const float x = src[(size_t)r * d + k];
            const float sg = x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f);
            dst[(size_t)r * d + k] = x + sg * u[q] * sc;
        }
    }
}

// ================================================================================================
// CLeaR spectral-feature-augmentation L1 term (attack/White/CLeaR.py:98-125)
// ================================================================================================
// H = rows of X with multiplicities w (H is never materialised: a real user's row appears T times, a target's row U times,
// a negative's row as often as it closes somebody's top-k list).  With q = H r0, r = H^T q, s = H r, Q = |r|^2:
//   H_aug - H = -s r^T / Q   =>   loss = mean|H_aug - H| = (sum_i |s_i|)(sum_j |r_j|) / (numel(H) Q)
// and, through everything (r is not detached in the reference),
//   dloss/dh_i = c1 sgn(s_i) r + q_i g_r + (h_i . g_r) r0,   c1 = A/(numel Q),
//   g_r = [ (A/Q) a + (S/Q) sgn(r) - (2 S A / Q^2) r ] / numel,   a = H^T sgn(s), S = sum|s|, A = sum|r|.
constexpr int kSfaStride = 264;          // floats per partial record: d (<= 256) accumulators + 1 scalar, padded
constexpr int kSfaMaxBlocks = 1024;

__device__ __forceinline__ float sgnf(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }

// PASS 1: dotv = r0, rowdot -> q, acc = sum w q x.      PASS 2: dotv = r, rowdot -> s, acc = sum w sgn(s) x, scalar = sum w |s|.
template <int PASS>
__global__ __launch_bounds__(kBlock) void sfa_reduce_pass_kernel(const float *__restrict__ X, const float *__restrict__ w, const float *__restrict__ dotv,
                                                                 int n, int d, float *__restrict__ rowdot, float *__restrict__ part) {
    __shared__ float red[kWavesPerBlock][kSfaStride];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float v[4], acc[4] = {0.f, 0.f, 0.f, 0.f}, sc = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = (lane + 64 * j < d) ? dotv[lane + 64 * j] : 0.f;
    for (int row = blockIdx.x * kWavesPerBlock + wave; row < n; row += gridDim.x * kWavesPerBlock) {
        const float wr = w[row];
        if (wr == 0.f) { if (lane == 0) rowdot[row] = 0.f; continue; }
        float x[4], t = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { x[j] = (lane + 64 * j < d) ? X[(size_t)row * d + lane + 64 * j] : 0.f; t = fmaf(x[j], v[j], t); }
        t = wave_sum(t);
        if (lane == 0) rowdot[row] = t;
        const float c = PASS == 1 ? wr * t : wr * sgnf(t);
        if (PASS == 2) sc += wr * fabsf(t);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = fmaf(c, x[j], acc[j]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) if (lane + 64 * j < d) red[wave][lane + 64 * j] = acc[j];
    if (lane == 0) red[wave][kSfaStride - 1] = sc;
    __syncthreads();
    for (int k = threadIdx.x; k < kSfaStride; k += kBlock) {
        if (k < d || k == kSfaStride - 1) {
            float t = 0.f;
#pragma unroll
            for (int wv = 0; wv < kWavesPerBlock; ++wv) t += red[wv][k];
            part[(size_t)blockIdx.x * kSfaStride + k] = t;
        }
    }
}

__device__ __forceinline__ float block_sum_256(float v, float *sh) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

// coef layout: [0,256) r   [256,512) a, then g_r   [512] c1   [513] loss   [514] S
// Column-wise fold of the per-block partial records in a fixed order: block j sums column j (j < d) or the scalar slot (j == d).
__global__ __launch_bounds__(kBlock) void sfa_fold_kernel(const float *__restrict__ part, int nblk, int d, float *__restrict__ out, float *__restrict__ scalar_out) {
    __shared__ float sh[4];
    const int j = blockIdx.x;
    const int slot = j < d ? j : kSfaStride - 1;
    float t = 0.f;
    for (int b = threadIdx.x; b < nblk; b += kBlock) t += part[(size_t)b * kSfaStride + slot];
    t = block_sum_256(t, sh);
    if (threadIdx.x == 0) { if (j < d) out[j] = t; else scalar_out[0] = t; }
}

__global__ __launch_bounds__(kBlock) void sfa_finalize_kernel(int d, float inv_numel, float *__restrict__ coef, float *__restrict__ loss_out) {
    __shared__ float sh[4];
    const int k = threadIdx.x;
    const float a = k < d ? coef[256 + k] : 0.f, S = coef[514];
    const float r = k < d ? coef[k] : 0.f;
    const float A = block_sum_256(fabsf(r), sh);
    const float Q = block_sum_256(r * r, sh);
    if (k < d) coef[256 + k] = inv_numel * ((A / Q) * a + (S / Q) * sgnf(r) - (2.f * S * A / (Q * Q)) * r);
    if (k == 0) { coef[512] = inv_numel * A / Q; coef[513] = inv_numel * S * A / Q; loss_out[0] = coef[513]; }
}

__global__ __launch_bounds__(kBlock) void sfa_grad_kernel(const float *__restrict__ X, const float *__restrict__ w, const float *__restrict__ r0,
                                                          const float *__restrict__ q, const float *__restrict__ s, const float *__restrict__ coef,
                                                          int n, int d, float scale, int accumulate, float *__restrict__ G) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (row >= n) return;
    const float wr = w[row];
    if (wr == 0.f) {
        if (!accumulate) for (int k = lane; k < d; k += kWave) G[(size_t)row * d + k] = 0.f;
        return;
    }
    float x[4], t = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { x[j] = (lane + 64 * j < d) ? X[(size_t)row * d + lane + 64 * j] : 0.f; t = fmaf(x[j], (lane + 64 * j < d) ? coef[256 + lane + 64 * j] : 0.f, t); }
    t = wave_sum(t);                                      // h . g_r
    const float c_r = coef[512] * sgnf(s[row]), c_g = q[row], ws = wr * scale;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = lane + 64 * j;
        if (k < d) {
            const float g = ws * (c_r * coef[k] + c_g * coef[256 + k] + t * r0[k]);
            G[(size_t)row * d + k] = accumulate ? G[(size_t)row * d + k] + g : g;
        }
    }
}

// ================================================================================================
// Attack primitives
// ================================================================================================
// out[t, j] += <dY[rows[t]], X[col_off+j]>.  Block = 64 items staged in LDS (stride d+1: conflict-free column
// reads), 4 waves stride over the selected rows; the dY row is wave-uniform (broadcast loads).
__global__ __launch_bounds__(kBlock) void sddmm_rows_dense_kernel(const float *__restrict__ dY, const float *__restrict__ X, int d,
                                                                   const int32_t *__restrict__ rows, int n_sel, long long col_off, int n_cols,
                                                                   float *__restrict__ out) {
    extern __shared__ float xs[];            // [64][d+1]
    const int j0 = blockIdx.x * 64;
    const int ld = d + 1;
    for (int t = threadIdx.x; t < 64 * d; t += kBlock) {
        const int jj = t / d, k = t % d;
        xs[jj * ld + k] = (j0 + jj < n_cols) ? X[(size_t)(col_off + j0 + jj) * d + k] : 0.f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int j = j0 + lane;
    for (int t = wv; t < n_sel; t += kWavesPerBlock) {
        const float *dy = dY + (size_t)rows[t] * d;
        float s = 0.f;
        for (int k = 0; k < d; ++k) s = fmaf(dy[k], xs[lane * ld + k], s);
        if (j < n_cols) out[(size_t)t * n_cols + j] += s;
    }
}

// d = 64 form: a lane keeps ITS item row in 64 registers, a wave owns 64 items and walks all listed rows; dY[rows[t]] is wave-uniform, so
// its 64 values arrive through scalar loads and every product is one v_fmac with a scalar operand (the general kernel above spends a
// broadcast global load and an LDS read per FMA).
__global__ __launch_bounds__(kBlock) void sddmm_rows_dense64_kernel(const float *__restrict__ dY, const float *__restrict__ X, const int32_t *__restrict__ rows,
                                                                     int n_sel, long long col_off, int n_cols, float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int j = (blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6)) * 64 + lane;
    float x[64];
    const float4 *xr = reinterpret_cast<const float4 *>(X + (size_t)(col_off + min(j, n_cols - 1)) * 64);
#pragma unroll
    for (int q = 0; q < 16; ++q) { const float4 v = xr[q]; x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w; }
    for (int t = 0; t < n_sel; ++t) {
        const float *dy = dY + (size_t)__builtin_amdgcn_readfirstlane(rows[t]) * 64;
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 64; ++k) s = fmaf(dy[k], x[k], s);
        if (j < n_cols) out[(size_t)t * n_cols + j] += s;
    }
}

// S = clamp(S - 0.2*tanh(g)), g = dinv_r[row] * dinv_c[col] * grad, and g = 0 where S has no stored entry (S == 0):
// autograd.grad w.r.t. the sparse adjacency only yields pattern entries (attack/White/PGA.py:117-139).
// ---- the F x I fake-user block of the poisoned adjacency as two dense products (attack/White/PGA.py:118-134: the reference multiplies
// by a dense (U+F+I)^2 matrix; the factored operator needs only S (D^-1/2 X)_items for the fake users' rows and S^T (D^-1/2 X)_fake for
// the item rows).  Both kernels stage a 64 x 64 tile of S in LDS per wave and read it back as broadcast float4s: four FMAs per LDS
// read, lane = embedding column, the accumulators of a tile in registers; plain fp32 FMA chains in a fixed order (deterministic).
constexpr int kFbT = 64;                 // tile edge (rows of S, items)
constexpr int kFbChunk = 64;             // items per wave task of the row product (split-K: partials are combined in task order)

__device__ __forceinline__ void fb_stage_tile(float *tile, const float *__restrict__ S, int F, long long I, int f0, long long i0, int lane) {
    // tile[f][j] = S[f0 + f][i0 + j] (0 where out of range); 16 passes of 64 lanes x 4 floats
#pragma unroll 4
    for (int pass = 0; pass < 16; ++pass) {
        const int idx = pass * 64 + lane, f = idx >> 4, part = idx & 15;
        const long long i = i0 + 4 * part;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (f0 + f < F) {
            const float *src = S + (long long)(f0 + f) * I + i;
            if (i + 3 < I && ((((long long)(f0 + f) * I + i) & 3) == 0)) v = *reinterpret_cast<const float4 *>(src);
            else { if (i < I) v.x = src[0]; if (i + 1 < I) v.y = src[1]; if (i + 2 < I) v.z = src[2]; if (i + 3 < I) v.w = src[3]; }
        }
        *reinterpret_cast<float4 *>(tile + f * kFbT + 4 * part) = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// partial[task][f][c] = sum over the task's items (kFbChunk, in 64-item tiles) of S[f0 + f][i] * X[i][c0 + c]
__global__ __launch_bounds__(kBlock) void fake_block_rows_kernel(const float *__restrict__ S, int F, long long I, const float *__restrict__ X, int d,
                                                                  int n_chunks, float *__restrict__ partial) {
    __shared__ float tiles[kWavesPerBlock][kFbT * kFbT];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int chunk = blockIdx.x * kWavesPerBlock + wv;
    if (chunk >= n_chunks) return;
    const int c0 = blockIdx.y * 64, f0 = blockIdx.z * kFbT;
    const int col = c0 + lane;
    float *tile = tiles[wv];
    float acc[kFbT];
#pragma unroll
    for (int f = 0; f < kFbT; ++f) acc[f] = 0.f;
    for (int sub = 0; sub < kFbChunk / kFbT; ++sub) {
        const long long i0 = (long long)chunk * kFbChunk + sub * kFbT;
        if (i0 >= I) break;
        float x[kFbT];                                             // the tile's operand rows, this lane's column: all loads in flight together
#pragma unroll
        for (int j = 0; j < kFbT; ++j) x[j] = (i0 + j < I && col < d) ? X[(i0 + j) * d + col] : 0.f;
        __builtin_amdgcn_wave_barrier();
        fb_stage_tile(tile, S, F, I, f0, i0, lane);
#pragma unroll
        for (int f = 0; f < kFbT; ++f) {
#pragma unroll
            for (int j4 = 0; j4 < kFbT / 4; ++j4) {
                const float4 s4 = *reinterpret_cast<const float4 *>(tile + f * kFbT + 4 * j4);
                acc[f] = fmaf(s4.x, x[4 * j4], acc[f]); acc[f] = fmaf(s4.y, x[4 * j4 + 1], acc[f]);
                acc[f] = fmaf(s4.z, x[4 * j4 + 2], acc[f]); acc[f] = fmaf(s4.w, x[4 * j4 + 3], acc[f]);
            }
        }
    }
    float *o = partial + (((size_t)blockIdx.z * gridDim.y + blockIdx.y) * n_chunks + chunk) * (kFbT * 64);
#pragma unroll
    for (int f = 0; f < kFbT; ++f) o[f * 64 + lane] = acc[f];
}

// Y[f][c] += alpha * rscale[f] * sum_task partial[task][f][c].  One 1024-thread workgroup per output row and 64-column block: lane =
// column, the 16 waves take the tasks round-robin (task k -> wave k % 16, ascending), their sums are added in wave order: a fixed tree.
constexpr int kFbFinishWaves = 16;
__global__ __launch_bounds__(kFbFinishWaves * 64) void fake_block_rows_finish_kernel(const float *__restrict__ partial, int n_chunks, int F, int d,
                                                                                    const float *__restrict__ rscale, float alpha, float *__restrict__ Y) {
    __shared__ float part[kFbFinishWaves][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int f = blockIdx.x, cy = blockIdx.y, ny = gridDim.y;
    const int fz = f / kFbT;
    const float *p = partial + ((size_t)fz * ny + cy) * n_chunks * (kFbT * 64) + (f % kFbT) * 64 + lane;
    float sum = 0.f;
    for (int k = wv; k < n_chunks; k += kFbFinishWaves) sum += p[(size_t)k * (kFbT * 64)];
    part[wv][lane] = sum;
    __syncthreads();
    const int c = cy * 64 + lane;
    if (wv == 0 && c < d) {
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < kFbFinishWaves; ++w) tot += part[w][lane];
        Y[(size_t)f * d + c] += alpha * (rscale ? rscale[f] : 1.f) * tot;
    }
}

// Y[i][c] += alpha * rscale[i] * sum_f S[f][i] * Xf[f][c]: a wave owns 64 items x 64 columns
__global__ __launch_bounds__(kBlock) void fake_block_cols_kernel(const float *__restrict__ S, int F, long long I, const float *__restrict__ Xf, int d,
                                                                  const float *__restrict__ rscale, float alpha, float *__restrict__ Y) {
    __shared__ float tiles[kWavesPerBlock][kFbT * kFbT];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long i0 = ((long long)blockIdx.x * kWavesPerBlock + wv) * kFbT;
    if (i0 >= I) return;
    const int col = blockIdx.y * 64 + lane;
    float *tile = tiles[wv];
    float acc[kFbT];
#pragma unroll
    for (int j = 0; j < kFbT; ++j) acc[j] = 0.f;
    for (int f0 = 0; f0 < F; f0 += kFbT) {
        __builtin_amdgcn_wave_barrier();
        fb_stage_tile(tile, S, F, I, f0, i0, lane);
        float xfv[kFbT];                                           // this lane's column of the tile's fake-user rows, loaded together
#pragma unroll
        for (int f = 0; f < kFbT; ++f) xfv[f] = (f0 + f < F && col < d) ? Xf[(size_t)(f0 + f) * d + col] : 0.f;
#pragma unroll
        for (int f = 0; f < kFbT; ++f) {                           // rows past F were staged as zeros
            const float xf = xfv[f];
#pragma unroll
            for (int j4 = 0; j4 < kFbT / 4; ++j4) {
                const float4 s4 = *reinterpret_cast<const float4 *>(tile + f * kFbT + 4 * j4);
                acc[4 * j4] = fmaf(s4.x, xf, acc[4 * j4]); acc[4 * j4 + 1] = fmaf(s4.y, xf, acc[4 * j4 + 1]);
                acc[4 * j4 + 2] = fmaf(s4.z, xf, acc[4 * j4 + 2]); acc[4 * j4 + 3] = fmaf(s4.w, xf, acc[4 * j4 + 3]);
            }
        }
    }
    if (col >= d) return;
#pragma unroll
    for (int j = 0; j < kFbT; ++j) {
        const long long i = i0 + j;
        if (i < I) Y[i * d + col] += alpha * (rscale ? rscale[i] : 1.f) * acc[j];
    }
}

__global__ __launch_bounds__(kBlock) void pga_update_kernel(float *__restrict__ S, const float *__restrict__ grad, const float *__restrict__ dinv_r,
                                                             const float *__restrict__ dinv_c, long long rows, long long cols) {
    const long long n = rows * cols;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) {
        const long long r = i / cols, c = i - r * cols;
        const float s0 = S[i];
        float g = grad[i];
        if (dinv_r) g *= dinv_r[r];
        if (dinv_c) g *= dinv_c[c];
        if (s0 == 0.f) g = 0.f;
        float s = s0 - 0.2f * tanhf(g);
        if (s > 1.f) s = 1.f;
        if (s <= 0.f) s = 10e-8f;
        S[i] = s;
    }
}

// ---- streaming score + mask + top-k -------------------------------------------------------------
// Candidate ordering key: larger score first, then smaller item id.  Packed so that a plain 64-bit unsigned
// compare orders candidates: high 32 bits = order-preserving map of the float, low 32 bits = ~item.
__device__ __forceinline__ unsigned long long pack_cand(float s, int item) {
    unsigned u = __float_as_uint(s);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((unsigned long long)u << 32) | (unsigned)(~(unsigned)item);
}
__device__ __forceinline__ float cand_score(unsigned long long c) {
    unsigned u = (unsigned)(c >> 32);
    u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    return __uint_as_float(u);
}
__device__ __forceinline__ int cand_item(unsigned long long c) { return (int)(~(unsigned)(c & 0xffffffffu)); }

constexpr int kTU = 16;          // users per block
constexpr int kTI = 256;         // items per tile (one per thread)
constexpr int kCap = 512;        // candidate slots per user (>= kTI + 2*k_max)

// bitonic sort (descending) of one user's kCap candidate keys by one wavefront
__device__ void wave_sort_desc(unsigned long long *c, int lane) {
    for (int k = 2; k <= kCap; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = lane; t < kCap; t += kWave) {
                const int ixj = t ^ j;
                if (ixj > t) {
                    const unsigned long long a = c[t], b = c[ixj];
                    const bool desc = ((t & k) == 0);
                    if (desc ? (a < b) : (a > b)) { c[t] = b; c[ixj] = a; }
                }
            }
            __builtin_amdgcn_wave_barrier();
            __threadfence_block();
        }
    }
}

__global__ __launch_bounds__(kBlock) void score_mask_topk_kernel(const float *__restrict__ Pu, const float *__restrict__ Pi, int U, int I, int d,
                                                                  const int32_t *__restrict__ mrp, const int32_t *__restrict__ mcol, int k,
                                                                  int32_t *__restrict__ top_idx, float *__restrict__ top_val) {
    extern __shared__ unsigned char smem_raw[];
    unsigned long long *cand = reinterpret_cast<unsigned long long *>(smem_raw);          // [kTU][kCap]
    float *us = reinterpret_cast<float *>(cand + kTU * kCap);                               // [kTU][d]
    unsigned long long *thr = reinterpret_cast<unsigned long long *>(us + kTU * 256);     // [kTU]
    int *cnt = reinterpret_cast<int *>(thr + kTU);                                          // [kTU]
    const int u0 = blockIdx.x * kTU;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int t = tid; t < kTU * d; t += kBlock) {
        const int uu = t / d, kk = t % d;
        us[uu * 256 + kk] = (u0 + uu < U) ? Pu[(size_t)(u0 + uu) * d + kk] : 0.f;
    }
    for (int t = tid; t < kTU * kCap; t += kBlock) cand[t] = 0ull;       // 0 sorts below every real candidate
    if (tid < kTU) { thr[tid] = 0ull; cnt[tid] = 0; }
    __syncthreads();
    for (int i0 = 0; i0 < I; i0 += kTI) {
        const int item = i0 + tid;
        if (item < I) {
            const float *pi = Pi + (size_t)item * d;
            float sc[kTU];
#pragma unroll
            for (int uu = 0; uu < kTU; ++uu) sc[uu] = 0.f;
            for (int kk = 0; kk < d; kk += 4) {
                const float4 x = *reinterpret_cast<const float4 *>(pi + kk);
#pragma unroll
                for (int uu = 0; uu < kTU; ++uu) {
                    const float4 y = *reinterpret_cast<const float4 *>(us + uu * 256 + kk);
                    sc[uu] = fmaf(x.x, y.x, sc[uu]); sc[uu] = fmaf(x.y, y.y, sc[uu]); sc[uu] = fmaf(x.z, y.z, sc[uu]); sc[uu] = fmaf(x.w, y.w, sc[uu]);
                }
            }
#pragma unroll
            for (int uu = 0; uu < kTU; ++uu) {
                if (u0 + uu >= U) continue;
                unsigned long long key = pack_cand(sc[uu], item);
                if (key > thr[uu]) {
                    if (mrp) {          // interacted -> -10e8 (only evaluated for the rare candidates that pass the threshold)
                        int lo = mrp[u0 + uu], hi = mrp[u0 + uu + 1];
                        const int end = hi;
                        while (lo < hi) { const int mid = (lo + hi) >> 1; if (mcol[mid] < item) lo = mid + 1; else hi = mid; }
                        if (lo < end && mcol[lo] == item) key = pack_cand(-10e8f, item);
                    }
                    if (key > thr[uu]) {
                        const int slot = atomicAdd(&cnt[uu], 1);
                        cand[uu * kCap + slot] = key;       // slot < kCap guaranteed: cnt <= kCap - kTI at tile start
                    }
                }
            }
        }
        __syncthreads();
        // compaction: any user whose buffer could overflow in the next tile is sorted and cut to its top k
        for (int uu = wv; uu < kTU; uu += kWavesPerBlock) {
            if (cnt[uu] > kCap - kTI) {
                unsigned long long *c = cand + uu * kCap;
                for (int t = cnt[uu] + lane; t < kCap; t += kWave) c[t] = 0ull;
                __builtin_amdgcn_wave_barrier();
                __threadfence_block();
                wave_sort_desc(c, lane);
                if (lane == 0) { thr[uu] = c[k - 1]; cnt[uu] = k; }
            }
        }
        __syncthreads();
    }
    for (int uu = wv; uu < kTU; uu += kWavesPerBlock) {
        if (u0 + uu >= U) continue;
        unsigned long long *c = cand + uu * kCap;
        for (int t = cnt[uu] + lane; t < kCap; t += kWave) c[t] = 0ull;
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
        wave_sort_desc(c, lane);
        for (int t = lane; t < k; t += kWave) {
            top_idx[(size_t)(u0 + uu) * k + t] = cand_item(c[t]);
            top_val[(size_t)(u0 + uu) * k + t] = cand_score(c[t]);
        }
    }
}

// ---- MFMA form of the streaming score + mask + top-k (d in {16,32,64,128}, k <= 64) ---------------------------------------
// scores = Pu . Pi^T is the one dense contraction of the path (2*U*I*d flop; 1.3e13 at cfg2), so it goes on the matrix cores.
// A block owns 128 users for the whole kernel (their A fragments stay in registers) and streams item tiles of Pi through a
// ring of four LDS images without block barriers (per-slot fill / done counters, global loads two stages ahead; calls over long
// item streams open with a bootstrap pass that pre-sets the users' thresholds -- both described in the kernel).  8 waves, 16 users each
// (16x16 MFMA shapes): lane l supplies A[user l&15][k] and B[k][item l&15] for its contiguous k range [Q*(l>>4), Q*(l>>4)+Q)
// (a permutation of the contraction index, irrelevant to the sum); C: item = l&15, user row = 4*(l>>4) + reg.
// Top-k per user is a sorted list in registers (see `tk_hi/tk_lo` in the kernel).
// Measured on MI355X: the f32 MFMA shares the SIMD's issue with the VALU (a second wave per SIMD hides latencies but its VALU
// work does NOT overlap the other wave's f32 MFMAs), so every VALU instruction of the pre-filter/insert path is paid in full --
// hence one subtract + one funnel shift per score, one ballot per phase, and the split-bf16 form below for the contraction.
#ifndef ARL_TOPK_SPLIT_MODE
#define ARL_TOPK_SPLIT_MODE 2
#endif
// Waves per workgroup (16 users each).  The fp16-split forms run more than two waves per SIMD -- further waves to fill the matrix pipe and
// the vector issue while the others wait: d = 64 with 16 waves = four per SIMD (124-126 registers once the fragments of a stage are loaded
// two sub-tiles at a time; 8 -> 12 -> 16 waves: 19.6 -> 17.6 -> 16.3 ms at 200 K x 100 K, cfg2 masked pass 86 -> 69 -> 63 ms), d = 128 with
// 12 = three per SIMD (146 registers with the fragments loaded one sub-tile at a time: 25.8 -> 21.7 ms at 192 K x 100 K); the exact-fp32 forms (174-182 registers) keep 8.
#ifndef ARL_TOPK_D128_WAVES
#define ARL_TOPK_D128_WAVES 12
#endif
#ifndef ARL_TOPK_D64_WAVES
#define ARL_TOPK_D64_WAVES 16
#endif
#ifndef ARL_TOPK_MIN_WAVES_EU
#define ARL_TOPK_MIN_WAVES_EU 1                   // (developer knob: smaller workgroups, e.g. -DARL_TOPK_D64_WAVES=8, need 4 here to stay at 128 registers = two workgroups per CU)
#endif
constexpr int topk_waves(int D, bool SPLIT) { return (SPLIT && ARL_TOPK_SPLIT_MODE == 2) ? (D == 64 ? ARL_TOPK_D64_WAVES : (D == 128 ? ARL_TOPK_D128_WAVES : 8)) : 8; }
#ifdef ARL_TOPK_PROF
#define ARL_PROF_DECL long long P_acc[7] = {0, 0, 0, 0, 0, 0, 0}, P_t0 = clock64(); const long long P_start = P_t0;
#define ARL_PROF_TICK(SLOT) { const long long P_t = clock64(); P_acc[SLOT] += P_t - P_t0; P_t0 = P_t; }
#else
#define ARL_PROF_DECL
#define ARL_PROF_TICK(SLOT)
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x16 __attribute__((ext_vector_type(16)));
constexpr int kBloomWords = 32;      // 1024 bits per user
// Candidate queues (ARL_TOPK_QUEUE = 1): a pre-filter survivor is APPENDED to its user row's queue in LDS by the lane that found it (one LDS
// atomic + one 8-byte store, all survivors of a 16-item x 16-user sub-tile in parallel) instead of being broadcast and inserted by the whole
// wave on the spot; a row's queue is merged into its sorted register list when it holds kQFlush entries (looked at every ARL_TOPK_QCHK-th stage) or is full,
// and at the end of the stream.  Thresholds move only at a merge -- a threshold that lags admits a few candidates more (they fall off at the
// merge), it never loses one -- and the per-candidate broadcast / row-select / threshold-select work of the immediate insert is paid once per merge.
#ifndef ARL_TOPK_QUEUE
#define ARL_TOPK_QUEUE 1
#endif
#ifndef ARL_TOPK_QCAP
#define ARL_TOPK_QCAP 16
#endif
#ifndef ARL_TOPK_QFLUSH
#define ARL_TOPK_QFLUSH 12
#endif
constexpr int kQCap = ARL_TOPK_QCAP;             // queue slots per user row and wave
constexpr int kQFlush = ARL_TOPK_QFLUSH;         // merge a row's queue at the end of a stage once it holds this many
constexpr int kQWords = 16 + 16 * kQCap * 2;      // per wave: 16 counters + 16 x kQCap 8-byte keys, in 4-byte words
#ifndef ARL_TOPK_RING
#define ARL_TOPK_RING 4
#endif
constexpr int kTopkRing = ARL_TOPK_RING;         // staged item tiles in LDS (slots of the ring), a power of two
#ifndef ARL_TOPK_LEAD
#define ARL_TOPK_LEAD 2
#endif
#ifndef ARL_TOPK_ESCALE
#define ARL_TOPK_ESCALE 1.05f                    // (probes only: anything below 1.05 voids the bound)
#endif
#ifndef ARL_TOPK_RGRP
#define ARL_TOPK_RGRP 2
#endif
#ifndef ARL_TOPK_PAIRS
#define ARL_TOPK_PAIRS 0
#endif
#ifndef ARL_TOPK_REFINE
#define ARL_TOPK_REFINE 1                        // fp16 split form: stream scores from the high pieces only, exact three-product scores for queued candidates (see the kernel)
#endif
#ifndef ARL_TOPK_SUBSKIP
#define ARL_TOPK_SUBSKIP 1
#endif
#ifndef ARL_TOPK_POLL8
#define ARL_TOPK_POLL8 0
#endif
#ifndef ARL_TOPK_SIGNAL_WAIT
#define ARL_TOPK_SIGNAL_WAIT 1
#endif
#ifndef ARL_TOPK_QCHK
#define ARL_TOPK_QCHK 2                          // the per-row queue counts are looked at every ARL_TOPK_QCHK-th stage (full queues are merged when they fill, whatever this is)
#endif
#ifndef ARL_TOPK_PIPE
#define ARL_TOPK_PIPE 1
#endif
constexpr int kTopkLead = ARL_TOPK_LEAD;         // a wave writes its share of stage s + kTopkLead while it consumes stage s (even, < ring)
#ifndef ARL_TOPK_BOOT_ITEMS
#define ARL_TOPK_BOOT_ITEMS 4096
#endif
constexpr int kTopkBootItems = ARL_TOPK_BOOT_ITEMS; // items scored by the bootstrap pass of a cold call (a multiple of every stage size)
__device__ __forceinline__ unsigned bloom_hash(int item) { return ((unsigned)item * 2654435761u) >> 22; }

// SPLIT = true: the contraction runs on the bf16 matrix path with every fp32 operand split into three bf16 pieces
// (x = hi + mid + lo exactly up to 2^-25 |x|) and the six partial products hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi
// accumulated in fp32, smallest first: the dropped terms are <= 3 * 2^-24 of |a||b| per product, the size of the rounding
// differences between two fp32 summation orders.  Six bf16 MFMAs at 16x the fp32 rate = 2.7x fewer matrix cycles, and -- unlike
// the fp32 MFMA -- they leave the SIMD's vector issue free for the other wave's pre-filter and inserts.  `Pi` is then the
// pre-split image [I][3][D] bf16 written by split_bf16x3_kernel.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) unsigned lds_u32;

// Split form used by SPLIT = true (compile-time choice; both kept):
//   1: three bf16 pieces, six products (above);
//   2: TWO fp16 pieces of the operand scaled by a power of two, three products ah*bh + ah*bl + al*bh.  Each table is scaled so that its
//      largest magnitude lands in [2^13, 2^14): the high piece keeps 11 significant bits, the low piece the next 11 (it stays a normal
//      fp16 number for every element within 2^12 of the table's maximum), every product of two pieces is exact in fp32, and the dropped
//      al*bl term is <= 2^-22 |a||b| -- 2^-21 per product in all, the size of the rounding differences between two fp32 summation orders
//      of a d = 64 dot product.  Half the matrix work and 2/3 of the LDS and global bytes of form 1.  Scores, thresholds and the sorted
//      lists live in the scaled domain (scale = 2^(eu + ei), exact); only the final values are scaled back.
constexpr int kSplitMode = ARL_TOPK_SPLIT_MODE;
constexpr int kSplitPlanes = kSplitMode == 1 ? 3 : 2;

// largest |x| of a table as float bits (non-negative floats order like unsigned ints): *out must be zeroed first
__global__ __launch_bounds__(kBlock) void absmax_bits_kernel(const float *__restrict__ X, long long n, unsigned *__restrict__ out) {
    // 16-byte loads, one atomic per WORKGROUP: 65 K same-address atomics (one per wave of the old form) serialise at ~12 ns each -- 0.8 ms
    // for a 256 MB table that streams in 50 us
    __shared__ unsigned part[kWavesPerBlock];
    unsigned m = 0u;
    const bool al = ((uintptr_t)X & 15u) == 0;
    const long long n4 = al ? n / 4 : 0;
    const float4 *X4 = reinterpret_cast<const float4 *>(X);
    for (long long t = (long long)blockIdx.x * kBlock + threadIdx.x; t < n4; t += (long long)gridDim.x * kBlock) {
        const float4 v = X4[t];
        m = max(max(m, __float_as_uint(fabsf(v.x))), max(__float_as_uint(fabsf(v.y)), max(__float_as_uint(fabsf(v.z)), __float_as_uint(fabsf(v.w)))));
    }
    for (long long t = n4 * 4 + (long long)blockIdx.x * kBlock + threadIdx.x; t < n; t += (long long)gridDim.x * kBlock) m = max(m, __float_as_uint(fabsf(X[t])));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kWavesPerBlock; ++w) m = max(m, part[w]);
        if (m) atomicMax(out, m);
    }
}
// power of two that brings a table whose largest magnitude has float bits `mbits` into [2^13, 2^14) (1 for an all-zero or non-finite table)
__device__ __forceinline__ float split_scale(unsigned mbits) {
    const int e = (int)(mbits >> 23);                              // biased exponent of the maximum
    if (e == 0 || e >= 255) return 1.f;
    const int se = min(max(127 + 13 - (e - 127), 127 - 40), 127 + 40);      // 2^(13 - (e - 127)), kept within 2^+-40 (the product of two scales must not overflow)
    return __uint_as_float((unsigned)se << 23);
}
// `order` (optional): image row r holds table row order[r] -- the item stream in another order than the table's (see arl_score_mask_topk_f32)
__global__ __launch_bounds__(kBlock) void split_f16x2_kernel(const float *__restrict__ X, long long n, int d, const unsigned *__restrict__ mbits,
                                                              _Float16 *__restrict__ out, const int32_t *__restrict__ order) {
    const long long t = (long long)blockIdx.x * kBlock + threadIdx.x;          // one element each
    if (t >= n) return;
    const long long row = t / d;
    const int kcol = (int)(t - row * d);
    const float x = (order ? X[(long long)order[row] * d + kcol] : X[t]) * split_scale(*mbits);
    const _Float16 h = (_Float16)x;
    _Float16 *o = out + row * 2 * d + kcol;
    o[0] = h; o[d] = (_Float16)(x - (float)h);
}

__global__ __launch_bounds__(kBlock) void invert_perm_kernel(const int32_t *__restrict__ order, int n, int32_t *__restrict__ pos_of) {
    const int t = blockIdx.x * kBlock + threadIdx.x;
    if (t < n) pos_of[order[t]] = t;
}

__global__ __launch_bounds__(kBlock) void split_bf16x3_kernel(const float *__restrict__ X, long long n, int d, __bf16 *__restrict__ out) {
    const long long t = (long long)blockIdx.x * kBlock + threadIdx.x;          // one element each
    if (t >= n) return;
    const long long row = t / d;
    const int kcol = (int)(t - row * d);
    const float x = X[t];
    const __bf16 h = (__bf16)x;
    const float r1 = x - (float)h;
    const __bf16 m = (__bf16)r1;
    const __bf16 l = (__bf16)(r1 - (float)m);
    __bf16 *o = out + row * 3 * d + kcol;
    o[0] = h; o[d] = m; o[2 * d] = l;
}

// largest scaled row norm |b * scale| of every stage of the item stream (positions [st * mst, st * mst + mst) of the staged image) and, behind
// them, of the whole table (float bits, an atomic maximum: zeroed first): the per-stage factor of the high-piece stream's bound E
__global__ __launch_bounds__(kWave) void stage_norm_kernel(const float *__restrict__ X, int I, int d, int mst, const unsigned *__restrict__ mbits,
                                                            const int32_t *__restrict__ order, int nstages, float *__restrict__ out) {
    const int st = blockIdx.x, lane = threadIdx.x;
    const int p = st * mst + lane;
    float n2 = 0.f;
    if (lane < mst && p < I) {
        const float *x = X + (size_t)(order ? order[p] : p) * d;
        for (int t = 0; t < d; t += 4) {
            const float4 v = *reinterpret_cast<const float4 *>(x + t);
            n2 = fmaf(v.x, v.x, n2); n2 = fmaf(v.y, v.y, n2); n2 = fmaf(v.z, v.z, n2); n2 = fmaf(v.w, v.w, n2);
        }
    }
    float nm = sqrtf(n2) * split_scale(*mbits);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) nm = fmaxf(nm, __shfl_xor(nm, off));
    if (lane == 0) {
        out[st] = nm;
        atomicMax(reinterpret_cast<unsigned *>(out + nstages), __float_as_uint(nm));
    }
}

#ifndef ARL_TOPK_EXIT
#define ARL_TOPK_EXIT 1                          // exact early exit of the norm-ordered stream (see the kernel); 0 = the round-3 form
#endif
#ifndef ARL_TOPK_EXIT_EVERY
#define ARL_TOPK_EXIT_EVERY 8                    // a wave tests its 16 users' bound every this-many stages (a power of two)
#endif
// suffix maxima of the per-stage norms: out[s] = max(sn[s..nst)), out[nst] = 0 -- the largest scaled item norm any LATER stage of the stream can hold
// (the early exit's bound; for a norm-ordered stream it equals sn).  One wave.
// pick[0] / pick[1] (optional): 1 / 0 when the early exit can pay on this norm profile, 0 / 1 otherwise (the gates of the two kernel builds).  A row can
// only finish when the remaining items' norms have fallen below cos * (norm of its k-th best item), cos = the cosine of that pair -- well under 1.  The
// exit is enabled when the stage at 7/8 of the stream lies below half of the stage that holds the 4096th largest norm (where the bootstrap sample ends:
// the region the first thresholds come from): measured on 1 M x 100 K, i.i.d. normal tables (ratio 0.73) and one-hop propagated tables (0.58) skip nothing
// and keep the plain build; log-normal item norms (0.05) skip 84 % of the stream.
__global__ __launch_bounds__(kWave) void stage_sufmax_kernel(const float *__restrict__ sn, int nst, float *__restrict__ out, int mst, int *__restrict__ pick) {
    const int lane = threadIdx.x;
    if (pick != nullptr) {                                          // (suffix maxima of the two stages, so that a stream in any other order than by norm keeps the plain build)
        const int head = min(nst - 1, 4096 / mst), tail = min(nst - 1, nst - nst / 8);
        float mh = 0.f, mt = 0.f;
        for (int s = head + lane; s < nst; s += kWave) { const float v = sn[s]; mh = fmaxf(mh, v); if (s >= tail) mt = fmaxf(mt, v); }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { mh = fmaxf(mh, __shfl_xor(mh, off)); mt = fmaxf(mt, __shfl_xor(mt, off)); }
        if (lane == 0) {
            const int on = (ARL_TOPK_EXIT != 0) && head < tail && mt < 0.5f * mh;
            pick[0] = on; pick[1] = !on;
        }
    }
    float carry = 0.f;
    for (int base = ((nst - 1) / kWave) * kWave; base >= 0; base -= kWave) {
        const int s = base + lane;
        float v = s < nst ? sn[s] : 0.f;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
            const float o = __shfl_down(v, off);
            if (lane + off < kWave) v = fmaxf(v, o);
        }
        v = fmaxf(v, carry);
        if (s < nst) out[s] = v;
        carry = __shfl(v, 0);
    }
    if (lane == 0) out[nst] = 0.f;
}

constexpr int kExitWords = 20;                   // LDS words behind the ring counters: [0] stop stage, [4 .. 4 + waves) one "finished" word per wave
// Early exit bound.  A high-piece score is <ah, bh> accumulated in fp32: |score| <= |ah| |bh| (1 + 2^-18) (Cauchy-Schwarz; 64 or 128 exact products,
// fp32 adds) <= |a'| |b'| (1 + 2^-10 + 2^-17) + Eabs (each piece within 2^-11 relative of its element, or 2^-25 absolute when subnormal: that part is
// what Eabs bounds).  With 1.05 * 2^-10 instead of 2^-10 covering the roundings of the two norms themselves, an item of scaled norm n cannot pass the
// pre-filter `score >= threshold - (Ereg n + Eabs)`, Ereg = ESCALE * 2^-10 |a'|, of a row with  (|a'| (1 + 1.05 * 2^-10) + Ereg) n + 2 Eabs < threshold.
// No register is spent on |a'|: it is Ereg / (ESCALE * 2^-10), so the bound is kExitK * Ereg * n + 2 Eabs (the kernel runs at 128 VGPRs with no slack).
constexpr float kExitK = (1.f + 1.05f * 0.0009765625f) / (ARL_TOPK_ESCALE * 0.0009765625f > 0.f ? ARL_TOPK_ESCALE * 0.0009765625f : 1.f) + 1.f;

template <int D, bool SPLIT, bool WARM, bool XIT = false>
__global__ __launch_bounds__(64 * topk_waves(D, SPLIT), ARL_TOPK_MIN_WAVES_EU) void score_mask_topk_mfma16_kernel(const float *__restrict__ Pu, const void *__restrict__ Pi_image, int U, int I,
                                                                            const int32_t *__restrict__ mrp, const int32_t *__restrict__ mcol, int k,
                                                                            int32_t *__restrict__ top_idx, float *__restrict__ top_val,
                                                                            const float *__restrict__ Pi_f32, const int32_t *__restrict__ warm_idx,
                                                                            int *__restrict__ underflow, const unsigned *__restrict__ table_max_bits,
                                                                            const int32_t *__restrict__ item_order, const int32_t *__restrict__ item_pos,
                                                                            const int *__restrict__ gate, const float *__restrict__ stage_norm,
                                                                            unsigned long long *__restrict__ stats, const int *__restrict__ gate2) {
    // gate (optional): the launch is the cold repeat of a warm-started call and runs only if that call raised its underflow flag -- decided here,
    // on the device, so that the host never waits for the flag (every thread of the grid takes the same branch)
    if (gate != nullptr && *gate == 0) return;
    // gate2 (optional): which BUILD of the kernel runs this call.  XIT = true carries the early exit of the norm-ordered stream; its extra code costs ~5 % on
    // tables where nothing can be skipped (the stream loop has no register to spare), so both builds are launched and the device picks one from the
    // items' norm profile (stage_sufmax_kernel) -- the other returns here.
    if (gate2 != nullptr && *gate2 == 0) return;
    // item_order / item_pos (both or neither): the staged image holds table row item_order[p] at position p (item_pos = the inverse).  Scores,
    // stages and the bootstrap sample are in POSITIONS; a candidate becomes an item id when it is packed into a key, so masks, keys (ties:
    // lower item id first) and results are those of the table order -- only the order in which thresholds rise changes.
    constexpr int kM16Block = 64 * topk_waves(D, SPLIT), kMU = 16 * topk_waves(D, SPLIT);       // threads / users per workgroup
    constexpr int Q = D / 4;                                       // contraction indices per lane: [Q*g, Q*g + Q)
    constexpr int SRCB = SPLIT ? kSplitPlanes * D * 2 : D * 4;     // bytes per item row in global memory
    // LDS image of a staged tile.  The hardware services a ds_read_b128 in four fixed 16-lane groups that mix lanes of two
    // k-groups (g, g+1); with plain [row][plane][k] rows every group hit two banks twice (SQ_LDS_BANK_CONFLICT = 44 % of the LDS
    // cycles).  Two half images -- k-groups with g even / g odd -- a multiple of 256 B apart, rows of an odd number of 16-B units:
    // every group tiles the 64 banks exactly.
    //   offset(row, plane, g, piece) = (g&1)*HALF + row*RH + ((plane*2 + (g>>1))*PPG + piece)*16
    constexpr int NPL = SPLIT ? kSplitPlanes : 1;                  // planes (pieces of the split) per item row
    constexpr int PPG = SPLIT ? Q / 8 : Q / 4;                     // 16-byte pieces per k-group and plane
    constexpr int RH = NPL * 2 * PPG * 16 + 16;
    constexpr int HALF = (D <= 16 ? 128 : (D <= 64 ? 64 : 32)) * RH;
    static_assert(HALF % 256 == 0 && (RH / 16) % 2 == 1, "half images must be bank-aligned");
    constexpr int MST = D <= 16 ? 128 : (D <= 64 ? 64 : 32);       // items per stage (one slot of the ring)
    constexpr int NSUB = MST / 16;                                 // 16-item sub-tiles per stage
    constexpr int SPP = NSUB >= 2 ? 2 : 1;                         // sub-tiles per insert phase (at most 32 items)
    [[maybe_unused]] constexpr int NPH = NSUB / SPP;
    [[maybe_unused]] constexpr int NSC = 4 * SPP;                  // scores per lane and phase
    static_assert(!SPLIT || (D % 32 == 0), "the bf16 path contracts 32 indices per MFMA");
    extern __shared__ unsigned char smem_raw[];
    unsigned char *bt = smem_raw;                                  // two staged item tiles (STAGEB bytes each), then the Bloom filters
    const unsigned char *Pi = reinterpret_cast<const unsigned char *>(Pi_image);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int u_base = blockIdx.x * kMU + wv * 16;                 // this wave's 16 users
    float a[Q];                                                    // A fragment: user (u_base + c), columns [Q*g, Q*g + Q)
    {
        const int u = u_base + c;
#pragma unroll
        for (int t = 0; t < Q; t += 4) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (u < U) v = *reinterpret_cast<const float4 *>(Pu + (size_t)u * D + Q * g + t);
            a[t] = v.x; a[t + 1] = v.y; a[t + 2] = v.z; a[t + 3] = v.w;
        }
    }
    constexpr int KS = SPLIT ? Q / 8 : 1;                          // 16-bit MFMAs (8 indices per lane each) per operand pair
    constexpr bool F16 = SPLIT && kSplitMode == 2;
    constexpr bool REFINE = F16 && ARL_TOPK_REFINE && ARL_TOPK_QUEUE;      // see the staging constants below
    float Ereg[4] = {0.f, 0.f, 0.f, 0.f};                          // REFINE: E of user rows 4g + reg for an item of scaled norm n is Ereg * n + Eabs
    float Eabs = 0.f;
    bf16x8 af[3][KS];
    f16x8 ah[2][KS];
    // scaled domain of the fp16 form: scores = true scores * score_scale (a power of two); 1 otherwise
    float score_scale = 1.f, score_unscale = 1.f;
    if constexpr (F16) {
        const float su = split_scale(table_max_bits[1]), si = split_scale(table_max_bits[0]);
        score_scale = su * si; score_unscale = (1.f / su) * (1.f / si);
        float n2 = 0.f;
#pragma unroll
        for (int t = 0; t < Q; ++t) {
            const float x = a[t] * su;
            const _Float16 h = (_Float16)x;
            ah[0][t / 8][t % 8] = h; ah[1][t / 8][t % 8] = (_Float16)(x - (float)h);
            n2 = fmaf(x, x, n2);
        }
        if constexpr (REFINE) {
            n2 += __shfl_xor(n2, 16); n2 += __shfl_xor(n2, 32);   // |a'|^2 of user c (the four k-groups of a column hold a quarter each)
            // E(user c, item) = 1.05 * 2^-10 * |a'| * |b'|: the stream takes |b'| per STAGE (stage_norm, the largest scaled row norm of the stage's
            // items -- decreasing along a norm-ordered stream), the bootstrap and the warm start the table's largest.
            // (+ an absolute term for pieces that are subnormal fp16 numbers -- elements 2^27 below their table's maximum: rounding error <= 2^-25
            // instead of 2^-11 relative; it matters only for rows that are zero to fp32 precision next to the rest of their table)
            const float bm = __uint_as_float(table_max_bits[0]) * si, am = __uint_as_float(table_max_bits[1]) * su;
            Eabs = 1.1920929e-7f * (float)D * (bm + am);
            const float Ec = ARL_TOPK_ESCALE * 0.0009765625f * sqrtf(n2);
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) Ereg[reg] = __shfl(Ec, 4 * g + reg);
        }
    } else if constexpr (SPLIT) {
#pragma unroll
        for (int t = 0; t < Q; ++t) {
            const __bf16 h = (__bf16)a[t];
            const float r1 = a[t] - (float)h;
            const __bf16 m = (__bf16)r1;
            af[0][t / 8][t % 8] = h; af[1][t / 8][t % 8] = m; af[2][t / 8][t % 8] = (__bf16)(r1 - (float)m);
        }
    }
    // REFINE: the stream contracts the HIGH fp16 pieces only (one product instead of three, half the staged bytes and fragment reads); that score
    // differs from the three-product score by at most E = 1.05 * 2^-10 * |a| * max|b| (each piece keeps 11 bits; Cauchy-Schwarz), so the pre-filter
    // runs against threshold - E and every queued candidate gets its three-product score -- the same MFMA sequence on the same operands as
    // the one-pass form, bit for bit -- when its row's queue is merged (up to 8 candidates per 16 x 16 tile, fragments gathered from the image).
    constexpr int C16 = SRCB / 16;                                 // 16-byte pieces per item row
    constexpr int C16S = REFINE ? C16 / kSplitPlanes : C16;        // ... of which the stream stages these (the first plane)
    constexpr int F4 = MST * C16S;
    constexpr int PER = (F4 + kM16Block - 1) / kM16Block;
    static_assert(PER >= 1 && PER <= 6, "staging assumes one to six 16-byte pieces per thread");
    const int nstages = (I + MST - 1) / MST;                       // item stages
    // Calls over long item streams open with a BOOTSTRAP pass over the first kTopkBootItems items (NB stages; scores only, no lists): every lane keeps the
    // four best scores it sees for each of its four user rows (a lane sees the items congruent to its column mod 16), so a row's 16
    // lanes end with 64 scores of DISTINCT items.  The (k + m)-th best of those -- m = the row's interacted items inside the sample,
    // which the mask may take out -- is a lower bound of the row's final k-th best score: the stream then restarts from item 0 with it
    // as the starting threshold (the warm-start mechanism), and the ~k ln(I/k) record-setters of a cold stream (half of them in its
    // first 2 %) drop to ~k (1 + ln(I / sample)).  Same instruction sequence on the same data: the sample's scores are bit-identical in
    // both passes, so the k items the bound rests on pass it again.
    // sample size: kTopkBootItems, at most a quarter of the stream, an even number of stages; streams under 32 K items run cold
    const int NB = I >= 32768 ? (min(kTopkBootItems, I / 4) / (2 * MST)) * 2 : 0;
    const int nvirt = NB + nstages;                                // stages as the ring counts them: the bootstrap's, then the stream's
    auto item_stage = [&](int v) { return v < NB ? v : v - NB; };
    auto stage_ptr = [&](int st, int p) {
        const int f = tid + p * kM16Block;
        const int item = min(st * MST + f / C16S, I - 1);          // clamped, never selected on: rows past I are masked out of pm
        return reinterpret_cast<const float4 *>(Pi + (size_t)item * SRCB + (f % C16S) * 16);
    };
    auto lds_ptr = [&](unsigned char *buf, int p) {
        const int f = tid + p * kM16Block;
        const int row = f / C16S, piece = f % C16S;                // piece = plane * (4*PPG) + j, j-th 16-B run of the plane
        const int pl = piece / (4 * PPG), j = piece % (4 * PPG), gq = j / PPG, ks = j % PPG;
        return reinterpret_cast<float4 *>(buf + (gq & 1) * HALF + row * RH + ((pl * 2 + (gq >> 1)) * PPG + ks) * 16);
    };
    // Warm start (optional): `warm_idx` holds k distinct candidate items per user -- typically the previous call's result, when the
    // tables moved a little (the surrogate loops of CLeaR / DLAttack, the per-epoch evaluation).  Their scores under the CURRENT
    // tables, lowered by a bound on the difference between this plain fp32 dot product and the streamed contraction, give a valid
    // starting threshold: all k candidates will pass it when they stream by, so the result is unchanged, but the ~k ln(I/k)
    // record-setters of a cold stream shrink to about k.  (A candidate that has become masked breaks the guarantee: rows that end
    // with fewer than k keys raise `underflow` and the caller repeats the call cold.)
    float thr0v = -INFINITY;                                       // lane r: starting threshold of user row r
    if constexpr (WARM) {
        for (int r = 0; r < 16; ++r) {
            const int u = u_base + r;
            if (u >= U) break;
            const float *pu = Pu + (size_t)u * D;
            const int cand_j = lane < k ? min(max(warm_idx[(size_t)u * k + lane], 0), I - 1) : 0;
            const float *pi = Pi_f32 + (size_t)cand_j * D;
            float sdot = 0.f, ni = 0.f, nu = 0.f;
            for (int t = 0; t < D; t += 4) {
                const float4 x = *reinterpret_cast<const float4 *>(pu + t), y = *reinterpret_cast<const float4 *>(pi + t);
                sdot = fmaf(x.x, y.x, sdot); sdot = fmaf(x.y, y.y, sdot); sdot = fmaf(x.z, y.z, sdot); sdot = fmaf(x.w, y.w, sdot);
                ni = fmaf(y.x, y.x, ni); ni = fmaf(y.y, y.y, ni); ni = fmaf(y.z, y.z, ni); ni = fmaf(y.w, y.w, ni);
                nu = fmaf(x.x, x.x, nu); nu = fmaf(x.y, x.y, nu); nu = fmaf(x.z, x.z, nu); nu = fmaf(x.w, x.w, nu);
            }
            float lb = lane < k ? (sdot - 8e-6f * sqrtf(nu * ni)) * score_scale - 1e-30f : INFINITY;     // thresholds live in the scaled domain
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) lb = fminf(lb, __shfl_xor(lb, off));
            if (lane == r) thr0v = lb;
        }
    }
    float thrf[4];                                                 // exact running k-th best score of user rows 4g + reg (pre-filter)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const float t0 = WARM ? __shfl(thr0v, 4 * g + reg) : -INFINITY;
        thrf[reg] = (u_base + 4 * g + reg < U) ? t0 : INFINITY;    // users past U never insert (cold calls: reset after the bootstrap)
    }
    // The running top-k of each of the wave's 16 users is a SORTED list held in registers: lane j of tk[r] is the j-th largest key
    // of user row r (k <= 64 = one key per lane; 0 = empty, below every real key).  An insert is one 64-bit compare + ballot for
    // the position and a one-lane DPP shift of the tail -- no candidate buffers, no compaction, no final sort, and the threshold
    // (lane k-1) is exact after every insert, so exactly the stream's record-setters (about k ln(I/k) per user) are ever handled.
    // (two 16-register vectors indexed by the wave-uniform row number: the compiler addresses them through M0, no 16-way branch)
    u32x16 tk_hi = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tk_lo = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    // Interacted-item mask: a 1024-bit Bloom filter per user in LDS answers "not interacted" for ~97 % of the pre-filter
    // survivors with one LDS read; only filter hits pay the binary search in global memory (7 dependent L2 round trips).
    constexpr int STAGEB = 2 * HALF;
    unsigned *ring_ctr = reinterpret_cast<unsigned *>(bt + kTopkRing * STAGEB);                             // fill[kTopkRing], done[kTopkRing]
    constexpr int kQAll = ARL_TOPK_QUEUE ? kQWords * (kM16Block / kWave) : 0;
    unsigned *exit_st = ring_ctr + 2 * kTopkRing;                                                            // [0] stop stage (virtual index), [4 + w] wave w's rows are finished
    unsigned *qcnt = ring_ctr + 2 * kTopkRing + kExitWords + wv * kQWords;                                                // this wave's 16 queue counters ...
    unsigned long long *qkey = reinterpret_cast<unsigned long long *>(qcnt + 16);                           // ... and its [16][kQCap] keys (8-byte aligned)
    unsigned *bloom = ring_ctr + 2 * kTopkRing + kExitWords + kQAll + wv * 16 * kBloomWords;                             // [16][kBloomWords]
    if (tid < 2 * kTopkRing) ring_ctr[tid] = 0u;
    if (tid < kExitWords) exit_st[tid] = tid == 0 ? 0x7fffffffu : 0u;
    if (ARL_TOPK_QUEUE && lane < 16) qcnt[lane] = 0u;
    if (mrp) {
        for (int t = lane; t < 16 * kBloomWords; t += kWave) bloom[t] = 0u;
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
        for (int ulw = 0; ulw < 16; ++ulw) {
            const int u = u_base + ulw;
            if (u >= U) break;
            for (int e = mrp[u] + lane, end = mrp[u + 1]; e < end; e += kWave) {
                const unsigned hb = bloom_hash(mcol[e]);
                atomicOr(&bloom[ulw * kBloomWords + (hb >> 5)], 1u << (hb & 31u));
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
#if !ARL_TOPK_QUEUE
    // insert `key` into the sorted list of user row `row` (wave-uniform); returns the row's new k-th best score = its threshold
    auto insert_sorted = [&](int row, unsigned long long key) -> float {
        const unsigned Kh = tk_hi[row], Kl = tk_lo[row];
        const unsigned long long K = ((unsigned long long)Kh << 32) | Kl;
        const int pos = __popcll(__ballot(K > key));               // keys above the new one: a prefix of the lanes
        unsigned nh = Kh, nl = Kl;
        if (pos < k) {                                             // wave-uniform
            const unsigned sl = __builtin_amdgcn_update_dpp(Kl, Kl, 0x138, 0xf, 0xf, false);            // wave_shr:1
            const unsigned sh = __builtin_amdgcn_update_dpp(Kh, Kh, 0x138, 0xf, 0xf, false);
            nl = lane < pos ? Kl : (lane == pos ? (unsigned)key : sl);
            nh = lane < pos ? Kh : (lane == pos ? (unsigned)(key >> 32) : sh);
            tk_hi[row] = nh; tk_lo[row] = nl;
        }
        const unsigned th = (unsigned)__builtin_amdgcn_readlane((int)nh, k - 1);
        const unsigned tl = (unsigned)__builtin_amdgcn_readlane((int)nl, k - 1);
        const float t0 = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(thr0v), row));
        return (th | tl) ? fmaxf(cand_score((unsigned long long)th << 32), t0) : t0;     // t0: the warm-start / bootstrap bound (-inf if none)
    };
    // Pre-filter + inserts for the scores of one stage (16*SPP items x 16 users per phase: NSC scores per lane,
    // bit b = 4*s2 + reg  <->  item item0 + 16*s2 + c, user row 4g + reg).
    auto bookkeeping = [&](const f32x4 (&ac)[NSUB], int st) {
#pragma unroll
        for (int ph = 0; ph < NPH; ++ph) {
            const int item0 = st * MST + ph * 16 * SPP;
            float scv[NSC];
#pragma unroll
            for (int b = 0; b < NSC; ++b) scv[b] = ac[SPP * ph + (b >> 2)][b & 3];
            // sign(score - threshold) shifted into a mask: two VALU ops per score, no VCC round trip (score == threshold passes;
            // threshold -inf always passes, +inf never)
            unsigned fails = 0u;
#pragma unroll
            for (int b = NSC - 1; b >= 0; --b) fails = __builtin_amdgcn_alignbit(fails, __float_as_uint(scv[b] - thrf[b & 3]), 31);
            unsigned pm = ~fails & ((1u << NSC) - 1u);
            if (st == nstages - 1) {                                       // rows past I were staged as copies of item I-1
#pragma unroll
                for (int s2 = 0; s2 < SPP; ++s2)
                    if (item0 + 16 * s2 + c >= I) pm &= ~(0xfu << (4 * s2));
            }
            // Survivors are taken one at a time by the whole wave.  Each pending lane first prepares ITS lowest pending candidate
            // (score select, key packing) in parallel; the serial part only broadcasts it.
            for (unsigned long long pend = __ballot(pm != 0u); pend != 0ull;) {
                const int b = __ffs(pm) - 1;                               // per lane (garbage where pm == 0: never read)
                const float s01 = (b & 1) ? scv[1] : scv[0], s23 = (b & 1) ? scv[3] : scv[2];
                float sc = (b & 2) ? s23 : s01;
                if constexpr (NSC == 8) {
                    const float s45 = (b & 1) ? scv[5] : scv[4], s67 = (b & 1) ? scv[7] : scv[6];
                    const float s47 = (b & 2) ? s67 : s45;
                    sc = (b & 4) ? s47 : sc;
                }
                const int item = item0 + 16 * (b >> 2) + c;
                const int ulw = 4 * g + (b & 3);
                if (mrp && pm != 0u) {                                     // interacted -> -10e8 (pre-filter survivors only)
                    const unsigned hb = bloom_hash(item);
                    if ((bloom[ulw * kBloomWords + (hb >> 5)] >> (hb & 31u)) & 1u) {
                        const int u = u_base + ulw;
                        int lo = mrp[u], hi = mrp[u + 1];
                        const int end = hi;
                        while (lo < hi) { const int mid = (lo + hi) >> 1; if (mcol[mid] < item) lo = mid + 1; else hi = mid; }
                        if (lo < end && mcol[lo] == item) sc = -10e8f * score_scale;      // (-10e8 once scaled back) stays in the list only while it holds fewer than k real items
                    }
                }
                const unsigned long long mykey = pack_cand(sc, item);
                pm &= pm - 1u;                                             // this round's candidate leaves the mask (0 stays 0)
                for (unsigned long long round = pend; round != 0ull; round &= round - 1ull) {
                    const int src = __ffsll((long long)round) - 1;
                    const unsigned kh = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(mykey >> 32), src);
                    const unsigned kl = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)mykey, src);
                    const int row = __builtin_amdgcn_readlane(ulw, src);
                    const float nt = insert_sorted(row, ((unsigned long long)kh << 32) | kl);
#pragma unroll
                    for (int j = 0; j < 4; ++j) thrf[j] = (4 * g + j == row) ? nt : thrf[j];
                }
                pend = __ballot(pm != 0u);
            }
        }
    };
#endif
#if ARL_TOPK_QUEUE
#ifndef ARL_TOPK_EXP
#define ARL_TOPK_EXP 0
#endif
    float exp_sink = 0.f;                                              // (developer experiments only: keeps values alive)
    // merge the queued candidates of user row `row` (wave-uniform) into its sorted list; the row's threshold becomes exact again.  A queue
    // entry is the raw (score, item) pair; the interacted-item mask and the key packing happen here, one candidate per lane.
    auto flush_row = [&](int row) {
        const int cnt = __builtin_amdgcn_readfirstlane((int)*(volatile lds_u32 *)(qcnt + row));
        const int m = min(cnt, kQCap);
        unsigned long long cj = 0ull;
        unsigned long long e = 0ull;
        if (lane < m) e = qkey[row * kQCap + lane];
        float sc_exact = 0.f;
        if constexpr (REFINE) {
            // three-product scores of the queued candidates: candidate j is column j of one 16 x 16 tile (fragments gathered from the staged
            // image at its stream position), the wave's own user fragments are the rows; the row's scores sit in the lanes of k-group row / 4
            static_assert(kQCap <= 16, "a queue's candidates are the columns of one tile");
            const int pos_c = __shfl((int)(unsigned)(e >> 32), c);             // 0 for columns without a candidate: row 0 of the image, never read back
            const unsigned char *src = Pi + (size_t)pos_c * SRCB + (size_t)(g * PPG) * 16;
            f16x8 fb[2][KS];
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) fb[pl][ks] = *reinterpret_cast<const f16x8 *>(src + pl * (D * 2) + ks * 16);
            constexpr int TA[3] = {0, 1, 0}, TB[3] = {1, 0, 0};                 // the one-pass form's order: ah*bl, al*bh, ah*bh
            f32x4 ca = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int term = 0; term < 3; ++term)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) ca = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[TA[term]][ks], fb[TB[term]][ks], ca, 0, 0, 0);
            const int rr = row & 3;
            const float pick = rr == 0 ? ca[0] : (rr == 1 ? ca[1] : (rr == 2 ? ca[2] : ca[3]));
            sc_exact = __shfl(pick, c + 16 * (row >> 2));
        }
        if (lane < m) {
            float sc = REFINE ? sc_exact : __uint_as_float((unsigned)e);
            const int pos = (int)(unsigned)(e >> 32);
            const int item = item_order ? item_order[pos] : pos;
            if (mrp) {                                                 // interacted -> -10e8 (pre-filter survivors only)
                const unsigned hb = bloom_hash(item);
                if ((bloom[row * kBloomWords + (hb >> 5)] >> (hb & 31u)) & 1u) {
                    const int u = u_base + row;
                    int lo = mrp[u], hi = mrp[u + 1];
                    const int end = hi;
                    while (lo < hi) { const int mid = (lo + hi) >> 1; if (mcol[mid] < item) lo = mid + 1; else hi = mid; }
                    if (lo < end && mcol[lo] == item) sc = -10e8f * score_scale;      // (-10e8 once scaled back) stays in the list only while it holds fewer than k real items
                }
            }
            cj = pack_cand(sc, item);
        }
        if (lane == 0) qcnt[row] = 0u;
        if (ARL_TOPK_EXP == 4) { exp_sink += (float)(unsigned)cj; return; }       // experiment: appends + queue reads, no list merges (results are wrong)
        unsigned Kh = tk_hi[row], Kl = tk_lo[row];
        // only candidates above the list's k-th key as it stands can enter (the queue's thresholds lag, and REFINE's are lowered by E)
        const unsigned long long Kk = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)Kh, k - 1) << 32) | (unsigned)__builtin_amdgcn_readlane((int)Kl, k - 1);
        for (unsigned long long todo = __ballot(lane < m && cj > Kk); todo != 0ull; todo &= todo - 1ull) {
            const int j = __ffsll((long long)todo) - 1;
            const unsigned ch = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(cj >> 32), j);
            const unsigned cl = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)cj, j);
            const unsigned long long key = ((unsigned long long)ch << 32) | cl, K = ((unsigned long long)Kh << 32) | Kl;
            const int pos = __popcll(__ballot(K > key));               // keys above the new one: a prefix of the lanes
            if (pos < k) {                                             // wave-uniform
                const unsigned sl = __builtin_amdgcn_update_dpp(Kl, Kl, 0x138, 0xf, 0xf, false);            // wave_shr:1
                const unsigned sh = __builtin_amdgcn_update_dpp(Kh, Kh, 0x138, 0xf, 0xf, false);
                Kl = lane < pos ? Kl : (lane == pos ? cl : sl);
                Kh = lane < pos ? Kh : (lane == pos ? ch : sh);
            }
        }
        tk_hi[row] = Kh; tk_lo[row] = Kl;
        const unsigned th = (unsigned)__builtin_amdgcn_readlane((int)Kh, k - 1);
        const unsigned tl = (unsigned)__builtin_amdgcn_readlane((int)Kl, k - 1);
        const float t0 = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(thr0v), row));
        const float nt = (th | tl) ? __builtin_fmaxf(cand_score((unsigned long long)th << 32), t0) : t0;     // t0: the warm-start / bootstrap bound (-inf if none)
        const bool mine = (g == (row >> 2));
        const int rj = row & 3;
#pragma unroll
        for (int j = 0; j < 4; ++j) thrf[j] = (mine && rj == j) ? nt : thrf[j];
    };
    auto flush_rows_with = [&](unsigned at_least) {                    // every row whose queue holds at least `at_least` candidates
        const unsigned cn = lane < 16 ? *(volatile lds_u32 *)(qcnt + lane) : 0u;
        for (unsigned long long need = __ballot(cn >= at_least); need != 0ull; need &= need - 1ull) flush_row(__ffsll((long long)need) - 1);
    };
    // Pre-filter + queue appends for the scores of one stage (accumulator ac[sub][reg] <-> item st*MST + 16*sub + c, user row 4g + reg).
    // Straight line: one compare per score; if any lane of the wave passes anywhere, every passing (sub, reg) appends its lanes' (score, item)
    // pairs -- an LDS atomic for the slot, one 8-byte store.  A queue that is full sends its lanes to the rare path at the end.
    auto bookkeeping = [&](const f32x4 (&ac)[NSUB], int st, float sn) {      // sn: the stage's largest scaled item norm (REFINE)
        if (ARL_TOPK_EXP == 1 || ARL_TOPK_EXP >= 5) {                  // experiment: scores only, no pre-filter, no lists
#pragma unroll
            for (int sub = 0; sub < NSUB; ++sub) exp_sink += ac[sub][0] + ac[sub][3];
            return;
        }
        f32x4 sv[NSUB];
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) sv[sub] = ac[sub];
        if (st == nstages - 1) {                                       // rows past I were staged as copies of item I-1: their scores never pass (NaN)
#pragma unroll
            for (int sub = 0; sub < NSUB; ++sub)
                if (st * MST + 16 * sub + c >= I) sv[sub] = f32x4{__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf("")};
        }
        float tf[4];                                                   // the stage's pre-filter thresholds: exact thresholds lowered by the stage's E
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) tf[reg] = REFINE ? thrf[reg] - fmaf(Ereg[reg], sn, Eabs) : thrf[reg];
        bool pass[NSUB][4], psub[NSUB];
        bool some = false;
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) {
            psub[sub] = false;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                pass[sub][reg] = sv[sub][reg] >= tf[reg];              // one compare per score (-inf always passes, +inf and NaN never)
                psub[sub] = psub[sub] || pass[sub][reg];
            }
            some = some || psub[sub];
        }
        if (ARL_TOPK_EXP == 2) { exp_sink += some ? 1.f : 0.f; return; }    // experiment: compares only
        if (ARL_TOPK_EXP == 3) { exp_sink += some ? 1.f : 0.f; flush_rows_with((unsigned)kQFlush); return; }    // experiment: compares + the stage-end queue check (queues stay empty)
        if (__builtin_amdgcn_ballot_w64(some) != 0ull) {
            unsigned left = 0u;                                        // bit 4*sub + reg: this lane's candidate found its queue full
            const int item0 = st * MST + c;
#pragma unroll
            for (int sub = 0; sub < NSUB; ++sub) {
                // two levels of skipping: a wave sees ~2 survivors per stage, so most sub-tiles have none (one branch instead of four)
                if (ARL_TOPK_SUBSKIP && __builtin_amdgcn_ballot_w64(psub[sub]) == 0ull) continue;
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    if (pass[sub][reg]) {
                        const int ulw = 4 * g + reg;
                        const unsigned slot = __hip_atomic_fetch_add((lds_u32 *)(qcnt + ulw), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (slot < (unsigned)kQCap) qkey[ulw * kQCap + slot] = ((unsigned long long)(unsigned)(item0 + 16 * sub) << 32) | __float_as_uint(sv[sub][reg]);
                        else left |= 1u << (4 * sub + reg);
                    }
                }
            }
            // rare path (cold starts, bursts): merge the full queues, then the lanes left over try again
            while (__builtin_amdgcn_ballot_w64(left != 0u) != 0ull) {
                flush_rows_with((unsigned)kQCap);
                if (left != 0u) {                                      // one candidate per lane and round
                    const int b = __ffs(left) - 1;
                    float sc = sv[0][0];
#pragma unroll
                    for (int q = 1; q < 4 * NSUB; ++q) sc = (b == q) ? sv[q >> 2][q & 3] : sc;
                    const int ulw = 4 * g + (b & 3);
                    const unsigned slot = __hip_atomic_fetch_add((lds_u32 *)(qcnt + ulw), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (slot < (unsigned)kQCap) {
                        qkey[ulw * kQCap + slot] = ((unsigned long long)(unsigned)(item0 + 16 * (b >> 2)) << 32) | __float_as_uint(sc);
                        left &= left - 1u;
                    }
                }
            }
        }
        if (ARL_TOPK_QCHK == 1 || (st % ARL_TOPK_QCHK) == 0) flush_rows_with((unsigned)kQFlush);
    };
#endif
    // ---- bootstrap pass state: bl[reg][j] = j-th best score this lane has seen for user row 4g + reg (descending)
    float bl[4][4];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg)
#pragma unroll
        for (int j = 0; j < 4; ++j) bl[reg][j] = -INFINITY;
    auto boot_book = [&](const f32x4 (&ac)[NSUB]) {
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {                    // sorted insert, all four levels from the OLD values: 4 independent ops
                const float sc = ac[sub][reg];
                const float n3 = __builtin_amdgcn_fmed3f(bl[reg][2], bl[reg][3], sc), n2 = __builtin_amdgcn_fmed3f(bl[reg][1], bl[reg][2], sc);
                const float n1 = __builtin_amdgcn_fmed3f(bl[reg][0], bl[reg][1], sc), n0 = fmaxf(bl[reg][0], sc);
                bl[reg][0] = n0; bl[reg][1] = n1; bl[reg][2] = n2; bl[reg][3] = n3;
            }
    };
    // the rows' (k + m)-th best of their 64 sample scores -> thr0v (lane r = row r) and thrf
    auto boot_finish = [&]() {
        int m16 = 0;                                               // lane r < 16: interacted items of row r inside the sample
        if (mrp && item_pos) {                                     // permuted stream: count the row's interacted items whose position is in the sample
            const int lim = NB * MST;
            for (int r = 0; r < 16; ++r) {
                const int u = u_base + r;
                if (u >= U) break;
                int cntr = 0;
                for (int e = mrp[u] + lane, end = mrp[u + 1]; __any(e < end); e += kWave)
                    cntr += __popcll(__ballot(e < end && item_pos[mcol[e]] < lim));
                if (lane == r) m16 = cntr;
            }
        } else if (mrp && lane < 16 && u_base + lane < U) {
            const int b0 = mrp[u_base + lane];
            int lo = b0, hi = mrp[u_base + lane + 1];
            const int lim = NB * MST;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (mcol[mid] < lim) lo = mid + 1; else hi = mid; }
            m16 = lo - b0;
        }
        float t0v = -INFINITY;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int m = __shfl(m16, 4 * g + reg);
            const int nrem = 64 - (k + m);                         // how many of the 64 lie below the (k + m)-th best
            for (int it = 0; it < 64 - k; ++it) {                  // wave-uniform trip count; a group stops removing at its own nrem
                float gm = bl[reg][3];
#pragma unroll
                for (int off = 1; off < 16; off <<= 1) gm = fminf(gm, __shfl_xor(gm, off, 16));
                const unsigned long long eq = __ballot(bl[reg][3] == gm && it < nrem);
                const unsigned mine = (unsigned)(eq >> (16 * g)) & 0xffffu;
                if (mine != 0u && (__ffs(mine) - 1) == c) {        // the group's first lane holding the minimum drops it
                    bl[reg][3] = bl[reg][2]; bl[reg][2] = bl[reg][1]; bl[reg][1] = bl[reg][0]; bl[reg][0] = INFINITY;
                }
            }
            float gm = bl[reg][3];
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) gm = fminf(gm, __shfl_xor(gm, off, 16));
            float t0 = (nrem >= 0 && u_base + 4 * g + reg < U) ? gm - (REFINE ? fmaf(Ereg[reg], stage_norm[nstages], Eabs) : 0.f) : -INFINITY;      // (REFINE: the sample's scores are high-piece scores, each within E of the true one)
            if constexpr (WARM) t0 = fmaxf(t0, __shfl(thr0v, 4 * g + reg));       // both are valid lower bounds: keep the better one
            thrf[reg] = (u_base + 4 * g + reg < U) ? t0 : INFINITY;
            const float bc = __shfl(t0, 16 * (lane >> 2) + 0);     // lane r < 16 reads group r / 4 ...
            if (lane < 16 && (lane & 3) == reg) t0v = bc;          // ... when this is row r's register
        }
        thr0v = t0v;
    };
    ARL_PROF_DECL
    // Scores of one staged tile + their bookkeeping.
    auto compute = [&](auto boot_tag, const unsigned char *buf, int st, unsigned *done_slot) {
        float sn = 0.f;
        if constexpr (REFINE && !decltype(boot_tag)::value) sn = stage_norm[st - NB];      // wave-uniform (a scalar load, long done when the products are)
        f32x4 accs[NSUB];
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) accs[sub] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (F16) {
            // fragments of GRP sub-tiles at a time (all of the stage, or two when the workgroup runs four waves per SIMD on 128 registers)
            constexpr int GRP = (topk_waves(D, SPLIT) >= 16 && NSUB > 2) ? 2 : ((D >= 128 && topk_waves(D, SPLIT) >= 12) ? 1 : NSUB);
            constexpr int TA[3] = {0, 1, 0}, TB[3] = {1, 0, 0};    // ah*bl, al*bh, ah*bh: smallest products first
            if constexpr (REFINE) {                                // high pieces only: the fragments of RG sub-tiles at a time, then their products
                constexpr int RG = ARL_TOPK_RGRP < NSUB ? ARL_TOPK_RGRP : NSUB;
#pragma unroll
                for (int s0 = 0; s0 < NSUB; s0 += RG) {
                    f16x8 pb[RG][KS];
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                        for (int sub = 0; sub < RG; ++sub)
                            pb[sub][ks] = *reinterpret_cast<const f16x8 *>(buf + (g & 1) * HALF + ((s0 + sub) * 16 + c) * RH + ((g >> 1) * PPG + ks) * 16);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                        for (int sub = 0; sub < RG; ++sub)
                            accs[s0 + sub] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[0][ks], pb[sub][ks], accs[s0 + sub], 0, 0, 0);
                    if constexpr (RG < NSUB) __builtin_amdgcn_sched_barrier(0);
                }
            } else if constexpr (ARL_TOPK_EXP == 5) {              // probe: ring + staging only (no fragment reads, no MFMAs)
            } else if constexpr (ARL_TOPK_EXP == 6) {              // probe: MFMAs on whatever the registers hold (fragments read in the first stage only)
                f16x8 pb[2][KS];
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) { pb[pl][ks] = ah[pl][ks]; asm volatile("" : "+v"(pb[pl][ks])); }
#pragma unroll
                for (int sub = 0; sub < NSUB; ++sub)
#pragma unroll
                    for (int term = 0; term < 3; ++term)
#pragma unroll
                        for (int ks = 0; ks < KS; ++ks)
                            accs[sub] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[TA[term]][ks], pb[TB[term]][ks], accs[sub], 0, 0, 0);
            } else if constexpr (ARL_TOPK_EXP == 7) {              // probe: fragment reads only (kept alive through an empty asm), no MFMAs
#pragma unroll
                for (int s0 = 0; s0 < NSUB; s0 += 2) {
                    f16x8 v[2][2][KS];
#pragma unroll
                    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                            for (int ks = 0; ks < KS; ++ks)
                                v[sub][pl][ks] = *reinterpret_cast<const f16x8 *>(buf + (g & 1) * HALF + ((s0 + sub) * 16 + c) * RH + ((pl * 2 + (g >> 1)) * PPG + ks) * 16);
                    __builtin_amdgcn_sched_barrier(0);
                    asm volatile("" ::"v"(v[0][0][0]), "v"(v[0][0][1]), "v"(v[0][1][0]), "v"(v[0][1][1]), "v"(v[1][0][0]), "v"(v[1][0][1]), "v"(v[1][1][0]), "v"(v[1][1][1]));
                }
            } else if constexpr (ARL_TOPK_PIPE && GRP == 2 && NSUB > 2) {
                // Software-pipelined form: the fragments of sub-tile s + 1 are requested BEFORE the six MFMAs of sub-tile s are issued (two
                // 16-register buffers = the registers of the two-sub-tile group form), so a wave's LDS latency runs behind its own matrix
                // work instead of behind the other three waves' of its SIMD only.
                f16x8 pb[2][2][KS];
                auto rd = [&](int sub, int b) {
#pragma unroll
                    for (int plo = 0; plo < 2; ++plo)
#pragma unroll
                        for (int ks = 0; ks < KS; ++ks) {
                            const int pl = 1 - plo;
                            pb[b][pl][ks] = *reinterpret_cast<const f16x8 *>(buf + (g & 1) * HALF + (sub * 16 + c) * RH + ((pl * 2 + (g >> 1)) * PPG + ks) * 16);
                        }
                };
                rd(0, 0);
#pragma unroll
                for (int sub = 0; sub < NSUB; ++sub) {
                    if (sub + 1 < NSUB) rd(sub + 1, (sub + 1) & 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int term = 0; term < 3; ++term)
#pragma unroll
                        for (int ks = 0; ks < KS; ++ks)
                            accs[sub] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[TA[term]][ks], pb[sub & 1][TB[term]][ks], accs[sub], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else
#pragma unroll
            for (int s0 = 0; s0 < NSUB; s0 += GRP) {
                f16x8 bfr[GRP][2][KS];
#pragma unroll
                for (int plo = 0; plo < 2; ++plo)                  // the low pieces first: the first (smallest) term uses them
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                        for (int sub = 0; sub < GRP; ++sub) {
                            const int pl = 1 - plo;
                            bfr[sub][pl][ks] = *reinterpret_cast<const f16x8 *>(buf + (g & 1) * HALF + ((s0 + sub) * 16 + c) * RH + ((pl * 2 + (g >> 1)) * PPG + ks) * 16);
                        }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int term = 0; term < 3; ++term)
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                        for (int sub = 0; sub < GRP; ++sub)
                            accs[s0 + sub] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[TA[term]][ks], bfr[sub][TB[term]][ks], accs[s0 + sub], 0, 0, 0);
                if constexpr (GRP < NSUB) __builtin_amdgcn_sched_barrier(0);
            }
        } else if constexpr (SPLIT) {
            bf16x8 bfr[NSUB][3][KS];
#pragma unroll
            for (int plo = 0; plo < 3; ++plo)                      // planes in the order the terms below first use them: 0, 2, 1
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int sub = 0; sub < NSUB; ++sub) {
                        const int pl = plo == 0 ? 0 : 3 - plo;
                        bfr[sub][pl][ks] = *reinterpret_cast<const bf16x8 *>(buf + (g & 1) * HALF + (sub * 16 + c) * RH + ((pl * 2 + (g >> 1)) * PPG + ks) * 16);
                    }
            __builtin_amdgcn_sched_barrier(0);
            constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};      // (A piece, B piece), smallest products first
#pragma unroll
            for (int term = 0; term < 6; ++term)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int sub = 0; sub < NSUB; ++sub)
                        accs[sub] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[TA[term]][ks], bfr[sub][TB[term]][ks], accs[sub], 0, 0, 0);
        } else {
            // all B fragments of the stage go to registers first (the loads overlap the MFMA chains instead of a read-wait-use
            // sequence per 4 MFMAs), and the sub-tiles' accumulation chains are interleaved so no MFMA waits on its predecessor
            float4 bf[NSUB][Q / 4];
#pragma unroll
            for (int sub = 0; sub < NSUB; ++sub) {
                const float *brow = reinterpret_cast<const float *>(buf + (g & 1) * HALF + (sub * 16 + c) * RH + (g >> 1) * PPG * 16);
#pragma unroll
                for (int t = 0; t < Q; t += 4) bf[sub][t / 4] = *reinterpret_cast<const float4 *>(brow + t);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < Q; t += 4) {
#pragma unroll
                for (int sub = 0; sub < NSUB; ++sub) accs[sub] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], bf[sub][t / 4].x, accs[sub], 0, 0, 0);
#pragma unroll
                for (int sub = 0; sub < NSUB; ++sub) accs[sub] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t + 1], bf[sub][t / 4].y, accs[sub], 0, 0, 0);
#pragma unroll
                for (int sub = 0; sub < NSUB; ++sub) accs[sub] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t + 2], bf[sub][t / 4].z, accs[sub], 0, 0, 0);
#pragma unroll
                for (int sub = 0; sub < NSUB; ++sub) accs[sub] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t + 3], bf[sub][t / 4].w, accs[sub], 0, 0, 0);
            }
        }
        // the tile's fragments are in registers: the slot can be refilled while this wave does its bookkeeping
        asm volatile("" ::: "memory");
        if (done_slot != nullptr && lane == 0) __hip_atomic_fetch_add((lds_u32 *)done_slot, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#ifdef ARL_TOPK_PROF
        { float sink = accs[0][0] + accs[NSUB - 1][3]; asm volatile("" ::"v"(sink)); ARL_PROF_TICK(1) }
#endif
        if constexpr (decltype(boot_tag)::value) boot_book(accs);  // bootstrap stage (its own loop below, so `bl` is dead in the stream's)
        else bookkeeping(accs, st - NB, sn);
        ARL_PROF_TICK(2)
    };
    // Staging: global -> registers (one stage of lead: the loads of stage t + 1 are issued right after stage t went to LDS) -> a RING
    // of kTopkRing tiles in LDS.  No block barrier in the loop: every wave writes ITS share of stage s + kTopkLead into slot
    // (s + kTopkLead) % kTopkRing once all waves have read the stage that occupied it (`done` counter), and consumes stage s once all
    // shares of it are there (`fill` counter).  A wave may therefore run up to two stages ahead of the slowest one: the inserts come in
    // bursts (half of them in the first 2 % of the stream) and with a barrier per stage every burst of one wave stalled the other
    // seven (barrier wait was 18-34 % of the kernel); drifting apart, the waves also stop issuing their 24 KB of fragment reads at the
    // same moment.  Counters only grow (stage t expects (t / ring + 1) * waves); LDS executes a wave's operations in order, so a
    // wave's tile writes are in place before its increment is.
    // Two register sets: the loads of stage t + 2 are in flight while stage t goes to LDS.  (Issuing them through inline assembly with
    // counted `vmcnt` waits was tried: no gain, and unsafe -- the register allocator may split the live range of an asm output around
    // a cold block, e.g. the mask's binary search, and copy the registers before the load has landed.)
    // (with 8-wave workgroups every thread moves the same number of pieces; other workgroup sizes leave the last pass partial)
    f32x4 nb[PER], nc[PER];
#pragma unroll
    for (int p = 0; p < PER; ++p) nb[p] = nc[p] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto fetch = [&](int st, f32x4 (&r)[PER]) {
#pragma unroll
        for (int p = 0; p < PER; ++p)
            if ((p + 1) * kM16Block <= F4 || tid + p * kM16Block < F4) r[p] = *reinterpret_cast<const f32x4 *>(stage_ptr(st, p));
    };
    auto stash = [&](unsigned char *buf, const f32x4 (&r)[PER]) {
#pragma unroll
        for (int p = 0; p < PER; ++p)
            if ((p + 1) * kM16Block <= F4 || tid + p * kM16Block < F4) *reinterpret_cast<f32x4 *>(lds_ptr(buf, p)) = r[p];
    };
    auto slot = [&](int st) { return bt + (st & (kTopkRing - 1)) * STAGEB; };
    auto signal = [&](unsigned *ctr) {
        // the LDS executes one wave's operations in issue order: the counter's atomic lands after the tile writes issued before it, no wait
        // for their completion needed (ARL_TOPK_SIGNAL_WAIT = 1: the round-3 form, which waited for them first -- one exposed LDS round trip per stage)
        if (ARL_TOPK_SIGNAL_WAIT) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        else asm volatile("" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add((lds_u32 *)ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    [[maybe_unused]] auto wait_ge = [&](unsigned *ctr, unsigned want) {             // explicit LDS loads: a flat load would also wait for the global prefetch
        if (ARL_TOPK_EXP == 8) return;                             // probe (scores only, racy): no waits at all -- what the ring's synchronisation costs
        int spins = 0;
        while ((unsigned)__builtin_amdgcn_readfirstlane((int)*(volatile lds_u32 *)ctr) < want) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1 << 26)) __builtin_trap();             // never reached: every wave signals every stage; a trap beats a hang
        }
        asm volatile("" ::: "memory");
    };
    [[maybe_unused]] auto poll_all = [&]() -> unsigned {
        const unsigned v = *(volatile lds_u32 *)(ring_ctr + min(lane, 2 * kTopkRing - 1));
        asm volatile("" ::: "memory");
        return v;
    };
    unsigned *fill_ctr = ring_ctr, *done_ctr = ring_ctr + kTopkRing;
    constexpr unsigned NWV = kM16Block / kWave;
    // ---- Exact early exit (REFINE, with per-stage norms; pays on a norm-ordered stream).  Row r is FINISHED at stage s when no item of a later
    // stage can pass its pre-filter: kExitK * Ereg_r * sufmax[s + 1] + 2 Eabs < threshold_r (Cauchy-Schwarz, see kExitK).  Thresholds only rise and the
    // suffix maxima only fall, so a finished row stays finished.  Every ARL_TOPK_EXIT_EVERY stages a wave whose 16 rows are finished sets ITS word
    // (idempotent) and looks at all of them; a wave that finds every word set, in its step s, lowers `stop` to s + lead + 1 (atomic minimum).  Every wave
    // reads `stop` in the same LDS instruction as the fill counter it waits for before consuming a stage, and leaves the loop at the first stage
    // >= stop without consuming it.  Why that is safe for whichever wave publishes: the fill counter of stage t completes only when every wave has run
    // step t - lead, so (1) no wave has consumed stage s + lead + 1 while the publisher is in step s, and (2) a wave that sees stage s + lead + 1 filled
    // sees a counter the publisher raised in its step s + 1, after its atomic on `stop` (one wave's LDS operations execute in order).  All waves leave at the
    // same stage, the counters of every stage below it complete as before; the stages skipped could not have produced a candidate, so lists, values
    // and tie order are those of the full stream, bit for bit.  (No per-wave state in registers: the kernel has none to spare.)
    constexpr bool EXIT = XIT && REFINE && ARL_TOPK_EXIT && !ARL_TOPK_POLL8 && !ARL_TOPK_PAIRS && (ARL_TOPK_ESCALE >= 1.05f);
    // The poll of the fill counter is the plain wait_ge() with the counter's top bit masked off: the publisher of `stop` sets that bit on all fill
    // counters AFTER storing `stop` and BEFORE its next fill signal, so a wave that sees stage s + lead + 1 (or any later one) filled sees the bit on the
    // counter it just polled and only then -- once per workgroup and kernel -- reads the stop stage itself.
    [[maybe_unused]] auto wait_fill_stop = [&](unsigned *ctr, unsigned want) -> bool {     // true: the top bit is set (a stop stage has been published)
        unsigned v;
        int spins = 0;
        while (v = (unsigned)__builtin_amdgcn_readfirstlane((int)*(volatile lds_u32 *)ctr), (v & 0x7fffffffu) < want) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1 << 26)) __builtin_trap();
        }
        asm volatile("" ::: "memory");
        return (v >> 31) != 0u;
    };
    [[maybe_unused]] auto vote = [&](int st) {                         // this wave's rows are finished after stage st (virtual index)
        unsigned *fw = ring_ctr + 2 * kTopkRing + 4;
        int lv = lane;
        asm volatile("" : "+v"(lv));                                   // (opaque copy: the per-lane LDS addresses below are formed HERE, not hoisted into registers held over the whole stream)
        if (lv == 0) *(volatile lds_u32 *)(fw + wv) = 1u;
        const unsigned seen = lv < (int)NWV ? *(volatile lds_u32 *)(fw + lv) : 1u;      // (after this wave's own store: LDS operations of a wave execute in order)
        if (__builtin_amdgcn_ballot_w64(seen == 0u) == 0ull) {
            if (lv == 0) __hip_atomic_fetch_min((lds_u32 *)(ring_ctr + 2 * kTopkRing), (unsigned)(st + kTopkLead + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (lv < kTopkRing) __hip_atomic_fetch_or((lds_u32 *)(ring_ctr + lv), 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // (issued after the minimum: in order)
        }
    };
    __syncthreads();                                               // counters zeroed (the only block barrier of the kernel)
    // one step of the pipeline: stage st + 2 goes from register set r to LDS, set r is refilled with stage st + 4, stage st is consumed
    [[maybe_unused]] auto step = [&](auto boot_tag, int st, f32x4 (&r)[PER]) -> bool {
        const int t = st + kTopkLead;

#if ARL_TOPK_POLL8
        // one LDS round trip reads all 2 * ring counters (lane l < 2 * ring holds counter l); in the steady state both conditions of the step
        // (the slot to refill has been read by every wave; the stage to consume is complete) hold in that one snapshot
        unsigned snap = poll_all();
        if (t < nvirt) {
            for (int spins = 0; (unsigned)__builtin_amdgcn_readlane((int)snap, kTopkRing + (t & (kTopkRing - 1))) < NWV * (unsigned)(t / kTopkRing); snap = poll_all()) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1 << 26)) __builtin_trap();
            }
#else
        if (t < nvirt) {
            wait_ge(done_ctr + (t & (kTopkRing - 1)), NWV * (unsigned)(t / kTopkRing));          // stage t - ring has been read by every wave
#endif
#ifdef ARL_TOPK_PROF
            ARL_PROF_TICK(4)
            asm volatile("s_waitcnt vmcnt(1)" ::: "memory");          // (profiling build: the older of the two loads in flight, timed apart)
            ARL_PROF_TICK(5)
#endif
            stash(slot(t), r);
            signal(fill_ctr + (t & (kTopkRing - 1)));
#ifdef ARL_TOPK_PROF
            ARL_PROF_TICK(6)
#endif
            if (t + 2 < nvirt) fetch(item_stage(t + 2), r);
        }
        ARL_PROF_TICK(0)
#if ARL_TOPK_POLL8
        for (int spins = 0; (unsigned)__builtin_amdgcn_readlane((int)snap, st & (kTopkRing - 1)) < NWV * (unsigned)(st / kTopkRing + 1); snap = poll_all()) {
            if (spins) __builtin_amdgcn_s_sleep(1);
            if (++spins > (1 << 26)) __builtin_trap();
        }
        asm volatile("" ::: "memory");
#else
        if constexpr (EXIT && !decltype(boot_tag)::value) {
            if (wait_fill_stop(fill_ctr + (st & (kTopkRing - 1)), NWV * (unsigned)(st / kTopkRing + 1))) {
                const int stop = __builtin_amdgcn_readfirstlane((int)*(volatile lds_u32 *)(ring_ctr + 2 * kTopkRing));
                if (st >= stop) return true;                           // wave-uniform; every wave of the workgroup leaves at this stage
            }
        } else wait_ge(fill_ctr + (st & (kTopkRing - 1)), NWV * (unsigned)(st / kTopkRing + 1));
#endif
        ARL_PROF_TICK(3)
        compute(boot_tag, slot(st), st, done_ctr + (st & (kTopkRing - 1)));
        if constexpr (EXIT && !decltype(boot_tag)::value) {
            // are this wave's 16 rows finished -- can no item of a LATER stage pass their pre-filter?  Tested every ARL_TOPK_EXIT_EVERY-th stage, here, after
            // the stage's accumulators have died (the kernel runs at 128 VGPRs with none to spare; the constant rides on the wave-uniform factor so that
            // the compiler has no per-lane product to hoist into a register held over the whole stream)
            if ((st & (ARL_TOPK_EXIT_EVERY - 1)) == ARL_TOPK_EXIT_EVERY - 1) {
                const float kx = kExitK * stage_norm[nstages + 1 + (st - NB) + 1];      // suffix maximum: the largest scaled norm of any later stage (scalar load)
                bool f = true;
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) f = f && (fmaf(Ereg[reg], kx, 2.f * Eabs) < thrf[reg]);       // (+inf for rows past U; NaN never finishes)
                if (__builtin_amdgcn_ballot_w64(!f) == 0ull) vote(st);
            }
        }
        return false;
    };
#if ARL_TOPK_PAIRS
    // Stages are synchronised in PAIRS (slots {0,1} and {2,3} of the ring; one fill and one done counter per pair): half the counter polls, tile-write
    // waits and signals per item of the per-stage form -- with REFINE a stage is eight MFMAs per wave and those round trips were most of the loop.
    static_assert(kTopkLead == 2 && kTopkRing == 4, "pairs: two stages of lead in a ring of four");
    auto pair_step = [&](auto boot_tag, int st) {                  // st even: stages st + 2, st + 3 go to LDS, st + 4, st + 5 are requested, st and st + 1 are consumed
        const int t = st + 2;
        if (t < nvirt) {
            const int ps = (t >> 1) & 1;
            wait_ge(done_ctr + ps, NWV * (unsigned)(t / 4));        // the pair that held these two slots has been read by every wave
            ARL_PROF_TICK(4)
            stash(slot(t), nb);
            if (t + 1 < nvirt) stash(slot(t + 1), nc);
            signal(fill_ctr + ps);
            ARL_PROF_TICK(6)
            if (t + 2 < nvirt) fetch(item_stage(t + 2), nb);
            if (t + 3 < nvirt) fetch(item_stage(t + 3), nc);
        }
        ARL_PROF_TICK(0)
        const int pc = (st >> 1) & 1;
        wait_ge(fill_ctr + pc, NWV * (unsigned)(st / 4 + 1));
        ARL_PROF_TICK(3)
        const int last = min(st + 1, nvirt - 1);
#pragma nounroll
        for (int sx = st; sx <= last; ++sx)                        // (a loop, not two calls: one copy of the stage body in the instruction cache)
            compute(boot_tag, slot(sx), sx, sx == last ? done_ctr + pc : (unsigned *)nullptr);
    };
    if (0 < nvirt) fetch(item_stage(0), nb);
    if (1 < nvirt) fetch(item_stage(1), nc);
    if (0 < nvirt) stash(slot(0), nb);
    if (1 < nvirt) stash(slot(1), nc);
    signal(fill_ctr + 0);
    if (2 < nvirt) fetch(item_stage(2), nb);
    if (3 < nvirt) fetch(item_stage(3), nc);
    for (int st = 0; st < NB; st += 2) pair_step(std::true_type{}, st);      // NB is even
    if (NB > 0) boot_finish();
    for (int st = NB; st < nvirt; st += 2) pair_step(std::false_type{}, st);
#else
    // prologue: the first kTopkLead stages go to their slots, the next two are in flight in the two register sets
    for (int s0 = 0; s0 < kTopkLead; s0 += 2) {
        if (s0 < nvirt) fetch(item_stage(s0), nb);
        if (s0 + 1 < nvirt) fetch(item_stage(s0 + 1), nc);
        if (s0 < nvirt) { stash(slot(s0), nb); signal(fill_ctr + (s0 & (kTopkRing - 1))); }
        if (s0 + 1 < nvirt) { stash(slot(s0 + 1), nc); signal(fill_ctr + ((s0 + 1) & (kTopkRing - 1))); }
    }
    if (nvirt > kTopkLead) fetch(item_stage(kTopkLead), nb);
    if (nvirt > kTopkLead + 1) fetch(item_stage(kTopkLead + 1), nc);
    for (int st = 0; st < NB; st += 2) {                           // NB is even
        step(std::true_type{}, st, nb);
        step(std::true_type{}, st + 1, nc);
    }
    if (NB > 0) boot_finish();
    int st_end = NB;
    for (; st_end < nvirt; st_end += 2) {
        if (step(std::false_type{}, st_end, nb)) break;
        if (st_end + 1 < nvirt && step(std::false_type{}, st_end + 1, nc)) { st_end += 1; break; }
    }
    if (stats != nullptr && tid == 0) {                            // stream stages this workgroup consumed (all of them without an exit), and one workgroup
        atomicAdd(stats, (unsigned long long)(min(st_end, nvirt) - NB));
        atomicAdd(stats + 1, 1ull);
    }
#endif
#if ARL_TOPK_QUEUE
    flush_rows_with(1u);                                           // whatever is still queued
    if (ARL_TOPK_EXP && exp_sink == 12345.678f) top_val[0] = exp_sink;
#endif
#ifdef ARL_TOPK_PROF
    const long long P_loop = clock64() - P_start;
#endif
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int u = u_base + r;
        if (u < U && lane < k) {
            const unsigned long long key = ((unsigned long long)tk_hi[r] << 32) | tk_lo[r];
            top_idx[(size_t)u * k + lane] = cand_item(key);
            top_val[(size_t)u * k + lane] = cand_score(key) * score_unscale;
            // the warm threshold excluded too much: the list ends short -- or with an interacted item's -10e8, which only belongs there when fewer than k
            // other items exist (a warm candidate that has become interacted can set a threshold above every free item; the repeat is then cold)
            if (WARM && lane == k - 1 && (key == 0ull || cand_score(key) <= -5e8f * score_scale)) atomicOr(underflow, 1);
        }
    }
#ifdef ARL_TOPK_PROF
    if (lane == 0 && u_base < U) {
        float *o = top_val + (size_t)u_base * k;
        o[0] = (float)P_acc[0]; o[1] = (float)P_acc[1]; o[2] = (float)P_acc[2]; o[3] = (float)P_acc[3]; o[4] = (float)P_loop; o[5] = (float)(clock64() - P_start);
        o[6] = (float)P_acc[4]; o[7] = (float)P_acc[5]; o[8] = (float)P_acc[6];
    }
#endif
}

// ================================================================================================
// score_mask_topk, second form of the fp16 split path's stream (round 4; d = 64 / 128, k <= 64, needs the caller's USER workspace).
// The first form (above) spends its time on issue slots, not on products: 104 vector + 130 scalar instructions per wave and 64-item stage of
// 16 users, a counter ring polled five times per stage, MFMA busy 15 %.  This form turns the tile around:
//   * a wave owns 32 users and contracts with v_mfma_f32_32x32x16_f16, ITEMS as rows and USERS as columns: all 16 accumulators of a lane belong
//     to ONE user (column lane % 32), so the pre-filter is one subtract + one v_alignbit per score against one threshold register, collected
//     into a per-lane bit mask -- no per-score scalar mask, no per-(sub-tile, row) branch;
//   * 16 waves = 512 users share a staged tile (half the staging traffic per score of the first form's 256), stages are 128 items (d = 64)
//     behind ONE workgroup barrier each (double-buffered tiles) instead of the ring's counters;
//   * the sorted lists live IN THE OUTPUT ARRAYS (top_val = mapped score bits, top_idx = ~item while the kernel runs; L2-resident) and not in
//     registers: a merge loads / stores 8 B per lane, and the kernel fits 128 registers with four waves per SIMD;
//   * starting thresholds come from two small kernels of their own (bootstrap sample / warm-start candidates) through the user workspace.
// What reaches the lists is unchanged: a queued candidate gets the first form's exact three-product score (the same v_mfma_f32_16x16x32_f16
// sequence on the same pieces, user row against candidate columns) when its row's queue is merged, keys and tie order are the same, so the
// results are bit-identical to the first form's (and the bound behind the pre-filter is the same E: any fp32-accumulated contraction of the
// high pieces lies within it).
// ================================================================================================
typedef float f32x16 __attribute__((ext_vector_type(16)));
#ifndef ARL_TOPK2
#define ARL_TOPK2 1                                // 0: the first form everywhere (A/B builds)
#endif
#ifndef ARL_TOPK2_EXP
#define ARL_TOPK2_EXP 0                            // developer probes (results wrong by construction)
#endif
#ifndef ARL_TOPK2_UA_LATE
#define ARL_TOPK2_UA_LATE 1                        // the user's pieces of a merge are loaded when it completes (1) or carried with the pending state (0: 16 registers more)
#endif
#ifndef ARL_TOPK2_PREFETCH
#define ARL_TOPK2_PREFETCH 0
#endif
#ifndef ARL_TOPK2_D128
#define ARL_TOPK2_D128 0
#endif
constexpr int kT2Waves = 16;                       // waves per workgroup
constexpr int kT2Users = 32 * kT2Waves;            // users per workgroup
#ifndef ARL_TOPK2_RING
#define ARL_TOPK2_RING 3
#endif
constexpr int kT2Ring = ARL_TOPK2_RING;                         // staged item tiles in LDS
constexpr int kT2QCap = 16;                        // queue slots per user (the candidates of one merge are the columns of one 16 x 16 tile)
#ifndef ARL_TOPK2_QFLUSH
#define ARL_TOPK2_QFLUSH 12
#endif
constexpr int kT2QFlush = ARL_TOPK2_QFLUSH;        // a row's queue is merged at the end of a stage once it holds this many
#ifndef ARL_TOPK2_BOOT_ITEMS
#define ARL_TOPK2_BOOT_ITEMS 16384
#endif
constexpr int kT2BootItems = ARL_TOPK2_BOOT_ITEMS; // sample of the bootstrap phase (streams of >= 32 768 items; a multiple of 128)
__host__ __device__ constexpr int t2_mst(int D) { return D <= 64 ? 128 : 64; }       // items per stage
__host__ __device__ constexpr int t2_rs(int D) { return D * 2 + 16; }                // LDS bytes per staged item row: the high pieces + 16 (conflict-free ds_read_b128 over 16 rows)
__host__ __device__ constexpr size_t t2_lds_bytes(int D, bool masked) {
    return kT2Ring * (size_t)t2_mst(D) * t2_rs(D) + 128 + sizeof(unsigned) * kT2Users * (1 + kT2QCap) + (masked ? sizeof(unsigned) * kT2Users * kBloomWords : 0);
}

// starting thresholds from the warm-start candidates (first form: the WARM prologue): thr0[u] = the lowest of the k candidates' fp32 scores, lowered by
// the bound on the difference to the streamed contraction, in the scaled domain
__global__ __launch_bounds__(kBlock) void topk2_warm_kernel(const float *__restrict__ Pu, const float *__restrict__ Pi_f32, int U, int I, int D, int k,
                                                            const int32_t *__restrict__ warm_idx, const unsigned *__restrict__ table_max_bits,
                                                            float *__restrict__ thr0, const int *__restrict__ gate, const int *__restrict__ gate2) {
    if (gate != nullptr && *gate == 0) return;
    if (gate2 != nullptr && *gate2 == 0) return;
    // D / 4 lanes x float4 own one candidate row (d = 64: 16 lanes, four candidates per pass): a load instruction touches whole 256-byte rows
    const int lane = threadIdx.x & 63, LPR = D >> 2, RPP = 64 / LPR, sub = lane / LPR, ch = lane % LPR;
    const float score_scale = split_scale(table_max_bits[1]) * split_scale(table_max_bits[0]);
    for (int u = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6); u < U; u += gridDim.x * kWavesPerBlock) {
        const float4 x = *reinterpret_cast<const float4 *>(Pu + (size_t)u * D + 4 * ch);
        float nu = fmaf(x.x, x.x, fmaf(x.y, x.y, fmaf(x.z, x.z, x.w * x.w)));
        for (int off = 1; off < LPR; off <<= 1) nu += __shfl_xor(nu, off);
        float lb = INFINITY;
        for (int j0 = 0; j0 < k; j0 += RPP) {
            const int j = j0 + sub;
            const int cand = j < k ? min(max(warm_idx[(size_t)u * k + j], 0), I - 1) : 0;
            const float4 y = *reinterpret_cast<const float4 *>(Pi_f32 + (size_t)cand * D + 4 * ch);
            float sd = fmaf(x.x, y.x, fmaf(x.y, y.y, fmaf(x.z, y.z, x.w * y.w)));
            float ni = fmaf(y.x, y.x, fmaf(y.y, y.y, fmaf(y.z, y.z, y.w * y.w)));
            for (int off = 1; off < LPR; off <<= 1) { sd += __shfl_xor(sd, off); ni += __shfl_xor(ni, off); }
            const float v = (sd - 8e-6f * sqrtf(nu * ni)) * score_scale - 1e-30f;       // (the slack covers any fp32 summation order of the two contractions)
            if (j < k) lb = fminf(lb, v);
        }
        for (int off = LPR; off < 64; off <<= 1) lb = fminf(lb, __shfl_xor(lb, off));
        if (lane == 0) thr0[u] = lb;
    }
}

template <int D, bool XIT>
__global__ __launch_bounds__(64 * kT2Waves) void topk2_main_kernel(const _Float16 *__restrict__ uimg, const _Float16 *__restrict__ img, int U, int I,
                                                                  const int32_t *__restrict__ mrp, const int32_t *__restrict__ mcol, int k,
                                                                  int32_t *__restrict__ top_idx, float *__restrict__ top_val, const float *__restrict__ thr0,
                                                                  int *__restrict__ underflow, int warm, const unsigned *__restrict__ table_max_bits,
                                                                  const int32_t *__restrict__ item_order, const int *__restrict__ gate,
                                                                  const float *__restrict__ stage_norm, unsigned long long *__restrict__ stats,
                                                                  const int *__restrict__ gate2, int boot, const int32_t *__restrict__ item_pos) {
    if (gate != nullptr && *gate == 0) return;
    if (gate2 != nullptr && *gate2 == 0) return;
    constexpr int MST = t2_mst(D), RS = t2_rs(D), TB = MST * RS, KS = D / 16, KS16 = D / 32, NT = 64 * kT2Waves;
    constexpr int C16 = D * 2 / 16;                                // 16-byte pieces of an item row's high plane
    constexpr int PER = MST * C16 / NT;
    static_assert(MST * C16 % NT == 0 && PER >= 1, "every thread moves the same number of pieces");
    extern __shared__ unsigned char smem_raw[];
    unsigned *ring_ctr = reinterpret_cast<unsigned *>(smem_raw + kT2Ring * TB);      // fill[kT2Ring], done[kT2Ring] (the main stream's ring, below)
    unsigned *qcnt_all = ring_ctr + 32;                            // ([6] the early exit's stop stage, [8 .. 8 + waves) one "finished" word per wave)
    unsigned *qpos_all = qcnt_all + kT2Users;
    unsigned *bloom_all = qpos_all + kT2Users * kT2QCap;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = lane & 31, h = lane >> 5, c = lane & 15, g = lane >> 4;
    const int u_base = blockIdx.x * kT2Users + wv * 32;
    unsigned *qcnt = qcnt_all + wv * 32, *qpos = qpos_all + wv * 32 * kT2QCap, *bloom = bloom_all + wv * 32 * kBloomWords;
    const int u = u_base + n, uc = min(u, U - 1);
    const float su = split_scale(table_max_bits[1]), si = split_scale(table_max_bits[0]);
    const float score_scale = su * si, score_unscale = (1.f / su) * (1.f / si);
    const float Eabs = 1.1920929e-7f * (float)D * (__uint_as_float(table_max_bits[0]) * si + __uint_as_float(table_max_bits[1]) * su);
    f16x8 bu[KS];
    float Ereg;
    {
        float n2 = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bu[ks] = *reinterpret_cast<const f16x8 *>(uimg + (size_t)uc * 2 * D + 16 * ks + 8 * h);
            const f16x8 lo = *reinterpret_cast<const f16x8 *>(uimg + (size_t)uc * 2 * D + D + 16 * ks + 8 * h);
#pragma unroll
            for (int t = 0; t < 8; ++t) { const float x = (float)bu[ks][t] + (float)lo[t]; n2 = fmaf(x, x, n2); }
        }
        n2 += __shfl_xor(n2, 32);
        Ereg = ARL_TOPK_ESCALE * 0.0009765625f * sqrtf(n2) * 1.0000005f;
    }
    float thr0v = u < U ? (thr0 != nullptr ? thr0[u] : -INFINITY) : INFINITY;      // lanes n and n + 32: user n's starting threshold (warm start; the bootstrap below raises it)
    float thr = thr0v;                                             // running exact threshold of this lane's user (users past U never append)
    // lists: empty
    for (int r = 0; r < 32; ++r) {
        const int ur = u_base + r;
        if (ur < U && lane < k) { top_idx[(size_t)ur * k + lane] = 0; reinterpret_cast<unsigned *>(top_val)[(size_t)ur * k + lane] = 0u; }
    }
    if (lane < 32) qcnt[lane] = 0u;
    if (mrp) {
        for (int t = lane; t < 32 * kBloomWords; t += kWave) bloom[t] = 0u;
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
        for (int r = 0; r < 32; ++r) {
            const int ur = u_base + r;
            if (ur >= U) break;
            for (int e = mrp[ur] + lane, end = mrp[ur + 1]; e < end; e += kWave) {
                const unsigned hb = bloom_hash(mcol[e]);
                atomicOr(&bloom[r * kBloomWords + (hb >> 5)], 1u << (hb & 31u));
            }
        }
    }
    __threadfence();                                               // (the zeroed lists are read back by this wave only; L2 is the point of coherence)
    const int nst = (I + MST - 1) / MST;

    // Merging the queue of user row r (wave-uniform) into its list, in two halves so that the global round trips (the candidates' rows of the image, the
    // user's pieces, the list, the item ids) run behind a stage of matrix work: flush_issue() snapshots the queue into registers, frees it and issues
    // the loads; flush_complete() -- one stage later for the stage-end merges, at once for a full queue -- scores, packs, merges and stores.  One merge
    // may be pending per wave; a row whose queue fills while its own merge is pending is completed first (flush_row), so no list update is lost.
#ifdef ARL_TOPK2_PROF
    long long Q_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, Q_t0 = clock64();
#define T2_TICK(SLOT) { const long long Q_t = clock64(); Q_acc[SLOT] += Q_t - Q_t0; Q_t0 = Q_t; }
#define T2_COUNT(SLOT, N) { Q_acc[SLOT] += (N); }
#else
#define T2_TICK(SLOT)
#define T2_COUNT(SLOT, N)
#endif
    int pend_r = -1, pend_m = 0;                                   // wave-uniform
    unsigned p_pos = 0u, p_Kh = 0u, p_Kl = 0u;
    int p_item = 0;
    f16x8 p_fb[2][KS16];
#if !ARL_TOPK2_UA_LATE
    f16x8 p_ua[2][KS16];
#endif
    auto flush_issue = [&](int r) {
#if ARL_TOPK2_EXP == 2
        if (lane == 0) *(volatile lds_u32 *)(qcnt + r) = 0u;          // probe: appends only, queues dropped
        return;
#endif
        const int cnt = __builtin_amdgcn_readfirstlane((int)*(volatile lds_u32 *)(qcnt + r));
        const int m = min(cnt, kT2QCap);
        const int ur = u_base + r;
        p_pos = 0u;
        if (lane < m) p_pos = *(volatile lds_u32 *)(qpos + r * kT2QCap + lane);
        p_Kh = 0u; p_Kl = 0u;
        if (lane < k) { p_Kh = reinterpret_cast<const unsigned *>(top_val)[(size_t)ur * k + lane]; p_Kl = (unsigned)top_idx[(size_t)ur * k + lane]; }
        const int pos_c = __shfl((int)p_pos, c);                   // 0 for columns without a candidate: row 0 of the image, never read back
        const _Float16 *src = img + (size_t)pos_c * 2 * D + g * (KS16 * 8);
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int ks = 0; ks < KS16; ++ks) p_fb[pl][ks] = *reinterpret_cast<const f16x8 *>(src + pl * D + ks * 8);
#if !ARL_TOPK2_UA_LATE
        {
            const _Float16 *usr = uimg + (size_t)ur * 2 * D + g * (KS16 * 8);
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                for (int ks = 0; ks < KS16; ++ks) p_ua[pl][ks] = *reinterpret_cast<const f16x8 *>(usr + pl * D + ks * 8);
        }
#endif
        p_item = 0;
        if (lane < m) p_item = item_order ? item_order[p_pos] : (int)p_pos;
        if (lane == 0) *(volatile lds_u32 *)(qcnt + r) = 0u;
        pend_r = r; pend_m = m;
    };
    auto flush_complete = [&]() {
        if (pend_r < 0) return;
        const int r = pend_r, m = pend_m, ur = u_base + r;
        pend_r = -1;
        // the user's own pieces (every row of the 16 x 16 tile is this user; an L2 hit: the wave read these rows at its start): loaded here, not
        // carried across the stage -- 16 registers the tile loop needs more
#if ARL_TOPK2_UA_LATE
        f16x8 p_ua[2][KS16];
        {
            const _Float16 *usr = uimg + (size_t)ur * 2 * D + g * (KS16 * 8);
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                for (int ks = 0; ks < KS16; ++ks) p_ua[pl][ks] = *reinterpret_cast<const f16x8 *>(usr + pl * D + ks * 8);
        }
#endif
        constexpr int TA[3] = {0, 1, 0}, TB3[3] = {1, 0, 0};       // the first form's order: ah*bl, al*bh, ah*bh
        f32x4 ca = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int term = 0; term < 3; ++term)
#pragma unroll
            for (int ks = 0; ks < KS16; ++ks) ca = __builtin_amdgcn_mfma_f32_16x16x32_f16(p_ua[TA[term]][ks], p_fb[TB3[term]][ks], ca, 0, 0, 0);
        // every row of the tile is user ur: lane j < 16 (k-group 0, register 0 = row 0) holds candidate j's score
        unsigned long long cj = 0ull;
        if (lane < m) {
            float sc = ca[0];
            const int item = p_item;
            if (mrp) {
                const unsigned hb = bloom_hash(item);
                if ((bloom[r * kBloomWords + (hb >> 5)] >> (hb & 31u)) & 1u) {
                    int lo = mrp[ur], hi = mrp[ur + 1];
                    const int end = hi;
                    while (lo < hi) { const int mid = (lo + hi) >> 1; if (mcol[mid] < item) lo = mid + 1; else hi = mid; }
                    if (lo < end && mcol[lo] == item) sc = -10e8f * score_scale;
                }
            }
            cj = pack_cand(sc, item);
        }
        unsigned Kh = p_Kh, Kl = p_Kl;
        const unsigned long long Kk = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)Kh, k - 1) << 32) | (unsigned)__builtin_amdgcn_readlane((int)Kl, k - 1);
        bool changed = false;
        for (unsigned long long todo = __ballot(lane < m && cj > Kk); todo != 0ull; todo &= todo - 1ull) {
            const int j = __ffsll((long long)todo) - 1;
            const unsigned ch = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(cj >> 32), j);
            const unsigned cl = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)cj, j);
            const unsigned long long key = ((unsigned long long)ch << 32) | cl, K = ((unsigned long long)Kh << 32) | Kl;
            const int pos = __popcll(__ballot(K > key));
            if (pos < k) {
                const unsigned sl = __builtin_amdgcn_update_dpp(Kl, Kl, 0x138, 0xf, 0xf, false);            // wave_shr:1
                const unsigned sh = __builtin_amdgcn_update_dpp(Kh, Kh, 0x138, 0xf, 0xf, false);
                Kl = lane < pos ? Kl : (lane == pos ? cl : sl);
                Kh = lane < pos ? Kh : (lane == pos ? ch : sh);
                changed = true;
            }
        }
        if (changed) {
            if (lane < k) { reinterpret_cast<unsigned *>(top_val)[(size_t)ur * k + lane] = Kh; top_idx[(size_t)ur * k + lane] = (int)Kl; }
            const unsigned th = (unsigned)__builtin_amdgcn_readlane((int)Kh, k - 1);
            const unsigned tl = (unsigned)__builtin_amdgcn_readlane((int)Kl, k - 1);
            const float t0 = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(thr0v), r));
            const float nt = (th | tl) ? __builtin_fmaxf(cand_score((unsigned long long)th << 32), t0) : t0;
            thr = (n == r) ? nt : thr;
        }
    };
    auto flush_row = [&](int r) { T2_TICK(1) flush_complete(); flush_issue(r); flush_complete(); T2_TICK(5) T2_COUNT(6, 1) };       // at once (a full queue; the end of the stream)
    auto flush_rows_with = [&](unsigned at_least) {
        const unsigned cn = lane < 32 ? *(volatile lds_u32 *)(qcnt + lane) : 0u;
        for (unsigned long long need = __ballot(cn >= at_least); need != 0ull; need &= need - 1ull) flush_row(__ffsll((long long)need) - 1);
    };
    // a slot of the pipelined merges (two per stage: after half of the tiles and at the end): the pending merge is completed, the next one -- a row whose
    // queue has reached kT2QFlush -- is issued
    auto flush_slot = [&]() {
        flush_complete();
        const unsigned cn = lane < 32 ? *(volatile lds_u32 *)(qcnt + lane) : 0u;
        const unsigned long long need = __ballot(cn >= (unsigned)kT2QFlush);
        if (need != 0ull) { flush_issue(__ffsll((long long)need) - 1); T2_COUNT(7, 1) }
    };

    // staging: global -> registers -> LDS, one stage ahead in registers, one in LDS
    float4 stg[PER];
    auto gload = [&](int s) {
#pragma unroll
        for (int p = 0; p < PER; ++p) {
            const int f = tid + p * NT;
            const int item = min(s * MST + f / C16, I - 1);        // clamped: rows past I are copies of item I - 1, masked out of the bit masks
            stg[p] = *reinterpret_cast<const float4 *>(reinterpret_cast<const unsigned char *>(img) + (size_t)item * (4 * D) + (f % C16) * 16);
        }
    };
    auto lwrite = [&](unsigned char *buf) {
#pragma unroll
        for (int p = 0; p < PER; ++p) {
            const int f = tid + p * NT;
            *reinterpret_cast<float4 *>(buf + (f / C16) * RS + (f % C16) * 16) = stg[p];
        }
    };
    // Bootstrap (streams of >= 32 768 items): the first kT2BootItems items are scored once without lists.  A lane keeps the maxima of 32 disjoint item
    // groups (its accumulator slots of a stage, over the sample's stages; the two half-waves see different rows): 64 group maxima of DISTINCT items per
    // user.  The (k + m)-th largest -- m = the user's interacted items inside the sample, which the mask may take out -- minus E is a lower bound of the
    // user's final k-th best score; selected per lane pair by a 32-step bisection over the order-preserving integer image of the floats.
    if (boot) {
        constexpr int NTL = MST / 32, GPT = 32 / NTL;              // tiles per stage, groups per tile and lane
        float gm[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) gm[j] = -INFINITY;
        const int nbs = boot / MST;                                // boot = the sample's size in items (a multiple of 128)
        gload(0);
        lwrite(smem_raw);
        gload(1);
        __syncthreads();
        for (int s = 0; s < nbs; ++s) {
            const unsigned char *buf = smem_raw + (s & 1) * TB;
            if (s + 1 < nbs) lwrite(smem_raw + ((s + 1) & 1) * TB);
            if (s + 2 < nbs) gload(s + 2);
#pragma unroll
            for (int t = 0; t < NTL; ++t) {
                f16x8 af[KS];
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) af[ks] = *reinterpret_cast<const f16x8 *>(buf + (t * 32 + n) * RS + ks * 32 + h * 16);
                f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks], bu[ks], acc, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < GPT; ++q) {
                    if constexpr (GPT == 8) gm[t * GPT + q] = fmaxf(gm[t * GPT + q], fmaxf(acc[2 * q], acc[2 * q + 1]));
                    else gm[t * GPT + q] = fmaxf(gm[t * GPT + q], acc[q]);
                }
            }
            __syncthreads();
        }
        int msk = 0;                                               // this user's interacted items inside the sample (both lanes of the pair compute it)
        if (mrp != nullptr && u < U) {
            const int b0 = mrp[u], e0 = mrp[u + 1];
            if (item_pos != nullptr) {
                for (int e = b0; e < e0; ++e) msk += item_pos[mcol[e]] < boot;
            } else {
                int lo = b0, hi = e0;
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (mcol[mid] < boot) lo = mid + 1; else hi = mid; }
                msk = lo - b0;
            }
        }
        const int K = k + msk;
        unsigned T = 0u;
        for (int bit = 31; bit >= 0; --bit) {
            const unsigned cand = T | (1u << bit);
            int cnt = 0;
#pragma unroll
            for (int j = 0; j < 32; ++j) { const unsigned b = __float_as_uint(gm[j]); cnt += ((b & 0x80000000u) ? ~b : (b | 0x80000000u)) >= cand; }
            cnt += __shfl_xor(cnt, 32);
            if (cnt >= K) T = cand;
        }
        float bound = -INFINITY;
        if (K <= 64 && T != 0u) {
            const unsigned b = (T & 0x80000000u) ? (T & 0x7fffffffu) : ~T;
            bound = __uint_as_float(b) - fmaf(Ereg, stage_norm[(I + (D <= 64 ? 64 : 32) - 1) / (D <= 64 ? 64 : 32)], Eabs);
            if (!(bound == bound)) bound = -INFINITY;
        }
        if (u < U) { thr0v = fmaxf(thr0v, bound); thr = thr0v; }
    }
    // The stream: a RING of kT2Ring staged tiles, no workgroup barrier in the loop.  Every wave writes ITS share of stage s + 1 into slot (s + 1) % ring once all
    // waves have read the stage that occupied it (`done` counter), and consumes stage s once all shares of it are there (`fill` counter); counters only
    // grow (stage t expects (t / ring + 1) * waves).  A wave may run up to ring - 1 stages ahead of the slowest: the merges come at random moments and take
    // thousands of cycles -- behind a barrier per stage every one of them stalled the other fifteen waves (a third of the kernel was barrier wait).
    // LDS executes one wave's operations in order, so a wave's tile writes are in place before its `fill` increment, its fragment reads before `done`'s.
    unsigned *fill_ctr = ring_ctr, *done_ctr = ring_ctr + kT2Ring;
    if (tid < 32) ring_ctr[tid] = tid == 6 ? 0x7fffffffu : 0u;
    __syncthreads();                                               // (also: the bootstrap's last tile has been read by every wave)
    // XIT: the exact early exit of the norm-ordered stream, as in the first form (bound: kExitK above).  Row r is FINISHED at stage s when no item of a later
    // stage can pass its pre-filter: kExitK * Ereg_r * sufmax(next stage) + 2 Eabs < threshold_r; thresholds only rise, suffix maxima only fall.  Every 8th
    // stage a wave whose 32 rows are finished sets its word and looks at all 16; a wave that finds them all set, at the END of its step s, lowers `stop` to
    // s + 2 (atomic minimum) and then sets the top bit of the fill counters.  Stage s + 2 is filled only once every wave has run the head of step s + 1 --
    // the publisher after its atomics (one wave's LDS operations execute in order) -- so nobody has consumed it yet, and whoever polls its fill counter sees
    // the bit, reads `stop` and leaves the loop there: all waves consume exactly the stages below `stop`, the shares of every such stage are written by
    // every wave as before, and the stages skipped could not have produced a candidate -- lists, values and tie order are the full stream's.
    [[maybe_unused]] auto ring_wait_flag = [&](unsigned *ctr, unsigned want) -> bool {
        unsigned v;
        int spins = 0;
        while (v = (unsigned)__builtin_amdgcn_readfirstlane((int)*(volatile lds_u32 *)ctr), (v & 0x7fffffffu) < want) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1 << 26)) __builtin_trap();
        }
        asm volatile("" ::: "memory");
        return (v >> 31) != 0u;
    };
    auto ring_wait = [&](unsigned *ctr, unsigned want) {
        int spins = 0;
        while ((unsigned)__builtin_amdgcn_readfirstlane((int)*(volatile lds_u32 *)ctr) < want) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1 << 26)) __builtin_trap();             // never reached: every wave signals every stage; a trap beats a hang
        }
        asm volatile("" ::: "memory");
    };
    auto ring_signal = [&](unsigned *ctr) {
        asm volatile("" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add((lds_u32 *)ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    gload(0);
    lwrite(smem_raw);
    ring_signal(fill_ctr + 0);
    if (nst > 1) gload(1);
    [[maybe_unused]] int exp_cnt = 0;
    T2_TICK(9)
    int s_end = nst;                                               // stages consumed (XIT: the stop stage)
    for (int s = 0; s < nst; ++s) {
        const unsigned char *buf = smem_raw + (s % kT2Ring) * TB;
        if (s + 1 < nst) {
            const int t = s + 1, sl = t % kT2Ring;
            ring_wait(done_ctr + sl, (unsigned)kT2Waves * (unsigned)(t / kT2Ring));      // the stage that held this slot has been read by every wave
            lwrite(smem_raw + sl * TB);
            ring_signal(fill_ctr + sl);
        }
        if (s + 2 < nst) gload(s + 2);
        if constexpr (XIT) {
            if (ring_wait_flag(fill_ctr + s % kT2Ring, (unsigned)kT2Waves * (unsigned)(s / kT2Ring + 1))) {
                const int stop = __builtin_amdgcn_readfirstlane((int)*(volatile lds_u32 *)(ring_ctr + 6));
                if (s >= stop) { s_end = s; break; }               // wave-uniform; every wave of the workgroup leaves at this stage
            }
        } else ring_wait(fill_ctr + s % kT2Ring, (unsigned)kT2Waves * (unsigned)(s / kT2Ring + 1));
        T2_TICK(4)
        unsigned pm[MST / 64];                                     // bit 16 * (t & 1) + i of pm[t / 2]: accumulator i of tile t passed
        constexpr int NTL = MST / 32;
        auto masks = [&](int t, const f32x16 &acc) {
            const float sn = stage_norm[(s * MST + t * 32) >> (D <= 64 ? 6 : 5)];      // the largest scaled item norm of the tile's 64-item (32 at d = 128) stretch of the stream
            const float tf = thr - fmaf(Ereg, sn, Eabs);
            // most tiles of a warm-started pass (and of the later part of a cold one) hold no survivor for any of the wave's 32 users: the lane's largest of its 16
            // scores (eight v_max3) against its threshold and one ballot first -- the 32-instruction bit mask only where somebody passes (warm 24.4 -> 23.9 ms)
            unsigned p16 = 0u;
            const float m1 = __builtin_fmaxf(__builtin_fmaxf(acc[0], acc[1]), acc[2]), m2 = __builtin_fmaxf(__builtin_fmaxf(acc[3], acc[4]), acc[5]);
            const float m3 = __builtin_fmaxf(__builtin_fmaxf(acc[6], acc[7]), acc[8]), m4 = __builtin_fmaxf(__builtin_fmaxf(acc[9], acc[10]), acc[11]);
            const float m5 = __builtin_fmaxf(__builtin_fmaxf(acc[12], acc[13]), acc[14]);
            const float mx = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(m1, m2), m3), __builtin_fmaxf(__builtin_fmaxf(m4, m5), acc[15]));
            if (__builtin_amdgcn_ballot_w64(!(mx < tf)) != 0ull) {         // (a -inf threshold passes, +inf never: as the sign test below)
                unsigned fails = 0u;
#pragma unroll
                for (int i = 15; i >= 0; --i) fails = __builtin_amdgcn_alignbit(fails, __float_as_uint(acc[i] - tf), 31);     // sign(score - threshold)
                p16 = ~fails & 0xffffu;
            }
            if (s == nst - 1) {                                    // rows past I
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (s * MST + t * 32 + 8 * (i >> 2) + 4 * h + (i & 3) >= I) p16 &= ~(1u << i);
            }
            if (t & 1) pm[t / 2] |= p16 << 16; else pm[t / 2] = p16;
        };
#if ARL_TOPK2_PREFETCH
        // the fragment reads of tile t + 1 are issued right behind the MFMAs of tile t, into the same registers (the hardware orders the overwrite behind
        // the MFMAs' operand reads): their LDS latency runs under the matrix work and the pre-filter of tile t
        f16x8 af[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) af[ks] = *reinterpret_cast<const f16x8 *>(buf + n * RS + ks * 32 + h * 16);
#pragma unroll
        for (int t = 0; t < NTL; ++t) {
            f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks], bu[ks], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (t + 1 < NTL) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) af[ks] = *reinterpret_cast<const f16x8 *>(buf + ((t + 1) * 32 + n) * RS + ks * 32 + h * 16);
            } else ring_signal(done_ctr + s % kT2Ring);            // this wave's fragment reads of the stage are done (issued: LDS executes a wave's operations in order)
            __builtin_amdgcn_sched_barrier(0);
            masks(t, acc);
            if (NTL >= 4 && t == NTL / 2 - 1) flush_slot();        // mid-stage slot: the pending merge has had half a stage of matrix work to land
        }
#else
#pragma unroll
        for (int t = 0; t < NTL; ++t) {
            f16x8 af[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) af[ks] = *reinterpret_cast<const f16x8 *>(buf + (t * 32 + n) * RS + ks * 32 + h * 16);
            f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks], bu[ks], acc, 0, 0, 0);
            masks(t, acc);
            if (NTL >= 4 && t == NTL / 2 - 1) flush_slot();        // mid-stage slot: the pending merge has had half a stage of matrix work to land
        }
        ring_signal(done_ctr + s % kT2Ring);                        // this wave's fragment reads of the stage are done
#endif
        // (tried and dropped, r04_experiments.md 5: a software pipeline over the tiles with two accumulator sets -- 61 ms, the second set does not fit 128
        // registers beside a pending merge; a wave-uniform bit mask of the rows whose queue has reached kT2QFlush, kept by the appends, instead of reading
        // the 32 counters back in every merge slot, with a 64-bit form of the append loop -- 28.8 vs 27.4 ms)
        T2_TICK(0)
#ifdef ARL_TOPK2_PROF
        { int pc = 0; for (int q = 0; q < MST / 64; ++q) pc += __popc(pm[q]); for (int off = 32; off > 0; off >>= 1) pc += __shfl_xor(pc, off); T2_COUNT(8, pc) Q_t0 = clock64(); }
#endif
#if ARL_TOPK2_EXP == 1
        { for (int q = 0; q < MST / 64; ++q) exp_cnt += __popc(pm[q]); continue; }      // probe: scores + masks only
#endif
        // appends: one candidate per lane and round
        for (;;) {
            bool have = false;
#pragma unroll
            for (int q = 0; q < MST / 64; ++q) have = have || pm[q] != 0u;
            if (__builtin_amdgcn_ballot_w64(have) == 0ull) break;
            bool ovf = false;
            if (have) {
                int q = 0;
#pragma unroll
                for (int x = MST / 64 - 1; x >= 0; --x) q = pm[x] != 0u ? x : q;
                unsigned w = pm[0];
#pragma unroll
                for (int x = 1; x < MST / 64; ++x) w = q == x ? pm[x] : w;
                const int b = __ffs(w) - 1;
                const unsigned pos = (unsigned)(s * MST + q * 64 + (b >> 4) * 32 + 8 * ((b & 15) >> 2) + 4 * h + (b & 3));
                const unsigned slot = __hip_atomic_fetch_add((lds_u32 *)(qcnt + n), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (slot < (unsigned)kT2QCap) {
                    *(volatile lds_u32 *)(qpos + n * kT2QCap + slot) = pos;
                    w &= w - 1u;
#pragma unroll
                    for (int x = 0; x < MST / 64; ++x) pm[x] = q == x ? w : pm[x];
                } else ovf = true;
            }
            if (__builtin_amdgcn_ballot_w64(ovf) != 0ull) flush_rows_with((unsigned)kT2QCap);
        }
        // stage end: the merge issued a stage ago is completed, the next one (a row whose queue has reached kT2QFlush) is issued
        T2_TICK(1)
        flush_slot();
        T2_TICK(3)
        if constexpr (XIT) {
            if ((s & 7) == 7) {
                const int nst64 = (I + (D <= 64 ? 64 : 32) - 1) / (D <= 64 ? 64 : 32);
                const int nx = min(((s + 1) * MST) >> (D <= 64 ? 6 : 5), nst64);       // the first later stretch of the stream, in the norm arrays' stages
                const float kx = kExitK * stage_norm[nst64 + 1 + nx];                   // suffix maximum: the largest scaled norm of any later item (0 behind the end)
                const bool fin = fmaf(Ereg, kx, 2.f * Eabs) < thr;                      // (+inf thresholds of users past U: finished; NaN never)
                if (__builtin_amdgcn_ballot_w64(!fin) == 0ull) {
                    unsigned *fw = ring_ctr + 8;
                    if (lane == 0) *(volatile lds_u32 *)(fw + wv) = 1u;
                    const unsigned seen = lane < kT2Waves ? *(volatile lds_u32 *)(fw + lane) : 1u;      // (after this wave's own store: in order)
                    if (__builtin_amdgcn_ballot_w64(seen == 0u) == 0ull) {
                        if (lane == 0) __hip_atomic_fetch_min((lds_u32 *)(ring_ctr + 6), (unsigned)(s + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (lane < kT2Ring) __hip_atomic_fetch_or((lds_u32 *)(fill_ctr + lane), 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // (issued after the minimum)
                    }
                }
            }
        }
    }
    flush_complete();
    flush_rows_with(1u);
#if ARL_TOPK2_EXP == 1
    if (u < U && h == 0) top_idx[(size_t)u * k] = exp_cnt;
    return;
#endif
    if (stats != nullptr && tid == 0) {                            // (the first form's counters, in its 64- / 32-item stages)
        const int nst64 = (I + (D <= 64 ? 64 : 32) - 1) / (D <= 64 ? 64 : 32);
        atomicAdd(stats, (unsigned long long)min(s_end * (MST / (D <= 64 ? 64 : 32)), nst64));
        atomicAdd(stats + 1, 1ull);
    }
    __threadfence();
    for (int r = 0; r < 32; ++r) {
        const int ur = u_base + r;
        if (ur < U && lane < k) {
            const unsigned kh = reinterpret_cast<const unsigned *>(top_val)[(size_t)ur * k + lane], kl = (unsigned)top_idx[(size_t)ur * k + lane];
            const unsigned long long key = ((unsigned long long)kh << 32) | kl;
            top_idx[(size_t)ur * k + lane] = cand_item(key);
            top_val[(size_t)ur * k + lane] = cand_score(key) * score_unscale;
            if (warm && lane == k - 1 && (key == 0ull || cand_score(key) <= -5e8f * score_scale)) atomicOr(underflow, 1);      // (as the first form)
        }
    }
#ifdef ARL_TOPK2_PROF
    if (lane < 10 && u_base < U) top_val[(size_t)u_base * k + lane] = (float)Q_acc[0 + lane];      // (over the first user's values; top_idx stays valid.  Dynamic index: the array lives in scratch -- a profiling build)
#endif
}

// ================================================================================================
// CW term of the attacks' surrogate loss from the users' top-k lists (attack/White/CLeaR.py:83-95, PGA.py:104-116):
//     L = c * sum over real users u and targets t of  <X_u, X_neg(u,t)> - <X_u, X_tg(t)>,   neg(u, t) = top_idx[u][k - 1 - t]  (successive .pop()s)
// and G = dL/dX on the packed table X = [user rows | item rows]: L is bilinear, L = 1/2 X^T M X, G = M X.  Round 3 built M as a CSR per step from ~25
// ATen launches (radix sort, searchsorted, cumsum, scatters) and ran an SpMM; here the term is five launches of its own and deterministic:
//   1 cw_user_kernel    user rows:  G_u = c (sum_t X_neg - sum_t X_tg), the loss terms <X_u, G_u>, column sums of the real users' rows, |X| maximum,
//                       and the histogram of the negatives (integer atomics: order-free);
//   2 cw_scan_kernel    folds the partials in a fixed order, cuts the items into groups of IPG rows and the groups' entries into slices, fixes the scale;
//   3 cw_fill_kernel    every (user, negative) entry goes to its group's bucket (slot by an integer atomic: the ORDER inside a bucket varies from run to
//                       run -- it does not matter, see 4);
//   4 cw_item_kernel    item rows:  sum over the entries of an item of X_u.  One workgroup per slice keeps the group's rows in LDS as 64-bit FIXED-POINT
//                       accumulators (value * 2^e, e chosen from the table's largest magnitude and the entry count so that no sum can overflow): integer
//                       addition is associative, so the sums are bit-identical whatever the order and exact to 2^-e (far below fp32 rounding of the
//                       same sum) -- no float atomics, no sort.  A slice's partial goes to the global accumulator by 64-bit integer atomics;
//   5 cw_finish_kernel  G_i = c (sum - [i is a target] * column sum) in fp32, and the SFA term's row multiplicities w (CLeaR.py:98-103).
// ================================================================================================
constexpr int kCwParts = 1024;       // workgroups of cw_user_kernel = partial sums folded by cw_scan_kernel
constexpr int kCwSlice = 8192;       // entries per workgroup of cw_item_kernel
constexpr int kCwItemThreads = 1024;
constexpr int kCwMaxGroups = 8192;   // item groups (LDS histograms of that many bins)
// item rows per group: IPG * d * 8 B of LDS accumulators -- 64 KB at d = 64 (two workgroups of cw_item_kernel per CU), 128 KB at d = 128 / 256
__host__ __device__ inline int cw_ipg(int d) { return d <= 128 ? 128 : 64; }

struct CwWs {                        // carved out of the caller's workspace (arl_cw_topk_term_workspace_bytes)
    float *part;                     // [kCwParts][d + 2]: column sums, loss terms, |x| maximum
    float *colsum;                   // [d]
    double *scale;                   // [0] 2^e, [1] 2^-e
    int32_t *neg_cnt;                // [I]: filled by cw_item_kernel (per-slice counts)
    int32_t *grp_tot;                // [G]: entries per item group, filled by cw_user_kernel (per-workgroup counts)
    int32_t *grp_off;                // [G + 1]
    int32_t *cursor;                 // [G]
    int32_t *slices;                 // [max_slices][4]: group, begin, end, -
    int32_t *n_slices;               // [1]
    int2 *bucket;                    // [n_real * T]: (user, item)
    long long *acc;                  // [I * d]
};

// LPR lanes x float4 own one user row (d = 64: 16 lanes, four rows per wave in flight); d % 4 == 0.
template <int LPR>
__global__ __launch_bounds__(kCwItemThreads) void cw_user_kernel(const float *__restrict__ X, int d, long long Up, int n_real, const int32_t *__restrict__ top_idx, int k,
                                                                 const long long *__restrict__ targets, int T, float c, float *__restrict__ G, float *__restrict__ w,
                                                                 CwWs W, int rows_per_wg, int ipg, int n_groups) {
    constexpr int NW = kCwItemThreads / kWave, RPV = kWave / LPR;  // waves per workgroup, rows per wave
    __shared__ float red[NW][256 + 2];
    __shared__ int ghist[kCwMaxGroups];                            // entries per item group seen by this workgroup (same-address GLOBAL atomics serialise: the
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;      //   tails of everybody's lists fall on few popular items -- counted in LDS, flushed once)
    const int q = lane % LPR, grp = lane / LPR;
    const bool act = q * 4 < d;
    for (int gI = threadIdx.x; gI < n_groups; gI += kCwItemThreads) ghist[gI] = 0;
    __syncthreads();
    float4 tgs = make_float4(0.f, 0.f, 0.f, 0.f), cs = tgs;
    if (act)
        for (int t = 0; t < T; ++t) tgs = add4(tgs, *reinterpret_cast<const float4 *>(X + (size_t)(Up + targets[t]) * d + q * 4));
    float dot = 0.f, amax = 0.f;
    const long long r0 = (long long)blockIdx.x * rows_per_wg, r1 = min(r0 + rows_per_wg, Up);
    for (long long r = r0 + wv * RPV + grp; r < r1; r += NW * RPV) {
        float4 *g = reinterpret_cast<float4 *>(G + (size_t)r * d + q * 4);
        if (r >= n_real) {                                         // fake users take no part in the CW pairs
            if (act) *g = make_float4(0.f, 0.f, 0.f, 0.f);
            if (w && q == 0) w[r] = 0.f;
            continue;
        }
        const int32_t *tl = top_idx + (size_t)r * k + (k - 1);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int t0 = 0; t0 < T; t0 += 8) {                        // eight gathers in flight per lane
            int it[8]; float4 x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) it[j] = t0 + j < T ? tl[-(t0 + j)] : -1;
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = (it[j] >= 0 && act) ? *reinterpret_cast<const float4 *>(X + (size_t)(Up + it[j]) * d + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc = add4(acc, x[j]);     // t ascending
        }
        for (int t = q; t < T; t += LPR) atomicAdd(&ghist[tl[-t] / ipg], 1);
        if (act) {
            const float4 x = *reinterpret_cast<const float4 *>(X + (size_t)r * d + q * 4);
            const float4 gv = make_float4(c * (acc.x - tgs.x), c * (acc.y - tgs.y), c * (acc.z - tgs.z), c * (acc.w - tgs.w));
            *g = gv;
            dot = fmaf(x.x, gv.x, dot); dot = fmaf(x.y, gv.y, dot); dot = fmaf(x.z, gv.z, dot); dot = fmaf(x.w, gv.w, dot);
            cs = add4(cs, x);
            amax = fmaxf(fmaxf(amax, fmaxf(fabsf(x.x), fabsf(x.y))), fmaxf(fabsf(x.z), fabsf(x.w)));
        }
        if (w && q == 0) w[r] = (float)T;
    }
    dot = wave_sum(dot);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off));
#pragma unroll
    for (int off = LPR; off < kWave; off <<= 1) {                  // the wave's row groups, fixed butterfly order
        cs.x += __shfl_xor(cs.x, off); cs.y += __shfl_xor(cs.y, off); cs.z += __shfl_xor(cs.z, off); cs.w += __shfl_xor(cs.w, off);
    }
    if (grp == 0 && act) { red[wv][q * 4] = cs.x; red[wv][q * 4 + 1] = cs.y; red[wv][q * 4 + 2] = cs.z; red[wv][q * 4 + 3] = cs.w; }
    if (lane == 0) { red[wv][256] = dot; red[wv][257] = amax; }
    __syncthreads();
    for (int gI = threadIdx.x; gI < n_groups; gI += kCwItemThreads)
        if (ghist[gI]) atomicAdd(W.grp_tot + gI, ghist[gI]);
    float *out = W.part + (size_t)blockIdx.x * (d + 2);
    for (int col = threadIdx.x; col < d + 2; col += kCwItemThreads) {
        const int src = col < d ? col : 256 + (col - d);
        float v = red[0][src];
        for (int qq = 1; qq < NW; ++qq) v = (col == d + 1) ? fmaxf(v, red[qq][src]) : v + red[qq][src];      // waves in a fixed order
        out[col] = v;
    }
}

__global__ __launch_bounds__(kCwItemThreads) void cw_scan_kernel(CwWs W, int d, int n_groups, int n_parts, int count_log2, float *__restrict__ loss) {
    __shared__ int s_tot[kCwItemThreads], s_sl[kCwItemThreads];
    __shared__ float fin[2];
    const int tid = threadIdx.x;
    if (tid < d + 2) {                                             // fold the workgroups' partials, parts in a fixed order, eight loads in flight
        float v = tid == d + 1 ? 0.f : 0.f;
        for (int p0 = 0; p0 < n_parts; p0 += 8) {
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = p0 + j < n_parts ? W.part[(size_t)(p0 + j) * (d + 2) + tid] : 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) v = (tid == d + 1) ? fmaxf(v, x[j]) : v + x[j];
        }
        if (tid < d) W.colsum[tid] = v;
        else fin[tid - d] = v;
    }
    // groups: thread t owns groups [t * GPT, (t + 1) * GPT); block-wide exclusive scan of (entries, slices)
    constexpr int GPT = kCwMaxGroups / kCwItemThreads;
    int tot[GPT], my_tot = 0, my_sl = 0;
#pragma unroll
    for (int j = 0; j < GPT; ++j) {
        const int g = tid * GPT + j;
        tot[j] = g < n_groups ? W.grp_tot[g] : 0;
        my_tot += tot[j];
        my_sl += (tot[j] + kCwSlice - 1) / kCwSlice;
    }
    s_tot[tid] = my_tot; s_sl[tid] = my_sl;
    __syncthreads();
    for (int off = 1; off < kCwItemThreads; off <<= 1) {
        const int a = tid >= off ? s_tot[tid - off] : 0, b = tid >= off ? s_sl[tid - off] : 0;
        __syncthreads();
        s_tot[tid] += a; s_sl[tid] += b;
        __syncthreads();
    }
    int off_e = s_tot[tid] - my_tot, off_s = s_sl[tid] - my_sl;   // exclusive prefixes
#pragma unroll
    for (int j = 0; j < GPT; ++j) {
        const int g = tid * GPT + j;
        if (g < n_groups) {
            W.grp_off[g] = off_e;
            W.cursor[g] = 0;
            for (int b = 0; b < tot[j]; b += kCwSlice) {
                int32_t *sl = W.slices + 4 * off_s++;
                sl[0] = g; sl[1] = off_e + b; sl[2] = off_e + min(b + kCwSlice, tot[j]);
            }
            off_e += tot[j];
        }
    }
    if (tid == kCwItemThreads - 1) { W.grp_off[n_groups] = s_tot[tid]; W.n_slices[0] = s_sl[tid]; }
    if (tid == 0) {
        loss[0] = fin[0];
        // scale 2^e with |x| < 2^(ea + 1) for every element and at most 2^count_log2 addends per sum: |sum * 2^e| < 2^(ea + 1 + count_log2 + e) = 2^61
        const float amax = fin[1];
        int e = 0;
        if (amax > 0.f && amax < INFINITY) e = 61 - (ilogbf(amax) + 1) - count_log2;
        e = max(min(e, 1000), -1000);
        W.scale[0] = ldexp(1.0, e); W.scale[1] = ldexp(1.0, -e);
    }
}

// One slot per entry inside its group's bucket.  The slot comes from an LDS counter per group (1 024 entries per workgroup), and ONE global atomic per
// workgroup and touched group reserves the workgroup's range: the lists' tails concentrate on few groups, and 5 M global atomics on a handful of addresses
// serialise (~12 ns each: +15 ms on the CLeaR step in the first version of this kernel).
__global__ __launch_bounds__(kCwItemThreads) void cw_fill_kernel(const int32_t *__restrict__ top_idx, int k, int T, long long n_entries, int ipg, int n_groups, CwWs W) {
    __shared__ int cnt[kCwMaxGroups], base[kCwMaxGroups];
    for (int g = threadIdx.x; g < n_groups; g += kCwItemThreads) cnt[g] = 0;
    __syncthreads();
    const long long e = (long long)blockIdx.x * kCwItemThreads + threadIdx.x;
    int u = 0, item = 0, g = 0, slot = 0;
    if (e < n_entries) {
        u = (int)(e / T);
        item = top_idx[(size_t)u * k + (k - 1 - (int)(e - (long long)u * T))];
        g = item / ipg;
        slot = atomicAdd(&cnt[g], 1);
    }
    __syncthreads();
    for (int qg = threadIdx.x; qg < n_groups; qg += kCwItemThreads)
        if (cnt[qg]) base[qg] = atomicAdd(W.cursor + qg, cnt[qg]);
    __syncthreads();
    if (e < n_entries) W.bucket[W.grp_off[g] + base[g] + slot] = make_int2(u, item);
}

__global__ __launch_bounds__(kCwItemThreads) void cw_item_kernel(const float *__restrict__ X, int d, int n_items, int ipg, CwWs W) {
    extern __shared__ unsigned long long cw_acc[];                 // [ipg][d] fixed-point sums of this slice
    __shared__ int icnt[128];                                      // entries per item of the group in this slice (the negatives' histogram)
    const int s = blockIdx.x;
    if (s >= W.n_slices[0]) return;                                // (the grid is the static upper bound of the slice count)
    const int grp = W.slices[4 * s], begin = W.slices[4 * s + 1], end = W.slices[4 * s + 2];
    const int cells = ipg * d;
    for (int i = threadIdx.x; i < cells; i += kCwItemThreads) cw_acc[i] = 0ull;
    if (threadIdx.x < 128) icnt[threadIdx.x] = 0;
    __syncthreads();
    const double scale = W.scale[0];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    constexpr int NW = kCwItemThreads / kWave, UNR = 8;            // 16 waves x 8 row gathers in flight
    for (int e0 = begin + wv * UNR; e0 < end; e0 += NW * UNR) {
        int2 en[UNR];
#pragma unroll
        for (int q = 0; q < UNR; ++q) en[q] = e0 + q < end ? W.bucket[e0 + q] : make_int2(-1, 0);        // wave-uniform
        for (int c0 = 0; c0 < d; c0 += 64) {                       // one pass per 64 columns (d = 64: one)
            const int col = c0 + lane;
            float x[UNR];
#pragma unroll
            for (int q = 0; q < UNR; ++q) x[q] = (en[q].x >= 0 && col < d) ? X[(size_t)en[q].x * d + col] : 0.f;
#pragma unroll
            for (int q = 0; q < UNR; ++q)
                if (en[q].x >= 0 && col < d)
                    atomicAdd(cw_acc + (size_t)(en[q].y - grp * ipg) * d + col, (unsigned long long)__double2ll_rn((double)x[q] * scale));
        }
        if (lane < UNR) {
            const int2 mine = e0 + lane < end ? W.bucket[e0 + lane] : make_int2(-1, 0);
            if (mine.x >= 0) atomicAdd(&icnt[mine.y - grp * ipg], 1);
        }
    }
    __syncthreads();
    const long long base = (long long)grp * ipg * d, lim = (long long)n_items * d;
    for (int i = threadIdx.x; i < cells; i += kCwItemThreads) {
        const unsigned long long v = cw_acc[i];
        if (v != 0ull && base + i < lim) atomicAdd(reinterpret_cast<unsigned long long *>(W.acc) + base + i, v);
    }
    if ((int)threadIdx.x < ipg && icnt[threadIdx.x] && grp * ipg + (int)threadIdx.x < n_items) atomicAdd(W.neg_cnt + grp * ipg + threadIdx.x, icnt[threadIdx.x]);
}

__global__ __launch_bounds__(kBlock) void cw_finish_kernel(int d, long long Up, int n_items, int n_real, const long long *__restrict__ targets, int T, float c,
                                                           float *__restrict__ G, float *__restrict__ w, CwWs W) {
    const long long t = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (t >= (long long)n_items * d) return;
    const int i = (int)(t / d), col = (int)(t - (long long)i * d);
    int tc = 0;
    for (int q = 0; q < T; ++q) tc += targets[q] == i;
    const float sum = (float)((double)W.acc[t] * W.scale[1]);
    G[(size_t)(Up + i) * d + col] = c * (sum - (float)tc * W.colsum[col]);
    if (w && col == 0) w[Up + i] = (float)W.neg_cnt[i] + (float)tc * (float)n_real;
}

// per-row top-n by n rounds of block arg-max over a scratch copy (n ~ average user degree, small)
__global__ __launch_bounds__(kBlock) void topn_project_kernel(const float *__restrict__ M, int cols, int n, float *__restrict__ out,
                                                               int32_t *__restrict__ idx, float *__restrict__ scratch) {
    __shared__ unsigned long long best[kBlock];
    const int r = blockIdx.x;
    const float *m = M + (size_t)r * cols;
    float *s = scratch + (size_t)r * cols, *o = out + (size_t)r * cols;
    for (int j = threadIdx.x; j < cols; j += kBlock) { s[j] = m[j]; o[j] = 0.f; }
    __syncthreads();
    for (int t = 0; t < n; ++t) {
        unsigned long long b = 0ull;
        for (int j = threadIdx.x; j < cols; j += kBlock) {
            const float v = s[j];
            if (v != -INFINITY) { const unsigned long long c = pack_cand(v, j); if (c > b) b = c; }
        }
        best[threadIdx.x] = b;
        __syncthreads();
        for (int st = kBlock / 2; st > 0; st >>= 1) {
            if ((int)threadIdx.x < st && best[threadIdx.x + st] > best[threadIdx.x]) best[threadIdx.x] = best[threadIdx.x + st];
            __syncthreads();
        }
        if (threadIdx.x == 0 && best[0] != 0ull) {
            const int j = cand_item(best[0]);
            o[j] = 1.f;
            if (idx) idx[(size_t)r * n + t] = j;
            s[j] = -INFINITY;
        }
        __syncthreads();
    }
}

inline unsigned grid_for(long long work_items, int per_block, unsigned cap = 2048u * 8u) {
    long long g = (work_items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (unsigned)g;
}

}  // namespace

// ordered BPR backward over the 3B contributions, kOrderedWindow per launch
static int launch_bpr_bwd(const float *emb, int64_t d, int64_t item_off, const int32_t *u, const int32_t *p, const int32_t *n, int64_t B, float reg,
                          float upstream, const float *ws, const float *norms, float *G, hipStream_t st, int distinct = 0) {
    const int64_t total = 3 * B;
    for (int64_t c0 = 0; c0 < total; c0 += kOrderedWindow) {
        const int64_t c1 = c0 + kOrderedWindow < total ? c0 + kOrderedWindow : total;
        hipLaunchKernelGGL(bpr_bwd_kernel, dim3((unsigned)((c1 - c0 + kWavesPerBlock - 1) / kWavesPerBlock)), dim3(kBlock), distinct ? 0 : (size_t)(c1 - c0) * 4, st, emb, (int)d,
                           (long long)item_off, u, p, n, (int)B, reg, upstream, ws, norms, G, (int)c0, (int)c1, distinct);
        ARL_LAUNCH_CHECK();
    }
    return ARL_OK;
}

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" {

int arl_norm_adj_values_f32(int64_t n_rows, const int32_t *rowptr, const int32_t *col, const float *w, float *dinv, float *val,
                            arl_stream_t stream) {
    if (!rowptr || !dinv) return ARL_E_NULL;
    if (n_rows < 0 || n_rows > 0x7fffffffll) return ARL_E_RANGE;
    if (n_rows == 0) return ARL_OK;
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)((n_rows + kWavesPerBlock - 1) / kWavesPerBlock);
    hipLaunchKernelGGL(row_dinv_kernel, dim3(grid), dim3(kBlock), 0, st, (int)n_rows, rowptr, w, dinv);
    ARL_LAUNCH_CHECK();
    if (val) {
        if (!col || !w) return ARL_E_NULL;
        hipLaunchKernelGGL(norm_vals_kernel, dim3(grid), dim3(kBlock), 0, st, (int)n_rows, rowptr, col, w, dinv, val);
        ARL_LAUNCH_CHECK();
    }
    return ARL_OK;
}

int arl_norm_vals_coo_f32(const int32_t *erow, const int32_t *col, const float *w, int64_t nnz, const float *dinv, float *val, arl_stream_t stream) {
    if (nnz < 0 || nnz > 0x7fffffffll) return ARL_E_RANGE;
    if (nnz == 0) return ARL_OK;
    if (!erow || !col || !w || !dinv || !val) return ARL_E_NULL;
    if (((uintptr_t)erow | (uintptr_t)col | (uintptr_t)w | (uintptr_t)val) & 15) return ARL_E_ARG;      // 16-B vector accesses
    const long long quads = (nnz + 3) / 4;
    hipLaunchKernelGGL(norm_vals_coo_kernel, dim3((unsigned)((quads + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream, (long long)nnz, erow, col, w,
                       dinv, val);
    ARL_LAUNCH_CHECK();
    return ARL_OK;
}

int arl_norm_adj_values_coo_f32(int64_t n_rows, const int32_t *rowptr, const int32_t *erow, const int32_t *col, const float *w, int64_t nnz,
                                float *dinv, float *val, arl_stream_t stream) {
    if (!rowptr || !dinv || !val) return ARL_E_NULL;
    if (n_rows < 0 || n_rows > 0x7fffffffll || nnz < 0 || nnz > 0x7fffffffll) return ARL_E_RANGE;
    if (n_rows == 0) return ARL_OK;
    if (nnz > 0 && (!erow || !col || !w)) return ARL_E_NULL;
    if (nnz > 0 && (((uintptr_t)erow | (uintptr_t)col | (uintptr_t)w | (uintptr_t)val) & 15)) return ARL_E_ARG;      // 16-B vector accesses
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(row_dinv_kernel, dim3((unsigned)((n_rows + kWavesPerBlock - 1) / kWavesPerBlock)), dim3(kBlock), 0, st, (int)n_rows, rowptr, w, dinv);
    ARL_LAUNCH_CHECK();
    if (nnz > 0) {
        const long long quads = (nnz + 3) / 4;
        hipLaunchKernelGGL(norm_vals_coo_kernel, dim3((unsigned)((quads + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, (long long)nnz, erow, col, w, dinv, val);
        ARL_LAUNCH_CHECK();
    }
    return ARL_OK;
}

int arl_spmm_csr_f32(const arl_csr *A, const float *X, int64_t d, float alpha, float beta, const float *Z, float *Y, arl_stream_t stream) {
    if (!Y) return ARL_E_NULL;
    if (beta != 0.f && !Z) return ARL_E_NULL;
    if (Y == X) return ARL_E_ARG;
    Epi ep = {};
    ep.alpha = alpha; ep.beta = beta; ep.Z = (beta != 0.f) ? Z : nullptr; ep.Y = Y;
    return launch_spmm<EPI_AXPBY>(A, X, d, ep, (hipStream_t)stream);
}

int arl_spmm_csr_rscale_f32(const arl_csr *A, const float *X, int64_t d, const float *row_scale, float alpha, float beta, const float *Z, float *Y,
                            arl_stream_t stream) {
    if (!Y || !row_scale) return ARL_E_NULL;
    if (beta != 0.f && !Z) return ARL_E_NULL;
    if (Y == X) return ARL_E_ARG;
    Epi ep = {};
    ep.alpha = alpha; ep.beta = beta; ep.Z = (beta != 0.f) ? Z : nullptr; ep.Y = Y; ep.rscale = row_scale;
    return launch_spmm<EPI_AXPBY>(A, X, d, ep, (hipStream_t)stream);
}

int arl_spmm_csr_layersum_f32(const arl_csr *A, const float *X, int64_t d, const float *S_in, float *S, float *Y, arl_stream_t stream) {
    if (!S_in || !S) return ARL_E_NULL;
    if (Y == X) return ARL_E_ARG;
    Epi ep = {};
    ep.S_in = S_in; ep.S = S; ep.Y = Y;
    return launch_spmm<EPI_LAYERSUM>(A, X, d, ep, (hipStream_t)stream);
}

int arl_spmm_blocked_f32(const arl_blocked *P, const float *X, int64_t d, float alpha, float beta, const float *Z, const uint8_t *zflags, float *Y,
                         arl_stream_t stream) {
    if (!Y) return ARL_E_NULL;
    if (beta != 0.f && !Z) return ARL_E_NULL;
    if (Y == X) return ARL_E_ARG;
    Epi ep = {};
    ep.alpha = alpha; ep.beta = beta; ep.Z = (beta != 0.f) ? Z : nullptr; ep.zflags = zflags; ep.Y = Y;
    return launch_spmm_blocked<EPI_AXPBY>(P, X, d, ep, (hipStream_t)stream);
}

int arl_spmm_blocked_rscale_f32(const arl_blocked *P, const float *X, int64_t d, const float *row_scale, float alpha, float beta, const float *Z,
                                float *Y, arl_stream_t stream) {
    if (!Y || !row_scale) return ARL_E_NULL;
    if (beta != 0.f && !Z) return ARL_E_NULL;
    if (Y == X) return ARL_E_ARG;
    Epi ep = {};
    ep.alpha = alpha; ep.beta = beta; ep.Z = (beta != 0.f) ? Z : nullptr; ep.Y = Y; ep.rscale = row_scale;
    return launch_spmm_blocked<EPI_AXPBY>(P, X, d, ep, (hipStream_t)stream);
}

int arl_spmm_blocked_layersum_f32(const arl_blocked *P, const float *X, int64_t d, const float *S_in, float *S, float *Y, arl_stream_t stream) {
    if (!S_in || !S) return ARL_E_NULL;
    if (Y == X) return ARL_E_ARG;
    Epi ep = {};
    ep.S_in = S_in; ep.S = S; ep.Y = Y;
    return launch_spmm_blocked<EPI_LAYERSUM>(P, X, d, ep, (hipStream_t)stream);
}

int arl_spmm_blocked_adam_f32(const arl_blocked *P, const float *X, int64_t d, float alpha, float beta, const float *Z, const uint8_t *zflags,
                              float *Pm, float *M, float *V, float lr, float beta1, float beta2, float eps, int64_t step, arl_stream_t stream) {
    if (!Pm || !M || !V) return ARL_E_NULL;
    if (beta != 0.f && !Z) return ARL_E_NULL;
    if (step < 1 || Pm == X) return ARL_E_ARG;
    Epi ep = {};
    ep.alpha = alpha; ep.beta = beta; ep.Z = (beta != 0.f) ? Z : nullptr; ep.zflags = zflags; ep.P = Pm; ep.M = M; ep.V = V;
    {
        const AdamScalars a = adam_scalars(lr, beta1, beta2, step);
        ep.step_size = a.step_size; ep.bc2_sqrt = a.bc2_sqrt; ep.w1 = a.w1; ep.b2 = a.b2; ep.w2 = a.w2; ep.eps = (float)typed_double(eps);
    }
    return launch_spmm_blocked<EPI_ADAM>(P, X, d, ep, (hipStream_t)stream);
}

int arl_spmm_tiled_f32(const arl_tiled *T, const float *X, int64_t d, float alpha, float beta, const float *Z, const uint8_t *zflags, float *Y,
                       arl_stream_t stream) {
    if (!Y) return ARL_E_NULL;
    if (beta != 0.f && !Z) return ARL_E_NULL;
    if (Y == X) return ARL_E_ARG;
    Epi ep = {};
    ep.alpha = alpha; ep.beta = beta; ep.Z = (beta != 0.f) ? Z : nullptr; ep.zflags = zflags; ep.Y = Y;
    return launch_spmm_tiled<EPI_AXPBY>(T, X, d, ep, (hipStream_t)stream);
}

int arl_spmm_tiled_adam_f32(const arl_tiled *T, const float *X, int64_t d, float alpha, float beta, const float *Z, const uint8_t *zflags,
                            float *P, float *M, float *V, float lr, float beta1, float beta2, float eps, int64_t step, arl_stream_t stream) {
    if (!P || !M || !V) return ARL_E_NULL;
    if (beta != 0.f && !Z) return ARL_E_NULL;
    if (step < 1 || P == X) return ARL_E_ARG;
    Epi ep = {};
    ep.alpha = alpha; ep.beta = beta; ep.Z = (beta != 0.f) ? Z : nullptr; ep.zflags = zflags; ep.P = P; ep.M = M; ep.V = V;
    {
        const AdamScalars a = adam_scalars(lr, beta1, beta2, step);
        ep.step_size = a.step_size; ep.bc2_sqrt = a.bc2_sqrt; ep.w1 = a.w1; ep.b2 = a.b2; ep.w2 = a.w2; ep.eps = (float)typed_double(eps);
    }
    return launch_spmm_tiled<EPI_ADAM>(T, X, d, ep, (hipStream_t)stream);
}

int arl_spmm_csr_flagged_f32(const arl_csr *A, const float *X, int64_t d, const uint32_t *xbits, float alpha, float beta, const float *Z,
                             const uint8_t *zflags, float *Y, arl_stream_t stream) {
    if (!Y) return ARL_E_NULL;
    if (beta != 0.f && !Z) return ARL_E_NULL;
    if (Y == X) return ARL_E_ARG;
    Epi ep = {};
    ep.alpha = alpha; ep.beta = beta; ep.Z = (beta != 0.f) ? Z : nullptr; ep.zflags = zflags; ep.Y = Y;
    return launch_spmm<EPI_AXPBY>(A, X, d, ep, (hipStream_t)stream, xbits);
}

int arl_spmm_csr_adam_f32(const arl_csr *A, const float *X, int64_t d, float alpha, float beta, const float *Z, const uint8_t *zflags, float *P,
                          float *M, float *V, float lr, float beta1, float beta2, float eps, int64_t step, arl_stream_t stream) {
    if (!P || !M || !V) return ARL_E_NULL;
    if (beta != 0.f && !Z) return ARL_E_NULL;
    if (step < 1) return ARL_E_ARG;
    if (P == X) return ARL_E_ARG;
    Epi ep = {};
    ep.alpha = alpha; ep.beta = beta; ep.Z = (beta != 0.f) ? Z : nullptr; ep.zflags = zflags; ep.P = P; ep.M = M; ep.V = V;
    {
        const AdamScalars a = adam_scalars(lr, beta1, beta2, step);
        ep.step_size = a.step_size; ep.bc2_sqrt = a.bc2_sqrt; ep.w1 = a.w1; ep.b2 = a.b2; ep.w2 = a.w2; ep.eps = (float)typed_double(eps);
    }
    return launch_spmm<EPI_ADAM>(A, X, d, ep, (hipStream_t)stream);
}

int64_t arl_spmm_csr_rows_workspace_bytes(int64_t n_rows_sel, int64_t nsplit, int64_t d) {
    return (n_rows_sel < 0 || nsplit < 1 || d < 0) ? 0 : (int64_t)sizeof(float) * n_rows_sel * nsplit * d;
}

int arl_spmm_csr_rows_f32(const arl_csr *A, const float *X, int64_t d, const int32_t *rows, int64_t n_rows_sel, int64_t nsplit,
                          const float *const *layers, int64_t n_layers, float alpha, const float *row_weight, float *out_c, void *workspace,
                          arl_stream_t stream) {
    if (!A || !X || !rows || !out_c || !workspace || !A->rowptr) return ARL_E_NULL;
    if (A->nnz > 0 && (!A->col || !A->val)) return ARL_E_NULL;
    if (n_layers < 0 || n_layers > 8 || (n_layers > 0 && !layers)) return ARL_E_ARG;
    if (d <= 0 || d > 256 || (d & 3)) return ARL_E_DIM;
    if (n_rows_sel < 0 || nsplit < 1 || nsplit > 1024 || n_rows_sel * nsplit > 0x7fffffffll) return ARL_E_RANGE;
    if (n_rows_sel == 0) return ARL_OK;
    hipStream_t st = (hipStream_t)stream;
    CsrDev D = {};
    D.n_rows = (int)A->n_rows; D.rowptr = A->rowptr; D.col = A->col; D.val = A->val;
    LayerPtrs LP = {};
    LP.n = (int)n_layers;
    for (int k = 0; k < LP.n; ++k) { if (!layers[k]) return ARL_E_NULL; LP.p[k] = layers[k]; }
    const long long tasks = n_rows_sel * nsplit;
    const unsigned grid = (unsigned)((tasks + kWavesPerBlock - 1) / kWavesPerBlock);
    const int di = (int)d, n = (int)n_rows_sel, ns = (int)nsplit;
    float *part = (float *)workspace;
#define ARL_SUBSET_CASE(LPRV)                                                                                                  \
    do {                                                                                                                       \
        hipLaunchKernelGGL((spmm_subset_kernel<LPRV>), dim3(grid), dim3(kBlock), 0, st, D, X, di, rows, n, ns, part);          \
        ARL_LAUNCH_CHECK();                                                                                                    \
        const unsigned g2 = (unsigned)((n + kWavesPerBlock * (kWave / LPRV) - 1) / (kWavesPerBlock * (kWave / LPRV)));         \
        hipLaunchKernelGGL((subset_finish_kernel<LPRV>), dim3(g2), dim3(kBlock), 0, st, part, n, ns, di, rows, LP, alpha, out_c, row_weight); \
        ARL_LAUNCH_CHECK();                                                                                                    \
    } while (0)
    if (d <= 16) ARL_SUBSET_CASE(4);
    else if (d <= 32) ARL_SUBSET_CASE(8);
    else if (d <= 64) ARL_SUBSET_CASE(16);
    else if (d <= 128) ARL_SUBSET_CASE(32);
    else ARL_SUBSET_CASE(64);
#undef ARL_SUBSET_CASE
    return ARL_OK;
}

int arl_mark_rows_u8(uint8_t *flags, const int32_t *idx, int64_t n, int32_t value, arl_stream_t stream) {
    if (!flags || !idx) return ARL_E_NULL;
    if (n < 0 || n > 0x7fffffffll) return ARL_E_ARG;
    if (n == 0) return ARL_OK;
    hipLaunchKernelGGL(mark_rows_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream, flags, idx, (int)n, (int)value);
    ARL_LAUNCH_CHECK();
    return ARL_OK;
}

int arl_mark_rows_bits_u32(uint32_t *bits, const int32_t *idx, int64_t n, int32_t set, arl_stream_t stream) {
    if (!bits || !idx) return ARL_E_NULL;
    if (n < 0 || n > 0x7fffffffll) return ARL_E_ARG;
    if (n == 0) return ARL_OK;
    hipLaunchKernelGGL(mark_bits_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream, bits, idx, (int)n, (int)set);
    ARL_LAUNCH_CHECK();
    return ARL_OK;
}

int arl_batch_rows_set_f32(float *G, uint8_t *flags, uint32_t *bits, const int32_t *idx, int64_t n, int64_t d, const float *src, float scale,
                           const float *row_scale, uint32_t *dup_bits, arl_stream_t stream) {
    if (!G || !flags || !bits || !idx || !src) return ARL_E_NULL;
    if (n < 0 || n > 0x7fffffffll || d <= 0 || d > 0x7fffffffll) return ARL_E_ARG;
    if (n == 0) return ARL_OK;
    if (dup_bits) {
        hipLaunchKernelGGL(rows_mark_dups_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream, bits, dup_bits, idx, (int)n, row_scale);
        ARL_LAUNCH_CHECK();
    }
    for (int64_t c0 = 0; c0 < n; c0 += kOrderedWindow) {                 // ordered (atomic-free) accumulation, one window per launch
        const int64_t c1 = c0 + kOrderedWindow < n ? c0 + kOrderedWindow : n;
        hipLaunchKernelGGL(rows_add_ordered_kernel<true>, dim3((unsigned)((c1 - c0 + kWavesPerBlock - 1) / kWavesPerBlock)), dim3(kBlock), (size_t)(c1 - c0) * 4,
                           (hipStream_t)stream, G, flags, bits, idx, (int)c0, (int)c1, (int)d, src, scale, row_scale, (const uint32_t *)dup_bits);
        ARL_LAUNCH_CHECK();
    }
    return ARL_OK;
}

int arl_batch_rows_clear_f32(float *G, uint8_t *flags, uint32_t *bits, const int32_t *idx, int64_t n, int64_t d, uint32_t *dup_bits, arl_stream_t stream) {
    if (!G || !flags || !bits || !idx) return ARL_E_NULL;
    if (n < 0 || n > 0x7fffffffll || d <= 0 || d > 0x7fffffffll) return ARL_E_ARG;
    if (n == 0) return ARL_OK;
    hipLaunchKernelGGL(batch_rows_clear_kernel, dim3((unsigned)((n + kWavesPerBlock - 1) / kWavesPerBlock)), dim3(kBlock), 0, (hipStream_t)stream, G, flags,
                       bits, idx, (int)n, (int)d, dup_bits);
    ARL_LAUNCH_CHECK();
    return ARL_OK;
}

int arl_zero_rows_f32(float *dst, const int32_t *idx, int64_t n, int64_t d, arl_stream_t stream) {
    if (!dst || !idx) return ARL_E_NULL;
    if (n < 0 || d <= 0 || n > 0x7fffffffll || d > 0x7fffffffll) return ARL_E_ARG;
    if (n == 0) return ARL_OK;
    hipLaunchKernelGGL(zero_rows_kernel, dim3((unsigned)((n + kWavesPerBlock - 1) / kWavesPerBlock)), dim3(kBlock), 0, (hipStream_t)stream, dst, idx, (int)n, (int)d);
    ARL_LAUNCH_CHECK();
    return ARL_OK;
}

int64_t arl_bpr_l2_workspace_bytes(int64_t B) { return B < 0 ? 0 : (int64_t)sizeof(float) * 4 * B; }

int arl_bpr_l2_fwd_bwd_f32(const float *emb, int64_t d, int64_t item_off, const int32_t *u, const int32_t *p, const int32_t *n, int64_t B,
                           float reg, float upstream, float *loss_out, float *G, void *workspace, int32_t distinct_rows, arl_stream_t stream) {
    if (!emb || !u || !p || !n || !loss_out || !workspace) return ARL_E_NULL;
    if (d <= 0 || B <= 0 || item_off < 0) return ARL_E_ARG;
    if (B > 0x7fffffffll / 4 || d > 0x7fffffffll) return ARL_E_RANGE;
    hipStream_t st = (hipStream_t)stream;
    float *ws = (float *)workspace;
    const unsigned grid = (unsigned)((B + kWavesPerBlock - 1) / kWavesPerBlock);
    hipLaunchKernelGGL(bpr_fwd_kernel, dim3(grid), dim3(kBlock), 0, st, emb, (int)d, (long long)item_off, u, p, n, (int)B, (float)B, ws);
    ARL_LAUNCH_CHECK();
    hipLaunchKernelGGL(bpr_finalize_kernel, dim3(1), dim3(kBlock), 0, st, (int)B, reg, ws, loss_out, 0);
    ARL_LAUNCH_CHECK();
    if (G) {
        const int rc = launch_bpr_bwd(emb, d, item_off, u, p, n, B, reg, upstream, ws, loss_out, G, st, distinct_rows != 0);
        if (rc) return rc;
    }
    return ARL_OK;
}

int arl_bpr_l2_partial_f32(const float *emb, int64_t d, int64_t item_off, const int32_t *u, const int32_t *p, const int32_t *n, int64_t B_local,
                           int64_t B_global, float *sums_out, void *workspace, arl_stream_t stream) {
    if (!emb || !sums_out || !workspace) return ARL_E_NULL;
    if (B_local > 0 && (!u || !p || !n)) return ARL_E_NULL;
    if (d <= 0 || B_local < 0 || B_global < B_local || B_global <= 0 || item_off < 0) return ARL_E_ARG;
    if (B_local > 0x7fffffffll / 4 || d > 0x7fffffffll) return ARL_E_RANGE;
    hipStream_t st = (hipStream_t)stream;
    float *ws = (float *)workspace;
    if (B_local > 0) {
        const unsigned grid = (unsigned)((B_local + kWavesPerBlock - 1) / kWavesPerBlock);
        hipLaunchKernelGGL(bpr_fwd_kernel, dim3(grid), dim3(kBlock), 0, st, emb, (int)d, (long long)item_off, u, p, n, (int)B_local, (float)B_global, ws);
        ARL_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(bpr_finalize_kernel, dim3(1), dim3(kBlock), 0, st, (int)B_local, 0.f, ws, sums_out, 1);
    ARL_LAUNCH_CHECK();
    return ARL_OK;
}

int arl_bpr_l2_backward_f32(const float *emb, int64_t d, int64_t item_off, const int32_t *u, const int32_t *p, const int32_t *n, int64_t B_local,
                            float reg, float upstream, const float *norms4, float *G, const void *workspace, arl_stream_t stream) {
    if (!emb || !norms4 || !G || !workspace) return ARL_E_NULL;
    if (B_local > 0 && (!u || !p || !n)) return ARL_E_NULL;
    if (d <= 0 || B_local < 0 || item_off < 0) return ARL_E_ARG;
    if (B_local > 0x7fffffffll / 4 || d > 0x7fffffffll) return ARL_E_RANGE;
    if (B_local == 0) return ARL_OK;
    return launch_bpr_bwd(emb, d, item_off, u, p, n, B_local, reg, upstream, (const float *)workspace, norms4, G, (hipStream_t)stream);
}

int arl_adam_dense_f32(float *p, const float *g, float *m, float *v, int64_t n, float lr, float beta1, float beta2, float eps, int64_t step,
                       arl_stream_t stream) {
    if (!p || !g || !m || !v) return ARL_E_NULL;
    if (n < 0 || step < 1) return ARL_E_ARG;
    if (n == 0) return ARL_OK;
    const AdamScalars a = adam_scalars(lr, beta1, beta2, step);
    hipStream_t st = (hipStream_t)stream;
    const bool al = (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15u) == 0;
    const long long n4 = al ? n / 4 : 0;
    if (n4 > 0) {
        hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n4, kBlock)), dim3(kBlock), 0, st, (float4 *)p, (const float4 *)g, (float4 *)m, (float4 *)v, n4,
                           a.step_size, a.bc2_sqrt, a.w1, a.b2, a.w2, eps);
        ARL_LAUNCH_CHECK();
    }
    if (n4 * 4 < n) {
        hipLaunchKernelGGL(adam_scalar_kernel, dim3(grid_for(n - n4 * 4, kBlock)), dim3(kBlock), 0, st, p, g, m, v, n4 * 4, (long long)n, a.step_size,
                           a.bc2_sqrt, a.w1, a.b2, a.w2, eps);
        ARL_LAUNCH_CHECK();
    }
    return ARL_OK;
}

int arl_sgd_dense_f32(float *p, const float *g, int64_t n, float lr, arl_stream_t stream) {
    if (!p || !g) return ARL_E_NULL;
    if (n < 0) return ARL_E_ARG;
    if (n == 0) return ARL_OK;
    hipLaunchKernelGGL(sgd_kernel, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, p, g, (long long)n, lr);
    ARL_LAUNCH_CHECK();
    return ARL_OK;
}

int arl_tables_sum_f32(const float *const *tables, int64_t n_tables, int64_t n_elems, float alpha, float *out, arl_stream_t stream) {
    if (!tables || !out) return ARL_E_NULL;
    if (n_tables < 1 || n_tables > 8 || n_elems < 0 || (n_elems & 3)) return ARL_E_ARG;
    if (n_elems == 0) return ARL_OK;
    LayerPtrs LP = {};
    LP.n = (int)n_tables;
    for (int k = 0; k < LP.n; ++k) { if (!tables[k]) return ARL_E_NULL; LP.p[k] = tables[k]; }
    const long long n4 = n_elems / 4;
    hipLaunchKernelGGL(tables_sum_kernel, dim3(grid_for(n4, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, LP, n4, alpha, (float4 *)out);
    ARL_LAUNCH_CHECK();
    return ARL_OK;
}

int arl_gather_rows_f32(const float *src, const int32_t *idx, int64_t n, int64_t d, float *dst, arl_stream_t stream) {
    if (!src || !idx || !dst) return ARL_E_NULL;
    if (n < 0 || d <= 0 || n > 0x7fffffffll || d > 0x7fffffffll) return ARL_E_ARG;
    if (n == 0) return ARL_OK;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((n + kWavesPerBlock - 1) / kWavesPerBlock)), dim3(kBlock), 0, (hipStream_t)stream, src, idx,
                       (int)n, (int)d, dst);
    ARL_LAUNCH_CHECK();
    return ARL_OK;
}

int arl_scatter_add_rows_f32(float *dst, const int32_t *idx, int64_t n, int64_t d, const float *src, float scale, arl_stream_t stream) {
    if (!src || !idx || !dst) return ARL_E_NULL;
    if (n < 0 || d <= 0 || n > 0x7fffffffll || d > 0x7fffffffll) return ARL_E_ARG;
    if (n == 0) return ARL_OK;
    for (int64_t c0 = 0; c0 < n; c0 += kOrderedWindow) {
        const int64_t c1 = c0 + kOrderedWindow < n ? c0 + kOrderedWindow : n;
        hipLaunchKernelGGL(rows_add_ordered_kernel<false>, dim3((unsigned)((c1 - c0 + kWavesPerBlock - 1) / kWavesPerBlock)), dim3(kBlock), (size_t)(c1 - c0) * 4,
                           (hipStream_t)stream, dst, (uint8_t *)nullptr, (uint32_t *)nullptr, idx, (int)c0, (int)c1, (int)d, src, scale, (const float *)nullptr,
                           (const uint32_t *)nullptr);
        ARL_LAUNCH_CHECK();
    }
    return ARL_OK;
}

int arl_rows_axpy_unique_f32(float *dst, const float *src, const int32_t *idx, int64_t n, int64_t d, float alpha, const uint32_t *dup_bits,
                             arl_stream_t stream) {
    if (!src || !idx || !dst) return ARL_E_NULL;
    if (n < 0 || d <= 0 || n > 0x7fffffffll || d > 0x7fffffffll || dst == src) return ARL_E_ARG;
    for (int64_t c0 = 0; c0 < n; c0 += kOrderedWindow) {
        const int64_t c1 = c0 + kOrderedWindow < n ? c0 + kOrderedWindow : n;
        const int lds_n = n <= kOrderedWindow ? (int)n : 0;
        hipLaunchKernelGGL(rows_axpy_unique_kernel, dim3((unsigned)((c1 - c0 + kWavesPerBlock - 1) / kWavesPerBlock)), dim3(kBlock), (size_t)lds_n * 4,
                           (hipStream_t)stream, dst, src, idx, (int)c0, (int)c1, (int)d, alpha, dup_bits, lds_n);
        ARL_LAUNCH_CHECK();
    }
    return ARL_OK;
}

int arl_shard_batch_prep_i32(const int32_t *u, const int32_t *p, const int32_t *n, int64_t B, int64_t u0, int64_t u1, int32_t *lu, float *own,
                             int32_t *item_rows, int32_t *rows_l, arl_stream_t stream) {
    if (!u || !p || !n || !lu || !own || !item_rows || !rows_l) return ARL_E_NULL;
    if (B < 0 || B > 0x7fffffffll / 3 || u0 < 0 || u1 < u0 || u1 > 0x7fffffffll) return ARL_E_ARG;
    if (B == 0) return ARL_OK;
    hipLaunchKernelGGL(shard_batch_prep_kernel, dim3((unsigned)((B + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream, u, p, n, (int)B, (int)u0,
                       (int)u1, lu, own, item_rows, rows_l);
    ARL_LAUNCH_CHECK();
    return ARL_OK;
}

// number of splits of the streamed table when the batch is the resident side: enough workgroups to fill the chip several times over
static int nce_allrows_splits(int64_t nA, int64_t nV) {
    const int64_t row_blocks = (nA + 63) / 64;
    int64_t s = (2048 + row_blocks - 1) / row_blocks;
    const int64_t max_s = (nV + 4095) / 4096;                          // at least 4 096 streamed rows per split
    if (s > max_s) s = max_s;
    if (s > 512) s = 512;
    return (int)(s < 1 ? 1 : s);
}

int64_t arl_nce_allrows_workspace_bytes(int64_t nA, int64_t nV, int64_t d) {
    if (nA <= 0 || nV <= 0 || d <= 0) return 0;
    return (int64_t)sizeof(float) * nce_allrows_splits(nA, nV) * nA * (d + 4);      // per split: an nA x d tile block + nA row sums (padded)
}

#define ARL_NCE_DISPATCH(DV, GRADV, LSERV, FUSEDV, GRID, ...)                                                                      \
    do {                                                                                                                          \
        hipLaunchKernelGGL((nce_allrows_kernel<DV, GRADV, LSERV, FUSEDV>), GRID, dim3(kBlock), 0, st, __VA_ARGS__);               \
        ARL_LAUNCH_CHECK();                                                                                                       \
    } while (0)
#define ARL_NCE_BY_WIDTH(GRADV, LSERV, FUSEDV, GRID, ...)                                                                         \
    do {                                                                                                                          \
        if (d == 16) ARL_NCE_DISPATCH(16, GRADV, LSERV, FUSEDV, GRID, __VA_ARGS__);                                               \
        else if (d == 32) ARL_NCE_DISPATCH(32, GRADV, LSERV, FUSEDV, GRID, __VA_ARGS__);                                          \
        else if (d == 64) ARL_NCE_DISPATCH(64, GRADV, LSERV, FUSEDV, GRID, __VA_ARGS__);                                          \
        else ARL_NCE_DISPATCH(128, GRADV, LSERV, FUSEDV, GRID, __VA_ARGS__);                                                      \
    } while (0)

int arl_nce_allrows_lse_f32(const float *A, int64_t nA, const float *V, int64_t nV, int64_t d, float tau, float *lse, void *workspace,
                            arl_stream_t stream) {
    if (!A || !V || !lse || !workspace) return ARL_E_NULL;
    if (d != 16 && d != 32 && d != 64 && d != 128) return ARL_E_DIM;
    if (nA <= 0 || nV <= 0 || nA > 0x7fffffffll / 128 || nV > 0x7fffffffll / 128 || !(tau > 0.f)) return ARL_E_ARG;
    if (((uintptr_t)A | (uintptr_t)V) & 15) return ARL_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int ns = nce_allrows_splits(nA, nV);
    const int split_len = (int)((nV + ns - 1) / ns);
    float *part = (float *)workspace;
    ARL_NCE_BY_WIDTH(false, true, false, dim3((unsigned)((nA + 63) / 64), (unsigned)ns), A, (int)nA, V, (int)nV, split_len, 1.0f / tau, (const float *)nullptr, part,
                     (float *)nullptr);
    hipLaunchKernelGGL(nce_allrows_lse_finish_kernel, dim3((unsigned)((nA + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, part, ns, (int)nA, 1.0f / tau, lse);
    ARL_LAUNCH_CHECK();
    return ARL_OK;
}

int arl_nce_allrows_grad_f32(const float *A, int64_t nA, const float *V, int64_t nV, int64_t d, float tau, float *lse, int32_t lse_given, float *dA,
                             float *dV, void *workspace, arl_stream_t stream) {
    if (!A || !V || !lse || !workspace || (!dA && !dV)) return ARL_E_NULL;
    if (!lse_given && !dA) return ARL_E_ARG;                          // the log-sum-exp comes out of the dA pass
    if (d != 16 && d != 32 && d != 64 && d != 128) return ARL_E_DIM;
    if (nA <= 0 || nV <= 0 || nA > 0x7fffffffll / 128 || nV > 0x7fffffffll / 128 || !(tau > 0.f)) return ARL_E_ARG;
    if (((uintptr_t)A | (uintptr_t)V | (uintptr_t)dA | (uintptr_t)dV | (uintptr_t)workspace) & 15) return ARL_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int ns = nce_allrows_splits(nA, nV);
    const int split_len = (int)((nV + ns - 1) / ns);
    float *part = (float *)workspace;
    if (dA) {       // the batch resident, the table streamed in splits, partial tiles folded in split order
        const long long n4 = nA * d / 4;
        const dim3 grid((unsigned)((nA + 63) / 64), (unsigned)ns);
        if (lse_given) {
            ARL_NCE_BY_WIDTH(true, true, false, grid, A, (int)nA, V, (int)nV, split_len, 1.0f / tau, (const float *)lse, part, (float *)nullptr);
            hipLaunchKernelGGL(nce_allrows_fold_kernel, dim3(grid_for(n4, kBlock)), dim3(kBlock), 0, st, (const float4 *)part, ns, n4, (float4 *)dA);
        } else {
            float *sums = part + (size_t)ns * nA * d;
            ARL_NCE_BY_WIDTH(true, true, true, grid, A, (int)nA, V, (int)nV, split_len, 1.0f / tau, (const float *)nullptr, part, sums);
            hipLaunchKernelGGL(nce_allrows_fold_norm_kernel, dim3(grid_for(n4, kBlock)), dim3(kBlock), 0, st, (const float4 *)part, (const float *)sums, ns, (int)nA,
                               (int)(d / 4), 1.0f / tau, lse, (float4 *)dA);
        }
        ARL_LAUNCH_CHECK();
    }
    if (dV) {       // the table resident (16 rows per wave), the whole batch streamed
        ARL_NCE_BY_WIDTH(true, false, false, dim3((unsigned)((nV + 63) / 64), 1u), V, (int)nV, A, (int)nA, (int)nA, 1.0f / tau, (const float *)lse, dV, (float *)nullptr);
    }
    return ARL_OK;
}

static int normalize_rows_launch(bool bwd, const float *X, const float *dY, float *nrm, int64_t n, int64_t d, float scale, const float *scale_dev, float *out,
                                 hipStream_t st) {
    int lpr = 1;
    while (lpr * 4 < d) lpr <<= 1;
    const long long waves = (n + 64 / lpr - 1) / (64 / lpr);
    const long long blocks = (waves + kWavesPerBlock - 1) / kWavesPerBlock;
    const unsigned grid = (unsigned)(blocks < 16384 ? blocks : 16384);
    if (bwd) hipLaunchKernelGGL((rows_normalize_kernel<true>), dim3(grid), dim3(kBlock), 0, st, X, dY, nrm, (long long)n, (int)d, lpr, scale, scale_dev, out);
    else hipLaunchKernelGGL((rows_normalize_kernel<false>), dim3(grid), dim3(kBlock), 0, st, X, dY, nrm, (long long)n, (int)d, lpr, scale, scale_dev, out);
    ARL_LAUNCH_CHECK();
    return ARL_OK;
}

int arl_normalize_rows_f32(const float *X, int64_t n, int64_t d, float *Y, float *nrm, arl_stream_t stream) {
    if (!X || !Y || !nrm) return ARL_E_NULL;
    if (n < 0 || d <= 0 || d > 256 || (d & 3)) return ARL_E_DIM;
    if (((uintptr_t)X | (uintptr_t)Y) & 15) return ARL_E_ARG;
    if (n == 0) return ARL_OK;
    return normalize_rows_launch(false, X, nullptr, nrm, n, d, 1.f, nullptr, Y, (hipStream_t)stream);
}

int arl_normalize_rows_bwd_f32(const float *Y, const float *nrm, const float *dY, int64_t n, int64_t d, float scale, const float *scale_dev, float *dX,
                               arl_stream_t stream) {
    if (!Y || !nrm || !dY || !dX) return ARL_E_NULL;
    if (n < 0 || d <= 0 || d > 256 || (d & 3)) return ARL_E_DIM;
    if (((uintptr_t)Y | (uintptr_t)dY | (uintptr_t)dX) & 15) return ARL_E_ARG;
    if (n == 0) return ARL_OK;
    return normalize_rows_launch(true, Y, dY, const_cast<float *>(nrm), n, d, scale, scale_dev, dX, (hipStream_t)stream);
}

int64_t arl_infonce_workspace_bytes(int64_t n, int64_t d) { return (n < 0 || d < 0) ? 0 : (int64_t)sizeof(float) * (4 * n * d + 4 * n); }

int arl_infonce_fwd_bwd_f32(const float *v1, const float *v2, int64_t n, int64_t d, float tau, float upstream, float *loss_out, float *dv1,
                            float *dv2, void *workspace, arl_stream_t stream) {
    if (!v1 || !v2 || !loss_out || !workspace) return ARL_E_NULL;
    if ((dv1 == nullptr) != (dv2 == nullptr)) return ARL_E_ARG;
    if (n <= 0 || n > 8192 || d <= 0 || d > 256 || (d & 3)) return ARL_E_DIM;
    hipStream_t st = (hipStream_t)stream;
    float *a = (float *)workspace, *b = a + n * d, *n1 = b + n * d, *n2 = n1 + n, *ttl = n2 + n, *rowloss = ttl + n;
    const unsigned gw = (unsigned)((n + kWavesPerBlock - 1) / kWavesPerBlock);
    const float inv_tau = 1.0f / tau;
    hipLaunchKernelGGL(nce_normalize_kernel, dim3(gw), dim3(kBlock), 0, st, v1, (int)n, (int)d, a, n1);
    ARL_LAUNCH_CHECK();
    hipLaunchKernelGGL(nce_normalize_kernel, dim3(gw), dim3(kBlock), 0, st, v2, (int)n, (int)d, b, n2);
    ARL_LAUNCH_CHECK();
    hipLaunchKernelGGL(nce_rowsum_kernel, dim3((unsigned)n), dim3(kBlock), 0, st, a, b, (int)n, (int)d, inv_tau, ttl, rowloss);
    ARL_LAUNCH_CHECK();
    hipLaunchKernelGGL(nce_loss_kernel, dim3(1), dim3(kBlock), 0, st, rowloss, (int)n, loss_out);
    ARL_LAUNCH_CHECK();
    if (dv1) {
        const float scale = upstream / ((float)n * tau);
        const size_t shm = sizeof(float) * (256 + (size_t)n + kBlock);
        hipLaunchKernelGGL((nce_grad_kernel<true>), dim3((unsigned)n), dim3(kBlock), shm, st, a, b, ttl, n1, (int)n, (int)d, inv_tau, scale, dv1);
        ARL_LAUNCH_CHECK();
        hipLaunchKernelGGL((nce_grad_kernel<false>), dim3((unsigned)n), dim3(kBlock), shm, st, b, a, ttl, n2, (int)n, (int)d, inv_tau, scale, dv2);
        ARL_LAUNCH_CHECK();
    }
    return ARL_OK;
}

static unsigned stream_grid(long long n4) { const long long b = (n4 + kBlock - 1) / kBlock; return (unsigned)(b < 16384 ? (b > 0 ? b : 1) : 16384); }

int arl_ngcf_combine_f32(const float *P, const float *E, int64_t n, int64_t d, float *ST, arl_stream_t stream) {
    if (!P || !E || !ST) return ARL_E_NULL;
    if (n < 0 || d <= 0 || (d & 3)) return ARL_E_DIM;
    if (n == 0) return ARL_OK;
    hipLaunchKernelGGL(ngcf_combine_kernel, dim3(stream_grid(n * d / 4)), dim3(kBlock), 0, (hipStream_t)stream, (const float4 *)P, (const float4 *)E, (float4 *)ST,
                       (long long)(n * d / 4), (int)(d / 4));
    ARL_LAUNCH_CHECK();
    return ARL_OK;
}

int arl_ngcf_act_f32(float *Z, float *acc, int64_t n, int64_t d, float slope, arl_stream_t stream) {
    if (!Z) return ARL_E_NULL;
    if (n < 0 || d <= 0 || (d & 3)) return ARL_E_DIM;
    if (n == 0) return ARL_OK;
    hipLaunchKernelGGL(ngcf_act_kernel, dim3(stream_grid(n * d / 4)), dim3(kBlock), 0, (hipStream_t)stream, (float4 *)Z, (float4 *)acc, (long long)(n * d / 4), slope);
    ARL_LAUNCH_CHECK();
    return ARL_OK;
}

int arl_ngcf_act_bwd_f32(const float *gOut, const float *Out, int64_t n, int64_t d, float slope, float *gZ, arl_stream_t stream) {
    if (!gOut || !Out || !gZ) return ARL_E_NULL;
    if (n < 0 || d <= 0 || (d & 3)) return ARL_E_DIM;
    if (n == 0) return ARL_OK;
    hipLaunchKernelGGL(ngcf_act_bwd_kernel, dim3(stream_grid(n * d / 4)), dim3(kBlock), 0, (hipStream_t)stream, (const float4 *)gOut, (const float4 *)Out,
                       (float4 *)gZ, (long long)(n * d / 4), slope);
    ARL_LAUNCH_CHECK();
    return ARL_OK;
}

int arl_ngcf_combine_bwd_f32(const float *gST, const float *P, const float *E, int64_t n, int64_t d, float *gP, float *gE, arl_stream_t stream) {
    if (!gST || !P || !E || !gP || !gE) return ARL_E_NULL;
    if (n < 0 || d <= 0 || (d & 3)) return ARL_E_DIM;
    if (n == 0) return ARL_OK;
    hipLaunchKernelGGL(ngcf_combine_bwd_kernel, dim3(stream_grid(n * d / 4)), dim3(kBlock), 0, (hipStream_t)stream, (const float4 *)gST, (const float4 *)P,
                       (const float4 *)E, (float4 *)gP, (float4 *)gE, (long long)(n * d / 4), (int)(d / 4));
    ARL_LAUNCH_CHECK();
    return ARL_OK;
}

static int ngcf_dense_args(int64_t n, int64_t d) {
    if (n < 0 || n > 0x7fffffffll * 16) return ARL_E_RANGE;
    if (d != 16 && d != 32 && d != 64 && d != 128) return ARL_E_DIM;
    return ARL_OK;
}

int arl_ngcf_dense_fwd_f32(const float *P, const float *E, const float *W, int64_t n, int64_t d, float slope, float *out, arl_stream_t stream) {
    if (!P || !E || !W || !out) return ARL_E_NULL;
    const int rc = ngcf_dense_args(n, d);
    if (rc != ARL_OK) return rc;
    if (n == 0) return ARL_OK;
    const int64_t tiles = (n + 15) / 16;
    const int64_t wpb = d >= 128 ? 16 : 4;                           // waves per workgroup (NgcfBlk)
    const int64_t want_blocks = (tiles + wpb - 1) / wpb, cap_blocks = d >= 128 ? 512 : 2048;
    const unsigned grid = (unsigned)(want_blocks < cap_blocks ? want_blocks : cap_blocks);
    const size_t shm = sizeof(float) * 2 * (size_t)d * (size_t)(d + 4);
#define ARL_NGCF_FWD(DV)                                                                                                                         \
    do {                                                                                                                                         \
        hipError_t e1 = hipFuncSetAttribute((const void *)ngcf_dense_fwd_kernel<DV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);     \
        if (e1 != hipSuccess) return (int)e1;                                                                                                    \
        hipLaunchKernelGGL((ngcf_dense_fwd_kernel<DV>), dim3(grid), dim3(NgcfBlk<DV>::value), shm, (hipStream_t)stream, P, E, W, (long long)n, slope, out); \
    } while (0)
    if (d == 16) ARL_NGCF_FWD(16); else if (d == 32) ARL_NGCF_FWD(32); else if (d == 64) ARL_NGCF_FWD(64); else ARL_NGCF_FWD(128);
#undef ARL_NGCF_FWD
    ARL_LAUNCH_CHECK();
    return ARL_OK;
}

int arl_ngcf_dense_dgrad_f32(const float *gOut, const float *Out, const float *P, const float *E, const float *Wt, int64_t n, int64_t d, float slope,
                             float *gZ, float *gP, float *gE, arl_stream_t stream) {
    if (!gOut || !Out || !P || !E || !Wt || !gZ || !gP || !gE) return ARL_E_NULL;
    const int rc = ngcf_dense_args(n, d);
    if (rc != ARL_OK) return rc;
    if (n == 0) return ARL_OK;
    const int64_t tiles = (n + 15) / 16;
    const int64_t wpb = d >= 128 ? 16 : 4;
    const int64_t want_blocks = (tiles + wpb - 1) / wpb, cap_blocks = d >= 128 ? 512 : 2048;
    const unsigned grid = (unsigned)(want_blocks < cap_blocks ? want_blocks : cap_blocks);
    const size_t shm = sizeof(float) * (size_t)d * (size_t)(2 * d + 4);
#define ARL_NGCF_DG(DV)                                                                                                                          \
    do {                                                                                                                                         \
        hipError_t e1 = hipFuncSetAttribute((const void *)ngcf_dense_dgrad_kernel<DV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);   \
        if (e1 != hipSuccess) return (int)e1;                                                                                                    \
        hipLaunchKernelGGL((ngcf_dense_dgrad_kernel<DV>), dim3(grid), dim3(NgcfBlk<DV>::value), shm, (hipStream_t)stream, gOut, Out, P, E, Wt, (long long)n, slope, gZ, gP, gE); \
    } while (0)
    if (d == 16) ARL_NGCF_DG(16); else if (d == 32) ARL_NGCF_DG(32); else if (d == 64) ARL_NGCF_DG(64); else ARL_NGCF_DG(128);
#undef ARL_NGCF_DG
    ARL_LAUNCH_CHECK();
    return ARL_OK;
}

int64_t arl_ngcf_wgrad_workspace_bytes(int64_t n, int64_t d) {
    if (n <= 0 || d <= 0) return 0;
    const int64_t tiles = (n + 15) / 16;
    const int64_t cap = 512;
    const int64_t blocks = tiles < cap ? tiles : cap;
    return blocks * 2 * d * d * (int64_t)sizeof(float);
}

int arl_ngcf_dense_wgrad_f32(const float *P, const float *E, const float *gZ, int64_t n, int64_t d, float *gW, void *workspace, arl_stream_t stream) {
    if (!P || !E || !gZ || !gW || !workspace) return ARL_E_NULL;
    const int rc = ngcf_dense_args(n, d);
    if (rc != ARL_OK) return rc;
    if (n == 0) { hipError_t e = hipMemsetAsync(gW, 0, sizeof(float) * 2 * d * d, (hipStream_t)stream); return e == hipSuccess ? ARL_OK : (int)e; }
    const int64_t tiles = (n + 15) / 16;
    const int64_t cap = 512;
    const unsigned blocks = (unsigned)(tiles < cap ? tiles : cap);
    float *part = (float *)workspace;
    if (d == 16) hipLaunchKernelGGL((ngcf_dense_wgrad_kernel<16>), dim3(blocks), dim3(kNgcfBlock), 0, (hipStream_t)stream, P, E, gZ, (long long)n, part);
    else if (d == 32) hipLaunchKernelGGL((ngcf_dense_wgrad_kernel<32>), dim3(blocks), dim3(kNgcfBlock), 0, (hipStream_t)stream, P, E, gZ, (long long)n, part);
    else if (d == 64) hipLaunchKernelGGL((ngcf_dense_wgrad_kernel<64>), dim3(blocks), dim3(kNgcfBlock), 0, (hipStream_t)stream, P, E, gZ, (long long)n, part);
    else hipLaunchKernelGGL((ngcf_dense_wgrad_kernel<128>), dim3(blocks), dim3(kNgcfBlock), 0, (hipStream_t)stream, P, E, gZ, (long long)n, part);
    ARL_LAUNCH_CHECK();
    const int n_elem = (int)(2 * d * d);
    hipLaunchKernelGGL(ngcf_wgrad_fold_kernel, dim3((unsigned)((n_elem + 15) / 16)), dim3(kBlock), 0, (hipStream_t)stream, part, (int)blocks, n_elem, gW);
    ARL_LAUNCH_CHECK();
    return ARL_OK;
}


int arl_simgcl_perturb_rng_f32(const float *src, float *dst, int64_t n, int64_t d, const int32_t *row_ids, float eps, uint64_t seed, uint64_t stream_id,
                               arl_stream_t stream) {
    if (!src || !dst) return ARL_E_NULL;
    if (n < 0 || d <= 0 || d > 256 || n > 0x7fffffffll) return ARL_E_ARG;
    if (n == 0) return ARL_OK;
    unsigned long long key = seed ^ (stream_id * 0x9E3779B97F4A7C15ull);          // one hash of (seed, stream) keys the whole call
    key += 0x9E3779B97F4A7C15ull; key = (key ^ (key >> 30)) * 0xBF58476D1CE4E5B9ull; key = (key ^ (key >> 27)) * 0x94D049BB133111EBull; key ^= key >> 31;
    hipLaunchKernelGGL(simgcl_perturb_rng_kernel, dim3((unsigned)((n + kWavesPerBlock - 1) / kWavesPerBlock)), dim3(kBlock), 0, (hipStream_t)stream, src, dst,
                       (int)n, (int)d, row_ids, eps, key);
    ARL_LAUNCH_CHECK();
    return ARL_OK;
}

int arl_simgcl_perturb_f32(float *E, const float *noise, int64_t n, int64_t d, float eps, arl_stream_t stream) {
    if (!E || !noise) return ARL_E_NULL;
    if (n < 0 || d <= 0 || n > 0x7fffffffll || d > 0x7fffffffll) return ARL_E_ARG;
    if (n == 0) return ARL_OK;
    hipLaunchKernelGGL(simgcl_perturb_kernel, dim3((unsigned)((n + kWavesPerBlock - 1) / kWavesPerBlock)), dim3(kBlock), 0, (hipStream_t)stream, E, noise,
                       (int)n, (int)d, eps);
    ARL_LAUNCH_CHECK();
    return ARL_OK;
}

int64_t arl_sfa_workspace_bytes(int64_t n_rows, int64_t d) {
    return (n_rows < 0 || d <= 0) ? 0 : (int64_t)sizeof(float) * (2 * n_rows + (int64_t)kSfaMaxBlocks * kSfaStride + 520);
}

int arl_sfa_l1_fwd_bwd_f32(const float *X, const float *w, const float *r0, int64_t n_rows, int64_t d, int64_t numel_h, float scale,
                           int32_t accumulate, float *loss_out, float *G, void *workspace, arl_stream_t stream) {
    if (!X || !w || !r0 || !loss_out || !workspace) return ARL_E_NULL;
    if (n_rows <= 0 || n_rows > 0x7fffffffll || numel_h <= 0) return ARL_E_ARG;
    if (d <= 0 || d > 256) return ARL_E_DIM;
    hipStream_t st = (hipStream_t)stream;
    float *q = (float *)workspace, *s = q + n_rows, *part = s + n_rows, *coef = part + (size_t)kSfaMaxBlocks * kSfaStride;
    const int64_t want = (n_rows + kWavesPerBlock - 1) / kWavesPerBlock;
    const int nblk = (int)(want < kSfaMaxBlocks ? want : kSfaMaxBlocks);
    hipLaunchKernelGGL((sfa_reduce_pass_kernel<1>), dim3(nblk), dim3(kBlock), 0, st, X, w, r0, (int)n_rows, (int)d, q, part);
    ARL_LAUNCH_CHECK();
    hipLaunchKernelGGL(sfa_fold_kernel, dim3((unsigned)d), dim3(kBlock), 0, st, part, nblk, (int)d, coef, coef + 514);
    ARL_LAUNCH_CHECK();
    hipLaunchKernelGGL((sfa_reduce_pass_kernel<2>), dim3(nblk), dim3(kBlock), 0, st, X, w, coef, (int)n_rows, (int)d, s, part);
    ARL_LAUNCH_CHECK();
    hipLaunchKernelGGL(sfa_fold_kernel, dim3((unsigned)d + 1), dim3(kBlock), 0, st, part, nblk, (int)d, coef + 256, coef + 514);
    ARL_LAUNCH_CHECK();
    hipLaunchKernelGGL(sfa_finalize_kernel, dim3(1), dim3(kBlock), 0, st, (int)d, (float)(1.0 / (double)numel_h), coef, loss_out);
    ARL_LAUNCH_CHECK();
    if (G) {
        hipLaunchKernelGGL(sfa_grad_kernel, dim3((unsigned)want), dim3(kBlock), 0, st, X, w, r0, q, s, coef, (int)n_rows, (int)d, scale, (int)accumulate, G);
        ARL_LAUNCH_CHECK();
    }
    return ARL_OK;
}

// The same computation cut at its two global reductions, for row sets that are partitioned over ranks (user-sharded CLeaR step): the
// caller sum-all-reduces r (d floats) between stage 1 and 2 and [a | S] (d + 1 floats) between stage 2 and 3.  One workspace for all three.
static int sfa_stage_args(const float *X, const float *w, int64_t n_rows, int64_t d, void *workspace) {
    if (!X || !w || !workspace) return ARL_E_NULL;
    if (n_rows <= 0 || n_rows > 0x7fffffffll) return ARL_E_ARG;
    if (d <= 0 || d > 256) return ARL_E_DIM;
    return ARL_OK;
}

int arl_sfa_stage1_f32(const float *X, const float *w, const float *r0, int64_t n_rows, int64_t d, float *r_out, void *workspace, arl_stream_t stream) {
    const int rc = sfa_stage_args(X, w, n_rows, d, workspace);
    if (rc != ARL_OK) return rc;
    if (!r0 || !r_out) return ARL_E_NULL;
    hipStream_t st = (hipStream_t)stream;
    float *q = (float *)workspace, *s = q + n_rows, *part = s + n_rows, *coef = part + (size_t)kSfaMaxBlocks * kSfaStride;
    const int64_t want = (n_rows + kWavesPerBlock - 1) / kWavesPerBlock;
    const int nblk = (int)(want < kSfaMaxBlocks ? want : kSfaMaxBlocks);
    hipLaunchKernelGGL((sfa_reduce_pass_kernel<1>), dim3(nblk), dim3(kBlock), 0, st, X, w, r0, (int)n_rows, (int)d, q, part);
    ARL_LAUNCH_CHECK();
    hipLaunchKernelGGL(sfa_fold_kernel, dim3((unsigned)d), dim3(kBlock), 0, st, part, nblk, (int)d, r_out, coef + 514);
    ARL_LAUNCH_CHECK();
    return ARL_OK;
}

int arl_sfa_stage2_f32(const float *X, const float *w, const float *r, int64_t n_rows, int64_t d, float *as_out, void *workspace, arl_stream_t stream) {
    const int rc = sfa_stage_args(X, w, n_rows, d, workspace);
    if (rc != ARL_OK) return rc;
    if (!r || !as_out) return ARL_E_NULL;
    hipStream_t st = (hipStream_t)stream;
    float *q = (float *)workspace, *s = q + n_rows, *part = s + n_rows;
    const int64_t want = (n_rows + kWavesPerBlock - 1) / kWavesPerBlock;
    const int nblk = (int)(want < kSfaMaxBlocks ? want : kSfaMaxBlocks);
    hipLaunchKernelGGL((sfa_reduce_pass_kernel<2>), dim3(nblk), dim3(kBlock), 0, st, X, w, r, (int)n_rows, (int)d, s, part);
    ARL_LAUNCH_CHECK();
    hipLaunchKernelGGL(sfa_fold_kernel, dim3((unsigned)d + 1), dim3(kBlock), 0, st, part, nblk, (int)d, as_out, as_out + d);
    ARL_LAUNCH_CHECK();
    return ARL_OK;
}

int arl_sfa_stage3_f32(const float *X, const float *w, const float *r0, const float *r, const float *as, int64_t n_rows, int64_t d, int64_t numel_h,
                       float scale, int32_t accumulate, float *loss_out, float *G, void *workspace, arl_stream_t stream) {
    const int rc = sfa_stage_args(X, w, n_rows, d, workspace);
    if (rc != ARL_OK) return rc;
    if (!r0 || !r || !as || !loss_out) return ARL_E_NULL;
    if (numel_h <= 0) return ARL_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    float *q = (float *)workspace, *s = q + n_rows, *part = s + n_rows, *coef = part + (size_t)kSfaMaxBlocks * kSfaStride;
    hipError_t e = hipMemcpyAsync(coef, r, sizeof(float) * d, hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(coef + 256, as, sizeof(float) * d, hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(coef + 514, as + d, sizeof(float), hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(sfa_finalize_kernel, dim3(1), dim3(kBlock), 0, st, (int)d, (float)(1.0 / (double)numel_h), coef, loss_out);
    ARL_LAUNCH_CHECK();
    if (G) {
        const int64_t want = (n_rows + kWavesPerBlock - 1) / kWavesPerBlock;
        hipLaunchKernelGGL(sfa_grad_kernel, dim3((unsigned)want), dim3(kBlock), 0, st, X, w, r0, q, s, coef, (int)n_rows, (int)d, scale, (int)accumulate, G);
        ARL_LAUNCH_CHECK();
    }
    return ARL_OK;
}

int arl_sddmm_rows_dense_f32(const float *dY, const float *X, int64_t d, const int32_t *rows, int64_t n_rows_sel, int64_t col_off, int64_t n_cols,
                             float *out, arl_stream_t stream) {
    if (!dY || !X || !rows || !out) return ARL_E_NULL;
    if (d <= 0 || d > 256 || n_rows_sel < 0 || n_cols < 0 || col_off < 0) return ARL_E_ARG;
    if (n_cols > 0x7fffffffll || n_rows_sel > 0x7fffffffll) return ARL_E_RANGE;
    if (n_rows_sel == 0 || n_cols == 0) return ARL_OK;
    if (d == 64) {
        const long long waves = (n_cols + 63) / 64;
        hipLaunchKernelGGL(sddmm_rows_dense64_kernel, dim3((unsigned)((waves + kWavesPerBlock - 1) / kWavesPerBlock)), dim3(kBlock), 0, (hipStream_t)stream, dY, X, rows,
                           (int)n_rows_sel, (long long)col_off, (int)n_cols, out);
        ARL_LAUNCH_CHECK();
        return ARL_OK;
    }
    const size_t shm = sizeof(float) * 64 * (size_t)(d + 1);
    if (shm > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)sddmm_rows_dense_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(sddmm_rows_dense_kernel, dim3((unsigned)((n_cols + 63) / 64)), dim3(kBlock), shm, (hipStream_t)stream, dY, X, (int)d, rows,
                       (int)n_rows_sel, (long long)col_off, (int)n_cols, out);
    ARL_LAUNCH_CHECK();
    return ARL_OK;
}

int arl_sddmm_csr_f32(const int32_t *rowptr, const int32_t *col, int64_t n_rows, int64_t d, const float *dY, const float *X, float alpha, float *gval,
                      arl_stream_t stream) {
    if (!rowptr || !col || !dY || !X || !gval) return ARL_E_NULL;
    if (d <= 0 || d > 256 || (d & 3)) return ARL_E_DIM;
    if (n_rows < 0 || n_rows > 0x7fffffffll) return ARL_E_RANGE;
    if (((uintptr_t)dY | (uintptr_t)X) & 15) return ARL_E_ARG;
    if (n_rows == 0) return ARL_OK;
    int lpr = 1;
    while (lpr * 4 < d) lpr <<= 1;
    hipLaunchKernelGGL(sddmm_csr_kernel, dim3(grid_for(n_rows, kWavesPerBlock)), dim3(kBlock), 0, (hipStream_t)stream, rowptr, col, (long long)n_rows, (int)d, lpr, dY, X,
                       alpha, gval);
    ARL_LAUNCH_CHECK();
    return ARL_OK;
}

int arl_pga_update_f32(float *S, const float *grad, const float *dinv_rows, const float *dinv_cols, int64_t rows, int64_t cols, arl_stream_t stream) {
    if (!S || !grad) return ARL_E_NULL;
    if (rows < 0 || cols < 0) return ARL_E_ARG;
    if (rows == 0 || cols == 0) return ARL_OK;
    hipLaunchKernelGGL(pga_update_kernel, dim3(grid_for(rows * cols, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, S, grad, dinv_rows, dinv_cols,
                       (long long)rows, (long long)cols);
    ARL_LAUNCH_CHECK();
    return ARL_OK;
}

int64_t arl_fake_block_rows_workspace_bytes(int64_t F, int64_t I, int64_t d) {
    if (F <= 0 || I <= 0 || d <= 0) return 0;
    const int64_t n_chunks = (I + kFbChunk - 1) / kFbChunk;
    return (int64_t)sizeof(float) * ((F + kFbT - 1) / kFbT) * ((d + 63) / 64) * n_chunks * kFbT * 64;
}

int arl_fake_block_rows_f32(const float *S, int64_t F, int64_t I, const float *X, int64_t d, const float *rscale, float alpha, float *Y, void *workspace,
                            arl_stream_t stream) {
    if (!S || !X || !Y || !workspace) return ARL_E_NULL;
    if (F < 0 || I < 0 || F > 0x7fffffffll || F * d > 0x7fffffffll) return ARL_E_ARG;
    if (d <= 0 || d > 256 || (d & 3)) return ARL_E_DIM;
    if (F == 0 || I == 0) return ARL_OK;
    const int n_chunks = (int)((I + kFbChunk - 1) / kFbChunk);
    const dim3 grid((unsigned)((n_chunks + kWavesPerBlock - 1) / kWavesPerBlock), (unsigned)((d + 63) / 64), (unsigned)((F + kFbT - 1) / kFbT));
    hipLaunchKernelGGL(fake_block_rows_kernel, grid, dim3(kBlock), 0, (hipStream_t)stream, S, (int)F, (long long)I, X, (int)d, n_chunks, (float *)workspace);
    ARL_LAUNCH_CHECK();
    hipLaunchKernelGGL(fake_block_rows_finish_kernel, dim3((unsigned)F, (unsigned)((d + 63) / 64)), dim3(kFbFinishWaves * 64), 0, (hipStream_t)stream, (const float *)workspace, n_chunks,
                       (int)F, (int)d, rscale, alpha, Y);
    ARL_LAUNCH_CHECK();
    return ARL_OK;
}

int arl_fake_block_cols_f32(const float *S, int64_t F, int64_t I, const float *Xf, int64_t d, const float *rscale, float alpha, float *Y, arl_stream_t stream) {
    if (!S || !Xf || !Y) return ARL_E_NULL;
    if (F < 0 || I < 0 || F > 0x7fffffffll) return ARL_E_ARG;
    if (d <= 0 || d > 256 || (d & 3)) return ARL_E_DIM;
    if (F == 0 || I == 0) return ARL_OK;
    const long long n_tiles = (I + kFbT - 1) / kFbT;
    const dim3 grid((unsigned)((n_tiles + kWavesPerBlock - 1) / kWavesPerBlock), (unsigned)((d + 63) / 64), 1);
    hipLaunchKernelGGL(fake_block_cols_kernel, grid, dim3(kBlock), 0, (hipStream_t)stream, S, (int)F, (long long)I, Xf, (int)d, rscale, alpha, Y);
    ARL_LAUNCH_CHECK();
    return ARL_OK;
}

static int64_t cw_align(int64_t b) { return (b + 255) / 256 * 256; }
static int64_t cw_layout(int64_t n_items, int64_t d, int64_t n_real, int64_t T, char *base, CwWs *W) {
    const int64_t ipg = cw_ipg((int)d), G = (n_items + ipg - 1) / ipg, n_ent = n_real * T, max_sl = G + (n_ent + kCwSlice - 1) / kCwSlice + 1;
    int64_t off = 0;
    auto take = [&](int64_t bytes) { char *p = base ? base + off : nullptr; off += cw_align(bytes); return p; };
    char *p_part = take(sizeof(float) * kCwParts * (d + 2)), *p_col = take(sizeof(float) * d), *p_sc = take(sizeof(double) * 2), *p_neg = take(sizeof(int32_t) * n_items);
    char *p_tot = take(sizeof(int32_t) * G), *p_off = take(sizeof(int32_t) * (G + 1)), *p_cur = take(sizeof(int32_t) * G), *p_sl = take(sizeof(int32_t) * 4 * max_sl), *p_ns = take(sizeof(int32_t) * 4);
    char *p_b = take(sizeof(int2) * (n_ent > 0 ? n_ent : 1)), *p_acc = take(sizeof(long long) * n_items * d);
    if (W) {
        W->part = (float *)p_part; W->colsum = (float *)p_col; W->scale = (double *)p_sc; W->neg_cnt = (int32_t *)p_neg; W->grp_tot = (int32_t *)p_tot; W->grp_off = (int32_t *)p_off;
        W->cursor = (int32_t *)p_cur; W->slices = (int32_t *)p_sl; W->n_slices = (int32_t *)p_ns; W->bucket = (int2 *)p_b; W->acc = (long long *)p_acc;
    }
    return off;
}

int64_t arl_cw_topk_term_workspace_bytes(int64_t n_items, int64_t d, int64_t n_real, int64_t n_targets) {
    if (n_items <= 0 || d <= 0 || n_real < 0 || n_targets <= 0) return 0;
    return cw_layout(n_items, d, n_real, n_targets, nullptr, nullptr);
}

int arl_cw_topk_term_f32(const float *X, int64_t n_user_rows, int64_t n_items, int64_t d, int64_t n_real, const int32_t *top_idx, int64_t k,
                         const int64_t *targets, int64_t n_targets, float c, float *G, float *loss, float *w_sfa, void *workspace, arl_stream_t stream) {
    if (!X || !top_idx || !targets || !G || !loss || !workspace) return ARL_E_NULL;
    if (d <= 0 || d > 256 || (d & 3)) return ARL_E_DIM;
    if (n_user_rows <= 0 || n_items <= 0 || n_real < 0 || n_real > n_user_rows || n_targets <= 0 || n_targets > 64 || k < n_targets) return ARL_E_ARG;
    if (n_user_rows + n_items > 0x7fffffffll || n_real * n_targets > 0x7fffffffll || n_items * d > 0x7fffffffffll) return ARL_E_RANGE;
    if ((((uintptr_t)workspace) & 7) || (((uintptr_t)X | (uintptr_t)G) & 15)) return ARL_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    CwWs W;
    cw_layout(n_items, d, n_real, n_targets, (char *)workspace, &W);
    const int ipg = cw_ipg((int)d), n_groups = (int)((n_items + ipg - 1) / ipg);
    if (n_groups > kCwMaxGroups) return ARL_E_RANGE;                  // (1 048 576 items at d <= 128; the group histograms live in LDS)
    const long long n_ent = (long long)n_real * n_targets;
    if (hipMemsetAsync(W.neg_cnt, 0, sizeof(int32_t) * n_items, st) != hipSuccess) return ARL_E_ARG;
    if (hipMemsetAsync(W.grp_tot, 0, sizeof(int32_t) * n_groups, st) != hipSuccess) return ARL_E_ARG;
    if (hipMemsetAsync(W.acc, 0, sizeof(long long) * n_items * d, st) != hipSuccess) return ARL_E_ARG;
    const int rows_per_wg = (int)((n_user_rows + kCwParts - 1) / kCwParts);
    const int n_parts = (int)((n_user_rows + rows_per_wg - 1) / rows_per_wg);
#define ARL_CW_USER(LPRV) hipLaunchKernelGGL((cw_user_kernel<LPRV>), dim3((unsigned)n_parts), dim3(kCwItemThreads), 0, st, X, (int)d, (long long)n_user_rows, (int)n_real, \
                                             top_idx, (int)k, (const long long *)targets, (int)n_targets, c, G, w_sfa, W, rows_per_wg, ipg, n_groups)
    if (d <= 16) ARL_CW_USER(4);
    else if (d <= 32) ARL_CW_USER(8);
    else if (d <= 64) ARL_CW_USER(16);
    else if (d <= 128) ARL_CW_USER(32);
    else ARL_CW_USER(64);
#undef ARL_CW_USER
    ARL_LAUNCH_CHECK();
    int count_log2 = 0;
    while ((1ll << count_log2) < (n_ent > 1 ? n_ent : 1)) ++count_log2;
    hipLaunchKernelGGL(cw_scan_kernel, dim3(1), dim3(kCwItemThreads), 0, st, W, (int)d, n_groups, n_parts, count_log2, loss);
    ARL_LAUNCH_CHECK();
    if (n_ent > 0) {
        hipLaunchKernelGGL(cw_fill_kernel, dim3((unsigned)((n_ent + kCwItemThreads - 1) / kCwItemThreads)), dim3(kCwItemThreads), 0, st, top_idx, (int)k, (int)n_targets, n_ent, ipg, n_groups, W);
        ARL_LAUNCH_CHECK();
        const size_t shm = sizeof(unsigned long long) * (size_t)ipg * (size_t)d;
        hipError_t ea = hipFuncSetAttribute((const void *)cw_item_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        if (ea != hipSuccess) return (int)ea;
        const unsigned max_sl = (unsigned)(n_groups + (n_ent + kCwSlice - 1) / kCwSlice + 1);
        hipLaunchKernelGGL(cw_item_kernel, dim3(max_sl), dim3(kCwItemThreads), shm, st, X, (int)d, (int)n_items, ipg, W);
        ARL_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(cw_finish_kernel, dim3((unsigned)((n_items * d + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, (int)d, (long long)n_user_rows, (int)n_items,
                       (int)n_real, (const long long *)targets, (int)n_targets, c, G, w_sfa, W);
    ARL_LAUNCH_CHECK();
    return ARL_OK;
}

int64_t arl_score_mask_topk_workspace_bytes(int64_t I, int64_t d) { return (I <= 0 || d <= 0) ? 0 : 6 * I * d + 64; }
// byte offset, inside the workspace, of the pass's two 8-byte counters [stream stages consumed summed over workgroups, workgroups] (fp16 split path,
// d = 64 / 128): consumed / (workgroups * ceil(I / stage items)) is the share of the item stream the early exit did not skip
int64_t arl_score_mask_topk_stats_offset(int64_t I, int64_t d) {
    if (I <= 0 || d <= 0) return 0;
    const int64_t mst = d <= 16 ? 128 : (d <= 64 ? 64 : 32), nst = (I + mst - 1) / mst;
    return (4 * I * d + 16 + 4 * I + 4 * (2 * nst + 2) + 7) / 8 * 8;
}

static int t2_env_items(const char *name, int dflt) {     // developer knob: bootstrap sample in items (0 or a multiple of 128, at most 32 768)
    const char *e = getenv(name);
    if (!e) return dflt;
    const int v = atoi(e);
    return (v < 0 || v > 32768 || (v & 127)) ? dflt : v;
}

int64_t arl_score_mask_topk_user_workspace_bytes(int64_t U, int64_t d) { return (U <= 0 || d <= 0) ? 0 : 4 * U * d + 4 * U + 64; }

int arl_score_mask_topk_f32(const float *Pu, const float *Pi, int64_t U, int64_t I, int64_t d, const int32_t *mask_rowptr, const int32_t *mask_col,
                            int64_t k, int32_t *top_idx, float *top_val, void *workspace, const int32_t *warm_idx, int32_t *underflow,
                            const int32_t *item_order, int32_t exit_mode, void *user_workspace, arl_stream_t stream) {
    if (!Pu || !Pi || !top_idx || !top_val) return ARL_E_NULL;
    if (warm_idx && !underflow) return ARL_E_NULL;
    if (mask_rowptr && !mask_col) return ARL_E_NULL;
    if (d <= 0 || d > 256 || (d & 3)) return ARL_E_DIM;
    if (k <= 0 || k > 128 || k > I) return ARL_E_ARG;
    if (U < 0 || I <= 0 || U > 0x7fffffffll || I > 0x7fffffffll) return ARL_E_RANGE;
    if (U == 0) return ARL_OK;
    if (k <= 64 && (d == 16 || d == 32 || d == 64 || d == 128)) {        // matrix-core path
        const bool split = workspace != nullptr && (d == 64 || d == 128);
        const int mst = d <= 16 ? 128 : (d <= 64 ? 64 : 32);
        const size_t stageb = 2 * (size_t)mst * ((split ? 2 * kSplitPlanes : 4) * (size_t)d / 2 + 16);     // two half images of mst rows (STAGEB in the kernel)
        const int nwaves = topk_waves((int)d, split);
        const int users_per_wg = 16 * nwaves;
        const size_t shm_m = kTopkRing * stageb + (2 * kTopkRing + kExitWords) * sizeof(unsigned) + (ARL_TOPK_QUEUE ? sizeof(unsigned) * kQWords * nwaves : 0) +
                             (mask_rowptr ? sizeof(unsigned) * users_per_wg * kBloomWords : 0);
        const unsigned grid_m = (unsigned)((U + users_per_wg - 1) / users_per_wg);
        const void *image = Pi;
        const unsigned *max_bits = nullptr;
        const int32_t *order_d = nullptr;
        int32_t *pos_d = nullptr;
        float *snorm_d = nullptr;
        unsigned long long *stats_d = nullptr;
        int *pick_d = nullptr;
        if (split) {
            const long long n = (long long)I * d;
            if (kSplitMode == 2) {
                // workspace: [I][2][d] fp16 image (4 I d bytes), then the two tables' largest magnitudes (float bits: items, users)
                unsigned *mb = reinterpret_cast<unsigned *>(static_cast<char *>(workspace) + 4 * (size_t)n);
                if (hipMemsetAsync(mb, 0, 2 * sizeof(unsigned), (hipStream_t)stream) != hipSuccess) return ARL_E_ARG;
                hipLaunchKernelGGL(absmax_bits_kernel, dim3(grid_for(n / 4 + 1, kBlock, 2048u)), dim3(kBlock), 0, (hipStream_t)stream, Pi, n, mb);
                ARL_LAUNCH_CHECK();
                hipLaunchKernelGGL(absmax_bits_kernel, dim3(grid_for(U * d / 4 + 1, kBlock, 2048u)), dim3(kBlock), 0, (hipStream_t)stream, Pu, (long long)(U * d), mb + 1);
                ARL_LAUNCH_CHECK();
                if (item_order && ARL_TOPK_QUEUE) {                 // the item stream in the caller's order: inverse map behind the two maxima (the 6 I d byte workspace has room)
                    order_d = item_order;
                    pos_d = reinterpret_cast<int32_t *>(static_cast<char *>(workspace) + 4 * (size_t)n + 16);
                    hipLaunchKernelGGL(invert_perm_kernel, dim3((unsigned)((I + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream, item_order, (int)I, pos_d);
                    ARL_LAUNCH_CHECK();
                }
                hipLaunchKernelGGL(split_f16x2_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream, Pi, n, (int)d, mb,
                                   (_Float16 *)workspace, order_d);
                max_bits = mb;
                {   // per-stage and whole-table largest scaled row norms (the factor of the high-piece stream's bound), behind the inverse map's room
                    const int nst = (int)((I + mst - 1) / mst);
                    snorm_d = reinterpret_cast<float *>(static_cast<char *>(workspace) + 4 * (size_t)n + 16 + 4 * (size_t)I);
                    if (hipMemsetAsync(snorm_d + nst, 0, sizeof(float), (hipStream_t)stream) != hipSuccess) return ARL_E_ARG;
                    hipLaunchKernelGGL(stage_norm_kernel, dim3((unsigned)nst), dim3(kWave), 0, (hipStream_t)stream, Pi, (int)I, (int)d, mst, mb, order_d, nst, snorm_d);
                    ARL_LAUNCH_CHECK();
                    // behind them: the suffix maxima [nst + 1] (the early exit's bound) and two 8-byte counters (stream stages consumed, workgroups)
                    stats_d = reinterpret_cast<unsigned long long *>(static_cast<char *>(workspace) + arl_score_mask_topk_stats_offset(I, d));
                    pick_d = reinterpret_cast<int *>(stats_d + 2);     // [exit build, plain build]
                    hipLaunchKernelGGL(stage_sufmax_kernel, dim3(1), dim3(kWave), 0, (hipStream_t)stream, snorm_d, nst, snorm_d + nst + 1, mst, pick_d);
                    ARL_LAUNCH_CHECK();
                    if (hipMemsetAsync(stats_d, 0, 2 * sizeof(unsigned long long), (hipStream_t)stream) != hipSuccess) return ARL_E_ARG;      // (before the kernels; the two gate words behind them are written by stage_sufmax_kernel)
                }
            } else {
                hipLaunchKernelGGL(split_bf16x3_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream, Pi, n, (int)d,
                                   (__bf16 *)workspace);
            }
            ARL_LAUNCH_CHECK();
            image = workspace;
        }
#define ARL_TOPK_CASE3(DV, SP, WM, XT, WARMP, GATE, GATE2)                                                                              \
        do {                                                                                                                           \
            hipError_t em = hipFuncSetAttribute((const void *)score_mask_topk_mfma16_kernel<DV, SP, WM, XT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm_m); \
            if (em != hipSuccess) return (int)em;                                                                                      \
            hipLaunchKernelGGL((score_mask_topk_mfma16_kernel<DV, SP, WM, XT>), dim3(grid_m), dim3(64 * topk_waves(DV, SP)), shm_m, (hipStream_t)stream, Pu, image, (int)U, (int)I, \
                               mask_rowptr, mask_col, (int)k, top_idx, top_val, Pi, WARMP, underflow, max_bits, order_d, pos_d, GATE, (const float *)snorm_d, stats_d, GATE2);  \
        } while (0)
        /* the fp16 split path launches both builds of the kernel (with / without the early exit); the device runs one (gate2), the other returns at once */
#define ARL_TOPK_CASE2_true(DV, WM, WARMP, GATE)                                                                                       \
        do {                                                                                                                           \
            if (exit_mode != 0) { ARL_TOPK_CASE3(DV, true, WM, true, WARMP, GATE, (const int *)pick_d); ARL_TOPK_CASE3(DV, true, WM, false, WARMP, GATE, (const int *)(pick_d + 1)); } \
            else ARL_TOPK_CASE3(DV, true, WM, false, WARMP, GATE, (const int *)nullptr);                                                \
        } while (0)
#define ARL_TOPK_CASE2_false(DV, WM, WARMP, GATE) ARL_TOPK_CASE3(DV, false, WM, false, WARMP, GATE, (const int *)nullptr)
        /* a warm-started call is followed by its own cold repeat, gated on the underflow flag on the device: valid results without a host round trip */
#define ARL_TOPK_CASE(DV, SP)                                                                                                          \
        do {                                                                                                                           \
            if (warm_idx) { ARL_TOPK_CASE2_##SP(DV, true, warm_idx, (const int *)nullptr); ARL_TOPK_CASE2_##SP(DV, false, (const int32_t *)nullptr, (const int *)underflow); } \
            else ARL_TOPK_CASE2_##SP(DV, false, (const int32_t *)nullptr, (const int *)nullptr);                                        \
        } while (0)
        // second form of the fp16 split stream (topk2_*): needs the caller's user workspace ([U][2][d] fp16 image of the users, then [U] starting thresholds)
        const bool form2 = split && kSplitMode == 2 && ARL_TOPK_REFINE && ARL_TOPK_QUEUE && ARL_TOPK2 && user_workspace != nullptr && (d == 64 || (d == 128 && ARL_TOPK2_D128));
        if (form2) {
            if (((uintptr_t)user_workspace | (uintptr_t)workspace) & 15) return ARL_E_ARG;      // 16-byte fragment loads from both images
            hipStream_t st = (hipStream_t)stream;
            _Float16 *uimg = (_Float16 *)user_workspace;
            float *thr0 = reinterpret_cast<float *>(static_cast<char *>(user_workspace) + 4 * (size_t)U * (size_t)d);
            const long long nu = (long long)U * d;
            hipLaunchKernelGGL(split_f16x2_kernel, dim3((unsigned)((nu + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, Pu, nu, (int)d, max_bits + 1, uimg, (const int32_t *)nullptr);
            ARL_LAUNCH_CHECK();
            const bool boot = I >= 32768;
            // bootstrap sample: a cold pass gains from a large one (fewer candidates, no burst of merges while the sample is re-scored); a warm-started
            // pass already has near-final thresholds and takes the small one as a safety net against stale candidates, and so does the exit build (it runs
            // where most of the stream is skipped: a 16 K sample would be as long as what is left)
            static const int boot_cold = t2_env_items("ARL_TOPK2_BOOT_COLD", kT2BootItems), boot_warm = t2_env_items("ARL_TOPK2_BOOT_WARM", 4096);
            const size_t shm2 = t2_lds_bytes((int)d, mask_rowptr != nullptr);
            const unsigned grid2 = (unsigned)((U + kT2Users - 1) / kT2Users);
#define ARL_TOPK2_MAIN(DV, XT, WARMF, GATE, GATE2)                                                                                      \
            do {                                                                                                                       \
                hipError_t em2 = hipFuncSetAttribute((const void *)topk2_main_kernel<DV, XT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm2); \
                if (em2 != hipSuccess) return (int)em2;                                                                                \
                hipLaunchKernelGGL((topk2_main_kernel<DV, XT>), dim3(grid2), dim3(64 * kT2Waves), shm2, st, (const _Float16 *)uimg, (const _Float16 *)workspace, (int)U, (int)I, mask_rowptr, mask_col, \
                                   (int)k, top_idx, top_val, (WARMF) ? (const float *)thr0 : (const float *)nullptr, underflow, (WARMF) ? 1 : 0, max_bits, order_d, GATE, (const float *)snorm_d, \
                                   stats_d, GATE2, boot ? (((WARMF) || (XT)) ? boot_warm : boot_cold) : 0, (const int32_t *)pos_d);       \
                ARL_LAUNCH_CHECK();                                                                                                    \
            } while (0)
            /* exit_mode = 1: both builds of the kernel (with / without the early exit) are launched, the device runs one (pick_d, see stage_sufmax_kernel) */
#define ARL_TOPK2_PASS(DV, WARMF, GATE)                                                                                                \
            do {                                                                                                                       \
                if (WARMF) {                                                                                                           \
                    hipLaunchKernelGGL(topk2_warm_kernel, dim3(grid_for(U, kWavesPerBlock, 8192u)), dim3(kBlock), 0, st, Pu, Pi, (int)U, (int)I, (int)d, (int)k, warm_idx, max_bits, thr0, GATE, \
                                       (const int *)nullptr);                                                                          \
                    ARL_LAUNCH_CHECK();                                                                                                \
                }                                                                                                                      \
                if (exit_mode != 0) { ARL_TOPK2_MAIN(DV, true, WARMF, GATE, (const int *)pick_d); ARL_TOPK2_MAIN(DV, false, WARMF, GATE, (const int *)(pick_d + 1)); } \
                else ARL_TOPK2_MAIN(DV, false, WARMF, GATE, (const int *)nullptr);                                                     \
            } while (0)
#define ARL_TOPK2_CALL(DV)                                                                                                             \
            do {                                                                                                                       \
                if (warm_idx) { ARL_TOPK2_PASS(DV, true, (const int *)nullptr); ARL_TOPK2_PASS(DV, false, (const int *)underflow); }  \
                else ARL_TOPK2_PASS(DV, false, (const int *)nullptr);                                                                  \
            } while (0)
            if (d == 64) ARL_TOPK2_CALL(64); else ARL_TOPK2_CALL(128);
#undef ARL_TOPK2_CALL
#undef ARL_TOPK2_PASS
#undef ARL_TOPK2_MAIN
        }
        else if (d == 16) ARL_TOPK_CASE(16, false);
        else if (d == 32) ARL_TOPK_CASE(32, false);
        else if (d == 64) { if (split) ARL_TOPK_CASE(64, true); else ARL_TOPK_CASE(64, false); }
        else { if (split) ARL_TOPK_CASE(128, true); else ARL_TOPK_CASE(128, false); }
#undef ARL_TOPK_CASE2_true
#undef ARL_TOPK_CASE2_false
#undef ARL_TOPK_CASE3
#undef ARL_TOPK_CASE
        ARL_LAUNCH_CHECK();
        return ARL_OK;
    }
    const size_t shm = sizeof(unsigned long long) * kTU * kCap + sizeof(float) * kTU * 256 + sizeof(unsigned long long) * kTU + sizeof(int) * kTU;
    hipError_t e = hipFuncSetAttribute((const void *)score_mask_topk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(score_mask_topk_kernel, dim3((unsigned)((U + kTU - 1) / kTU)), dim3(kBlock), shm, (hipStream_t)stream, Pu, Pi, (int)U, (int)I,
                       (int)d, mask_rowptr, mask_col, (int)k, top_idx, top_val);
    ARL_LAUNCH_CHECK();
    return ARL_OK;
}

int arl_topn_project_rows_f32(const float *M, int64_t rows, int64_t cols, int64_t n, float *out, int32_t *idx, float *scratch, arl_stream_t stream) {
    if (!M || !out || !scratch) return ARL_E_NULL;
    if (rows < 0 || cols <= 0 || n < 0 || n > cols || cols > 0x7fffffffll || rows > 0x7fffffffll) return ARL_E_ARG;
    if (rows == 0) return ARL_OK;
    hipLaunchKernelGGL(topn_project_kernel, dim3((unsigned)rows), dim3(kBlock), 0, (hipStream_t)stream, M, (int)cols, (int)n, out, idx, scratch);
    ARL_LAUNCH_CHECK();
    return ARL_OK;
}

}  // extern "C"
