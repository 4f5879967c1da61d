// arl_comm.hip -- the item-table exchange of the user-sharded step behind the C ABI: a sum-all-reduce of the replicated
// I x d block as a DIRECT reduce-scatter + all-gather over the point-to-point xGMI links of one node.
//
// The reference has no multi-GPU path (main.py:19 pins one device); this is new design.  A ring all-reduce sends 2(P-1)/P of the buffer
// through every link in 2(P-1) dependent steps; on a fully connected xGMI mesh (7 links per GPU) every rank can instead send shard q of its
// partial straight to rank q on its own link (one step), each rank sums the P partials of ITS shard in rank order (deterministic, and the
// owner is the only writer, so all replicas receive the same bits), and sends the result straight to every peer (one step): 2 steps of
// (P-1)/P of the buffer spread over P-1 links.  The buffer is cut into chunks: the receive workspace is (P-1) slots of one chunk, and the
// caller's compute stream (the exchange runs on its own stream) meets shorter communication kernels to interleave with.
//
// RCCL is bound at run time (dlopen of the librccl the process already has -- PyTorch's -- or the system one): libarlib_amd.so itself has no
// link dependency on it, a single-GPU box without RCCL still loads the library, and arl_comm_* fail with ARL_E_ARG there.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdint.h>
#include <string.h>
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
// A box without the RCCL headers still builds the library (the functions are bound by dlopen at run time): the few declarations used here, with the
// values of nccl.h (ncclFloat32 = 7, ncclSum = 0, a 128-byte unique id).
extern "C" {
typedef struct ncclComm *ncclComm_t;
typedef enum { ncclSuccess = 0 } ncclResult_t;
#define NCCL_UNIQUE_ID_BYTES 128
typedef struct { char internal[NCCL_UNIQUE_ID_BYTES]; } ncclUniqueId;
typedef enum { ncclFloat = 7 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
}
#endif
#include "arlib_amd.h"

namespace {

struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
};
RcclApi g_rccl;

bool load_rccl(const char *path) {
    if (g_rccl.handle) return true;
    void *h = nullptr;
    if (path && *path) h = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
    const char *names[] = {"librccl.so.1", "librccl.so"};
    for (int pass = 0; pass < 2 && !h; ++pass)                     // first an instance the process has already mapped, then a fresh load
        for (const char *n : names) {
            h = dlopen(n, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
            if (h) break;
        }
    if (!h) return false;
    RcclApi a;
    a.handle = h;
#define ARL_SYM(F) a.F = (decltype(a.F))dlsym(h, "nccl" #F); if (!a.F) { dlclose(h); return false; }
    ARL_SYM(GetUniqueId) ARL_SYM(CommInitRank) ARL_SYM(CommDestroy) ARL_SYM(Send) ARL_SYM(Recv) ARL_SYM(GroupStart) ARL_SYM(GroupEnd) ARL_SYM(AllReduce)
#undef ARL_SYM
    g_rccl = a;
    return true;
}

struct Comm {
    ncclComm_t nc;
    int rank, world;
};

// shard q of an n-element buffer over `world` ranks: [lo, hi), sizes differing by at most one 4-element group (16-byte aligned shard starts)
inline void shard_range(int64_t n, int world, int q, int64_t *lo, int64_t *hi) {
    const int64_t g = (n + 3) / 4;                                 // 4-element groups
    const int64_t a = (g * q) / world * 4, b = (g * (q + 1)) / world * 4;
    *lo = a < n ? a : n;
    *hi = b < n ? b : n;
}
// chunk c of `chunks` of a shard [lo, hi)
inline void chunk_range(int64_t lo, int64_t hi, int chunks, int c, int64_t *clo, int64_t *chi) {
    const int64_t len = hi - lo, g = (len + 3) / 4;
    const int64_t a = lo + (g * c) / chunks * 4, b = lo + (g * (c + 1)) / chunks * 4;
    *clo = a < hi ? a : hi;
    *chi = b < hi ? b : hi;
}

// out[i] = sum over ranks r = 0..world-1 (in that order) of part_r[i]; part_me = out itself (the rank's own partial), the others = tmp slots
__global__ __launch_bounds__(256) void shard_sum_kernel(float *__restrict__ own, const float *__restrict__ tmp, long long slot_stride, int me, int world,
                                                        long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float acc = 0.f;
        for (int r = 0; r < world; ++r) acc += (r == me) ? own[i] : tmp[(long long)(r < me ? r : r - 1) * slot_stride + i];
        own[i] = acc;
    }
}

}  // namespace

extern "C" {

int arl_comm_load(const char *rccl_path) { return load_rccl(rccl_path) ? ARL_OK : ARL_E_ARG; }

int arl_comm_unique_id(void *id128) {
    if (!id128) return ARL_E_NULL;
    if (!load_rccl(nullptr)) return ARL_E_ARG;
    ncclUniqueId id;
    const ncclResult_t r = g_rccl.GetUniqueId(&id);
    if (r != ncclSuccess) return 1000 + (int)r;
    memcpy(id128, id.internal, NCCL_UNIQUE_ID_BYTES);
    return ARL_OK;
}

int arl_comm_init(const void *id128, int64_t rank, int64_t world, int64_t device, arl_comm_t *out) {
    if (!id128 || !out) return ARL_E_NULL;
    if (world < 1 || rank < 0 || rank >= world) return ARL_E_ARG;
    if (!load_rccl(nullptr)) return ARL_E_ARG;
    ncclUniqueId id;
    memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
    // the communicator binds to the CURRENT device of the calling thread: make it the caller's device for the call (device < 0: leave it alone)
    int prev = -1;
    if (device >= 0) {
        if (hipGetDevice(&prev) != hipSuccess) return ARL_E_ARG;
        const hipError_t e = hipSetDevice((int)device);
        if (e != hipSuccess) return (int)e;
    }
    Comm *c = new Comm{nullptr, (int)rank, (int)world};
    const ncclResult_t r = g_rccl.CommInitRank(&c->nc, (int)world, id, (int)rank);
    if (device >= 0 && prev >= 0 && prev != (int)device) (void)hipSetDevice(prev);
    if (r != ncclSuccess) { delete c; return 1000 + (int)r; }
    *out = (arl_comm_t)c;
    return ARL_OK;
}

int arl_comm_destroy(arl_comm_t comm) {
    if (!comm) return ARL_E_NULL;
    Comm *c = (Comm *)comm;
    if (g_rccl.handle && c->nc) g_rccl.CommDestroy(c->nc);
    delete c;
    return ARL_OK;
}

/* The partition the exchange uses, exposed for the host-side tests: range [lo, hi) of chunk c of shard q.  Pure arithmetic, no GPU. */
int arl_item_exchange_range(int64_t n_elems, int64_t world, int64_t n_chunks, int64_t shard, int64_t chunk, int64_t *lo, int64_t *hi) {
    if (!lo || !hi) return ARL_E_NULL;
    if (n_elems < 0 || world < 1 || n_chunks < 1 || shard < 0 || shard >= world || chunk < 0 || chunk >= n_chunks) return ARL_E_ARG;
    int64_t a, b;
    shard_range(n_elems, (int)world, (int)shard, &a, &b);
    chunk_range(a, b, (int)n_chunks, (int)chunk, lo, hi);
    return ARL_OK;
}

int64_t arl_allreduce_item_workspace_bytes(int64_t n_elems, int64_t world, int64_t n_chunks) {
    if (n_elems < 0 || world < 1 || n_chunks < 1) return 0;
    // (world - 1) receive slots of one chunk each (the largest chunk of the largest shard), double-buffered over chunk parity
    const int64_t shard = ((n_elems + 3) / 4 + world - 1) / world * 4;
    const int64_t chunk = ((shard + 3) / 4 + n_chunks - 1) / n_chunks * 4 + 4;
    return (int64_t)sizeof(float) * 2 * (world - 1) * chunk;
}

int arl_allreduce_item_f32(arl_comm_t comm, float *buf, int64_t n_elems, int64_t n_chunks, void *workspace, arl_stream_t stream) {
    if (!comm || !buf) return ARL_E_NULL;
    if (n_elems < 0 || n_chunks < 1 || n_chunks > 64 || n_elems > 0x7fffffffffll) return ARL_E_ARG;
    Comm *c = (Comm *)comm;
    const int P = c->world, me = c->rank;
    if (P == 1 || n_elems == 0) return ARL_OK;
    if (!workspace) return ARL_E_NULL;
    if (((uintptr_t)buf | (uintptr_t)workspace) & 15) return ARL_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int64_t shard_max = ((n_elems + 3) / 4 + P - 1) / P * 4;
    const int64_t slot = ((shard_max + 3) / 4 + n_chunks - 1) / n_chunks * 4 + 4;          // floats per receive slot
    float *ws = (float *)workspace;
    int64_t mlo, mhi;
    shard_range(n_elems, P, me, &mlo, &mhi);
#define ARL_NCCL(CALL) do { const ncclResult_t r__ = (CALL); if (r__ != ncclSuccess) return 1000 + (int)r__; } while (0)
    // inside a GroupStart / GroupEnd pair: close the group before leaving, an open group would swallow every later RCCL call of the thread
#define ARL_NCCL_G(CALL) do { const ncclResult_t r__ = (CALL); if (r__ != ncclSuccess) { (void)g_rccl.GroupEnd(); return 1000 + (int)r__; } } while (0)
    for (int ch = 0; ch < (int)n_chunks; ++ch) {
        float *tmp = ws + (int64_t)(ch & 1) * (P - 1) * slot;
        int64_t clo, chi;
        chunk_range(mlo, mhi, (int)n_chunks, ch, &clo, &chi);
        // reduce-scatter, direct: chunk ch of shard q of my partial goes to rank q; the peers' pieces of MY shard arrive in the slots
        ARL_NCCL(g_rccl.GroupStart());
        for (int q = 0; q < P; ++q) {
            if (q == me) continue;
            int64_t qlo, qhi, a, b;
            shard_range(n_elems, P, q, &qlo, &qhi);
            chunk_range(qlo, qhi, (int)n_chunks, ch, &a, &b);
            if (b > a) ARL_NCCL_G(g_rccl.Send(buf + a, (size_t)(b - a), ncclFloat, q, c->nc, st));
            if (chi > clo) ARL_NCCL_G(g_rccl.Recv(tmp + (int64_t)(q < me ? q : q - 1) * slot, (size_t)(chi - clo), ncclFloat, q, c->nc, st));
        }
        ARL_NCCL(g_rccl.GroupEnd());
        if (chi > clo) {
            const long long n = chi - clo;
            const unsigned grid = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
            hipLaunchKernelGGL(shard_sum_kernel, dim3(grid), dim3(256), 0, st, buf + clo, tmp, (long long)slot, me, P, n);
            const hipError_t e = hipGetLastError();
            if (e != hipSuccess) return (int)e;
        }
        // all-gather, direct: my reduced chunk goes to every peer, theirs land in place
        ARL_NCCL(g_rccl.GroupStart());
        for (int q = 0; q < P; ++q) {
            if (q == me) continue;
            int64_t qlo, qhi, a, b;
            shard_range(n_elems, P, q, &qlo, &qhi);
            chunk_range(qlo, qhi, (int)n_chunks, ch, &a, &b);
            if (chi > clo) ARL_NCCL_G(g_rccl.Send(buf + clo, (size_t)(chi - clo), ncclFloat, q, c->nc, st));
            if (b > a) ARL_NCCL_G(g_rccl.Recv(buf + a, (size_t)(b - a), ncclFloat, q, c->nc, st));
        }
        ARL_NCCL(g_rccl.GroupEnd());
    }
#undef ARL_NCCL
#undef ARL_NCCL_G
    return ARL_OK;
}

}  // extern "C"
