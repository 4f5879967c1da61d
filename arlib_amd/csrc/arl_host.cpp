// arl_host.cpp -- host side of libarlib_amd.so: the bit-exact BPR pair sampler.
//
// Replaces util/sampler.py:4-30 (next_batch_pairwise).  The reference draws from CPython's global
// `random` (MT19937, seeded by util/tool.py:101-108).  To stay a drop-in, the caller hands us
// random.getstate()[1] (624 words + index), we advance it exactly as CPython would, and the caller
// puts it back with random.setstate().  Sequential by nature (rejection loops consume a
// data-dependent number of draws), so this lives on the host and feeds the GPU through pinned
// buffers; it costs ~10 ns per draw versus ~2.5 us per sample in the reference's Python loop.
#include <cstdint>
#include <cstddef>
#include <queue>
#include <cmath>
#include <algorithm>
#include <utility>
#include <vector>
#include "arlib_amd.h"

namespace {

// View over a CPython MT19937 state vector (624 words followed by the read position).
class PyMersenne {
public:
    explicit PyMersenne(uint32_t *state) : w_(state), pos_(state[kN]) {}
    ~PyMersenne() { w_[kN] = pos_; }

    uint32_t next32() {
        if (pos_ >= kN) refill();
        uint32_t y = w_[pos_++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        return y ^ (y >> 18);
    }

    // Random._randbelow_with_getrandbits for n < 2^32: draw bit_length(n) top bits until < n.
    uint32_t below(uint32_t n) {
        if (n == 0) return 0;
        const int shift = __builtin_clz(n);          // 32 - bit_length(n)
        uint32_t r;
        do { r = next32() >> shift; } while (r >= n);
        return r;
    }

    static void seed_by_array(uint32_t *w, const uint32_t *key, int64_t len) {
        w[0] = 19650218u;
        for (uint32_t i = 1; i < kN; ++i) w[i] = 1812433253u * (w[i - 1] ^ (w[i - 1] >> 30)) + i;
        uint32_t i = 1;
        int64_t j = 0;
        for (int64_t k = (len > (int64_t)kN ? len : (int64_t)kN); k > 0; --k) {
            w[i] = (w[i] ^ ((w[i - 1] ^ (w[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
            if (++i >= kN) { w[0] = w[kN - 1]; i = 1; }
            if (++j >= len) j = 0;
        }
        for (uint32_t k = kN - 1; k > 0; --k) {
            w[i] = (w[i] ^ ((w[i - 1] ^ (w[i - 1] >> 30)) * 1566083941u)) - i;
            if (++i >= kN) { w[0] = w[kN - 1]; i = 1; }
        }
        w[0] = 0x80000000u;
        w[kN] = kN;
    }

private:
    static constexpr uint32_t kN = 624, kM = 397;
    static uint32_t twist(uint32_t hi, uint32_t lo, uint32_t far) {
        const uint32_t y = (hi & 0x80000000u) | (lo & 0x7fffffffu);
        return far ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    void refill() {
        for (uint32_t k = 0; k < kN - kM; ++k) w_[k] = twist(w_[k], w_[k + 1], w_[k + kM]);
        for (uint32_t k = kN - kM; k < kN - 1; ++k) w_[k] = twist(w_[k], w_[k + 1], w_[k + kM - kN]);
        w_[kN - 1] = twist(w_[kN - 1], w_[0], w_[kM - 1]);
        pos_ = 0;
    }
    uint32_t *w_;
    uint32_t pos_;
};

// `item in training_set_u[user]` on the CSR image of the dict-of-dicts (sorted item ids per user).
inline bool user_has_item(const int64_t *rowptr, const int32_t *items, int64_t rows, int32_t user, int32_t item) {
    if (user < 0 || user >= rows) return false;
    const int32_t *lo = items + rowptr[user], *hi = items + rowptr[user + 1];
    while (lo < hi) {
        const int32_t *mid = lo + (hi - lo) / 2;
        if (*mid < item) lo = mid + 1; else hi = mid;
    }
    return lo < items + rowptr[user + 1] && *lo == item;
}

}  // namespace

extern "C" {

int arl_abi_version(void) { return 24; }

int arl_mt_seed(uint32_t *mt_state, const uint32_t *key, int64_t key_len) {
    if (!mt_state || !key) return ARL_E_NULL;
    if (key_len < 1) return ARL_E_ARG;
    PyMersenne::seed_by_array(mt_state, key, key_len);
    return ARL_OK;
}

int arl_sampler_shuffle(uint32_t *mt_state, int32_t *pairs, int64_t nnz) {
    if (!mt_state || (!pairs && nnz > 0)) return ARL_E_NULL;
    if (nnz < 0 || nnz > 0xFFFFFFFFll) return ARL_E_RANGE;
    if (mt_state[624] > 624) return ARL_E_ARG;
    PyMersenne rng(mt_state);
    // random.shuffle: for i in reversed(range(1, n)): j = randbelow(i + 1); x[i], x[j] = x[j], x[i]
    int64_t *rows = reinterpret_cast<int64_t *>(pairs);      // one (user,item) pair = 8 bytes
    const bool aligned = (reinterpret_cast<uintptr_t>(pairs) & 7u) == 0;
    // The partner indices depend on the generator alone, not on the swaps: they are drawn a block ahead and their rows prefetched, then the swaps run in
    // the reference's order.  (32 M pairs = 256 MB: every swap partner is a cache miss; 0.45 -> ~0.2 s per cfg2 epoch, the serial part of device_epoch.)
    constexpr int kAhead = 512;
    int64_t js[kAhead];
    for (int64_t i = nnz - 1; i >= 1;) {
        const int cnt = (int)(i < kAhead ? i : kAhead);               // swaps i, i - 1, ..., i - cnt + 1 (all >= 1)
        for (int t = 0; t < cnt; ++t) {
            js[t] = rng.below((uint32_t)(i - t + 1));
            __builtin_prefetch(pairs + 2 * js[t], 1, 0);
        }
        if (aligned) {
            for (int t = 0; t < cnt; ++t) { const int64_t a = i - t, j = js[t], v = rows[a]; rows[a] = rows[j]; rows[j] = v; }
        } else {
            for (int t = 0; t < cnt; ++t)
                for (int c = 0; c < 2; ++c) { const int64_t a = i - t, j = js[t]; const int32_t v = pairs[2 * a + c]; pairs[2 * a + c] = pairs[2 * j + c]; pairs[2 * j + c] = v; }
        }
        i -= cnt;
    }
    return ARL_OK;
}

int arl_mt_sample_range(uint32_t *mt_state, int64_t n, int64_t k, int32_t use_pool, int32_t *out, int32_t *scratch) {
    // random.sample(range(n), k) of CPython 3.10 (Lib/random.py:sample): the pool form when the caller's set-size rule says so
    // (scratch: n int32), rejection against the already selected values otherwise (scratch: (n + 31) / 32 words, zeroed here).
    if (!mt_state || (!out && k > 0) || !scratch) return ARL_E_NULL;
    if (n < 0 || k < 0 || k > n || n > 0x7FFFFFFFll) return ARL_E_ARG;
    if (mt_state[624] > 624) return ARL_E_ARG;
    PyMersenne rng(mt_state);
    if (use_pool) {
        for (int64_t i = 0; i < n; ++i) scratch[i] = (int32_t)i;
        for (int64_t i = 0; i < k; ++i) {
            const uint32_t j = rng.below((uint32_t)(n - i));
            out[i] = scratch[j];
            scratch[j] = scratch[n - i - 1];
        }
    } else {
        uint32_t *seen = reinterpret_cast<uint32_t *>(scratch);
        for (int64_t w = 0; w < (n + 31) / 32; ++w) seen[w] = 0u;
        for (int64_t i = 0; i < k; ++i) {
            uint32_t j = rng.below((uint32_t)n);
            while ((seen[j >> 5] >> (j & 31)) & 1u) j = rng.below((uint32_t)n);
            seen[j >> 5] |= 1u << (j & 31);
            out[i] = (int32_t)j;
        }
    }
    return ARL_OK;
}

int arl_sampler_next_batch(uint32_t *mt_state, const int32_t *pairs, int64_t begin, int64_t count, int32_t n_items,
                           const int64_t *memb_rowptr, const int32_t *memb_items, int64_t memb_rows,
                           int32_t *out_u, int32_t *out_p, int32_t *out_n) {
    if (!mt_state || !pairs || !out_u || !out_p || !out_n) return ARL_E_NULL;
    if (memb_rows > 0 && (!memb_rowptr || !memb_items)) return ARL_E_NULL;
    if (begin < 0 || count < 0 || n_items <= 0) return ARL_E_ARG;
    if (mt_state[624] > 624) return ARL_E_ARG;
    PyMersenne rng(mt_state);
    for (int64_t b = 0; b < count; ++b) {
        const int32_t user = pairs[2 * (begin + b)], pos = pairs[2 * (begin + b) + 1];
        int32_t neg;
        do { neg = (int32_t)rng.below((uint32_t)n_items); }
        while (user_has_item(memb_rowptr, memb_items, memb_rows, user, neg));
        out_u[b] = user; out_p[b] = pos; out_n[b] = neg;
    }
    return ARL_OK;
}

// Plan helper of the register-blocked SpMM: longest-processing-time dealing.  Rows arrive sorted by weight (edge count),
// heaviest first; each goes to the least loaded bin (wave) that still has a free slot; ties go to the lowest bin id, so the
// plan is reproducible.  n_bins * cap >= n is required.
int arl_lpt_deal(int64_t n, const int32_t *weight_desc, int64_t n_bins, int64_t cap, int32_t *bin_out, int32_t *slot_out) {
    if (n < 0 || n_bins < 0 || cap < 1 || n_bins > 0x7fffffffll || n_bins * cap < n) return ARL_E_ARG;
    if (n == 0) return ARL_OK;
    if (!weight_desc || !bin_out || !slot_out) return ARL_E_NULL;
    typedef std::pair<int64_t, int32_t> Load;                      // (edges so far, bin)
    std::priority_queue<Load, std::vector<Load>, std::greater<Load>> open;
    std::vector<int32_t> fill((size_t)n_bins, 0);
    for (int64_t b = 0; b < n_bins; ++b) open.push(Load(0, (int32_t)b));
    for (int64_t r = 0; r < n; ++r) {
        if (weight_desc[r] < 0 || (r > 0 && weight_desc[r] > weight_desc[r - 1])) return ARL_E_ARG;
        const Load top = open.top();
        open.pop();
        bin_out[r] = top.second;
        slot_out[r] = fill[top.second]++;
        if (fill[top.second] < cap) open.push(Load(top.first + weight_desc[r], top.second));
    }
    return ARL_OK;
}

// ---- SYN-v1: the synthetic interaction graphs of the benchmark, generated natively.  Every random number is a counter-based hash
//   h(stream, index) = splitmix64(splitmix64(seed ^ stream * PHI) ^ index)
// so this generator and the numpy one (arlib_amd/util/synthetic.py) emit the same pair list, checked by arl_graph_digest.
static inline uint64_t syn_splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static inline uint64_t syn_hash(uint64_t base, uint64_t index) { return syn_splitmix64(index ^ base); }
static inline uint64_t syn_base(uint64_t seed, uint64_t stream) { return syn_splitmix64(seed ^ (stream * 0x9E3779B97F4A7C15ull)); }
static inline double syn_uniform(uint64_t h) { return ((double)(h >> 11) + 0.5) * (1.0 / 9007199254740992.0); }

/* Writes the user-major sorted, de-duplicated (user, item) pairs into pairs_out [capacity][2] and returns their number (or the number
 * needed, without writing, when pairs_out is NULL or capacity is too small); negative = argument error. */
int64_t arl_syn_v1_pairs(int64_t n_users, int64_t n_items, double mean_deg, uint64_t seed, double sigma, int64_t deg_min, int64_t deg_max,
                         int32_t *pairs_out, int64_t capacity) {
    if (n_users <= 0 || n_items <= 0 || n_users > 0x7fffffffll || n_items > 0x7fffffffll || !(mean_deg > 0) || !(sigma >= 0) || deg_min < 0 || deg_max < deg_min)
        return ARL_E_ARG;
    const uint64_t b1 = syn_base(seed, 1), b2 = syn_base(seed, 2), b3 = syn_base(seed, 3), b4 = syn_base(seed, 4), b5 = syn_base(seed, 5);
    const double mu = std::log(mean_deg) - 0.5 * sigma * sigma;
    const double two_pi = 2.0 * 3.14159265358979323846;
    const int64_t hi = deg_max < n_items ? deg_max : n_items;
    std::vector<int64_t> deg((size_t)n_users);
    int64_t total = 0;
    for (int64_t u = 0; u < n_users; ++u) {
        const double u1 = syn_uniform(syn_hash(b1, (uint64_t)u)), u2 = syn_uniform(syn_hash(b2, (uint64_t)u));
        const double z = std::sqrt(-2.0 * std::log(u1)) * std::cos(two_pi * u2);
        double dg = std::nearbyint(std::exp(mu + sigma * z));          // numpy rint: round half to even (default FP environment)
        if (dg < (double)deg_min) dg = (double)deg_min;
        if (dg > (double)hi) dg = (double)hi;
        deg[(size_t)u] = (int64_t)dg;
        total += deg[(size_t)u];
    }
    // pi = stable argsort of h(4, j)
    std::vector<std::pair<uint64_t, int64_t>> hp((size_t)n_items);
    for (int64_t j = 0; j < n_items; ++j) hp[(size_t)j] = std::make_pair(syn_hash(b4, (uint64_t)j), j);
    std::stable_sort(hp.begin(), hp.end(), [](const std::pair<uint64_t, int64_t> &a, const std::pair<uint64_t, int64_t> &b) { return a.first < b.first; });
    std::vector<int64_t> keys;
    keys.reserve((size_t)(total + n_items));
    int64_t k = 0;
    for (int64_t u = 0; u < n_users; ++u)
        for (int64_t t = 0; t < deg[(size_t)u]; ++t, ++k) {
            const double r = syn_uniform(syn_hash(b3, (uint64_t)k));
            int64_t pos = (int64_t)((double)n_items * r * r);
            if (pos > n_items - 1) pos = n_items - 1;
            keys.push_back(u * n_items + hp[(size_t)pos].second);
        }
    for (int64_t j = 0; j < n_items; ++j) keys.push_back((int64_t)(syn_hash(b5, (uint64_t)j) % (uint64_t)n_users) * n_items + j);      // no isolated item
    std::sort(keys.begin(), keys.end());
    keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
    const int64_t nnz = (int64_t)keys.size();
    if (!pairs_out || capacity < nnz) return nnz;
    for (int64_t e = 0; e < nnz; ++e) {
        pairs_out[2 * e] = (int32_t)(keys[(size_t)e] / n_items);
        pairs_out[2 * e + 1] = (int32_t)(keys[(size_t)e] % n_items);
    }
    return nnz;
}

/* Order-sensitive 64-bit digest of a pair list (the numpy twin is synthetic.graph_digest). */
uint64_t arl_graph_digest(const int32_t *pairs, int64_t n) {
    uint64_t acc = 0;
    if (pairs)
        for (int64_t e = 0; e < n; ++e) {
            const uint64_t a = (uint64_t)(int64_t)pairs[2 * e], b = (uint64_t)(int64_t)pairs[2 * e + 1];
            acc ^= syn_splitmix64((a * 0x100000001B3ull) ^ syn_splitmix64(b) ^ (uint64_t)e);
        }
    return acc ^ (uint64_t)(n > 0 ? n : 0);
}

}  // extern "C"
