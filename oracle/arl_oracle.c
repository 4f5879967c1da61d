/*
 * arl_oracle.c -- CPU restatement ("oracle") of the ARLib hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under arlib_amd/ may import, link or call this file.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only as the
 * checker / reported CPU baseline -- never as the thing shipped or measured as the product.
 *
 * Parity pin: every function here is checked against golden vectors captured from the reference
 * itself (tests/golden/gen_golden.py imports /root/reference on CPU) in tests/test_oracle_golden.py.
 *
 * Each function cites the reference file:line (relative to /root/reference) it restates.
 * Accumulations inside reductions are done in double and rounded to fp32 once, so the oracle sits
 * at least as close to exact arithmetic as either the reference (torch CPU fp32) or the HIP path;
 * tensors between ops are fp32 exactly like the reference's.
 */
#include <stdint.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <math.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------
 * CPython `random` (MT19937).  Reference contract: util/tool.py:101-108 seedSet() -> random.seed(seed);
 * util/sampler.py:9 shuffle(), :24-28 choice().  The generator itself is CPython's _randommodule.c
 * (third-party to ARLib, absent from /root/reference): Matsumoto/Nishimura MT19937 with
 * init_by_array; random.seed(int) feeds the 32-bit little-endian words of abs(seed).
 * State layout = random.getstate()[1]: 624 words + index.
 * ------------------------------------------------------------------------------------------ */
#define MT_N 624
#define MT_M 397

static void mt_init_genrand(uint32_t *mt, uint32_t s)
{
    mt[0] = s;
    for (int i = 1; i < MT_N; i++)
        mt[i] = 1812433253U * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
    mt[MT_N] = MT_N;
}

void orc_mt_seed_by_array(uint32_t *st, const uint32_t *key, int64_t len)
{
    uint32_t *mt = st;
    mt_init_genrand(mt, 19650218U);
    int64_t i = 1, j = 0, k = (MT_N > len ? MT_N : len);
    for (; k; k--) {
        mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525U)) + key[j] + (uint32_t)j;
        i++; j++;
        if (i >= MT_N) { mt[0] = mt[MT_N - 1]; i = 1; }
        if (j >= len) j = 0;
    }
    for (k = MT_N - 1; k; k--) {
        mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941U)) - (uint32_t)i;
        i++;
        if (i >= MT_N) { mt[0] = mt[MT_N - 1]; i = 1; }
    }
    mt[0] = 0x80000000U;
    st[MT_N] = MT_N;
}

static inline uint32_t mt_genrand(uint32_t *st)
{
    static const uint32_t mag01[2] = {0x0U, 0x9908b0dfU};
    uint32_t *mt = st, y;
    if (st[MT_N] >= MT_N) {
        int kk;
        for (kk = 0; kk < MT_N - MT_M; kk++) {
            y = (mt[kk] & 0x80000000U) | (mt[kk + 1] & 0x7fffffffU);
            mt[kk] = mt[kk + MT_M] ^ (y >> 1) ^ mag01[y & 1U];
        }
        for (; kk < MT_N - 1; kk++) {
            y = (mt[kk] & 0x80000000U) | (mt[kk + 1] & 0x7fffffffU);
            mt[kk] = mt[kk + (MT_M - MT_N)] ^ (y >> 1) ^ mag01[y & 1U];
        }
        y = (mt[MT_N - 1] & 0x80000000U) | (mt[0] & 0x7fffffffU);
        mt[MT_N - 1] = mt[MT_M - 1] ^ (y >> 1) ^ mag01[y & 1U];
        st[MT_N] = 0;
    }
    y = mt[st[MT_N]++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680U;
    y ^= (y << 15) & 0xefc60000U;
    y ^= (y >> 18);
    return y;
}

/* random.random(): 53-bit double */
double orc_mt_random(uint32_t *st)
{
    uint32_t a = mt_genrand(st) >> 5, b = mt_genrand(st) >> 6;
    return (a * 67108864.0 + b) * (1.0 / 9007199254740992.0);
}

/* Random._randbelow_with_getrandbits(n), n < 2^32: k=n.bit_length(); r=getrandbits(k) until r<n */
static inline uint32_t mt_randbelow(uint32_t *st, uint32_t n)
{
    if (!n) return 0;
    int k = 32 - __builtin_clz(n);
    uint32_t r = mt_genrand(st) >> (32 - k);
    while (r >= n) r = mt_genrand(st) >> (32 - k);
    return r;
}

uint32_t orc_mt_randbelow(uint32_t *st, uint32_t n) { return mt_randbelow(st, n); }

/* util/sampler.py:9  shuffle(training_data): for i in reversed(range(1,n)): j=randbelow(i+1); swap */
void orc_shuffle_pairs(uint32_t *st, int32_t *pairs, int64_t nnz)
{
    for (int64_t i = nnz - 1; i >= 1; i--) {
        int64_t j = mt_randbelow(st, (uint32_t)(i + 1));
        int32_t a = pairs[2 * i], b = pairs[2 * i + 1];
        pairs[2 * i] = pairs[2 * j]; pairs[2 * i + 1] = pairs[2 * j + 1];
        pairs[2 * j] = a; pairs[2 * j + 1] = b;
    }
}

static inline int memb_contains(const int64_t *rowptr, const int32_t *items, int64_t n_rows, int32_t u, int32_t it)
{
    if (u >= n_rows) return 0;           /* users appended after DataLoader init have an empty training_set_u */
    int64_t lo = rowptr[u], hi = rowptr[u + 1];
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (items[mid] < it) lo = mid + 1; else hi = mid;
    }
    return lo < rowptr[u + 1] && items[lo] == it;
}

/* util/sampler.py:20-29: per sample, neg = choice(item_list) until neg not in training_set_u[user].
 * item_list = list(data.item.keys()) so item_list[j] is the item with internal id j. */
void orc_sample_batch(uint32_t *st, const int32_t *pairs, int64_t begin, int64_t count, int32_t n_items,
                      const int64_t *memb_rowptr, const int32_t *memb_items, int64_t memb_rows,
                      int32_t *out_u, int32_t *out_p, int32_t *out_n)
{
    for (int64_t b = 0; b < count; b++) {
        int32_t u = pairs[2 * (begin + b)], p = pairs[2 * (begin + b) + 1];
        int32_t neg = (int32_t)mt_randbelow(st, (uint32_t)n_items);
        while (memb_contains(memb_rowptr, memb_items, memb_rows, u, neg))
            neg = (int32_t)mt_randbelow(st, (uint32_t)n_items);
        out_u[b] = u; out_p[b] = p; out_n[b] = neg;
    }
}

/* ------------------------------------------------------------------------------------------
 * Normalised adjacency values.  util/DataLoader.py:73-87 normalize_graph_mat (d_inv[isinf]=0 guard) and
 * recommender/LightGCN.py:212-215 _init_uiAdj (no guard; an isolated node simply has no entries).
 * val[e] = (dinv[row] * w[e]) * dinv[col]  -- fp32, same association as diag @ A @ diag.
 * ------------------------------------------------------------------------------------------ */
void orc_norm_adj_values(int64_t n, const int64_t *rowptr, const int32_t *col, const float *w, float *val)
{
    float *dinv = (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    for (int64_t r = 0; r < n; r++) {
        float s = 0.f;                                   /* scipy sums fp32 rows in fp32 */
        for (int64_t e = rowptr[r]; e < rowptr[r + 1]; e++) s += w[e];
        dinv[r] = s > 0.f ? (float)(1.0f / sqrtf(s)) : 0.f;
    }
    for (int64_t r = 0; r < n; r++)
        for (int64_t e = rowptr[r]; e < rowptr[r + 1]; e++)
            val[e] = (dinv[r] * w[e]) * dinv[col[e]];
    free(dinv);
}

/* ------------------------------------------------------------------------------------------
 * SpMM  Y = alpha * (A X) + beta * Z.   recommender/LightGCN.py:234 torch.sparse.mm(sparse_norm_adj, ego)
 * ------------------------------------------------------------------------------------------ */
void orc_spmm_csr(int64_t n_rows, const int64_t *rowptr, const int32_t *col, const float *val,
                  const float *X, int64_t d, float alpha, float beta, const float *Z, float *Y)
{
#pragma omp parallel
    {
        double *acc = (double *)malloc(sizeof(double) * (size_t)d);
#pragma omp for schedule(dynamic, 64)
        for (int64_t r = 0; r < n_rows; r++) {
            for (int64_t k = 0; k < d; k++) acc[k] = 0.0;
            for (int64_t e = rowptr[r]; e < rowptr[r + 1]; e++) {
                const float *x = X + (int64_t)col[e] * d;
                double v = val[e];
                for (int64_t k = 0; k < d; k++) acc[k] += v * x[k];
            }
            for (int64_t k = 0; k < d; k++) {
                float y = (float)acc[k];
                Y[r * d + k] = (beta != 0.f && Z) ? alpha * y + beta * Z[r * d + k] : alpha * y;
            }
        }
        free(acc);
    }
}

/* fp32-accumulate variant used only for the timed CPU baseline (same arithmetic class as torch CPU) */
void orc_spmm_csr_f32acc(int64_t n_rows, const int64_t *rowptr, const int32_t *col, const float *val,
                         const float *X, int64_t d, float alpha, float beta, const float *Z, float *Y)
{
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t r = 0; r < n_rows; r++) {
        float acc[256];
        for (int64_t k = 0; k < d; k++) acc[k] = 0.f;
        for (int64_t e = rowptr[r]; e < rowptr[r + 1]; e++) {
            const float *x = X + (int64_t)col[e] * d;
            float v = val[e];
            for (int64_t k = 0; k < d; k++) acc[k] += v * x[k];
        }
        for (int64_t k = 0; k < d; k++)
            Y[r * d + k] = (beta != 0.f && Z) ? alpha * acc[k] + beta * Z[r * d + k] : alpha * acc[k];
    }
}

/* ------------------------------------------------------------------------------------------
 * BPR + un-squared L2 on gathered rows, forward and backward.
 * util/loss.py:5-9 bpr_loss (eps = 10e-8 = 1e-7), :25-29 l2_reg_loss (reg * sum of Frobenius norms),
 * gather rec_user_emb[user_idx] etc. recommender/LightGCN.py:51-54; backward = autograd of those lines:
 *   x_b = <u,p> - <u,n>, s = sigmoid(x_b), g_b = -s(1-s)/((1e-7+s) B)
 *   du = g_b (p - n) + reg u/||U_b||_F ; dp = g_b u + reg p/||P_b||_F ; dn = -g_b u ; rows scatter-ADDED.
 * `emb` is the combined [N,d] table (users first, items at row offset item_off).  G must be pre-zeroed.
 * ------------------------------------------------------------------------------------------ */
void orc_bpr_l2_fwd_bwd(const float *emb, int64_t d, int64_t item_off, const int32_t *ui, const int32_t *pi,
                        const int32_t *ni, int64_t B, float reg, float *loss_bpr, float *loss_reg, float *G)
{
    double lsum = 0.0, nu = 0.0, np_ = 0.0;
    float *gb = (float *)malloc(sizeof(float) * (size_t)(B > 0 ? B : 1));
    for (int64_t b = 0; b < B; b++) {
        const float *u = emb + (int64_t)ui[b] * d, *p = emb + (item_off + pi[b]) * d, *n = emb + (item_off + ni[b]) * d;
        double ps = 0.0, ns = 0.0;
        for (int64_t k = 0; k < d; k++) { ps += (double)u[k] * p[k]; ns += (double)u[k] * n[k]; nu += (double)u[k] * u[k]; np_ += (double)p[k] * p[k]; }
        float x = (float)ps - (float)ns;
        float s = 1.0f / (1.0f + expf(-x));
        lsum += -(double)logf(1e-7f + s);
        gb[b] = -(s * (1.0f - s)) / ((1e-7f + s) * (float)B);
    }
    float nrm_u = (float)sqrt(nu), nrm_p = (float)sqrt(np_);
    *loss_bpr = (float)(lsum / (double)B);
    *loss_reg = reg * (nrm_u + nrm_p);
    if (G) {
        for (int64_t b = 0; b < B; b++) {
            int64_t ru = ui[b], rp = item_off + pi[b], rn = item_off + ni[b];
            const float *u = emb + ru * d, *p = emb + rp * d, *n = emb + rn * d;
            float g = gb[b];
            for (int64_t k = 0; k < d; k++) {
                float uk = u[k], pk = p[k], nk = n[k];
                G[ru * d + k] += g * (pk - nk) + (nrm_u > 0.f ? reg * uk / nrm_u : 0.f);
                G[rp * d + k] += g * uk + (nrm_p > 0.f ? reg * pk / nrm_p : 0.f);
                G[rn * d + k] += -g * uk;
            }
        }
    }
    free(gb);
}

/* ------------------------------------------------------------------------------------------
 * torch.optim.Adam as used by the reference (recommender/LightGCN.py:33,64): betas (0.9,0.999), eps 1e-8,
 * no weight decay, amsgrad off.  Restates torch/optim/adam.py _single_tensor_adam:
 *   m.lerp_(g, 1-b1); v = v*b2 + (1-b2) g*g; step_size = lr/(1-b1^t); denom = sqrt(v)/sqrt(1-b2^t) + eps;
 *   p -= step_size * m/denom.      SGD (attack/White/PGA.py:59): p -= lr * g.
 * torch keeps lr and the betas as Python doubles and rounds the DERIVED weights to fp32: (float)(1 - 0.999) = 0.001f, not 1.0f - 0.999f
 * = 0.00100005.  This interface carries floats; the double the caller typed is recovered as the shortest decimal that rounds to the float.
 * ------------------------------------------------------------------------------------------ */
static double orc_typed_double(float x)
{
    char buf[32];
    snprintf(buf, sizeof buf, "%.7g", (double)x);
    double d = strtod(buf, NULL);
    return (float)d == x ? d : (double)x;
}

void orc_adam_step(float *p, const float *g, float *m, float *v, int64_t n, float lr, float b1, float b2, float eps, int64_t t)
{
    double B1 = orc_typed_double(b1), B2 = orc_typed_double(b2), LR = orc_typed_double(lr);
    double bc1 = 1.0 - pow(B1, (double)t), bc2 = 1.0 - pow(B2, (double)t);
    float step_size = (float)(LR / bc1);
    float bc2_sqrt = (float)sqrt(bc2);
    float w1 = (float)(1.0 - B1), w2 = (float)(1.0 - B2), b2f = (float)B2;
#pragma omp parallel for
    for (int64_t i = 0; i < n; i++) {
        float gi = g[i];
        float mi = m[i] + (gi - m[i]) * w1;
        float vi = v[i] * b2f + w2 * gi * gi;
        m[i] = mi; v[i] = vi;
        float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = p[i] - step_size * (mi / denom);
    }
}

void orc_sgd_step(float *p, const float *g, int64_t n, float lr)
{
#pragma omp parallel for
    for (int64_t i = 0; i < n; i++) p[i] = p[i] - lr * g[i];
}

/* ------------------------------------------------------------------------------------------
 * InfoNCE forward + backward.  util/loss.py:42-49:
 *   a=normalize(v1), b=normalize(v2) (F.normalize eps 1e-12); pos_i=exp(<a_i,b_i>/t); ttl_i=sum_j exp(<a_i,b_j>/t);
 *   loss = mean(-log(pos/ttl)).   Backward derived by hand:
 *   P_ij = exp(s_ij/t)/ttl_i ;  dL/ds_ij = (P_ij - [i==j]) / (n t)
 *   da_i = sum_j dS_ij b_j ; db_j = sum_i dS_ij a_i ; then through normalize: dx = (dy - y <y,dy>)/max(|x|,eps)
 * ------------------------------------------------------------------------------------------ */
void orc_infonce_fwd_bwd(const float *v1, const float *v2, int64_t n, int64_t d, float tau, float *loss, float *dv1, float *dv2)
{
    float *a = (float *)malloc(sizeof(float) * (size_t)(n * d)), *b = (float *)malloc(sizeof(float) * (size_t)(n * d));
    float *n1 = (float *)malloc(sizeof(float) * (size_t)n), *n2 = (float *)malloc(sizeof(float) * (size_t)n);
    double *da = (double *)calloc((size_t)(n * d), sizeof(double)), *db = (double *)calloc((size_t)(n * d), sizeof(double));
    for (int64_t i = 0; i < n; i++) {
        double s1 = 0, s2 = 0;
        for (int64_t k = 0; k < d; k++) { s1 += (double)v1[i * d + k] * v1[i * d + k]; s2 += (double)v2[i * d + k] * v2[i * d + k]; }
        n1[i] = fmaxf((float)sqrt(s1), 1e-12f); n2[i] = fmaxf((float)sqrt(s2), 1e-12f);
        for (int64_t k = 0; k < d; k++) { a[i * d + k] = v1[i * d + k] / n1[i]; b[i * d + k] = v2[i * d + k] / n2[i]; }
    }
    double total = 0.0;
    double *e = (double *)malloc(sizeof(double) * (size_t)n);
    for (int64_t i = 0; i < n; i++) {
        double ttl = 0.0, sii = 0.0;
        for (int64_t j = 0; j < n; j++) {
            double s = 0.0;
            for (int64_t k = 0; k < d; k++) s += (double)a[i * d + k] * b[j * d + k];
            if (j == i) sii = s;
            e[j] = exp((double)(float)s / tau);
            ttl += e[j];
        }
        total += -(sii / tau - log(ttl));
        if (dv1) {
            for (int64_t j = 0; j < n; j++) {
                double ds = (e[j] / ttl - (j == i ? 1.0 : 0.0)) / ((double)n * tau);
                for (int64_t k = 0; k < d; k++) { da[i * d + k] += ds * b[j * d + k]; db[j * d + k] += ds * a[i * d + k]; }
            }
        }
    }
    *loss = (float)(total / (double)n);
    if (dv1) {
        for (int64_t i = 0; i < n; i++) {
            double dot1 = 0, dot2 = 0;
            for (int64_t k = 0; k < d; k++) { dot1 += a[i * d + k] * da[i * d + k]; dot2 += b[i * d + k] * db[i * d + k]; }
            for (int64_t k = 0; k < d; k++) {
                dv1[i * d + k] = (float)((da[i * d + k] - a[i * d + k] * dot1) / n1[i]);
                dv2[i * d + k] = (float)((db[i * d + k] - b[i * d + k] * dot2) / n2[i]);
            }
        }
    }
    free(a); free(b); free(n1); free(n2); free(da); free(db); free(e);
}

/* ------------------------------------------------------------------------------------------
 * SimGCL perturbation.  recommender/SimGCL.py:203-205:  E += sign(E) * normalize(noise, dim=-1) * eps
 * (F.normalize eps 1e-12; noise comes from torch.rand_like -- injected here for parity).
 * ------------------------------------------------------------------------------------------ */
void orc_simgcl_perturb(float *E, const float *noise, int64_t n, int64_t d, float eps)
{
#pragma omp parallel for
    for (int64_t r = 0; r < n; r++) {
        double s = 0;
        for (int64_t k = 0; k < d; k++) s += (double)noise[r * d + k] * noise[r * d + k];
        float nr = fmaxf((float)sqrt(s), 1e-12f);
        for (int64_t k = 0; k < d; k++) {
            float x = E[r * d + k];
            float sg = (x > 0.f) ? 1.f : (x < 0.f ? -1.f : 0.f);
            E[r * d + k] = x + sg * (noise[r * d + k] / nr) * eps;
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * SDDMM on selected rows: gradient of a loss w.r.t. the adjacency VALUES, restricted to the given rows.
 * attack/White/PGA.py:117 autograd.grad(Loss, sparse_norm_adj): for Y = A X, dA[i,j] = <dY[i], X[j]> on the
 * pattern; summed over layers by the caller.  out[t*n_cols + j] += <dY[rows[t]], X[col_off + j]> for ALL j
 * in [0,n_cols) (dense block form: PGA's fake rows carry a dense weight row, attack/White/PGA.py:69-73).
 * ------------------------------------------------------------------------------------------ */
void orc_sddmm_rows_dense(const float *dY, const float *X, int64_t d, const int32_t *rows, int64_t n_rows,
                          int64_t col_off, int64_t n_cols, float *out)
{
#pragma omp parallel for
    for (int64_t t = 0; t < n_rows; t++)
        for (int64_t j = 0; j < n_cols; j++) {
            double s = 0;
            const float *a = dY + (int64_t)rows[t] * d, *b = X + (col_off + j) * d;
            for (int64_t k = 0; k < d; k++) s += (double)a[k] * b[k];
            out[t * n_cols + j] += (float)s;
        }
}

/* ------------------------------------------------------------------------------------------
 * scores = Pu @ Pi^T, mask interacted -> -10e8, top-k per user.
 * attack/White/DLAttack.py:73-83, attack/White/CLeaR.py:75-82 (mask uses the CSR of uiAdj2.nonzero()).
 * Ties: lower item id first (torch.topk leaves ties unspecified; fixtures avoid exact ties).
 * mask_rowptr may be NULL (PGA.py:101-102 takes top-50 with no mask).
 * ------------------------------------------------------------------------------------------ */
void orc_score_mask_topk(const float *Pu, const float *Pi, int64_t U, int64_t I, int64_t d,
                         const int64_t *mask_rowptr, const int32_t *mask_col, int64_t k, int32_t *top_idx, float *top_val)
{
#pragma omp parallel
    {
        float *sc = (float *)malloc(sizeof(float) * (size_t)I);
#pragma omp for schedule(dynamic, 8)
        for (int64_t u = 0; u < U; u++) {
            for (int64_t i = 0; i < I; i++) {
                double s = 0;
                for (int64_t c = 0; c < d; c++) s += (double)Pu[u * d + c] * Pi[i * d + c];
                sc[i] = (float)s;
            }
            if (mask_rowptr)
                for (int64_t e = mask_rowptr[u]; e < mask_rowptr[u + 1]; e++) sc[mask_col[e]] = -10e8f;
            for (int64_t t = 0; t < k; t++) {           /* selection: k small (50) */
                int64_t best = -1;
                for (int64_t i = 0; i < I; i++)
                    if (best < 0 || sc[i] > sc[best]) best = i;
                top_idx[u * k + t] = (int32_t)best; top_val[u * k + t] = sc[best];
                sc[best] = -INFINITY;
            }
        }
        free(sc);
    }
}

/* attack/White/PGA.py:153-158, CLeaR.py:161-166, DLAttack.py:127-132: per row top-n indices -> {0,1} row.
 * Ties: lower index first. */
void orc_topn_project_rows(const float *M, int64_t rows, int64_t cols, int64_t n, float *out, int32_t *idx)
{
    float *tmp = (float *)malloc(sizeof(float) * (size_t)cols);
    for (int64_t r = 0; r < rows; r++) {
        memcpy(tmp, M + r * cols, sizeof(float) * (size_t)cols);
        for (int64_t j = 0; j < cols; j++) out[r * cols + j] = 0.f;
        for (int64_t t = 0; t < n; t++) {
            int64_t best = -1;
            for (int64_t j = 0; j < cols; j++)
                if (best < 0 || tmp[j] > tmp[best]) best = j;
            out[r * cols + best] = 1.f;
            if (idx) idx[r * n + t] = (int32_t)best;
            tmp[best] = -INFINITY;
        }
    }
    free(tmp);
}

/* attack/White/PGA.py:117-139.  doubleGrad = autograd.grad(Loss, sparse_norm_adj) exists only on the stored pattern,
 * is scaled D^-1/2 . D^-1/2 (:118-126), densified and sliced to the fake rows (:128-134); then
 * S -= 0.2*tanh(grad); S[S>1]=1; S[S<=0]=10e-8 (:135-139).  Entries with S == 0 are not in the pattern -> grad 0. */
void orc_pga_update(float *S, const float *grad, const float *dinv_rows, const float *dinv_cols, int64_t rows, int64_t cols)
{
    for (int64_t r = 0; r < rows; r++)
        for (int64_t c = 0; c < cols; c++) {
            int64_t i = r * cols + c;
            float g = grad[i];
            if (dinv_rows) g *= dinv_rows[r];
            if (dinv_cols) g *= dinv_cols[c];
            if (S[i] == 0.f) g = 0.f;
            float s = S[i] - 0.2f * tanhf(g);
            if (s > 1.f) s = 1.f;
            if (s <= 0.f) s = 10e-8f;
            S[i] = s;
        }
}

int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
