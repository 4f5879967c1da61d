/*
 * arlib_amd.h -- C ABI of libarlib_amd.so: the MI355X (gfx950) implementation of ARLib's
 * embedding-recommender training + white-box attack hot path.
 *
 * The reference (CoderWZW/ARLib) is 100 % Python and has no FFI: its "kernels" are implicit ATen ops.
 * Each entry point below therefore cites the reference Python lines whose device work it replaces
 * (paths relative to the reference root).  INTEGRATION.md shows the ctypes binding a maintainer adds.
 *
 * Conventions
 *   - return 0 on success; negative = argument error (ARL_E_*); positive = hipError_t.
 *   - never throws across the ABI, never allocates or frees device memory, takes no ownership:
 *     every buffer and workspace is caller-allocated; sizes of workspaces come from *_workspace_bytes.
 *   - device entry points are asynchronous on `stream` (a hipStream_t passed as void*; NULL = default
 *     stream) and re-entrant across distinct streams as long as workspaces are distinct.
 *   - all matrices are row-major fp32; all indices int32; row offsets int32 (nnz < 2^31).
 *   - "emb" tables are ONE contiguous [N,d] buffer: user rows first, item rows at row offset item_off.
 *   - d must be a multiple of 4 and <= 256 for the propagation kernels (reference default d = 64).
 */
#ifndef ARLIB_AMD_H
#define ARLIB_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ARL_OK 0
#define ARL_E_NULL (-1)      /* required pointer is NULL                     */
#define ARL_E_DIM (-2)       /* unsupported embedding size / shape           */
#define ARL_E_RANGE (-3)     /* size out of the int32 index range            */
#define ARL_E_ARG (-4)       /* inconsistent arguments                       */

typedef void *arl_stream_t;

/* ABI version; bump on any signature change. */
int arl_abi_version(void);

/* ------------------------------------------------------------------------------------------------
 * Host sampler -- replaces util/sampler.py:4-30 (next_batch_pairwise) bit-exactly, including the
 * CPython `random` MT19937 stream it consumes (util/tool.py:101-108 seedSet -> random.seed).
 * `mt_state` is random.getstate()[1] as 625 uint32 (624 words + index); it is advanced in place so
 * the caller can random.setstate() it back and stay in lock-step with the reference.
 * Not thread-safe per state object.
 * ---------------------------------------------------------------------------------------------- */
/* random.seed(int): key = 32-bit little-endian words of abs(seed) */
int arl_mt_seed(uint32_t *mt_state, const uint32_t *key, int64_t key_len);
/* util/sampler.py:9   shuffle(training_data) in place; pairs = int32 [nnz][2] (user id, item id) */
int arl_sampler_shuffle(uint32_t *mt_state, int32_t *pairs, int64_t nnz);
/* random.sample(range(n), k) of CPython 3.10 on the caller's MT19937 state -- what the graph augmentations of the
 * contrastive encoders draw (recommender/SGL.py:281-299: edge / node dropout).  use_pool: the caller evaluates CPython's
 * set-size rule (n <= 21 + (k > 5 ? 4 ** ceil(log(3k, 4)) : 0)).  scratch: n int32 (pool form) or (n+31)/32 int32. */
int arl_mt_sample_range(uint32_t *mt_state, int64_t n, int64_t k, int32_t use_pool, int32_t *out, int32_t *scratch);
/* util/sampler.py:12-29  one batch: positives pairs[begin..begin+count), one negative per positive drawn
 * by choice(item_list) with rejection against training_set_u[user] (given as a CSR with sorted item
 * ids; users >= memb_rows have an empty set, which is what the reference's defaultdict gives users
 * injected after DataLoader construction). */
int arl_sampler_next_batch(uint32_t *mt_state, const int32_t *pairs, int64_t begin, int64_t count,
                           int32_t n_items, const int64_t *memb_rowptr, const int32_t *memb_items,
                           int64_t memb_rows, int32_t *out_u, int32_t *out_p, int32_t *out_n);

/* ------------------------------------------------------------------------------------------------
 * CSR adjacency + long-row plan.  The plan splits rows longer than `chunk` edges into chunk tasks so
 * a 100k-edge popular-item row does not serialise one wavefront; partial sums go to `partial`
 * ([n_chunks][d] fp32, caller-allocated) and are combined in chunk order (deterministic).
 * n_chunks == 0 means "no plan": every row is processed whole.
 * ---------------------------------------------------------------------------------------------- */
typedef struct arl_csr {
    int64_t n_rows;
    int64_t nnz;
    const int32_t *rowptr;      /* [n_rows+1] device */
    const int32_t *col;         /* [nnz] device      */
    const float *val;           /* [nnz] device      */
    int32_t chunk;              /* rows with more than `chunk` edges are split (0 = never)       */
    int64_t n_chunks;
    const int32_t *chunk_row;   /* [n_chunks] device: output row of the task                      */
    const int32_t *chunk_begin; /* [n_chunks] device: first edge                                  */
    const int32_t *chunk_end;   /* [n_chunks] device: one past last edge                          */
    int64_t n_long;
    const int32_t *long_row;    /* [n_long] device                                                */
    const int32_t *long_first;  /* [n_long] device: first chunk slot of the row                   */
    const int32_t *long_count;  /* [n_long] device: number of chunk slots                         */
    float *partial;             /* [n_chunks * d] device workspace                                */
    const int32_t *row_tasks;   /* optional [n_rows][4] = (row, first edge, one past last edge, 0): the rows in the order the flag-masked
                                 * hop should take them -- sorted by edge count, so that the four lane groups of a wave own rows of equal
                                 * length (a wave lasts as long as its longest row).  NULL: rows in index order.               */
} arl_csr;

/* Degree-normalised edge values on device.  Replaces util/DataLoader.py:73-87 (normalize_graph_mat) and
 * recommender/LightGCN.py:212-215 (_init_uiAdj): val[e] = (dinv[row]*w[e])*dinv[col[e]], dinv = rowsum^-1/2
 * (0 for an empty row).  dinv: [n_rows] device workspace (kept: PGA re-uses it, attack/White/PGA.py:118-126). */
int arl_norm_adj_values_f32(int64_t n_rows, const int32_t *rowptr, const int32_t *col, const float *w,
                            float *dinv, float *val, arl_stream_t stream);
/* Same result for callers that keep the row id of every edge (erow[e], CSR order): the edge values are computed edge-parallel
 * (no wave per row).  All arrays 16-byte aligned. */
int arl_norm_adj_values_coo_f32(int64_t n_rows, const int32_t *rowptr, const int32_t *erow, const int32_t *col,
                                const float *w, int64_t nnz, float *dinv, float *val, arl_stream_t stream);
/* The value pass alone, for a caller that already has dinv (PGA: the row sums of its fixed-pattern graph follow from the F x I fake
 * block in O(F I), so the 77 M-edge row-sum pass is not needed): val[e] = (dinv[erow[e]] * w[e]) * dinv[col[e]]. */
int arl_norm_vals_coo_f32(const int32_t *erow, const int32_t *col, const float *w, int64_t nnz, const float *dinv, float *val,
                          arl_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * SpMM family -- replaces torch.sparse.mm(sparse_norm_adj, ego) and its autograd
 * (recommender/LightGCN.py:234, recommender/SimGCL.py:202) plus the fused neighbours:
 *   arl_spmm_csr_f32          Y = alpha*(A X) + beta*Z            (Z may be NULL iff beta == 0)
 *   arl_spmm_csr_layersum_f32 Y = A X ; S = S_in + Y              (torch.stack(...).mean running sum,
 *                                                                  LightGCN.py:236-237; S may alias S_in)
 *   arl_spmm_csr_adam_f32     g = alpha*(A X) + beta*Z ; Adam(P,M,V; g)   (last backward hop fused with
 *                                                                  torch.optim.Adam.step, LightGCN.py:64)
 * X, Y, Z, S, P, M, V: [n_rows, d] (square adjacency).  Y must not alias X.
 * ---------------------------------------------------------------------------------------------- */
int arl_spmm_csr_f32(const arl_csr *A, const float *X, int64_t d, float alpha, float beta, const float *Z,
                     float *Y, arl_stream_t stream);
int arl_spmm_csr_layersum_f32(const arl_csr *A, const float *X, int64_t d, const float *S_in, float *S,
                              float *Y, arl_stream_t stream);
int arl_spmm_csr_adam_f32(const arl_csr *A, const float *X, int64_t d, float alpha, float beta, const float *Z,
                          const uint8_t *zflags, float *P, float *M, float *V, float lr, float beta1, float beta2,
                          float eps, int64_t step, arl_stream_t stream);

/* Sparse-batch-gradient forms.  In one training step dL/d(out) (= G) is non-zero on at most 3B rows and the last forward
 * hop is consumed on those rows only, so two of the 2L full-graph hops of recommender/LightGCN.py:230-240 + autograd
 * collapse to work proportional to the batch neighbourhood (results identical to the dense hops):
 *   arl_spmm_csr_flagged_f32  Y = alpha*(A X) + beta*Z where X is zero except on rows whose bit is set in the bitmap xbits
 *                             (uint32 words, bit c&31 of word c>>5; NULL = dense gather) and Z is read only where
 *                             zflags != 0 (NULL = everywhere).  xbits: ceil(n_cols/32) words, zflags: [n_rows] bytes.
 *   arl_spmm_csr_rows_f32     out_c[t] = alpha * w[t] * ( sum_k layers[k][rows[t]] + (A X)[rows[t]] ), t < n_rows_sel: only the
 *                             listed rows are produced (duplicates allowed).  Each row is cut into `nsplit` edge ranges;
 *                             workspace = arl_spmm_csr_rows_workspace_bytes(n_rows_sel, nsplit, d).  layers: HOST array of
 *                             n_layers (<= 8) device pointers to [n_rows, d] tables.  row_weight: optional [n_rows_sel] factors
 *                             w (NULL = 1; the user-sharded step passes 1 for samples whose user the rank owns, 0 otherwise).
 *   arl_spmm_csr_adam_f32's zflags has the same meaning (NULL = read Z everywhere).
 *   arl_mark_rows_u8 / arl_zero_rows_f32: flags[idx[t]] = value / dst[idx[t], :] = 0 -- set and clear the batch's sparse state. */
int arl_spmm_csr_flagged_f32(const arl_csr *A, const float *X, int64_t d, const uint32_t *xbits, float alpha, float beta,
                             const float *Z, const uint8_t *zflags, float *Y, arl_stream_t stream);
int64_t arl_spmm_csr_rows_workspace_bytes(int64_t n_rows_sel, int64_t nsplit, int64_t d);
int arl_spmm_csr_rows_f32(const arl_csr *A, const float *X, int64_t d, const int32_t *rows, int64_t n_rows_sel,
                          int64_t nsplit, const float *const *layers, int64_t n_layers, float alpha, const float *row_weight,
                          float *out_c, void *workspace, arl_stream_t stream);
int arl_mark_rows_u8(uint8_t *flags, const int32_t *idx, int64_t n, int32_t value, arl_stream_t stream);
/* set (set != 0) or clear the bits idx[t] of a bitmap with atomic OR / AND (duplicates allowed) */
int arl_mark_rows_bits_u32(uint32_t *bits, const int32_t *idx, int64_t n, int32_t set, arl_stream_t stream);
int arl_zero_rows_f32(float *dst, const int32_t *idx, int64_t n, int64_t d, arl_stream_t stream);
/* The three per-batch updates of the sparse-batch step on one index list, fused: G[idx[t]] += scale * row_scale[t] * src[t]
 * (row_scale optional, NULL = 1), flags[idx[t]] = 1, bit idx[t] of `bits` set -- and the clearing counterpart (rows zeroed, flag and
 * bit cleared).  Duplicates accumulate IN INDEX ORDER, one wave per distinct row, starting from the value already in G: the association
 * of CPU index_put_(accumulate=True), i.e. of the reference's gathers under autograd (recommender/LightGCN.py:51-56); no float atomics,
 * bit-identical from run to run.
 * dup_bits (optional, a second bitmap of the size of `bits`, all-zero on entry like `bits`): a first launch records in it the rows the list
 * names more than once, so that only those are summed by the ordered scan and every other row is a plain add (44 -> ~10 us at cfg2); the
 * clearing call takes the same pointer and zeroes it again.
 * An entry with row_scale[t] == 0 is ABSENT: nothing is added, its row is neither flagged nor marked nor counted as a duplicate (the user-sharded
 * step gives foreign samples factor 0 on a clamped row to keep static shapes: they must not pile up on that row). */
int arl_batch_rows_set_f32(float *G, uint8_t *flags, uint32_t *bits, const int32_t *idx, int64_t n, int64_t d,
                           const float *src, float scale, const float *row_scale, uint32_t *dup_bits, arl_stream_t stream);
int arl_batch_rows_clear_f32(float *G, uint8_t *flags, uint32_t *bits, const int32_t *idx, int64_t n, int64_t d,
                             uint32_t *dup_bits, arl_stream_t stream);

/* Register-blocked SpMM (d = 64: one column per lane; d = 128: two adjacent columns per lane, 16 operand rows in flight): the same three operations as arl_spmm_csr_f32 / _layersum_f32 / _adam_f32, for the rows
 * of a PLAN.  A wave owns up to rows_per_wave (16 or 32) output rows with register accumulators and consumes one record stream
 * sorted by (column block, row slot); waves carry equal edge counts (arl_lpt_deal) and are all resident, so they sweep X in step
 * and share the gathered rows through the L2.  Rows absent from the plan (longer than its hub threshold) are not written: the
 * caller runs them through the chunked CSR kernel.  Plans are built by arlib_amd/ops.py:BlockedPlan.
 * Summation order differs from the CSR kernel: results agree to fp32 rounding, not bitwise; each is deterministic. */
typedef struct arl_blocked {
    int64_t n_waves, rows_per_wave;
    int64_t loads_in_flight;    /* 16 or 32 operand rows requested per wave before the first is consumed            */
    const int32_t *wave_ptr;    /* [n_waves + 1] record offsets, multiples of 64                          */
    const int32_t *wave_rows;   /* [n_waves][rows_per_wave] output row id, -1 = unused slot               */
    const int32_t *rec_col;     /* column | slot << 24 (columns < 2^24); padding records have val 0       */
    const float *rec_val;
    /* split rows: a row longer than the plan's hub threshold is dealt as several PIECES (piece p of P takes every P-th edge of the
     * column-sorted row, so each piece sweeps the whole column range like an ordinary row); a piece's slot holds -(2 + t) in
     * wave_rows and its raw sum goes to partial[t]; after the sweep the n_split rows are combined in piece order (deterministic)
     * and run the epilogue.  n_split = 0: no split rows (partial and the split_* arrays may be NULL). */
    int64_t n_split;
    const int32_t *split_row;   /* [n_split] output row                                                    */
    const int32_t *split_first; /* [n_split] first piece                                                   */
    const int32_t *split_count; /* [n_split] number of pieces                                              */
    float *partial;             /* [n_pieces][d] workspace                                                 */
    int64_t waves_per_group;    /* 1, 2 or 4 wavefronts per workgroup (0 = 4): the dispatcher balances workgroups, not waves, over
                                 * the CUs, so a launch of ~12 waves per CU is spread evenly only with small workgroups         */
} arl_blocked;
int arl_spmm_blocked_f32(const arl_blocked *P, const float *X, int64_t d, float alpha, float beta, const float *Z,
                         const uint8_t *zflags, float *Y, arl_stream_t stream);
/* Y = alpha * diag(row_scale) (A X) + beta * Z: the product with a diagonal factor in the epilogue (PGA applies D^-1/2 W D^-1/2 in
 * factors and keeps W's values fixed); CSR and blocked schedules. */
int arl_spmm_csr_rscale_f32(const arl_csr *A, const float *X, int64_t d, const float *row_scale, float alpha, float beta,
                            const float *Z, float *Y, arl_stream_t stream);
int arl_spmm_blocked_rscale_f32(const arl_blocked *P, const float *X, int64_t d, const float *row_scale, float alpha,
                                float beta, const float *Z, float *Y, arl_stream_t stream);
int arl_spmm_blocked_layersum_f32(const arl_blocked *P, const float *X, int64_t d, const float *S_in, float *S, float *Y,
                                  arl_stream_t stream);
int arl_spmm_blocked_adam_f32(const arl_blocked *P, const float *X, int64_t d, float alpha, float beta, const float *Z,
                              const uint8_t *zflags, float *Pm, float *M, float *V, float lr, float beta1, float beta2,
                              float eps, int64_t step, arl_stream_t stream);
/* Host helper of the plan: rows sorted by edge count (descending) are dealt to the least loaded of n_bins waves with a free
 * slot (cap slots each); bin_out / slot_out receive the placement. */
int arl_lpt_deal(int64_t n, const int32_t *weight_desc, int64_t n_bins, int64_t cap, int32_t *bin_out, int32_t *slot_out);

/* SYN-v1, the benchmark's synthetic interaction graphs (no reference counterpart: ARLib ships fixed datasets, data/clean/),
 * generated natively.  Every random number is a counter-based hash h(stream, index) = splitmix64(splitmix64(seed ^ stream * PHI) ^ index):
 * log-normal user degrees (streams 1, 2), item draws with popularity density ~ x^-1/2 through a hash-derived item permutation (streams 3, 4),
 * one extra edge per item so that none is isolated (stream 5); user-major sorted, de-duplicated int32 (user, item) pairs.  Returns the pair
 * count; writes nothing and returns the count needed when pairs_out is NULL or capacity is too small.  arl_graph_digest is the
 * order-sensitive checksum both this generator and the numpy one (arlib_amd/util/synthetic.py) are compared by. */
int64_t arl_syn_v1_pairs(int64_t n_users, int64_t n_items, double mean_deg, uint64_t seed, double sigma, int64_t deg_min, int64_t deg_max,
                         int32_t *pairs_out, int64_t capacity);
uint64_t arl_graph_digest(const int32_t *pairs, int64_t n);

/* L2-blocked ("tiled") SpMM: same results as arl_spmm_csr_f32 / _adam_f32 on the same adjacency, different schedule.
 * Output rows are dealt into BINS of <= cap rows with equal edge counts; one persistent workgroup per CU keeps a bin's fp32
 * accumulators in LDS for a sweep and walks column blocks of `col_block` rows in ascending order, so the gathered rows of X
 * are served from the XCD's L2 instead of the fabric.  The plan is built once per graph (arlib_amd/ops.py:TiledPlan).
 * Every local row of a bin is owned by one lane group for the whole sweep; a bin's edges are sorted by (owner group, column
 * block, local row, column): each group streams one contiguous list that walks the column blocks in ascending order, LDS
 * accumulators are updated without atomics and the result is deterministic.
 * Summation order differs from the CSR kernel: results agree to fp32 rounding, not bitwise. */
typedef struct arl_tiled {
    int64_t n_sweeps, n_slots, cap, n_cb, nnz;
    int64_t n_groups;           /* owner groups per workgroup = 16 waves * (64 / lanes-per-row); fixed by d's width class  */
    const int32_t *bin_rows;    /* [n_sweeps*n_slots][cap] global row id, -1 = empty          */
    const int32_t *seg_ptr;     /* [n_sweeps*n_slots*n_groups + 1] offsets of the (bin, owner group) edge lists          */
    const int32_t *e_col;       /* [nnz] */
    const float *e_val;         /* [nnz] */
    const uint16_t *e_row;      /* [nnz] local row in the bin (cap <= 65535)                   */
} arl_tiled;
int arl_spmm_tiled_f32(const arl_tiled *T, const float *X, int64_t d, float alpha, float beta, const float *Z,
                       const uint8_t *zflags, float *Y, arl_stream_t stream);
int arl_spmm_tiled_adam_f32(const arl_tiled *T, const float *X, int64_t d, float alpha, float beta, const float *Z,
                            const uint8_t *zflags, float *P, float *M, float *V, float lr, float beta1, float beta2,
                            float eps, int64_t step, arl_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * BPR + un-squared L2 on gathered rows, forward + backward in one call.
 * Replaces util/loss.py:5-9 (bpr_loss, eps 10e-8), :25-29 (l2_reg_loss), the three gathers
 * rec_user_emb[user_idx] ... (recommender/LightGCN.py:51-52) and their autograd (index_put accumulate).
 *   loss_out[0] = mean(-log(1e-7 + sigmoid(<u,p>-<u,n>)))   loss_out[1] = reg*(||U_b||_F + ||P_b||_F)
 *   loss_out[2] = ||U_b||_F   loss_out[3] = ||P_b||_F
 *   G[row] += upstream * d(loss)/d(emb[row]): duplicates accumulate IN SAMPLE ORDER (a user row: its samples' terms in order; an item
 *   row: its positive-role terms in order, then its negative-role ones -- autograd's three index_put_(accumulate) of the reference),
 *   one wave per distinct row, no float atomics, bit-identical from run to run; G pre-zeroed or holding other gradient terms.
 *   G may be NULL for forward only.  distinct_rows != 0: the caller guarantees that the 3B rows u[b], item_off + p[b], item_off + n[b]
 *   are pairwise distinct (the compact [3B, d] batch form) -- the ownership scan is skipped.
 * workspace: arl_bpr_l2_workspace_bytes(B) bytes of device memory.
 * ---------------------------------------------------------------------------------------------- */
int64_t arl_bpr_l2_workspace_bytes(int64_t B);
int arl_bpr_l2_fwd_bwd_f32(const float *emb, int64_t d, int64_t item_off, const int32_t *u, const int32_t *p,
                           const int32_t *n, int64_t B, float reg, float upstream, float *loss_out, float *G,
                           void *workspace, int32_t distinct_rows, arl_stream_t stream);

/* User-sharded form of the same loss (one process per GPU holds a block of users): the batch is split by
 * user shard, so the mean (1/B_global) and the two Frobenius norms need the whole batch.
 *   arl_bpr_l2_partial_f32  : per-sample coefficients into `workspace`, and sums_out[0..2] = sum of -log(...) terms,
 *                             sum |u|^2, sum |p|^2 over the B_local samples -> caller all-reduces (RCCL) the 3 floats.
 *   arl_bpr_l2_backward_f32 : scatter-adds the gradient of the B_local samples into G using norms4[2] = ||U_b||_F and
 *                             norms4[3] = ||P_b||_F of the WHOLE batch (same layout as loss_out above). */
int arl_bpr_l2_partial_f32(const float *emb, int64_t d, int64_t item_off, const int32_t *u, const int32_t *p,
                           const int32_t *n, int64_t B_local, int64_t B_global, float *sums_out, void *workspace,
                           arl_stream_t stream);
int arl_bpr_l2_backward_f32(const float *emb, int64_t d, int64_t item_off, const int32_t *u, const int32_t *p,
                            const int32_t *n, int64_t B_local, float reg, float upstream, const float *norms4, float *G,
                            const void *workspace, arl_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Dense optimisers -- torch.optim.Adam (betas, eps, no weight decay; recommender/LightGCN.py:33,64) and
 * torch.optim.SGD (attack/White/PGA.py:59) over a whole table.  `step` is the 1-based step count.
 * ---------------------------------------------------------------------------------------------- */
int arl_adam_dense_f32(float *p, const float *g, float *m, float *v, int64_t n, float lr, float beta1,
                       float beta2, float eps, int64_t step, arl_stream_t stream);
int arl_sgd_dense_f32(float *p, const float *g, int64_t n, float lr, arl_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Row gather / scatter-add helpers (advanced indexing of the propagated tables and its backward).
 *   gather:       dst[t] = src[idx[t]]
 *   scatter_add:  dst[idx[t]] += scale * src[t]   (duplicates accumulate in index order -- torch's CPU index_put_(accumulate=True)
 *                 association, the backward of the reference's advanced-index gathers; ordered, no float atomics, deterministic)
 *   axpy_unique:  dst[r] += alpha * src[r] once for every DISTINCT row r in idx (src, dst: tables of the same shape); dup_bits: optional bitmap
 *                 (as filled by arl_batch_rows_set_f32 for a list containing idx) of the rows that may be named twice -- the others skip the scan
 *   shard_batch_prep: user-sharded batch bookkeeping in one launch -- lu[b] = clamp(u[b] - u0, 0, Ul - 1), own [3B] = [(u0 <= u[b] < u1) | 1 | 1]
 *                 (factors of the batch's user / positive / negative contributions), item_rows = [p | n], rows_l = [lu | Ul + p | Ul + n] with Ul = u1 - u0 (no reference counterpart: main.py:19 pins one device)
 * ---------------------------------------------------------------------------------------------- */
int arl_gather_rows_f32(const float *src, const int32_t *idx, int64_t n, int64_t d, float *dst, arl_stream_t stream);
int arl_scatter_add_rows_f32(float *dst, const int32_t *idx, int64_t n, int64_t d, const float *src, float scale,
                             arl_stream_t stream);
int arl_rows_axpy_unique_f32(float *dst, const float *src, const int32_t *idx, int64_t n, int64_t d, float alpha,
                             const uint32_t *dup_bits, arl_stream_t stream);
int arl_shard_batch_prep_i32(const int32_t *u, const int32_t *p, const int32_t *n, int64_t B, int64_t u0, int64_t u1,
                             int32_t *lu, float *own, int32_t *item_rows, int32_t *rows_l, arl_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * InfoNCE forward + backward -- replaces util/loss.py:42-49 and its autograd.
 *   loss_out[0] = mean_i( -log( exp(<a_i,b_i>/tau) / sum_j exp(<a_i,b_j>/tau) ) ), a,b = row-normalised v1,v2
 *   dv1, dv2 = upstream * gradients (may be NULL for forward only).   n <= 8192, d <= 256.
 * workspace: arl_infonce_workspace_bytes(n, d).
 * ---------------------------------------------------------------------------------------------- */
int64_t arl_infonce_workspace_bytes(int64_t n, int64_t d);
int arl_infonce_fwd_bwd_f32(const float *v1, const float *v2, int64_t n, int64_t d, float tau, float upstream,
                            float *loss_out, float *dv1, float *dv2, void *workspace, arl_stream_t stream);

/* SimGCL perturbation -- recommender/SimGCL.py:203-205: E += sign(E) * normalize(noise, dim=-1) * eps, in place. */
int arl_simgcl_perturb_f32(float *E, const float *noise, int64_t n, int64_t d, float eps, arl_stream_t stream);
/* The same perturbation with the noise drawn inside the kernel (recommender/SimGCL.py:203-205: random_noise = torch.rand_like(ego)):
 *   dst[r, k] = src[r, k] + sign(src[r, k]) * u[r, k] / max(||u[r, :]||, 1e-12) * eps,
 *   u[r, k] = uniform [0, 1) from a counter-based hash of (seed, stream_id, row * d + k), row = row_ids ? row_ids[r] : r.
 * No noise table, no clone of the operand (dst may alias src); with row_ids the operand is a compact slice of a larger table and receives
 * exactly the noise its rows get in a full-table call with the same (seed, stream_id).  d <= 256. */
int arl_simgcl_perturb_rng_f32(const float *src, float *dst, int64_t n, int64_t d, const int32_t *row_ids, float eps, uint64_t seed,
                               uint64_t stream_id, arl_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * NGCF layer glue -- recommender/NGCF.py:200-208:  E' = leaky_relu((P + E) W1 + (P * E) W2),  P = A_hat E.
 * The d x d products are ONE GEMM of [S | T] (N x 2d) with [W1; W2] (the caller's BLAS); these are the passes around it.
 *   combine      : ST[row] = [ P[row] + E[row] | P[row] * E[row] ]                       (ST: [n, 2d])
 *   act          : Z <- leaky_relu(Z, slope) in place;  acc += Z when acc != NULL        (layer sum)
 *   act_bwd      : gZ = gOut * (Out > 0 ? 1 : slope)
 *   combine_bwd  : from gST = [gS | gT]:  gP = gS + gT * E,  gE = gS + gT * P
 * d % 4 == 0.
 * ---------------------------------------------------------------------------------------------- */
int arl_ngcf_combine_f32(const float *P, const float *E, int64_t n, int64_t d, float *ST, arl_stream_t stream);
int arl_ngcf_act_f32(float *Z, float *acc, int64_t n, int64_t d, float slope, arl_stream_t stream);
int arl_ngcf_act_bwd_f32(const float *gOut, const float *Out, int64_t n, int64_t d, float slope, float *gZ, arl_stream_t stream);
int arl_ngcf_combine_bwd_f32(const float *gST, const float *P, const float *E, int64_t n, int64_t d, float *gP, float *gE,
                             arl_stream_t stream);
/* The same layer with its dense part on the matrix cores (v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulation; d in {16, 32, 64, 128}):
 * the [N, 2d] operand [P + E | P * E] of recommender/NGCF.py:200-208 is formed in registers, never in memory.
 *   fwd   : out = leaky_relu((P + E) W1 + (P * E) W2),  W = [W1; W2] as [2d, d] row-major
 *   dgrad : gZ = gOut * act'(Out);  gP = gS + gT * E,  gE = gS + gT * P  with [gS | gT] = gZ W^T;  Wt = W^T as [d, 2d] row-major; gZ is an output (wgrad reads it)
 *   wgrad : gW [2d, d] = [P + E | P * E]^T gZ, per-workgroup partials in `workspace` (arl_ngcf_wgrad_workspace_bytes) summed in a fixed order */
int arl_ngcf_dense_fwd_f32(const float *P, const float *E, const float *W, int64_t n, int64_t d, float slope, float *out, arl_stream_t stream);
int arl_ngcf_dense_dgrad_f32(const float *gOut, const float *Out, const float *P, const float *E, const float *Wt, int64_t n, int64_t d, float slope,
                             float *gZ, float *gP, float *gE, arl_stream_t stream);
int64_t arl_ngcf_wgrad_workspace_bytes(int64_t n, int64_t d);
int arl_ngcf_dense_wgrad_f32(const float *P, const float *E, const float *gZ, int64_t n, int64_t d, float *gW, void *workspace, arl_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * White-box attack primitives.
 * ---------------------------------------------------------------------------------------------- */
/* ------------------------------------------------------------------------------------------------
 * CLeaR spectral-feature-augmentation L1 term -- replaces spectral_feature_augmentation(H, 1) + F.l1_loss(SFA, H)
 * and their autograd (attack/White/CLeaR.py:98-125) for H = the rows of X taken w[row] times each
 * (H = cat(Pu[users], Pi[pos], Pi[neg]) repeats table rows; it is never materialised).
 *   r = H^T H r0;  loss_out[0] = mean| H - H r r^T/|r|^2  -  H |  over numel_h = rows(H) * d elements;
 *   G[row] (= or +=, per `accumulate`) scale * d loss / d X[row]   (G may be NULL: loss only).
 * X: [n_rows, d] row-major, d <= 256; w: [n_rows] multiplicities (0 = row not in H); r0: [d].
 * workspace: arl_sfa_workspace_bytes(n_rows, d).  Deterministic (two-stage reductions, no atomics).
 * ---------------------------------------------------------------------------------------------- */
int64_t arl_sfa_workspace_bytes(int64_t n_rows, int64_t d);
int arl_sfa_l1_fwd_bwd_f32(const float *X, const float *w, const float *r0, int64_t n_rows, int64_t d, int64_t numel_h,
                           float scale, int32_t accumulate, float *loss_out, float *G, void *workspace, arl_stream_t stream);
/* The same term cut at its two global reductions, for row sets partitioned over ranks (the user-sharded CLeaR surrogate step: the rows of H
 * generated by a rank's own users -- their rows, the targets with weight = local real users, the local negatives' histogram): the caller
 * sum-all-reduces r_out[d] after stage 1 and as_out[d + 1] = [a | S] after stage 2, then hands the reduced vectors to stage 3, which
 * writes loss_out (the GLOBAL loss when numel_h is the global element count) and this rank's rows of G.  One workspace for all stages
 * (arl_sfa_workspace_bytes).  stage1; stage2; stage3 without reductions = arl_sfa_l1_fwd_bwd_f32 (attack/White/CLeaR.py:98-125). */
int arl_sfa_stage1_f32(const float *X, const float *w, const float *r0, int64_t n_rows, int64_t d, float *r_out, void *workspace,
                       arl_stream_t stream);
int arl_sfa_stage2_f32(const float *X, const float *w, const float *r, int64_t n_rows, int64_t d, float *as_out, void *workspace,
                       arl_stream_t stream);
int arl_sfa_stage3_f32(const float *X, const float *w, const float *r0, const float *r, const float *as, int64_t n_rows, int64_t d,
                       int64_t numel_h, float scale, int32_t accumulate, float *loss_out, float *G, void *workspace, arl_stream_t stream);

/* Gradient w.r.t. adjacency values restricted to `rows`, dense over the item block (the only entries PGA
 * uses; replaces autograd.grad(Loss, sparse_norm_adj) + to_dense() + slicing, attack/White/PGA.py:117-134):
 *   out[t, j] += <dY[rows[t]], X[col_off + j]> ,  0 <= j < n_cols.   out: [n_rows_sel, n_cols]. */
int arl_sddmm_rows_dense_f32(const float *dY, const float *X, int64_t d, const int32_t *rows, int64_t n_rows_sel,
                             int64_t col_off, int64_t n_cols, float *out, arl_stream_t stream);
/* Gradient w.r.t. EVERY stored entry of the adjacency (CSR order): gval[e] += alpha * <dY[row(e)], X[col[e]]> -- the autograd of
 * torch.sparse.mm with respect to its sparse argument, as accumulated by train(requires_adjgrad=True) (recommender/LightGCN.py:41-43,58-59);
 * d % 4 == 0, d <= 256, 16-byte aligned tables.  One writer per entry (deterministic). */
int arl_sddmm_csr_f32(const int32_t *rowptr, const int32_t *col, int64_t n_rows, int64_t d, const float *dY, const float *X, float alpha,
                      float *gval, arl_stream_t stream);
/* out = alpha * (tables[0] + ... + tables[n_tables-1]), element-wise over n_elems floats (n_elems % 4 == 0, 1 <= n_tables <= 8): the mean
 * over the propagated layers (recommender/LightGCN.py:236-240: torch.stack(...).mean) in one pass.  tables: HOST array of device pointers;
 * out may alias one of them. */
int arl_tables_sum_f32(const float *const *tables, int64_t n_tables, int64_t n_elems, float alpha, float *out, arl_stream_t stream);

/* The F x I fake-user block S of the poisoned adjacency applied as two dense products (attack/White/PGA.py:118-134 multiplies by the
 * dense (U+F+I)^2 matrix; the factored operator needs only these two blocks of it).  S: [F, I] row-major, d % 4 == 0, d <= 256.
 *   rows:  Y[f, :] += alpha * rscale[f] * sum_i S[f, i] * X[i, :]      X: [I, d] (the item rows), Y: [F, d], rscale: [F] or NULL (= 1)
 *   cols:  Y[i, :] += alpha * rscale[i] * sum_f S[f, i] * Xf[f, :]     Xf: [F, d] (the fake users' rows), Y: [I, d], rscale: [I] or NULL
 * fp32 FMA chains in a fixed order (the row product is split over 64-item chunks whose partials are added in a fixed tree):
 * deterministic.  workspace (rows only): arl_fake_block_rows_workspace_bytes(F, I, d) bytes of device memory. */
int64_t arl_fake_block_rows_workspace_bytes(int64_t F, int64_t I, int64_t d);
int arl_fake_block_rows_f32(const float *S, int64_t F, int64_t I, const float *X, int64_t d, const float *rscale, float alpha, float *Y,
                            void *workspace, arl_stream_t stream);
int arl_fake_block_cols_f32(const float *S, int64_t F, int64_t I, const float *Xf, int64_t d, const float *rscale, float alpha, float *Y,
                            arl_stream_t stream);
/* Projected-gradient step on the fake-user block S [rows, cols] (attack/White/PGA.py:118-139):
 *   g = dinv_rows[r] * grad[r,c] * dinv_cols[c]   (D^-1/2 grad D^-1/2; either scale vector may be NULL = 1)
 *   g = 0 where S[r,c] == 0                       (autograd.grad on a sparse tensor only yields pattern entries)
 *   S = S - 0.2*tanh(g);  S > 1 -> 1;  S <= 0 -> 10e-8 */
int arl_pga_update_f32(float *S, const float *grad, const float *dinv_rows, const float *dinv_cols, int64_t rows,
                       int64_t cols, arl_stream_t stream);
/* Streaming scores + interacted mask + top-k, never materialising U x I.  Replaces the chunked Pu@Pi.T into a
 * host buffer, scores[nonzero] = -10e8 and torch.topk (attack/White/DLAttack.py:73-83, CLeaR.py:75-82,
 * PGA.py:100-102 with mask_rowptr == NULL).  Output sorted by descending score, ties by ascending item id.
 * k <= 128.
 * workspace == NULL: scores are the exact fp32 contraction (bitwise an fmaf chain over d).
 * workspace != NULL (arl_score_mask_topk_workspace_bytes(I, d) bytes) and d in {64, 128}: the contraction runs on
 * the 16-bit matrix path with every fp32 operand (scaled by a power of two chosen per table on the device) split in
 * two fp16 pieces, three partial products accumulated in fp32 -- scores within ~1e-6 relative of the exact ones
 * (the size of fp32 summation-order differences), 3x faster; the workspace receives the split image of Pi and the
 * two tables' largest magnitudes.  Other d ignore the workspace.  (How the three products are scheduled is internal: the
 * stream contracts the high pieces only, filters against threshold - E with E a rigorous bound on the two other products,
 * and every candidate that reaches a list is scored with all three -- the scores returned are those of the three-product form.)
 * warm_idx (optional, matrix-core path only): [U, k] DISTINCT candidate items per user, e.g. the previous call's top_idx when
 * the tables moved little; it only pre-sets each user's threshold (same result, ~6x fewer list inserts).  If a candidate has
 * become masked the threshold may exclude too much: *underflow (int32, zeroed by the caller) is then set non-zero and the
 * same call repeats the pass cold -- a second launch that every workgroup leaves at once unless the flag is set, decided on
 * the device: top_idx / top_val are valid in stream order either way and the host never has to read the flag (it stays
 * readable as a statistic).
 * item_order (optional, used on the fp16 matrix path; a permutation of [0, I)): the items are STREAMED in this order instead of table
 * order -- typically by descending row norm, so that the items most users rank high come first and the thresholds rise early (6-20 %
 * less time at 1 M x 100 K).  Masks, warm_idx, top_idx and the tie order (lower item id first) are in item ids as always: the result is
 * the one of the table order, bit for bit.
 * Early exit (fp16 matrix path): with the stream in descending row-norm order a workgroup stops streaming once, for each of its users,
 * |a_u| * (largest norm of any later item) lies below the user's current k-th best score by more than the pre-filter's slack (Cauchy-Schwarz:
 * no later item can enter a list) -- the stages skipped cannot change the result, which stays that of the full stream bit for bit.  What was
 * skipped is readable after the pass: two uint64 counters at byte arl_score_mask_topk_stats_offset(I, d) of the workspace,
 * [stream stages consumed summed over workgroups, workgroups], followed by two int32 [exit build picked, plain build picked].
 * The exit's code costs the stream loop ~5 % where nothing can be skipped (the loop has no register to spare), so it lives in a second BUILD of
 * the kernel.  exit_mode = 1: both builds are launched and the device runs one of them, chosen from the items' norm profile (the exit build iff the
 * norms at 7/8 of the stream are below half of the 4096th largest); exit_mode = 0: the plain build alone (a caller that has seen the counters report
 * nothing skipped on these tables saves the 5 %).  Results are identical in every case.
 * user_workspace (optional; arl_score_mask_topk_user_workspace_bytes(U, d) bytes, 16-byte aligned like the workspace; fp16 matrix path, d = 64): selects the SECOND FORM of the stream --
 * a wave owns 32 users and contracts 32 x 32 tiles with the items as rows, so that all of a lane's scores belong to one user and the pre-filter is one
 * subtract + one bit shift per score; 512 users share a staged tile; the sorted lists are kept in top_idx / top_val themselves while the pass runs;
 * the starting thresholds (bootstrap sample, warm-start candidates) come from two small launches of their own.  The workspace receives the split
 * image of Pu and the starting thresholds.  Candidates, exact scores, keys and tie order are those of the first form: results identical bit for bit,
 * 1.6x less time at 1 M x 100 K.  NULL: the first form.  (exit_mode = 1 launches the second form's own two builds; the device picks as before.) */
int64_t arl_score_mask_topk_workspace_bytes(int64_t I, int64_t d);
int64_t arl_score_mask_topk_user_workspace_bytes(int64_t U, int64_t d);
int64_t arl_score_mask_topk_stats_offset(int64_t I, int64_t d);
int arl_score_mask_topk_f32(const float *Pu, const float *Pi, int64_t U, int64_t I, int64_t d,
                            const int32_t *mask_rowptr, const int32_t *mask_col, int64_t k, int32_t *top_idx,
                            float *top_val, void *workspace, const int32_t *warm_idx, int32_t *underflow,
                            const int32_t *item_order, int32_t exit_mode, void *user_workspace, arl_stream_t stream);
/* CW term of the white-box attacks' surrogate loss, from the users' top-k lists (attack/White/CLeaR.py:83-95, PGA.py:104-116; the reference builds
 * three Python lists of U*T ids and gathers a [U*T, d] matrix per side).  X [n_user_rows + n_items, d] = the packed propagated tables (users first);
 * pairs = (real user u < n_real) x (target t): negative item neg(u, t) = top_idx[u][k - 1 - t] (the successive .pop()s of CLeaR.py:84-88), positive
 * = targets[t] (item ids, int64, on the device).
 *     *loss = c * sum_{u,t} <X_u, X_neg(u,t)> - <X_u, X_tg(t)>          (c = 1 / (n_real * T) gives the reference's mean)
 *     G     = d loss / d X on every row of X (rows of fake users n_real <= r < n_user_rows: zero)
 *     w_sfa (optional, [n_user_rows + n_items]) = how often each row occurs in H = cat(users, positives, negatives) of CLeaR.py:98-103: T per real
 *             user, n_real per target occurrence, the negatives' histogram -- the row weights of arl_sfa_l1_fwd_bwd_f32.
 * Deterministic: the item-side sums run in 64-bit fixed point (integer addition is associative; scale chosen on the device so that no sum can
 * overflow, resolution far below fp32 rounding of the same sum); no float atomics, no sort.  top_idx entries must be item ids in [0, n_items)
 * (the caller's own arl_score_mask_topk_f32 output).  1 <= T <= 64, k >= T, d <= 256 with d % 4 == 0, X and G 16-byte aligned,
 * n_items <= 8192 * 128 (d <= 128) or 8192 * 64 (d = 256). */
int64_t arl_cw_topk_term_workspace_bytes(int64_t n_items, int64_t d, int64_t n_real, int64_t n_targets);
int arl_cw_topk_term_f32(const float *X, int64_t n_user_rows, int64_t n_items, int64_t d, int64_t n_real, const int32_t *top_idx, int64_t k,
                         const int64_t *targets, int64_t n_targets, float c, float *G, float *loss, float *w_sfa, void *workspace,
                         arl_stream_t stream);
/* Per-row top-n -> {0,1} rows (+ indices, descending value, ties ascending column).  Replaces project()
 * (attack/White/PGA.py:153-158, CLeaR.py:161-166, DLAttack.py:127-132).  scratch: [rows*cols] fp32. */
int arl_topn_project_rows_f32(const float *M, int64_t rows, int64_t cols, int64_t n, float *out, int32_t *idx,
                              float *scratch, arl_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * All-rows InfoNCE -- replaces the nA x nV logit matrices of recommender/NCL.py:96-115 (ssl_layer_loss: F.normalize, two
 * torch.matmul against ALL users / items, exp, sum, log) and attack/White/InfoAttack.py:214-230, forward and backward, without ever
 * storing a logit.  A [nA, d] and V [nV, d] are row-NORMALISED fp32 tables (|<a, v>| <= 1 is what makes the fixed shift of the
 * log-sum-exp safe), d in {16, 32, 64, 128}.
 *   arl_nce_allrows_lse_f32   lse[b] = log sum_j exp(<a_b, v_j> / tau)
 *   arl_nce_allrows_grad_f32  dA[b] = sum_j P_bj v_j,  dV[j] = sum_b P_bj a_b  with  P_bj = exp(<a_b, v_j>/tau - lse[b])
 *                             (the caller applies 1/tau, the positive pairs' -v_idx / -a_b terms and the upstream gradient);
 *                             dA or dV may be NULL when only one side is differentiated (InfoAttack: the other view is a constant).
 *                             lse_given != 0: lse is an input (from arl_nce_allrows_lse_f32).  lse_given == 0: lse is an OUTPUT of the
 *                             dA pass (dA required), which then accumulates unnormalised exp((s - 1)/tau) terms and divides in the fold:
 *                             forward + backward in two passes over the table instead of three.
 * Exact fp32 products on the matrix cores (v_mfma_f32_16x16x4_f32); partial results are combined in a fixed order (deterministic).
 * workspace: arl_nce_allrows_workspace_bytes(nA, nV, d) bytes, 16-byte aligned like A, V and dA.
 * ---------------------------------------------------------------------------------------------- */
int64_t arl_nce_allrows_workspace_bytes(int64_t nA, int64_t nV, int64_t d);
int arl_nce_allrows_lse_f32(const float *A, int64_t nA, const float *V, int64_t nV, int64_t d, float tau, float *lse,
                            void *workspace, arl_stream_t stream);
int arl_nce_allrows_grad_f32(const float *A, int64_t nA, const float *V, int64_t nV, int64_t d, float tau, float *lse, int32_t lse_given,
                             float *dA, float *dV, void *workspace, arl_stream_t stream);

/* F.normalize(x, dim=1) of a whole table and its autograd in one pass each (the normalisations around the all-rows InfoNCE,
 * recommender/NCL.py:98-99, 110-111; d % 4 == 0, d <= 256, 16-byte aligned tables):
 *   Y = X / max(||X_r||, 1e-12), nrm[r] = max(||X_r||, 1e-12);   dX = scale * (dY - Y <Y, dY>) / nrm  (dX may alias dY;
 *   scale_dev, optional: a device scalar multiplied into scale -- the upstream gradient without a host round trip). */
int arl_normalize_rows_f32(const float *X, int64_t n, int64_t d, float *Y, float *nrm, arl_stream_t stream);
int arl_normalize_rows_bwd_f32(const float *Y, const float *nrm, const float *dY, int64_t n, int64_t d, float scale, const float *scale_dev,
                               float *dX, arl_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Item-table exchange of the user-sharded step (no reference counterpart: main.py:19 pins one device).
 * One process per GPU.  arl_comm_unique_id (rank 0) -> the 128 bytes travel to every rank by any side channel (torch.distributed
 * broadcast) -> arl_comm_init on every rank.  arl_allreduce_item_f32 sum-all-reduces buf[0, n_elems) IN PLACE, asynchronously on `stream`,
 * as a direct reduce-scatter + all-gather over the point-to-point xGMI links (every rank owns one shard, sums the P partials of it in rank
 * order -- deterministic, one writer per element -- and sends the result to every peer), cut into n_chunks chunks (1..64); workspace =
 * arl_allreduce_item_workspace_bytes(n_elems, world, n_chunks), 16-byte aligned like buf.  RCCL is bound at run time (dlopen of the
 * instance the process already has, or `rccl_path`); without it the arl_comm_* calls return ARL_E_ARG.  Returns 1000 + ncclResult_t on an
 * RCCL error.  arl_item_exchange_range: the [lo, hi) element range of chunk `chunk` of shard `shard` (host-side arithmetic, for tests).
 * ---------------------------------------------------------------------------------------------- */
typedef void *arl_comm_t;
int arl_comm_load(const char *rccl_path);
int arl_comm_unique_id(void *id128);
/* device: HIP device index the communicator binds to (made current for the call and restored); < 0 = the calling thread's current device */
int arl_comm_init(const void *id128, int64_t rank, int64_t world, int64_t device, arl_comm_t *out);
int arl_comm_destroy(arl_comm_t comm);
int arl_item_exchange_range(int64_t n_elems, int64_t world, int64_t n_chunks, int64_t shard, int64_t chunk, int64_t *lo, int64_t *hi);
int64_t arl_allreduce_item_workspace_bytes(int64_t n_elems, int64_t world, int64_t n_chunks);
int arl_allreduce_item_f32(arl_comm_t comm, float *buf, int64_t n_elems, int64_t n_chunks, void *workspace, arl_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ARLIB_AMD_H */
