"""ctypes loader for the in-tree libarlib_amd.so (C ABI declared in include/arlib_amd.h).

There is no CPU fallback: if the HIP library is missing every op raises.  Build it with
`python -c "import __graft_entry__ as g; g.build()"` or `make -C arlib_amd/csrc`.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('ARLIB_AMD_LIB') or os.path.join(_HERE, 'lib', 'libarlib_amd.so')      # override: developer builds (e.g. `make prof`)
ABI_VERSION = 24
_lib = None


class ArlError(RuntimeError):
    pass


class arl_csr(C.Structure):
    _fields_ = [('n_rows', C.c_int64), ('nnz', C.c_int64), ('rowptr', C.c_void_p), ('col', C.c_void_p), ('val', C.c_void_p),
                ('chunk', C.c_int32), ('n_chunks', C.c_int64), ('chunk_row', C.c_void_p), ('chunk_begin', C.c_void_p),
                ('chunk_end', C.c_void_p), ('n_long', C.c_int64), ('long_row', C.c_void_p), ('long_first', C.c_void_p),
                ('long_count', C.c_void_p), ('partial', C.c_void_p), ('row_tasks', C.c_void_p)]


class arl_blocked(C.Structure):
    _fields_ = [('n_waves', C.c_int64), ('rows_per_wave', C.c_int64), ('loads_in_flight', C.c_int64), ('wave_ptr', C.c_void_p), ('wave_rows', C.c_void_p), ('rec_col', C.c_void_p),
                ('rec_val', C.c_void_p), ('n_split', C.c_int64), ('split_row', C.c_void_p), ('split_first', C.c_void_p), ('split_count', C.c_void_p),
                ('partial', C.c_void_p), ('waves_per_group', C.c_int64)]


class arl_tiled(C.Structure):
    _fields_ = [('n_sweeps', C.c_int64), ('n_slots', C.c_int64), ('cap', C.c_int64), ('n_cb', C.c_int64), ('nnz', C.c_int64),
                ('n_groups', C.c_int64), ('bin_rows', C.c_void_p), ('seg_ptr', C.c_void_p), ('e_col', C.c_void_p), ('e_val', C.c_void_p),
                ('e_row', C.c_void_p)]


_vp, _i64, _i32, _f = C.c_void_p, C.c_int64, C.c_int32, C.c_float
_SIGS = {
    'arl_abi_version': (C.c_int, []),
    'arl_mt_seed': (C.c_int, [_vp, _vp, _i64]),
    'arl_sampler_shuffle': (C.c_int, [_vp, _vp, _i64]),
    'arl_mt_sample_range': (C.c_int, [_vp, _i64, _i64, C.c_int32, _vp, _vp]),
    'arl_sampler_next_batch': (C.c_int, [_vp, _vp, _i64, _i64, _i32, _vp, _vp, _i64, _vp, _vp, _vp]),
    'arl_norm_adj_values_f32': (C.c_int, [_i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    'arl_norm_adj_values_coo_f32': (C.c_int, [_i64, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    'arl_norm_vals_coo_f32': (C.c_int, [_vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    'arl_spmm_csr_f32': (C.c_int, [C.POINTER(arl_csr), _vp, _i64, _f, _f, _vp, _vp, _vp]),
    'arl_spmm_csr_layersum_f32': (C.c_int, [C.POINTER(arl_csr), _vp, _i64, _vp, _vp, _vp, _vp]),
    'arl_spmm_csr_adam_f32': (C.c_int, [C.POINTER(arl_csr), _vp, _i64, _f, _f, _vp, _vp, _vp, _vp, _vp, _f, _f, _f, _f, _i64, _vp]),
    'arl_spmm_blocked_f32': (C.c_int, [C.POINTER(arl_blocked), _vp, _i64, _f, _f, _vp, _vp, _vp, _vp]),
    'arl_spmm_csr_rscale_f32': (C.c_int, [C.POINTER(arl_csr), _vp, _i64, _vp, _f, _f, _vp, _vp, _vp]),
    'arl_spmm_blocked_rscale_f32': (C.c_int, [C.POINTER(arl_blocked), _vp, _i64, _vp, _f, _f, _vp, _vp, _vp]),
    'arl_spmm_blocked_layersum_f32': (C.c_int, [C.POINTER(arl_blocked), _vp, _i64, _vp, _vp, _vp, _vp]),
    'arl_spmm_blocked_adam_f32': (C.c_int, [C.POINTER(arl_blocked), _vp, _i64, _f, _f, _vp, _vp, _vp, _vp, _vp, _f, _f, _f, _f, _i64, _vp]),
    'arl_lpt_deal': (C.c_int, [_i64, _vp, _i64, _i64, _vp, _vp]),
    'arl_syn_v1_pairs': (_i64, [_i64, _i64, C.c_double, C.c_uint64, C.c_double, _i64, _i64, _vp, _i64]),
    'arl_graph_digest': (C.c_uint64, [_vp, _i64]),
    'arl_spmm_tiled_f32': (C.c_int, [C.POINTER(arl_tiled), _vp, _i64, _f, _f, _vp, _vp, _vp, _vp]),
    'arl_spmm_tiled_adam_f32': (C.c_int, [C.POINTER(arl_tiled), _vp, _i64, _f, _f, _vp, _vp, _vp, _vp, _vp, _f, _f, _f, _f, _i64, _vp]),
    'arl_spmm_csr_flagged_f32': (C.c_int, [C.POINTER(arl_csr), _vp, _i64, _vp, _f, _f, _vp, _vp, _vp, _vp]),
    'arl_spmm_csr_rows_workspace_bytes': (_i64, [_i64, _i64, _i64]),
    'arl_spmm_csr_rows_f32': (C.c_int, [C.POINTER(arl_csr), _vp, _i64, _vp, _i64, _i64, _vp, _i64, _f, _vp, _vp, _vp, _vp]),
    'arl_mark_rows_u8': (C.c_int, [_vp, _vp, _i64, _i32, _vp]),
    'arl_mark_rows_bits_u32': (C.c_int, [_vp, _vp, _i64, _i32, _vp]),
    'arl_zero_rows_f32': (C.c_int, [_vp, _vp, _i64, _i64, _vp]),
    'arl_batch_rows_set_f32': (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _vp, _f, _vp, _vp, _vp]),
    'arl_batch_rows_clear_f32': (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp]),
    'arl_bpr_l2_workspace_bytes': (_i64, [_i64]),
    'arl_bpr_l2_fwd_bwd_f32': (C.c_int, [_vp, _i64, _i64, _vp, _vp, _vp, _i64, _f, _f, _vp, _vp, _vp, _i32, _vp]),
    'arl_bpr_l2_partial_f32': (C.c_int, [_vp, _i64, _i64, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp]),
    'arl_bpr_l2_backward_f32': (C.c_int, [_vp, _i64, _i64, _vp, _vp, _vp, _i64, _f, _f, _vp, _vp, _vp, _vp]),
    'arl_adam_dense_f32': (C.c_int, [_vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _i64, _vp]),
    'arl_sgd_dense_f32': (C.c_int, [_vp, _vp, _i64, _f, _vp]),
    'arl_gather_rows_f32': (C.c_int, [_vp, _vp, _i64, _i64, _vp, _vp]),
    'arl_scatter_add_rows_f32': (C.c_int, [_vp, _vp, _i64, _i64, _vp, _f, _vp]),
    'arl_rows_axpy_unique_f32': (C.c_int, [_vp, _vp, _vp, _i64, _i64, _f, _vp, _vp]),
    'arl_shard_batch_prep_i32': (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp]),
    'arl_infonce_workspace_bytes': (_i64, [_i64, _i64]),
    'arl_infonce_fwd_bwd_f32': (C.c_int, [_vp, _vp, _i64, _i64, _f, _f, _vp, _vp, _vp, _vp, _vp]),
    'arl_simgcl_perturb_f32': (C.c_int, [_vp, _vp, _i64, _i64, _f, _vp]),
    'arl_simgcl_perturb_rng_f32': (C.c_int, [_vp, _vp, _i64, _i64, _vp, _f, C.c_uint64, C.c_uint64, _vp]),
    'arl_ngcf_combine_f32': (C.c_int, [_vp, _vp, _i64, _i64, _vp, _vp]),
    'arl_ngcf_act_f32': (C.c_int, [_vp, _vp, _i64, _i64, _f, _vp]),
    'arl_ngcf_act_bwd_f32': (C.c_int, [_vp, _vp, _i64, _i64, _f, _vp, _vp]),
    'arl_ngcf_combine_bwd_f32': (C.c_int, [_vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp]),
    'arl_ngcf_dense_fwd_f32': (C.c_int, [_vp, _vp, _vp, _i64, _i64, _f, _vp, _vp]),
    'arl_ngcf_dense_dgrad_f32': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _f, _vp, _vp, _vp, _vp]),
    'arl_ngcf_wgrad_workspace_bytes': (_i64, [_i64, _i64]),
    'arl_ngcf_dense_wgrad_f32': (C.c_int, [_vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp]),
    'arl_sfa_workspace_bytes': (_i64, [_i64, _i64]),
    'arl_sfa_l1_fwd_bwd_f32': (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, _f, C.c_int32, _vp, _vp, _vp, _vp]),
    'arl_sfa_stage1_f32': (C.c_int, [_vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp]),
    'arl_sfa_stage2_f32': (C.c_int, [_vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp]),
    'arl_sfa_stage3_f32': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _f, C.c_int32, _vp, _vp, _vp, _vp]),
    'arl_sddmm_csr_f32': (C.c_int, [_vp, _vp, _i64, _i64, _vp, _vp, _f, _vp, _vp]),
    'arl_sddmm_rows_dense_f32': (C.c_int, [_vp, _vp, _i64, _vp, _i64, _i64, _i64, _vp, _vp]),
    'arl_pga_update_f32': (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _vp]),
    'arl_tables_sum_f32': (C.c_int, [_vp, _i64, _i64, C.c_float, _vp, _vp]),
    'arl_fake_block_rows_workspace_bytes': (_i64, [_i64, _i64, _i64]),
    'arl_fake_block_rows_f32': (C.c_int, [_vp, _i64, _i64, _vp, _i64, _vp, C.c_float, _vp, _vp, _vp]),
    'arl_fake_block_cols_f32': (C.c_int, [_vp, _i64, _i64, _vp, _i64, _vp, C.c_float, _vp, _vp]),
    'arl_cw_topk_term_workspace_bytes': (_i64, [_i64, _i64, _i64, _i64]),
    'arl_cw_topk_term_f32': (C.c_int, [_vp, _i64, _i64, _i64, _i64, _vp, _i64, _vp, _i64, C.c_float, _vp, _vp, _vp, _vp, _vp]),
    'arl_score_mask_topk_workspace_bytes': (_i64, [_i64, _i64]),
    'arl_score_mask_topk_stats_offset': (_i64, [_i64, _i64]),
    'arl_score_mask_topk_user_workspace_bytes': (_i64, [_i64, _i64]),
    'arl_score_mask_topk_f32': (C.c_int, [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int32, _vp, _vp]),
    'arl_normalize_rows_f32': (C.c_int, [_vp, _i64, _i64, _vp, _vp, _vp]),
    'arl_normalize_rows_bwd_f32': (C.c_int, [_vp, _vp, _vp, _i64, _i64, _f, _vp, _vp, _vp]),
    'arl_nce_allrows_workspace_bytes': (_i64, [_i64, _i64, _i64]),
    'arl_nce_allrows_lse_f32': (C.c_int, [_vp, _i64, _vp, _i64, _i64, _f, _vp, _vp, _vp]),
    'arl_nce_allrows_grad_f32': (C.c_int, [_vp, _i64, _vp, _i64, _i64, _f, _vp, C.c_int32, _vp, _vp, _vp, _vp]),
    'arl_comm_load': (C.c_int, [C.c_char_p]),
    'arl_comm_unique_id': (C.c_int, [_vp]),
    'arl_comm_init': (C.c_int, [_vp, _i64, _i64, _i64, C.POINTER(C.c_void_p)]),
    'arl_comm_destroy': (C.c_int, [_vp]),
    'arl_item_exchange_range': (C.c_int, [_i64, _i64, _i64, _i64, _i64, C.POINTER(_i64), C.POINTER(_i64)]),
    'arl_allreduce_item_workspace_bytes': (_i64, [_i64, _i64, _i64]),
    'arl_allreduce_item_f32': (C.c_int, [_vp, _vp, _i64, _i64, _vp, _vp]),
    'arl_topn_project_rows_f32': (C.c_int, [_vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp]),
}
EXPORTS = tuple(_SIGS)


def lib():
    """Load libarlib_amd.so or fail loudly."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ArlError('arlib_amd: %s is missing -- the HIP extension is the product and there is no CPU fallback. '
                           'Build it with `make -C arlib_amd/csrc` (hipcc --offload-arch=gfx950).' % LIB_PATH)
        # torch bundles its own libamdhip64.so (SONAME libamdhip64.so.7).  It must be loaded first so that our NEEDED
        # entry resolves to that same runtime instance: two HIP runtimes in one process do not share device state.
        import torch  # noqa: F401
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)            # AttributeError here = ABI mismatch: fail loudly
            fn.restype, fn.argtypes = res, args
        if l.arl_abi_version() != ABI_VERSION:
            raise ArlError('arlib_amd: ABI version mismatch (lib %d, python %d): rebuild' % (l.arl_abi_version(), ABI_VERSION))
        _lib = l
    return _lib


_ERR = {-1: 'ARL_E_NULL (required pointer is NULL)', -2: 'ARL_E_DIM (unsupported embedding size/shape)',
        -3: 'ARL_E_RANGE (size outside the int32 index range)', -4: 'ARL_E_ARG (inconsistent arguments)'}


def check(rc, what):
    if rc != 0:
        raise ArlError('%s failed: %s' % (what, _ERR.get(rc, 'hipError_t %d' % rc)))
