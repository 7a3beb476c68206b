"""ctypes binding of libdbnet_hip.so (the C ABI declared in include/dbnet_hip.h).

The product path has no CPU or PyTorch-op fallback: if the HIP library is
missing or fails to load, importing a compute entry point raises.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# DBN_LIB_PATH: a test vehicle built from the same sources (make RACE=1 -> libdbnet_hip_race.so, csrc/common.h); the default is the product
LIB_PATH = os.environ.get('DBN_LIB_PATH') or os.path.join(_HERE, 'libdbnet_hip.so')
CSRC = os.path.join(_HERE, 'csrc')

_P, _I, _L, _F = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float

# name -> argument kinds (p = device/host pointer, i = int, l = long, f = float); all return int
SIGNATURES = {
    'dbn_pack_weights': 'piiiiiipp',
    'dbn_igemm_panel_floats': 'iiiiii',
    'dbn_igemm_packed_floats': 'ii',
    'dbn_igemm_f32': 'pppp' + 'i' * 14 + 'p',
    'dbn_igemm_tile_config': 'ii',
    'dbn_igemm_tile_config_ns': 'iii',
    'dbn_conv_bn_ws_floats': 'iiiiii',
    'dbn_pack_weights_batched': 'piip',
    'dbn_igemm_splitk_plan': 'iiii',
    'dbn_igemm_splitk_plan_ns': 'iiiii',
    'dbn_igemm_splitk_f32': 'pppp' + 'i' * 16 + 'pp',
    'dbn_deform_im2col': 'ppp' + 'i' * 11 + 'p',
    'dbn_deform_col2im': 'ppppp' + 'i' + 'p' + 'i' * 11 + 'p',
    'dbn_deform_col2im_ws_bytes': 'i' * 8,
    'dbn_permute_weight': 'ppiiiifp',
    'dbn_binarize_u8': 'piiiifpp',
    'dbn_box_scores': 'piipiipp',
    'dbn_pyramid_conv_ws_floats': 'iiii',
    'dbn_pyramid_conv_f32': 'p' * 10 + 'i' * 7 + 'pp' + 'ff' + 'ppppppp' + 'p',
    'dbn_conv_bn_f32': 'pppp' + 'i' * 15 + 'pp' + 'ff' + 'ppppppp' + 'p',
    'dbn_pack_weights_bf16s': 'piiiiiiipp',
    'dbn_igemm_bf16s_panel_floats': 'iiiiiii',
    'dbn_igemm_bf16s': 'pppp' + 'i' * 15 + 'p',
    'dbn_wgrad_bf16s': 'pppp' + 'i' * 12 + 'fip',
    'dbn_wgrad_splitk': 'iiiiiii',
    'dbn_wgrad_slab_floats': 'iiiiiii',
    'dbn_wgrad_splitk_hw': 'iiiiiiiii',
    'dbn_wgrad_slab_floats_hw': 'iiiiiiiiii',
    'dbn_set_index_limits': 'lll',
    'dbn_wgrad_f32': 'pppp' + 'i' * 12 + 'fp',
    'dbn_reduce_ws_floats': 'i',
    'dbn_bn_train_stats': 'piippffppppppp' + 'p',
    'dbn_bn_eval_coef': 'ippppfppp',
    'dbn_bn_apply': 'pppppppliip',
    'dbn_bn_backward': 'pppppppp' + 'pp' + 'i' + 'pp' + 'ii' + 'f' + 'pp',
    'dbn_bn_backward_from_sums': 'ppppppppp' + 'pp' + 'i' + 'pp' + 'ii' + 'f' + 'pp',
    'dbn_bn_backward_ex': 'ppppppppp' + 'pp' + 'i' + 'ppp' + 'ii' + 'f' + 'pp',
    'dbn_col_sum': 'piipfpp',
    'dbn_bnrelu_maxpool_fwd': 'ppppiiiip',
    'dbn_bnrelu_maxpool_bwd': 'ppppppiiiip',
    'dbn_nearest_up_fwd': 'ppp' + 'i' * 8 + 'p',
    'dbn_nearest_up_bwd': 'pp' + 'i' * 9 + 'p',
    'dbn_bilinear_fwd': 'ppliiiip',
    'dbn_bilinear_bwd': 'ppliiiip',
    'dbn_fpn_combine_weights': 'piiiipp',
    'dbn_fpn_scatter_wgrad': 'ppppiipp',
    'dbn_nchw3_to_nhwc4': 'ppiiip',
    'dbn_add_inplace': 'pplp',
    'dbn_head_tail_fwd': 'p' * 11 + 'iiii' + 'f' + 'p',
    'dbn_head_tail_bwd_ws_floats': '',
    'dbn_head_tail_bwd': 'p' * 21 + 'iiii' + 'ff' + 'pp',
    'dbn_db_loss_ws_bytes': '',
    'dbn_db_loss_fwd': 'pp' + 'iiii' + 'ffff' + 'pppp',
    'dbn_db_loss_sum_fwd': 'pp' + 'iiii' + 'ffff' + 'pppp',
    'dbn_db_loss_frac_fwd': 'pp' + 'iiii' + 'ffff' + 'i' + 'pppp',
    'dbn_db_loss_bwd': 'pppp' + 'ff' + 'iiii' + 'pp',
    'dbn_db_loss_ohem_ws_bytes': 'iii',
    'dbn_db_loss_ohem_fwd': 'pp' + 'iiii' + 'ffff' + 'pppp',
    'dbn_db_loss_ohem_bwd': 'ppppp' + 'ff' + 'iiii' + 'pp',
    'dbn_pixel_confusion': 'plppiiifpp',
    'dbn_adam_step': 'pppp' + 'l' + 'ffff' + 'i' + 'f' + 'p',
    'dbn_clock_probe': 'pip',
    'dbn_has_experiments': '',
    'dbn_wall_clock_khz': '',
}
# `_t` forms: activation storage type first (0 fp32, 1 bf16, 2 fp16), see include/dbnet_hip.h
SIGNATURES.update({
    'dbn_igemm_t': 'ii' + 'pppp' + 'i' * 14 + 'i' + 'p' + 'p',
    'dbn_conv_bn_t': 'i' + SIGNATURES['dbn_conv_bn_f32'],
    'dbn_pyramid_conv_t': 'i' + SIGNATURES['dbn_pyramid_conv_f32'],
    'dbn_pyramid_conv_from_t': 'ii' + SIGNATURES['dbn_pyramid_conv_f32'],
    'dbn_igemm_bn_rows': 'i' * 15,
    'dbn_igemm_bnsums_t': 'ii' + 'pppp' + 'i' * 14 + 'ppppppp' + 'pppp' + 'p' + 'p',
    'dbn_igemm_bn_final_counters': 'ii',
    'dbn_igemm_bn_final_group_floats': 'ii',
    'dbn_wgrad_t': 'ii' + 'pppp' + 'i' * 12 + 'f' + 'p',
    'dbn_wgrad_phase_t': 'iii' + 'pppp' + 'i' * 12 + 'f' + 'p',
    'dbn_wgrad_reduce_describe': 'ii' + 'pppp' + 'i' * 12 + 'f' + 'p',
    'dbn_wgrad_reduce_many': 'ppiiip',
    'dbn_wgrad_tile_config': 'ii',
    'dbn_set_wgrad_variant': 'i',
    'dbn_set_convt_kernel': 'i',
    'dbn_igemm_splitk_slab_floats': 'iiiii',
    'dbn_wgrad_kernel_config': 'iiiii',
    'dbn_wgrad_kernel_config_hw': 'i' * 12,
    'dbn_set_patch_conv': 'i',
    'dbn_set_wres16': 'i',
    'dbn_set_pyramid_wide': 'i',
    'dbn_pyramid_wide_would_run': 'iiiiii',
    'dbn_conv_bn_set_final': 'pp',
    'dbn_conv_bn_final_group_doubles': 'ii',
    'dbn_wres16_would_run': 'iiiiiiiii',
    'dbn_mfma_sustained': 'ippip',
    'dbn_mfma_sustained_flops': 'ii',
    'dbn_set_stagger': 'i',
    'dbn_winograd_eligible': 'iiiii',
    'dbn_winograd_panel_floats': 'ii',
    'dbn_winograd_pack': 'piiiipp',
    'dbn_winograd_rows': 'iii',
    'dbn_winograd_pack_batched': 'pip',
    'dbn_winograd_dgrad_bnsums_f32': 'ppp' + 'iiiiii' + 'pppp' + 'ppp' + 'pppp' + 'p' + 'p',
    'dbn_winograd_ws_floats': 'iiii',
    'dbn_winograd_wgrad_eligible': 'iiiiii',
    'dbn_winograd_wgrad_linear': 'ii',
    'dbn_winograd_wgrad_slab_floats': 'iiiii',
    'dbn_winograd_wgrad_f32': 'ipppppp' + 'iiiiii' + 'fp',
    'dbn_winograd_conv_bn_act_f32': 'pppppp' + 'iiiii' + 'pp' + 'ff' + 'ppppppp' + 'p',
    'dbn_winograd_conv_bn_f32': 'pppp' + 'iiiii' + 'pp' + 'ff' + 'ppppppp' + 'p',
    'dbn_set_phase_priority': 'i',
    'dbn_set_winograd_persistent': 'i',
    'dbn_set_winograd_stagger': 'i',
    'dbn_set_winograd_blocks_per_barrier': 'i',
    'dbn_fold_bn_eval': 'pilppppp' + 'f' + 'ppp',
    'dbn_igemm_act_t': 'ii' + 'pppp' + 'i' + 'p' + 'i' * 13 + 'p',
    'dbn_winograd_conv_act_f32': 'pppp' + 'i' + 'p' + 'iiiii' + 'p',
    'dbn_pyramid_conv_act_t': 'ii' + 'p' * 9 + 'i' + 'p' + 'i' * 6 + 'p',
    'dbn_set_trace': 'pl',
    'dbn_igemm_kernel_config': 'i' * 16,
    'dbn_split3': 'pplp',
    'dbn_deform_im2col_t': 'i' + SIGNATURES['dbn_deform_im2col'],
    'dbn_deform_col2im_t': 'i' + SIGNATURES['dbn_deform_col2im'],
    'dbn_deform_col2im_gather_t': 'i' + SIGNATURES['dbn_deform_col2im'],
    'dbn_deform_col2im_gather_ws_bytes': 'iii',
    'dbn_deform_offset_absmax_t': 'iplpp',
    'dbn_pw16_eligible': 'iiiiii',
    'dbn_pw16_panel_bytes': '',
    'dbn_pw16_pack': 'ipipp',
    'dbn_pw16_act_t': 'ipppipiiiip',
    'dbn_set_head_tail_wide': 'i',
    'dbn_stem16_pool_eligible': 'iiii',
    'dbn_stem16_conv_bn_relu_pool_t': 'ipppppiiip',
    'dbn_head16_eligible': 'iiii',
    'dbn_head16_tail_eval_t': 'i' + 'p' * 15 + 'iii' + 'p',
    'dbn_cast_f32': 'ipplp',
    'dbn_pack_weights_t': 'ip' + 'i' * 7 + 'pp',
    'dbn_igemm_panel_floats_t': 'i' * 8,
    'dbn_bn_train_stats_t': 'i' + SIGNATURES['dbn_bn_train_stats'],
    'dbn_bn_apply_t': 'i' + SIGNATURES['dbn_bn_apply'],
    'dbn_bn_backward_t': 'ipi' + SIGNATURES['dbn_bn_backward_ex'][1:],
    'dbn_col_sum_t': 'i' + SIGNATURES['dbn_col_sum'],
    'dbn_bnrelu_maxpool_fwd_t': 'i' + SIGNATURES['dbn_bnrelu_maxpool_fwd'],
    'dbn_bnrelu_maxpool_bwd_t': 'i' + SIGNATURES['dbn_bnrelu_maxpool_bwd'][:-1] + 'pppp',
    'dbn_maxpool_bwd_parts': 'iiii',
    'dbn_bnrelu_maxpool_fwd_arg_t': 'ipppppp' + 'iiii' + 'p',
    'dbn_maxpool_bn_backward_ws_floats': 'iiii',
    'dbn_maxpool_bn_backward_t': 'i' + 'p' * 10 + 'iiii' + 'f' + 'pp',
    'dbn_nearest_up_fwd_t': 'i' + SIGNATURES['dbn_nearest_up_fwd'],
    'dbn_nearest_up_bwd_t': 'i' + SIGNATURES['dbn_nearest_up_bwd'],
    'dbn_nchw3_to_nhwc4_t': 'i' + SIGNATURES['dbn_nchw3_to_nhwc4'],
    'dbn_nchw3_to_nhwc4_packed_t': 'i' + SIGNATURES['dbn_nchw3_to_nhwc4'],
    'dbn_nchw3_to_nhwc16_and_4_t': 'ipppiiip',
    'dbn_convt16_rows': '',
    'dbn_convt16_panel_bytes': '',
    'dbn_convt16_eligible': 'iiiiii',
    'dbn_convt16_pack': 'ippp',
    'dbn_convt16_bn_t': 'ippppiii' + 'pp' + 'ff' + 'ppppppp' + 'p',
    'dbn_stem16_padded_h': 'i',
    'dbn_stem16_padded_w': 'i',
    'dbn_stem16_rows': '',
    'dbn_stem16_panel_bytes': '',
    'dbn_stem16_eligible': 'iiii',
    'dbn_stem16_pack': 'ippp',
    'dbn_nchw3_to_padded4_t': 'ipppiiip',
    'dbn_stem16_conv_bn_t': 'ipppiii' + 'pp' + 'ff' + 'ppppppp' + 'p',
    'dbn_head_tail_fwd_t': 'i' + SIGNATURES['dbn_head_tail_fwd'],
    'dbn_head_tail_bwd_t': 'i' + SIGNATURES['dbn_head_tail_bwd'],
})
LONG_RETURN = {'dbn_mfma_sustained_flops', 'dbn_conv_bn_final_group_doubles', 'dbn_maxpool_bn_backward_ws_floats', 'dbn_pw16_panel_bytes', 'dbn_stem16_panel_bytes', 'dbn_convt16_panel_bytes', 'dbn_winograd_panel_floats', 'dbn_winograd_wgrad_slab_floats', 'dbn_winograd_ws_floats', 'dbn_igemm_splitk_slab_floats', 'dbn_deform_col2im_ws_bytes', 'dbn_deform_col2im_gather_ws_bytes', 'dbn_igemm_bn_final_counters', 'dbn_igemm_bn_final_group_floats', 'dbn_igemm_panel_floats_t', 'dbn_wgrad_slab_floats_hw', 'dbn_wgrad_slab_floats', 'dbn_igemm_panel_floats', 'dbn_igemm_bf16s_panel_floats', 'dbn_db_loss_ohem_ws_bytes', 'dbn_conv_bn_ws_floats', 'dbn_pyramid_conv_ws_floats'}
_KIND = {'p': _P, 'i': _I, 'l': _L, 'f': _F}


class HipLibraryError(RuntimeError):
    pass


class WgradReduceJob(ctypes.Structure):
    """dbn_wgrad_reduce_job of include/dbnet_hip.h: one layer's slab reduction, for dbn_wgrad_reduce_many."""
    _fields_ = ([('slab', _P), ('grad', _P)] + [(f, _I) for f in ('splitk', 'O', 'J', 'Jp', 'BM', 'BN', 'Cb', 'I', 'RS', 'G', 'natural', 'blocks')]
                + [('scale', _F), ('smem_bytes', _I)])


class BnbFinal(ctypes.Structure):
    """dbn_bnb_final of include/dbnet_hip.h: the in-kernel finalize of a data gradient's BatchNorm-backward sums."""
    _fields_ = [('counters', _P), ('group', _P), ('c1c2', _P), ('dgamma', _P), ('dbeta', _P), ('c1c2_2', _P), ('dgamma_2', _P),
                ('dbeta_2', _P), ('grad_scale', _F)]


def build(verbose=False):
    """Compile csrc/*.hip for gfx950 into libdbnet_hip.so (hipcc cross-compiles without a GPU)."""
    res = subprocess.run(['make', '-C', CSRC, '-j8'], capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout[-4000:])
        print(res.stderr[-4000:])
    if res.returncode != 0:
        raise HipLibraryError('building libdbnet_hip.so failed (see output above)')
    return LIB_PATH


def source_stamp():
    """sha256 over the kernel sources (csrc/*.hip, *.h, Makefile, sorted by name): recorded beside profile summaries
    (tools/pmc_traffic.py) so that bench.py can tell whether a committed per-kernel figure still describes these kernels."""
    import hashlib
    h = hashlib.sha256()
    for name in sorted(os.listdir(CSRC)):
        if name.endswith(('.hip', '.h')) or name == 'Makefile':
            h.update(name.encode())
            with open(os.path.join(CSRC, name), 'rb') as f:
                h.update(f.read())
    return h.hexdigest()[:16]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryError('%s not found: run `python -c "import __graft_entry__ as g; g.build()"` '
                                  '(there is no CPU fallback for the DBNet hot path)' % LIB_PATH)
        try:
            l = ctypes.CDLL(LIB_PATH)
        except OSError as e:
            raise HipLibraryError('cannot load %s: %s' % (LIB_PATH, e))
        for name, sig in SIGNATURES.items():
            fn = getattr(l, name)  # AttributeError if the symbol is missing
            fn.argtypes = [_KIND[k] for k in sig]
            fn.restype = _L if name in LONG_RETURN else _I
        _lib = l
    return _lib


def check(rc, what=''):
    if rc != 0:
        if rc == 1:
            raise RuntimeError('libdbnet_hip: invalid argument in %s' % what)
        raise RuntimeError('libdbnet_hip: %s failed with hipError %d' % (what, rc - 1000))
