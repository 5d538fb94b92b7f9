"""ctypes binding of libmelogan_hip.so (include/melo_gan_hip.h).  Fails loudly if the library
is absent -- there is deliberately no fallback path."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmelogan_hip.so")
if os.environ.get("MELO_LIB_VARIANT"):      # A/B measurements: a tools/build_variant.sh build of the same sources
    LIB_PATH = os.path.join(os.path.dirname(_HERE), "tools", "_build", "libmelogan_" + os.environ["MELO_LIB_VARIANT"] + ".so")

ACT_NONE, ACT_RELU, ACT_LRELU, ACT_GELU, ACT_TANH = 0, 1, 2, 3, 4

vp, i32, i64, f32, sz = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_size_t


class Epilogue(C.Structure):
    _fields_ = [("bias", vp), ("scale", vp), ("shift", vp), ("zout", vp), ("act", i32),
                ("gref", vp), ("gact", i32), ("emul", vp), ("gscale", vp), ("accumulate", i32)]


class EpilogueBf16(C.Structure):
    _fields_ = [("scale", vp), ("shift", vp), ("zout", vp), ("act", i32), ("gref", vp), ("gact", i32), ("gscale", vp),
                ("accumulate", i32)]


class Conv16Extra(C.Structure):
    _fields_ = [("part", vp), ("pool", vp), ("pool_scale", f32), ("y_perm", i32), ("mix_real", vp), ("mix_alpha", vp),
                ("mix_out", vp), ("mix_rows", i32), ("bnb_a", vp), ("bnb_z", vp), ("bnb_mean", vp), ("bnb_invstd", vp),
                ("bnb_part", vp), ("bnb_act", i32)]


class ChainOp(C.Structure):
    _fields_ = [("kind", i32), ("a", i32), ("b", i32), ("n0", i32), ("n1", i32), ("act", i32), ("i0", i32), ("i1", i32), ("f0", f32),
                ("p0", vp), ("ld0", i64), ("p1", vp), ("ld1", i64), ("p2", vp), ("ld2", i64), ("p3", vp), ("ld3", i64),
                ("q0", vp), ("lq0", i64), ("q1", vp), ("lq1", i64), ("t0", vp)]


CH_LOAD, CH_STORE, CH_LAYERNORM, CH_LIN_FWD, CH_LIN_DGRAD, CH_SOFTMAX_CE, CH_DHEAD, CH_MEAN_T, CH_COPY = 1, 2, 3, 4, 5, 6, 7, 8, 9
CHAIN_MAX_OPS, CHAIN_SLOTS, CHAIN_MAX_VEC = 16, 6, 512


class StageJob(C.Structure):
    _fields_ = [("src", vp), ("dst", vp), ("idx", vp), ("row_bytes", i64), ("src_rows", i64), ("dst_pitch", i64), ("rows", i64)]


MAX_STAGE_JOBS = 8


class WinoWJob(C.Structure):
    _fields_ = [("w", vp), ("wt", vp), ("N", i32), ("Cin", i32), ("w_sn", i64), ("w_sc", i64), ("flip", i32)]


MAX_WINO_WJOBS = 8


class SnJob(C.Structure):
    _fields_ = [("w_orig", vp), ("w_eff", vp), ("u", vp), ("v", vp), ("sigma", vp), ("dw", vp), ("rows", i32), ("cols", i32)]


MAX_SN_JOBS = 8


class WgradJob(C.Structure):
    _fields_ = [("s0", vp), ("l0", vp), ("nb0", i32), ("s1", vp), ("l1", vp), ("nb1", i32),
                ("out", vp), ("bias_out", vp), ("bias_from", i32), ("Ts", i32), ("Tl", i32), ("A", i32), ("Bc", i32)]


MAX_WGRAD_JOBS = 8


class WqEntry(C.Structure):
    _fields_ = [("start", i64), ("N", i32), ("Cc", i32), ("K", i32), ("w_sn", i32), ("w_sc", i32), ("dst", vp)]


MAX_WQ_ENTRIES = 8

# name -> (restype, argtypes); this table is also what tests/test_abi.py checks against the header.
SIGNATURES = {
    "mg_version": (i32, []),
    "mg_last_error": (C.c_char_p, []),
    "mg_conv_workspace_bytes": (sz, [i32, i32, i32]),
    "mg_conv_set_lds_pad": (i32, [i64]),
    "mg_conv1d_gather": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i64, i64, C.POINTER(Epilogue), vp, sz, vp]),
    "mg_conv1d_scatter2": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i64, i64, C.POINTER(Epilogue), vp, sz, vp]),
    "mg_wq_relayout": (i32, [vp, vp, i32, i32, i32, i32, i32, vp]),
    "mg_conv16_supported": (i32, [i32, i32, i32, i32, i32, i32]),
    "mg_conv16": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, i64, i64, C.POINTER(Epilogue), vp]),
    "mg_conv16_plan": (i32, [i32, i32, i32, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]),
    "mg_conv16_stats": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, i64, i64, C.POINTER(Epilogue), vp, vp]),
    "mg_conv16_ex": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, i64, i64, C.POINTER(Epilogue), C.POINTER(Conv16Extra), vp]),
    "mg_conv16_poolable": (i32, [i32, i32, i32, i32]),
    "mg_conv16_pool": (i32, [vp, vp, vp, i32, i32, i32, i32, i64, i64, C.POINTER(Epilogue), vp, f32, vp]),
    "mg_bn_train_fwd_parts": (i32, [vp, i32, i32, vp, vp, i64, i32, vp, vp, vp, vp, f32, f32, vp, vp, i32, vp]),
    "mg_linear_workspace_bytes": (sz, [i32, i32, i32]),
    "mg_linear": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, C.POINTER(Epilogue), vp, sz, vp]),
    "mg_linear_perm": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, C.POINTER(Epilogue), i32, vp, sz, vp]),
    "mg_conv_tile_config": (i32, [i64, i32, i32]),
    "mg_conv_thin_route": (i32, [vp, i64, i32, i32, i32, i32, i32]),
    "mg_wb_relayout": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "mg_conv1d_s1_bf16_supported": (i32, [i32, i32, i32, i32, i32]),
    "mg_conv1d_s1_bf16": (i32, [vp, i32, vp, vp, i32, i32, i32, i32, i32, i32, C.POINTER(EpilogueBf16), vp]),
    "mg_meanT_fwd_bf16": (i32, [vp, vp, i32, i32, i32, vp]),
    "mg_meanT_bwd_bf16": (i32, [vp, vp, vp, i32, vp, i32, i32, i32, vp]),
    "mg_wgrad_workspace_bytes": (sz, [i32, i32, i32, i32, i32]),
    "mg_wgrad": (i32, [vp, vp, i32, vp, vp, i32, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, sz, vp]),
    "mg_wgrad_multi": (i32, [vp, i32, i32, i32, vp, sz, vp]),
    "mg_colsum_workspace_bytes": (sz, [i32]),
    "mg_colsum": (i32, [vp, i64, i32, vp, vp, vp, sz, vp]),
    "mg_bn_workspace_bytes": (sz, [i32]),
    "mg_bn_train_fwd": (i32, [vp, vp, i64, i32, vp, vp, vp, vp, f32, f32, vp, vp, i32, vp, sz, vp]),
    "mg_bn_groups_workspace_bytes": (sz, [i32, i32]),
    "mg_bn_train_fwd_groups": (i32, [vp, vp, i64, i32, i32, vp, vp, vp, vp, f32, f32, vp, vp, i32, vp, sz, vp]),
    "mg_bn_train_bwd": (i32, [vp, vp, vp, vp, i64, i32, vp, vp, vp, vp, vp, vp, i32, vp, sz, vp]),
    "mg_bn_train_bwd_parts": (i32, [vp, i32, vp, vp, vp, vp, i64, i32, vp, vp, vp, vp, vp, vp, i32, vp]),
    "mg_bn_eval_fwd": (i32, [vp, vp, i64, i32, vp, vp, vp, vp, f32, i32, vp]),
    "mg_bn_fold": (i32, [vp, vp, vp, vp, vp, f32, vp, vp, i32, vp]),
    "mg_meanT_fwd": (i32, [vp, vp, i32, i32, i32, vp]),
    "mg_meanT_bwd": (i32, [vp, vp, i32, i32, i32, vp, i32, vp, vp]),
    "mg_layernorm_fwd": (i32, [vp, vp, vp, i32, i32, vp, vp, f32, vp]),
    "mg_layernorm_bwd_params": (i32, [vp, vp, vp, vp, i32, i32, vp]),
    "mg_dhead_fwd": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "mg_dhead_bwd": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "mg_dhead_fwd_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "mg_dhead_wgrad": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "mg_dhead_wgrad_loss": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, f32, vp, vp, i32, vp]),
    "mg_meanT_bwd_mean": (i32, [vp, vp, i32, i32, i32, vp, i32, vp, vp, vp, i32, f32, vp]),
    "mg_row_chain": (i32, [vp, i32, i32, vp]),
    "mg_mean_scaled": (i32, [vp, vp, i32, f32, vp]),
    "mg_stamp": (i32, [vp, vp]),
    "mg_spectral_norm_fwd": (i32, [vp, i32, i32, f32, vp]),
    "mg_spectral_norm_bwd": (i32, [vp, i32, vp]),
    "mg_conv1d_wino3_supported": (i32, [i32, i32, i32, i32]),
    "mg_wino3_weights": (i32, [vp, vp, i32, i32, i64, i64, i32, vp]),
    "mg_conv1d_wino3": (i32, [vp, vp, vp, i32, i32, i32, i32, vp, vp]),
    "mg_wino3_weights_multi": (i32, [vp, i32, vp]),
    "mg_gp_interp": (i32, [vp, vp, vp, vp, i32, i64, vp]),
    "mg_gp_penalty": (i32, [vp, vp, vp, vp, f32, i32, i64, vp]),
    "mg_wgan_d_loss": (i32, [vp, vp, f32, vp, i32, vp]),
    "mg_wgan_d_loss_gp": (i32, [vp, vp, f32, vp, vp, i32, vp]),
    "mg_softmax_ce": (i32, [vp, vp, vp, vp, f32, i32, i32, vp]),
    "mg_neg_mean": (i32, [vp, vp, i32, vp]),
    "mg_fill": (i32, [vp, f32, i64, vp]),
    "mg_axpby": (i32, [vp, vp, f32, f32, i64, vp]),
    "mg_copy_cols": (i32, [vp, i32, i32, vp, i32, i32, i32, i32, i32, vp]),
    "mg_transpose_bcl_blc": (i32, [vp, vp, i32, i32, i32, vp, i32, vp]),
    "mg_stage_rows": (i32, [vp, i32, i32, vp]),
    "mg_stage_rows_cursor": (i32, [vp, i32, i32, vp, i64, vp, vp, vp]),
    "mg_act_bwd": (i32, [vp, vp, i32, vp, vp, i64, vp]),
    "mg_rng_fill": (i32, [vp, i64, vp, i64, vp, i64, vp, i64, f32, C.c_uint64, vp, vp]),
    "mg_rng_fill_tick": (i32, [vp, i64, vp, i64, vp, i64, vp, i64, f32, C.c_uint64, vp, vp, f32, f32, vp]),
    "mg_rng_fill_tick2": (i32, [vp, i64, vp, i64, vp, i64, vp, i64, f32, C.c_uint64, vp, vp, vp, f32, f32, vp]),
    "mg_rng_fill_tick2_stage": (i32, [vp, i64, vp, i64, vp, i64, vp, i64, f32, C.c_uint64, vp, vp, vp, f32, f32, vp, i32, i32, vp, i64, vp, vp]),
    "mg_adam_flat": (i32, [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, vp, f32, vp, vp]),
    "mg_adam_flat_ticked": (i32, [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, vp, f32, vp, vp, vp]),
    "mg_adam_flat_wq": (i32, [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, vp, f32, vp, i32, vp, vp, i32, vp]),
    "mg_grad_norm_workspace_bytes": (sz, [i64]),
    "mg_grad_norm_clip": (i32, [vp, i64, f32, vp, vp, sz, vp]),
    "mg_reparam_fwd": (i32, [vp, vp, vp, vp, i64, vp]),
    "mg_reparam_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, i64, vp]),
    "mg_vae_loss_workspace_bytes": (sz, []),
    "mg_vae_loss": (i32, [vp, vp, i64, vp, vp, i64, f32, vp, vp, vp, vp, vp, sz, vp]),
    "mg_graph_begin": (i32, [vp]),
    "mg_graph_end": (i32, [vp, C.POINTER(vp)]),
    "mg_graph_end_n": (i32, [vp, C.POINTER(vp), i32]),
    "mg_graph_last_kernel_nodes": (i32, []),
    "mg_graph_launch": (i32, [vp, vp]),
    "mg_graph_destroy": (i32, [vp]),
    "mg_event_create": (i32, [C.POINTER(vp)]),
    "mg_event_record": (i32, [vp, vp]),
    "mg_event_elapsed_ms": (i32, [vp, vp, C.POINTER(f32)]),
    "mg_event_destroy": (i32, [vp]),
}

_lib = None


def load():
    """Returns the loaded library; raises RuntimeError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python melo-gan_amd/build.py` "
            "(or __graft_entry__.build()).  There is no fallback path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class HipError(RuntimeError):
    pass


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().mg_last_error()
        raise HipError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")
