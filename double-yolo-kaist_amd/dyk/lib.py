"""ctypes binding of libdyk_hip.so (C ABI declared in include/dyk_hip.h).

The product path has no CPU fallback: if the shared library cannot be loaded, anything that
needs it raises ``DykLibraryError``.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC_DIR = os.path.join(os.path.dirname(_HERE), "csrc")
LIB_PATH = os.path.join(CSRC_DIR, "libdyk_hip.so")

DYK_F32, DYK_BF16, DYK_U8 = 0, 1, 2
ACT_CODES = {"linear": 0, "leaky": 1, "mish": 2, "relu": 3, "relu6": 4, "hard-sigmoid": 5, "hard-swish": 6}
EPI_AFFINE, EPI_RESIDUAL, EPI_STATS, EPI_ACCUM, EPI_OUT_F32, EPI_BNBWD, EPI_ADDEND, EPI_BNFWD = 1, 2, 4, 8, 16, 32, 64, 128
EW_ACCUM = 1
EW_SKIP = 2
MAX_TAPS = 25
SE_POOL_SPLITS = 16        # DYK_SE_POOL_SPLITS (include/dyk_hip.h): partial-sum planes of the pixel-split SE pool

# op codes of DykCommand (include/dyk_hip.h)
(OP_CONV, OP_WGRAD, OP_BN_FINALIZE, OP_BN_ACT_FWD, OP_BN_BWD_REDUCE, OP_BN_BWD_APPLY, OP_AXPBY, OP_DOT,
 OP_UPSAMPLE_FWD, OP_UPSAMPLE_BWD, OP_MAXPOOL_FWD, OP_MAXPOOL_BWD, OP_SE_POOL, OP_SE_FC_FWD, OP_SE_FC_BWD,
 OP_SE_SCALE, OP_BN_BWD_PARAMS, OP_BN_FOLD, OP_WFUSE_WEIGHTS, OP_WFUSE_BWD_PARAMS, OP_HEAD_PERMUTE_FWD,
 OP_HEAD_PERMUTE_BWD, OP_PATCH_GATHER, OP_MEMSET, OP_YOLO_DECODE, OP_DW_FWD, OP_DW_DGRAD, OP_DW_WGRAD,
 OP_CAST_PAD_ROWS, OP_BN_FWD_FUSED, OP_GRAD_REDUCE, OP_STEM_FWD, OP_STEM_WGRAD) = range(1, 34)


class DykLibraryError(RuntimeError):
    pass


class DykError(RuntimeError):
    pass


_i8, _i32, _i64, _f32, _vp = ctypes.c_int8, ctypes.c_int32, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p


class DykConvDesc(ctypes.Structure):
    _fields_ = [
        ("x", _vp), ("w", _vp), ("y", _vp), ("scale", _vp), ("shift", _vp), ("res", _vp), ("stats", _vp),
        ("aux0", _vp), ("aux1", _vp), ("add", _vp),
        ("y2", _vp), ("bn_gamma", _vp), ("bn_beta", _vp), ("bn_running_mean", _vp), ("bn_running_var", _vp),
        ("bn_save_mean", _vp), ("bn_save_rstd", _vp), ("bn_counter", _vp),
        ("dtype", _i32), ("ldx", _i32), ("ldy", _i32), ("ldr", _i32),
        ("B", _i32), ("Hi", _i32), ("Wi", _i32), ("Cin", _i32), ("Cout", _i32),
        ("Hg", _i32), ("Wg", _i32), ("Ho", _i32), ("Wo", _i32),
        ("isy", _i32), ("isx", _i32), ("osy", _i32), ("osx", _i32), ("ooy", _i32), ("oox", _i32),
        ("ntaps", _i32),
        ("tdy", _i8 * MAX_TAPS), ("tdx", _i8 * MAX_TAPS), ("twt", _i8 * MAX_TAPS), ("_pad", _i8),
        ("ncls", _i8), ("cls_first", _i8 * 4), ("cls_ntaps", _i8 * 4), ("cls_ooy", _i8 * 4), ("cls_oox", _i8 * 4),
        ("_pad2", _i8 * 3),
        ("act", _i32), ("flags", _i32), ("stats_slots", _i32),
        ("ldy2", _i32), ("bn_count", _i32), ("bn_momentum", _f32), ("bn_eps", _f32), ("tune", _i32),
        ("twin", _vp),
        ("sk_ws", _vp), ("sk_cnt", _vp), ("sk_ws_bytes", _i64), ("sk_cnt_n", _i32), ("splitk", _i32),
    ]


class DykWgradDesc(ctypes.Structure):
    _fields_ = [
        ("x", _vp), ("dy", _vp), ("dw", _vp), ("part", _vp), ("part_stride", _i64),
        ("dtype", _i32), ("ldx", _i32), ("lddy", _i32),
        ("B", _i32), ("Hi", _i32), ("Wi", _i32), ("Cin", _i32), ("Ho", _i32), ("Wo", _i32), ("Cout", _i32),
        ("isy", _i32), ("isx", _i32), ("ntaps", _i32),
        ("tdy", _i8 * MAX_TAPS), ("tdx", _i8 * MAX_TAPS), ("twt", _i8 * MAX_TAPS), ("_pad", _i8),
        ("splits", _i32), ("lddw", _i32), ("tune", _i32),
        ("twin", _vp),
        ("sk_ws", _vp), ("sk_cnt", _vp), ("sk_ws_bytes", _i64), ("sk_cnt_n", _i32), ("_pad2", _i32),
        ("group", _vp), ("group_n", _i32), ("_pad3", _i32),
    ]


class DykWgradGroupEntry(ctypes.Structure):
    _fields_ = [("x", _vp), ("dy", _vp), ("dw", _vp), ("part", _vp)]


class DykEwDesc(ctypes.Structure):
    _fields_ = [
        ("a", _vp), ("b", _vp), ("out", _vp), ("p0", _vp), ("p1", _vp), ("p2", _vp), ("p3", _vp), ("red", _vp),
        ("aux", _vp), ("aux2", _vp),
        ("dtype", _i32), ("npix", _i32), ("C", _i32), ("lda", _i32), ("ldb", _i32), ("ldo", _i32),
        ("act", _i32), ("flags", _i32), ("B", _i32), ("H", _i32), ("W", _i32), ("k", _i32),
        ("alpha", _f32), ("beta", _f32), ("slots", _i32),
        ("twin", _vp),
    ]


class DykBnFinalizeDesc(ctypes.Structure):
    _fields_ = [
        ("stats", _vp), ("gamma", _vp), ("beta", _vp), ("running_mean", _vp), ("running_var", _vp),
        ("scale", _vp), ("shift", _vp), ("save_mean", _vp), ("save_rstd", _vp),
        ("C", _i32), ("count", _i32), ("momentum", _f32), ("eps", _f32), ("slots", _i32),
        ("twin", _vp),
    ]


class DykSeFcDesc(ctypes.Structure):
    _fields_ = [
        ("pooled", _vp), ("w1", _vp), ("b1", _vp), ("w2", _vp), ("b2", _vp), ("scale", _vp), ("dscale", _vp),
        ("dpooled", _vp), ("dw1", _vp), ("db1", _vp), ("dw2", _vp), ("db2", _vp), ("ws", _vp),
        ("B", _i32), ("C", _i32), ("Cs", _i32),
    ]


class DykTransposeEntry(ctypes.Structure):
    _fields_ = [("src_off", _i64), ("dst_off", _i64), ("taps", _i32), ("rows", _i32), ("cols", _i32),
                ("tile_begin", _i32), ("dst_ld", _i32), ("_pad", _i32)]


class DykPadEntry(ctypes.Structure):
    _fields_ = [("src", _vp), ("dst", _vp), ("rows", _i32), ("cols", _i32), ("cpad", _i32), ("blk_begin", _i32),
                ("transpose_f32", _i32), ("_pad", _i32)]


class DykDwDesc(ctypes.Structure):
    _fields_ = [("x", _vp), ("y", _vp), ("w", _vp), ("dw", _vp), ("stats", _vp), ("part", _vp),
                ("dtype", _i32), ("ldx", _i32), ("ldy", _i32),
                ("B", _i32), ("Hi", _i32), ("Wi", _i32), ("Ho", _i32), ("Wo", _i32), ("C", _i32),
                ("k", _i32), ("stride", _i32), ("pad", _i32), ("flags", _i32), ("stats_slots", _i32),
                ("res", _vp), ("bn", _vp), ("ldr", _i32), ("act", _i32),
                ("pre", _vp), ("pre_act", _i32), ("_pad", _i32)]


class DykGradReduceEntry(ctypes.Structure):
    _fields_ = [("g_off", _i64), ("part_off", _i64), ("plane", _i64), ("n", _i32), ("splits", _i32),
                ("chunk_begin", _i32), ("_pad", _i32)]


class DykMiscDesc(ctypes.Structure):
    _fields_ = [("p", _vp * 6), ("n", _i64), ("i", _i32 * 12), ("f", _f32 * 4)]


class DykCommand(ctypes.Structure):
    _fields_ = [("op", _i32), ("lane", _i32), ("desc", _vp)]


class DykStemDesc(ctypes.Structure):
    _fields_ = [("img", _vp), ("wt", _vp), ("y", _vp), ("stats", _vp), ("scale", _vp), ("shift", _vp), ("dy", _vp),
                ("dw", _vp), ("part", _vp),
                ("bn_da", _vp), ("bn_yraw", _vp), ("bn_vecs", _vp), ("bn_red", _vp), ("bn_dgamma", _vp), ("bn_dbeta", _vp),
                ("dtype", _i32), ("in_u8", _i32),
                ("B", _i32), ("H", _i32), ("W", _i32), ("Cout", _i32), ("k", _i32), ("stride", _i32), ("pad", _i32),
                ("Ho", _i32), ("Wo", _i32), ("ldy", _i32), ("lddy", _i32), ("act", _i32), ("stats_slots", _i32),
                ("bn_fused", _i32), ("bn_slots", _i32)]


class DykSchedEntry(ctypes.Structure):
    _fields_ = [("cmd", _i32), ("stream", ctypes.c_int16), ("nwait", _i8), ("record", _i8), ("wait", _i32 * 7),
                ("cmd2", _i32)]


class DykDecodeDesc(ctypes.Structure):
    _fields_ = [("p", _vp), ("io", _vp), ("B", _i32), ("na", _i32), ("ny", _i32), ("nx", _i32), ("no", _i32),
                ("rows_total", _i32), ("row_offset", _i32), ("v4", _i32), ("stride", _f32),
                ("anchor_vec", _f32 * 16)]


class DykTargetsDesc(ctypes.Structure):
    _fields_ = [("targets", _vp), ("nt", _i32), ("nheads", _i32), ("na", _i32), ("ny", _i32 * 3), ("nx", _i32 * 3),
                ("anchor_vec", (_f32 * 16) * 3), ("iou_t", _f32), ("counts", _vp), ("indices", _vp), ("tbox", _vp),
                ("anch", _vp), ("tcls", _vp)]


class DykLossDesc(ctypes.Structure):
    _fields_ = [("p", _vp * 3), ("dp", _vp * 3), ("tobj", _vp * 3), ("nheads", _i32), ("B", _i32), ("no", _i32),
                ("nc", _i32), ("v4", _i32), ("ciou", _i32), ("hyp_box", _f32), ("hyp_obj", _f32), ("hyp_cls", _f32),
                ("cls_pw", _f32), ("obj_pw", _f32), ("gr", _f32), ("fl_gamma", _f32), ("fl_alpha", _f32),
                ("acc", _vp), ("out", _vp), ("flag", _vp)]


class DykOptimDesc(ctypes.Structure):
    _fields_ = [("p", _vp), ("g", _vp), ("m", _vp), ("v", _vp), ("wc", _vp), ("mask", _vp), ("n", _i64), ("lr", _f32), ("beta1", _f32),
                ("beta2", _f32), ("eps", _f32), ("weight_decay", _f32), ("grad_scale", _f32), ("step", _i32),
                ("zero_grad", _i32)]


class DykNmsDesc(ctypes.Structure):
    _fields_ = [("pred", _vp), ("out", _vp), ("out_rows", _vp), ("counts", _vp), ("ws", _vp), ("ws_per_image", _i64),
                ("B", _i32), ("N", _i32), ("no", _i32), ("conf_thres", _f32), ("iou_thres", _f32),
                ("multi_label", _i32), ("agnostic", _i32), ("max_num", _i32), ("n_classes", _i32),
                ("classes", _i32 * 16)]


_lib = None
_P = ctypes.POINTER

# name -> (restype, argtypes); tests/test_abi.py checks this table against include/dyk_hip.h.
SIGNATURES = {
    "dyk_abi_version": (_i32, []),
    "dyk_build_sha": (ctypes.c_char_p, []),
    "dyk_error_string": (ctypes.c_char_p, [_i32]),
    "dyk_conv_igemm": (_i32, [_P(DykConvDesc), _vp]),
    "dyk_conv_bnfwd_max_grid": (_i32, []),
    "dyk_conv_grid": (_i32, [_P(DykConvDesc)]),
    "dyk_conv_splitk_ws_bytes": (_i64, [_P(DykConvDesc), _P(_i32)]),
    "dyk_conv_wgrad": (_i32, [_P(DykWgradDesc), _vp]),
    "dyk_conv_wgrad_splits": (_i32, [_P(DykWgradDesc)]),
    "dyk_conv_wgrad_variant": (_i32, [_P(DykWgradDesc)]),
    "dyk_conv_wgrad_fold_ws_bytes": (_i64, [_P(DykWgradDesc), _P(_i32)]),
    "dyk_grad_reduce": (_i32, [_vp, _vp, _vp, _i32, _i32, _vp]),
    "dyk_bn_finalize": (_i32, [_P(DykBnFinalizeDesc), _vp]),
    "dyk_bn_finalize_act_fwd": (_i32, [_P(DykBnFinalizeDesc), _P(DykEwDesc), _vp]),
    "dyk_bn_fold": (_i32, [_vp, _vp, _vp, _vp, _f32, _vp, _vp, _i32, _vp]),
    "dyk_bn_act_fwd": (_i32, [_P(DykEwDesc), _vp]),
    "dyk_bn_act_bwd_reduce": (_i32, [_P(DykEwDesc), _vp]),
    "dyk_bn_bwd_params": (_i32, [_vp, _vp, _vp, _i32, _i32, _vp]),
    "dyk_bn_act_bwd_apply": (_i32, [_P(DykEwDesc), _vp]),
    "dyk_axpby": (_i32, [_P(DykEwDesc), _vp]),
    "dyk_dot": (_i32, [_P(DykEwDesc), _vp]),
    "dyk_wfuse_weights": (_i32, [_vp, _vp, _i32, _vp]),
    "dyk_wfuse_bwd_params": (_i32, [_vp, _vp, _vp, _i32, _vp]),
    "dyk_upsample2x_fwd": (_i32, [_P(DykEwDesc), _vp]),
    "dyk_upsample2x_bwd": (_i32, [_P(DykEwDesc), _vp]),
    "dyk_maxpool_fwd": (_i32, [_P(DykEwDesc), _vp, _vp]),
    "dyk_maxpool_bwd": (_i32, [_P(DykEwDesc), _vp, _vp]),
    "dyk_se_pool": (_i32, [_P(DykEwDesc), _vp, _vp]),
    "dyk_se_fc_fwd": (_i32, [_P(DykSeFcDesc), _vp]),
    "dyk_se_fc_bwd": (_i32, [_P(DykSeFcDesc), _vp]),
    "dyk_se_scale": (_i32, [_P(DykEwDesc), _vp]),
    "dyk_head_permute_fwd": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "dyk_head_permute_bwd": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "dyk_patch_gather": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _i32, _vp]),
    "dyk_image_prep": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _vp]),
    "dyk_pack_conv_weight": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "dyk_nchw_to_nhwc": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _i32, _vp]),
    "dyk_nhwc_to_nchw": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "dyk_cast_f32": (_i32, [_vp, _vp, _i64, _i32, _vp]),
    "dyk_cast_pad_rows": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "dyk_cast_pad_table": (_i32, [_vp, _i32, _i32, _i32, _vp]),
    "dyk_transpose_taps": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "dyk_dwconv_fwd": (_i32, [_P(DykDwDesc), _vp]),
    "dyk_dwconv_tile_ok": (_i32, [_P(DykDwDesc)]),
    "dyk_dwconv_dgrad": (_i32, [_P(DykDwDesc), _vp]),
    "dyk_dwconv_wgrad": (_i32, [_P(DykDwDesc), _vp]),
    "dyk_dwconv_wgrad_rows": (_i32, [_P(DykDwDesc)]),
    "dyk_run_commands": (_i32, [_P(DykCommand), _i32, _vp, _P(_i32)]),
    "dyk_run_command_pair": (_i32, [_P(DykCommand), _P(DykCommand), _vp]),
    "dyk_run_schedule_timed": (_i32, [_P(DykCommand), _P(DykSchedEntry), _i32, _vp, _P(_f32)]),
    "dyk_sched_stream": (_i32, [_i32, _P(_vp)]),
    "dyk_run_commands_overlap": (_i32, [_P(DykCommand), _i32, _vp, _P(_i32)]),
    "dyk_run_schedule": (_i32, [_P(DykCommand), _P(DykSchedEntry), _i32, _i32, _i32, _vp, _P(_i32)]),
    "dyk_dag_graph_create": (_i32, [_P(DykCommand), _i32, _P(_i32), _P(_i32), _P(_vp), _P(_i32)]),
    "dyk_schedule_graph_launch": (_i32, [_vp, _vp]),
    "dyk_schedule_graph_destroy": (_i32, [_vp]),
    "dyk_yolo_decode": (_i32, [_P(DykDecodeDesc), _vp]),
    "dyk_build_targets": (_i32, [_P(DykTargetsDesc), _vp]),
    "dyk_yolo_loss": (_i32, [_P(DykLossDesc), _P(DykTargetsDesc), _vp]),
    "dyk_nms_workspace_bytes": (_i64, [_i32, _i32, _i32]),
    "dyk_nms": (_i32, [_P(DykNmsDesc), _vp]),
    "dyk_adam_step": (_i32, [_P(DykOptimDesc), _vp]),
    "dyk_sgd_step": (_i32, [_P(DykOptimDesc), _vp]),
    "dyk_run_commands_timed": (_i32, [_P(DykCommand), _i32, _vp, _P(_f32)]),
    "dyk_loss_scale_grads": (_i32, [_vp, _i64, _i32, _vp, _vp]),
    "dyk_loss_scale_grads3": (_i32, [_vp, _i64, _i32, _vp, _vp, _vp, _vp]),
    "dyk_stem_conv_fwd": (_i32, [_P(DykStemDesc), _vp]),
    "dyk_stem_conv_wgrad": (_i32, [_P(DykStemDesc), _vp]),
    "dyk_stem_wgrad_planes": (_i32, [_P(DykStemDesc)]),
    "dyk_stem_wgrad_bn_fusable": (_i32, [_P(DykStemDesc)]),
    "dyk_box_convert": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "dyk_scale_coords": (_i32, [_vp, _i32, _i32, _f32, _f32, _f32, _f32, _f32, _i32, _vp]),
}


ABI_VERSION = 5          # = DYK_ABI_VERSION of include/dyk_hip.h: a stale .so with older descriptor layouts is refused


def load(path=None):
    """Load (once) and return the ctypes handle of libdyk_hip.so."""
    global _lib
    if _lib is not None:
        return _lib
    p = path or os.environ.get("DYK_LIB", LIB_PATH)
    if not os.path.exists(p):
        raise DykLibraryError(
            "libdyk_hip.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C double-yolo-kaist_amd/csrc`). There is no CPU fallback." % p)
    try:
        lib = ctypes.CDLL(p)
    except OSError as e:  # pragma: no cover
        raise DykLibraryError("cannot load %s: %s" % (p, e))
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise DykLibraryError("%s does not export %s (stale build?)" % (p, name))
        fn.restype = res
        fn.argtypes = args
    if lib.dyk_abi_version() != ABI_VERSION:
        raise DykLibraryError("ABI version mismatch")
    if path is None and "DYK_LIB" not in os.environ and os.environ.get("DYK_ALLOW_STALE_LIB", "0") == "0":
        # the in-tree library must be the build of the in-tree sources (a variant library named by DYK_LIB / `path` is the
        # caller's business: tools/ab.sh compares builds of different sources)
        from .buildinfo import native_sha
        built, tree = lib.dyk_build_sha().decode(), native_sha()
        if tree is not None and built != tree:      # (None: no sources beside the library -- a packaged copy; nothing to check)
            raise DykLibraryError("%s was built from other sources (digest %s, tree %s): rebuild it with "
                                  "`python -c 'import __graft_entry__ as g; g.build()'`" % (p, built, tree))
    _lib = lib
    return lib


def check(code, what=""):
    if code != 0:
        msg = load().dyk_error_string(code).decode()
        raise DykError("%s failed: %s (%d)" % (what or "dyk call", msg, code))
