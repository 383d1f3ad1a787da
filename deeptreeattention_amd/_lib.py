"""ctypes binding of libdta_hip.so (C ABI declared in include/dta_hip.h).

The HIP library is the product; there is no CPU or PyTorch fallback.  If the library is missing or a call
fails, a RuntimeError is raised with dta_last_error().
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DTA_DEV_LIB=1 (a choice of FILE made here, in Python): load the developer library -- the same sources built with
# -DDTA_DEV_SWITCHES, whose launch plans can be switched through DTA_* environment variables for same-box A/B runs
# (tools/, tests/test_kernel_variants_gpu.py).  The product library reads nothing from the environment.
LIB_PATH = os.path.join(_HERE, "libdta_hip_dev.so" if os.environ.get("DTA_DEV_LIB") == "1" else "libdta_hip.so")

DTA_F32, DTA_BF16 = 0, 1
ABI_VERSION = 2    # DTA_ABI_VERSION
MAX_YEARS = 16   # DTA_MAX_YEARS
FORWARD_ONLY = 8   # DTA_FORWARD_ONLY (heads_mask flag)
REUSE_PACKED = 32  # DTA_REUSE_PACKED (heads_mask flag: frozen-weight inference keeps the re-laid-out weights in the workspace)
XCHG_HANDLE_BYTES = 128   # DTA_XCHG_HANDLE_BYTES
SKIP_BLEND = 16    # DTA_SKIP_BLEND (heads_mask flag): the blend is left to dta_net_loss
NET_HANG2020, NET_SPECTRAL, NET_SPATIAL, NET_VANILLA = 0, 1, 2, 3
SITE_CONV_FWD, SITE_CONV_WGRAD, SITE_CONV_DGRAD, SITE_STAGE_FWD, SITE_STAGE_BWD, SITE_GEMM = 0, 3, 6, 9, 12, 15
_DTYPES = {"fp32": DTA_F32, "f32": DTA_F32, "float32": DTA_F32, "bf16": DTA_BF16, "bfloat16": DTA_BF16}

c_float_p = C.POINTER(C.c_float)
c_double_p = C.POINTER(C.c_double)
c_ll_p = C.POINTER(C.c_longlong)


class NetDesc(C.Structure):
    _fields_ = [("batch", C.c_int), ("bands", C.c_int), ("height", C.c_int), ("width", C.c_int),
                ("classes", C.c_int), ("kind", C.c_int), ("dtype", C.c_int), ("training", C.c_int),
                ("heads_mask", C.c_int), ("bn_momentum", C.c_float), ("bn_eps", C.c_float)]


class CropDesc(C.Structure):      # mirrors dta_crop_desc
    _fields_ = [(n, C.c_int) for n in ("batch", "bands_raw", "clip", "size", "flip", "layout", "dtype")]


CROP_F32, CROP_I16, CROP_U8 = 0, 1, 2
CROP_CHW, CROP_HWC = 0, 1


class SubnetParams(C.Structure):
    _fields_ = [("conv_w", C.c_void_p * 3), ("conv_b", C.c_void_p * 3), ("bn_w", C.c_void_p * 3),
                ("bn_b", C.c_void_p * 3), ("bn_rm", C.c_void_p * 3), ("bn_rv", C.c_void_p * 3),
                ("bn_nbt", C.c_void_p * 3), ("att", (C.c_void_p * 6) * 3), ("fc_w", C.c_void_p * 3),
                ("fc_b", C.c_void_p * 3)]


class SubnetGrads(C.Structure):
    _fields_ = [("conv_w", C.c_void_p * 3), ("conv_b", C.c_void_p * 3), ("bn_w", C.c_void_p * 3),
                ("bn_b", C.c_void_p * 3), ("att", (C.c_void_p * 6) * 3), ("fc_w", C.c_void_p * 3),
                ("fc_b", C.c_void_p * 3)]


class ConvModuleDesc(C.Structure):
    _fields_ = [("batch", C.c_int), ("in_channels", C.c_int), ("filters", C.c_int), ("height", C.c_int),
                ("width", C.c_int), ("pool", C.c_int), ("training", C.c_int), ("dtype", C.c_int),
                ("bn_momentum", C.c_float), ("bn_eps", C.c_float)]


class AttentionDesc(C.Structure):
    _fields_ = [("batch", C.c_int), ("filters", C.c_int), ("height", C.c_int), ("width", C.c_int), ("kind", C.c_int)]


class AdamSegment(C.Structure):     # dta_adam_segment
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("n", C.c_size_t),
                ("active", C.c_void_p), ("dev_step", C.c_void_p), ("dev_step_next", C.c_void_p), ("step", C.c_int),
                ("lr", C.c_float)]


class Level(C.Structure):           # dta_level
    _fields_ = [("classes", C.c_int), ("first", C.c_int), ("count", C.c_int), ("labels", C.c_void_p), ("weight", C.c_void_p),
                ("mean_scores", C.c_void_p), ("kept", C.c_void_p), ("loss", C.c_void_p), ("dscore", C.c_void_p),
                ("scratch", C.c_void_p)]


MAX_LEVELS = 8   # DTA_MAX_LEVELS


ADAM_MAX_SEGMENTS = 16   # DTA_ADAM_MAX_SEGMENTS


class MetaParams(C.Structure):      # dta_meta_params
    _fields_ = [("emb", C.c_void_p), ("bn_w", C.c_void_p), ("bn_b", C.c_void_p), ("bn_rm", C.c_void_p), ("bn_rv", C.c_void_p),
                ("bn_nbt", C.c_void_p), ("mlp_w", C.c_void_p), ("mlp_b", C.c_void_p), ("fc_w", C.c_void_p), ("fc_b", C.c_void_p)]


class MetaGrads(C.Structure):       # dta_meta_grads
    _fields_ = [("emb", C.c_void_p), ("bn_w", C.c_void_p), ("bn_b", C.c_void_p), ("mlp_w", C.c_void_p), ("mlp_b", C.c_void_p),
                ("fc_w", C.c_void_p), ("fc_b", C.c_void_p)]


PtrArray6 = C.c_void_p * 6
ScoreTable = (C.c_void_p * 3) * 2

_lib = None


def lib():
    """Load the shared library once; fail loudly when it is absent (no fallback path exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -m deeptreeattention_amd.build` "
                "(hipcc, gfx950).  deeptreeattention_amd has no CPU/PyTorch fallback.")
        L = C.CDLL(LIB_PATH)
        L.dta_abi_version.restype = C.c_int
        L.dta_last_error.restype = C.c_char_p
        L.dta_build_id.restype = C.c_char_p
        L.dta_net_workspace_bytes.restype = C.c_size_t
        L.dta_net_workspace_bytes.argtypes = [C.POINTER(NetDesc)]
        L.dta_net_forward.restype = C.c_int
        L.dta_net_forward.argtypes = [C.POINTER(NetDesc), C.POINTER(SubnetParams), C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.POINTER(ScoreTable), C.c_void_p, C.c_void_p]
        L.dta_net_backward.restype = C.c_int
        L.dta_net_backward.argtypes = [C.POINTER(NetDesc), C.POINTER(SubnetParams), C.c_void_p, C.c_void_p,
                                       C.POINTER(ScoreTable), C.c_void_p, C.POINTER(SubnetGrads), C.c_void_p,
                                       C.c_int, C.c_void_p]
        L.dta_net_forward_tiles.restype = C.c_int
        L.dta_net_forward_tiles.argtypes = L.dta_net_forward.argtypes
        L.dta_net_backward_tiles.restype = C.c_int
        L.dta_net_backward_tiles.argtypes = [C.POINTER(NetDesc), C.POINTER(SubnetParams), C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.POINTER(ScoreTable), C.c_void_p, C.POINTER(SubnetGrads), C.c_void_p,
                                             C.c_int, C.c_void_p]
        L.dta_net_backward_dp.restype = C.c_int
        L.dta_net_backward_dp.argtypes = [C.POINTER(NetDesc), C.POINTER(SubnetParams), C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.POINTER(ScoreTable), C.c_void_p, C.POINTER(SubnetGrads), C.c_void_p, C.c_void_p,
                                          C.c_int, C.c_void_p]
        L.dta_adam_step_dp.restype = C.c_int
        L.dta_adam_step_dp.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float,
                                       C.c_float, C.c_float, C.c_int, C.c_void_p]
        L.dta_preprocess_crops_tiles.restype = C.c_int
        L.dta_preprocess_crops_tiles.argtypes = [C.POINTER(CropDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                 C.c_void_p, C.c_void_p]
        L.dta_ensemble_workspace_bytes.restype = C.c_size_t
        L.dta_ensemble_workspace_bytes.argtypes = [C.POINTER(NetDesc), C.c_int]
        L.dta_ensemble_forward.restype = C.c_int
        L.dta_ensemble_forward.argtypes = [C.POINTER(NetDesc), C.c_int, C.POINTER(SubnetParams),
                                           C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_void_p]
        L.dta_ensemble_backward.restype = C.c_int
        L.dta_ensemble_backward.argtypes = [C.POINTER(NetDesc), C.c_int, C.POINTER(SubnetParams), C.c_void_p,
                                            C.c_void_p, C.POINTER(SubnetGrads), C.c_void_p]
        L.dta_preprocess_out_bands.restype = C.c_int
        L.dta_preprocess_out_bands.argtypes = [C.c_int, C.c_int]
        L.dta_preprocess_crops.restype = C.c_int
        L.dta_preprocess_crops.argtypes = [C.POINTER(CropDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p]
        L.dta_meta_head_workspace_bytes.restype = C.c_size_t
        L.dta_meta_head_workspace_bytes.argtypes = [C.c_int, C.c_int, C.c_int]
        L.dta_meta_head_forward.restype = C.c_int
        L.dta_meta_head_forward.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.POINTER(MetaParams),
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.dta_meta_head_loss.restype = C.c_int
        L.dta_meta_head_loss.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.dta_meta_head_backward.restype = C.c_int
        L.dta_meta_head_backward.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(MetaParams), C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(MetaGrads), C.c_void_p, C.c_void_p]
        L.dta_net_loss.restype = C.c_int
        L.dta_net_loss.argtypes = [C.POINTER(NetDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.dta_ensemble_forward_loss.restype = C.c_int
        L.dta_ensemble_forward_loss.argtypes = [C.POINTER(NetDesc), C.c_int, C.POINTER(SubnetParams), C.POINTER(C.c_void_p), C.c_void_p,
                                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.c_void_p, C.c_void_p]
        L.dta_adam_step_multi.restype = C.c_int
        L.dta_adam_step_multi.argtypes = [C.c_int, C.POINTER(AdamSegment), C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                          C.c_int, C.c_void_p]
        L.dta_multistage_workspace_bytes.restype = C.c_size_t
        L.dta_multistage_workspace_bytes.argtypes = [C.POINTER(NetDesc), C.c_int, C.POINTER(Level)]
        L.dta_multistage_forward_loss.restype = C.c_int
        L.dta_multistage_forward_loss.argtypes = [C.POINTER(NetDesc), C.c_int, C.POINTER(Level), C.POINTER(SubnetParams),
                                                  C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_void_p]
        L.dta_multistage_forward.restype = C.c_int
        L.dta_multistage_forward.argtypes = [C.POINTER(NetDesc), C.c_int, C.POINTER(Level), C.POINTER(SubnetParams),
                                             C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_void_p]
        L.dta_multistage_predict.restype = C.c_int
        L.dta_multistage_predict.argtypes = [C.POINTER(NetDesc), C.c_int, C.POINTER(Level), C.POINTER(SubnetParams),
                                             C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p),
                                             C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_void_p]
        L.dta_multistage_backward.restype = C.c_int
        L.dta_multistage_backward.argtypes = [C.POINTER(NetDesc), C.c_int, C.POINTER(Level), C.POINTER(SubnetParams), C.c_void_p,
                                              C.POINTER(SubnetGrads), C.c_void_p, C.c_void_p]
        L.dta_net_forward_loss.restype = C.c_int
        L.dta_net_forward_loss.argtypes = [C.POINTER(NetDesc), C.POINTER(SubnetParams), C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p]
        L.dta_weighted_ce.restype = C.c_int
        L.dta_weighted_ce.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p]
        L.dta_weighted_ce_scaled.restype = C.c_int
        L.dta_weighted_ce_scaled.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_void_p]
        L.dta_weighted_ce_scaled_dev.restype = C.c_int
        L.dta_weighted_ce_scaled_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                                 C.c_void_p, C.c_void_p, C.c_void_p]
        L.dta_year_flags.restype = C.c_int
        L.dta_year_flags.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
        L.dta_ensemble_forward_gated.restype = C.c_int
        L.dta_ensemble_forward_gated.argtypes = [C.POINTER(NetDesc), C.c_int, C.POINTER(SubnetParams), C.POINTER(C.c_void_p),
                                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.dta_adam_step.restype = C.c_int
        L.dta_adam_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float,
                                    C.c_float, C.c_float, C.c_void_p]
        L.dta_adam_step_zero_grad.restype = C.c_int
        L.dta_adam_step_zero_grad.argtypes = L.dta_adam_step.argtypes
        L.dta_adam_step_gated.restype = C.c_int
        L.dta_adam_step_gated.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int,
                                          C.c_void_p]
        L.dta_ensemble_backward_phased.restype = C.c_int
        L.dta_ensemble_backward_phased.argtypes = [C.POINTER(NetDesc), C.c_int, C.POINTER(SubnetParams), C.c_void_p,
                                                   C.c_void_p, C.POINTER(SubnetGrads), C.c_int, C.c_void_p]
        L.dta_ensemble_backward_gated.restype = C.c_int
        L.dta_ensemble_backward_gated.argtypes = [C.POINTER(NetDesc), C.c_int, C.POINTER(SubnetParams), C.c_void_p,
                                                  C.c_void_p, C.POINTER(SubnetGrads), C.c_void_p, C.c_int, C.c_void_p]
        L.dta_ensemble_backward_xchg.restype = C.c_int
        L.dta_ensemble_backward_xchg.argtypes = [C.POINTER(NetDesc), C.c_int, C.POINTER(SubnetParams), C.c_void_p,
                                                 C.c_void_p, C.POINTER(SubnetGrads), C.c_void_p, C.c_void_p, C.c_void_p]
        vp = C.c_void_p
        L.dta_conv_module_workspace_bytes.restype = C.c_size_t
        L.dta_conv_module_workspace_bytes.argtypes = [C.POINTER(ConvModuleDesc)]
        L.dta_conv_module_forward.restype = C.c_int
        L.dta_conv_module_forward.argtypes = [C.POINTER(ConvModuleDesc)] + [vp] * 11
        L.dta_conv_module_backward.restype = C.c_int
        L.dta_conv_module_backward.argtypes = [C.POINTER(ConvModuleDesc)] + [vp] * 10
        L.dta_attention_workspace_bytes.restype = C.c_size_t
        L.dta_attention_workspace_bytes.argtypes = [C.POINTER(AttentionDesc)]
        L.dta_attention_forward.restype = C.c_int
        L.dta_attention_forward.argtypes = [C.POINTER(AttentionDesc), C.POINTER(PtrArray6), vp, vp, vp, vp, vp]
        L.dta_attention_backward.restype = C.c_int
        L.dta_attention_backward.argtypes = [C.POINTER(AttentionDesc), C.POINTER(PtrArray6), vp, vp, vp, vp, vp,
                                             C.POINTER(PtrArray6), vp]
        L.dta_linear_forward.restype = C.c_int
        L.dta_linear_forward.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp]
        L.dta_linear_backward.restype = C.c_int
        L.dta_linear_backward.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp]
        L.dta_softmax_top2.restype = C.c_int
        L.dta_softmax_top2.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp, vp]
        L.dta_profile_enable.restype = C.c_int
        L.dta_profile_enable.argtypes = [C.c_int]
        L.dta_profile_set_stride.restype = C.c_int
        L.dta_profile_set_stride.argtypes = [C.c_int]
        L.dta_profile_collect.restype = C.c_int
        L.dta_profile_collect.argtypes = [C.POINTER(C.c_float), C.c_int]
        L.dta_profile_collect_site.restype = C.c_int
        L.dta_profile_collect_site.argtypes = [C.c_int, C.POINTER(C.c_float), C.c_int]
        L.dta_dev_reload_switches.restype = C.c_int
        L.dta_dev_switches_enabled.restype = C.c_int
        # peer gradient exchange (opaque handle = void*)
        L.dta_xchg_create.restype = C.c_int
        L.dta_xchg_create.argtypes = [C.c_int, C.c_int, C.c_size_t, C.POINTER(C.c_void_p)]
        L.dta_xchg_grad_buffer.restype = C.c_void_p
        L.dta_xchg_grad_buffer.argtypes = [C.c_void_p]
        L.dta_xchg_grad_capacity.restype = C.c_size_t
        L.dta_xchg_grad_capacity.argtypes = [C.c_void_p]
        L.dta_xchg_export.restype = C.c_int
        L.dta_xchg_export.argtypes = [C.c_void_p, C.c_void_p]
        L.dta_xchg_connect.restype = C.c_int
        L.dta_xchg_connect.argtypes = [C.c_void_p, C.c_void_p]
        L.dta_xchg_set_timeout.restype = None
        L.dta_xchg_set_timeout.argtypes = [C.c_void_p, C.c_double]
        L.dta_xchg_set_max_workgroups.restype = None
        L.dta_xchg_set_max_workgroups.argtypes = [C.c_void_p, C.c_int]
        L.dta_xchg_allreduce.restype = C.c_int
        L.dta_xchg_allreduce.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p]
        L.dta_xchg_reduce_head.restype = C.c_int
        L.dta_xchg_reduce_head.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p]
        L.dta_xchg_adam_step.restype = C.c_int
        L.dta_xchg_adam_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                         C.c_longlong, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float,
                                         C.c_float, C.c_float, C.c_int, C.c_void_p]
        L.dta_xchg_set_split.restype = C.c_int
        L.dta_xchg_set_split.argtypes = [C.c_void_p, C.c_size_t]
        L.dta_net_backward_xchg.restype = C.c_int
        L.dta_net_backward_xchg.argtypes = [C.POINTER(NetDesc), C.POINTER(SubnetParams), C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.POINTER(ScoreTable), C.c_void_p, C.POINTER(SubnetGrads), C.c_void_p, C.c_void_p,
                                            C.c_longlong, C.c_void_p]
        L.dta_xchg_disconnect.restype = C.c_int
        L.dta_xchg_disconnect.argtypes = [C.c_void_p]
        L.dta_xchg_selftest_fill.restype = C.c_int
        L.dta_xchg_selftest_fill.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.dta_xchg_selftest_verify.restype = C.c_int
        L.dta_xchg_selftest_verify.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.dta_xchg_selftest_mismatches.restype = C.c_int
        L.dta_xchg_selftest_mismatches.argtypes = [C.c_void_p]
        L.dta_xchg_status.restype = C.c_int
        L.dta_xchg_status.argtypes = [C.c_void_p]
        L.dta_xchg_last_timing.restype = C.c_int
        L.dta_xchg_last_timing.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.dta_xchg_destroy.restype = C.c_int
        L.dta_xchg_destroy.argtypes = [C.c_void_p]
        if L.dta_abi_version() != ABI_VERSION:
            raise RuntimeError("libdta_hip.so ABI version mismatch: the library says {}, this binding is written for {} "
                               "(rebuild: python -m deeptreeattention_amd.build --force)".format(L.dta_abi_version(), ABI_VERSION))
        _lib = L
    return _lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed: {lib().dta_last_error().decode()}")


def dtype_code(name):
    try:
        return _DTYPES[str(name).lower()]
    except KeyError:
        raise ValueError(f"precision must be one of {sorted(_DTYPES)}, got {name!r}")


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def current_stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
