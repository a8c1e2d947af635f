"""ctypes binding of libneuradar_hip.so (C ABI in include/neuradar_hip.h).

The HIP library IS the product path: there is no CPU / eager fallback.  If the shared object is
missing this module raises at import of the first op (build it with `python -c "import
__graft_entry__ as g; g.build()"` or `make -C neuradar_amd/csrc`).
"""
import ctypes
import os
import subprocess
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int64, c_uint32, c_void_p

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB_PATH = os.environ.get("NR_LIB_PATH") or os.path.join(CSRC, "libneuradar_hip.so")  # (NR_LIB_PATH: A/B runs of two builds)
NR_MAX_LAYERS = 8
NR_EINVAL = -1
NR_LOSS_SLOTS = 1024
NR_ABI_VERSION = 28
NR_DTYPES = {"float32": 0, "bfloat16": 1, "float16": 2}  # nr_field_t.dtype
# nr_amp state layout (include/neuradar_hip.h)
NR_AMP_MAX_GROUPS, NR_AMP_SCALE, NR_AMP_GROWTH_TRACKER, NR_AMP_INV_SCALE, NR_AMP_SKIPPED_PREV, NR_AMP_SKIPPED_TOTAL = 8, 0, 1, 2, 3, 4
NR_AMP_FOUND, NR_AMP_FOUND_PREV, NR_AMP_FLOATS = 8, 16, 24


class NrMlp(Structure):
    _fields_ = [("num_layers", c_int), ("in_dim", c_int), ("width", c_int), ("out_dim", c_int),
                ("weight", c_void_p * NR_MAX_LAYERS), ("bias", c_void_p * NR_MAX_LAYERS)]


class NrMlpGrads(Structure):
    _fields_ = [("weight", c_void_p * NR_MAX_LAYERS), ("bias", c_void_p * NR_MAX_LAYERS)]


class NrRadarHeads(Structure):  # nr_radar_heads_t / nr_radar_heads_grads_t: [head][layer]
    _fields_ = [("weight", (c_void_p * 3) * 3), ("bias", (c_void_p * 3) * 3)]


class NrField(Structure):
    _fields_ = [("geo", NrMlp), ("feat", NrMlp), ("beta", c_void_p), ("packed", c_void_p), ("stash", c_void_p),
                ("dtype", c_int), ("sample_dirs", c_void_p), ("grad_scale", c_float), ("amp", c_void_p), ("amp_groups", c_uint32)]


class NrConv7List(Structure):
    _fields_ = [("n", c_int), ("offset", c_int64 * 16), ("bias_offset", c_int64 * 16)]


class NrConv7Fold(Structure):
    _fields_ = [("n", c_int), ("weight", c_void_p * 16), ("stride_o", c_int64 * 16), ("stride_t", c_int64 * 16), ("stride_i", c_int64 * 16),
                ("bias", c_void_p * 16), ("gamma", c_void_p * 16), ("beta", c_void_p * 16), ("mean", c_void_p * 16), ("var", c_void_p * 16),
                ("eps", c_float * 16)]


_ENC_PARAMS = ("in_proj_weight", "in_proj_bias", "out_proj_weight", "out_proj_bias", "linear1_weight", "linear1_bias",
               "linear2_weight", "linear2_bias", "norm1_weight", "norm1_bias", "norm2_weight", "norm2_bias", "norm_weight", "norm_bias")


class NrEncoder(Structure):
    _fields_ = [(k, c_void_p) for k in _ENC_PARAMS] + [("d_model", c_int), ("dim_feedforward", c_int), ("eps", c_float),
                                                       ("p_drop", c_float), ("seed", c_uint32), ("seed_epoch", c_void_p)]


class NrEncoderGrads(Structure):
    _fields_ = [(k, c_void_p) for k in _ENC_PARAMS]


class NrLidarSup(Structure):
    _fields_ = [("is_lidar", c_void_p), ("did_return", c_void_p), ("range", c_void_p), ("carving_epsilon", c_float),
                ("non_return_distance", c_float), ("weight", c_float), ("depth_weight", c_float), ("non_return_loss_mult", c_float)]


class NrLidarLosses(Structure):
    _fields_ = [("did_return", c_void_p), ("range", c_void_p), ("target_intensity", c_void_p), ("row0", c_int64), ("n", c_int64),
                ("non_return_distance", c_float), ("non_return_loss_mult", c_float), ("quantile", c_float), ("depth_mult", c_float),
                ("intensity_mult", c_float), ("ray_drop_mult", c_float)]


class NrFieldGrads(Structure):
    _fields_ = [("geo", NrMlpGrads), ("feat", NrMlpGrads), ("beta", c_void_p)]


P, I, L, F = c_void_p, c_int, c_int64, c_float

# name -> argtypes (return type is int unless listed in _RESTYPES).  Mirrors include/neuradar_hip.h.
PROTOTYPES = {
    "nr_abi_version": [],
    "nr_target_arch": [],
    "nr_init": [],
    "nr_hash_mark_vertices": [P, P, I, I, L, P, P, P],
    "nr_adam_step_split": [P, P, P, P, L, F, F, F, F, P, P, P, P, I, I, P],
    "nr_set_tuning": [I, I],
    "nr_hash_encode_fwd": [P, P, P, P, I, I, I, P, L, L, L, I, P],
    "nr_hash_encode_bwd": [P, P, P, I, I, I, P, L, L, P, L, I, P],
    "nr_hash_encode_bwd_tuned": [P, P, P, I, I, I, P, L, L, P, L, I, I, P],
    "nr_hash_encode_bwd_marked": [P, P, P, I, I, I, P, L, L, P, L, I, I, P, P],
    "nr_hash_encode_bwd_shared": [P, P, P, I, I, I, P, L, L, P, L, P, P],
    "nr_hash_encode_bwd_shared_split": [P, P, P, I, I, I, P, L, L, P, L, P, I, P],
    "nr_hash_encode_bwd_binned_workspace_bytes": [I, I, I, L],
    "nr_hash_encode_bwd_binned": [P, P, P, I, I, I, P, L, L, P, L, P, P],
    "nr_prop_density_scatter_binned": [P, P, P, I, I, I, P, L, L, P, P, I, L, P, P, L, P, P],
    "nr_hash_encode_bwd_binned_lp": [P, P, P, I, I, I, P, L, L, P, L, P, I, P],
    "nr_prop_density_scatter_binned_lp": [P, P, P, I, I, I, P, L, L, P, P, I, L, P, P, L, P, I, P],
    "nr_prop_density_scatter_binned2": [P, P, P, P, I, L, P, P, P, P, I, L, L, P, I, I, I, L, P, P, P, P, P],
    "nr_attention_workspace_floats": [L, L, I],
    "nr_attention_fwd": [P, P, P, L, L, I, F, c_uint32, P, P, P, P, P, P],
    "nr_attention_bwd": [P, P, P, P, P, P, L, L, I, F, c_uint32, P, P, P, P, P, P, P],
    "nr_radar_assign_workspace_bytes": [I, L, I],
    "nr_radar_assign_status_offset": [I, L, I],
    "nr_radar_assign": [P, I, L, P, I, P, I, I, P, P, P],
    "nr_radar_loss": [P, I, L, P, I, P, P, I, F, P, P, P],
    "nr_radar_points_fwd": [P, P, L, P, P, I, P, P, P, P],
    "nr_radar_points_bwd": [P, P, L, P, P],
    "nr_radar_heads_fwd": [P, P, I, P, L, P, P],
    "nr_radar_heads_bwd": [P, P, I, P, L, P, P, P, P],
    "nr_conv7_image_bytes": [],
    "nr_conv7_pack": [P, POINTER(NrConv7List), I, P, P],
    "nr_conv7_fold_pack": [POINTER(NrConv7Fold), I, P, P, I, P],
    "nr_param_generation": [],
    "nr_conv7_fwd": [P, P, P, I, P, I, I, I, I, P],
    "nr_conv7_wgrad_workspace_bytes": [],
    "nr_conv7_wgrad": [P, P, P, P, I, P, I, I, I, I, P],
    "nr_pw_fwd": [P, I, P, P, P, I, L, I, I, I, I, I, I, I, P],
    "nr_pw_bwd_data": [P, P, I, P, P, I, L, I, I, I, I, I, I, P, I, P],
    "nr_pw_workspace_bytes": [],
    "nr_pw_bwd_weight": [P, I, P, P, I, P, P, I, P, L, I, I, I, I, I, I, I, P],
    "nr_encoder_pre_fwd": [P, P, P, L, P, P, P, P],
    "nr_encoder_post_fwd": [P, P, P, L, P, P],
    "nr_encoder_post_bwd": [P, P, P, P, L, P, P, P, P],
    "nr_encoder_pre_bwd": [P, P, P, P, P, P, P, L, P, P, P],
    "nr_bn_act_workspace_floats": [L, I],
    "nr_bn_act_fwd": [P, P, L, I, I, P, P, F, F, P, P, I, P, P, P, P, P],
    "nr_bn_act_bwd": [P, P, P, L, I, I, P, P, P, I, P, P, P, P, P, P],
    "nr_tcnn_grid_param_count": [I, I, I, I, I, F],
    "nr_tcnn_grid_geometry": [I, I, I, I, F, P, P, P],
    "nr_tcnn_grid_fwd": [P, P, I, I, I, I, I, F, P, L, P],
    "nr_tcnn_grid_bwd": [P, I, I, I, I, I, F, P, P, L, P],
    "nr_hash_encode_bwd_input": [P, P, P, P, I, I, I, P, L, L, P, L, P],
    "nr_actor_keyframes": [P, L, P, I, P, P, P, P],
    "nr_actor_candidates": [P, P, P, L, I, P, P, P, P, P, P, I, I, P, P, P],
    "nr_actor_w2b_fwd": [P, L, I, I, P, P, P, P, P, P, P, P],
    "nr_actor_w2b_bwd": [P, L, I, I, P, P, P, P, P, P, P, P, P],
    "nr_actor_assign": [P, P, P, P, L, I, I, P, I, P, P, P, F, P, P, P, P, P, P],
    "nr_actor_encode_fwd": [P, P, P, P, I, L, I, I, P, P, P, I, I, I, P, L, L, I, P],
    "nr_actor_encode_bwd": [P, P, P, P, I, L, I, I, P, P, P, I, I, I, P, L, L, I, P, P, P, P, P, P, F, P, P, P],
    "nr_contract_gaussians": [P, P, P, P, L, I, F, I, P, P, P],
    "nr_mlp_fwd": [POINTER(NrMlp), P, L, P, P],
    "nr_mlp_bwd": [POINTER(NrMlp), P, P, L, P, POINTER(NrMlpGrads), P],
    "nr_field_bwd_workspace_floats": [POINTER(NrField), L],
    "nr_field_stash_floats": [POINTER(NrField), L],
    "nr_field_image_floats": [POINTER(NrField)],
    "nr_field_pack": [POINTER(NrField), P, P],
    "nr_field_fwd": [POINTER(NrField), P, L, L, I, P, I, I, L, P, P, P, P],
    "nr_field_fwd_gather": [POINTER(NrField), P, P, P, P, I, I, I, P, L, P, I, I, L, P, P, P, P],
    "nr_field_bwd": [POINTER(NrField), P, L, L, I, P, I, I, L, P, P, P, P, POINTER(NrFieldGrads), P, P],
    "nr_field_grad_reduce": [POINTER(NrField), P, L, POINTER(NrFieldGrads), P],
    "nr_sh4_fwd": [P, L, P, P],
    "nr_prop_density_fwd": [P, L, L, I, P, I, L, I, I, P, P],
    "nr_prop_field_fwd": [P, P, P, P, I, I, I, P, P, L, L, L, I, I, P, P],
    "nr_prop_density_bwd": [P, L, L, I, P, I, L, I, I, P, P, P, P, P],
    "nr_power_bins": [P, P, P, L, I, F, F, P, P, P],
    "nr_power_bins_contract": [P, P, P, P, P, P, L, I, F, F, F, I, P, P, P, P, P],
    "nr_proposal_round": [P, P, P, P, P, P, P, P, P, L, I, I, F, F, F, F, I, P, P, P, P, P, P, P],
    "nr_weights_from_density_fwd": [P, P, L, I, P, P],
    "nr_weights_from_density_bwd": [P, P, P, L, I, P, P],
    "nr_pdf_resample": [P, P, P, P, P, L, I, I, F, F, F, P, P, P],
    "nr_composite_fwd": [P, P, P, L, I, I, P, P, P, P, P],
    "nr_composite_bwd": [P, P, P, P, P, P, P, P, L, I, I, P, P, P],
    "nr_render_weights_fwd": [P, P, P, L, I, P, P, P, P],
    "nr_render_weights_bwd": [P, P, P, P, P, P, P, L, I, P, P],
    "nr_accumulate_fwd": [P, P, L, I, I, P, P],
    "nr_accumulate_bwd": [P, P, P, L, I, I, P, P, P],
    "nr_render_train": [P, P, P, P, P, P, L, I, I, F, F, F, P, P, P, P, P, P, P, P, P, POINTER(NrLidarSup), P],
    "nr_lidar_depth_quantile": [P, POINTER(NrLidarLosses), P, P, P],
    "nr_lidar_losses": [P, P, POINTER(NrLidarLosses), P, P, P, P, P, P],
    "nr_appearance_concat_fwd": [P, I, P, I, P, P, F, I, L, L, P, P],
    "nr_appearance_concat_bwd": [P, I, I, P, P, F, I, L, L, P, P, L, P],
    "nr_lidar_head_loss": [P, P, P, L, P, F, F, P, P, P],
    "nr_depth_from_weights": [P, P, L, I, P, P],
    "nr_gen_rays_camera": [P, P, P, P, P, P, P, P, P, P, P, P, L, P, P, P, P, P, P],
    "nr_gen_rays_lidar": [P, P, I, P, P, P, L, P, P, P, P, P, P, P],
    "nr_gen_rays_lidar_sampled": [P, L, I, P, P, P, P, I, P, P, P, P, P, P, P, P, P, P, P],
    "nr_sample_radar_scans": [P, I, L, P, P],
    "nr_permutation_from_uniform": [P, I, P, P],
    "nr_gen_rays_radar": [P, L, P, P, F, F, I, F, F, I, P, P, P, P, P, P],
    "nr_adam_step": [P, P, P, P, L, F, F, F, F, F, I, I, F, I, P, P, P, P, I, P],
    "nr_apply_delta16": [P, P, L, L, L, P],
    "nr_grad_to16_clear": [P, P, L, P],
    "nr_adam_step_marked": [P, P, P, P, L, F, F, F, F, I, F, I, P, P, P, I, P],
    "nr_amp_init": [P, F, P],
    "nr_amp_update": [P, I, F, F, I, P],
    "nr_nonfinite_check": [P, L, P, P],
    "nr_unscale_add_16": [P, P, L, I, P, P, P],
    "nr_supervision_loss": [P, I, P, I, P, P, L, F, F, P, P, P, P],
    "nr_distortion_loss": [P, I, P, I, I, L, F, P, P, P],
    "nr_interlevel_loss": [P, I, P, I, I, P, P, I, L, F, F, P, P, P],
    "nr_interlevel_loss_to_density": [P, I, P, I, I, P, P, P, P, I, L, F, F, P, P, POINTER(NrLidarSup), P],
    "nr_adam_hyper": [P, P, F, F, I, I, F, F, P, I, P],
    "nr_grad_compact": [P, L, I, L, P, P, P, P],
    "nr_grad_apply": [P, P, P, L, I, P, P],
    "nr_grad_apply_guarded": [P, P, P, I, I, I, L, I, P, P, P],
    "nr_grad_compact_shards": [P, L, I, I, P, P, P, P, P],
    "nr_grad_lists_apply": [P, P, L, P, P, I, I, I, I, P, P, P],
    "nr_grad_lists_restore": [P, P, L, P, P, I, I, L, I, P, P, P],
    "nr_gen_rays_camera_patches": [P, L, I, I, I, I, I, F, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P],
    "nr_uniform_fill": [P, L, c_uint32, P, P],
}
_RESTYPES = {"nr_target_arch": c_char_p, "nr_param_generation": c_void_p, "nr_field_bwd_workspace_floats": c_int64, "nr_field_image_floats": c_int64,
             "nr_field_stash_floats": c_int64, "nr_hash_encode_bwd_binned_workspace_bytes": c_int64,
             "nr_tcnn_grid_param_count": c_int64, "nr_attention_workspace_floats": c_int64,
             "nr_radar_assign_workspace_bytes": c_int64, "nr_radar_assign_status_offset": c_int64,
             "nr_bn_act_workspace_floats": c_int64, "nr_conv7_image_bytes": c_int64,
             "nr_conv7_wgrad_workspace_bytes": c_int64, "nr_pw_workspace_bytes": c_int64}

_lib = None


def build(force: bool = False) -> str:
    """Compile the HIP library for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    if force or not os.path.exists(LIB_PATH) or _stale():
        subprocess.check_call(["make", "-C", CSRC, "-j4"], stdout=subprocess.DEVNULL)
    return LIB_PATH


def _stale() -> bool:
    so = os.path.getmtime(LIB_PATH)
    inc = os.path.join(os.path.dirname(CSRC), "..", "include", "neuradar_hip.h")
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))] + [inc]
    return any(os.path.getmtime(s) > so for s in srcs if os.path.exists(s))


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension is the only implementation of this path "
                "(no CPU fallback). Build it with `make -C neuradar_amd/csrc`.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, argtypes in PROTOTYPES.items():
            fn = getattr(handle, name)  # AttributeError here = ABI mismatch, fail loudly
            fn.argtypes = argtypes
            fn.restype = _RESTYPES.get(name, c_int)
        if handle.nr_abi_version() != NR_ABI_VERSION:
            raise RuntimeError("libneuradar_hip.so ABI version mismatch")
        for env, hint in RETIRED_ENV.items():
            if os.environ.get(env):
                raise RuntimeError(f"{env} is set but this build no longer reads it ({hint})")
        for knob, (env, _) in enumerate(TUNING):  # launch-shape knobs for A/B runs: read ONCE, here (the library never reads the environment)
            value = os.environ.get(env)
            if value:
                if handle.nr_set_tuning(knob, int(value)) != 0:
                    raise RuntimeError(f"{env}={value}: nr_set_tuning refused it")
        _lib = handle
    return _lib


# nr_set_tuning knobs (include/neuradar_hip.h, NR_TUNE_*), in enum order: (environment variable forwarded at load, meaning)
TUNING = (("NR_CONV7_BLOCKS", "persistent blocks of nr_conv7_fwd"), ("NR_BIN_BLOCKS_PER_CU", "bin blocks per half CU"),
          ("NR_SHARED_BLOCKS", "blocks of nr_hash_encode_bwd_shared"), ("NR_FIELD_FWD_BLOCKS", "blocks of nr_field_fwd*"),
          ("NR_FIELD_BWD_BLOCKS", "blocks of nr_field_bwd*"), ("NR_PDBWD_BLOCKS", "blocks of nr_prop_density_bwd"),
          ("NR_ADAM_BLOCKS", "blocks of nr_adam_step*"), ("NR_PW_MFMA_OFF", "1: generic kernels for the transposed convolution"),
          ("NR_SHARED_SPLIT", "threads per row of nr_hash_encode_bwd_shared* (1, 2, 4): overrides the caller's choice"))
# variables earlier rounds read and this build ignores: setting one is an A/B run that compares identical code -- refuse it loudly
RETIRED_ENV = {"NR_PW_MFMA": "use NR_PW_MFMA_OFF=1", "NR_PROP_SHARED_OFF": "no kernel ever read it (removed in ABI v27)",
               "NR_PROP_SHARED_BLOCKS": "no kernel ever read it (removed in ABI v27)"}

def set_tuning(env_name: str, value: int) -> None:
    """nr_set_tuning by the knob's environment-variable name (tests / probes switching a knob inside one process)."""
    knob = [e for e, _ in TUNING].index(env_name)
    check(lib().nr_set_tuning(knob, int(value)), f"nr_set_tuning({env_name})")


_inited_devices: set = set()


def ensure_device(index: int) -> None:
    """nr_init() once per device (dynamic-LDS attributes of the > 64 KB kernels): called by ops._stream() before every launch
    (a set look-up); the caller's current device must be `index`."""
    if index not in _inited_devices:
        check(lib().nr_init(), "nr_init")
        _inited_devices.add(index)


class NeuradarHipError(RuntimeError):
    pass


def check(rc: int, what: str) -> None:
    if rc == 0:
        return
    if rc == NR_EINVAL:
        raise NeuradarHipError(f"{what}: invalid argument (NR_EINVAL)")
    raise NeuradarHipError(f"{what}: HIP error {rc}")
