"""ctypes binding of libsvdd_hip.so (include/svdd_hip.h).

The HIP library is the product: there is NO CPU or PyTorch fallback for the hot-path
operators. Importing this module without the built library, or calling an operator
without a gfx950 device, raises.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# SVDD_HIP_LIB: load another build of the library instead (the timing-experiment scripts under tools/ build patched
# copies of the kernels in a scratch directory; the tracked sources are never edited in place)
SO_PATH = os.environ.get("SVDD_HIP_LIB") or os.path.join(CSRC, "libsvdd_hip.so")
ABI_VERSION = 11

OK, E_ARG, E_LAUNCH, E_NODEVICE = 0, -1, -2, -3
LAYOUT_BLV, LAYOUT_BVL = 0, 1
RNG_REPLAY, RNG_PHILOX = 0, 1
SELECT_ARGMAX, SELECT_MULTINOMIAL = 0, 1
PRECISIONS = {"f32": 0, "f16x3": 1, "bf16x3": 2, "f16": 3, "bf16": 4}      # enum SVDD_PREC_* of include/svdd_hip.h
MAX_M = 1024

EXPORTS = (
    "svdd_abi_version", "svdd_device_info", "svdd_propose", "svdd_sample_categorical", "svdd_select", "svdd_x0hat",
    "svdd_finalize", "svdd_transform_samples", "svdd_subs_logp", "svdd_tds_resample",
    "svdd_set_option", "svdd_selftest_fastmath", "svdd_profile_enable", "svdd_profile_collect",
    "svdd_gru_bidir_f32", "svdd_gru_bidir_train_f32", "svdd_gru_bidir_bwd_f32", "svdd_epilogue_ln_f32", "svdd_conv1d_cl_f32",
    "svdd_conv1d_set_dynamic", "svdd_gru_set_mode", "svdd_conv_tower_f32", "svdd_backbone_cnn_f32", "svdd_value_tail_f32",
    "svdd_candidate_windows", "svdd_conv_tower_windows_f32", "svdd_k1_stats",
    "svdd_backbone_cnn_lp", "svdd_conv_tower_lp", "svdd_conv_tower_windows_lp", "svdd_gru_bidir_lp", "svdd_value_tail_lp",
    "svdd_compact_flags", "svdd_compact_by_key", "svdd_gather_rows", "svdd_advance_rows", "svdd_select_compact", "svdd_set_tower_version", "svdd_set_backbone_packing",
    "svdd_trunk_gemm", "svdd_trunk_act_split", "svdd_trunk_layernorm_split", "svdd_trunk_attn_pool", "svdd_trunk_stem_unfold", "svdd_trunk_attn_small",
    "svdd_trunk_windows", "svdd_trunk_stem_unfold_win", "svdd_trunk_attn_pool_win",
    "svdd_bb_layer_fwd_f32", "svdd_bb_layer_bwd_f32", "svdd_mt19937_uniform_f32",
    "svdd_backbone_set_workspace", "svdd_backbone_split_status", "svdd_backbone_cnn_save_f32", "svdd_backbone_cnn_grad_f32",
    "svdd_dps_probs", "svdd_dps_probs_bwd", "svdd_dps_guided_q", "svdd_reward_stem_f32", "svdd_reward_stem_bwd_f32",
    "svdd_conv1d_cl_gated_f32", "svdd_reward_tail_grad_f32", "svdd_sum_gate_f32", "svdd_gru_bidir_train2_f32", "svdd_gru_bidir_bwd2_f32",
)
OPT_FORCE_EXACT = 0


class SvddRng(ctypes.Structure):
    """struct svdd_rng (include/svdd_hip.h)."""
    _fields_ = [("kind", ctypes.c_int32), ("step", ctypes.c_uint32), ("uniforms", ctypes.c_void_p),
                ("seed", ctypes.c_uint64), ("row_offset", ctypes.c_uint64),
                ("uniforms_layout", ctypes.c_int32), ("uniforms_rows", ctypes.c_int32)]


class SvddError(RuntimeError):
    pass


def build(force=False):
    """Compile every csrc/*.hip for gfx950 (hipcc cross-compiles without a GPU); stale = any *.hip / *.h / Makefile newer."""
    import glob
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(CSRC, "Makefile")])
    srcs.append(os.path.join(os.path.dirname(_HERE), "include", "svdd_hip.h"))
    if os.environ.get("SVDD_HIP_LIB"):
        return SO_PATH                                   # an explicitly chosen build is used as it is
    stale = not os.path.exists(SO_PATH) or os.path.getmtime(SO_PATH) < max(os.path.getmtime(f) for f in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", CSRC, "-B", "-j", str(min(8, os.cpu_count() or 1))])   # one job per translation unit
    return SO_PATH


_lib = None


def lib():
    """Load libsvdd_hip.so. torch is imported first so that the library binds to the HIP
    runtime already resident in the process (same libamdhip64 SONAME) instead of a second copy."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  (loads torch's bundled libamdhip64.so.7 first)
    if not os.path.exists(SO_PATH):
        raise SvddError(f"{SO_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                        "or `make -C svdd_amd/csrc`. There is no CPU fallback for the SVDD hot path.")
    L = ctypes.CDLL(SO_PATH)
    for name in EXPORTS:
        if not hasattr(L, name):
            raise SvddError(f"libsvdd_hip.so does not export {name}")
    if L.svdd_abi_version() != ABI_VERSION:
        raise SvddError(f"libsvdd_hip.so ABI {L.svdd_abi_version()} != binding ABI {ABI_VERSION}; rebuild")
    vp, f32, i32 = ctypes.c_void_p, ctypes.c_float, ctypes.c_int
    L.svdd_propose.argtypes = [vp, vp, f32, f32, i32, i32, i32, i32, ctypes.POINTER(SvddRng), vp, vp, vp, vp]
    L.svdd_sample_categorical.argtypes = [vp, vp, i32, i32, i32, i32, ctypes.POINTER(SvddRng), vp, vp, vp]
    L.svdd_select.argtypes = [vp, vp, i32, i32, i32, i32, ctypes.POINTER(SvddRng), vp, vp, vp, vp]
    L.svdd_x0hat.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp]
    L.svdd_finalize.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp]
    L.svdd_transform_samples.argtypes = [vp, i32, i32, i32, vp, vp]
    L.svdd_subs_logp.argtypes = [vp, vp, i32, i32, i32, vp, vp]
    L.svdd_tds_resample.argtypes = [vp, vp, ctypes.c_double, vp, vp, i32, i32, vp, vp, vp, vp]
    L.svdd_set_option.argtypes = [i32, i32]
    L.svdd_mt19937_uniform_f32.argtypes = [vp, vp, ctypes.c_longlong, vp]
    L.svdd_backbone_set_workspace.argtypes = [vp, ctypes.c_longlong]
    L.svdd_backbone_split_status.argtypes = [ctypes.POINTER(ctypes.c_int)]
    L.svdd_selftest_fastmath.argtypes = [ctypes.POINTER(ctypes.c_double)]
    L.svdd_gru_bidir_f32.argtypes = [vp, vp, vp, vp, i32, i32, vp, vp]
    L.svdd_gru_bidir_train_f32.argtypes = [vp, vp, vp, vp, vp, i32, i32, vp]
    L.svdd_gru_bidir_bwd_f32.argtypes = [vp, vp, vp, vp, vp, i32, i32, vp]
    L.svdd_conv1d_set_dynamic.argtypes = [i32]
    L.svdd_gru_set_mode.argtypes = [i32]
    L.svdd_conv_tower_f32.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, vp, vp]
    L.svdd_value_tail_f32.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp, vp]
    L.svdd_candidate_windows.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp, vp]
    L.svdd_conv_tower_windows_f32.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp]
    L.svdd_backbone_cnn_f32.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, ctypes.POINTER(ctypes.c_int), vp, vp, i32, vp]
    L.svdd_backbone_cnn_save_f32.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, ctypes.POINTER(ctypes.c_int), vp, vp, vp, vp]
    L.svdd_backbone_cnn_grad_f32.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, ctypes.POINTER(ctypes.c_int), vp]
    L.svdd_conv1d_cl_f32.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, i32, vp, vp, vp, vp, vp]
    L.svdd_epilogue_ln_f32.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, ctypes.c_int64, i32, i32, vp]
    L.svdd_k1_stats.argtypes = [vp]
    L.svdd_set_tower_version.argtypes = [i32]
    L.svdd_set_backbone_packing.argtypes = [i32]
    L.svdd_compact_flags.argtypes = [vp, i32, vp, vp, vp, vp]
    L.svdd_compact_by_key.argtypes = [vp, i32, vp, vp, vp, i32, vp]
    L.svdd_gather_rows.argtypes = [vp, vp, vp, i32, i32, vp, vp]
    L.svdd_advance_rows.argtypes = [vp, vp, vp, i32, i32, i32, vp, vp]
    L.svdd_select_compact.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, ctypes.POINTER(SvddRng), vp, vp, vp, vp, vp, vp]
    L.svdd_conv_tower_lp.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, i32, vp]
    L.svdd_conv_tower_windows_lp.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, i32, vp]
    L.svdd_gru_bidir_lp.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, vp, i32, vp]
    L.svdd_value_tail_lp.argtypes = [vp, vp, vp, vp, vp, vp, f32, vp, i32, i32, i32, vp, i32, vp]
    L.svdd_backbone_cnn_lp.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, ctypes.POINTER(ctypes.c_int), i32, vp, vp, i32, vp]
    i64 = ctypes.c_int64
    L.svdd_trunk_gemm.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, i32, vp, vp, vp, vp, i32, i32, vp]
    L.svdd_trunk_act_split.argtypes = [vp, vp, vp, i32, i64, i32, i32, i32, vp, vp, vp, vp]
    L.svdd_trunk_layernorm_split.argtypes = [vp, vp, vp, f32, i64, i32, vp, vp, vp, i32, vp]
    L.svdd_trunk_attn_pool.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, i32, vp]
    L.svdd_trunk_stem_unfold.argtypes = [vp, i32, i32, vp, vp, vp]
    L.svdd_bb_layer_fwd_f32.argtypes = [vp, vp, vp, vp, vp, vp, f32, vp, vp, vp, i64, i32, i32, vp]
    L.svdd_bb_layer_bwd_f32.argtypes = [vp, vp, vp, vp, f32, vp, vp, vp, vp, i64, i32, i32, vp]
    L.svdd_trunk_windows.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp]
    L.svdd_trunk_stem_unfold_win.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp, vp, vp]
    L.svdd_trunk_attn_pool_win.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp]
    L.svdd_trunk_attn_small.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp]
    L.svdd_dps_probs.argtypes = [vp, vp, i32, i32, vp, vp]
    L.svdd_dps_probs_bwd.argtypes = [vp, vp, vp, i32, i32, vp, vp, vp]
    L.svdd_dps_guided_q.argtypes = [vp, vp, vp, vp, f32, f32, f32, i32, i32, vp, vp]
    L.svdd_reward_stem_f32.argtypes = [vp, vp, vp, vp, i32, i32, i32, vp]
    L.svdd_reward_stem_bwd_f32.argtypes = [vp, vp, vp, i32, i32, i32, vp]
    L.svdd_conv1d_cl_gated_f32.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp]
    L.svdd_reward_tail_grad_f32.argtypes = [vp, vp, vp, vp, vp, vp, vp, f32, i32, i32, vp, vp, vp]
    L.svdd_sum_gate_f32.argtypes = [vp, vp, vp, vp, i64, vp]
    L.svdd_gru_bidir_train2_f32.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, vp]
    L.svdd_gru_bidir_bwd2_f32.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i32, vp]
    L.svdd_profile_enable.argtypes = [i32]
    L.svdd_profile_collect.argtypes = [i32, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)]
    L.svdd_device_info.argtypes = [ctypes.c_char_p, i32, ctypes.POINTER(ctypes.c_int)]
    for name in EXPORTS:
        getattr(L, name).restype = ctypes.c_int
    _lib = L
    return L


def device_info():
    buf = ctypes.create_string_buffer(64)
    ncu = ctypes.c_int(0)
    rc = lib().svdd_device_info(buf, 64, ctypes.byref(ncu))
    if rc != OK:
        raise SvddError("no HIP device visible to libsvdd_hip.so")
    return buf.value.decode().split(":")[0], ncu.value


_OPTIONS = {}          # (key, sub) -> the value this process last set through set_option (absent: the library default)
_OPTION_DEFAULTS = {(4, "version"): 2, (4, "height"): 40, (4, "concurrency"): 51}


def _option_slot(key, value):
    """SVDD_OPT_TRUNK_GEMM_VERSION multiplexes three settings on one key (include/svdd_hip.h): kernel version, tile height (40 - 42),
    chains sharing the chip (51 - 54). Each is remembered separately."""
    if key == 4:
        return (4, "height" if 40 <= value <= 42 else "concurrency" if 51 <= value <= 54 else "version")
    return (key, None)


def current_option(key, like=0):
    """The value last set for `key` (for key 4: for the setting `like` belongs to)."""
    slot = _option_slot(key, like)
    return _OPTIONS.get(slot, _OPTION_DEFAULTS.get(slot, 0))


def set_option(key, value):
    """svdd_set_option through one door that remembers what was set -> the PREVIOUS value of that setting, so that a scoped change
    can put back what its caller had chosen instead of a constant."""
    key, value = int(key), int(value)
    prev = current_option(key, value)
    check(lib().svdd_set_option(key, value), f"svdd_set_option({key}, {value})")
    _OPTIONS[_option_slot(key, value)] = value
    return prev


def set_force_exact(on):
    """A/B switch: K1 evaluates every draw in the exact arithmetic (same results, slower)."""
    set_option(OPT_FORCE_EXACT, int(bool(on)))


def profile_enable(on=True):
    check(lib().svdd_profile_enable(int(bool(on))), "svdd_profile_enable")


def profile_collect(kernel):
    """(total_ms, launches) of kernel 0 = propose / 1 = select since profiling was enabled."""
    tot, n = ctypes.c_double(0.0), ctypes.c_int(0)
    check(lib().svdd_profile_collect(kernel, ctypes.byref(tot), ctypes.byref(n)), "svdd_profile_collect")
    return tot.value, n.value


def selftest_fastmath():
    """(max rel err of fast g over all 2^24 uniforms, of fast exp on [-80,0], of fast log on (1,4])."""
    out = (ctypes.c_double * 3)()
    check(lib().svdd_selftest_fastmath(out), "svdd_selftest_fastmath")
    return tuple(out)


_ERR = {E_ARG: "invalid argument", E_LAUNCH: "kernel launch failed", E_NODEVICE: "no HIP device"}


def check(rc, what):
    if rc != OK:
        raise SvddError(f"{what}: {_ERR.get(rc, rc)}")
