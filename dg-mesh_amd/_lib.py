"""ctypes binding of libdgmesh_hip.so (the C ABI declared in include/dgmesh_hip.h).

The HIP library is the product: there is NO fallback.  If it cannot be loaded every entry point raises,
so a GPU box can never silently run something else.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# DGM_LIB_PATH: developer override used by tools/build_variant.sh A/B runs (another build of the SAME library)
LIB_PATH = os.environ.get("DGM_LIB_PATH") or os.path.join(_HERE, "lib", "libdgmesh_hip.so")
CSRC = os.path.join(_HERE, "csrc")

_c = ctypes
_vp = _c.c_void_p
ALLOC_FN = _c.CFUNCTYPE(_c.c_void_p, _c.c_void_p, _c.c_size_t)


class StateLayout(_c.Structure):
    """Mirror of dgm_state_layout (include/dgmesh_hip.h)."""
    _fields_ = [(n, _c.c_size_t) for n in (
        "rec", "depth", "radii", "tiles_touched", "offs", "cov3D", "clamped", "block_sums", "hist",
        "tile_count", "tile_offset", "big_list", "counters", "geometry_bytes",
        "inst", "point_list", "slab", "live", "ckpt", "ckpt64", "ulist_full", "binning_bytes",
        "final_T", "n_contrib", "ranges", "nproc", "cfin", "ulist_last", "image_bytes")] + [
        (n, _c.c_int) for n in ("tiles_x", "tiles_y", "n_chunks", "chunk_size")]


class MlpParams(_c.Structure):
    """Mirror of dgm_mlp_params (include/dgmesh_hip.h)."""
    _fields_ = [(n, _c.c_int) for n in ("n_layers", "width", "emb_dim", "t_dim", "skip_layer", "n_out")] + [
        ("W", _vp * 8), ("b", _vp * 8), ("Wh", _vp), ("bh", _vp)]


# name -> (restype, argtypes); must list every symbol declared in include/dgmesh_hip.h
_f, _i = _c.c_float, _c.c_int
SYMBOLS = {
    "dgm_abi_version": (_i, []),
    "dgm_last_error": (_c.c_char_p, []),
    "dgm_rasterize_forward": (_i, [ALLOC_FN, _vp, ALLOC_FN, _vp, ALLOC_FN, _vp, _i, _i, _i, _vp, _i, _i, _vp, _vp, _vp,
                                   _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _f, _f, _i, _vp, _vp, _i, _vp,
                                   _c.POINTER(_i)]),
    "dgm_rasterize_backward": (_i, [_i, _i, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _f, _f,
                                    _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "dgm_rasterize_forward_split_sh": (_i, [ALLOC_FN, _vp, ALLOC_FN, _vp, ALLOC_FN, _vp, _i, _i, _i, _vp, _i, _i, _vp, _vp, _vp,
                                            _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _f, _f, _i, _vp, _vp, _i, _vp,
                                            _c.POINTER(_i)]),
    "dgm_rasterize_forward_capacity": (_i, [ALLOC_FN, _vp, ALLOC_FN, _vp, ALLOC_FN, _vp, _i, _i, _i, _vp, _i, _i, _vp, _vp, _vp,
                                            _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _f, _f, _i, _vp, _vp, _i, _vp,
                                            _i, _vp]),
    "dgm_rasterize_backward_split_sh": (_i, [_i, _i, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp,
                                             _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                             _vp, _vp, _i, _vp]),
    "dgm_mark_visible": (_i, [_i, _vp, _vp, _vp, _vp, _vp]),
    "dgm_geometry_bytes": (_c.c_size_t, [_i, _i, _i]),
    "dgm_binning_bytes": (_c.c_size_t, [_i]),
    "dgm_image_bytes": (_c.c_size_t, [_i, _i]),
    "dgm_describe_state": (_i, [_i, _i, _i, _i, _c.POINTER(StateLayout)]),
    "dgm_set_profiling": (None, [_i]),
    "dgm_set_profiling_sampling": (None, [_i]),
    "dgm_get_stage_ms": (_i, [_c.POINTER(_f), _i]),
    "dgm_collect_stage_ms": (_i, [_c.POINTER(_f), _c.POINTER(_i), _i]),
    "dgm_stage_name": (_c.c_char_p, [_i]),
    "dgm_knn_scratch_bytes": (_c.c_size_t, [_i]),
    "dgm_knn_mean_dist2": (_i, [_i, _vp, _vp, _vp, _vp]),
    "dgm_image_loss_workspace_bytes": (_c.c_size_t, [_i, _i, _i]),
    "dgm_image_loss_forward": (_i, [_vp, _vp, _i, _i, _i, _f, _vp, _vp, _vp]),
    "dgm_image_loss_backward": (_i, [_vp, _vp, _i, _i, _i, _f, _vp, _vp, _vp, _vp]),
    "dgm_gaussian_apply_forward": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "dgm_gaussian_apply_backward": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "dgm_se3_exp_forward": (_i, [_i, _vp, _i, _vp, _vp]),
    "dgm_se3_exp_backward": (_i, [_i, _vp, _i, _vp, _vp, _i, _vp]),
    "dgm_se3_transform_forward": (_i, [_i, _vp, _vp, _vp, _vp]),
    "dgm_se3_transform_backward": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dgm_cycle_loss_workspace_bytes": (_c.c_size_t, [_i]),
    "dgm_cycle_loss_forward": (_i, [_i, _vp, _vp, _i, _vp, _vp, _vp]),
    "dgm_cycle_loss_backward": (_i, [_i, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "dgm_adam_step": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _f, _vp]),
    "dgm_densify_stats": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dgm_densify_scratch_bytes": (_c.c_size_t, [_i]),
    "dgm_densify_totals_offset": (_c.c_size_t, [_i]),
    "dgm_densify_decide": (_i, [_i, _vp, _vp, _vp, _vp, _f, _f, _f, _f, _vp, _vp, _vp]),
    "dgm_densify_apply": (_i, [_i, _c.c_uint, _c.c_uint, _c.c_uint, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "dgm_dpsr_splat_forward": (_i, [_i, _i, _vp, _vp, _vp, _vp]),
    "dgm_dpsr_splat_backward": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dgm_dpsr_interp_forward": (_i, [_i, _i, _vp, _vp, _vp, _vp]),
    "dgm_dpsr_interp_backward": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dgm_dpsr_spectral": (_i, [_i, _f, _vp, _vp, _i, _vp]),
    "dgm_laplace_scratch_floats": (_c.c_size_t, [_i]),
    "dgm_laplace_forward": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp]),
    "dgm_laplace_backward": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp]),
    "dgm_opacity_field_scratch_bytes": (_c.c_size_t, [_i]),
    "dgm_opacity_field": (_i, [_i, _vp, _vp, _vp, _vp, _f, _i, _i, _f, _vp, _vp, _vp, _vp]),
    "dgm_mlp_set_gemm": (_i, [_i]),
    "dgm_timenet_forward": (_i, [_vp, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp]),
    "dgm_timenet_backward": (_i, [_vp, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dgm_mlp_workspace_bytes": (_c.c_size_t, [_i]),
    "dgm_mlp_describe_workspace": (_i, [_i, _c.POINTER(_c.c_size_t), _i]),
    "dgm_mlp_forward": (_i, [_c.POINTER(MlpParams), _i, _vp, _vp, _i, _vp, _vp, _vp]),
    "dgm_mlp_backward": (_i, [_c.POINTER(MlpParams), _i, _vp, _i, _vp, _vp * 8, _vp * 8, _vp, _vp, _vp, _vp]),
    "dgm_mlp_backward_dx": (_i, [_c.POINTER(MlpParams), _i, _vp, _i, _vp, _vp * 8, _vp * 8, _vp, _vp, _vp, _vp, _vp, _vp]),
}

_LIB = None
ABI_VERSION = 5  # DGM_ABI_VERSION of include/dgmesh_hip.h


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "-j8"] + (["-B"] if force else [])
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if out.returncode != 0:
        raise RuntimeError("building libdgmesh_hip.so failed:\n" + out.stdout[-4000:])
    if verbose:
        print(out.stdout)
    return LIB_PATH


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(dg-mesh_amd has no CPU or PyTorch fallback for its kernels)")
        # torch must be imported first: it ships its own libamdhip64 and all device pointers / streams we are
        # handed belong to that runtime.  Loading ours first would put a second HIP runtime in the process.
        import torch  # noqa: F401

        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(handle, name)  # AttributeError here = header and library out of sync
            fn.restype = res
            fn.argtypes = args
        if handle.dgm_abi_version() != ABI_VERSION:  # no bypass: struct layouts and argtypes below belong to exactly this version
            raise RuntimeError(f"libdgmesh_hip.so ABI version {handle.dgm_abi_version()} != {ABI_VERSION} (rebuild: __graft_entry__.build())")
        _LIB = handle
    return _LIB


def stream_ptr():
    """Raw handle of torch's current stream on the current device, as the C ABI takes it (torch.cuda.current_stream() builds a
    Stream object through three Python layers: ~11 us, and the step asks ~10 times)."""
    import torch
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


class _NoGuard:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def device_guard(device):
    """`with torch.cuda.device(device)` when `device` is not already current; nothing otherwise (the common case)."""
    import torch
    idx = device.index
    if idx is None or idx == torch._C._cuda_getDevice():
        return _NO_GUARD
    return torch.cuda.device(device)


def check(status):
    if status != 0:
        raise RuntimeError(lib().dgm_last_error().decode() or "libdgmesh_hip error")


STAGE_COUNT = 12


def stage_ms():
    buf = (_f * STAGE_COUNT)()
    n = lib().dgm_get_stage_ms(buf, STAGE_COUNT)
    return {lib().dgm_stage_name(i).decode(): float(buf[i]) for i in range(n)}


def collect_stage_ms():
    """Deferred profiling (dgm_set_profiling(2)): {stage: (avg_ms, launches)} since the mode was set."""
    ms = (_f * STAGE_COUNT)()
    cnt = (_i * STAGE_COUNT)()
    n = lib().dgm_collect_stage_ms(ms, cnt, STAGE_COUNT)
    return {lib().dgm_stage_name(i).decode(): (float(ms[i]), int(cnt[i])) for i in range(n)}
