"""Drop-in replacement of the `diff_gaussian_rasterization` Python package on top of libdgmesh_hip.so.

Same names, argument order, return values and error behaviour as the reference:
  DGR/diff_gaussian_rasterization/__init__.py:21-220   (rasterize_gaussians, _RasterizeGaussians,
                                                        GaussianRasterizationSettings, GaussianRasterizer)
  DGR/rasterize_points.cu:35-217 / rasterize_points.h   (the three functions the pybind module `_C` exports)
(DGR/ = /root/reference/dgmesh/submodules/diff-gaussian-rasterization/.)

PyTorch is plumbing here: it owns device memory, the stream and autograd bookkeeping.  All arithmetic happens
in the HIP library; tensors are handed over as raw device pointers through ctypes.
"""
import ctypes
import time
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _lib

NUM_CHANNELS = 3  # DGR/cuda_rasterizer/config.h:15
LAST_NUM_RENDERED = 0  # num_rendered of the most recent forward (read by bench.py for the roofline bytes)
FORWARD_CALL_SECONDS = 0.0  # wall time spent inside dgm_rasterize_forward, i.e. mostly waiting on the R read-back
# Reporting aid (bench.py): with RECORD_LIVE_ROWS set, a backward counts the per-instance gradient rows it wrote and leaves the
# number in LAST_LIVE_ROWS.  Off by default, and only the int is kept: holding the binning buffer itself (rounds 4-5) kept it from
# returning to the caching allocator before the next forward asked for its own, which doubled the resident binning memory.
RECORD_LIVE_ROWS = False
LAST_LIVE_ROWS = None


def _count_live_rows(binning, P, W, H, R):
    """`live` bytes of the binning buffer that the backward set: the rows preprocess_bwd reads.  One device
    reduction + read-back."""
    lay = _lib.StateLayout()
    _lib.check(_lib.lib().dgm_describe_state(int(P), int(W), int(H), int(R), ctypes.byref(lay)))
    return int(binning[lay.live:lay.live + int(R)].sum(dtype=torch.int64).item())


def live_rows():
    """Rows counted by the most recent backward that ran with RECORD_LIVE_ROWS set (None if none did)."""
    return LAST_LIVE_ROWS


def _ptr(t):
    if t is None or t.numel() == 0:
        return None
    return ctypes.c_void_p(t.data_ptr())


def _f32c(t, name):
    if t is None:
        return None
    if t.numel() == 0:
        return t
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA/HIP tensor (dg-mesh_amd has no CPU path)")
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name} must be float32")
    return t.contiguous()


def _stream():
    return _lib.stream_ptr()


def _bucket(nbytes):
    """Sizes above 1 MiB rounded up to m * 2^k with m in 8..15 (at most 12.5 % more), above 2 GiB to the next power of two.  The
    binning buffer follows R, which differs from frame to frame and creeps upwards while the scene densifies or its splats grow:
    with exact sizes nearly every new maximum misses torch's caching allocator (a cached block serves smaller requests only) and
    costs a hipMalloc, which takes ~25 ms per GiB here.  With a few sizes per octave the same blocks come round again; where a
    block is several GiB (cfg4 on a scene inflating under the bench's random targets: 13 GB) one size per octave keeps the bytes
    allocated over a period of growth at twice the final size instead of nine times."""
    n = int(nbytes)
    if n <= (1 << 20):
        return n
    if n > (1 << 31):
        return 1 << (n - 1).bit_length()
    k = n.bit_length() - 4
    return ((n + (1 << k) - 1) >> k) << k


_BIN_CAPACITY = {}  # device index -> bytes: capacity the binning buffers of that device are allocated with


def _binning_capacity(device_index, nbytes):
    """Bytes to allocate for a binning buffer that must hold `nbytes`: a per-device capacity that only moves when a frame does not
    fit (then to the bucket of 1.25 x the need: headroom for the frames that follow) or needs less than a quarter of it.  R differs
    from view to view by a few per cent; with sizes that follow it a new maximum late in a run misses torch's caching allocator and
    costs a hipMalloc (~6 ms for this buffer at cfg2: 9 % of a 20-step timed region) -- with one capacity per device every forward
    asks for the same block and the allocator hands the previous step's back."""
    n = int(nbytes)
    cur = _BIN_CAPACITY.get(device_index, 0)
    if n > cur or 4 * n < cur:
        cur = _bucket(n + n // 4)
        _BIN_CAPACITY[device_index] = cur
    return cur


def _resizer(t, bucket=False):
    """resizeFunctional (DGR/rasterize_points.cu:27-33): grow a byte tensor, hand back its device pointer."""
    def cb(_ctx, nbytes):
        t.resize_(_binning_capacity(t.device.index or 0, nbytes) if bucket else int(nbytes))
        return t.data_ptr()
    return _lib.ALLOC_FN(cb)


# ---- the forward without the host round trip (dgm_rasterize_forward_capacity; DGM_SYNC_FREE=1 or rasterizer.SYNC_FREE = True) --------
# The binning buffer is sized for a CAPACITY of tile instances (1.25 x the largest R seen on the device, a few sizes per octave) before
# the frame's R is known; the call only enqueues, {R, flags} arrive in page-locked memory behind an event.  By default the wrapper
# waits for that event AFTER everything of the forward is enqueued (the GPU never idles waiting for the host; callers still get the
# true R and never see an overflowed frame: it is rendered again, transparently, with a larger capacity).  With DEFER_SETTLE set (the
# training loop) the forward returns at once and `settle()` is called after the backward has been enqueued: the host then waits on
# an event that has long fired.  The first frame on a device runs the synchronous protocol once to learn R.
import os as _os

SYNC_FREE = _os.environ.get("DGM_SYNC_FREE") == "1"
DEFER_SETTLE = False
SETTLE_WAIT_SECONDS = 0.0       # host time spent waiting in settle() / the inline wait (bench.py reports it)
OVERFLOW_REDOS = 0              # frames rendered again because R exceeded the capacity
UNIT_REDOS = 0                  # frames rendered again because capacity and R sat on different sides of the replay-unit boundary
INJECT_CAPACITY = 0             # test hook: the next capacity-mode forward uses this capacity (then the hook clears itself)
_SF = {}                        # device index -> {"cap": capacity in instances, "words": pinned int32[4], "event": torch.cuda.Event}
_PENDING = None                 # (device index, capacity) of a forward whose words have not been looked at yet
_CAP_OF = {}                    # data_ptr of a capacity-mode binning buffer -> (capacity, true R, numel): the reference-shaped API hands
                                # the backward ONE integer, the true R; the layout follows the capacity


def _sf_state(dev):
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    st = _SF.get(idx)
    if st is None:
        st = _SF[idx] = {"cap": 0, "words": torch.zeros(4, dtype=torch.int32).pin_memory(), "event": torch.cuda.Event(), "idx": idx}
    return st


_UNIT_BOUNDARY = 1 << 20  # DGM_FINE_UNITS_BELOW (csrc/dgm_common.hpp): frames below it are replayed in units of 32 entries, else 64


def _capacity_for(R):
    """Capacity (tile instances) for a frame of R: 1.25 x, rounded up to m * 2^k, m in 8..15 -- the sizes _bucket() gives bytes --
    but never across the replay-unit boundary: the unit length follows the number the layout is sized for, and it decides how the
    backward partitions its sums, so a frame is only ever accepted with its capacity on R's own side of the boundary (settle()):
    every bit of the gradients is then what the synchronous protocol gives."""
    n = max(int(R) + int(R) // 4, 4096)
    k = max(n.bit_length() - 4, 0)
    cap = ((n + (1 << k) - 1) >> k) << k
    if int(R) < _UNIT_BOUNDARY <= cap:
        cap = _UNIT_BOUNDARY - 1
    return cap


def _fixed_resizer(t, nbytes_cap):
    def cb(_ctx, nbytes):
        if int(nbytes) > nbytes_cap:
            raise RuntimeError("capacity-mode binning buffer smaller than the layout asks for")
        t.resize_(nbytes_cap)
        return t.data_ptr()
    return _lib.ALLOC_FN(cb)


def _read_words(st):
    w = st["words"]
    return int(w[0]) & 0xffffffff, int(w[1])


def settle():
    """Look at the words of the most recent capacity-mode forward (waits for its event).  True: the frame is good (LAST_NUM_RENDERED
    is its R).  False: R exceeded the capacity -- the frame was neutralised on the device, the capacity has been raised, render it
    again.  No pending forward: True."""
    global _PENDING, LAST_NUM_RENDERED, SETTLE_WAIT_SECONDS, OVERFLOW_REDOS, UNIT_REDOS
    if _PENDING is None:
        return True
    idx, cap = _PENDING
    _PENDING = None
    st = _SF[idx]
    t0 = time.perf_counter()
    st["event"].synchronize()
    SETTLE_WAIT_SECONDS += time.perf_counter() - t0
    R, flags = _read_words(st)
    if flags & 2:
        st["cap"] = max(st["cap"], _capacity_for(R))
        OVERFLOW_REDOS += 1
        return False
    if flags & 1:
        raise RuntimeError("Point is filtered although prefiltered is set. This shouldn't happen!")
    if (R < _UNIT_BOUNDARY) != (cap < _UNIT_BOUNDARY):
        # the capacity (left by a larger scene on this device) sits on the other side of the replay-unit boundary: the frame is
        # correct, but its gradients would be summed in another partition than the synchronous protocol's -- render it again
        st["cap"] = _capacity_for(R)
        UNIT_REDOS += 1
        return False
    if 4 * R < cap and not INJECT_CAPACITY:  # the scene shrank a lot (another scene on this device): follow it down
        st["cap"] = _capacity_for(R)
    LAST_NUM_RENDERED = R
    return True


def _forward_dispatch(dev, binning, run_sync, run_cap):
    """run_sync(binning_callback) -> R issues the synchronous C call; run_cap(binning_callback, capacity, words_ptr) the capacity one.
    Returns (R or None when deferred, R the layout / the backward follows)."""
    global _PENDING, INJECT_CAPACITY, LAST_NUM_RENDERED
    if not SYNC_FREE:
        R = run_sync(_resizer(binning, bucket=True))
        LAST_NUM_RENDERED = R
        return R, R
    st = _sf_state(dev)
    if st["cap"] == 0 and not INJECT_CAPACITY:  # first frame on this device: the reference's protocol once, to learn R
        R = run_sync(_resizer(binning, bucket=True))
        st["cap"] = _capacity_for(R)
        LAST_NUM_RENDERED = R
        return R, R
    settle()  # (a deferred frame nobody settled: its words are about to be overwritten)
    while True:
        cap = INJECT_CAPACITY or st["cap"]
        INJECT_CAPACITY = 0
        nbytes = int(_lib.lib().dgm_binning_bytes(cap)) + 256
        run_cap(_fixed_resizer(binning, nbytes), cap, ctypes.c_void_p(st["words"].data_ptr()))
        st["event"].record(torch.cuda.current_stream(dev))
        _PENDING = (st["idx"], cap)
        if DEFER_SETTLE:
            return None, cap
        if settle():
            return LAST_NUM_RENDERED, cap


class _CModule:
    """Stands in for the pybind extension `diff_gaussian_rasterization._C` (DGR/ext.cpp:15-18)."""

    @staticmethod
    def rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                            viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos,
                            prefiltered, debug):
        if means3D.ndimension() != 2 or means3D.size(1) != 3:
            raise RuntimeError("means3D must have dimensions (num_points, 3)")  # rasterize_points.cu:57-59
        L = _lib.lib()
        P, H, W = means3D.size(0), int(image_height), int(image_width)
        dev = means3D.device
        means3D = _f32c(means3D, "means3D")
        background, colors, opacity = _f32c(background, "background"), _f32c(colors, "colors"), _f32c(opacity, "opacity")
        scales, rotations, cov3D_precomp = _f32c(scales, "scales"), _f32c(rotations, "rotations"), _f32c(cov3D_precomp, "cov3D_precomp")
        viewmatrix, projmatrix, campos = _f32c(viewmatrix, "viewmatrix"), _f32c(projmatrix, "projmatrix"), _f32c(campos, "campos")
        sh = _f32c(sh, "sh")
        out_color = torch.empty((NUM_CHANNELS, H, W), dtype=torch.float32, device=dev)
        radii = torch.empty((P,), dtype=torch.int32, device=dev)
        geom = torch.empty((0,), dtype=torch.uint8, device=dev)
        binning = torch.empty((0,), dtype=torch.uint8, device=dev)
        img = torch.empty((0,), dtype=torch.uint8, device=dev)
        M = sh.size(1) if sh is not None and sh.numel() != 0 else 0
        rendered = ctypes.c_int(0)
        cb_g, cb_i = _resizer(geom), _resizer(img)
        keep = [cb_g, cb_i]  # (the ctypes callbacks must outlive the call)

        def run_sync(cb_b):
            keep.append(cb_b)
            _lib.check(L.dgm_rasterize_forward(
                cb_g, None, cb_b, None, cb_i, None, P, int(degree), M, _ptr(background), W, H, _ptr(means3D),
                _ptr(sh), _ptr(colors), _ptr(opacity), _ptr(scales), float(scale_modifier), _ptr(rotations),
                _ptr(cov3D_precomp), _ptr(viewmatrix), _ptr(projmatrix), _ptr(campos), float(tan_fovx), float(tan_fovy),
                int(bool(prefiltered)), _ptr(out_color), _ptr(radii), int(bool(debug)), _stream(),
                ctypes.byref(rendered)))
            return rendered.value

        def run_cap(cb_b, cap, words):
            keep.append(cb_b)
            _lib.check(L.dgm_rasterize_forward_capacity(
                cb_g, None, cb_b, None, cb_i, None, P, int(degree), M, _ptr(background), W, H, _ptr(means3D),
                _ptr(sh), None, _ptr(colors), _ptr(opacity), _ptr(scales), float(scale_modifier), _ptr(rotations),
                _ptr(cov3D_precomp), _ptr(viewmatrix), _ptr(projmatrix), _ptr(campos), float(tan_fovx), float(tan_fovy),
                int(bool(prefiltered)), _ptr(out_color), _ptr(radii), int(bool(debug)), _stream(), int(cap), words))

        global FORWARD_CALL_SECONDS
        t_call = time.perf_counter()
        defer = DEFER_SETTLE
        with _lib.device_guard(dev):
            if P == 0 or not SYNC_FREE or defer:  # (this entry hands the caller the true R: it always settles before it returns)
                R_true = run_sync(_resizer(binning, bucket=True))
                global LAST_NUM_RENDERED
                LAST_NUM_RENDERED = R_true
            else:
                R_true, R_layout = _forward_dispatch(dev, binning, run_sync, run_cap)
                if R_layout != R_true:
                    if len(_CAP_OF) > 64:
                        _CAP_OF.clear()
                    _CAP_OF[binning.data_ptr()] = (R_layout, R_true, binning.numel())
        FORWARD_CALL_SECONDS += time.perf_counter() - t_call  # (synchronous protocol: contains the step's only host<->device sync)
        return R_true, out_color, radii, geom, binning, img

    @staticmethod
    def rasterize_gaussians_backward(background, means3D, radii, colors, scales, rotations, scale_modifier,
                                     cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color, sh,
                                     degree, campos, geomBuffer, R, binningBuffer, imageBuffer, debug):
        L = _lib.lib()
        P = means3D.size(0)
        H, W = dL_dout_color.size(1), dL_dout_color.size(2)
        dev = means3D.device
        means3D = _f32c(means3D, "means3D")
        background, colors = _f32c(background, "background"), _f32c(colors, "colors")
        scales, rotations, cov3D_precomp = _f32c(scales, "scales"), _f32c(rotations, "rotations"), _f32c(cov3D_precomp, "cov3D_precomp")
        viewmatrix, projmatrix, campos = _f32c(viewmatrix, "viewmatrix"), _f32c(projmatrix, "projmatrix"), _f32c(campos, "campos")
        sh, dL = _f32c(sh, "sh"), _f32c(dL_dout_color, "dL_dout_color")
        radii = radii.contiguous()
        M = sh.size(1) if sh is not None and sh.numel() != 0 else 0
        new = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
        # the library overwrites every element; only dL_dsh needs zeros when the SH branch is off
        dL_dmeans3D, dL_dmeans2D, dL_dcolors = new(P, 3), new(P, 3), new(P, NUM_CHANNELS)
        dL_dconic, dL_dopacity, dL_dcov3D = new(P, 2, 2), new(P, 1), new(P, 6)
        dL_dscales, dL_drotations = new(P, 3), new(P, 4)
        sh_active = M > 0 and (colors is None or colors.numel() == 0)
        dL_dsh = new(P, M, 3) if sh_active else torch.zeros((P, M, 3), dtype=torch.float32, device=dev)
        cap = _CAP_OF.get(binningBuffer.data_ptr()) if binningBuffer is not None and binningBuffer.numel() else None
        if cap is not None and cap[1] == int(R) and cap[2] == binningBuffer.numel():
            R = cap[0]  # a capacity-mode forward: the layout follows the capacity, not the frame's R
        if P != 0:
            with _lib.device_guard(dev):
                _lib.check(L.dgm_rasterize_backward(
                    P, int(degree), M, int(R), _ptr(background), W, H, _ptr(means3D), _ptr(sh), _ptr(colors),
                    _ptr(scales), float(scale_modifier), _ptr(rotations), _ptr(cov3D_precomp), _ptr(viewmatrix),
                    _ptr(projmatrix), _ptr(campos), float(tan_fovx), float(tan_fovy), _ptr(radii), _ptr(geomBuffer),
                    _ptr(binningBuffer), _ptr(imageBuffer), _ptr(dL), _ptr(dL_dmeans2D), _ptr(dL_dconic),
                    _ptr(dL_dopacity), _ptr(dL_dcolors), _ptr(dL_dmeans3D), _ptr(dL_dcov3D), _ptr(dL_dsh),
                    _ptr(dL_dscales), _ptr(dL_drotations), int(bool(debug)), _stream()))
            if RECORD_LIVE_ROWS:
                global LAST_LIVE_ROWS
                LAST_LIVE_ROWS = _count_live_rows(binningBuffer, P, W, H, int(R))
        return dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations

    @staticmethod
    def mark_visible(means3D, viewmatrix, projmatrix):
        P = means3D.size(0)
        present = torch.zeros((P,), dtype=torch.bool, device=means3D.device)
        if P != 0:
            means3D = _f32c(means3D, "means3D")
            viewmatrix, projmatrix = _f32c(viewmatrix, "viewmatrix"), _f32c(projmatrix, "projmatrix")
            with _lib.device_guard(means3D.device):
                _lib.check(_lib.lib().dgm_mark_visible(P, _ptr(means3D), _ptr(viewmatrix), _ptr(projmatrix),
                                                       _ptr(present), _stream()))
        return present


_C = _CModule()


def cpu_deep_copy_tuple(input_tuple):
    return tuple(item.cpu().clone() if isinstance(item, torch.Tensor) else item for item in input_tuple)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings)


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                raster_settings):
        rs = raster_settings
        args = (rs.bg, means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
                rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, sh,
                rs.sh_degree, rs.campos, rs.prefiltered, rs.debug)
        if rs.debug:
            cpu_args = cpu_deep_copy_tuple(args)  # copy them before they can be corrupted
            try:
                num_rendered, color, radii, geomBuffer, binningBuffer, imgBuffer = _C.rasterize_gaussians(*args)
            except Exception as ex:
                torch.save(cpu_args, "snapshot_fw.dump")
                print("\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
                raise ex
        else:
            num_rendered, color, radii, geomBuffer, binningBuffer, imgBuffer = _C.rasterize_gaussians(*args)
        ctx.raster_settings = rs
        ctx.num_rendered = num_rendered
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer,
                              binningBuffer, imgBuffer)
        ctx.mark_non_differentiable(radii)
        return color, radii

    @staticmethod
    def backward(ctx, grad_out_color, _):
        num_rendered = ctx.num_rendered
        rs = ctx.raster_settings
        colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer, binningBuffer, imgBuffer = \
            ctx.saved_tensors
        args = (rs.bg, means3D, radii, colors_precomp, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
                rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, grad_out_color, sh, rs.sh_degree, rs.campos,
                geomBuffer, num_rendered, binningBuffer, imgBuffer, rs.debug)
        if rs.debug:
            cpu_args = cpu_deep_copy_tuple(args)
            try:
                grads = _C.rasterize_gaussians_backward(*args)
            except Exception as ex:
                torch.save(cpu_args, "snapshot_bw.dump")
                print("\nAn error occured in backward. Writing snapshot_bw.dump for debugging.\n")
                raise ex
        else:
            grads = _C.rasterize_gaussians_backward(*args)
        (grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh, grad_scales,
         grad_rotations) = grads
        return (grad_means3D, grad_means2D, grad_sh, grad_colors_precomp, grad_opacities, grad_scales, grad_rotations,
                grad_cov3Ds_precomp, None)


class _RasterizeGaussiansSplitSH(torch.autograd.Function):
    """The training loop's variant of _RasterizeGaussians: SH coefficients as the model stores them, in two tensors
    (features_dc (P,1,3), features_rest (P,M-1,3)), scales / rotations given, no precomputed colours or covariances.
    Same kernels through dgm_rasterize_{forward,backward}_split_sh; what it saves is get_features' torch.cat and the two
    strided copies autograd needs to split that concatenation's gradient again."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh_dc, sh_rest, opacities, scales, rotations, raster_settings):
        rs = raster_settings
        if means3D.ndimension() != 2 or means3D.size(1) != 3:
            raise RuntimeError("means3D must have dimensions (num_points, 3)")
        L = _lib.lib()
        P, H, W = means3D.size(0), int(rs.image_height), int(rs.image_width)
        if sh_dc.shape != (P, 1, 3) or sh_rest.dim() != 3 or sh_rest.shape[0] != P or sh_rest.shape[2] != 3 or sh_rest.shape[1] < 1:
            raise RuntimeError("split SH: features_dc (P,1,3) and features_rest (P,M-1,3) required")
        dev = means3D.device
        means3D, opacities = _f32c(means3D, "means3D"), _f32c(opacities, "opacity")
        scales, rotations = _f32c(scales, "scales"), _f32c(rotations, "rotations")
        sh_dc, sh_rest = _f32c(sh_dc, "features_dc"), _f32c(sh_rest, "features_rest")
        bg, view, proj, campos = (_f32c(rs.bg, "background"), _f32c(rs.viewmatrix, "viewmatrix"),
                                  _f32c(rs.projmatrix, "projmatrix"), _f32c(rs.campos, "campos"))
        out_color = torch.empty((NUM_CHANNELS, H, W), dtype=torch.float32, device=dev)
        radii = torch.empty((P,), dtype=torch.int32, device=dev)
        geom, binning, img = (torch.empty((0,), dtype=torch.uint8, device=dev) for _ in range(3))
        M = 1 + sh_rest.shape[1]
        rendered = ctypes.c_int(0)
        cb_g, cb_i = _resizer(geom), _resizer(img)
        keep = [cb_g, cb_i]

        def run_sync(cb_b):
            keep.append(cb_b)
            _lib.check(L.dgm_rasterize_forward_split_sh(
                cb_g, None, cb_b, None, cb_i, None, P, int(rs.sh_degree), M, _ptr(bg), W, H, _ptr(means3D), _ptr(sh_dc),
                _ptr(sh_rest), None, _ptr(opacities), _ptr(scales), float(rs.scale_modifier), _ptr(rotations), None, _ptr(view),
                _ptr(proj), _ptr(campos), float(rs.tanfovx), float(rs.tanfovy), int(bool(rs.prefiltered)), _ptr(out_color),
                _ptr(radii), int(bool(rs.debug)), _stream(), ctypes.byref(rendered)))
            return rendered.value

        def run_cap(cb_b, cap, words):
            keep.append(cb_b)
            _lib.check(L.dgm_rasterize_forward_capacity(
                cb_g, None, cb_b, None, cb_i, None, P, int(rs.sh_degree), M, _ptr(bg), W, H, _ptr(means3D), _ptr(sh_dc),
                _ptr(sh_rest), None, _ptr(opacities), _ptr(scales), float(rs.scale_modifier), _ptr(rotations), None, _ptr(view),
                _ptr(proj), _ptr(campos), float(rs.tanfovx), float(rs.tanfovy), int(bool(rs.prefiltered)), _ptr(out_color),
                _ptr(radii), int(bool(rs.debug)), _stream(), int(cap), words))

        global FORWARD_CALL_SECONDS
        t_call = time.perf_counter()
        with _lib.device_guard(dev):
            if P == 0:
                R_layout = run_sync(_resizer(binning, bucket=True))
            else:
                _, R_layout = _forward_dispatch(dev, binning, run_sync, run_cap)
        FORWARD_CALL_SECONDS += time.perf_counter() - t_call
        ctx.raster_settings, ctx.num_rendered, ctx.consts = rs, R_layout, (bg, view, proj, campos)
        ctx.save_for_backward(means3D, scales, rotations, radii, sh_dc, sh_rest, geom, binning, img)
        ctx.mark_non_differentiable(radii)
        return out_color, radii

    @staticmethod
    def backward(ctx, grad_out_color, _):
        L = _lib.lib()
        rs = ctx.raster_settings
        means3D, scales, rotations, radii, sh_dc, sh_rest, geom, binning, img = ctx.saved_tensors
        bg, view, proj, campos = ctx.consts
        P, dev = means3D.size(0), means3D.device
        dL = _f32c(grad_out_color, "dL_dout_color")
        H, W = dL.size(1), dL.size(2)
        M = 1 + sh_rest.shape[1]
        new = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
        dL_dmeans3D, dL_dmeans2D, dL_dcolors = new(P, 3), new(P, 3), new(P, NUM_CHANNELS)
        dL_dconic, dL_dopacity, dL_dcov3D = new(P, 2, 2), new(P, 1), new(P, 6)
        dL_dscales, dL_drotations = new(P, 3), new(P, 4)
        dL_ddc, dL_drest = torch.empty_like(sh_dc), torch.empty_like(sh_rest)
        if P != 0:
            with _lib.device_guard(dev):
                _lib.check(L.dgm_rasterize_backward_split_sh(
                    P, int(rs.sh_degree), M, int(ctx.num_rendered), _ptr(bg), W, H, _ptr(means3D), _ptr(sh_dc), _ptr(sh_rest),
                    None, _ptr(scales), float(rs.scale_modifier), _ptr(rotations), None, _ptr(view), _ptr(proj), _ptr(campos),
                    float(rs.tanfovx), float(rs.tanfovy), _ptr(radii), _ptr(geom), _ptr(binning), _ptr(img), _ptr(dL),
                    _ptr(dL_dmeans2D), _ptr(dL_dconic), _ptr(dL_dopacity), _ptr(dL_dcolors), _ptr(dL_dmeans3D), _ptr(dL_dcov3D),
                    _ptr(dL_ddc), _ptr(dL_drest), _ptr(dL_dscales), _ptr(dL_drotations), int(bool(rs.debug)), _stream()))
            if RECORD_LIVE_ROWS:
                global LAST_LIVE_ROWS
                LAST_LIVE_ROWS = _count_live_rows(binning, P, W, H, int(ctx.num_rendered))
        return dL_dmeans3D, dL_dmeans2D, dL_ddc, dL_drest, dL_dopacity, dL_dscales, dL_drotations, None


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        with torch.no_grad():
            rs = self.raster_settings
            return _C.mark_visible(positions, rs.viewmatrix, rs.projmatrix)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        rs = self.raster_settings
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        empty = torch.Tensor([])
        shs = empty if shs is None else shs
        colors_precomp = empty if colors_precomp is None else colors_precomp
        scales = empty if scales is None else scales
        rotations = empty if rotations is None else rotations
        cov3D_precomp = empty if cov3D_precomp is None else cov3D_precomp
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, rs)

    def forward_split_sh(self, means3D, means2D, opacities, shs_dc, shs_rest, scales, rotations):
        """forward(shs=cat(shs_dc, shs_rest, 1), scales=..., rotations=...) without forming the concatenation: the two SH
        tensors of the model (features_dc (P,1,3), features_rest (P,M-1,3)) go to the kernels as they are."""
        return _RasterizeGaussiansSplitSH.apply(means3D, means2D, shs_dc, shs_rest, opacities, scales, rotations,
                                                self.raster_settings)
