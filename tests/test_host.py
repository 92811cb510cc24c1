"""CPU-side tests of the host layer: the C-ABI library builds for gfx950, loads, and exports every symbol the
header declares (no compute without a GPU); the state layout is a pure function of sizes; the Python API
mirrors the reference's names / argument checks and refuses to run without the HIP path."""
import ctypes
import inspect
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, pkg


def test_build_and_exports():
    L = pkg("_lib")
    L.build()
    assert os.path.exists(L.LIB_PATH)
    lib = L.lib()
    header = open(os.path.join(ROOT, "include", "dgmesh_hip.h")).read()
    declared = set(re.findall(r"\b(dgm_[a-z0-9_]+)\s*\(", header))
    declared -= {"dgm_alloc_fn"}
    assert declared, "no declarations parsed"
    assert declared == set(L.SYMBOLS), (declared ^ set(L.SYMBOLS))
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.dgm_abi_version() == 5
    names = [lib.dgm_stage_name(i).decode() for i in range(L.STAGE_COUNT)]
    assert names[0] == "preprocess_fwd" and names[7] == "preprocess_bwd" and names[-2] == "mlp_layer_dw" and names[-1] == "mlp_bwd_pair"


def test_state_layout_is_pure_and_aligned():
    L = pkg("_lib")
    lib = L.lib()
    a, b = L.StateLayout(), L.StateLayout()
    assert lib.dgm_describe_state(100000, 800, 800, 2780000, ctypes.byref(a)) == 0
    assert lib.dgm_describe_state(100000, 800, 800, 2780000, ctypes.byref(b)) == 0
    for f, _ in L.StateLayout._fields_:
        assert getattr(a, f) == getattr(b, f)
    assert (a.tiles_x, a.tiles_y) == (50, 50)
    assert a.n_chunks * a.chunk_size >= 100000 and a.chunk_size % 512 == 0 and a.n_chunks <= 256
    offs = [a.rec, a.depth, a.radii, a.tiles_touched, a.offs, a.cov3D, a.clamped, a.block_sums, a.hist,
            a.tile_count, a.tile_offset, a.big_list, a.counters]
    assert offs == sorted(offs) and all(o % 256 == 0 for o in offs)
    assert a.depth - a.rec >= 100000 * 48
    assert a.geometry_bytes == lib.dgm_geometry_bytes(100000, 800, 800)
    assert a.binning_bytes == lib.dgm_binning_bytes(2780000) and a.binning_bytes >= 2780000 * (8 + 4 + 36 + 1)
    assert a.image_bytes == lib.dgm_image_bytes(800, 800)
    # geometry / image sizes must not depend on R (backward re-derives them)
    assert lib.dgm_describe_state(100000, 800, 800, 5, ctypes.byref(b)) == 0
    assert b.geometry_bytes == a.geometry_bytes and b.image_bytes == a.image_bytes and b.hist == a.hist


def test_c_abi_argument_errors_without_gpu():
    """Argument validation happens before any HIP call, so it is testable on a CPU-only box."""
    L = pkg("_lib")
    lib = L.lib()
    cb = L.ALLOC_FN(lambda ctx, n: 0)
    n = ctypes.c_int(-1)
    st = lib.dgm_rasterize_forward(cb, None, cb, None, cb, None, -1, 3, 16, None, 64, 64, None, None, None, None, None,
                                   1.0, None, None, None, None, None, 1.0, 1.0, 0, None, None, 0, None, ctypes.byref(n))
    assert st != 0 and b"bad sizes" in lib.dgm_last_error()
    null = ctypes.cast(None, L.ALLOC_FN)
    st = lib.dgm_rasterize_forward(null, None, null, None, null, None, 10, 3, 16, None, 64, 64, None, None, None, None,
                                   None, 1.0, None, None, None, None, None, 1.0, 1.0, 0, None, None, 0, None,
                                   ctypes.byref(n))
    assert st != 0 and b"allocator" in lib.dgm_last_error()
    assert lib.dgm_knn_mean_dist2(0, None, None, None, None) == 0      # P == 0 is a no-op
    assert lib.dgm_knn_mean_dist2(5, None, None, None, None) != 0
    assert lib.dgm_knn_scratch_bytes(0) == 0 and lib.dgm_knn_scratch_bytes(100000) > 100000 * 16  # caller-owned scratch, sized by the library
    assert lib.dgm_mark_visible(0, None, None, None, None, None) == 0


def test_python_api_surface_matches_reference():
    import diff_gaussian_rasterization as D
    import simple_knn._C as K

    assert D.GaussianRasterizationSettings._fields == (
        "image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix", "projmatrix",
        "sh_degree", "campos", "prefiltered", "debug")
    sig = inspect.signature(D.GaussianRasterizer.forward)
    assert list(sig.parameters) == ["self", "means3D", "means2D", "opacities", "shs", "colors_precomp", "scales",
                                    "rotations", "cov3D_precomp"]
    assert list(inspect.signature(D._C.rasterize_gaussians).parameters) == [
        "background", "means3D", "colors", "opacity", "scales", "rotations", "scale_modifier", "cov3D_precomp",
        "viewmatrix", "projmatrix", "tan_fovx", "tan_fovy", "image_height", "image_width", "sh", "degree", "campos",
        "prefiltered", "debug"]
    assert list(inspect.signature(D._C.rasterize_gaussians_backward).parameters) == [
        "background", "means3D", "radii", "colors", "scales", "rotations", "scale_modifier", "cov3D_precomp",
        "viewmatrix", "projmatrix", "tan_fovx", "tan_fovy", "dL_dout_color", "sh", "degree", "campos", "geomBuffer",
        "R", "binningBuffer", "imageBuffer", "debug"]
    assert callable(D._C.mark_visible) and callable(K.distCUDA2)
    rs = D.GaussianRasterizationSettings(8, 8, 1.0, 1.0, torch.zeros(3), 1.0, torch.eye(4), torch.eye(4), 3,
                                         torch.zeros(3), False, False)
    r = D.GaussianRasterizer(rs)
    x = torch.zeros(4, 3)
    with pytest.raises(Exception, match="SHs or precomputed colors"):
        r(means3D=x, means2D=x, opacities=torch.zeros(4, 1), scales=x, rotations=torch.zeros(4, 4))
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(means3D=x, means2D=x, opacities=torch.zeros(4, 1), shs=torch.zeros(4, 16, 3))
    # no CPU path: CPU tensors are refused loudly instead of being computed some other way
    with pytest.raises(RuntimeError, match="no CPU path"):
        r(means3D=x, means2D=x, opacities=torch.zeros(4, 1), shs=torch.zeros(4, 16, 3), scales=x,
          rotations=torch.zeros(4, 4))
    with pytest.raises(RuntimeError, match="no CPU path"):
        K.distCUDA2(x)


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under the product package may reference it."""
    pk = os.path.join(ROOT, "dg-mesh_amd")
    for dp, _, files in os.walk(pk):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")) or f == "Makefile":
                src = open(os.path.join(dp, f)).read()
                assert "oracle" not in src.replace("the oracle", "").replace("oracle f2i_sat", "").replace(
                    "to the oracle", "").replace("vs oracle", ""), os.path.join(dp, f)


def test_synthetic_camera_conventions(syn):
    cam = syn.make_camera(800, 800)
    w2c = cam.world_view_transform.T.astype(np.float64)
    # camera looks at the origin: origin maps to (0,0,+radius)
    o = w2c @ np.array([0, 0, 0, 1.0])
    np.testing.assert_allclose(o[:3], [0, 0, 4.0], atol=1e-5)
    np.testing.assert_allclose(np.linalg.det(w2c[:3, :3]), 1.0, atol=1e-5)
    np.testing.assert_allclose(-w2c[:3, :3].T @ w2c[:3, 3], cam.camera_center, atol=1e-5)
    # full projection: w component equals view-space z (P[3,2] = 1)
    p = np.array([0.3, -0.2, 0.1, 1.0]) @ cam.full_proj_transform.astype(np.float64)
    v = np.array([0.3, -0.2, 0.1, 1.0]) @ cam.world_view_transform.astype(np.float64)
    np.testing.assert_allclose(p[3], v[2], rtol=1e-5)
    g = syn.make_gaussians(500, seed=0)
    assert g["features_rest"].shape == (500, 15, 3) and np.allclose(g["rotation"][:, 0], 1)
    a = syn.activate(g)
    np.testing.assert_allclose(a["opacities"], 0.1, rtol=1e-5)
    np.testing.assert_allclose(np.linalg.norm(a["rotations"], axis=1), 1, rtol=1e-5)


def test_densification_stats_mask_form_equals_indexing():
    """GaussianModel.track_densification_stats (mask arithmetic, no nonzero()) == the reference's boolean-index form
    (train.py:489-496 + add_densification_stats)."""
    import torch
    S = pkg("scene")
    P = 50
    g1, g2 = S.GaussianModel(sh_degree=3, device="cpu"), S.GaussianModel(sh_degree=3, device="cpu")
    rng = np.random.RandomState(0)
    for g in (g1, g2):
        g.load_raw(rng.randn(P, 3), rng.randn(P, 1, 3), rng.randn(P, 15, 3), rng.randn(P, 3), rng.randn(P, 4), rng.randn(P, 1))
        g.training_setup(S.OptimizationParams())
    for step in range(3):
        vp = torch.zeros(P, 3, requires_grad=True)
        vp.grad = torch.tensor(rng.randn(P, 3).astype(np.float32))
        radii = torch.tensor(rng.randint(0, 40, P).astype(np.int32))
        vis = radii > 0
        g1.max_radii2D[vis] = torch.max(g1.max_radii2D[vis], radii[vis].to(g1.max_radii2D.dtype))
        g1.add_densification_stats(vp, vis)
        g2.track_densification_stats(vp, vis, radii)
    assert torch.equal(g1.max_radii2D, g2.max_radii2D)
    assert torch.allclose(g1.xyz_gradient_accum, g2.xyz_gradient_accum) and torch.equal(g1.denom, g2.denom)


def test_time_noise_schedule_and_shape():
    """get_linear_noise_func: linear (not log) interpolation with the delayed warm-up factor, as the reference's
    smooth_term (train.py:119-121); Trainer.time_input keeps a stride-0 view and only adds noise for real scenes."""
    import torch
    D, T = pkg("deform"), pkg("trainer")
    f = D.get_linear_noise_func(lr_init=0.1, lr_final=1e-15, lr_delay_mult=0.01, max_steps=20000)
    assert abs(f(0) - 0.1) < 1e-12 and abs(f(10000) - 0.05) < 1e-9 and f(20000) < 1e-12 and f(10 ** 9) < 1e-12
    g = D.get_linear_noise_func(1.0, 0.0, lr_delay_steps=100, lr_delay_mult=0.5, max_steps=1000)
    assert abs(g(0) - 0.5) < 1e-12 and abs(g(100) - 0.9) < 1e-9

    class Cam:
        fid = torch.tensor([0.25])

    tr = T.Trainer.__new__(T.Trainer)
    tr.is_blender, tr.time_interval, tr.smooth_term = True, 0.01, f
    t = tr.time_input(Cam, 7, 5000)
    assert t.shape == (7, 1) and t.stride(0) == 0 and float(t[3, 0]) == 0.25
    tr.is_blender = False
    torch.manual_seed(0)
    t2 = tr.time_input(Cam, 7, 5000)
    assert t2.stride(0) == 0 and float(t2[0, 0]) != 0.25 and abs(float(t2[0, 0]) - 0.25) < 0.01 * 0.1 * 6
    assert torch.equal(t2[0], t2[6])


def test_pack_heads_keeps_the_module_and_makes_the_heads_adjacent():
    """mlp_hip.pack_heads: head weights / biases re-homed back to back in output order (what lets the fused kernels read
    [Wh | bh] in place, without torch.cat) -- same outputs, same state_dict names and shapes, the layout survives
    load_state_dict, and _stacked() is then a view of the parameters' own memory."""
    import torch
    D, M = pkg("deform"), pkg("mlp_hip")
    torch.manual_seed(0)
    net = D.DeformNetworkNormal(is_blender=True, trunk_impl="torch")
    x, t = torch.randn(17, 3), torch.full((17, 1), 0.3)
    before = [o.detach().clone() for o in net(x, t)]
    keys = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
    heads = net.head_modules()
    assert not M._adjacent([m.weight for m in heads])
    ids = [id(p) for p in net.parameters()]
    M.pack_heads(heads)
    assert M._adjacent([m.weight for m in heads]) and M._adjacent([m.bias for m in heads])
    assert ids == [id(p) for p in net.parameters()]
    assert keys == [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
    for a, b in zip(before, net(x, t)):
        assert torch.equal(a, b)
    n_out = sum(m.weight.shape[0] for m in heads)
    Wh = M._stacked([m.weight for m in heads], (n_out, 256))
    assert Wh.data_ptr() == heads[0].weight.data_ptr() and torch.equal(Wh, torch.cat([m.weight for m in heads], 0))
    bh = M._stacked([m.bias for m in heads], (n_out,))
    assert bh.data_ptr() == heads[0].bias.data_ptr() and torch.equal(bh, torch.cat([m.bias for m in heads], 0))
    sd = {k: torch.randn_like(v) for k, v in net.state_dict().items()}
    net.load_state_dict(sd)
    assert M._adjacent([m.weight for m in heads]) and torch.equal(heads[1].weight, sd["gaussian_rotation.weight"])
    # an optimizer step through the parameters is visible through the stacked view (no stale copy)
    with torch.no_grad():
        heads[2].weight.add_(1.0)
    assert torch.equal(M._stacked([m.weight for m in heads], (n_out, 256)), torch.cat([m.weight for m in heads], 0))


def test_binning_buffer_sizes_are_bucketed():
    """rasterizer._bucket: exact up to 1 MiB, then m * 2^k with m in 8..15 (<= 12.5 % more), one size per octave above 2 GiB; monotone,
    never below the request -- the binning buffer's size follows R and must not miss the caching allocator at every new maximum."""
    R = pkg("rasterizer")
    assert [R._bucket(n) for n in (0, 1, 4096, 1 << 20)] == [0, 1, 4096, 1 << 20]
    last = 0
    for n in list(range((1 << 20) + 1, 1 << 22, 65537)) + [10 ** 8, 5 * 10 ** 8, (1 << 31), (1 << 31) + 1, 13 * 10 ** 9]:
        b = R._bucket(n)
        assert b >= n and b >= last
        last = b
        if n <= (1 << 31):
            assert b <= n * 1.125 + 1 and (b >> (b.bit_length() - 4)) << (b.bit_length() - 4) == b
        else:
            assert b & (b - 1) == 0 and b < 2 * n
    assert len({R._bucket(n) for n in range(400 * 10 ** 6, 500 * 10 ** 6, 10 ** 6)}) <= 4


def test_graft_entry_build_checks_the_current_abi():
    """__graft_entry__.build() must accept exactly the library it builds (it once pinned a literal ABI number)."""
    import re
    src = open(os.path.join(ROOT, "__graft_entry__.py")).read()
    assert "lib.ABI_VERSION" in src and not re.search(r"dgm_abi_version\(\)\s*==\s*\d", src)
    hdr = open(os.path.join(ROOT, "include", "dgmesh_hip.h")).read()
    assert int(re.search(r"#define DGM_ABI_VERSION (\d+)", hdr).group(1)) == pkg("_lib").ABI_VERSION
