"""The evaluation / rendering entry (models/neuradar.py:905-969, get_outputs_for_camera_ray_bundle) on the HIP path: strided
camera rays, ragged chunks, eval-mode samplers, one decode over the reading -- against the CPU oracle's eval pipeline and
against the same model evaluated in one piece."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from test_gpu_fullsize import _check, _oracle_params  # noqa: E402


def _model(chunk, decoders=True):
    from neuradar_amd.neurad_encoding import NeuRADHashEncodingConfig, StaticSettings
    from neuradar_amd.neurad_field import NeuRADFieldConfig
    from neuradar_amd.step import HotPathConfig, NeuRadarHotPath

    torch.manual_seed(3)
    cfg = HotPathConfig(field=NeuRADFieldConfig(grid=NeuRADHashEncodingConfig(static=StaticSettings(log2_hashmap_size=15))),
                        decoders=decoders, eval_num_rays_per_chunk=chunk)
    cfg.proposal_field_1.grid.static.log2_hashmap_size = 14
    cfg.proposal_field_2.grid.static.log2_hashmap_size = 14
    model = NeuRadarHotPath(cfg).to(DEV).eval()
    with torch.no_grad():  # a scene with structure: tables away from their near-zero initialisation
        model.field.hashgrid.static_grid.hash_table.uniform_(-0.5, 0.5)
        model.proposal_fields[1].hashgrid.static_grid.hash_table.uniform_(-0.5, 0.5)
    return model


def _camera_rays(H, W, gen):
    from neuradar_amd.rays import RayBundle

    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    d = torch.stack([(xs - W / 2) / W, (ys - H / 2) / W, torch.ones_like(xs)], -1).reshape(-1, 3)
    d = d / d.norm(dim=-1, keepdim=True)
    o = torch.tensor([1.0, -2.0, 0.5]).expand_as(d) + 0.01 * torch.randn(H * W, 3, generator=gen)
    area = torch.full((H * W, 1), 1e-6) * (1 + torch.rand(H * W, 1, generator=gen))
    fars = 40.0 + 200.0 * torch.rand(H * W, 1, generator=gen)
    return RayBundle(o.to(DEV).contiguous(), d.to(DEV).contiguous(), area.to(DEV), fars=fars.to(DEV))


def test_camera_image_in_ragged_chunks_vs_oracle_and_one_piece():
    from oracle import pipeline as op

    gen = torch.Generator().manual_seed(0)
    H, W = 26, 31  # not multiples of the upsampling factor: rows 1, 4, ..., 25 and columns 1, 4, ..., 28
    model = _model(chunk=37)
    bundle = _camera_rays(H, W, gen)
    keep = dict(o=bundle.origins.clone(), d=bundle.directions.clone(), a=bundle.pixel_area.clone(), f=bundle.fars.clone())
    out = model.get_outputs_for_camera_ray_bundle(bundle, image_shape=(H, W))
    h, w = len(range(1, H, 3)), len(range(1, W, 3))
    assert out["features"].shape == (h, w, 32) and out["depth"].shape == (h, w, 1) and out["accumulation"].shape == (h, w, 1)
    assert out["rgb"].shape == (3 * h, 3 * w, 3) and out["intensity"].shape == (h, w, 1) and out["ray_drop_prob"].shape == (h, w, 1)
    assert float(out["rgb"].min()) >= 0.0 and float(out["rgb"].max()) <= 1.0
    # the strided rays through the CPU oracle's eval pipeline (bin edges at linspace, no jitter)
    idx = (torch.arange(1, H, 3)[:, None] * W + torch.arange(1, W, 3)[None, :]).reshape(-1).to(DEV)
    fp, pp = _oracle_params(model)
    with torch.no_grad():
        ref = op.nff_outputs(fp, [pp, pp], dict(origins=keep["o"][idx].cpu(), directions=keep["d"][idx].cpu(),
                                               pixel_area=keep["a"][idx].cpu(), fars=keep["f"][idx].cpu()))
    _check(out["features"].reshape(-1, 32).cpu(), ref["features"], "rendered features (eval)", few=5e-3)  # (2 880 values: 4 of them = 1.4e-3)
    _check(out["depth"].reshape(-1, 1).cpu(), ref["depth"], "depth (eval)")
    _check(out["accumulation"].reshape(-1, 1).cpu(), ref["accumulation"], "accumulation (eval)")
    _check(out["prop_depth_1"].reshape(-1, 1).cpu(), ref["prop_depth_1"], "proposal depth (eval)", rtol=1e-3)
    # one piece: the same numbers (every ray is independent of its chunk)
    model.config.eval_num_rays_per_chunk = 1 << 15
    bundle2 = _camera_rays(H, W, torch.Generator().manual_seed(0))
    whole = model.get_outputs_for_camera_ray_bundle(bundle2, image_shape=(H, W))
    for k in ("features", "depth", "accumulation", "rgb", "intensity"):
        assert torch.equal(out[k], whole[k]), k
    # without the upsampling compensation every ray is shot and the image comes out 3x as large
    model.config.compensate_upsampling_when_rendering = False
    full = model.get_outputs_for_camera_ray_bundle(_camera_rays(6, 5, gen), image_shape=(6, 5))
    assert full["features"].shape == (6, 5, 32) and full["rgb"].shape == (18, 15, 3)


def test_lidar_and_radar_readings():
    from neuradar_amd.rays import RayBundle

    gen = torch.Generator().manual_seed(1)
    model = _model(chunk=64)
    n = 2 * 107  # two radar scans of 107 rays
    base = _camera_rays(n, 1, gen)
    sph = torch.stack([(torch.rand(n, generator=gen) - 0.5) * 1.5, (torch.rand(n, generator=gen) - 0.5) * 0.4], 1).to(DEV)
    lidar = RayBundle(base.origins, base.directions, base.pixel_area, fars=base.fars.clone(),
                      metadata={"is_lidar": torch.ones(n, 1, dtype=torch.bool, device=DEV)})
    out = model.get_outputs_for_camera_ray_bundle(lidar)
    assert out["depth"].shape == (n, 1) and out["intensity"].shape == (n, 1) and "rgb" not in out and "radar_output" not in out
    assert float(out["intensity"].min()) > 0.0 and float(out["intensity"].max()) < 1.0
    radar = RayBundle(base.origins, base.directions, base.pixel_area, fars=base.fars.clone(),
                      metadata={"is_radar": torch.ones(n, 1, dtype=torch.bool, device=DEV), "directions_spher": sph})
    ro = model.get_outputs_for_camera_ray_bundle(radar, num_radar_scans=2)
    assert ro["radar_output"].shape == (2, 107, 7) and bool(torch.isfinite(ro["radar_output"]).all())
    assert torch.equal(ro["depth"], out["depth"])  # same rays, same field: the sensor only selects the decoder
    # the decoded points are the rendered points plus a bounded offset (neuradar.py:486: 1.5 * tanh)
    xyz = ro["depth"].view(2, 107, 1) * torch.stack([torch.cos(sph[:, 0]) * torch.cos(sph[:, 1]), torch.sin(sph[:, 0]) * torch.cos(sph[:, 1]),
                                                     torch.sin(sph[:, 1])], -1).view(2, 107, 3)
    assert float((ro["radar_output"][..., 1:4] - xyz).abs().max()) <= 1.5 + 1e-4
    with pytest.raises(AssertionError):
        model.train().get_outputs_for_camera_ray_bundle(lidar)


@pytest.mark.parametrize("mlp_dtype", ["float32", "bfloat16"])
def test_fused_render_chain_equals_the_modular_eval_path(mlp_dtype, monkeypatch):
    """The forward-only launch chain of the rendering entry (fused_render.FusedRenderer: fused per-ray launches over
    preallocated buffers, sample-major rows, the main gather inside the field forward on 16-bit operands) against the
    modular modules' eval path (NR_FUSED_RENDER=0) on the same rays: the fused launches equal the chains they replace to
    2e-5 (tests/test_gpu_parity.py), so the rendered values agree to 1e-4 of their scale -- with ragged chunks, and for a
    lidar reading (rays without the camera's constant far plane)."""
    from neuradar_amd.rays import RayBundle

    gen = torch.Generator().manual_seed(7)
    H, W = 23, 40
    model = _model(chunk=101)
    model.field.config.mlp_dtype = mlp_dtype
    outs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("NR_FUSED_RENDER", mode)
        cam = model.get_outputs_for_camera_ray_bundle(_camera_rays(H, W, torch.Generator().manual_seed(7)), image_shape=(H, W))
        base = _camera_rays(150, 1, torch.Generator().manual_seed(8))
        lid = model.get_outputs_for_camera_ray_bundle(RayBundle(base.origins, base.directions, base.pixel_area, fars=base.fars.clone(),
                                                                metadata={"is_lidar": torch.ones(150, 1, dtype=torch.bool, device=DEV)}))
        outs[mode] = (cam, lid)
        assert (model._fused_renderer(101) is not None) == (mode == "1")
    for a, b, what in ((outs["1"][0], outs["0"][0], "camera"), (outs["1"][1], outs["0"][1], "lidar")):
        for k in ("features", "depth", "accumulation", "prop_depth_0", "prop_depth_1", "intensity"):
            _check(a[k].reshape(-1, a[k].shape[-1]).cpu(), b[k].reshape(-1, b[k].shape[-1]).cpu(), f"{what} {k} ({mlp_dtype})", rtol=1e-4, floor=1e-5,
                   few=5e-3)
    _check(outs["1"][0]["rgb"].reshape(-1, 3).cpu(), outs["0"][0]["rgb"].reshape(-1, 3).cpu(), "rgb", rtol=1e-3, floor=1e-4, few=5e-3)


@pytest.mark.parametrize("mlp_dtype", ["bfloat16", "float32"])
def test_render_after_raw_pointer_optimizer_steps_sees_the_new_parameters(mlp_dtype, monkeypatch):
    """render -> FlatAdam steps -> render.  FlatAdam (nr_adam_step) and the batch norms' running statistics
    (nr_bn_act_fwd) write through raw pointers, so torch's `_version` counters never move: the fused renderer's packed MLP image
    is rebuilt per rendered reading (one 23-us launch) and the CNN's BN-folded 7 x 7 images when the library's device-side
    parameter-generation word has moved (test_cnn_weight_images_are_rebuilt_exactly_when_parameters_moved).  The second
    fused render must equal the modular eval path (NR_FUSED_RENDER=0, which packs per call) on the UPDATED parameters -- and differ
    from the first render."""
    from neuradar_amd.step import FlatAdam

    H, W = 23, 40
    model = _model(chunk=101)
    model.field.config.mlp_dtype = mlp_dtype
    cam = lambda: _camera_rays(H, W, torch.Generator().manual_seed(7))  # noqa: E731
    monkeypatch.setenv("NR_FUSED_RENDER", "1")
    first = model.get_outputs_for_camera_ray_bundle(cam(), image_shape=(H, W))
    # "training": every field / decoder parameter moves through nr_adam_step on a synthetic gradient; running statistics of
    # the CNN's batch norms move through nr_bn_act_fwd itself
    params = [p for n_, p in model.named_parameters() if "hash_table" not in n_ and p.requires_grad]
    opt = FlatAdam(params, lr=5e-2, warmup_steps=0)
    versions = [p._version for p in params]
    torch.manual_seed(11)
    for _ in range(3):
        for _, g in opt.buffers:
            g.copy_(torch.randn_like(g))
        opt.advance()
        for i in range(len(opt.buffers)):
            opt.step_buffer(i)
    assert [p._version for p in params] == versions, "the premise: raw-pointer updates do not bump version counters"
    from neuradar_amd import ops

    for m_ in model._decoders.rgb_decoder.modules():
        if isinstance(m_, torch.nn.BatchNorm2d):  # a training-mode nr_bn_act_fwd: running statistics move, their version does not
            v0 = (m_.running_mean._version, m_.running_var._version)
            x = (3.0 * torch.randn(2, m_.num_features, 8, 8, device=DEV) + 1.0).contiguous(memory_format=torch.channels_last)
            with torch.no_grad():
                ops.bn_act(x, m_.weight, m_.bias, m_.running_mean, m_.running_var, None, 0.5, m_.eps, True)
            assert (m_.running_mean._version, m_.running_var._version) == v0
    torch.cuda.synchronize()
    second = model.get_outputs_for_camera_ray_bundle(cam(), image_shape=(H, W))
    monkeypatch.setenv("NR_FUSED_RENDER", "0")
    modular = model.get_outputs_for_camera_ray_bundle(cam(), image_shape=(H, W))
    for k in ("features", "depth", "accumulation", "intensity"):
        _check(second[k].reshape(-1, second[k].shape[-1]).cpu(), modular[k].reshape(-1, modular[k].shape[-1]).cpu(),
               f"{k} after the update ({mlp_dtype})", rtol=1e-4, floor=1e-5, few=5e-3)
    _check(second["rgb"].reshape(-1, 3).cpu(), modular["rgb"].reshape(-1, 3).cpu(), "rgb after the update", rtol=1e-3, floor=1e-4, few=5e-3)
    assert float((second["features"] - first["features"]).abs().max()) > 1e-3, "the update did not reach the rendered features"
    assert float((second["rgb"] - first["rgb"]).abs().max()) > 1e-3, "the update did not reach the image"


def test_cnn_weight_images_are_rebuilt_exactly_when_parameters_moved():
    """Round 6: the BN-folded 7 x 7 weight images of the rendering entry are built by ONE launch
    (nr_conv7_fold_pack) that early-outs ON THE DEVICE while the library's parameter-generation word -- bumped by every optimizer
    step (nr_adam_hyper), sharded delta apply and running-statistics update, graph replays included -- is what the images were
    built from.  Checked through the rebuild counter the kernel keeps (state[1]) and against a forced rebuild:
    render -> render without training: NO rebuild; after FlatAdam steps (raw pointers, version counters unmoved): one rebuild and
    the images equal a forced fold of the current parameters; after a training-mode nr_bn_act_fwd alone: rebuilt; after a
    torch-side in-place write (version counter): rebuilt; after an optimizer step inside a REPLAYED hipGraph: rebuilt."""
    from neuradar_amd import ops
    from neuradar_amd.step import FlatAdam

    H, W = 23, 40
    model = _model(chunk=101)
    model.field.config.mlp_dtype = "bfloat16"
    dec = model._decoders
    cam = lambda: _camera_rays(H, W, torch.Generator().manual_seed(7))  # noqa: E731
    render = lambda: model.get_outputs_for_camera_ray_bundle(cam(), image_shape=(H, W))  # noqa: E731

    def rebuilds():
        torch.cuda.synchronize()
        return int(dec._conv7_eval_cache["state"][1])

    def images_are_current():
        have = dec._conv7_eval_cache["images"][:, 0].clone()
        dec._conv7_eval_cache["key"] = None  # forces the next fold
        dec.prepare_conv7_eval(torch.bfloat16)
        torch.cuda.synchronize()
        return torch.equal(have, dec._conv7_eval_cache["images"][:, 0])

    first = render()
    n0 = rebuilds()
    assert n0 == 1
    again = render()
    assert rebuilds() == n0, "render -> render without training must not rebuild the weight images"
    assert torch.equal(first["rgb"], again["rgb"])
    params = [p for p in dec.rgb_decoder.parameters() if p.requires_grad]
    opt = FlatAdam(params, lr=5e-2, warmup_steps=0)
    n0 = rebuilds()  # (FlatAdam rebinds the parameters into its flat buffer: new data pointers -> the host key forces one rebuild)
    versions = [p._version for p in params]

    def train_step():
        for _, g in opt.buffers:
            g.copy_(torch.randn_like(g))
        opt.advance()
        for i in range(len(opt.buffers)):
            opt.step_buffer(i)

    torch.manual_seed(5)
    render()
    n1 = rebuilds()
    render()
    assert rebuilds() == n1
    train_step()
    assert [p._version for p in params] == versions, "the premise: raw-pointer updates do not bump version counters"
    second = render()
    assert rebuilds() == n1 + 1, "an optimizer step through raw pointers must be seen"
    assert float((second["rgb"] - again["rgb"]).abs().max()) > 1e-3
    n2 = rebuilds()
    assert images_are_current()
    n2 = rebuilds()
    # running statistics alone
    bn = next(m for m in dec.rgb_decoder.modules() if isinstance(m, torch.nn.BatchNorm2d))
    x = (3.0 * torch.randn(2, bn.num_features, 8, 8, device=DEV) + 1.0).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        ops.bn_act(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, None, 0.5, bn.eps, True)
    render()
    assert rebuilds() == n2 + 1, "a running-statistics update through nr_bn_act_fwd must be seen"
    assert images_are_current()
    n3 = rebuilds()
    # a torch-side write: the version counter
    conv7 = next(m for m in dec.rgb_decoder.modules() if isinstance(m, torch.nn.Conv2d) and m.kernel_size == (7, 7))
    with torch.no_grad():
        conv7.weight.mul_(1.01)
    render()
    assert rebuilds() == n3 + 1 and images_are_current()
    n3 = rebuilds()
    with torch.no_grad():  # ... and a parameter the images do not depend on (the 1 x 1 head) leaves them alone
        params[0].mul_(1.01)
    render()
    assert rebuilds() == n3
    n4 = rebuilds()
    # an optimizer step inside a replayed hipGraph: no Python runs, the device word still moves
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        train_step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    render()
    n5 = rebuilds()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        opt.advance()
        for i in range(len(opt.buffers)):
            opt.step_buffer(i)
    for _, g in opt.buffers:
        g.copy_(torch.randn_like(g))
    graph.replay()
    render()
    assert rebuilds() == n5 + 1, "an optimizer step inside a replayed graph must be seen"
    assert images_are_current()
    assert n4 >= 0


@pytest.mark.parametrize("mlp_dtype", ["bfloat16", "float16"])
def test_rendered_image_on_hand_written_pointwise_layers_equals_the_autocast_path(mlp_dtype, monkeypatch):
    """Round 6: the rendering entry runs the RGB decoder's three pointwise layers (1 x 1 head + ReLU, transposed 3 x 3 / stride 3,
    1 x 1 tail + sigmoid) on nr_pw_fwd like the training step does (Decoders.render_rgb16) instead of torch.autocast -> MIOpen
    (whose backward-data solver -- ConvTranspose2d's forward -- is slow once the overrunning NHWC one is switched off).  Same
    operands, same 16-bit rounding points: the image equals the autocast path's to a few units of the operand type."""
    H, W = 24, 42
    model = _model(chunk=500)
    model.field.config.mlp_dtype = mlp_dtype
    cam = lambda: _camera_rays(H, W, torch.Generator().manual_seed(9))  # noqa: E731
    monkeypatch.setenv("NR_RENDER_PW", "1")
    a = model.get_outputs_for_camera_ray_bundle(cam(), image_shape=(H, W))
    monkeypatch.setenv("NR_RENDER_PW", "0")
    b = model.get_outputs_for_camera_ray_bundle(cam(), image_shape=(H, W))
    assert a["rgb"].shape == b["rgb"].shape == (H, W, 3) and a["rgb"].dtype == torch.float32
    u = 2.0 ** -8 if mlp_dtype == "bfloat16" else 2.0 ** -11
    err = (a["rgb"] - b["rgb"]).abs()
    assert float(err.max()) <= 24 * u and float(err.mean()) <= 2 * u, (float(err.max()), float(err.mean()))
    assert float(a["rgb"].std()) > 1e-3, "a constant image would pass trivially"
    for k in ("features", "depth"):
        assert torch.equal(a[k], b[k])

