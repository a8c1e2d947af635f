"""CPU: pin the oracle (oracle/) against golden vectors produced by the reference itself
(tests/golden/make_golden.py).  These are the "not gpu" parity gate for the checker."""
import torch

from helpers import assert_close, load_golden, oracle_field_params, oracle_prop_params, t
from oracle import field, hashgrid, losses, pipeline, raygen, render, sampler


def test_level_scalings_match_reference_float32_quirk():
    g = load_golden("hash_encode")
    for key, args in {"scalings_static_main": (8, 32, 8192), "scalings_static_prop": (6, 128, 4096),
                      "scalings_actor": (4, 64, 1024), "scalings_class_default": (16, 16, 1024),
                      "scalings_nerfacto": (16, 16, 2048)}.items():
        assert torch.equal(hashgrid.level_scalings(*args), g[key]), key
    assert g["scalings_static_main"][-1].item() == 8191.0


def test_hash_slots_bit_exact():
    g = load_golden("hash_encode")
    c = g["hashfn_corners"]
    slots = hashgrid.hash_slots(c[..., 0], c[..., 1], c[..., 2], 2**14) + torch.arange(2) * 2**14
    assert torch.equal(slots, g["hashfn_slots"])


def test_hash_encode_forward_and_table_grad():
    g = load_golden("hash_encode")
    for tag in ("l8f4", "l6f1", "l16f2", "l4f4"):
        table = t(g[f"{tag}_table"], True)
        y = hashgrid.encode(g[f"{tag}_x"], table, g[f"{tag}_scalings"], 2 ** int(g[f"{tag}_log2t"]))
        assert_close(y, g[f"{tag}_out"], rtol=1e-6, atol_scale=1e-7, what=tag)
        (gt,) = torch.autograd.grad(y, table, g[f"{tag}_gout"])
        assert_close(gt, g[f"{tag}_gtable"], rtol=1e-5, atol_scale=1e-6, what=tag + " grad")


def test_gaussian_contraction_rescale():
    g = load_golden("gaussian_contraction")
    e = g["edges"]
    mean, std = field.isotropic_gaussian(g["origins"], g["directions"], e[:, :-1], e[:, 1:], g["pixel_area"])
    assert_close(mean, g["mean"], rtol=1e-6, atol_scale=1e-7)
    assert_close(std, g["std"], rtol=1e-6, atol_scale=1e-7)
    m01, s01 = field.scaled_contraction(mean, std, 100.0)
    assert_close(m01, g["mean01"], rtol=1e-6, atol_scale=1e-7)
    assert_close(s01, g["std01"], rtol=1e-6, atol_scale=1e-7)
    m01a, s01a = field.scaled_contraction(mean, std, 10.0)
    assert_close(m01a, g["mean01_actor"], rtol=1e-6, atol_scale=1e-7)
    assert_close(s01a, g["std01_actor"], rtol=1e-6, atol_scale=1e-7)
    grid = field.GridParams(g["table"], g["scalings"], int(g["log2t"]))
    feats = field.static_grid_features(mean, std, grid, 100.0)
    assert_close(feats, g["grid_features"], rtol=1e-5, atol_scale=1e-6)


def test_sh_and_mlp():
    g = load_golden("sh_mlp")
    assert_close(field.sh4(g["dirs"]), g["sh_raw"], rtol=1e-6, atol_scale=1e-7)
    assert_close(field.direction_encoding(g["dirs"]), g["sh_01"], rtol=1e-6, atol_scale=1e-7)
    layers = [(g[f"mlp_w{i}"], g[f"mlp_b{i}"]) for i in range(3)]
    assert_close(field.mlp(g["mlp_x"], layers), g["mlp_y"], rtol=1e-5, atol_scale=1e-6)
    # the lidar decoder (K7) is this MLP + split + sigmoid (neuradar.py:432-452)
    from oracle import decoders

    x = torch.as_tensor(g["mlp_x"])
    is_lidar = torch.zeros(x.shape[0], 1, dtype=torch.bool)
    is_lidar[::2] = True
    inten, drop = decoders.lidar_decode(x, is_lidar, [torch.as_tensor(w) for w, _ in layers], [torch.as_tensor(b) for _, b in layers])
    y = torch.as_tensor(g["mlp_y"])[::2]
    assert_close(inten, torch.sigmoid(y[:, :1]), rtol=1e-5, atol_scale=1e-6)
    assert_close(drop, y[:, 1:], rtol=1e-5, atol_scale=1e-6)
    assert decoders.lidar_decode(x, torch.zeros(x.shape[0], 1, dtype=torch.bool), [], []) == (None, None)


def _check_field(tag):
    g = load_golden(tag)
    p = oracle_field_params(g, requires_grad=True)
    e = g["edges"]
    feat, sdf, alpha = field.field_forward(p, g["origins"], g["directions"], e[:, :-1], e[:, 1:], g["pixel_area"])
    assert_close(feat, g["feature"], rtol=1e-5, atol_scale=1e-6, what="feature")
    assert_close(sdf, g["sdf"], rtol=1e-5, atol_scale=1e-6, what="sdf")
    assert_close(alpha, g["alpha"], rtol=1e-5, atol_scale=1e-6, what="alpha")
    loss = (feat * g["g_feature"]).sum() + (alpha * g["g_alpha"]).sum()
    grads = torch.autograd.grad(loss, p.tensors())
    names = ["table"]
    for grp, lst in (("geo", p.geo), ("feat", p.feat)):
        for i in range(len(lst)):
            names += [f"{grp}_w{i}", f"{grp}_b{i}"]
    names.append("beta")
    for n, gr in zip(names, grads):
        assert_close(gr, g["grad_" + n], rtol=1e-4, atol_scale=1e-5, what="grad " + n)
    return g


def test_field_neurad_dims_fwd_bwd():
    g = _check_field("field_neurad")
    pp = oracle_prop_params(g, requires_grad=True)
    e = g["edges"]
    dens = field.proposal_density(pp, g["origins"], g["directions"], e[:, :-1], e[:, 1:], g["pixel_area"])
    assert_close(dens, g["prop_density"], rtol=1e-5, atol_scale=1e-6)
    gt, gw = torch.autograd.grad((dens * g["prop_g_density"]).sum(), pp.tensors())
    assert_close(gt, g["prop_grad_table"], rtol=1e-4, atol_scale=1e-5)
    assert_close(gw, g["prop_grad_decoder"], rtol=1e-4, atol_scale=1e-5)


def test_field_l16f2_w64_fwd_bwd():
    _check_field("field_l16f2w64")


def test_power_bins_weights_pdf():
    g = load_golden("sampler")
    s_eval = sampler.power_bins(g["nears"], g["fars"], 128)
    assert_close(s_eval.spacing, g["power_eval_spacing"], rtol=1e-6, atol_scale=1e-7)
    assert_close(s_eval.euclid, g["power_eval_euclid"], rtol=1e-5, atol_scale=1e-7)
    s_train = sampler.power_bins(g["nears"], g["fars"], 128, g["power_train_t_rand"])
    assert_close(s_train.spacing, g["power_train_spacing"], rtol=1e-6, atol_scale=1e-7)
    assert_close(s_train.euclid, g["power_train_euclid"], rtol=1e-5, atol_scale=1e-7)
    assert_close(s_train.deltas, g["gw_deltas"], rtol=1e-5, atol_scale=1e-7)
    w = sampler.weights_from_density(g["gw_deltas"], g["gw_density"])
    assert_close(w, g["gw_weights"], rtol=1e-5, atol_scale=1e-7)
    r_eval = sampler.pdf_resample(s_train, g["gw_weights"], 64)
    assert_close(r_eval.spacing, g["pdf_eval_spacing"], rtol=1e-5, atol_scale=1e-6)
    assert_close(r_eval.euclid, g["pdf_eval_euclid"], rtol=1e-4, atol_scale=1e-6)
    r_train = sampler.pdf_resample(s_train, g["gw_weights"], 64, g["pdf_train_jitter"])
    assert_close(r_train.spacing, g["pdf_train_spacing"], rtol=1e-5, atol_scale=1e-6)
    assert_close(r_train.euclid, g["pdf_train_euclid"], rtol=1e-4, atol_scale=1e-6)


def test_pipeline_end_to_end_and_bench_loss_grads():
    g = load_golden("pipeline")
    fp = oracle_field_params(g, prefix="main_", log2t_key="main_log2t", requires_grad=True)
    pp = oracle_prop_params(g, prefix="prop1_", log2t_key="prop_log2t", requires_grad=True)
    bundle = {k: g[k] for k in ("origins", "directions", "pixel_area", "fars")}
    out = pipeline.nff_outputs(fp, [pp, pp], bundle, g["t_rand"], (g["jitter1"], g["jitter2"]))
    for i in (0, 1):
        assert_close(out[f"prop_spacing_{i}"], g[f"prop_spacing_{i}"], rtol=1e-5, atol_scale=1e-6)
        assert_close(out[f"prop_euclid_{i}"], g[f"prop_euclid_{i}"], rtol=1e-4, atol_scale=1e-6)
        assert_close(out[f"prop_weights_{i}"], g[f"prop_weights_{i}"], rtol=1e-4, atol_scale=1e-5)
        assert_close(out[f"prop_depth_{i}"], g[f"prop_depth_{i}"], rtol=1e-4, atol_scale=1e-5)
    assert_close(out["final_spacing"], g["final_spacing"], rtol=1e-5, atol_scale=1e-6)
    assert_close(out["final_euclid"], g["final_euclid"], rtol=1e-4, atol_scale=1e-6)
    assert_close(out["alpha"], g["alpha"], rtol=1e-4, atol_scale=1e-5)
    assert_close(out["weights"], g["weights"], rtol=1e-4, atol_scale=1e-5)
    assert_close(out["accumulation"], g["accumulation"], rtol=1e-4, atol_scale=1e-5)
    assert_close(out["features"], g["features"], rtol=1e-4, atol_scale=1e-5)
    assert_close(out["depth"], g["depth"], rtol=1e-4, atol_scale=1e-5)
    # in-repo cross-check of the unpinned nerfacc formula: same up to the +1e-7 per factor
    torch.testing.assert_close(render.render_weight_from_alpha(g["alpha"][..., 0])[0], g["weights_alt_inrepo"],
                               rtol=1e-4, atol=1e-5)
    inter = losses.zipnerf_interlevel_loss(out["c_list"], out["w_list"])
    dist = losses.distortion_loss(out["c_list"][-1], out["w_list"][-1])
    assert_close(inter, g["interlevel"], rtol=1e-4, atol_scale=1e-6)
    assert_close(dist, g["distortion"], rtol=1e-4, atol_scale=1e-6)
    loss = pipeline.train_loss(out, g["target_features"], g["target_depth"])
    assert_close(loss, g["loss"], rtol=1e-5, atol_scale=1e-6)
    main_t, prop_t = fp.tensors(), pp.tensors()
    grads = torch.autograd.grad(loss, main_t + prop_t)
    names = ["main_hashgrid_static_grid_hash_table"]
    for grp, n in (("mlp_geo", len(fp.geo)), ("mlp_feature", len(fp.feat))):
        for i in range(n):
            names += [f"main_{grp}_layers_{i}_weight", f"main_{grp}_layers_{i}_bias"]
    names += ["main_sdf_to_density_beta", "prop1_hashgrid_static_grid_hash_table", "prop1_density_decoder_weight"]
    for n, gr in zip(names, grads):
        assert_close(gr, g["grad_" + n], rtol=2e-4, atol_scale=2e-5, what=n)


def test_regularisers_on_handmade_histograms():
    g = load_golden("losses")
    ws = [t(g[f"w{i}"], True) for i in range(3)]
    cs = [g[f"c{i}"] for i in range(3)]
    inter = losses.zipnerf_interlevel_loss(cs, ws)
    dist = losses.distortion_loss(cs[-1], ws[-1])
    assert_close(inter, g["interlevel"], rtol=1e-5, atol_scale=1e-7)
    assert_close(dist, g["distortion"], rtol=1e-5, atol_scale=1e-7)
    gi = torch.autograd.grad(inter, ws[:2])
    (gd,) = torch.autograd.grad(dist, ws[2:])
    assert_close(gi[0], g["g_inter_w0"], rtol=1e-4, atol_scale=1e-6)
    assert_close(gi[1], g["g_inter_w1"], rtol=1e-4, atol_scale=1e-6)
    assert_close(gd, g["g_dist_w2"], rtol=1e-4, atol_scale=1e-6)


def test_raygen_camera_lidar_radar():
    g = load_golden("raygen")
    cam = raygen.camera_rays(g["cam_ray_indices"], g["cam_c2w"], g["cam_fx"], g["cam_fy"], g["cam_cx"], g["cam_cy"],
                             g["cam_times_in"], g["cam_vel"], g["cam_rs_offsets"], g["cam_heights"])
    for k in ("origins", "directions", "pixel_area", "times", "fars", "directions_norm"):
        assert_close(cam[k], g["cam_" + k], rtol=1e-5, atol_scale=1e-6, what="cam " + k)
    cam2 = raygen.camera_rays(g["cam_ray_indices"], g["cam_c2w"], g["cam_fx"], g["cam_fy"], g["cam_cx"], g["cam_cy"],
                              g["cam_times_in"])
    assert_close(cam2["origins"], g["cam_nors_origins"], rtol=1e-6, atol_scale=1e-7)
    assert_close(cam2["times"], g["cam_nors_times"], rtol=1e-6, atol_scale=1e-7)
    cam_args = (g["cam_ray_indices"], g["cam_c2w"], g["cam_fx"], g["cam_fy"], g["cam_cx"], g["cam_cy"], g["cam_times_in"])
    fe = raygen.camera_rays(*cam_args, distortion=g["cam_dist"], fisheye=torch.ones(5, dtype=torch.bool))  # ZOD model
    assert_close(fe["directions"], g["cam_fe_directions"], rtol=1e-6, atol_scale=1e-7, what="fisheye directions")
    assert_close(fe["pixel_area"], g["cam_fe_pixel_area"], rtol=1e-5, atol_scale=1e-7, what="fisheye pixel_area")
    pd = raygen.camera_rays(*cam_args, distortion=g["cam_dist"])
    assert_close(pd["directions"], g["cam_pd_directions"], rtol=1e-6, atol_scale=1e-7, what="undistorted pinhole directions")
    lid = raygen.lidar_rays(g["lid_indices"], g["lid_points"], g["lid_l2w"], g["lid_times_in"], g["lid_vel"])
    for k in ("origins", "directions", "pixel_area", "times", "fars", "directions_norm"):
        assert_close(lid[k], g["lid_" + k], rtol=1e-5, atol_scale=1e-6, what="lidar " + k)
    assert torch.equal(lid["did_return"], g["lid_did_return"]) and torch.equal(lid["is_lidar"], g["lid_is_lidar"])
    fov = [float(v) for v in g["rad_zod_fov"]]
    rad = raygen.radar_rays(g["rad_scans"], g["rad_r2w"], g["rad_times_in"][:, None], *fov)
    assert rad["directions"].shape[0] == 2 * 3531  # ZOD: 107 x 33 rays per scan
    for k in ("origins", "directions", "pixel_area", "times", "fars", "directions_spher", "directions_norm"):
        assert_close(rad[k], g["rad_" + k], rtol=1e-5, atol_scale=1e-6, what="radar " + k)
    assert torch.equal(rad["scan_of_ray"], g["rad_scan_of_ray"])
    for tag, n in (("default", 256), ("vod", 4545)):
        fov = [float(v) for v in g[f"rad_{tag}_fov"]]
        r = raygen.radar_rays(torch.tensor([1]), g["rad_r2w"], g["rad_times_in"][:, None], *fov)
        assert r["directions"].shape[0] == n, (tag, r["directions"].shape)
        assert_close(r["directions"], g[f"rad_{tag}_directions"], rtol=1e-5, atol_scale=1e-6)
        assert_close(r["directions_spher"], g[f"rad_{tag}_directions_spher"], rtol=1e-6, atol_scale=1e-7)
        assert_close(r["pixel_area"], g[f"rad_{tag}_pixel_area"], rtol=1e-6, atol_scale=1e-7)


def _actor_setup(g, requires_grad=False):
    from oracle import actors as oactors
    from oracle.field import GridParams

    state = oactors.ActorState(t(g["actor_positions"], requires_grad), t(g["actor_rotations_6d"], requires_grad),
                               g["actor_timestamps"], g["actor_present"], g["actor_sizes"], g["actor_padding"])
    fp = oracle_field_params(g, requires_grad=requires_grad)
    fp.actor_grids = [GridParams(t(g[f"actor{i}_table"], requires_grad), g["actor_scalings"], int(g["actor_log2t"])) for i in range(2)]
    pp = oracle_prop_params(g, requires_grad=requires_grad)
    pp.actor_grids = [GridParams(g[f"prop_actor{i}_table"], g["prop_actor_scalings"], int(g["actor_log2t"])) for i in range(2)]
    return state, fp, pp


def test_dynamic_actors_eval_and_train_with_trajectory_grads():
    """a10: culling, world->box transform, per-ray flip, per-actor grids, pose gradients."""
    from oracle import actors as oactors

    g = load_golden("actors")
    e = g["edges"]
    args = (g["origins"], g["directions"], e[:, :-1], e[:, 1:], g["pixel_area"])
    state, fp, pp = _actor_setup(g)
    mean, _ = field.isotropic_gaussian(*args)
    b2w, valid = oactors.boxes2world(state, g["times"][:, 0])
    ri, si, ai = oactors.actor_indices(mean, b2w, valid, oactors.pose_inverse(b2w), state.bounds())
    assert torch.equal(ri, g["actor_ray_idx"]) and torch.equal(si, g["actor_sample_idx"]) and torch.equal(ai, g["actor_actor_idx"])
    ctx = {"state": state, "times": g["times"][:, 0], "flip": None}
    feat, sdf, alpha = field.field_forward(fp, *args, actor_ctx=ctx)
    assert_close(feat, g["eval_feature"], rtol=1e-5, atol_scale=1e-6, what="eval feature")
    assert_close(alpha, g["eval_alpha"], rtol=1e-5, atol_scale=1e-6, what="eval alpha")
    dens = field.proposal_density(pp, *args, actor_ctx={**ctx, "require_grad": False})
    assert_close(dens, g["eval_prop_density"], rtol=1e-5, atol_scale=1e-6, what="eval prop density")
    # training: per-ray flip + gradients incl. the trajectory parameters
    state, fp, pp = _actor_setup(g, requires_grad=True)
    ctx = {"state": state, "times": g["times"][:, 0], "flip": g["flip"]}
    feat, sdf, alpha = field.field_forward(fp, *args, actor_ctx=ctx)
    assert_close(feat, g["train_feature"], rtol=1e-5, atol_scale=1e-6, what="train feature")
    assert_close(alpha, g["train_alpha"], rtol=1e-5, atol_scale=1e-6, what="train alpha")
    loss = (feat * g["g_feature"]).sum() + (alpha * g["g_alpha"]).sum()
    wrt = [fp.grid.table, fp.actor_grids[0].table, fp.actor_grids[1].table, state.positions, state.rotations_6d, fp.geo[0][0]]
    keys = ["hashgrid_static_grid_hash_table", "hashgrid_actor_grids_0_hash_table", "hashgrid_actor_grids_1_hash_table",
            "hashgrid_actors_actor_positions", "hashgrid_actors_actor_rotations_6d", "mlp_geo_layers_0_weight"]
    for k, gr in zip(keys, torch.autograd.grad(loss, wrt)):
        assert_close(gr, g["grad_" + k], rtol=1e-4, atol_scale=1e-5, what="grad " + k)


# ---- the reference's own tests that pin values on the path, restated on the oracle ----
def test_reference_frustum_midpoint_vector():
    """tests/cameras/test_rays.py:12-34: origin (0,1,2), direction (0,1,0), start 2, end 3 -> position (0,3.5,2)."""
    from oracle import field as ofield

    mean, _ = ofield.isotropic_gaussian(torch.tensor([[0.0, 1.0, 2.0]]), torch.tensor([[0.0, 1.0, 0.0]]), torch.tensor([[2.0]]),
                                        torch.tensor([[3.0]]), torch.ones(1, 1))
    assert torch.allclose(mean[0, 0], torch.tensor([0.0, 3.5, 2.0]), atol=1e-6)


def test_reference_spherical_harmonics_orthonormality():
    """tests/utils/test_math.py:7-16 (degree 4): (sh^T sh)/N * 4 pi == I, atol 1.5e-2."""
    from oracle import field as ofield

    torch.manual_seed(0)
    n = 1_000_000
    d = torch.nn.functional.normalize(torch.normal(0, 1, size=(n, 3)), dim=-1)
    sh = ofield.sh4(d)
    torch.testing.assert_close((sh.T @ sh) / n * 4 * torch.pi, torch.eye(16), rtol=0, atol=1.5e-2)


def test_reference_pinhole_camera_origin():
    """tests/cameras/test_cameras.py:109-121: identity pose, cx=cy=400, fx=fy=10 -> ray origins at 0."""
    from oracle import raygen

    idx = torch.tensor([[0, 0, 0], [0, 400, 400], [0, 799, 13]])
    out = raygen.camera_rays(idx, torch.eye(4)[None, :3, :], torch.tensor([10.0]), torch.tensor([10.0]), torch.tensor([400.0]),
                             torch.tensor([400.0]), torch.zeros(1))
    assert torch.allclose(out["origins"], torch.zeros(3, 3))
    assert torch.allclose(out["directions"].norm(dim=-1), torch.ones(3), atol=1e-6)


def test_field_density_branch_vs_reference_golden():
    """a16, use_sdf=False: DENSITY = trunc_exp(geo_out) (fields/neurad_field.py:149-150), forward and parameter gradients."""
    from oracle import field as of

    g = load_golden("field_density")
    req = lambda k: g[k].clone().requires_grad_(True)  # noqa: E731
    p = of.FieldParams(of.GridParams(req("table"), g["scalings"], int(g["log2t"])), [(req(f"geo_w{i}"), req(f"geo_b{i}")) for i in range(2)],
                       [(req(f"feat_w{i}"), req(f"feat_b{i}")) for i in range(3)], torch.ones(1))
    e = g["edges"]
    feature, density = of.field_forward(p, g["origins"], g["directions"], e[:, :-1], e[:, 1:], g["pixel_area"], use_sdf=False)
    torch.testing.assert_close(feature, g["feature"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(density, g["density"], rtol=1e-5, atol=1e-6)
    loss = (feature * g["g_feature"]).sum() + (density * g["g_density"]).sum()
    grads = torch.autograd.grad(loss, [p.grid.table, p.geo[0][0], p.geo[1][0], p.geo[1][1], p.feat[0][0]])
    for got, k in zip(grads, ("grad_table", "grad_geo_w0", "grad_geo_w1", "grad_geo_b1", "grad_feat_w0")):
        torch.testing.assert_close(got, g[k], rtol=1e-4, atol=1e-5 * float(g[k].abs().max()))
