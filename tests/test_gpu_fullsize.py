"""Full-size parity: the HIP path against the CPU oracle on BASELINE.json's real configurations (tables, widths,
sample counts), not only on the reference-generated toy fixtures -- 4 096 camera rays of configs[1] (main grid
L16/F2/T=2^19, 64-wide MLPs, proposal grids L6/F1/T=2^20) and a 2 048-ray camera + radar + lidar slice of the
configs[2] batch on NeuRadar's own field (L8/F4/T=2^22, 32-wide).  The oracle takes a few seconds per case on CPU."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _oracle_params(model):
    from oracle.field import FieldParams, GridParams, ProposalParams

    c = lambda t: t.detach().cpu().clone().requires_grad_(True)  # noqa: E731
    f = model.field
    sg = f.hashgrid.static_grid
    fp = FieldParams(GridParams(c(sg.hash_table), sg.scalings.cpu(), sg.log2_hashmap_size),
                     [(c(l.weight), c(l.bias)) for l in f.mlp_geo.layers], [(c(l.weight), c(l.bias)) for l in f.mlp_feature.layers],
                     c(f.sdf_to_density.beta), f.hashgrid.static_scale)
    p = model.proposal_fields[1]
    pg = p.hashgrid.static_grid
    pp = ProposalParams(GridParams(c(pg.hash_table), pg.scalings.cpu(), pg.log2_hashmap_size), c(p.density_decoder.weight),
                        p.hashgrid.static_scale)
    return fp, pp


def _check(got, want, what, rtol=1e-4, floor=1e-6, few=1e-3):
    """Element-wise: |got - want| <= rtol * |want| + floor * max|want| (the north star's "within 1e-4 rel"; the floor --
    one millionth of the tensor's scale -- stands in for fp32 cancellation in sums whose result is ~0).  At most a
    fraction `few` of the elements may exceed that, and those stay within rtol of the tensor's SCALE: a sample whose sdf
    sits on the steep part of sigmoid(-20 * sdf) turns a 1e-6 difference in summation order into 2e-5 of alpha (measured:
    1 of 65 536 weights of the mixed batch, 1.5e-5 at scale 0.76; 16 of 65 536 rendered features, 5.3e-6 at scale 0.35)."""
    want = want.to(got.dtype).reshape(got.shape)
    scale = float(want.abs().max())
    err = (got - want).abs()
    bad = err > rtol * want.abs() + floor * scale
    frac = float(bad.float().mean())
    assert frac <= few and float(err.max()) <= rtol * scale, (
        f"{what}: {int(bad.sum())} of {bad.numel()} elements outside rtol {rtol} + {floor}*scale; worst |d| = {float(err.max()):.3e} "
        f"at scale {scale:.3e}")


@pytest.mark.parametrize("workload,n_rays", [("cam4096_l16f2_w64", 4096), ("mixed16384_neuradar", 2048)])
def test_full_size_step_vs_oracle(workload, n_rays):
    import bench
    from neuradar_amd.batch_assembly import SensorBatchAssembler
    from neuradar_amd.rays import RayBundle
    from oracle import pipeline as op

    wl = bench.WORKLOADS[workload]
    model = bench.build_model(wl, torch.device(DEV)).train()
    with torch.no_grad():  # trained-looking tables: O(0.1) features instead of the 1e-3 initialisation, a density field with structure
        model.field.hashgrid.static_grid.hash_table.mul_(200.0)
        model.proposal_fields[1].hashgrid.static_grid.hash_table.mul_(1000.0)
    scene = bench.SyntheticScene(torch.device(DEV), seed=1000)
    gen = torch.Generator().manual_seed(7)
    if "cam_rays" in wl:  # 1 patch + a slice of a radar scan + lidar rays
        asm = SensorBatchAssembler(scene.cameras, scene.H, scene.W, scene.PATCH, scene.STRIDE, 1, lidars=scene.lidars,
                                   lidar_points=scene.lidar_points, points_per_lidar=scene.points_per_lidar, n_lidar_rays=600,
                                   radars=scene.radars, n_radar_scans=1, order=("camera", "radar", "lidar"))
        s = asm.assemble(torch.rand(asm.uniform_count(), generator=gen).to(DEV))
        keep = torch.cat([torch.arange(0, 1024), torch.arange(1024, 1024 + 424), torch.arange(asm.offset["lidar"], asm.offset["lidar"] + 600)]).to(DEV)
        o, d, area = s["origins"][keep], s["directions"][keep], s["pixel_area"][keep]
    else:
        asm = SensorBatchAssembler(scene.cameras, scene.H, scene.W, scene.PATCH, scene.STRIDE, n_rays // 1024)
        s = asm.assemble(torch.rand(asm.uniform_count(), generator=gen).to(DEV))
        o, d, area = s["origins"], s["directions"], s["pixel_area"]
    assert o.shape[0] == n_rays
    fars = torch.full((n_rays, 1), 1e6, device=DEV)
    t_rand = torch.rand(n_rays, 129, generator=gen)
    j1, j2 = torch.rand(n_rays, 1, generator=gen), torch.rand(n_rays, 1, generator=gen)
    tf, td = 0.1 * torch.randn(n_rays, 32, generator=gen), 5.0 + 50.0 * torch.rand(n_rays, 1, generator=gen)
    out = model.get_nff_outputs(RayBundle(o.clone(), d.clone(), area[:, None].clone(), fars=fars), t_rand=t_rand.to(DEV),
                                jitters=(j1.to(DEV), j2.to(DEV)))
    loss = model.bench_loss(out, tf.to(DEV), td.to(DEV))
    fp, pp = _oracle_params(model)
    ref = op.nff_outputs(fp, [pp, pp], {"origins": o.cpu(), "directions": d.cpu(), "pixel_area": area[:, None].cpu(), "fars": fars.cpu()},
                         t_rand, (j1, j2))
    ref_loss = op.train_loss(ref, tf, td)
    cpu = lambda t: t.detach().cpu()  # noqa: E731
    _check(cpu(out["ray_samples"].spacing), ref["final_spacing"].detach(), "final spacing")
    _check(cpu(out["weights"][..., 0]), ref["weights"].detach(), "weights")
    _check(cpu(out["accumulation"]), ref["accumulation"].detach(), "accumulation")
    _check(cpu(out["features"]), ref["features"].detach(), "features")
    _check(cpu(out["depth"]), ref["depth"].detach(), "depth", floor=1e-5)  # sum of w * t with t up to 2e4 m: a coarser floor
    for i in (0, 1):
        _check(cpu(out["weights_list"][i][..., 0]), ref[f"prop_weights_{i}"].detach(), f"proposal weights {i}", rtol=1e-3, floor=1e-5)
    assert abs(float(loss) - float(ref_loss)) <= 1e-4 * abs(float(ref_loss)), (float(loss), float(ref_loss))
    # every parameter gradient of the chain, as relative L2 errors (tables: the rows the step touched)
    names = {"main table": (model.field.hashgrid.static_grid.hash_table, fp.grid.table),
             "proposal table": (model.proposal_fields[1].hashgrid.static_grid.hash_table, pp.grid.table),
             "proposal decoder": (model.proposal_fields[1].density_decoder.weight, pp.decoder),
             "beta": (model.field.sdf_to_density.beta, fp.beta)}
    for i, l in enumerate(model.field.mlp_geo.layers):
        names[f"geo w{i}"] = (l.weight, fp.geo[i][0])
    for i, l in enumerate(model.field.mlp_feature.layers):
        names[f"feat w{i}"] = (l.weight, fp.feat[i][0])
    g_hip = torch.autograd.grad(loss, [a for a, _ in names.values()])
    g_ref = torch.autograd.grad(ref_loss, [b for _, b in names.values()])
    for k, gh, gr in zip(names, g_hip, g_ref):
        err = float((cpu(gh) - gr).norm() / gr.norm().clamp_min(1e-30))
        assert err < 2e-3, f"grad {k}: relative L2 error {err:.3e}"
        if "table" in k:  # the same rows are touched: where only one side is non-zero the value is a rounding residue
            a, b = cpu(gh), gr
            diff = (a != 0) != (b != 0)
            resid = float(torch.maximum(a.abs(), b.abs())[diff].max()) if bool(diff.any()) else 0.0
            assert resid <= 1e-6 * float(b.abs().max()), f"grad {k}: rows touched on one side only carry up to {resid:.3e}"
    if "cam_rays" not in wl:
        return
    # ---- the path bench.py TIMES, pinned directly: the fused step with bf16 MFMA operands on the same slice against the
    #      oracle -- rendered outputs within 5 u of their scale (u = 2^-8, bf16's unit roundoff: the operands of five chained
    #      layers are rounded, accumulation is fp32), the loss within 1e-3 relative
    from neuradar_amd.fused_step import FusedTrainStep

    model16 = bench.build_model(wl, torch.device(DEV), "bfloat16").train()
    model16.load_state_dict(model.state_dict())
    fused = FusedTrainStep(model16, n_rays, coherent_rays=1024 + 424)  # camera + radar rows sample-major, lidar rows ray-major
    floss = fused.forward_backward(o.contiguous(), d.contiguous(), area.contiguous(), fars[:, 0].contiguous(), tf.to(DEV), td[:, 0].to(DEV),
                                   t_rand.to(DEV), j1[:, 0].to(DEV), j2[:, 0].to(DEV))
    fo = fused.outputs()
    u = 2.0 ** -8
    for key, want in (("features", ref["features"]), ("depth", ref["depth"]), ("accumulation", ref["accumulation"]),
                      ("weights", ref["weights"]), ("prop_weights_0", ref["prop_weights_0"]), ("prop_weights_1", ref["prop_weights_1"])):
        got, want = cpu(fo[key]).reshape(-1), want.detach().reshape(-1)
        scale = float(want.abs().max())
        worst = float((got - want).abs().max())
        assert worst <= 5 * u * scale, f"fused bf16 {key}: worst |d| = {worst:.3e} at scale {scale:.3e} (> 5u)"
    assert abs(float(floss.sum()) - float(ref_loss)) <= 1e-3 * abs(float(ref_loss)), (float(floss.sum()), float(ref_loss))
