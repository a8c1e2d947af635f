"""GPU parity of the device-side batch assembly against the oracle's restatement of the reference's
samplers and merge (oracle/batch.py) fed the same random numbers: sampled indices bit-exact, rays to fp32 rounding."""
import pytest
import torch

from helpers import assert_close, load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rig(g):
    from neuradar_amd.sensors import Cameras, Lidars, Radars

    d = lambda k: g[k].to(DEV)  # noqa: E731
    cams = Cameras(d("cam_c2w"), d("cam_fx"), d("cam_fy"), d("cam_cx"), d("cam_cy"), d("cam_heights"), d("cam_times_in"),
                   d("cam_vel"), d("cam_rs_offsets"))
    n_l = g["lid_l2w"].shape[0]
    lid = Lidars(d("lid_l2w"), d("lid_times_in"), d("lid_vel"))
    n_r = g["rad_r2w"].shape[0]
    rad = Radars(d("rad_r2w"), d("rad_times_in"), radar_azimuth_ray_divergence=0.015, radar_elevation_ray_divergence=0.015,
                 min_azimuth=-0.80, max_azimuth=0.80, min_elevation=-0.08, max_elevation=0.4)
    return cams, lid, n_l, rad, n_r


@pytest.mark.parametrize("order", [("camera", "lidar", "radar"), ("camera", "radar", "lidar")])
def test_assembler_vs_oracle_samplers_and_merge(order):
    from neuradar_amd.batch_assembly import SensorBatchAssembler
    from oracle import batch as ob, raygen as org

    g = load_golden("raygen")
    cams, lid, n_l, rad, n_r = _rig(g)
    gen = torch.Generator().manual_seed(3)
    ppl = torch.randint(50, 400, (n_l,), generator=gen)
    total = int(ppl.sum())
    rng = 2.0 + 100.0 * torch.rand(total, generator=gen)
    rng[::11] = 2000.0  # non-returns (beyond the 1e3 m threshold, lidars.py:404)
    dirs = torch.nn.functional.normalize(torch.randn(total, 3, generator=gen), dim=-1)
    pts = torch.cat([dirs * rng[:, None], torch.rand(total, 1, generator=gen), 0.1 * torch.rand(total, 1, generator=gen)], dim=1)
    n_patches, n_lidar, n_scans = 2, 333, 3
    asm = SensorBatchAssembler(cams, 1080, 1920, 32, 3, n_patches, lidars=lid, lidar_points=pts.to(DEV), points_per_lidar=ppl,
                               n_lidar_rays=n_lidar, radars=rad, n_radar_scans=n_scans, order=order)
    assert asm.n == n_patches * 1024 + n_lidar + n_scans * asm.per_scan
    u = torch.rand(asm.uniform_count(), generator=gen)
    s = asm.assemble(u.to(DEV), slot=1)
    torch.cuda.synchronize()
    # ---- lidar: indices bit-exact against the reference's sampler on the same numbers
    n_u = 3 * n_patches
    u_l = u[n_u:n_u + n_lidar]
    shuffle = torch.argsort(u[n_u + n_lidar:n_u + n_lidar + n_l], stable=True)
    assert torch.equal(asm.lidar_order.cpu(), shuffle), "device permutation != stable argsort of the same uniforms"
    rpl = asm.rays_per_lidar
    # ray i of the device batch is (slot i // rpl, column i % rpl) of the reference's [num_lidars, rays_per_lidar] draw
    u_mat = torch.zeros(n_l, rpl, dtype=torch.float64)
    i = torch.arange(n_lidar)
    u_mat[shuffle[i // rpl], i % rpl] = u_l.double()
    idx, flat = ob.lidar_point_sample(u_mat, shuffle, ppl, n_lidar)
    assert torch.equal(s["lidar_indices"][:n_lidar].cpu(), idx)
    ref_l = org.lidar_rays(idx[:, 0], pts[flat], g["lid_l2w"], g["lid_times_in"], g["lid_vel"])
    # ---- radar scans
    scans = ob.radar_scan_choice(u[n_u + n_lidar + n_l:], n_scans, n_r)
    assert torch.equal(s["scan_indices"][:n_scans].cpu(), scans)
    ref_r = org.radar_rays(scans, g["rad_r2w"], g["rad_times_in"], -0.80, 0.80, 0.015, -0.08, 0.4, 0.015)
    # ---- camera patches: the index path of the same library (pinned against the reference in test_gpu_parity)
    b_c, idx_c = cams.generate_patch_rays(u[:n_u].view(n_patches, 3).to(DEV), 32, 3, 1080, 1920, area_scale=9.0, return_indices=True)
    cam = {"origins": b_c.origins.cpu(), "directions": b_c.directions.cpu(), "pixel_area": b_c.pixel_area.cpu(), "times": b_c.times.cpu()}
    want = ob.merge_img_lidar_radar({"camera": cam, "lidar": ref_l, "radar": ref_r}, order)
    got = asm.bundle(1)
    assert_close(got.origins.cpu(), want["origins"], rtol=1e-6, atol_scale=1e-6, what="origins")
    assert_close(got.directions.cpu(), want["directions"], rtol=1e-4, atol_scale=1e-5, what="directions")  # fp32 normalisation of far (2 km) points
    assert_close(got.pixel_area.cpu(), want["pixel_area"], rtol=1e-5, atol_scale=1e-7, what="pixel_area")
    assert_close(got.times.cpu(), want["times"], rtol=1e-6, atol_scale=1e-6, what="times")
    for k in ("is_lidar", "is_radar", "did_return"):
        assert torch.equal(got.metadata[k].cpu(), want[k].bool()), k
    assert_close(got.metadata["directions_spher"].cpu(), want["directions_spher"], rtol=1e-6, atol_scale=1e-6, what="spher")
    lid_seg = asm.seg("lidar")
    assert_close(got.metadata["directions_norm"][lid_seg].cpu(), want["directions_norm"][lid_seg], rtol=1e-5,
                 atol_scale=1e-6, what="lidar distance")
    assert float(got.fars.min()) == 1e6


def test_radar_scan_choice_edge_cases():
    from neuradar_amd import ops

    lib, p, st = ops._lib.lib(), ops._p, ops._stream
    out = torch.full((4,), -1, device=DEV, dtype=torch.int64)
    ops.check(lib.nr_sample_radar_scans(None, 4, 3, p(out), st()), "scan")  # fewer radars than scans: 0,1,2 then padded with 0
    assert out.tolist() == [0, 1, 2, 0]
    u = torch.tensor([0.0, 0.999999, 0.5, 0.25], device=DEV)
    ops.check(lib.nr_sample_radar_scans(p(u), 4, 10, p(out), st()), "scan")  # randint(0, 9): the last scan (9) is never drawn
    assert out.tolist() == [0, 8, 4, 2]
