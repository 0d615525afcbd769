"""cer_geo_consistency_f32 / cer-mvs_amd/fusion.py on the MI355X against captures of the reference's fusion.py
(tests/golden/fusion.npz) and the CPU oracle.  Thresholded outputs are compared as mismatch fractions: a pixel whose
reprojection error sits within an ulp of a threshold may legitimately fall on the other side (different fp32 evaluation
order of the 3x3 products on the device); everything continuous is compared in relative L1."""
import numpy as np
import pytest
import torch

from conftest import rel_l1

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


def _scene(g):
    from cer_mvs_amd.synthetic import synthetic_depth_maps, synthetic_scene, tensor_checksum
    H, W, V = int(g["H"]), int(g["W"]), int(g["V"])
    images, poses, intr, _ = synthetic_scene(H, W, V, seed=int(g["scene_seed"]))
    depths = synthetic_depth_maps(H, W, V, seed=int(g["scene_seed"]))
    assert tensor_checksum(depths) == int(g["depths_checksum"])
    return images, depths, intr[0], poses[0], V


def test_check_geometric_consistency_matches_reference_capture(dev, golden):
    from cer_mvs_amd import fusion
    g = golden("fusion")
    _, depths, K, E, V = _scene(g)
    S = V
    masks, mask, drep, xs, ys, rel = fusion.check_geometric_consistency(
        depths[0][None].repeat(S, 1, 1).to(dev), K[0][None].repeat(S, 1, 1), E[0][None].repeat(S, 1, 1), depths[1:].to(dev), K[1:], E[1:],
        4.0, 1300.0)
    got = torch.stack(masks).cpu().numpy()
    assert got.shape == g["cgc_masks"].shape
    assert (got != g["cgc_masks"]).mean() < 2e-4
    assert rel_l1(xs.cpu(), g["cgc_x_src"]) < 1e-6 and rel_l1(ys.cpu(), g["cgc_y_src"]) < 1e-6
    same = torch.from_numpy(got[-1] == g["cgc_masks"][-1])
    assert rel_l1(drep.cpu()[same], torch.from_numpy(g["cgc_depth_reprojected"])[same]) < 1e-6
    r, rr = rel.cpu(), torch.from_numpy(g["cgc_rel"])
    assert (r - rr).abs().max() < 1e-5


def test_fused_vote_equals_per_view_api_and_oracle(dev, golden):
    """The fused launch (what the loop uses) against the aggregation of the per-view API outputs (fusion.py:226-236) on
    the device, and against the CPU oracle."""
    from cer_mvs_amd import fusion
    from oracle import fusion_oracle as FO
    g = golden("fusion")
    _, depths, K, E, V = _scene(g)
    for ref in (0, 3):
        src = [j for j in range(V + 1) if j != ref]
        S, n = len(src), len(src) + 1
        cnt = torch.zeros(fusion.COUNTERS, device=dev, dtype=torch.int32)
        geo, est = fusion.vote(depths[ref].to(dev), K[ref], E[ref], depths[src].to(dev), K[src], E[src], 9.0, 2900.0, count=cnt)
        masks, mask, drep, _, _, _ = fusion.check_geometric_consistency(
            depths[ref][None].repeat(S, 1, 1).to(dev), K[ref][None].repeat(S, 1, 1), E[ref][None].repeat(S, 1, 1), depths[src].to(dev),
            K[src], E[src], 9.0, 2900.0)
        gsum = mask.sum(0)
        lit = gsum >= n
        for i in range(2, n):
            lit = lit | (masks[i - 2].sum(0) >= i)
        assert torch.equal(geo.bool(), lit)
        assert int(cnt.sum().item()) == int(lit.sum().item())
        assert rel_l1(est.cpu(), ((drep.sum(0) + depths[ref].to(dev)) / (gsum + 1)).cpu()) < 1e-6
        om, oe = FO.vote(depths[ref], K[ref], E[ref], depths[src], K[src], E[src], 9.0, 2900.0)
        assert (geo.bool().cpu() != om).float().mean() < 5e-4
        assert rel_l1(est.cpu(), oe) < 1e-5


def test_fusion_driver_matches_reference_capture(dev, golden, tmp_path):
    """The whole driver - PFMs in, ten bisection rounds, masks + point cloud out - against the capture of the reference's
    `fusion()` on the same depth maps."""
    from cer_mvs_amd import fusion
    from cer_mvs_amd.inference import write_pfm
    g = golden("fusion")
    images, depths, K, E, V = _scene(g)
    N = V + 1
    names = [f"{i:08d}" for i in range(N)]
    (tmp_path / "depths").mkdir()
    for i in range(N):
        write_pfm(tmp_path / "depths" / f"{names[i]}.pfm", depths[i].numpy())
    loader = []
    for i in range(N):
        order = [i] + [j for j in range(N) if j != i]
        loader.append((images[:, order].clone(), E[None][:, order].clone(), K[None][:, order].clone(), [(names[j],) for j in order], None))
    out = fusion.fusion(loader, tmp_path, glb=0.25)
    ref_masks = g["final_masks"] > 0
    assert out["masks"].shape == ref_masks.shape
    assert (out["masks"] != ref_masks).mean() < 1e-3
    assert abs(len(out["xyz"]) - len(g["ply_xyz"])) <= 16
    both = out["masks"] & ref_masks
    assert both.mean() > 0.2
    # the point cloud, compared where both agree on the mask (per view order is the same: row-major masked pixels)
    if np.array_equal(out["masks"], ref_masks):
        assert rel_l1(torch.from_numpy(out["xyz"]).float(), torch.from_numpy(g["ply_xyz"])) < 1e-5
        assert np.array_equal(out["rgb"], g["ply_rgb"])
    assert (tmp_path / "result.ply").exists() and (tmp_path / "mask").is_dir()
    with open(tmp_path / "result.ply", "rb") as f:
        head = f.read(200).decode("ascii", "replace")
    assert head.startswith("ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % len(out["xyz"]))


def test_full_size_properties(dev):
    """1600x1184, 10 source views (BASELINE configs[1] size): determinism, a view against itself is fully consistent, and a
    source view whose depth is off by 20 % votes for nothing."""
    from cer_mvs_amd import fusion
    from cer_mvs_amd.synthetic import synthetic_depth_maps, synthetic_scene
    H, W, V = 1184, 1600, 10
    _, poses, intr, _ = synthetic_scene(32, 32, V, seed=1)
    K, E = intr[0].clone(), poses[0]
    K[:, 0, 0] = K[:, 1, 1] = 1.8 * W
    K[:, 0, 2], K[:, 1, 2] = W / 2.0, H / 2.0
    depths = synthetic_depth_maps(H, W, V, seed=1, noise=0.0, outliers=0.0).to(dev)
    src = list(range(1, V + 1))
    a, ea = fusion.vote(depths[0], K[0], E[0], depths[src], K[src], E[src], 4.0, 1300.0)
    b, eb = fusion.vote(depths[0], K[0], E[0], depths[src], K[src], E[src], 4.0, 1300.0)
    assert torch.equal(a, b) and torch.equal(ea, eb)
    assert a.float().mean() > 0.9                                   # exact plane depths: (almost) every pixel is consistent
    assert rel_l1(ea.cpu()[a.bool().cpu()], depths[0].cpu()[a.bool().cpu()]) < 1e-4
    same = depths[0][None].repeat(3, 1, 1).contiguous()
    m, e = fusion.vote(depths[0], K[0], E[0], same, K[[0, 0, 0]], E[[0, 0, 0]], 4.0, 1300.0)
    assert m[1:-1, 1:-1].all() and rel_l1(e.cpu(), depths[0].cpu()) < 1e-6
    bad = (depths[src] * 1.2).contiguous()
    m2, e2 = fusion.vote(depths[0], K[0], E[0], bad, K[src], E[src], 4.0, 1300.0)
    assert m2.float().mean() < 1e-3 and rel_l1(e2.cpu(), depths[0].cpu()) < 1e-3
    # against the CPU oracle at full size, at a tight threshold that cuts through the noise (mask area ~0.5): the mask may
    # differ only in the borderline band (dist is a difference of O(W) coordinates), the averaged depth not at all
    from oracle import fusion_oracle as FO
    noisy = synthetic_depth_maps(H, W, V, seed=1)
    gm, est = fusion.vote(noisy[0].to(dev), K[0], E[0], noisy[src].to(dev), K[src], E[src], 33.0, 33.0 * 325)
    om, oe = FO.vote(noisy[0], K[0], E[0], noisy[src], K[src], E[src], 33.0, 33.0 * 325)
    assert 0.2 < om.float().mean() < 0.8
    assert (gm.bool().cpu() != om).float().mean() < 2e-3
    assert rel_l1(est.cpu(), oe) < 1e-6
