"""The fusion oracle (oracle/fusion_oracle.py) against captures of the reference's own fusion.py
(tools/gen_golden_fusion.py -> tests/golden/fusion.npz): `check_geometric_consistency` outputs, and the whole ten-round
`fusion()` loop (final masks and point cloud)."""
import numpy as np
import torch

from conftest import rel_l1


def _scene(g):
    from cer_mvs_amd.synthetic import synthetic_depth_maps, synthetic_scene, tensor_checksum
    H, W, V = int(g["H"]), int(g["W"]), int(g["V"])
    _, poses, intr, _ = synthetic_scene(H, W, V, seed=int(g["scene_seed"]))
    depths = synthetic_depth_maps(H, W, V, seed=int(g["scene_seed"]))
    assert tensor_checksum(depths) == int(g["depths_checksum"])
    return depths, intr[0], poses[0], V


def test_check_geometric_consistency_matches_reference(golden):
    from oracle import fusion_oracle as FO
    g = golden("fusion")
    depths, K, E, V = _scene(g)
    S = V
    masks, mask, drep, xs, ys, rel = FO.check_geometric_consistency(
        depths[0][None].repeat(S, 1, 1), K[0][None].repeat(S, 1, 1), E[0][None].repeat(S, 1, 1), depths[1:], K[1:], E[1:], 4.0, 1300.0)
    assert np.array_equal(torch.stack(masks).numpy(), g["cgc_masks"])
    assert np.array_equal(drep.numpy(), g["cgc_depth_reprojected"])
    assert np.array_equal(xs.numpy(), g["cgc_x_src"]) and np.array_equal(ys.numpy(), g["cgc_y_src"])
    assert np.array_equal(rel.numpy(), g["cgc_rel"], equal_nan=True)
    assert 0.05 < g["cgc_masks"][-1].mean() < 0.95                      # the capture exercises both outcomes


def test_fusion_loop_matches_reference(golden):
    from oracle import fusion_oracle as FO
    g = golden("fusion")
    depths, K, E, V = _scene(g)
    N = V + 1
    pairs = [(i, [j for j in range(N) if j != i]) for i in range(N)]
    masks, est, thre, hist = FO.fuse(depths, K, E, pairs, glb=0.25)
    assert np.array_equal(masks.numpy(), g["final_masks"] > 0)
    pts = torch.cat([FO.backproject(est[i], masks[i], K[i], E[i]) for i in range(N)])
    assert pts.shape[0] == g["ply_xyz"].shape[0]
    assert rel_l1(pts.float(), torch.from_numpy(g["ply_xyz"])) < 1e-6
    assert abs(hist[-1][1] - 0.25) < 0.02
