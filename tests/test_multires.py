"""Multi-resolution merge (reference: multires.py:16-40): oracle restatement vs an independent bilinear resize (CPU), HIP kernels
vs the oracle and the file-level driver (GPU)."""
import os

import numpy as np
import pytest
import torch

from conftest import rel_l1
from test_oracle_golden import hashed


def _maps(h1, w1, h2, w2, seed):
    a = hashed((h1, w1), seed, 0.5, 3.0).numpy()
    big = torch.nn.functional.interpolate(torch.from_numpy(a)[None, None], size=(h2, w2), mode="bicubic", align_corners=False)[0, 0].numpy()
    b = (big * (1 + 0.05 * hashed((h2, w2), seed + 1, -1.0, 1.0).numpy())).astype(np.float32)      # within / beyond 2 % of the scale-1 map
    return a, b


def test_oracle_resize_matches_independent_bilinear():
    """The cv2 INTER_LINEAR restatement (oracle/multires_oracle.py) against torch's bilinear (align_corners=False): the same
    half-pixel rule when up-sampling; identity when the size does not change; exact on constants."""
    from oracle import multires_oracle as M
    for (h, w, ho, wo) in [(37, 53, 74, 106), (40, 64, 100, 160), (31, 45, 47, 91)]:
        a = hashed((h, w), 11, 0.5, 3.0).numpy()
        ref = torch.nn.functional.interpolate(torch.from_numpy(a)[None, None], size=(ho, wo), mode="bilinear", align_corners=False)[0, 0].numpy()
        got = M.resize_linear(a, (ho, wo))
        # (torch forms the source coordinate with a float32 scale, OpenCV in double: the taps' weights differ in the last bits)
        assert got.shape == (ho, wo) and np.abs(got - ref).max() < 3e-5
    a = hashed((9, 13), 12).numpy()
    assert np.array_equal(M.resize_linear(a, (9, 13)), a)
    assert np.array_equal(M.resize_linear(np.full((5, 7), 2.5, np.float32), (11, 20)), np.full((11, 20), 2.5, np.float32))
    # down-sampling by 2 from even sizes: taps at 2x + 0.5 -> the mean of two neighbours in each direction
    b = hashed((8, 12), 13).numpy()
    assert np.allclose(M.resize_linear(b, (4, 6)), (b[0::2, 0::2] + b[0::2, 1::2] + b[1::2, 0::2] + b[1::2, 1::2]) / 4, atol=1e-6)


def test_oracle_merge_select():
    from oracle import multires_oracle as M
    im1 = np.array([[1.0, 2.0], [4.0, 0.0]], np.float32)
    im2 = np.array([[1.01, 2.1], [3.95, 0.5]], np.float32)
    out = M.merge(im1, im2, th=0.02)
    assert np.array_equal(out, np.array([[1.01, 2.0], [3.95, 0.0]], np.float32))          # |d| < 2 % of im1 keeps im2; im1 = 0 never does


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [((37, 53), (74, 106)), ((60, 80), (60, 80)), ((45, 64), (121, 173))])
@pytest.mark.parametrize("down", [1, 2])
def test_multires_kernels_match_oracle(shape, down):
    from cer_mvs_amd import multires as MR
    from oracle import multires_oracle as M
    (h1, w1), (h2, w2) = shape
    a, b = _maps(h1, w1, h2, w2, 21)
    want = M.merge(a, b, 0.02, down)
    got = MR.merge(a, b, 0.02, down).cpu().numpy()
    assert got.shape == want.shape
    if down == 1:
        # the select is discontinuous: a resized value within 1 ulp of the threshold may flip; everything else is bit-comparable
        flips = np.abs(got - want) > 1e-6 * np.abs(want)
        assert flips.mean() < 1e-3
        assert np.abs(got - want)[~flips].max() <= 1e-6 * np.abs(want).max()
        frac_im2 = (got == b).mean()
        assert 0.1 < frac_im2 < 0.9                                      # both branches are exercised
    else:
        assert rel_l1(torch.from_numpy(got), torch.from_numpy(want)) < 1e-4


@pytest.mark.gpu
def test_multires_driver_files(tmp_path):
    from cer_mvs_amd import multires as MR
    from cer_mvs_amd.fusion import read_pfm
    from cer_mvs_amd.inference import write_pfm
    from oracle import multires_oracle as M
    d = tmp_path / "depths"
    os.makedirs(d)
    want = {}
    for i, name in enumerate(["00000003", "00000011"]):
        a, b = _maps(30, 40, 60, 80, 31 + 2 * i)
        write_pfm(d / f"{name}_scale1_nf10.pfm", a)
        write_pfm(d / f"{name}_scale2_nf7.pfm", b)
        want[name] = M.merge(a, b, 0.02, 1)
    written = MR.multires(tmp_path, suffix1="_nf10", suffix2="_nf7", th=0.02)
    assert [p.name for p in written] == ["00000003_nf10_nf7_th0.02.pfm", "00000011_nf10_nf7_th0.02.pfm"]
    for name, w in want.items():
        got = read_pfm(d / f"{name}_nf10_nf7_th0.02.pfm")
        assert got.shape == w.shape and (np.abs(got - w) > 1e-6 * np.abs(w)).mean() < 1e-3


@pytest.mark.gpu
def test_inference_two_scales_then_multires(tmp_path):
    """The reference's demo chain (demo.py:28-62: inference at scale 1 and 2, then multires) on a synthetic scene: the file names
    one step writes are the ones the next step reads, and the merged map has the scale-2 size."""
    from cer_mvs_amd import RAFT
    from cer_mvs_amd import inference as I
    from cer_mvs_amd import multires as MR
    from cer_mvs_amd.fusion import read_pfm
    from cer_mvs_amd.synthetic import fill_state_dict, synthetic_scene
    dev = torch.device("cuda:0")
    images, poses, intr, scale = synthetic_scene(64, 96, 2, seed=5)
    model = RAFT(cascade=[(64, 64, 1), (-1, 320, 1)], test_mode=True)
    model.load_state_dict(fill_state_dict(model.state_dict(), seed=2))
    model = model.to(dev)
    loader = [(images, poses, intr, [["00000007"]], scale)]
    I.inference(loader, None, tmp_path, rescale=1, model=model, num_frames=3)
    I.inference(loader, None, tmp_path, rescale=2, model=model, num_frames=3)
    written = MR.multires(tmp_path, suffix1="_nf3", suffix2="_nf3", th=0.02)
    assert [p.name for p in written] == ["00000007_nf3_nf3_th0.02.pfm"]
    d1 = read_pfm(tmp_path / "depths" / "00000007_scale1_nf3.pfm")
    d2 = read_pfm(tmp_path / "depths" / "00000007_scale2_nf3.pfm")
    m = read_pfm(written[0])
    assert d1.shape == (16, 24) and d2.shape == (32, 48) and m.shape == d2.shape
    from oracle import multires_oracle as M
    want = M.merge(d1, d2, 0.02, 1)
    assert (np.abs(m - want) > 1e-6 * np.abs(want)).mean() < 1e-2
