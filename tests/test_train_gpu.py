"""GPU tests of the training row and of API surface the inference path does not touch: DirectCorr under
autograd, the deterministic correlation backward, the training-mode forward (prediction list) + sequence loss, the literal
ConvGRU.forward, per-stage GRU weights, packed-weight cache invalidation.  Run on the GPU box: pytest -m gpu."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import cached_scene, rel_l1
from test_oracle_golden import hashed

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


def _corr_case(radius=0):
    B, N, H, W, C = 1, 3, 8, 12, 64
    f1, f2 = hashed((B, H, W, C), 71), hashed((B, H, W, C), 72)
    xy = torch.stack([hashed((B, N, H, W), 73, -1.0, W + 0.5), hashed((B, N, H, W), 74, -1.0, H + 0.5)], -1).contiguous()
    g = hashed((B, N, (2 * radius + 1) ** 2, H, W), 75)
    return f1, f2, xy, g


@pytest.mark.parametrize("deterministic", [True, False])
def test_alt_corr_backward_both_paths(dev, deterministic):
    """fmap1 / fmap2 gradients of the radius-0 op equal autograd through the oracle's grid_sample form - for the one-call atomic
    form (the reference's, correlation_kernel.cu:122-256) and for the sorted segmented reduction."""
    from cer_mvs_amd import ops
    from oracle import cer_oracle as O
    f1, f2, xy, g = _corr_case()
    a, b = f1.clone().requires_grad_(True), f2.clone().requires_grad_(True)
    (O.alt_corr_forward(a, b, xy) * g).sum().backward()
    g1, g2, gc = ops.alt_corr_backward(f1.to(dev), f2.to(dev), xy.to(dev), g.to(dev), 0, deterministic=deterministic)
    assert rel_l1(g1.cpu(), a.grad) < 1e-5 and rel_l1(g2.cpu(), b.grad) < 1e-5
    assert float(gc.abs().max()) == 0.0               # the reference never writes coords_grad (correlation_kernel.cu:307)


def test_alt_corr_backward_deterministic_is_bit_reproducible(dev):
    """The sorted reduction gives the same bits on every run (many samples per texel: a summation order that atomics would
    scramble) and agrees with the atomic form to rounding, also for radius 1."""
    from cer_mvs_amd import ops
    B, N, H, W, C = 1, 16, 24, 32, 64
    f1, f2 = hashed((B, H, W, C), 81).to(dev), hashed((B, H, W, C), 82).to(dev)
    xy = torch.stack([hashed((B, N, H, W), 83, 2.0, 9.0), hashed((B, N, H, W), 84, 2.0, 7.0)], -1).contiguous().to(dev)   # crowded texels
    for r in (0, 1):
        g = hashed((B, N, (2 * r + 1) ** 2, H, W), 85 + r).to(dev)
        runs = [ops.alt_corr_backward(f1, f2, xy, g, r, deterministic=True) for _ in range(3)]
        assert all(torch.equal(runs[0][1], x[1]) and torch.equal(runs[0][0], x[0]) for x in runs[1:])
        atomic = ops.alt_corr_backward(f1, f2, xy, g, r, deterministic=False)
        assert rel_l1(runs[0][1].cpu(), atomic[1].cpu()) < 1e-5 and torch.equal(runs[0][0], atomic[0])


def test_direct_corr_autograd_function(dev):
    """DirectCorr.apply (core/corr.py:12-25): forward value and both feature gradients through torch autograd."""
    from cer_mvs_amd.corr import DirectCorr
    from oracle import cer_oracle as O
    f1, f2, xy, g = _corr_case()
    a, b = f1.clone().requires_grad_(True), f2.clone().requires_grad_(True)
    ref = O.alt_corr_forward(a, b, xy)
    (ref * g).sum().backward()
    x, y, c = f1.to(dev).requires_grad_(True), f2.to(dev).requires_grad_(True), xy.to(dev).requires_grad_(True)
    out = DirectCorr.apply(x, y, c)
    assert out.shape == ref.shape and rel_l1(out.detach().cpu(), ref.detach()) < 1e-5
    (out * g.to(dev)).sum().backward()
    assert rel_l1(x.grad.cpu(), a.grad) < 1e-5 and rel_l1(y.grad.cpu(), b.grad) < 1e-5
    assert float(c.grad.abs().max()) == 0.0


def test_convgru_forward_literal(dev):
    """ConvGRU.forward (core/update.py:17-25) on the HIP kernels against the same module evaluated with torch convs."""
    from cer_mvs_amd.update import ConvGRU
    h, w = 13, 21
    gru = ConvGRU(h_planes=64, i_planes=64 + 49 + 64)
    with torch.no_grad():
        for i, prm in enumerate(gru.parameters()):
            prm.copy_(hashed(tuple(prm.shape), 400 + i, -0.05, 0.05))
    net = torch.tanh(hashed((1, 64, h, w), 411, -2, 2))
    xs = [torch.relu(hashed((1, 64, h, w), 412, -1, 2)), hashed((1, 49, h, w), 413, -0.2, 0.2), torch.relu(hashed((1, 64, h, w), 414, -1, 2))]
    with torch.no_grad():
        x = torch.cat(xs, 1)
        hx = torch.cat([net, x], 1)
        z = torch.sigmoid(gru.convz(hx))
        r = torch.sigmoid(gru.convr(hx))
        q = torch.tanh(gru.convq(torch.cat([r * net, x], 1)))
        ref = (1 - z) * net + z * q
        got = gru.to(dev)(net.to(dev), *[t.to(dev) for t in xs])
    assert got.shape == ref.shape and rel_l1(got.cpu(), ref) < 1e-5


def _tiny_model(dev, golden, **kw):
    from cer_mvs_amd import RAFT
    from cer_mvs_amd.synthetic import fill_state_dict
    g = golden("e2e_tiny")
    cascade = [tuple(int(x) for x in c) for c in g["cascade"]]
    model = RAFT(cascade=cascade, **kw)
    model.load_state_dict(fill_state_dict(model.state_dict(), seed=int(g["weight_seed"])))
    images, poses, intr, scale = cached_scene(int(g["H"]), int(g["W"]), int(g["V"]), int(g["scene_seed"]))
    return model.to(dev), (images.to(dev), poses.to(dev), intr.to(dev)), scale, g, cascade


def test_training_mode_forward_and_loss(dev, golden):
    """RAFT.forward(test_mode=False) returns one prediction per GRU iteration (core/raft.py:103,109); the last one times the
    scale is the reference's test-mode output; the sequence loss (loss.py:5-41) back-propagates finite, non-zero gradients into
    the feature encoder (through the HIP correlation backward), the context encoder and the update block; inputs stay intact."""
    from cer_mvs_amd.train import sequence_loss
    model, inputs, scale, g, cascade = _tiny_model(dev, golden, test_mode=False)
    model.train()
    before = [t.clone() for t in inputs]
    preds = model(*inputs, scale=scale)
    assert isinstance(preds, list) and len(preds) == sum(c[2] for c in cascade)
    assert all(p.shape == preds[0].shape and p.requires_grad for p in preds)
    assert all(torch.equal(a, b) for a, b in zip(inputs, before))
    ref = torch.from_numpy(g["disp"])
    s = float(torch.as_tensor(scale).reshape(-1)[0])
    assert rel_l1((preds[-1] * s).detach().cpu(), ref) < 1e-4
    gt = (preds[-1].detach() * 1.1 + 1e-4).clamp_min(1e-4)
    loss, metrics = sequence_loss(preds, gt, gradual_weight=0.3)
    assert torch.isfinite(loss) and set(metrics) == {"mean_depth_error", "less3", "less10", "less25"}
    loss.backward()
    for name in ("fnet.conv1.weight", "fnet.conv2.weight", "cnet.conv2.weight", "update_block.gru.convz.weight",
                 "update_block.corr_encoder.0.weight", "update_block.delta0.0.weight", "update_block.delta1.2.weight"):
        grad = dict(model.named_parameters())[name].grad
        assert grad is not None and torch.isfinite(grad).all() and float(grad.abs().sum()) > 0, name


def test_training_row_matches_reference_capture(dev, golden):
    """The training row against the REFERENCE ITSELF (tests/golden/train_tiny.npz = tools/gen_golden.py --only train_tiny: the
    reference's RAFT(test_mode=False) + loss.sequence_loss under the generator's shims): the whole prediction list
    (core/raft.py:103,109), the loss and its metrics (loss.py:5-41), and the gradients of seven parameters - through the HIP
    correlation forward AND the deterministic HIP backward (core/corr.py:19-25)."""
    from cer_mvs_amd import RAFT
    from cer_mvs_amd.synthetic import fill_state_dict
    from cer_mvs_amd.train import sequence_loss
    g = golden("train_tiny")
    cascade = [tuple(int(x) for x in c) for c in g["cascade"]]
    model = RAFT(cascade=cascade, test_mode=False)
    model.load_state_dict(fill_state_dict(model.state_dict(), seed=int(g["weight_seed"])))
    model = model.to(dev).train()
    images, poses, intr, scale = cached_scene(int(g["H"]), int(g["W"]), int(g["V"]), int(g["scene_seed"]))
    preds = model(images.to(dev), poses.to(dev), intr.to(dev), scale=scale)
    want = torch.from_numpy(g["predictions"])
    assert len(preds) == want.shape[0]
    for i, p_ in enumerate(preds):
        assert rel_l1(p_.detach().cpu(), want[i]) < 1e-4, i
    gt = torch.from_numpy(g["gt"]).to(dev)
    loss, metrics = sequence_loss(list(preds), gt, gradual_weight=float(g["gradual_weight"]))
    assert abs(float(loss) - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    for k, v in zip(("mean_depth_error", "less3", "less10", "less25"), g["metrics"]):
        assert abs(metrics[k] - float(v)) <= 1e-4 * max(abs(float(v)), 1.0), k
    loss.backward()
    params = dict(model.named_parameters())
    for key in g.files:
        if not key.startswith("grad_"):
            continue
        name = key[5:]
        grad = params[name].grad.detach().reshape(-1).cpu()
        ref = torch.from_numpy(g[key])
        sub = grad if grad.numel() == ref.numel() else grad[::7]
        assert sub.numel() == ref.numel(), name
        assert rel_l1(sub, ref) < 1e-3, (name, rel_l1(sub, ref))
        assert abs(float(grad.double().abs().sum()) - float(g["gradsum_" + name])) <= 1e-3 * float(g["gradsum_" + name]), name


def test_sequence_loss_matches_restatement(dev):
    """sequence_loss against a direct numpy evaluation of loss.py:5-41 on small tensors."""
    from cer_mvs_amd.train import sequence_loss
    rng = np.random.RandomState(3)
    gt = torch.from_numpy(rng.uniform(0.0, 0.003, (1, 1, 8, 12)).astype(np.float32))
    gt[0, 0, :2] = 0.0                                         # invalid pixels
    est = [torch.from_numpy(rng.uniform(5e-4, 0.003, (1, 1, 4, 6)).astype(np.float32)) for _ in range(3)]
    loss, metrics = sequence_loss([e.clone() for e in est], gt, gradual_weight=0.25, gamma=0.9)
    up = [F.interpolate(e, [8, 12], mode="bilinear", align_corners=True).double().numpy() for e in est]
    g = gt.double().numpy()
    valid = g > 0
    want = 0.0
    for i, e in enumerate(up):
        wgt = 0.9 ** (3 - i - 1)
        ld = np.abs(e - g)
        lz = np.minimum(np.abs(1 / np.maximum(e, 1e-3) - 1 / np.maximum(g, 1e-3)), 100) / 3.6e5
        il = 0.25 * lz + 0.75 * ld
        want += wgt * (valid * il).mean() + 0.01 * wgt * il.mean()
    assert abs(float(loss) - want) < 1e-6 * max(abs(want), 1e-9) + 1e-12
    epe = np.abs(1 / np.maximum(up[-1], 1e-3) - 1 / g)[valid]
    assert abs(metrics["mean_depth_error"] - epe.mean()) < 1e-3 * epe.mean()


def test_unshared_gru_weights_use_their_own_hoisted_term(dev, golden):
    """UpdateBlock(share_gru=False): every stage's GRU has its own `inp` weights and biases, so the hoisted term is per stage
    (hoisting once with stage-0 weights was silently wrong).  Fast path == literal path == per-stage torch convs."""
    from cer_mvs_amd import RAFT
    from cer_mvs_amd.update import UpdateBlock
    from cer_mvs_amd.synthetic import fill_state_dict
    g = golden("e2e_tiny")
    cascade = [tuple(int(x) for x in c) for c in g["cascade"]]
    model = RAFT(cascade=cascade, test_mode=True)
    model.update_block = UpdateBlock(cascade=model.cascade, dim_net=64, dim_inp=64, share_gru=False)
    model.update_block.conv_mode = "s16"
    sd = fill_state_dict(model.state_dict(), seed=11)
    model.load_state_dict(sd)
    model = model.to(dev).eval()
    images, poses, intr, scale = cached_scene(int(g["H"]), int(g["W"]), int(g["V"]), int(g["scene_seed"]))
    args = (images.to(dev), poses.to(dev), intr.to(dev))
    with torch.no_grad():
        fast = model(*args, scale=scale)
        lit = model._forward_literal(*args, scale, False)
    assert rel_l1(fast.cpu(), lit.cpu()) < 1e-5
    w0, w1 = model.update_block.gru0.convz.weight, model.update_block.gru1.convz.weight
    assert not torch.equal(w0, w1)                              # the two stages really differ


def test_packed_encoder_weights_follow_parameter_reloads(dev, golden):
    """Loading new parameters through a submodule or a wrapper drops the packed encoder weights."""
    from cer_mvs_amd.synthetic import fill_state_dict
    model, inputs, scale, g, _ = _tiny_model(dev, golden, test_mode=True)
    model.eval()
    with torch.no_grad():
        a = model(*inputs, scale=scale)
        sd2 = fill_state_dict(model.state_dict(), seed=77)
        model.fnet.load_state_dict({k[5:]: v for k, v in sd2.items() if k.startswith("fnet.")})
        b = model(*inputs, scale=scale)
        torch.nn.DataParallel(model).load_state_dict({"module." + k: v for k, v in fill_state_dict(model.state_dict(), seed=int(g["weight_seed"])).items()})
        c = model(*inputs, scale=scale)
    assert not torch.equal(a, b) and torch.equal(a, c)


def test_packed_weights_follow_in_place_updates_and_deepcopy(dev, golden):
    """Packed weights are a cache of the parameters: an in-place update (what an optimizer step does) and a deep copy
    whose parameters are then changed must both be seen by the next test-mode forward - without refresh_weights()."""
    import copy
    model, inputs, scale, g, _ = _tiny_model(dev, golden, test_mode=True)
    model.eval()
    with torch.no_grad():
        a = model(*inputs, scale=scale)
        twin = copy.deepcopy(model)
        for p_ in list(twin.fnet.parameters())[:2] + [twin.update_block.gru.convq.weight]:
            p_.mul_(1.01)
        b = twin(*inputs, scale=scale)
        a2 = model(*inputs, scale=scale)
        model.update_block.delta1[2].bias.add_(1e-4)                 # in place on the original
        c = model(*inputs, scale=scale)
        fresh = copy.deepcopy(model)
        fresh.refresh_weights()
        c2 = fresh(*inputs, scale=scale)
    assert torch.equal(a, a2) and not torch.equal(a, b) and not torch.equal(a, c) and torch.equal(c, c2)


def test_forward_rejects_sizes_that_are_not_multiples_of_four(dev, golden):
    model, inputs, scale, _, _ = _tiny_model(dev, golden, test_mode=True)
    bad = inputs[0][..., :-2]
    with pytest.raises(RuntimeError, match="multiple of 4"):
        model(bad, inputs[1], inputs[2], scale=scale)
