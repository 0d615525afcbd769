"""Run-to-run determinism and the shipped regime.

Two rounds in a row a kernel VARIANT produced timing-dependent wrong output (round 2: a lookup specialisation under GPU sharing;
round 3: a disparity-generator variant of the s16 conv), was replaced by a form that never failed, and shipped without a root
cause.  These tests are the guard on what ships: every epilogue / arithmetic form / tile height of the update-block convolutions,
the lookup, the cost volume and the round-4 producer / consumer encoder kernels are launched hundreds of times on fixed inputs and
must reproduce the first launch bit for bit; the shipped inference regime (three depth maps in flight at the bench workload) must
reproduce the reference capture every time; and the default arithmetic (s16f8) must stay inside 5e-5 of the oracle when the
weights get larger / heavier-tailed than the golden ones."""
import pytest
import torch

from conftest import REPO, cached_scene, rel_l1
from test_oracle_golden import hashed

pytestmark = pytest.mark.gpu

import os

LAUNCHES = int(os.environ.get("CER_DET_LAUNCHES", "200"))       # (the schedule-fuzz build is run with 500: tools/fuzz_schedule.sh)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


def _nhwc(x):
    return x[0].permute(1, 2, 0).reshape(-1, x.shape[1]).contiguous()


def _repeat(fn, n=LAUNCHES):
    """fn() -> tensor or tuple of tensors (may alias persistent buffers: compared against clones of the first result)."""
    first, bad = None, 0
    for _ in range(n):
        out = fn()
        out = out if isinstance(out, (tuple, list)) else (out,)
        if first is None:
            first = [o.clone() for o in out]
            assert all(torch.isfinite(o.float()).all() for o in first)
        elif not all(torch.equal(a, b) for a, b in zip(out, first)):
            bad += 1
    return bad


@pytest.mark.parametrize("size", [(74, 100), (296, 400)], ids=["quarter", "bench"])
@pytest.mark.parametrize("mt", [2, 4])
@pytest.mark.parametrize("f8", [False, True, 6], ids=["s16", "s16f8", "s16f6"])
def test_update_block_convs_are_deterministic(dev, f8, mt, size):
    """z|r gates, GRU blend, fused delta head, ReLU conv (corr2 shape) and the hoisted linear conv - rim tiles, interior tiles, partial
    last tiles - 200 launches each, every output bit-identical to the first launch's."""
    from cer_mvs_amd import _lib as L, ops
    h, w = size
    if size == (296, 400) and mt == 2:
        pytest.skip("half-height tiles are selected for slabs only")
    ops.TILE_MT = mt
    try:
        P = h * w
        U, R, Dp = L.S16_UNIT, L.S16_RELU, L.S16_DISP
        net = torch.tanh(hashed((1, 64, h, w), 901, -2, 2))
        c2 = torch.relu(hashed((1, 64, h, w), 902, -1, 2))
        disp = hashed((P,), 903, 0.0005, 0.0025).to(dev)
        src = [(64, 2, U), (49, 1, Dp), (64, 2, R)]
        pzr = ops.PackedConvS16(hashed((128, 177, 3, 3), 905, -0.05, 0.05), None, src, dev, corr_fp8=f8)
        pq = ops.PackedConvS16(hashed((64, 177, 3, 3), 906, -0.05, 0.05), None, src, dev, corr_fp8=f8)
        pr = ops.PackedConvS16(hashed((64, 64, 3, 3), 907, -0.05, 0.05), hashed((64,), 908, -0.1, 0.1), [(64, 2, R)], dev, corr_fp8=f8)
        pd = ops.PackedConvS16(hashed((256, 64, 3, 3), 909, -0.08, 0.08), hashed((256,), 910, -0.1, 0.1), [(64, 2, U)], dev, corr_fp8=f8)
        proj = ops.delta_proj_pack_s16(hashed((1, 256, 3, 3), 911, -0.05, 0.05), dev)
        net_s = ops.to_frag16(_nhwc(net).to(dev), h, w, U)
        c2_s = ops.to_frag16(_nhwc(c2).to(dev), h, w, R)
        init = ops.s16_layout(hashed((P, 128), 904, -0.3, 0.3).to(dev), h, w, L.S16_ACC32)
        initq = ops.s16_layout(hashed((P, 64), 912, -0.3, 0.3).to(dev), h, w, L.S16_ACC32)
        z, rh = ops.conv3x3_s16(pzr, [net_s, disp, c2_s], h, w, L.EPI_GATES, aux=net_s, init=init, log2s_out=U, log2s_aux=U)
        z, rh = z.clone(), rh.clone()
        cases = {
            "gates": lambda: ops.conv3x3_s16(pzr, [net_s, disp, c2_s], h, w, L.EPI_GATES, aux=net_s, init=init, log2s_out=U, log2s_aux=U),
            "gru": lambda: ops.conv3x3_s16(pq, [rh, disp, c2_s], h, w, L.EPI_GRU, aux=net_s, aux2=z, init=initq, log2s_out=U, log2s_aux=U),
            "delta": lambda: ops.conv3x3_s16(pd, [net_s], h, w, L.EPI_DELTA, aux=proj),
            "relu": lambda: ops.conv3x3_s16(pr, [c2_s], h, w, L.EPI_RELU, log2s_out=R),
            "linear": lambda: ops.conv3x3_s16(pzr, [net_s, disp, c2_s], h, w, L.EPI_LINEAR),
        }
        for name, fn in cases.items():
            bad = _repeat(fn)
            assert bad == 0, f"{name}: {bad} of {LAUNCHES - 1} launches differ from the first (f8={f8}, mt={mt}, {h}x{w})"
        assert not ops.check_overflow(dev)
    finally:
        ops.TILE_MT = 0


def test_lookup_and_cost_volume_are_deterministic(dev):
    """cer_lookup_encode_f32 (persistent blocks, register prefetch) and cer_cost_lines_f32 (epipolar-line tiles + view reduction) on a
    quarter-size scene: 200 launches, bit-identical."""
    from cer_mvs_amd import RAFT, ops
    from cer_mvs_amd.projective import pij_matrices
    from cer_mvs_amd.synthetic import fill_state_dict
    H, W, V = 592, 800, 4
    images, poses, intr, scale = cached_scene(H, W, V, 21)
    model = RAFT(test_mode=True)
    model.load_state_dict(fill_state_dict(model.state_dict(), seed=9))
    model = model.to(dev).eval()
    h, w = H // 4, W // 4
    with torch.no_grad():
        net, inp, f1, f2 = model.encode(images.to(dev).float(), list(range(1, V + 1)), raw=True)
        p = poses.clone().float()
        p[..., :3, 3] *= float(scale)
        k = intr.clone().float()
        k[:, :, :2] /= 4
        Pij = pij_matrices(p[0], k[0], [0] * V, list(range(1, V + 1))).to(dev)
        disp = torch.zeros(h * w, device=dev)
        split = (ops.feat_split(f1), ops.feat_split(f2))
        (D, incre, _) = model.stages()[0]
        build = lambda: ops.cost_build(f1, f2, Pij, disp, D, incre, True, h, w, 3, fold=True, pyramid_scale=1.0 / V, split=split)
        assert _repeat(build) == 0
        vol, origin = build()
        ub = model.update_block
        ub.corr_fp8, ub.conv_mode = True, "s16"
        pk = ub.packed(0, dev)
        from cer_mvs_amd import _lib as L
        d1 = torch.full((h * w,), 0.0012, device=dev)
        look = lambda: ops.lookup_encode(vol, origin, d1, pk["w0t"], pk["b0"], D, incre, ub.num_levels, ub.radius, out_split=2, log2s=L.S16_RELU, img_w=w)
        assert _repeat(look) == 0
        # round 5: the forms RAFT.forward runs - level-0-only rows (pooled levels formed in LDS) and the previous iteration's disparity update
        # riding on the launch (wave 3 publishes the new disparities to the block through LDS: a new happens-before edge, csrc/lookup.hip)
        build_c = lambda: ops.cost_build(f1, f2, Pij, disp, D, incre, True, h, w, 3, fold=True, pyramid_scale=1.0 / V, split=split, compact=True)
        assert _repeat(build_c) == 0
        volc, _ = build_c()
        assert torch.equal(volc[:, :D], vol[:, :D])
        T = hashed((2, 9, h * w), 941, -0.02, 0.02).to(dev)

        def look5():
            d = d1.clone()
            return d, ops.lookup_encode(volc, origin, d, pk["w0t"], pk["b0"], D, incre, ub.num_levels, ub.radius, out_split=2, log2s=L.S16_RELU, img_w=w,
                                        delta=(T, 0.003))
        assert _repeat(look5) == 0
    assert not ops.check_overflow(dev)


@pytest.mark.parametrize("f6", [False, True])
@pytest.mark.parametrize("which", ["fnet", "cnet"])
def test_encoder_engine_is_deterministic(dev, which, f6):
    """The round-4 producer / consumer encoder (stem + convolutions with on-the-fly merges): 40 whole encoder passes over 3 images at
    592 x 800 (partial tiles in both directions at quarter resolution), every output bit-identical to the first pass."""
    from cer_mvs_amd import RAFT
    from cer_mvs_amd.encoder_hip import HipEncoder
    from cer_mvs_amd.synthetic import fill_state_dict
    images, _, _, _ = cached_scene(592, 800, 2, 21)
    model = RAFT(test_mode=True)
    model.load_state_dict(fill_state_dict(model.state_dict(), seed=13))
    eng = HipEncoder(getattr(model, which), dev)
    eng.f6 = f6                            # (round 6: the FP6-correction form - producers exchange block maxima by DPP, four 4-byte LDS stores per item)
    x = images[0].float().to(dev)
    with torch.no_grad():
        if which == "fnet":
            fn = lambda: eng.features(x, n_ref=1, raw=True)[:2]
        else:
            fn = lambda: eng.context(x[:1], raw=True)[:2]
        assert _repeat(fn, 40) == 0


def test_three_depth_maps_in_flight_reproduce_the_capture(dev, golden):
    """The shipped inference regime: pipeline.DepthMapPipeline(streams=3) at BASELINE configs[1] (1600x1184, 10 views, 32 iterations),
    36 forwards: each bit-identical to the first and within the bar of the reference's own output (tests/golden/e2e_cfg2.npz)."""
    from cer_mvs_amd import RAFT
    from cer_mvs_amd.pipeline import DepthMapPipeline
    from cer_mvs_amd.synthetic import fill_state_dict
    g = golden("e2e_cfg2")
    H, W, V = int(g["H"]), int(g["W"]), int(g["V"])
    cascade = [tuple(int(x) for x in c) for c in g["cascade"]]
    images, poses, intr, scale = cached_scene(H, W, V, int(g["scene_seed"]))
    model = RAFT(cascade=cascade, test_mode=True)
    model.load_state_dict(fill_state_dict(model.state_dict(), seed=int(g["weight_seed"])))
    model = model.to(dev).eval()
    scene = (images.to(dev), poses.to(dev), intr.to(dev), scale)
    pipe = DepthMapPipeline(model, streams=3)
    ref = torch.from_numpy(g["disp"])
    first, n = None, 0
    for out in pipe.map([scene] * 36):
        if first is None:
            first = out.clone()
            err = rel_l1(first.cpu(), ref)
            print(f"three in flight, cfg2: rel-L1 vs the reference capture {err:.3e}")
            assert err < 1e-4
        else:
            assert torch.equal(out, first), f"forward {n} differs from the first"
        n += 1
    assert n == 36 and pipe.check_overflow() == 0


@pytest.mark.parametrize("tail", ["uniform", "heavy"])
@pytest.mark.parametrize("gain", [1.0, 2.0, 4.0])
def test_precision_margin_under_weight_gain(dev, gain, tail):
    """the fp8-correction form keeps ~15 product bits in the update block's correction terms.  At 32 GRU iterations,
    with the update block's conv weights scaled by 1 / 2 / 4 (and a heavy-tailed filler: 2 % of the weights 8 x larger), every
    precision is run against the exact-fp32 kernels (gru_precision="fp32").  Measured (MI355X): s16f8 sits 22-38 x further from fp32
    than the all-f16 forms - 3e-6 / 5e-6 at gain 1, 1.2e-5 at gain 2, but 2.8e-4 at gain 2 heavy-tailed and 5.2e-4 at gain 4 - so it
    cannot be an unconditional default.  The default is "auto": the first forward of a set of weights runs both forms and keeps the
    fp8 one only if they agree within RAFT.AUTO_TOL.  Asserted here: "auto" ends below 5e-5 (half the bar) wherever the recurrence
    itself is well-conditioned, i.e. wherever the fp32-CLASS forms reproduce exact fp32 to 2e-5; at gain 4 heavy-tailed nothing does
    (s16 / f16x3: 5e-3 - the GRU amplifies one-ulp differences), and there "auto" must simply have picked the fp32-class form."""
    from cer_mvs_amd import RAFT
    from cer_mvs_amd.synthetic import fill_state_dict
    import warnings
    H, W, V = 160, 224, 3
    cascade = [(64, 64, 16), (-1, 320, 16)]
    images, poses, intr, scale = cached_scene(H, W, V, 33)
    base = RAFT(cascade=cascade, test_mode=True)
    sd = fill_state_dict(base.state_dict(), seed=41)
    gen = torch.Generator().manual_seed(5)
    for k_, v in sd.items():
        if k_.startswith("update_block.") and k_.endswith("weight") and v.dim() == 4 and "delta" not in k_:
            v = v * gain
            if tail == "heavy":
                m = torch.rand(v.shape, generator=gen) < 0.02
                v = torch.where(m, v * 8.0, v)
            sd[k_] = v
    outs, choice = {}, None
    for prec in ("fp32", "s16", "s16f8", "s16f8+e6", "s16f8+e6+c2", "f16x3", "auto"):     # (round 6: "+e6" the encoders' correction terms in FP6, "+c2" two-term cost-volume dots)
        sfx = prec.split("+")[1:]
        model = RAFT(cascade=cascade, test_mode=True, gru_precision=prec.split("+")[0], enc_precision="f6" if "e6" in sfx else "auto",
                     cost_precision="x2" if "c2" in sfx else "auto")
        model.load_state_dict(sd)
        model = model.to(dev).eval()
        model.overflow_policy = "ignore"
        with torch.no_grad(), warnings.catch_warnings():
            warnings.simplefilter("ignore")
            outs[prec] = model(images.to(dev), poses.to(dev), intr.to(dev), scale=scale).cpu()
            if prec == "auto":
                choice, cal = model.auto_choice, model.auto_error
                again = model(images.to(dev), poses.to(dev), intr.to(dev), scale=scale).cpu()     # calibrated: one forward, same form
                assert torch.equal(again, outs[prec]) and model.auto_choice == choice
        assert torch.isfinite(outs[prec]).all()
    e = {p: rel_l1(outs[p], outs["fp32"]) for p in ("s16", "s16f8", "f16x3", "auto")}
    print(f"gain x{gain} {tail}: rel-L1 vs exact fp32: s16 {e['s16']:.2e}  s16f8 {e['s16f8']:.2e}  f16x3 {e['f16x3']:.2e}  auto {e['auto']:.2e} "
          f"(kept {choice}, calibration {cal:.2e}; s16f8 / s16 = {e['s16f8'] / max(e['s16'], 1e-12):.0f})")
    assert choice in RAFT.AUTO_FORMS
    assert torch.equal(outs["auto"], outs[choice])
    if max(e["s16"], e["f16x3"]) < 2e-5:
        assert e["auto"] < 5e-5
        if e["s16f8"] > 5e-5:
            assert choice == "s16"
    else:
        assert choice == "s16"


def test_lookup_and_convs_reproduce_next_to_an_fp16_gemm(dev):
    """Round 4's root cause of the intermittent failures of rounds 2 and 3: packed-fp32 instructions that take their low
    result from the high half of src1 go wrong on MI355X while f16 / bf16 MFMA waves share the CU.  tests/test_isa_hazard_cpu.py keeps the
    form out of the library; this is the dynamic side: every kernel family of the forward (lookup in both row forms, the update-block
    convolutions in both arithmetic forms, the fused delta head, the cost volume, the encoder) runs on one stream while an fp16 GEMM
    (hipBLASLt) runs on another, and every launch must reproduce the solo result.  The library the process loaded must also be one a
    scan has vouched for (the record build() / the CPU test leave next to it: a travelled .so is matched by its sha256)."""
    import sys as _sys
    _sys.path.insert(0, os.path.join(REPO, "tools"))
    import check_isa
    from cer_mvs_amd import _lib as _L
    if check_isa.have_objdump():
        assert not check_isa.scan(_L.LIB_PATH)[2], "the loaded library contains the hazardous packed-fp32 form"
    else:
        assert check_isa.verify_sidecar(_L.LIB_PATH), "the loaded library carries no record of a passed ISA scan of these bytes"
    from cer_mvs_amd import _lib as L, ops
    h, w, D = 296, 400, 64
    P = h * w
    U, R, Dp = L.S16_UNIT, L.S16_RELU, L.S16_DISP
    vol = hashed((P, 112), 921, -8, 8).to(dev)
    origin = torch.full((P,), 0.00125, device=dev)
    incre = 0.0025 / 64
    disp = hashed((P,), 922, 0.0, 60 * incre).to(dev)
    wt, b = hashed((33, 64), 923, -0.5, 0.5).to(dev), hashed((64,), 924, -0.5, 0.5).to(dev)
    net_s = ops.to_frag16(_nhwc(torch.tanh(hashed((1, 64, h, w), 925, -2, 2))).to(dev), h, w, U)
    c2_s = ops.to_frag16(_nhwc(torch.relu(hashed((1, 64, h, w), 926, -1, 2))).to(dev), h, w, R)
    dsp = hashed((P,), 927, 0.0005, 0.0025).to(dev)
    src = [(64, 2, U), (49, 1, Dp), (64, 2, R)]
    wzr = hashed((128, 177, 3, 3), 928, -0.05, 0.05)
    pz = {f8: ops.PackedConvS16(wzr, None, src, dev, corr_fp8=f8) for f8 in (False, True)}
    cases = {
        "lookup": lambda: ops.lookup_encode(vol, origin, disp, wt, b, D, incre, 3, 5, out_split=2, log2s=4, img_w=w),
        "gates s16": lambda: ops.conv3x3_s16(pz[False], [net_s, dsp, c2_s], h, w, L.EPI_GATES, aux=net_s, log2s_out=U, log2s_aux=U),
        "gates s16f8": lambda: ops.conv3x3_s16(pz[True], [net_s, dsp, c2_s], h, w, L.EPI_GATES, aux=net_s, log2s_out=U, log2s_aux=U),
    }
    # round 5: every kernel family of the forward, not only the two that had failed - the level-0-row lookup with the
    # disparity update riding on it, the fused delta head, the 64-channel GRU conv, the cost volume (its own MFMA waves share its CUs: that is
    # where the vectorised reciprocal projection of profiles/r05_cost_lines_fastdiv_ab.txt went wrong) and the encoder's 32 -> 32 layers
    vol0 = vol[:, :64].contiguous()
    Tpl = hashed((2, 9, P), 931, -0.02, 0.02).to(dev)
    pd = ops.PackedConvS16(hashed((256, 64, 3, 3), 932, -0.08, 0.08), hashed((256,), 933, -0.1, 0.1), [(64, 2, U)], dev, corr_fp8=True)
    proj = ops.delta_proj_pack_s16(hashed((1, 256, 3, 3), 934, -0.05, 0.05), dev)
    pq = ops.PackedConvS16(hashed((64, 177, 3, 3), 935, -0.05, 0.05), None, src, dev, corr_fp8=True)
    z_s = torch.rand(ops.s16_pixels(h, w), 64, device=dev)

    def lookup_delta():
        d = disp.clone()
        return d, ops.lookup_encode(vol0, origin, d, wt, b, D, incre, 3, 5, out_split=2, log2s=4, img_w=w, delta=(Tpl, 0.01))
    cases["lookup level-0 rows + delta"] = lookup_delta
    cases["delta head s16f8"] = lambda: ops.conv3x3_s16(pd, [net_s], h, w, L.EPI_DELTA, aux=proj)
    cases["gru q s16f8"] = lambda: ops.conv3x3_s16(pq, [net_s, dsp, c2_s], h, w, L.EPI_GRU, aux=net_s, aux2=z_s, log2s_out=U, log2s_aux=U)
    from cer_mvs_amd import RAFT
    from cer_mvs_amd.encoder_hip import HipEncoder
    from cer_mvs_amd.projective import pij_matrices
    from cer_mvs_amd.synthetic import fill_state_dict, synthetic_scene
    m_ = RAFT(test_mode=True)
    m_.load_state_dict(fill_state_dict(m_.state_dict(), seed=13))
    imgs, poses, intr, _ = synthetic_scene(256, 320, 3, seed=8)
    eng = HipEncoder(m_.fnet, dev)
    x_img = (imgs[0].float() * (2 / 255.0) - 1).to(dev)
    cases["encoder trunk + head"] = lambda: eng.forward_nchw(x_img)
    hc, wc = 64, 80
    f1 = hashed((hc * wc, 64), 936, -4, 4).to(dev)
    f2 = torch.zeros(3, (hc + 4) * (wc + 4), 64, device=dev)
    f2.view(3, hc + 4, wc + 4, 64)[:, 2:-2, 2:-2] = hashed((3, hc, wc, 64), 937, -4, 4).to(dev)
    intr4 = intr.clone()
    intr4[:, :, :2] /= 4
    Pij = pij_matrices(poses[0], intr4[0], [0] * 3, [1, 2, 3]).to(dev)
    split = (ops.feat_split(f1), ops.feat_split(f2))
    d0 = torch.zeros(hc * wc, device=dev)
    cases["cost volume (line tiles)"] = lambda: ops.cost_build(f1, f2, Pij, d0, 64, incre, True, hc, wc, 3, fold=True, pyramid_scale=1.0 / 3,
                                                               split=split, compact=True)
    a16 = torch.randn(2048, 2048, device=dev, dtype=torch.float16)
    sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
    for name, fn in cases.items():
        solo = fn()
        solo = [o.clone() for o in (solo if isinstance(solo, (tuple, list)) else (solo,))]
        torch.cuda.synchronize()
        bad = 0
        for _ in range(10):
            with torch.cuda.stream(sB):
                for _ in range(30):
                    a16 @ a16
            outs = []
            with torch.cuda.stream(sA):
                for _ in range(10):
                    o = fn()
                    outs.append([t.clone() for t in (o if isinstance(o, (tuple, list)) else (o,))])
            torch.cuda.synchronize()
            bad += sum(0 if all(torch.equal(x, y) for x, y in zip(o, solo)) else 1 for o in outs)
        assert bad == 0, f"{name}: {bad} of 100 launches next to an fp16 GEMM differ from the solo result"


def test_pipeline_replicas_adopt_the_first_models_precision_decision(dev, golden):
    """gru_precision="auto" under DepthMapPipeline: the first model calibrates (its first AUTO_INPUTS submits), the replicas take over its decision instead of
    calibrating on whatever input reaches them first - one arithmetic form per pipeline, one calibration per set of weights - and calibrate
    again only after the weights changed."""
    from cer_mvs_amd import RAFT
    from cer_mvs_amd.pipeline import DepthMapPipeline
    from cer_mvs_amd.synthetic import fill_state_dict
    g = golden("e2e_cfg1")
    cascade = [tuple(int(x) for x in c) for c in g["cascade"]]
    model = RAFT(cascade=cascade, test_mode=True)
    model.load_state_dict(fill_state_dict(model.state_dict(), seed=int(g["weight_seed"])))
    model = model.to(dev).eval()
    images, poses, intr, scale = cached_scene(int(g["H"]), int(g["W"]), int(g["V"]), int(g["scene_seed"]))
    x = (images.to(dev), poses.to(dev), intr.to(dev), scale)
    pipe = DepthMapPipeline(model, streams=3)
    calls = []
    for m in pipe.models:
        orig = m._forward_calibrating
        m._forward_calibrating = (lambda *a, _o=orig, _m=m, **k: (calls.append(id(_m)), _o(*a, **k))[1])
    outs = list(pipe.map([x] * 7))
    n_cal = RAFT.AUTO_INPUTS                                  # (round 5: the decision rests on the first three inputs, all taken by the first model)
    assert calls == [id(pipe.models[0])] * n_cal, "only the first model calibrates"
    assert all(m.auto_choice == pipe.models[0].auto_choice and not m._auto_pending() for m in pipe.models)
    assert all(torch.equal(o, outs[0]) for o in outs)
    with torch.no_grad():
        for p_ in pipe.models[0].parameters():
            p_.mul_(1.0)                                   # in-place update: new version counters
    pipe.refresh_weights()
    calls.clear()
    outs2 = list(pipe.map([x] * 5))
    assert calls == [id(pipe.models[0])] * n_cal and all(torch.equal(o, outs[0]) for o in outs2)


def test_auto_precision_is_decided_on_the_worst_of_the_first_inputs(dev):
    """VERDICT r4 "weak" 1(c) / item 7(i): how far the fp8-correction form drifts from the fp32-class form depends on the SCENE as well
    as on the weights (tools/archive/r05/explore_auto.py: update-block convs x 1.5, heavy-tailed - 7.7e-6 on a noise image stack, 2.1e-5 / 4.7e-5
    on two textured scenes, 3.7e-5 on an untextured one; tolerance 2.5e-5).  A first input inside the tolerance and a second one outside:
    round 4's rule - decide on the first input - kept the fp8 form for good; the decision now rests on the first AUTO_INPUTS inputs and
    the model must end on "s16", with the second result already the fp32-class one."""
    from cer_mvs_amd import RAFT
    from cer_mvs_amd.synthetic import fill_state_dict
    import warnings
    H, W, V = 160, 224, 3
    cascade = [(64, 64, 16), (-1, 320, 16)]
    images, poses, intr, scale = cached_scene(H, W, V, 7)
    noise = torch.rand(images.shape, generator=torch.Generator().manual_seed(1)) * 255
    sd = fill_state_dict(RAFT(cascade=cascade, test_mode=True).state_dict(), seed=41)
    gen = torch.Generator().manual_seed(5)
    for k_, v in sd.items():
        if k_.startswith("update_block.") and k_.endswith("weight") and v.dim() == 4 and "delta" not in k_:
            v = v * 1.5
            sd[k_] = torch.where(torch.rand(v.shape, generator=gen) < 0.02, v * 8.0, v)

    def make(prec):
        m = RAFT(cascade=cascade, test_mode=True, gru_precision=prec, enc_precision="f16x3", cost_precision="x3")   # (the update block's walk: the rest pinned)
        m.load_state_dict(sd)
        m = m.to(dev).eval()
        m.overflow_policy = "ignore"
        return m
    hard = (images.to(dev), poses.to(dev), intr.to(dev))
    easy = (noise.to(dev), poses.to(dev), intr.to(dev))
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m8, m16 = make("s16f8"), make("s16")
        e_easy = rel_l1(m8(*easy, scale=scale).cpu(), m16(*easy, scale=scale).cpu())
        ref_hard = m16(*hard, scale=scale).cpu()
        e_hard = rel_l1(m8(*hard, scale=scale).cpu(), ref_hard)
        print(f"fp8-correction vs all-f16 form: first input {e_easy:.2e}, second input {e_hard:.2e} (tolerance {RAFT.AUTO_TOL:.1e})")
        assert e_easy <= RAFT.AUTO_TOL < e_hard, "the test's premise: an input inside and an input outside the tolerance"
        auto = make("auto")
        auto(*easy, scale=scale)
        assert auto.auto_choice == "s16f8" and auto._auto_pending()         # undecided: inside the tolerance so far
        out = auto(*hard, scale=scale).cpu()
        assert auto.auto_choice == "s16" and not auto._auto_pending()
        assert torch.equal(out, ref_hard)                                   # the harder input already got the fp32-class result
        assert torch.equal(auto(*hard, scale=scale).cpu(), ref_hard)
        # round 6: the calibration walks a LIST of candidates.  With the FP6-correction form in front (not a default candidate: DESIGN.md 3n)
        # the easy input keeps it or demotes to the fp8 form - whichever is inside the tolerance on it - and the hard input ends on "s16"
        m6 = make("s16f6")
        e6 = rel_l1(m6(*easy, scale=scale).cpu(), m16(*easy, scale=scale).cpu())
        walk = make("auto")
        walk.AUTO_FORMS = ("s16f6", "s16f8", "s16")
        walk.update_block.corr_fp8 = 6
        o_easy = walk(*easy, scale=scale).cpu()
        print(f"FP6-correction vs all-f16 form on the first input: {e6:.2e}; the walk kept {walk.auto_choice}")
        assert walk.auto_choice == ("s16f6" if e6 <= RAFT.AUTO_TOL else "s16f8") and walk._auto_pending()
        assert torch.equal(o_easy, (m6 if walk.auto_choice == "s16f6" else m8)(*easy, scale=scale).cpu())
        assert torch.equal(walk(*hard, scale=scale).cpu(), ref_hard) and walk.auto_choice == "s16" and not walk._auto_pending()
