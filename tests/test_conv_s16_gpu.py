"""GPU parity of the round-2 update-block convolutions (csrc/conv_s16.hip, through the C ABI) against torch / fp64 restatements
of core/update.py:13-25,61-71,80-85.  Run on the GPU box: pytest -m gpu."""
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l1
from test_oracle_golden import hashed

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


def nhwc(x):
    """[1,C,h,w] -> [h*w, C]"""
    return x[0].permute(1, 2, 0).reshape(-1, x.shape[1]).contiguous()


@pytest.fixture
def tile_mt():
    from cer_mvs_amd import ops
    yield ops
    ops.TILE_MT = 0


@pytest.fixture(params=[False, True, 6], ids=["f16x3terms", "fp8corr", "fp6corr"])
def f8(request, monkeypatch):
    """The three arithmetic forms of the tensor sources: three f16 MFMAs per product, the two correction terms on the fp8 matrix
    instruction (CER_EPI_CORR_FP8, gru_precision="s16f8"), or on its FP6 form with one E8M0 scale per 16-channel block (round 6:
    CER_EPI_CORR_FP6, gru_precision="s16f6").  Yields the factor by which the fp32-class tolerances widen: the 8- / 6-bit operands carry the
    2^-11 correction terms with 2^-4 relative precision (FP6: relative to the block maximum for the small elements of a block), i.e.
    2^-15..2^-16 of a product."""
    from cer_mvs_amd import ops
    if request.param:
        orig = ops.PackedConvS16.__init__
        form = request.param

        def init(self, weight, bias, sources, device, corr_fp8=form):
            orig(self, weight, bias, sources, device, corr_fp8=corr_fp8)
        monkeypatch.setattr(ops.PackedConvS16, "__init__", init)
    return {False: 1.0, True: 30.0, 6: 45.0}[request.param]


def frag(x, h, w, log2s):
    """[1,C,h,w] cpu -> frag16 device tensor"""
    from cer_mvs_amd import ops
    return ops.to_frag16(nhwc(x).cuda(), h, w, log2s)


def unfrag(t, h, w, log2s):
    from cer_mvs_amd import ops
    return ops.from_frag16(t, h, w, log2s).cpu().double()


def unacc(t, h, w, layout=None):
    from cer_mvs_amd import _lib as L, ops
    return ops.s16_layout(t, h, w, L.S16_ACC32 if layout is None else layout, inverse=True).cpu().double()


@pytest.mark.parametrize("h,w", [(8, 16), (13, 37), (30, 64)])
def test_s16_layouts_roundtrip(dev, h, w):
    """Plain [h*w, C] <-> the three m-tile-major layouts (cer_mvs.h): exact for the fp32 layouts, 2^-22-class for frag16; the
    frag16 row mover copies bits."""
    from cer_mvs_amd import _lib as L, ops
    x = hashed((h * w, 64), 901, -1, 1).to(dev)
    for layout in (L.S16_ACC32, L.S16_F32X8):
        t = ops.s16_layout(x, h, w, layout)
        assert t.shape == (ops.s16_pixels(h, w), 64)
        assert torch.equal(ops.s16_layout(t, h, w, layout, inverse=True), x)
    for log2s in (L.S16_UNIT, L.S16_RELU, 0):
        rt = ops.from_frag16(ops.to_frag16(x, h, w, log2s), h, w, log2s)
        assert (rt - x).abs().max() <= max(2.0 ** -21, 2.0 ** (-24 - log2s))
    big = torch.full((h * w, 32), 1e9, device=dev)
    assert torch.isfinite(ops.from_frag16(ops.to_frag16(big, h, w, 4), h, w, 4)).all()       # saturates at 65504 / 2^4
    # the two fp32 layouts really differ from each other and from the plain order
    assert not torch.equal(ops.s16_layout(x, h, w, L.S16_ACC32)[:h * w], x)
    # row mover: rows [y0, y0+n) out and back in
    t = ops.to_frag16(x, h, w, L.S16_UNIT)
    y0, n = 3, 4
    rows = torch.empty(n * w * 64, device=dev)
    ops.s16_rows(t, rows, h, w, y0, n, False)
    t2 = torch.zeros_like(t)
    ops.s16_rows(t2, rows, h, w, y0, n, True)
    back = ops.from_frag16(t2, h, w, L.S16_UNIT)
    ref = ops.from_frag16(t, h, w, L.S16_UNIT)
    assert torch.equal(back[y0 * w:(y0 + n) * w], ref[y0 * w:(y0 + n) * w]) and back[:y0 * w].abs().sum() == 0


@pytest.mark.parametrize("mt", [0, 2, 3, 4])
@pytest.mark.parametrize("h,w,cout", [(8, 16, 64), (11, 21, 64), (9, 17, 128), (24, 40, 256), (5, 70, 128), (37, 33, 64)])
def test_conv_s16_matches_torch(dev, tile_mt, f8, h, w, cout, mt):
    from cer_mvs_amd import _lib as L, ops
    ops.TILE_MT = mt
    cin = 64
    x = hashed((1, cin, h, w), 101)
    wt = hashed((cout, cin, 3, 3), 102, -0.1, 0.1)
    b = hashed((cout,), 103)
    ref = nhwc(F.conv2d(x.double(), wt.double(), b.double(), padding=1))
    pc = ops.PackedConvS16(wt, b, [(cin, 2, L.S16_UNIT)], dev)
    xs = frag(x, h, w, L.S16_UNIT)
    out = ops.conv3x3_s16(pc, [xs], h, w, L.EPI_LINEAR)
    assert rel_l1(unacc(out, h, w), ref) < 1e-6 * f8
    out = ops.conv3x3_s16(pc, [xs], h, w, L.EPI_LINEAR, out_split=True, log2s_out=L.S16_RELU)
    assert rel_l1(unfrag(out, h, w, L.S16_RELU), ref) < 1e-6 * f8
    out = ops.conv3x3_s16(pc, [xs], h, w, L.EPI_RELU, log2s_out=L.S16_RELU)
    assert rel_l1(unfrag(out, h, w, L.S16_RELU), F.relu(ref)) < 1e-6 * f8
    init = hashed((h * w, cout), 104, -0.3, 0.3)
    out = ops.conv3x3_s16(pc, [xs], h, w, L.EPI_LINEAR, init=ops.s16_layout(init.to(dev), h, w, L.S16_ACC32))
    assert rel_l1(unacc(out, h, w), ref - b.double() + init.double()) < 1e-6 * f8


def test_conv_s16_two_sources_and_error_bound(dev, f8):
    """Two tensor sources with different scale classes; error per output relative to sum |x||w| stays fp32-class."""
    from cer_mvs_amd import _lib as L, ops
    h, w, cout = 19, 45, 128
    a = torch.tanh(hashed((1, 64, h, w), 111, -2, 2))
    c = torch.relu(hashed((1, 32, h, w), 112, -1, 3))
    wt = hashed((cout, 96, 3, 3), 113, -0.07, 0.07)
    x = torch.cat([a, c], 1)
    ref = nhwc(F.conv2d(x.double(), wt.double(), None, padding=1))
    mag = nhwc(F.conv2d(x.abs().double(), wt.abs().double(), None, padding=1))
    pc = ops.PackedConvS16(wt, None, [(64, 2, L.S16_UNIT), (32, 2, L.S16_RELU)], dev)
    out = ops.conv3x3_s16(pc, [frag(a, h, w, L.S16_UNIT), frag(c, h, w, L.S16_RELU)], h, w, L.EPI_LINEAR)
    assert float(((unacc(out, h, w) - ref).abs() / mag).max()) < 1e-6 * f8


@pytest.mark.parametrize("mt", [2, 4])
@pytest.mark.parametrize("h,w,cout", [(30, 70, 128), (13, 101, 64), (41, 50, 64)])
def test_conv_s16_disparity_source(dev, tile_mt, f8, h, w, cout, mt):
    """Kind-1 source: 100 * (unfold7x7(disp) - disp) (core/update.py:80-85,97) generated in the kernel - collapsed 81-tap form
    on interior tiles, literal form on border tiles; both must match the literal convolution, also where the 9x9 window
    leaves the image."""
    from cer_mvs_amd import _lib as L, ops
    from oracle import cer_oracle as O
    ops.TILE_MT = mt if cout == 128 else {2: 2, 4: 3}[mt]
    disp = hashed((1, 1, h, w), 211, 0.0005, 0.0025)
    a = hashed((1, 32, h, w), 212)
    feat = 100 * O.disp_features(disp)
    wt = hashed((cout, 32 + 49, 3, 3), 213, -0.1, 0.1)
    ref = nhwc(F.conv2d(torch.cat([a, feat], 1).double(), wt.double(), None, padding=1))
    pc = ops.PackedConvS16(wt, None, [(32, 2, L.S16_UNIT), (49, 1, L.S16_DISP)], dev)
    assert pc.packed_c is not None
    srcs = [frag(a, h, w, L.S16_UNIT), disp.reshape(-1).to(dev)]
    outs = {}
    # (collapsed everywhere + rim correction: the default) / (collapsed on interior tiles, literal on border tiles) / (literal)
    for mode, (coll, edge) in {"rim": (True, True), "mixed": (True, False), "literal": (False, False)}.items():
        ops.COLLAPSE_DISP, ops.EDGE_CORRECT = coll, edge
        if f8 > 1 and mode != "rim":        # the fp8-correction kernels have the collapsed form with the rim correction only
            try:
                with pytest.raises(RuntimeError):
                    ops.conv3x3_s16(pc, srcs, h, w, L.EPI_LINEAR)
            finally:
                ops.COLLAPSE_DISP, ops.EDGE_CORRECT = True, True
            continue
        try:
            outs[mode] = unacc(ops.conv3x3_s16(pc, srcs, h, w, L.EPI_LINEAR), h, w)
        finally:
            ops.COLLAPSE_DISP, ops.EDGE_CORRECT = True, True
        assert rel_l1(outs[mode], ref) < 2e-6 * f8, mode
        assert (outs[mode] - ref).abs().max() < 5e-6 * f8 * ref.abs().max(), mode
    if f8 > 1:
        return
    assert not torch.equal(outs["rim"], outs["literal"])         # the collapsed path really ran
    th = 2 * ops.TILE_MT * (1 if cout == 128 else 2)
    if h >= 2 * th + 2:                                          # at least one interior tile row
        assert not torch.equal(outs["mixed"], outs["literal"])


@pytest.mark.parametrize("h,w", [(1, 16), (2, 3), (3, 40), (40, 2), (9, 17)])
def test_conv_s16_rim_correction_degenerate_images(dev, f8, h, w):
    """Images so small that a pixel lies on several edges at once (all four for 1 x n): the rim correction sums the edges and
    removes the corner taps they share."""
    from cer_mvs_amd import _lib as L, ops
    from oracle import cer_oracle as O
    disp = hashed((1, 1, h, w), 231, 0.0005, 0.0025)
    wt = hashed((64, 49, 3, 3), 233, -0.1, 0.1)
    ref = nhwc(F.conv2d((100 * O.disp_features(disp)).double(), wt.double(), None, padding=1))
    pc = ops.PackedConvS16(wt, None, [(49, 1, L.S16_DISP)], dev)
    out = unacc(ops.conv3x3_s16(pc, [disp.reshape(-1).to(dev)], h, w, L.EPI_LINEAR), h, w)
    assert rel_l1(out, ref) < 2e-6 and (out - ref).abs().max() < 5e-6 * ref.abs().max()


def test_conv_s16_disparity_only_source(dev, f8):
    from cer_mvs_amd import _lib as L, ops
    from oracle import cer_oracle as O
    h, w, cout = 33, 37, 64
    disp = hashed((1, 1, h, w), 221, 0.0, 0.0025)
    wt = hashed((cout, 49, 3, 3), 223, -0.1, 0.1)
    ref = nhwc(F.conv2d((100 * O.disp_features(disp)).double(), wt.double(), None, padding=1))
    pc = ops.PackedConvS16(wt, None, [(49, 1, L.S16_DISP)], dev)
    out = ops.conv3x3_s16(pc, [disp.reshape(-1).to(dev)], h, w, L.EPI_LINEAR)
    assert rel_l1(unacc(out, h, w), ref) < 2e-6


def test_conv_s16_gates_and_gru_epilogues(dev, f8):
    """z|r gates (sigmoid, r*h) and the GRU blend (core/update.py:17-25) on frag16 tensors vs an fp64 restatement."""
    from cer_mvs_amd import _lib as L, ops
    from oracle import cer_oracle as O
    h, w = 23, 50
    P = h * w
    U, R, Dp = L.S16_UNIT, L.S16_RELU, L.S16_DISP
    net = torch.tanh(hashed((1, 64, h, w), 501, -2, 2))
    c2 = torch.relu(hashed((1, 64, h, w), 502, -1, 2))
    disp = hashed((1, 1, h, w), 503, 0.0005, 0.0025)
    feat = 100 * O.disp_features(disp)
    wzr = hashed((128, 177, 3, 3), 505, -0.05, 0.05)
    wq = hashed((64, 177, 3, 3), 506, -0.05, 0.05)
    init = hashed((P, 128), 504, -0.3, 0.3)
    initq = hashed((P, 64), 507, -0.3, 0.3)
    x = torch.cat([net, feat, c2], 1).double()
    pre = nhwc(F.conv2d(x, wzr.double(), None, padding=1)) + init.double()
    z_ref = torch.sigmoid(pre[:, :64])
    r_ref = torch.sigmoid(pre[:, 64:])
    rh_ref = r_ref * nhwc(net).double()
    src = [(64, 2, U), (49, 1, Dp), (64, 2, R)]
    pzr = ops.PackedConvS16(wzr, None, src, dev)
    pq = ops.PackedConvS16(wq, None, src, dev)
    net_s, c2_s = frag(net, h, w, U), frag(c2, h, w, R)
    d = disp.reshape(-1).to(dev)
    acc = lambda t: ops.s16_layout(t.to(dev), h, w, L.S16_ACC32)
    z, rh = ops.conv3x3_s16(pzr, [net_s, d, c2_s], h, w, L.EPI_GATES, aux=net_s, init=acc(init), log2s_out=U, log2s_aux=U)
    assert rel_l1(unacc(z, h, w, L.S16_F32X8), z_ref) < 1e-6 * f8
    assert rel_l1(unfrag(rh, h, w, U), rh_ref) < 1e-6 * f8
    rh4 = rh_ref.t().reshape(1, 64, h, w)
    xq = torch.cat([rh4, feat.double(), c2.double()], 1)
    q_ref = torch.tanh(nhwc(F.conv2d(xq, wq.double(), None, padding=1)) + initq.double())
    new_ref = (1 - z_ref) * nhwc(net).double() + z_ref * q_ref
    new = ops.conv3x3_s16(pq, [rh, d, c2_s], h, w, L.EPI_GRU, out=net_s, aux=net_s, aux2=z, init=acc(initq), log2s_out=U, log2s_aux=U)
    assert new.data_ptr() == net_s.data_ptr()                    # in place, as the loop runs it
    assert rel_l1(unfrag(new, h, w, U), new_ref) < 1e-6 * f8


@pytest.mark.parametrize("mt", [2, 4])
def test_conv_s16_fused_delta_head(dev, tile_mt, f8, mt):
    """EPI_DELTA: hid = relu(conv3x3(net, 64 -> 256)) projected onto the nine taps of the 256 -> 1 conv (core/update.py:68-71);
    cer_delta_sum_f32 then gives delta = 0.01 * conv3x3(hid, w2) (core/update.py:114)."""
    from cer_mvs_amd import _lib as L, ops
    ops.TILE_MT = mt
    h, w = 27, 38
    P = h * w
    net = torch.tanh(hashed((1, 64, h, w), 601, -2, 2))
    w1 = hashed((256, 64, 3, 3), 602, -0.08, 0.08)
    b1 = hashed((256,), 603, -0.1, 0.1)
    w2 = hashed((1, 256, 3, 3), 604, -0.05, 0.05)
    hid = F.relu(F.conv2d(net.double(), w1.double(), b1.double(), padding=1))
    delta_ref = 0.01 * (F.conv2d(hid, w2.double(), None, padding=1) + 0.25).reshape(-1)
    pc = ops.PackedConvS16(w1, b1, [(64, 2, L.S16_UNIT)], dev)
    proj = ops.delta_proj_pack_s16(w2, dev)
    T = ops.conv3x3_s16(pc, [frag(net, h, w, L.S16_UNIT)], h, w, L.EPI_DELTA, aux=proj)
    assert T.shape == (2, 9, P)
    disp0 = hashed((P,), 605, 0.0005, 0.002).to(dev)
    disp1, delta = ops.delta_sum(T, 0.25, disp0, h, w)
    assert rel_l1(delta.cpu().double(), delta_ref) < 1e-6 * f8
    assert torch.equal(disp1, disp0 + delta)


def test_lookup_encode_frag16_output(dev):
    """cer_lookup_encode_f32 with out_split = 2 writes exactly frag16(relu(conv1x1(lookup)))."""
    from cer_mvs_amd import _lib as L, ops
    h, w, D = 25, 40, 64
    P = h * w
    _, _, rs = ops.row_layout(D, 3)
    vol = hashed((P, rs), 701, -1, 1).to(dev)
    origin = hashed((P,), 702, 0.001, 0.0015).to(dev)
    disp = hashed((P,), 703, 0.0005, 0.002).to(dev)
    w0t = hashed((33, 64), 704, -0.2, 0.2).to(dev)
    b0 = hashed((64,), 705, -0.1, 0.1).to(dev)
    incre = 0.0025 / 64
    a = ops.lookup_encode(vol, origin, disp, w0t, b0, D, incre, 3, 5)
    b = ops.lookup_encode(vol, origin, disp, w0t, b0, D, incre, 3, 5, out_split=2, log2s=L.S16_RELU, img_w=w)
    assert torch.equal(b, ops.to_frag16(a, h, w, L.S16_RELU))


def test_conv_s16_dynamic_range(dev):
    """Small and large activations keep fp32-class accuracy relative to sum |x||w|; values beyond 65504 / scale saturate."""
    from cer_mvs_amd import _lib as L, ops
    h, w, cin, cout = 16, 32, 32, 64
    x = hashed((1, cin, h, w), 121) * torch.logspace(-3, 2, h * w).view(1, 1, h, w)       # 1e-3 .. 1e2 per pixel
    wt = hashed((cout, cin, 3, 3), 122, -0.2, 0.2)
    ref = nhwc(F.conv2d(x.double(), wt.double(), None, padding=1))
    mag = nhwc(F.conv2d(x.abs().double(), wt.abs().double(), None, padding=1))
    wsum = nhwc(F.conv2d(torch.ones(1, cin, h, w).double(), wt.abs().double(), None, padding=1))
    pc = ops.PackedConvS16(wt, None, [(cin, 2, L.S16_RELU)], dev)
    got = unacc(ops.conv3x3_s16(pc, [frag(x, h, w, L.S16_RELU)], h, w, L.EPI_LINEAR), h, w)
    # per operand: relative 2^-22 or absolute 2^-25 / 2^log2s, whichever is larger
    assert bool(((got - ref).abs() <= 2e-6 * mag + 2.0 ** (-24 - L.S16_RELU) * wsum).all())
    big = ops.to_frag16(torch.full((h * w, cin), 1e6, device=dev), h, w, L.S16_RELU)
    assert torch.isfinite(ops.conv3x3_s16(pc, [big], h, w, L.EPI_LINEAR)).all()


@pytest.mark.variants
def test_conv_s16_producer_consumer_form_matches(dev):
    """csrc/experimental/conv_s16pc.hip (variants/libcermvs_optin.so only since round 5: cer_conv3x3_s16_pc(1)) against the default kernels: z|r gates, GRU blend, ReLU conv and the fused
    delta head on a size with rim tiles, partial last tiles and several tiles per persistent block; the two forms differ only in where
    the hoisted term is added (<= 2e-6 relative), and the producer / consumer form reproduces itself bit for bit over 100 launches."""
    from cer_mvs_amd import _lib as L, ops
    lib = L.load()
    if not L.has_variant_forms():
        pytest.skip("the producer / consumer form is not in the product library: run with CER_MVS_LIB=.../variants/libcermvs_optin.so (tools/archive/r05/test_variants.sh)")
    h, w = 118, 150
    P = h * w
    U, R, Dp = L.S16_UNIT, L.S16_RELU, L.S16_DISP
    net = torch.tanh(hashed((1, 64, h, w), 701, -2, 2))
    c2 = torch.relu(hashed((1, 64, h, w), 702, -1, 2))
    disp = hashed((P,), 703, 0.0005, 0.0025).to(dev)
    src = [(64, 2, U), (49, 1, Dp), (64, 2, R)]
    pzr = ops.PackedConvS16(hashed((128, 177, 3, 3), 705, -0.05, 0.05), None, src, dev, corr_fp8=True)
    pq = ops.PackedConvS16(hashed((64, 177, 3, 3), 706, -0.05, 0.05), None, src, dev, corr_fp8=True)
    pr = ops.PackedConvS16(hashed((64, 64, 3, 3), 707, -0.05, 0.05), hashed((64,), 708, -0.1, 0.1), [(64, 2, R)], dev, corr_fp8=True)
    pd = ops.PackedConvS16(hashed((256, 64, 3, 3), 709, -0.08, 0.08), hashed((256,), 710, -0.1, 0.1), [(64, 2, U)], dev, corr_fp8=True)
    proj = ops.delta_proj_pack_s16(hashed((1, 256, 3, 3), 711, -0.05, 0.05), dev)
    net_s, c2_s = frag(net, h, w, U), frag(c2, h, w, R)
    init = ops.s16_layout(hashed((P, 128), 704, -0.3, 0.3).to(dev), h, w, L.S16_ACC32)
    initq = ops.s16_layout(hashed((P, 64), 712, -0.3, 0.3).to(dev), h, w, L.S16_ACC32)

    def run():
        z, rh = ops.conv3x3_s16(pzr, [net_s, disp, c2_s], h, w, L.EPI_GATES, aux=net_s, init=init, log2s_out=U, log2s_aux=U)
        new = ops.conv3x3_s16(pq, [rh, disp, c2_s], h, w, L.EPI_GRU, aux=net_s, aux2=z, init=initq, log2s_out=U, log2s_aux=U)
        rl = ops.conv3x3_s16(pr, [c2_s], h, w, L.EPI_RELU, log2s_out=R)
        T = ops.conv3x3_s16(pd, [net_s], h, w, L.EPI_DELTA, aux=proj)
        return [unacc(z, h, w, L.S16_F32X8), unfrag(rh, h, w, U), unfrag(new, h, w, U), unfrag(rl, h, w, R), T.cpu().double()], [z, rh, new, rl, T]
    prev = lib.cer_conv3x3_s16_pc(0)
    try:
        ref, _ = run()
        lib.cer_conv3x3_s16_pc(1)
        got, raw0 = run()
        raw0 = [t.clone() for t in raw0]
        for name, a_, b_ in zip(("z", "r*h", "h'", "relu", "taps"), got, ref):
            assert rel_l1(a_, b_) < 2e-6, (name, rel_l1(a_, b_))
        for _ in range(100):
            _, raw = run()
            assert all(torch.equal(x, y) for x, y in zip(raw, raw0))
    finally:
        lib.cer_conv3x3_s16_pc(prev)


def test_rn_may_share_c1s_buffer(dev, monkeypatch):
    """ADVICE r5 on update.ALIAS_RN_C1 (r * h written into the lookup output's buffer).  (1) Whole forwards are bit-identical with the
    two tensors aliased and apart.  (2) What the m-tile-major scratch tensors hold in their PADDING slots (slots past the image's last
    row / column; the epilogues store whole m-tiles, so relu(conv) of a virtual pixel does land there) never reaches a result: every
    consumer masks pixels outside the image while it stages (conv_s16.hip: `valid`), so a workspace whose tensors were pre-filled with
    garbage gives the same bits.  Ragged size: 9 x 21 pixels at feature resolution = 5 x 2 m-tiles of 2 x 16 slots, 131 of 320 padding."""
    from cer_mvs_amd import RAFT, ops, update
    from cer_mvs_amd.synthetic import fill_state_dict, synthetic_scene
    H, W, V = 36, 84, 3
    h, w = H // 4, W // 4
    cascade = [(64, 64, 3), (-1, 320, 3)]
    images, poses, intr, scale = synthetic_scene(H, W, V, seed=2)
    outs = {}
    for alias, poison in ((True, False), (False, False), (True, True)):
        monkeypatch.setattr(update, "ALIAS_RN_C1", alias)
        model = RAFT(cascade=cascade, test_mode=True, gru_precision="s16f8")
        model.load_state_dict(fill_state_dict(model.state_dict(), seed=5))
        model = model.to(dev).eval()
        ws = model.update_block.workspace(h, w, dev)
        assert (ws["rn"].data_ptr() == ws["c1"].data_ptr()) == alias and ws["c1"].shape[0] == ops.s16_pixels(h, w) > h * w
        if poison:
            for name in ("c1", "c2", "z", "rn"):
                ws[name].view(torch.int32).fill_(0x5BCD5BCD)          # f16 halves of 249.6 / as fp32 1.16e17: finite garbage everywhere
        with torch.no_grad():
            outs[(alias, poison)] = model(images.to(dev), poses.to(dev), intr.to(dev), scale=scale).clone()
    assert torch.equal(outs[(True, False)], outs[(False, False)])
    assert torch.equal(outs[(True, False)], outs[(True, True)])
    assert float(outs[(True, False)].abs().sum()) > 0
