"""UpdateBlock / ConvGRU (reference: core/update.py:9-25, 29-120) on the HIP conv kernels.

Same constructor arguments, parameter names (state_dict keys) and ``forward(net, inp, disp,
corr_frames, stage) -> (net, delta)`` as the reference.  Internally activations are channels-last
[h*w, C]; one iteration of the update block is 6 kernel launches:

  lookup+mean+1x1+ReLU -> 3x3 64->64 ReLU -> {z,r} 3x3 (sigmoid, r*net) -> q 3x3 (tanh, GRU blend)
  -> delta 3x3 64->256 ReLU -> delta 3x3 256->1 (+ disp update)

Two exact algebraic reductions are used on the fast path (``step``): the GRU's `inp` slice of
convz/convr/convq is constant over iterations and stages, so its contribution is convolved once
per forward and fed to the MFMA accumulators as their initial value (``hoist``); and the 49-channel
disparity encoder is generated inside the conv kernel instead of being materialised."""
import torch
import torch.nn as nn

from . import _lib as L
from . import ops


USE_PLANS = True         # replay recorded launches for GRU iterations 2..T (False: every iteration through the checked wrappers)
ALIAS_RN_C1 = __import__("os").environ.get("CER_ALIAS_RN_C1", "1") == "1"      # (A/B switch, see UpdateBlock.workspace)


class ConvGRU(nn.Module):
    """Parameter container + literal forward (reference: core/update.py:9-25)."""

    def __init__(self, kernel_z=3, kernel_r=3, kernel_q=3, h_planes=None, i_planes=None):
        super().__init__()
        if (kernel_z, kernel_r, kernel_q) != (3, 3, 3):
            raise NotImplementedError("ConvGRU: only the reference's default 3x3 kernels are built")
        self.do_checkpoint = False
        self.h_planes, self.i_planes = h_planes, i_planes
        self.convz = nn.Conv2d(h_planes + i_planes, h_planes, 3, padding=1)
        self.convr = nn.Conv2d(h_planes + i_planes, h_planes, 3, padding=1)
        self.convq = nn.Conv2d(h_planes + i_planes, h_planes, 3, padding=1)
        self._packed = None

    def _invalidate(self):
        self._packed = None

    def _apply(self, fn, *a, **k):
        self._packed = None
        return super()._apply(fn, *a, **k)

    def forward(self, net, *inputs):
        """net [1,Ch,h,w], inputs: NCHW tensors concatenated along channels (any split)."""
        x = torch.cat(inputs, dim=1)
        _, ch, h, w = net.shape
        cx = x.shape[1]
        cpad = (cx + 31) // 32 * 32
        dev = net.device
        if self._packed is None or self._packed[0] != (cx, dev):
            wzr = torch.cat([self.convz.weight, self.convr.weight], 0)
            bzr = torch.cat([self.convz.bias, self.convr.bias], 0)
            srcs = [(ch, 0), (cx, 0)]
            self._packed = ((cx, dev), ops.PackedConv3x3(wzr, bzr, srcs, dev), ops.PackedConv3x3(self.convq.weight, self.convq.bias, srcs, dev))
        _, pzr, pq = self._packed
        net_l = ops.nchw_to_nhwc(net[0].float().contiguous())
        xp = torch.zeros(h * w, cpad, device=dev, dtype=torch.float32)
        xp[:, :cx] = ops.nchw_to_nhwc(x[0].float().contiguous())
        pzr_run = _with_stride(pzr, [(ch, 0), (cpad, 0)])
        pq_run = _with_stride(pq, [(ch, 0), (cpad, 0)])
        z, rn = ops.conv3x3(pzr_run, [net_l, xp], h, w, L.EPI_GATES, aux=net_l)
        new = ops.conv3x3(pq_run, [rn, xp], h, w, L.EPI_GRU, aux=net_l, aux2=z)
        return ops.nhwc_to_nchw(new).view(1, ch, h, w)


class _Strided:
    """A PackedConv3x3 viewed with run-time source strides (padded tensors)."""

    def __init__(self, pc, sources):
        self.packed, self.packed_x, self.bias, self.cout, self.sources = pc.packed, pc.packed_x, pc.bias, pc.cout, sources
        self.packed_c = None              # (no kind-1 disparity source in the literal ConvGRU: nothing to collapse)


def _with_stride(pc, sources):
    return _Strided(pc, sources)


class UpdateBlock(nn.Module):
    def __init__(self, kernel_corr=3, dim0_corr=64, dim1_corr=64, dim_net=None, dim_inp=None, dim0_delta=256,
                 kernel0_delta=3, kernel1_delta=3, num_levels=3, radius=5, size_disp_enc=7, kernel0_vis=3, kernel1_vis=3,
                 share_corr=True, share_gru=True, share_delta=False, aggregation=("mean",), cascade=None):
        super().__init__()
        if (kernel_corr, kernel0_delta, kernel1_delta, size_disp_enc) != (3, 3, 3, 7):
            raise NotImplementedError("UpdateBlock: only the reference's default kernel sizes are built")
        if (dim0_corr, dim1_corr, dim_net, dim_inp, dim0_delta) != (64, 64, 64, 64, 256):
            raise NotImplementedError("UpdateBlock: only the reference's default widths (64/64/64/64/256) are built")
        self.num_levels, self.radius, self.size_disp_enc = num_levels, radius, size_disp_enc
        self.share_corr, self.share_gru, self.share_delta = share_corr, share_gru, share_delta
        self.aggregation = list(aggregation)
        self.cascade = cascade
        self.dim_net, self.dim_inp = dim_net, dim_inp
        cor_planes = len(self.aggregation) * num_levels * (2 * radius + 1)
        n_cascade = len(cascade)
        for i in (range(n_cascade) if not share_corr else [""]):
            setattr(self, f"corr_encoder{i}", nn.Sequential(
                nn.Conv2d(cor_planes, dim0_corr, 1, padding=0), nn.ReLU(inplace=True),
                nn.Conv2d(dim0_corr, dim1_corr, 3, padding=1), nn.ReLU(inplace=True)))
        for i in (range(n_cascade) if not share_delta else [""]):
            setattr(self, f"delta{i}", nn.Sequential(
                nn.Conv2d(dim_net, dim0_delta, 3, padding=1), nn.ReLU(inplace=True), nn.Conv2d(dim0_delta, 1, 3, padding=1)))
        i_planes = dim_inp + dim1_corr + size_disp_enc ** 2
        for i in (range(n_cascade) if not share_gru else [""]):
            setattr(self, f"gru{i}", ConvGRU(h_planes=dim_net, i_planes=i_planes))
        # arithmetic of the 3x3 convolutions: "s16" (split-f16 MFMA, one accumulator, barrier-light kernels of conv_s16.hip: the
        # fast path), "f16x3" (round-1 split-f16 kernels, two accumulators) or "fp32" (exact fp32 MFMA)
        self.conv_mode = "s16"
        self._packed = {}
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.refresh_weights())

    # ------------------------------------------------------------------ weight packing
    def refresh_weights(self):
        """Drop packed weights (call after mutating parameters)."""
        self._packed = {}
        for m in self.modules():
            if isinstance(m, ConvGRU):
                m._invalidate()

    def _apply(self, fn, *a, **k):
        self._packed = {}
        return super()._apply(fn, *a, **k)

    def _names(self, stage):
        return (f"corr_encoder{stage if not self.share_corr else ''}", f"gru{stage if not self.share_gru else ''}",
                f"delta{stage if not self.share_delta else ''}")

    def packed(self, stage, device):
        """Packed weights for ``stage`` on ``device`` (built once, cached)."""
        f8 = 6 if self.corr_fp8 == 6 else bool(self.corr_fp8)      # (False: three f16 terms, True: fp8 corrections, 6: FP6 corrections)
        key = (stage, str(device), self.conv_mode, f8)  # (the dict's contents depend on the arithmetic mode: s16 packs exist only for "s16")
        if key in self._packed:
            return self._packed[key]
        cn, gn, dn = self._names(stage)
        ce, gru, de = getattr(self, cn), getattr(self, gn), getattr(self, dn)
        dn_, di = self.dim_net, self.dim_inp
        f32 = lambda t: t.detach().to(device, torch.float32).contiguous()
        p = {}
        p["w0t"] = f32(ce[0].weight[:, :, 0, 0].t())                       # [K,64]
        p["b0"] = f32(ce[0].bias)
        p["corr2"] = ops.PackedConv3x3(ce[2].weight, ce[2].bias, [(64, 0)], device)
        wzr = torch.cat([gru.convz.weight, gru.convr.weight], 0).detach()
        bzr = torch.cat([gru.convz.bias, gru.convr.bias], 0).detach()
        wq, bq = gru.convq.weight.detach(), gru.convq.bias.detach()
        # input channel order of the GRU convs (core/update.py:18-19,112): net | inp | disp49 | corr
        s_net, s_inp = slice(0, dn_), slice(dn_, dn_ + di)
        rest = list(range(0, dn_)) + list(range(dn_ + di, wzr.shape[1]))
        full_src = [(dn_, 0), (di, 0), (49, 1), (64, 0)]
        rest_src = [(dn_, 0), (49, 1), (64, 0)]
        p["zr_full"] = ops.PackedConv3x3(wzr, bzr, full_src, device)
        p["q_full"] = ops.PackedConv3x3(wq, bq, full_src, device)
        p["zr_rest"] = ops.PackedConv3x3(wzr[:, rest], None, rest_src, device)
        p["q_rest"] = ops.PackedConv3x3(wq[:, rest], None, rest_src, device)
        p["zr_inp"] = ops.PackedConv3x3(wzr[:, s_inp], bzr, [(di, 0)], device)
        p["q_inp"] = ops.PackedConv3x3(wq[:, s_inp], bq, [(di, 0)], device)
        p["d1"] = ops.PackedConv3x3(de[0].weight, de[0].bias, [(dn_, 0)], device)
        if self.conv_mode == "s16":
            # s16 fast path (csrc/conv_s16.hip): every loop tensor lives in HBM in the split16 layout with a per-class power-of-two
            # scale (hidden state and r*h: |x| <= 1; ReLU outputs; generated disparity features).  PackedConvS16 raises when the
            # weights of one conv do not fit a shared scale (cer_conv3x3_s16_scale): callers of ``packed`` see that error and may
            # switch ``conv_mode`` to "f16x3" (RAFT does, with a warning).
            U, R, Dp = L.S16_UNIT, L.S16_RELU, L.S16_DISP
            p["s_corr2"] = ops.PackedConvS16(ce[2].weight, ce[2].bias, [(64, 2, R)], device, corr_fp8=f8)
            p["s_zr"] = ops.PackedConvS16(wzr[:, rest], None, [(dn_, 2, U), (49, 1, Dp), (64, 2, R)], device, corr_fp8=f8)
            p["s_q"] = ops.PackedConvS16(wq[:, rest], None, [(dn_, 2, U), (49, 1, Dp), (64, 2, R)], device, corr_fp8=f8)
            p["s_zr_inp"] = ops.PackedConvS16(wzr[:, s_inp], bzr, [(di, 2, R)], device, corr_fp8=f8)
            p["s_q_inp"] = ops.PackedConvS16(wq[:, s_inp], bq, [(di, 2, R)], device, corr_fp8=f8)
            p["s_d1"] = ops.PackedConvS16(de[0].weight, de[0].bias, [(dn_, 2, U)], device, corr_fp8=f8)
            p["s_d2proj"] = ops.delta_proj_pack_s16(de[2].weight, device)
        p["d2w"] = f32(de[2].weight[0].permute(1, 2, 0).reshape(9, -1))    # [tap, C]
        p["d2proj"] = ops.delta_proj_pack(de[2].weight, device)
        p["d2b"] = float(de[2].bias.detach().float().cpu()[0])
        self._packed[key] = p
        return p

    # ------------------------------------------------------------------ fast path (channels-last)
    def hoist(self, inp_l, h, w, stage=0):
        """Contribution of the constant `inp` slice (+ biases) to the z|r and q pre-activations."""
        p = self.packed(stage, inp_l.device)
        if self.conv_mode == "s16":     # (the results are in the acc32 layout: they seed the accumulators of the loop's convs)
            inp_s = ops.to_frag16(inp_l, h, w, L.S16_RELU)
            if self.CHECK_OVERFLOW:
                ops.scan_overflow(inp_s)
            return (ops.conv3x3_s16(p["s_zr_inp"], [inp_s], h, w, L.EPI_LINEAR), ops.conv3x3_s16(p["s_q_inp"], [inp_s], h, w, L.EPI_LINEAR))
        return (ops.conv3x3(p["zr_inp"], [inp_l], h, w, L.EPI_LINEAR, mode=self.conv_mode), ops.conv3x3(p["q_inp"], [inp_l], h, w, L.EPI_LINEAR, mode=self.conv_mode))

    def hoist_all(self, inp_l, h, w, n_stages):
        """``hoist`` for every cascade stage: one result shared by all stages when the GRU weights are shared (the reference's
        default, core/update.py:47), one per stage otherwise (gru{stage} has its own `inp` weights and biases)."""
        if self.share_gru:
            return [self.hoist(inp_l, h, w, 0)] * n_stages
        return [self.hoist(inp_l, h, w, s) for s in range(n_stages)]

    # f16x3 path: the loop's activations (hidden state, corr features, r*h) live in HBM in the "split32" layout - hi|lo f16
    # pairs in the fp32 slots (cer_mvs.h) - written by the producers' epilogues, so that every conv stages its tensor sources
    # with plain 16-byte copies instead of re-splitting them (staging was ~10 % of the conv time, VALU-bound).  The hidden
    # state is therefore carried at 2^-22 relative resolution (the resolution the f16x3 products see anyway).
    SPLIT_ACTS = True
    # s16 path: the two correction terms of the split-f16 product (2^-11 of the main term) of every tensor source on the block-scaled
    # fp8 matrix instruction (twice the f16 rate): gru_precision="s16f8".  End to end 4e-6 instead of 2e-7 relative L1 from fp32.
    corr_fp8 = False
    CHECK_OVERFLOW = True       # s16 path: scan the ReLU-class activation tensors for saturation once per stage (ops.check_overflow reads the flag)

    def split_acts(self):
        return self.SPLIT_ACTS and self.conv_mode == "f16x3"

    def prepare_net(self, net_l, h, w):
        """Hidden state [h*w,64] fp32 -> the layout ``step`` keeps it in (frag16 on the s16 path, split32 on the f16x3 path)."""
        if self.conv_mode == "s16":
            return ops.to_frag16(net_l, h, w, L.S16_UNIT)
        return ops.split32(net_l) if self.split_acts() else net_l

    def restore_net(self, net_l, h, w):
        """Inverse of ``prepare_net``: the hidden state as plain fp32 [h*w,64]."""
        if self.conv_mode == "s16":
            return ops.from_frag16(net_l, h, w, L.S16_UNIT)
        return ops.split32(net_l, inverse=True) if self.split_acts() else net_l

    # s16 path: the disparity update of iteration i (the 18-tap gather of the fused delta head's tap planes, ``ops.delta_sum``) rides on the
    # lookup launch of iteration i + 1 (csrc/lookup.hip, round 5): one launch less per iteration; the last iteration of a stage ends with
    # the stand-alone kernel.  Off where something reads ``disp`` between two iterations (the row-slab exchange: ``run(after=...)``).
    FUSE_DELTA = __import__("os").environ.get("CER_FUSE_DELTA", "1") == "1"      # (A/B switch)

    def step(self, vol, origin, net_l, disp, hoisted, stage, h, w, D, incre, ws, delta="now"):
        """One GRU iteration on the folded volume; updates ``net_l`` [P,64] (see ``prepare_net``) and ``disp`` [P] in place.
        ``ws``: dict of scratch tensors (c1, c2, z, rn, hid) reused across iterations.  ``delta`` (s16 path): "now" - the disparity update
        is a launch of its own at the end of the step; "defer" - it is left pending in ``ws["T"]`` for the next step's lookup;
        "apply+defer" - this step's lookup first applies the pending update of the previous step, and leaves its own pending."""
        p = self.packed(stage, net_l.device)
        hzr, hq = hoisted
        if self.conv_mode == "s16":
            U, R = L.S16_UNIT, L.S16_RELU
            ops.lookup_encode(vol, origin, disp, p["w0t"], p["b0"], D, incre, self.num_levels, self.radius, out=ws["c1"], out_split=2, log2s=R,
                              img_w=w, delta=(ws["T"], p["d2b"]) if delta == "apply+defer" else None)
            ops.conv3x3_s16(p["s_corr2"], [ws["c1"]], h, w, L.EPI_RELU, out=ws["c2"], out_split=True, log2s_out=R)
            ops.conv3x3_s16(p["s_zr"], [net_l, disp, ws["c2"]], h, w, L.EPI_GATES, out=ws["z"], out2=ws["rn"], aux=net_l, init=hzr,
                            log2s_out=U, log2s_aux=U)
            ops.conv3x3_s16(p["s_q"], [ws["rn"], disp, ws["c2"]], h, w, L.EPI_GRU, out=net_l, aux=net_l, aux2=ws["z"], init=hq,
                            log2s_out=U, log2s_aux=U)
            ops.conv3x3_s16(p["s_d1"], [net_l], h, w, L.EPI_DELTA, out=ws["T"], aux=p["s_d2proj"])
            if delta == "now":
                ops.delta_sum(ws["T"], p["d2b"], disp, h, w, disp_out=disp, want_delta=False)
            return
        if self.split_acts():
            ops.lookup_encode(vol, origin, disp, p["w0t"], p["b0"], D, incre, self.num_levels, self.radius, out=ws["c1"], out_split=True)
            ops.conv3x3(p["corr2"], [ws["c1"]], h, w, L.EPI_RELU, out=ws["c2"], kinds=[3], out_split=True)
            ops.conv3x3(p["zr_rest"], [net_l, disp, ws["c2"]], h, w, L.EPI_GATES, out=ws["z"], out2=ws["rn"], aux=net_l, init=hzr,
                        kinds=[3, 1, 3], out_split=True, aux_split=True)
            ops.conv3x3(p["q_rest"], [ws["rn"], disp, ws["c2"]], h, w, L.EPI_GRU, out=net_l, aux=net_l, aux2=ws["z"], init=hq,
                        kinds=[3, 1, 3], out_split=True, aux_split=True)
            ops.conv3x3(p["d1"], [net_l], h, w, L.EPI_DELTA, mode="f16x3", out=ws["T"], aux=p["d2proj"], kinds=[3])
            ops.delta_sum(ws["T"], p["d2b"], disp, h, w, disp_out=disp, want_delta=False)
            return
        ops.lookup_encode(vol, origin, disp, p["w0t"], p["b0"], D, incre, self.num_levels, self.radius, out=ws["c1"])
        ops.conv3x3(p["corr2"], [ws["c1"]], h, w, L.EPI_RELU, out=ws["c2"])
        ops.conv3x3(p["zr_rest"], [net_l, disp, ws["c2"]], h, w, L.EPI_GATES, out=ws["z"], out2=ws["rn"], aux=net_l, init=hzr)
        ops.conv3x3(p["q_rest"], [ws["rn"], disp, ws["c2"]], h, w, L.EPI_GRU, out=net_l, aux=net_l, aux2=ws["z"], init=hq)
        if self.conv_mode == "f16x3":       # fused delta head: the 256-channel hidden map never reaches HBM
            ops.conv3x3(p["d1"], [net_l], h, w, L.EPI_DELTA, mode="f16x3", out=ws["T"], aux=p["d2proj"])
            ops.delta_sum(ws["T"], p["d2b"], disp, h, w, disp_out=disp, want_delta=False)
        else:
            ops.conv3x3(p["d1"], [net_l], h, w, L.EPI_RELU, mode=self.conv_mode, out=ws["hid"])
            ops.delta_tail(ws["hid"], p["d2w"], p["d2b"], disp, h, w, disp_out=disp, want_delta=False)

    def run(self, iters, vol, origin, net_l, disp, hoisted, stage, h, w, D, incre, ws, after=None):
        """``iters`` GRU iterations on fixed buffers: the first executes through the checked wrappers while its raw launches
        are recorded (``_lib.LaunchPlan``), the rest replay them.  ``after(i)``: called after every iteration (the slab
        exchange of the sharded forward)."""
        plan = None
        p = self.packed(stage, net_l.device)    # weight packing (host work) must not end up in the recorded plan
        fuse = self.FUSE_DELTA and self.conv_mode == "s16" and after is None and iters >= 2
        for i in range(iters):
            if fuse and i == 0:                 # nothing pending yet: through the checked wrappers, its update left pending
                self.step(vol, origin, net_l, disp, hoisted, stage, h, w, D, incre, ws, delta="defer")
                continue
            mode = "apply+defer" if fuse else "now"
            if not USE_PLANS:
                self.step(vol, origin, net_l, disp, hoisted, stage, h, w, D, incre, ws, delta=mode)
            elif plan is None:
                plan = L.LaunchPlan(keep=(vol, origin, net_l, disp, hoisted, ws))
                with L.recording(plan):
                    self.step(vol, origin, net_l, disp, hoisted, stage, h, w, D, incre, ws, delta=mode)
            else:
                plan.replay()
            if after is not None:
                after(i)
        if fuse:                                # the last iteration's update
            ops.delta_sum(ws["T"], p["d2b"], disp, h, w, disp_out=disp, want_delta=False)
        if self.conv_mode == "s16" and iters > 0 and self.CHECK_OVERFLOW:
            # the s16 layouts clamp ReLU-class activations beyond 4094 (65504 / 2^4): saturation must not be silent - scan what the
            # last iteration left in memory (2 x ~8 us; the delta head's hidden map is checked inside its kernel)
            # (c1 is checked by the lookup kernel itself since round 5: its buffer holds r * h by now - ALIAS_RN_C1)
            ops.scan_overflow(ws["c2"])

    def workspace(self, h, w, device):
        """Scratch tensors of the loop for an h x w image (s16 path: m-tile-major layouts over whole m-tiles, ops.s16_pixels)."""
        P = h * w
        if self.conv_mode == "s16":
            # persistent per (size, device): allocated (and zero-filled) once - 4 x 30 MB of memsets per forward otherwise; every data
            # pixel is overwritten by the first iteration of a forward, the padding slots are never consumed (see below)
            key = (h, w, str(device))
            cache = self.__dict__.setdefault("_ws_cache", {})
            if key not in cache:
                while len(cache) >= 8:            # (row slabs of different heights share a process in the simulated-rank tests)
                    cache.pop(next(iter(cache)))
                z = lambda c: torch.zeros(ops.s16_pixels(h, w), c, device=device, dtype=torch.float32)
                cache[key] = {"c1": z(64), "c2": z(64), "z": z(64), "T": torch.empty(2, 9, P, device=device, dtype=torch.float32)}
                # r*h is written by the z|r launch, after the only reader of c1 (the corr2 launch) has run, and read by the q launch,
                # before the next lookup writes c1 again: with ALIAS_RN_C1 the two tensors ARE ONE 30 MB buffer (ws["rn"] is ws["c1"];
                # round 5: the iteration's working set - 283 MB at 296 x 400 - sits just above the 256 MB Infinity Cache).  Legal because
                # nothing reads c1 behind the z|r launch of an iteration.  (The padding slots of these tensors need NOT stay zero: the
                # epilogues store whole m-tiles, and every consumer masks pixels outside the image while it stages;
                # tests/test_conv_s16_gpu.py::test_rn_may_share_c1s_buffer runs a forward on a garbage-filled workspace.)  Tools that
                # want two independent buffers set CER_ALIAS_RN_C1=0.
                cache[key]["rn"] = cache[key]["c1"] if ALIAS_RN_C1 else z(64)
            return cache[key]
        e = lambda c: torch.empty(P, c, device=device, dtype=torch.float32)
        return {"c1": e(64), "c2": e(64), "z": e(64), "rn": e(64), "hid": e(256),
                "T": torch.empty(2, 9, P, device=device, dtype=torch.float32)}

    # ------------------------------------------------------------------ literal API
    def disp_encoder(self, disp):
        """[B,1,h,w] -> [B,49,h,w] (reference: core/update.py:80-85); not used by the HIP path, which
        generates these channels inside the conv kernel."""
        batch, _, ht, wd = disp.shape
        k = self.size_disp_enc
        u = torch.nn.functional.unfold(disp, [k, k], padding=k // 2).view(batch, k * k, ht, wd)
        return u - disp.view(batch, 1, ht, wd)

    def forward(self, net, inp, disp, corr_frames, stage, parts=None):
        """net, inp [B,num,64,h,w]; disp [B,1,h,w]; corr_frames [B,V,33,h,w] -> (net [B,num,64,h,w], delta [B,num,h,w]).
        ``parts`` (extension): the view aggregates [33,h,w] in the order mean, max, std, already reduced over all views - the
        view-sharded literal forward passes them (dist.aggregate_views) instead of ``corr_frames``."""
        if not net.is_cuda:
            raise RuntimeError("net must be a CUDA tensor")
        batch, num, ch, ht, wd = net.shape
        if batch * num != 1:
            raise RuntimeError("UpdateBlock.forward: batch*num must be 1")
        P = ht * wd
        dev = net.device
        p = self.packed(stage, dev)
        net_l = ops.nchw_to_nhwc(net.reshape(ch, P).float().contiguous())
        inp_l = ops.nchw_to_nhwc(inp.reshape(-1, P).float().contiguous())
        disp_l = disp.reshape(P).float().contiguous()
        if parts is not None:
            agg = torch.stack([t.float() for t in parts], dim=1).reshape(1, -1, P).contiguous()
            c1 = ops.corr_encode(agg, p["w0t"], p["b0"])
            feats = None
        else:
            feats = corr_frames[0].float()
            parts = []
        if feats is None:
            pass
        elif "mean" in self.aggregation and len(self.aggregation) == 1:
            c1 = ops.corr_encode(feats.reshape(feats.shape[0], -1, P).contiguous(), p["w0t"], p["b0"])
        else:
            if "mean" in self.aggregation:
                parts.append(torch.mean(feats, dim=0))
            if "max" in self.aggregation:
                parts.append(torch.max(feats, dim=0).values)
            if "std" in self.aggregation:
                parts.append(torch.std(feats, dim=0))
            # stack(dim=2).view(...) in the reference interleaves [part][channel]: channel-major then part
            agg = torch.stack(parts, dim=1).reshape(1, -1, P).contiguous()
            c1 = ops.corr_encode(agg, p["w0t"], p["b0"])
        mode = "f16x3" if self.conv_mode == "s16" else self.conv_mode      # the literal API runs on the general round-1 kernels
        c2 = ops.conv3x3(p["corr2"], [c1], ht, wd, L.EPI_RELU, mode=mode)
        z, rn = ops.conv3x3(p["zr_full"], [net_l, inp_l, disp_l, c2], ht, wd, L.EPI_GATES, mode=mode, aux=net_l)
        new = ops.conv3x3(p["q_full"], [rn, inp_l, disp_l, c2], ht, wd, L.EPI_GRU, mode=mode, aux=net_l, aux2=z)
        if mode == "f16x3":
            T = ops.conv3x3(p["d1"], [new], ht, wd, L.EPI_DELTA, mode="f16x3", aux=p["d2proj"])
            _, delta = ops.delta_sum(T, p["d2b"], disp_l, ht, wd)
        else:
            hid = ops.conv3x3(p["d1"], [new], ht, wd, L.EPI_RELU, mode=mode)
            _, delta = ops.delta_tail(hid, p["d2w"], p["d2b"], disp_l, ht, wd)
        net_out = ops.nhwc_to_nchw(new).view(batch, num, ch, ht, wd)
        return net_out, delta.view(batch, num, ht, wd)
