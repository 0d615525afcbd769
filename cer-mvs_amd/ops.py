"""Python faces of the HIP kernels: shape bookkeeping + output allocation around include/cer_mvs.h.
All tensors are CUDA float32; every call enqueues on torch's current stream."""
import ctypes

import torch

from . import _lib as L


def row_layout(D, num_levels, compact=False):
    """Volume row layout [level0 | level1 | ...] -> (offsets, lengths, row_stride multiple of 4).  ``compact`` (round 5, the folded
    volume of RAFT.forward): the row holds level 0 only - the lookup kernels form the pooled levels on the fly, bit-identically -
    and the stride is D rounded up to 4."""
    offs, lens, n, off = [], [], D, 0
    for _ in range(num_levels):
        offs.append(off)
        lens.append(n)
        off += n
        n //= 2
    return offs, lens, ((D if compact else off) + 3) // 4 * 4


def _level0_only(vol, D, num_levels, level0_only):
    """The row form of a volume for the lookup entry points (an explicit argument of the C ABI since 1060).  Given, it is
    used; else the mark the builders of this module leave on the volumes they return (``vol.level0_only``) - a view of a volume
    loses it; else the stride decides where it can: rows shorter than the whole pyramid hold level 0 only, and a stride that fits
    both forms (D <= 5) is refused instead of guessed."""
    if level0_only is not None:
        return int(bool(level0_only))
    mark = getattr(vol, "level0_only", None)
    if mark is not None:
        return int(bool(mark))
    offs, lens, rs_full = row_layout(D, num_levels)
    rs = vol.shape[-1]
    if rs < offs[-1] + lens[-1]:
        return 1
    if num_levels > 1 and rs == row_layout(D, num_levels, compact=True)[2]:
        raise ValueError(f"lookup: a row stride of {rs} floats fits both row forms at D = {D}, {num_levels} levels: pass level0_only")
    return 0


def alt_corr_forward(fmap1, fmap2, coords, radius):
    B, H1, W1, C = fmap1.shape
    _, H2, W2, _ = fmap2.shape
    N = coords.shape[1]
    rd = 2 * radius + 1
    corr = torch.empty(B, N, rd * rd, H1, W1, device=fmap1.device, dtype=torch.float32)
    L.check(L.load().cer_alt_corr_forward_f32(L.dev_ptr(fmap1, "fmap1"), L.dev_ptr(fmap2, "fmap2"), L.dev_ptr(coords, "coords"),
                                              L.dev_ptr(corr, "corr"), B, N, H1, W1, H2, W2, C, radius, L.cur_stream()),
            "alt_corr_forward")
    return corr


DETERMINISTIC_BACKWARD = True      # fmap2 gradient by a sorted segmented reduction (False: float atomics, like the reference)


def alt_corr_backward(fmap1, fmap2, coords, corr_grad, radius, deterministic=None):
    """-> (fmap1_grad, fmap2_grad, coords_grad = 0) (correlation.cpp:36-48).  ``deterministic`` (default
    ``DETERMINISTIC_BACKWARD``): fmap2_grad without atomics - tuples kernel, stable sort of the texel keys (torch), segmented
    reduction kernel: bit-identical from run to run."""
    B, H1, W1, C = fmap1.shape
    _, H2, W2, _ = fmap2.shape
    N = coords.shape[1]
    if DETERMINISTIC_BACKWARD if deterministic is None else deterministic:
        lib, dev = L.load(), fmap1.device
        g1 = torch.empty_like(fmap1)
        gc = torch.zeros_like(coords)
        L.check(lib.cer_alt_corr_backward_f32(L.dev_ptr(fmap1, "fmap1"), L.dev_ptr(fmap2, "fmap2"), L.dev_ptr(coords, "coords"),
                                              L.dev_ptr(corr_grad, "corr_grad"), L.dev_ptr(g1, "g1"), None, None, B, N, H1, W1, H2, W2, C,
                                              radius, L.cur_stream()), "alt_corr_backward[f1]")
        # one batch element (= one source view under DirectCorr) at a time: its tuples (key, coefficient, source pixel: ~32 B each with
        # the sort's temporaries) are N*H1*W1*(2r+2)^2 - 1/B of the transient memory of sorting all views at once (~1.7 GB per stage
        # at 10 views x 64 hypotheses x 128x160), and B small sorts instead of one large one; keys are partitioned by b anyway
        n = N * H1 * W1 * (2 * radius + 2) ** 2
        keys = torch.empty(n, device=dev, dtype=torch.int64)
        coef = torch.empty(n, device=dev, dtype=torch.float32)
        src = torch.empty(n, device=dev, dtype=torch.int32)
        T = H2 * W2
        bounds = torch.arange(T + 1, device=dev, dtype=torch.int64)
        g2 = torch.empty_like(fmap2)
        for b in range(B):
            L.check(lib.cer_alt_corr_bwd_tuples_f32(L.dev_ptr(coords[b:b + 1], "coords"), L.dev_ptr(corr_grad[b:b + 1], "corr_grad"),
                                                    L.dev_ptr(keys, "keys", torch.int64), L.dev_ptr(coef, "coef"), L.dev_ptr(src, "src", torch.int32),
                                                    1, N, H1, W1, H2, W2, radius, L.cur_stream()), "alt_corr_bwd_tuples")
            skeys, order = torch.sort(keys, stable=True)
            seg = torch.searchsorted(skeys, bounds).contiguous()
            L.check(lib.cer_alt_corr_bwd_reduce_f32(L.dev_ptr(fmap1[b:b + 1], "fmap1"), L.dev_ptr(order.contiguous(), "order", torch.int64),
                                                    L.dev_ptr(coef, "coef"), L.dev_ptr(src, "src", torch.int32), L.dev_ptr(seg, "seg", torch.int64),
                                                    L.dev_ptr(g2[b:b + 1], "g2"), T, C, L.cur_stream()), "alt_corr_bwd_reduce")
        return g1, g2, gc
    g1 = torch.empty_like(fmap1)
    g2 = torch.empty_like(fmap2)
    gc = torch.empty_like(coords)
    L.check(L.load().cer_alt_corr_backward_f32(L.dev_ptr(fmap1, "fmap1"), L.dev_ptr(fmap2, "fmap2"), L.dev_ptr(coords, "coords"),
                                               L.dev_ptr(corr_grad, "corr_grad"), L.dev_ptr(g1, "g1"), L.dev_ptr(g2, "g2"),
                                               L.dev_ptr(gc, "gc"), B, N, H1, W1, H2, W2, C, radius, L.cur_stream()),
            "alt_corr_backward")
    return g1, g2, gc


_OVERFLOW = {}


def overflow_flag(device):
    """Sticky device int32 or-ed by kernels that had to saturate an operand (feature split beyond +-1023, ...): read it with
    ``check_overflow`` when convenient (reading synchronises)."""
    key = str(device)
    f = _OVERFLOW.get(key)
    if f is None:
        f = torch.zeros(1, device=device, dtype=torch.int32)
        _OVERFLOW[key] = f
        L.check(L.load().cer_overflow_flag(L.dev_ptr(f, "flag", torch.int32)), "overflow_flag")     # in-kernel checks report here too
    return f


def scan_overflow(t, bit=4):
    """Or ``bit`` into the device's overflow flag if any half of the split-f16 tensor ``t`` (frag16 activations, split feature
    rows; any dtype, the bytes are scanned) is saturated or not finite.  ~8 us per 30 MB."""
    nbytes = t.numel() * t.element_size()
    L.check(L.load().cer_f16_scan_overflow(ctypes.c_void_p(t.data_ptr()), nbytes, L.dev_ptr(overflow_flag(t.device), "flag", torch.int32),
                                           int(bit), L.cur_stream()), "scan_overflow")


def check_overflow(device, reset=True):
    """Non-zero if any kernel since the last reset saturated an operand on ``device`` (one device->host read): bit 1 = cost-volume
    feature rows, 2 = hidden map of the delta head, 4 = an update-block activation tensor (``scan_overflow``)."""
    f = overflow_flag(device)
    hit = int(f.item())
    if reset and hit:
        f.zero_()
    return hit


_OVERFLOW_SNAP = {}


def overflow_snapshot(device):
    """Enqueue an asynchronous copy of the flag into pinned host memory (end of a forward); ``overflow_poll`` reads it later
    without synchronising."""
    key = str(device)
    snap = _OVERFLOW_SNAP.get(key)
    if snap is None:
        snap = [torch.zeros(1, dtype=torch.int32).pin_memory(), torch.cuda.Event()]
        _OVERFLOW_SNAP[key] = snap
    snap[0].copy_(overflow_flag(device), non_blocking=True)
    snap[1].record()


def overflow_poll(device):
    """Bits of the last completed ``overflow_snapshot`` (the device flag is cleared on a hit), 0 if clean, None if the
    snapshot has not completed yet or none was taken - never blocks."""
    snap = _OVERFLOW_SNAP.get(str(device))
    if snap is None or not snap[1].query():
        return None
    bits = int(snap[0][0])
    if bits:
        overflow_flag(device).zero_()
        snap[0].zero_()
    return bits


def feat_split(x, out=None):
    """fp32 rows [texels, 64] (one block: the reference map) or [blocks, texels, 64] (source views) -> the split-f16 operand
    planes of ``cost_build(..., split=...)`` / cer_cost_lines_f32 (cer_mvs.h: per block 8 planes [texels][16], same bytes)."""
    if x.shape[-1] != 64 or x.dim() not in (2, 3):
        raise RuntimeError("feat_split: [texels, 64] or [blocks, texels, 64]")
    blocks, texels = (1, x.shape[0]) if x.dim() == 2 else (x.shape[0], x.shape[1])
    if out is None:
        out = torch.empty(x.shape[:-1] + (128,), device=x.device, dtype=torch.float16)
    elif out.numel() != 2 * x.numel() or out.device != x.device:
        raise RuntimeError(f"feat_split: out must hold {2 * x.numel()} halves on {x.device} (got {out.numel()} on {out.device})")
    L.check(L.load().cer_feat_split_f16(L.dev_ptr(x, "x"), L.dev_ptr(out, "out", torch.float16), blocks, texels, 64,
                                        L.dev_ptr(overflow_flag(x.device), "flag", torch.int32), L.cur_stream()), "feat_split")
    return out


_LINES_WS = {}


def release_lines_workspace(device, stream=None):
    """Forget the workspace of ``stream`` (a torch.cuda.Stream) - or of every stream of ``device`` - so that its memory returns to the
    caching allocator (every inference() call creates new streams; their entries used to accumulate)."""
    dev = str(device)
    for key in [k for k in _LINES_WS if k[0] == dev and (stream is None or k[1] == stream.cuda_stream)]:
        del _LINES_WS[key]


def lines_workspace(V, h1, w1, D, device):
    """The (persistent, per device and stream) workspace of the epipolar-line-tile cost volume: partial volumes + tile parameters."""
    return _lines_workspace(V, h1, w1, D, device)


def _lines_workspace(V, h1, w1, D, device):
    need = int(L.load().cer_cost_lines_workspace(V, h1, w1, D))
    key = (str(device), torch.cuda.current_stream(device).cuda_stream)       # per stream: forwards on different streams overlap
    ws = _LINES_WS.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, device=device, dtype=torch.uint8)
        _LINES_WS[key] = ws
    return ws


def cost_build(fmap1, fmap2, Pij, disp_in, D, incre, shift, h1, w1, num_levels, fold, vol=None, accumulate=False, src_hw=None, y0=0,
               pyramid_scale=None, split=None, compact=False, two_term=False):
    """fmap1 [P,C], fmap2 [V,(h2+4)*(w2+4),C] (NHWC, pre-scaled, 2-texel zero border), Pij [V,4,4], disp_in [P].
    (h1, w1): reference grid of this call, first image row ``y0`` (row slabs); ``src_hw``: source-map size (default h1, w1).
    Returns (vol [V,P,rs] or [P,rs], origin [P]).  Level 0 only; call ``pyramid`` next - unless ``pyramid_scale`` is given
    (fold, no accumulate, D <= 64): then the kernel's epilogue scales level 0 and writes the pooled levels itself.
    Fold modes at C = 64, D <= 64 run on the epipolar-line-tile kernel (cer_cost_lines_f32) unless ``cer_cost_build_algo(1)``
    selects the walk; ``split`` = (feat_split(fmap1), feat_split(fmap2)) if the caller already has them (they are the same for
    every stage of a forward), else they are made here; a third element ``slots`` (int32 [V], device) says that view v's split
    rows are block slots[v] of the second element (any leading shape; the sharded forward's gathered buffer) - ``fmap2`` may then
    be None and ``V`` = len(slots).  ``compact``: level-0-only rows (``row_layout``); with ``pyramid_scale`` the epilogue then only
    scales (both builders: ``fuse_levels = 1``).  ``two_term`` (round 6; epipolar-line-tile kernel only, ignored by the walk): the source
    texels' lo planes are not read - half the tile kernel's fragment bytes, 1.2e-5 relative L1 on the volume (cer_mvs.h)."""
    if fmap2 is None:
        if split is None or len(split) < 3 or src_hw is None:
            raise ValueError("cost_build: without fmap2 the split rows, their view slots and src_hw are required")
        V, C = int(split[2].numel()), 64
        P2 = (src_hw[0] + 4) * (src_hw[1] + 4)
    else:
        V, P2, C = fmap2.shape
    P = h1 * w1
    h2, w2 = src_hw if src_hw is not None else (h1, w1)
    if P2 != (h2 + 4) * (w2 + 4):
        raise RuntimeError("cost_build: fmap2 must carry a 2-texel zero border ([V,(h+4)*(w+4),C])")
    _, _, rs = row_layout(D, num_levels, compact)
    fuse = pyramid_scale is not None
    if fuse and (not fold or accumulate or D > 64):
        raise ValueError("cost_build: the fused pyramid needs fold=True, accumulate=False and D <= 64")
    lib = L.load()
    on_lines = fold and C == 64 and D <= 64 and (fmap2 is None or lib.cer_cost_build_algo(-1) != 1)
    fuse_levels = 0 if not fuse else (1 if compact else num_levels)      # (1 = scale only, in both builders since ABI 1060)
    if vol is None:
        shape = (P, rs) if fold else (V, P, rs)
        if fuse:
            # fused pyramid (the forward's hot path): the kernel writes every level of every row - no 50 MB zero fill; only the
            # alignment pad of a row (rs - used, <= 3 floats) is cleared
            vol = torch.empty(shape, device=fmap1.device, dtype=torch.float32)
            offs, lens, _ = row_layout(D, num_levels)
            used = D if compact else offs[-1] + lens[-1]
            if used < rs:
                vol[..., used:] = 0
        else:
            vol = torch.zeros(shape, device=fmap1.device, dtype=torch.float32)     # pooled levels stay 0 until ``pyramid`` runs
    origin = torch.empty(P, device=fmap1.device, dtype=torch.float32)
    mode = (2 if accumulate else 1) if fold else 0
    slots = split[2] if (split is not None and len(split) > 2) else None
    if on_lines:
        f1s, f2s = split[:2] if split is not None else (None, None)
        if f1s is None:
            f1s = feat_split(fmap1)
        if f2s is None:
            f2s = feat_split(fmap2)
        ws = _lines_workspace(V, h1, w1, D, fmap1.device)
        L.check(lib.cer_cost_lines_f32(L.dev_ptr(f1s, "fmap1_split", torch.float16), L.dev_ptr(f2s, "fmap2_split", torch.float16),
                                       L.dev_ptr(slots, "view_slot", torch.int32), L.dev_ptr(Pij, "Pij"), L.dev_ptr(disp_in, "disp_in"), L.dev_ptr(vol, "vol"),
                                       L.dev_ptr(origin, "origin"), L.dev_ptr(ws, "workspace", torch.uint8), V, h1, w1, h2, w2, C, D, rs,
                                       float(incre), int(bool(shift)), mode, int(y0), fuse_levels,
                                       float(pyramid_scale) if fuse else 1.0, int(bool(two_term)), L.cur_stream()), "cost_lines")
    else:
        if fmap2 is None:
            raise RuntimeError("cost_build: this shape needs the fp32 walk, which takes fmap2 itself")
        L.check(lib.cer_cost_build_f32(L.dev_ptr(fmap1, "fmap1"), L.dev_ptr(fmap2, "fmap2"), L.dev_ptr(Pij, "Pij"),
                                       L.dev_ptr(disp_in, "disp_in"), L.dev_ptr(vol, "vol"), L.dev_ptr(origin, "origin"),
                                       V, h1, w1, h2, w2, C, D, rs, float(incre), int(bool(shift)), mode, int(y0),
                                       fuse_levels, float(pyramid_scale) if fuse else 1.0, L.cur_stream()),
                "cost_build")
    vol.level0_only = bool(compact)        # the row form, for the lookup wrappers (``_level0_only``)
    return vol, origin


def cost_lines_views(f1s, f2s, slots, Pij, disp_in, V, v0, nv, h1, w1, D, incre, shift, src_hw=None, y0=0, ws=None, two_term=False):
    """First half of the epipolar-line-tile cost volume for views v0 .. v0 + nv - 1 of V: their partial volumes go to the device's
    lines workspace (cer_cost_lines_views_f32).  ``cost_lines_reduce`` finishes the volume once every view has been built."""
    h2, w2 = src_hw if src_hw is not None else (h1, w1)
    if ws is None:                                           # (callers that split the two halves over streams pass one workspace to both)
        ws = _lines_workspace(V, h1, w1, D, f1s.device)
    L.check(L.load().cer_cost_lines_views_f32(L.dev_ptr(f1s, "fmap1_split", torch.float16), L.dev_ptr(f2s, "fmap2_split", torch.float16),
                                              L.dev_ptr(slots, "view_slot", torch.int32), L.dev_ptr(Pij, "Pij"), L.dev_ptr(disp_in, "disp_in"),
                                              L.dev_ptr(ws, "workspace", torch.uint8), V, int(v0), int(nv), h1, w1, h2, w2, 64, D, float(incre),
                                              int(bool(shift)), int(y0), int(bool(two_term)), L.cur_stream()), "cost_lines_views")


def cost_lines_reduce(disp_in, V, h1, w1, D, incre, shift, num_levels, pyramid_scale=None, vol=None, accumulate=False, ws=None, compact=False):
    """Second half: sum of the V partial volumes -> (vol [P, rs], origin [P]) with the fused view-mean scale + pooled levels when
    ``pyramid_scale`` is given (as ``cost_build``; ``compact``: level-0-only rows, scale only)."""
    P = h1 * w1
    _, _, rs = row_layout(D, num_levels, compact)
    fuse = pyramid_scale is not None and (num_levels > 1 or compact) and not accumulate
    dev = disp_in.device
    if vol is None:
        vol = torch.empty(P, rs, device=dev, dtype=torch.float32) if fuse else torch.zeros(P, rs, device=dev, dtype=torch.float32)
        if fuse:
            offs, lens, _ = row_layout(D, num_levels)
            used = D if compact else offs[-1] + lens[-1]
            if used < rs:
                vol[..., used:] = 0
    origin = torch.empty(P, device=dev, dtype=torch.float32)
    if ws is None:
        ws = _lines_workspace(V, h1, w1, D, dev)
    L.check(L.load().cer_cost_lines_reduce_f32(L.dev_ptr(ws, "workspace", torch.uint8), L.dev_ptr(disp_in, "disp_in"), L.dev_ptr(vol, "vol"),
                                               L.dev_ptr(origin, "origin"), V, h1, w1, D, rs, float(incre), int(bool(shift)),
                                               2 if accumulate else 1, (1 if compact else num_levels) if fuse else 0,
                                               float(pyramid_scale) if fuse else 1.0, L.cur_stream()), "cost_lines_reduce")
    if pyramid_scale is not None and not fuse:
        pyramid(vol, D, 1 if compact else num_levels, scale=float(pyramid_scale))
    vol.level0_only = bool(compact)
    return vol, origin


def pyramid(vol, D, num_levels, scale=1.0):
    rs = vol.shape[-1]
    rows = vol.numel() // rs
    L.check(L.load().cer_pyramid_f32(L.dev_ptr(vol, "vol"), rows, D, rs, num_levels, float(scale), L.cur_stream()), "pyramid")
    return vol


def corr_lookup(vol, origin, disp, D, incre, num_levels, radius, per_view_disp=False, level0_only=None):
    """vol [nv,P,rs] or [P,rs]; origin [P]; disp [P] (or [nv,P]) -> [nv, L*(2r+1), P].  ``level0_only``: see ``_level0_only``."""
    l0 = _level0_only(vol, D, num_levels, level0_only)
    if vol.dim() == 2:
        vol = vol[None]
    nv, P, rs = vol.shape
    out = torch.empty(nv, num_levels * (2 * radius + 1), P, device=vol.device, dtype=torch.float32)
    L.check(L.load().cer_corr_lookup_f32(L.dev_ptr(vol, "vol"), L.dev_ptr(origin, "origin"), L.dev_ptr(disp, "disp"),
                                         P if per_view_disp else 0, L.dev_ptr(out, "out"), nv, P, D, rs, float(incre),
                                         num_levels, radius, l0, L.cur_stream()), "corr_lookup")
    return out


def corr_encode(feats, w_t, b, out=None):
    """feats [nv,K,P] planar; w_t [K,64]; b [64] -> [P,64] (mean over views, 1x1 conv, ReLU)."""
    nv, K, P = feats.shape
    if out is None:
        out = torch.empty(P, 64, device=feats.device, dtype=torch.float32)
    L.check(L.load().cer_corr_encode_f32(L.dev_ptr(feats, "feats"), L.dev_ptr(w_t, "w"), L.dev_ptr(b, "b"), L.dev_ptr(out, "out"),
                                         nv, K, P, 64, L.cur_stream()), "corr_encode")
    return out


def lookup_encode(vol, origin, disp, w_t, b, D, incre, num_levels, radius, out=None, out_split=False, log2s=0, img_w=0, delta=None,
                  level0_only=None):
    """Folded volume [P,rs] -> relu(conv1x1(lookup)) [P,64] (``out_split``: True / 1 = split32 layout, see ``split32``;
    2 = frag16 layout of an image ``img_w`` pixels wide with scale 2^log2s, see ``s16_layout``: ``out`` [s16_pixels, 64]).
    ``level0_only`` (default: the builder's mark on ``vol``, see ``_level0_only``): the rows hold level 0 only (``row_layout(...,
    compact=True)``) and the kernel forms the pooled levels itself.
    ``delta`` = (T [nhalf,9,P], bias): the previous iteration's disparity update (``delta_sum``) is applied to ``disp`` IN PLACE by this
    launch before the lookup reads it (needs ``img_w``)."""
    P, rs = vol.shape
    l0 = _level0_only(vol, D, num_levels, level0_only)
    if out is None:
        rows = s16_pixels(P // img_w, img_w) if int(out_split) == 2 else P
        out = torch.zeros(rows, 64, device=vol.device, dtype=torch.float32)
    T, nhalf, dbias = (delta[0], int(delta[0].shape[0]), float(delta[1])) if delta is not None else (None, 0, 0.0)
    L.check(L.load().cer_lookup_encode_f32(L.dev_ptr(vol, "vol"), L.dev_ptr(origin, "origin"), L.dev_ptr(disp, "disp"),
                                           L.dev_ptr(w_t, "w"), L.dev_ptr(b, "b"), L.dev_ptr(out, "out"), P, D, rs, float(incre),
                                           num_levels, radius, 64, int(out_split), int(log2s), int(img_w), L.dev_ptr(T, "delta_taps"), nhalf, dbias,
                                           l0, L.cur_stream()), "lookup_encode")
    return out


class PackedConv3x3:
    """A 3x3 conv's weights packed for the MFMA kernels, for a given K-source list.
    ``sources``: list of (channels, kind) with kind 0 = tensor, 1 = disparity encoder (49).
    Two packings are kept: exact-fp32 B fragments (``packed``, v_mfma_f32_16x16x4_f32) and split
    hi|lo f16 fragments (``packed_x``, 3 x v_mfma_f32_32x32x16_f16, fp32-equivalent accuracy)."""

    def __init__(self, weight, bias, sources, device, corr_fp8=False):
        lib = L.load()
        w = weight.detach().to("cpu", torch.float32).contiguous()
        Cout, Cin = w.shape[0], w.shape[1]
        self.sources = list(sources)
        self.corr_fp8 = bool(corr_fp8)      # tensor sources packed for CER_EPI_CORR_FP8 launches (cer_mvs.h)
        n = len(sources)
        ch = (ctypes.c_int * n)(*[c for c, _ in sources])
        kind = (ctypes.c_int * n)(*[k for _, k in sources])
        kpad = sum(64 if k == 1 else (c + 31) // 32 * 32 for c, k in sources)
        size = lib.cer_conv3x3_packed_size(Cout, kpad)
        if size <= 0:
            raise RuntimeError(f"conv3x3 pack: unsupported shape Cout={Cout} Kpad={kpad}")
        packed = torch.empty(size, dtype=torch.float32)
        L.check(lib.cer_conv3x3_pack_f32(ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(packed.data_ptr()), Cout, Cin, ch, kind, n),
                "conv3x3_pack")
        self.packed = packed.to(device)
        size_x = lib.cer_conv3x3_f16x3_packed_size(Cout, kpad)
        if size_x <= 0:
            raise RuntimeError(f"conv3x3 f16x3 pack: unsupported shape Cout={Cout} Kpad={kpad}")
        packed_x = torch.empty(size_x, dtype=torch.float16)
        L.check(lib.cer_conv3x3_f16x3_pack(ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(packed_x.data_ptr()), Cout, Cin, ch, kind, n),
                "conv3x3_f16x3_pack")
        self.packed_x = packed_x.to(device)
        self.packed_c = None                 # collapsed-disparity packing (interior tiles), only with a kind-1 source
        if any(k == 1 for _, k in sources):
            size_c = lib.cer_conv3x3_f16x3_collapsed_size(Cout, ch, kind, n)
            if size_c <= 0:
                raise RuntimeError(f"conv3x3 collapsed pack: unsupported shape Cout={Cout}")
            packed_c = torch.empty(size_c, dtype=torch.float16)
            L.check(lib.cer_conv3x3_f16x3_pack_collapsed(ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(packed_c.data_ptr()), Cout, Cin,
                                                         ch, kind, n), "conv3x3_f16x3_pack_collapsed")
            self.packed_c = packed_c.to(device)
        self.bias = None if bias is None else bias.detach().to(device, torch.float32).contiguous()
        self.cout = Cout


CONV_MODE = "f16x3"      # default arithmetic of ops.conv3x3: "f16x3" (split-f16 MFMA, fp32-equivalent) or "fp32" (exact fp32 MFMA)


COLLAPSE_DISP = True     # interior tiles of a conv with a disparity source use the 81-tap collapsed form (cer_mvs.h)


def conv3x3(pc, srcs, h, w, epi, out=None, out2=None, aux=None, aux2=None, init=None, use_bias=True, mode=None, out_split=False,
            aux_split=False, kinds=None):
    """srcs: list of tensors matching ``pc.sources`` ([P,ch] for kind 0, disp [P] for kind 1).
    f16x3 mode only: ``kinds`` overrides the source kinds of this call (3 = the tensor is in the split32 layout, see
    ``split32``); ``out_split`` writes the (RELU/LINEAR out, GATES out2, GRU out) result in that layout, ``aux_split`` reads
    the previous hidden state ``aux`` from it."""
    mode = mode or CONV_MODE
    dev = srcs[0].device
    P = h * w
    ci = L.ConvInputs()
    ci.nsrc = len(srcs)
    for i, (t, (c, k)) in enumerate(zip(srcs, pc.sources)):
        ci.src[i] = t.data_ptr()
        L.dev_ptr(t, f"src{i}")
        ci.ch[i] = c
        ci.kind[i] = k if kinds is None else kinds[i]
    oc = pc.cout // 2 if epi == L.EPI_GATES else pc.cout
    if epi == L.EPI_DELTA:
        if mode != "f16x3":
            raise ValueError("EPI_DELTA is only built for the f16x3 kernel")
        if out is None:
            out = torch.empty(pc.cout // 128, 9, P, device=dev, dtype=torch.float32)
    if out is None:
        out = torch.empty(P, oc, device=dev, dtype=torch.float32)
    if epi == L.EPI_GATES and out2 is None:
        out2 = torch.empty(P, oc, device=dev, dtype=torch.float32)
    bias = pc.bias if (use_bias and init is None) else None
    aux_p = L.dev_ptr(aux, "aux", torch.float16) if epi == L.EPI_DELTA else L.dev_ptr(aux, "aux")
    tail = (L.dev_ptr(bias, "bias"), L.dev_ptr(init, "init"), L.dev_ptr(out, "out"), L.dev_ptr(out2, "out2"), aux_p,
            L.dev_ptr(aux2, "aux2"), h, w, pc.cout, epi, L.cur_stream())
    if mode == "f16x3":
        coll = L.dev_ptr(pc.packed_c, "packed_collapsed", torch.float16) if (COLLAPSE_DISP and pc.packed_c is not None) else None
        flags = (L.EPI_OUT_SPLIT if out_split else 0) | (L.EPI_AUX_SPLIT if aux_split else 0)
        tail = tail[:9] + (epi | flags,) + tail[10:]
        rc = L.load().cer_conv3x3_f16x3(ctypes.byref(ci), L.dev_ptr(pc.packed_x, "packed_w", torch.float16), coll, *tail)
    elif mode == "fp32":
        if out_split or aux_split or (kinds is not None and 3 in kinds):
            raise ValueError("the split32 layout exists only for the f16x3 kernels")
        rc = L.load().cer_conv3x3_f32(ctypes.byref(ci), L.dev_ptr(pc.packed, "packed_w"), *tail)
    else:
        raise ValueError(f"conv3x3: unknown mode {mode!r}")
    L.check(rc, f"conv3x3[{mode}]")
    return (out, out2) if epi == L.EPI_GATES else out


def delta_proj_pack(w2, device):
    """w2 [1,C,3,3] (delta{s}.2.weight) -> packed projection fragments for EPI_DELTA."""
    lib = L.load()
    w = w2.detach().to("cpu", torch.float32).contiguous()
    C = w.shape[1]
    size = lib.cer_delta_proj_packed_size(C)
    if size <= 0:
        raise RuntimeError(f"delta projection pack: unsupported C={C}")
    packed = torch.empty(size, dtype=torch.float16)
    L.check(lib.cer_delta_proj_pack(ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(packed.data_ptr()), C), "delta_proj_pack")
    return packed.to(device)


def delta_sum(T, bias, disp_in, h, w, disp_out=None, want_delta=True):
    """T [nhalf,9,P] tap planes -> (disp_out [P], delta [P] or None)."""
    nhalf, _, P = T.shape
    if disp_out is None:
        disp_out = torch.empty(P, device=T.device, dtype=torch.float32)
    delta = torch.empty(P, device=T.device, dtype=torch.float32) if want_delta else None
    L.check(L.load().cer_delta_sum_f32(L.dev_ptr(T, "T"), nhalf, float(bias), L.dev_ptr(disp_in, "disp_in"), L.dev_ptr(disp_out, "disp_out"),
                                       L.dev_ptr(delta, "delta"), h, w, L.cur_stream()), "delta_sum")
    return disp_out, delta


def delta_tail(hid, w_tap_c, bias, disp_in, h, w, disp_out=None, want_delta=True):
    """hid [P,C]; w_tap_c [9,C]; returns (disp_out [P], delta [P] or None)."""
    P, C = hid.shape
    if disp_out is None:
        disp_out = torch.empty(P, device=hid.device, dtype=torch.float32)
    delta = torch.empty(P, device=hid.device, dtype=torch.float32) if want_delta else None
    L.check(L.load().cer_delta_tail_f32(L.dev_ptr(hid, "hid"), L.dev_ptr(w_tap_c, "w"), float(bias), L.dev_ptr(disp_in, "disp_in"),
                                        L.dev_ptr(disp_out, "disp_out"), L.dev_ptr(delta, "delta"), h, w, C, L.cur_stream()),
            "delta_tail")
    return disp_out, delta


def nchw_to_nhwc(x, scale=1.0):
    """[C,h,w] (or [C,P]) -> [P,C] * scale."""
    C = x.shape[0]
    P = x.numel() // C
    out = torch.empty(P, C, device=x.device, dtype=torch.float32)
    L.check(L.load().cer_nchw_to_nhwc_f32(L.dev_ptr(x, "src"), L.dev_ptr(out, "dst"), C, P, float(scale), L.cur_stream()), "nchw_to_nhwc")
    return out


def nhwc_to_nchw(x, scale=1.0):
    """[P,C] -> [C,P] * scale."""
    P, C = x.shape
    out = torch.empty(C, P, device=x.device, dtype=torch.float32)
    L.check(L.load().cer_nhwc_to_nchw_f32(L.dev_ptr(x, "src"), L.dev_ptr(out, "dst"), C, P, float(scale), L.cur_stream()), "nhwc_to_nchw")
    return out


def plane_stats(x, eps=1e-5):
    """x [N,C,H,W] contiguous -> stats [N*C,2] = (mean, rstd) per (image, channel) plane."""
    N, C, H, W = x.shape
    stats = torch.empty(N * C, 2, device=x.device, dtype=torch.float32)
    L.check(L.load().cer_plane_stats_f32(L.dev_ptr(x, "x"), L.dev_ptr(stats, "stats"), N * C, H * W, float(eps), L.cur_stream()), "plane_stats")
    return stats


def norm_act(x, x_stats=None, res=None, res_stats=None, relu_a=False, relu_b=False, relu_out=False, out=None):
    """out = relu_out( relu_a(norm(x)) + relu_b(norm_r(res)) ) on NCHW tensors; ``out`` may be ``x``."""
    N, C, H, W = x.shape
    if out is None:
        out = torch.empty_like(x)
    flags = (1 if relu_a else 0) | (2 if relu_b else 0) | (4 if relu_out else 0)
    L.check(L.load().cer_norm_act_f32(L.dev_ptr(x, "x"), L.dev_ptr(x_stats, "x_stats"), L.dev_ptr(res, "res"), L.dev_ptr(res_stats, "res_stats"),
                                      L.dev_ptr(out, "out"), N * C, H * W, flags, L.cur_stream()), "norm_act")
    return out


def copy_segments(pairs):
    """One launch copying up to 4 contiguous fp32 ranges: pairs = [(src, dst), ...] with equal numel per pair
    (the row-slab halo pack / refresh, slab.py)."""
    if len(pairs) > L.COPY_MAX_SEG:
        raise ValueError(f"copy_segments: at most {L.COPY_MAX_SEG} segments")
    seg = L.CopySegments()
    for i, (src, dst) in enumerate(pairs):
        if src.numel() != dst.numel() or not src.is_contiguous() or not dst.is_contiguous():
            raise ValueError("copy_segments: each pair must be contiguous with equal numel")
        L.dev_ptr(src, f"src{i}")
        L.dev_ptr(dst, f"dst{i}")
        seg.src[i] = src.data_ptr()
        seg.dst[i] = dst.data_ptr()
        seg.n[i] = src.numel()
    L.check(L.load().cer_copy_segments_f32(ctypes.byref(seg), L.cur_stream()), "copy_segments")


def split32(x, inverse=False, out=None):
    """fp32 [P, C] (C % 32 == 0) -> the split32 layout of the f16x3 kernels (same shape / dtype, per pixel and 32-channel chunk
    32 hi halves | 32 lo halves), or back (``inverse``; hi + 2^-11 lo, exact to 2^-22 relative)."""
    P, C = x.shape
    if out is None:
        out = torch.empty_like(x)
    L.check(L.load().cer_split32_f32(L.dev_ptr(x, "src"), L.dev_ptr(out, "dst"), P, C, int(bool(inverse)), L.cur_stream()), "split32")
    return out


# ------------------------------------------------------------------------------------ "s16" convolutions (csrc/conv_s16.hip)
def s16_pixels(h, w):
    """Rows of an s16 tensor of an h x w image: whole m-tiles (2 rows x 16 columns), cer_s16_padded_pixels."""
    return (h + 1) // 2 * ((w + 15) // 16) * 32


def s16_layout(x, h, w, layout, log2s=0, inverse=False, out=None):
    """Plain fp32 [h*w, C] -> m-tile-major layout ``layout`` (L.S16_FRAG16 with scale 2^log2s | L.S16_ACC32 | L.S16_F32X8, see
    cer_mvs.h) as a [s16_pixels(h, w), C] tensor, or back (``inverse``)."""
    C = x.shape[1]
    if out is None:
        # (the padding pixels of the m-tile-major layouts must be zero; when the image is whole m-tiles - 296 x 400 - there are none to clear)
        alloc = torch.empty if (inverse or s16_pixels(h, w) == h * w) else torch.zeros
        out = alloc(h * w if inverse else s16_pixels(h, w), C, device=x.device, dtype=torch.float32)
    if x.shape[0] != (s16_pixels(h, w) if inverse else h * w):
        raise ValueError(f"s16_layout: {tuple(x.shape)} does not match a {h}x{w} image")
    L.check(L.load().cer_s16_layout_f32(L.dev_ptr(x, "src"), L.dev_ptr(out, "dst"), h, w, C, int(layout), int(log2s), int(bool(inverse)),
                                        L.cur_stream()), "s16_layout")
    return out


def s16_rows(tensor, rows, h, w, y0, nrows, to_tensor):
    """Image rows [y0, y0+nrows) of a frag16 tensor [s16_pixels(h,w), C] <-> ``rows`` (flat, nrows*w*C floats), bit-exact."""
    C = tensor.shape[1]
    if rows.numel() != nrows * w * C:
        raise ValueError("s16_rows: buffer size mismatch")
    L.check(L.load().cer_s16_rows_f32(L.dev_ptr(tensor, "tensor"), L.dev_ptr(rows, "rows"), h, w, C, int(y0), int(nrows), int(bool(to_tensor)),
                                      L.cur_stream()), "s16_rows")


def to_frag16(x, h, w, log2s, out=None):
    return s16_layout(x, h, w, L.S16_FRAG16, log2s, out=out)


def from_frag16(x, h, w, log2s):
    return s16_layout(x, h, w, L.S16_FRAG16, log2s, inverse=True)


class PackedConvS16:
    """A 3x3 conv's weights packed for ``conv3x3_s16``.  ``sources``: list of (channels, kind, log2 scale) in the weight's
    input-channel order; kind 2 = frag16 tensor, kind 1 = disparity encoder (49 channels, generated in the kernel)."""

    def __init__(self, weight, bias, sources, device, corr_fp8=False):
        lib = L.load()
        w = weight.detach().to("cpu", torch.float32).contiguous()
        Cout, Cin = w.shape[0], w.shape[1]
        self.sources = list(sources)
        self.corr_fp8 = 6 if corr_fp8 == 6 else int(bool(corr_fp8))      # 1: tensor sources packed for CER_EPI_CORR_FP8 launches, 6: for CER_EPI_CORR_FP6 (cer_mvs.h)
        n = len(sources)
        self.ch = (ctypes.c_int * n)(*[c for c, _, _ in sources])
        self.kind = (ctypes.c_int * n)(*[k for _, k, _ in sources])
        self.log2sx = (ctypes.c_int * L.CONV_MAX_SRC)(*([s for _, _, s in sources] + [0] * (L.CONV_MAX_SRC - n)))
        wp = ctypes.c_void_p(w.data_ptr())
        self.log2S = lib.cer_conv3x3_s16_scale(wp, Cout, Cin, self.ch, self.kind, self.log2sx, n)
        if self.log2S < -1000:
            raise RuntimeError("conv3x3_s16: cannot scale these weights / sources (one source's weights are too small next to another's "
                               "for a shared split-f16 scale); use gru_precision='f16x3'")

        def pack(collapsed):
            size = lib.cer_conv3x3_s16_packed_size(Cout, self.ch, self.kind, n, collapsed)
            if size <= 0:
                raise RuntimeError(f"conv3x3_s16 pack: unsupported shape Cout={Cout} sources={sources}")
            t = torch.empty(size, dtype=torch.float16)
            L.check(lib.cer_conv3x3_s16_pack(wp, ctypes.c_void_p(t.data_ptr()), Cout, Cin, self.ch, self.kind, self.log2sx, n,
                                             collapsed | (4 if self.corr_fp8 == 6 else 2 if self.corr_fp8 else 0), self.log2S), "conv3x3_s16_pack")
            return t.to(device)
        self.packed = pack(0)
        self.packed_c, self.edge = None, None
        if any(k == 1 for _, k, _ in sources):
            self.packed_c = pack(1)
            edge = torch.empty(lib.cer_conv3x3_s16_edge_size(Cout), dtype=torch.float16)
            L.check(lib.cer_conv3x3_s16_edge_pack(wp, ctypes.c_void_p(edge.data_ptr()), Cout, Cin, self.ch, self.kind, self.log2sx, n, self.log2S),
                    "conv3x3_s16_edge_pack")
            self.edge = edge.to(device)
        self.bias = None if bias is None else bias.detach().to(device, torch.float32).contiguous()
        self.cout = Cout


def delta_proj_pack_s16(w2, device):
    """w2 [1,C,3,3] (delta{s}.2.weight) -> (packed projection fragments for the s16 EPI_DELTA, log2 of their scale)."""
    lib = L.load()
    w = w2.detach().to("cpu", torch.float32).contiguous()
    C = w.shape[1]
    size = lib.cer_delta_proj_s16_packed_size(C)
    if size <= 0:
        raise RuntimeError(f"delta projection pack: unsupported C={C}")
    packed = torch.empty(size, dtype=torch.float16)
    log2s = ctypes.c_int(0)
    L.check(lib.cer_delta_proj_s16_pack(ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(packed.data_ptr()), C, ctypes.byref(log2s)), "delta_proj_s16_pack")
    return packed.to(device), log2s.value


TILE_MT = 0              # 0: let the library choose the tile height; tests force 3 / 4 / 5
EDGE_CORRECT = True      # collapsed disparity form on every tile + rim correction (False: border tiles run the literal form)


def conv3x3_s16(pc, srcs, h, w, epi, out=None, out2=None, aux=None, aux2=None, init=None, use_bias=True, out_split=False,
                log2s_out=0, log2s_aux=0):
    """srcs: tensors matching ``pc.sources`` (frag16 [s16_pixels(h,w), ch] for kind 2, plain disp [h*w] for kind 1).  Every
    tensor except the disparity and the DELTA planes is in an m-tile-major layout (cer_mvs.h, ``s16_layout``):
    LINEAR: ``out`` acc32 (or frag16 with scale 2^log2s_out when ``out_split``); RELU: frag16; GATES: (z f32x8, r*h frag16),
    aux = h frag16 (2^log2s_aux); GRU: new h frag16, aux = h frag16, aux2 = z f32x8; DELTA: plain tap planes [Cout/128, 9, h*w],
    aux = (packed projection, log2 scale); ``init``: acc32."""
    dev = srcs[0].device
    P, PP = h * w, s16_pixels(h, w)
    ci = L.ConvInputs()
    ci.nsrc = len(srcs)
    for i, (t, (c, k, _)) in enumerate(zip(srcs, pc.sources)):
        if t.numel() != (P if k == 1 else PP * c):
            raise ValueError(f"conv3x3_s16: source {i} has {t.numel()} elements, expected {(P if k == 1 else PP * c)}")
        ci.src[i] = t.data_ptr()
        L.dev_ptr(t, f"src{i}")
        ci.ch[i] = c
        ci.kind[i] = k
    oc = pc.cout // 2 if epi == L.EPI_GATES else pc.cout
    if epi == L.EPI_DELTA:
        aux, log2s_aux = aux
        if out is None:
            out = torch.empty(pc.cout // 128, 9, P, device=dev, dtype=torch.float32)
    alloc = torch.empty if PP == P else torch.zeros         # (padding pixels of the m-tile-major layouts stay zero; whole-m-tile images have none)
    if out is None:
        out = alloc(PP, oc, device=dev, dtype=torch.float32)
    if epi == L.EPI_GATES and out2 is None:
        out2 = alloc(PP, oc, device=dev, dtype=torch.float32)
    for name, t in (("out", out), ("out2", out2), ("aux", aux if epi != L.EPI_DELTA else None), ("aux2", aux2), ("init", init)):
        if t is not None and epi != L.EPI_DELTA and t.shape[0] != PP:
            raise ValueError(f"conv3x3_s16: {name} must have {PP} (padded) pixel rows, got {tuple(t.shape)}")
    bias = pc.bias if (use_bias and init is None) else None
    aux_p = L.dev_ptr(aux, "aux", torch.float16) if epi == L.EPI_DELTA else L.dev_ptr(aux, "aux")
    coll = L.dev_ptr(pc.packed_c, "packed_collapsed", torch.float16) if (COLLAPSE_DISP and pc.packed_c is not None) else None
    edge = L.dev_ptr(pc.edge, "edge_w", torch.float16) if (EDGE_CORRECT and coll is not None) else None
    flags = L.EPI_OUT_SPLIT if (out_split or epi in (L.EPI_RELU, L.EPI_GATES, L.EPI_GRU)) else 0
    if pc.corr_fp8:
        flags |= L.EPI_CORR_FP6 if pc.corr_fp8 == 6 else L.EPI_CORR_FP8
    rc = L.load().cer_conv3x3_s16(ctypes.byref(ci), pc.log2sx, L.dev_ptr(pc.packed, "packed_w", torch.float16), coll, edge, pc.log2S,
                                  L.dev_ptr(bias, "bias"), L.dev_ptr(init, "init"), L.dev_ptr(out, "out"), L.dev_ptr(out2, "out2"), aux_p,
                                  L.dev_ptr(aux2, "aux2"), h, w, pc.cout, epi | flags, int(log2s_out), int(log2s_aux), int(TILE_MT),
                                  L.cur_stream())
    L.check(rc, "conv3x3[s16]")
    return (out, out2) if epi == L.EPI_GATES else out
