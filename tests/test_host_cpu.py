"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol include/cer_mvs.h declares,
argument errors are reported without touching a device, the host-side weight packing / geometry / driver
transforms are right, and the product path refuses to run without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import REPO, rel_l1
from test_oracle_golden import blank_state_dict, hashed


def header_symbols():
    text = open(os.path.join(REPO, "include", "cer_mvs.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cer_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from cer_mvs_amd import _lib
    lib = _lib.load()
    syms = header_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/cer_mvs.h but not exported"
    assert sorted(_lib.exported_symbols()) == syms, "python binding table and header disagree"
    assert lib.cer_abi_version() == _lib.ABI_VERSION


def test_variant_only_entry_points_are_not_in_the_product_library():
    """Round 5: the opt-in kernel forms of round 4 (measured slower) and their switches are declared in include/cer_mvs_variants.h and
    exported by csrc/variants/libcermvs_optin.so only."""
    from cer_mvs_amd import _lib
    text = open(os.path.join(REPO, "include", "cer_mvs_variants.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    vsyms = sorted(set(re.findall(r"\b(cer_[a-z0-9_]+)\s*\(", text)))
    assert vsyms == sorted(_lib._VARIANT_SIGNATURES)
    product = ctypes.CDLL(os.path.join(REPO, "cer-mvs_amd", "csrc", "libcermvs.so"))
    for s in vsyms:
        assert not hasattr(product, s), f"{s} is a variant-only entry point but the product library exports it"
    optin = os.path.join(REPO, "cer-mvs_amd", "csrc", "variants", "libcermvs_optin.so")
    if os.path.exists(optin):
        v = ctypes.CDLL(optin)
        for s in vsyms + header_symbols():
            assert hasattr(v, s), f"variants/libcermvs_optin.so does not export {s}"


def test_argument_errors_without_device():
    from cer_mvs_amd import _lib
    lib = _lib.load()
    null = ctypes.c_void_p(0)
    assert lib.cer_alt_corr_forward_f32(null, null, null, null, 1, 1, 4, 4, 4, 4, 64, 0, null) == -1
    assert lib.cer_cost_build_f32(null, null, null, null, null, null, 1, 4, 4, 4, 4, 64, 64, 112, ctypes.c_double(0.1), 1, 1, 0, 0, 1.0, null) == -1
    assert lib.cer_pyramid_f32(null, 10, 64, 112, 3, 1.0, null) == -1
    fake = ctypes.c_void_p(0x1000)
    assert lib.cer_alt_corr_forward_f32(fake, fake, fake, fake, 1, 1, 4, 4, 4, 4, 60, 0, null) == -2      # C % 64
    assert lib.cer_pyramid_f32(fake, 10, 64, 100, 3, 1.0, null) == -2                                      # row too short
    misaligned = ctypes.c_void_p(0x1004)
    assert lib.cer_alt_corr_forward_f32(misaligned, fake, fake, fake, 1, 1, 4, 4, 4, 4, 64, 0, null) == -3
    assert b"CER_ESHAPE" in lib.cer_error_string(-2)
    # round-3 entry points: the epipolar-line-tile cost volume and its operand split (argument checks only - nothing is launched)
    dbl = ctypes.c_double(0.1)
    assert lib.cer_feat_split_f16(null, null, 1, 10, 64, null, null) == -1
    assert lib.cer_feat_split_f16(fake, fake, 1, 10, 32, null, null) == -2                                     # C != 64
    assert lib.cer_feat_split_f16(misaligned, fake, 1, 10, 64, null, null) == -3
    # partial volumes + tile parameters + the hand-over list of the eight-line form (one entry per one-line tile of a view) + (round 6) one
    # 48-byte band record per tile (cost_lines_bands_kernel)
    tiles = max(13 * ((296 + 32 + 15) // 16 * 16), 10 * ((400 + 32 + 15) // 16 * 16))
    assert lib.cer_cost_lines_workspace(10, 296, 400, 64) == 10 * 296 * 400 * 64 * 4 + 10 * 16 + 256 + 10 * (tiles + 1) * 8 + 64 + 10 * tiles * 48 + 64
    assert lib.cer_cost_lines_workspace(0, 1, 1, 1) == -1
    if _lib.has_variant_forms():    # (variants/libcermvs_optin.so: the switch of round 4's multi-line form)
        prev = lib.cer_cost_lines_form(-1)
        assert prev in (0, 1) and lib.cer_cost_lines_form(1) == prev and lib.cer_cost_lines_form(prev) == 1 and lib.cer_cost_lines_form(-1) == prev
    args = (1, 4, 4, 4, 4, 64, 64, 112, dbl, 1)
    assert lib.cer_cost_lines_f32(null, null, null, null, null, null, null, null, *args, 1, 0, 0, 1.0, 0, null) == -1
    assert lib.cer_cost_lines_f32(fake, fake, null, fake, fake, fake, fake, fake, *args, 0, 0, 0, 1.0, 0, null) == -1      # mode 0: per-view volumes are the walk's
    assert lib.cer_cost_lines_f32(fake, fake, null, fake, fake, fake, fake, fake, 1, 4, 4, 4, 4, 64, 80, 144, dbl, 1, 1, 0, 0, 1.0, 0, null) == -2   # D > 64
    assert lib.cer_cost_lines_f32(fake, fake, null, fake, fake, fake, fake, fake, 1, 4, 4, 4, 4, 128, 64, 112, dbl, 1, 1, 0, 0, 1.0, 0, null) == -2  # C != 64
    assert lib.cer_cost_lines_views_f32(fake, fake, null, fake, fake, fake, 4, 3, 2, 4, 4, 4, 4, 64, 64, dbl, 1, 0, 0, null) == -1      # views 3..4 of 4
    assert lib.cer_cost_lines_views_f32(fake, fake, null, fake, fake, fake, 4, 0, 2, 4, 4, 4, 4, 64, 64, dbl, 1, 0, 2, null) == -1      # two_term is 0 or 1 (ABI 1070)
    assert lib.cer_cost_lines_reduce_f32(fake, fake, fake, null, 2, 4, 4, 64, 60, dbl, 1, 1, 0, 1.0, null) == -1           # row shorter than D
    assert lib.cer_cost_lines_reduce_f32(fake, fake, fake, null, 2, 4, 4, 64, 112, dbl, 1, 2, 3, 1.0, null) == -1          # fused pyramid needs mode 1
    assert lib.cer_f16_scan_overflow(fake, 24, fake, 4, null) == -1 and lib.cer_f16_scan_overflow(misaligned, 32, fake, 4, null) == -3
    assert lib.cer_overflow_flag(null) == 0 and lib.cer_cost_build_algo(-1) in (0, 1)


def test_depth_map_pipeline_argument_checks():
    """pipeline.DepthMapPipeline refuses a CPU model (there is no CPU path) and a stream count below one."""
    import pytest, torch
    from cer_mvs_amd import RAFT
    from cer_mvs_amd.pipeline import DepthMapPipeline
    model = RAFT(cascade=[(64, 64, 1), (-1, 320, 1)], test_mode=True)
    with pytest.raises(RuntimeError, match="GPU"):
        DepthMapPipeline(model, streams=2)
    with pytest.raises(ValueError):
        DepthMapPipeline(model, streams=0)


def test_argument_errors_of_the_conv_and_encoder_entry_points():
    from cer_mvs_amd import _lib
    lib = _lib.load()
    null, fake = ctypes.c_void_p(0), ctypes.c_void_p(0x1000)
    ci = _lib.ConvInputs()
    ci.nsrc = 1
    ci.src[0] = 0x1000
    ci.ch[0] = 48            # not a multiple of 32
    ci.kind[0] = 0
    assert lib.cer_conv3x3_f16x3(ctypes.byref(ci), fake, null, null, null, fake, null, null, null, 8, 8, 64, 1, null) == -2
    assert lib.cer_conv3x3_f32(ctypes.byref(ci), fake, null, null, fake, null, null, null, 8, 8, 64, 1, null) == -2
    ci.ch[0] = 64
    assert lib.cer_conv3x3_f16x3(ctypes.byref(ci), fake, null, null, null, fake, null, null, null, 8, 8, 64, 2, null) == -1   # GATES needs out2 + aux
    assert lib.cer_conv3x3_f16x3(ctypes.byref(ci), fake, null, null, null, fake, null, null, null, 8, 8, 96, 1, null) == -2   # Cout % 64
    assert lib.cer_conv3x3_f16x3_packed_size(64, 192) == 6 * 9 * 2 * 2048
    assert lib.cer_conv3x3_f16x3_packed_size(48, 192) == -2
    assert lib.cer_delta_proj_packed_size(256) == 2 * 8 * 2 * 512 and lib.cer_delta_proj_packed_size(200) == -2
    assert lib.cer_enc_conv_packed_size(64, 32, 9) == 9 * 2 * 2048 and lib.cer_enc_conv_packed_size(64, 32, 4) == -2
    buf16 = (ctypes.c_char * 16)()
    for pack in (lib.cer_enc_conv_pack, lib.cer_enc_conv_pack_f6):                # (round 6: the FP6-correction form's packer takes the same arguments)
        assert pack(None, buf16, 32, 32, 9) == -1 and pack(buf16, None, 32, 32, 9) == -1
        assert pack(buf16, buf16, 48, 32, 9) == -2 and pack(buf16, buf16, 32, 32, 4) == -2
    assert lib.cer_enc_conv_f16x3(fake, null, 0, fake, null, fake, null, null, 1, 8, 8, 48, 64, 9, 1, 0, 0, 1.0, null) == -2
    assert lib.cer_enc_conv_f16x3(fake, null, 0, fake, null, fake, null, null, 1, 8, 8, 32, 32, 9, 2, 0, 0, 1.0, null) == -2   # stride 2 needs Cout % 64
    assert lib.cer_enc_conv_f16x3(null, null, 0, fake, null, fake, null, null, 1, 8, 8, 32, 32, 9, 1, 0, 0, 1.0, null) == -1
    assert lib.cer_enc_conv_tiles(296, 400, 1, 9, 64) == 13 * 37 and lib.cer_enc_conv_tiles(296, 400, 2, 9, 64) == 13 * 148
    assert lib.cer_enc_stem_tiles(592, 800) == 925          # 592 rows x 400 pixel PAIRS / 256 threads
    assert lib.cer_enc_stem_s16_tiles(592, 800) == 74 * 25 and lib.cer_enc_stem_s16_tiles(9, 33) == 4      # 8 x 32-pixel tiles
    assert lib.cer_enc_stem_s16_packed_size() == 2 * 14 * 2 * 64 * 8          # two layouts: round-2 kernel | producer / consumer kernel
    assert lib.cer_enc_pc_supported(32, 32, 9, 1, 0) == 1 and lib.cer_enc_pc_supported(64, 64, 9, 1, 0) == 1 and lib.cer_enc_pc_supported(128, 128, 9, 1, 0) == 0
    assert lib.cer_enc_pc_tiles(296, 400, 64, 9, 1) == 13 * 37 and lib.cer_enc_pc_tiles(296, 400, 64, 9, 2) == 13 * 148 and lib.cer_enc_pc_tiles(296, 400, 128, 1, 1) == 13 * 74
    assert lib.cer_enc_pc_conv(fake, null, null, null, 1, null, fake, null, fake, null, null, 1, 8, 8, 48, 64, 9, 1, 0, 0, 1.0, null) == -2     # shape outside the HR encoder's
    assert lib.cer_enc_pc_conv(null, null, null, null, 1, null, fake, null, fake, null, null, 1, 8, 8, 32, 32, 9, 1, 0, 0, 1.0, null) == -1
    assert lib.cer_enc_pc_conv(fake, null, null, null, 1, fake, fake, null, fake, null, null, 1, 8, 8, 32, 32, 9, 1, 0, 0, 1.0, null) == -1     # merged_out needs srcB
    import numpy as np
    w = np.zeros((32, 3, 7, 7), dtype=np.float32)
    w[5, 1, 2, 6] = 0.75                                    # channel 5, ci 1, ky 2, kx 6 -> step 2*2 + 1, kg 1, element 0*4 + 1
    pk = np.zeros(lib.cer_enc_stem_s16_packed_size(), dtype=np.float16)
    k = ctypes.c_int(0)
    assert lib.cer_enc_stem_s16_pack(w.ctypes.data_as(ctypes.c_void_p), pk.ctypes.data_as(ctypes.c_void_p), ctypes.byref(k)) == 0
    assert k.value == 14                                    # 0.75 * 2^14 = 12288 < 16384
    pk = pk.reshape(2, 14, 2, 64, 8)
    assert pk[0, 5, 0, 32 + 5, 1] == 12288.0 and np.count_nonzero(pk[0]) == 1
    # second layout (producer / consumer kernel): the padding column in front, i.e. kernel column 6 sits at index 7 = 4 * 1 + 2 * 1 + 1
    assert pk[1, 5, 0, 32 + 5, 4 + 1] == 12288.0 and np.count_nonzero(pk[1]) == 1
    assert lib.cer_enc_stem_s16_pack(None, pk.ctypes.data_as(ctypes.c_void_p), ctypes.byref(k)) == -1
    assert lib.cer_enc_merge_f32(fake, null, null, null, fake, 1, 10, 30, 0, null) == -2                      # C % 4
    assert lib.cer_delta_sum_f32(null, 2, 0.0, fake, fake, null, 4, 4, null) == -1
    from cer_mvs_amd._lib import CopySegments
    seg = CopySegments()
    assert lib.cer_copy_segments_f32(None, null) == -1
    assert lib.cer_copy_segments_f32(ctypes.byref(seg), null) == 0          # all segments empty: nothing launched
    seg.n[1] = 16                                                          # non-empty segment without pointers
    assert lib.cer_copy_segments_f32(ctypes.byref(seg), null) == -1


def test_f16x3_weight_packing_splits_hi_lo():
    """cer_conv3x3_f16x3_pack (host code): hi = f16(w), lo = f16((w - hi) * 2^11), fragment order
    [chunk32][tap][ntile32][k16-step][hi|lo][lane][8]; hi + lo/2048 reproduces w to 2^-22."""
    from cer_mvs_amd import _lib
    lib = _lib.load()
    cout, cin = 64, 64
    w = hashed((cout, cin, 3, 3), 17, -0.3, 0.3)
    size = lib.cer_conv3x3_f16x3_packed_size(cout, cin)
    packed = torch.empty(size, dtype=torch.float16)
    ch, kind = (ctypes.c_int * 1)(cin), (ctypes.c_int * 1)(0)
    assert lib.cer_conv3x3_f16x3_pack(ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(packed.data_ptr()), cout, cin, ch, kind, 1) == 0
    pk = packed.view(cin // 32, 9, cout // 32, 2, 2, 64, 8).float()
    for kc, tap, nt, ks, lane, e in [(0, 0, 0, 0, 0, 0), (1, 5, 1, 1, 37, 3), (1, 8, 0, 0, 63, 7), (0, 2, 1, 1, 31, 4)]:
        co, ci_ = nt * 32 + lane % 32, kc * 32 + ks * 16 + (lane // 32) * 8 + e
        ref = float(w[co, ci_, tap // 3, tap % 3])
        hi, lo = float(pk[kc, tap, nt, ks, 0, lane, e]), float(pk[kc, tap, nt, ks, 1, lane, e])
        assert hi == float(torch.tensor(ref).half())
        assert abs(hi + lo / 2048.0 - ref) <= abs(ref) * 2.0 ** -21


def test_collapsed_disparity_weights_equal_the_literal_conv():
    """cer_conv3x3_f16x3_pack_collapsed (host code): for a pixel whose 3x3 neighbours are inside the image,
    conv3x3(100*(unfold7x7(d) - d)) == sum_s W9[s] * 100*(dz[p+s-4] - d[p]) with the pre-summed 81-tap filter."""
    from cer_mvs_amd import _lib
    from oracle import cer_oracle as O
    lib = _lib.load()
    cout, cin = 32, 49
    w = hashed((cout, cin, 3, 3), 23, -0.2, 0.2)
    ch, kind = (ctypes.c_int * 1)(cin), (ctypes.c_int * 1)(1)
    size = lib.cer_conv3x3_f16x3_collapsed_size(cout, ch, kind, 1)
    assert size == 3 * 1 * 2048
    packed = torch.empty(size, dtype=torch.float16)
    assert lib.cer_conv3x3_f16x3_pack_collapsed(ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(packed.data_ptr()), cout, cin, ch, kind, 1) == 0
    pk = packed.view(3, 1, 2, 2, 64, 8).double()                       # [step][ntile][k16][hi|lo][lane][8]
    w9 = torch.zeros(cout, 96, dtype=torch.float64)
    for kc in range(3):
        for ks in range(2):
            for lane in range(64):
                for e in range(8):
                    k = kc * 32 + ks * 16 + (lane // 32) * 8 + e
                    w9[lane % 32, k] = pk[kc, 0, ks, 0, lane, e] + pk[kc, 0, ks, 1, lane, e] / 2048.0
    assert torch.all(w9[:, 81:] == 0)
    h, wd = 9, 11
    d = hashed((1, 1, h, wd), 24, 0.0005, 0.0025)
    lit = F.conv2d(100 * O.disp_features(d).double(), w.double(), None, padding=1)[0]          # [cout, h, w]
    dz = F.pad(d[0, 0].double(), (4, 4, 4, 4))
    for (y, x) in [(1, 1), (4, 5), (h - 2, wd - 2), (1, wd - 2), (3, 1)]:
        win = 100 * (dz[y:y + 9, x:x + 9] - d[0, 0, y, x].double()).reshape(81)
        got = w9[:, :81] @ win
        assert torch.allclose(got, lit[:, y, x], rtol=0, atol=2e-6 * float(lit.abs().max())), (y, x)
    assert lib.cer_conv3x3_f16x3_collapsed_size(48, ch, kind, 1) == -2


def test_product_path_refuses_cpu_tensors():
    from cer_mvs_amd import RAFT, CorrBlock, alt_cuda_corr
    m = RAFT(test_mode=True)
    with pytest.raises(RuntimeError, match="CUDA"):
        m(torch.zeros(1, 3, 3, 32, 32), torch.eye(4).repeat(1, 3, 1, 1), torch.eye(3).repeat(1, 3, 1, 1), scale=1.0)
    with pytest.raises(RuntimeError, match="CUDA"):
        alt_cuda_corr.forward(torch.zeros(1, 4, 4, 64), torch.zeros(1, 4, 4, 64), torch.zeros(1, 1, 4, 4, 2), 0)
    with pytest.raises(RuntimeError, match="CUDA"):
        CorrBlock(torch.zeros(1, 2, 64, 4, 4), torch.eye(4).repeat(1, 2, 1, 1), torch.eye(3).repeat(1, 2, 1, 1), [0], [1], 64, 0.1,
                  torch.zeros(1, 1, 4, 4), True, 3, 5)


def test_missing_library_is_a_hard_error(monkeypatch, tmp_path):
    from cer_mvs_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()


def test_no_product_module_imports_the_oracle():
    """The oracle is test infrastructure: nothing under cer-mvs_amd/ may import, link or execute it."""
    pkg = os.path.join(REPO, "cer-mvs_amd")
    pat = re.compile(r"^\s*(from|import)\s+[\w.]*oracle|cer_oracle|oracle/", re.M)
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", "Makefile")):
                assert not pat.search(open(os.path.join(root, f)).read()), f"{f} references the oracle"


def test_state_dict_is_reference_compatible():
    from cer_mvs_amd import RAFT
    from cer_mvs_amd.synthetic import fill_state_dict
    m = RAFT(test_mode=True)
    sd = m.state_dict()
    want = blank_state_dict()
    assert sorted(sd) == sorted(want)
    for k in want:
        assert tuple(sd[k].shape) == tuple(want[k].shape), k
    filled = fill_state_dict(sd, seed=1)
    m.load_state_dict(filled, strict=True)
    m.load_state_dict({"module." + k: v for k, v in filled.items()}, strict=True)      # DataParallel checkpoints
    assert torch.equal(m.state_dict()["update_block.gru.convq.weight"], filled["update_block.gru.convq.weight"])


def test_cascade_resolution_matches_oracle():
    from cer_mvs_amd import RAFT
    from oracle import cer_oracle as O
    for cascade in ([(64, 64, 8), (-1, 320, 8)], [(64, 64, 16), (-1, 320, 16)], [(32, 32, 4)]):
        assert RAFT(cascade=cascade, test_mode=True).stages() == O.resolve_cascade(cascade)


def test_pij_matches_oracle():
    from cer_mvs_amd.projective import pij_matrices
    from cer_mvs_amd.synthetic import synthetic_scene
    from oracle import cer_oracle as O
    _, poses, intr, _ = synthetic_scene(8, 8, 5, seed=2)
    a = pij_matrices(poses[0], intr[0], [0] * 5, [1, 2, 3, 4, 5])
    b = O.pij_matrices(poses[0], intr[0], [0] * 5, [1, 2, 3, 4, 5])
    assert torch.equal(a, b)


def test_row_layout():
    from cer_mvs_amd.ops import row_layout
    assert row_layout(64, 3) == ([0, 64, 96], [64, 32, 16], 112)
    assert row_layout(44, 3) == ([0, 44, 66], [44, 22, 11], 80)


def test_conv_weight_packing_order():
    """cer_conv3x3_pack_f32 (host code) puts W[co, ci, ky, kx] at [kc16][tap][ntile][lane][s] with
    co = ntile*16 + lane%16, padded-K index = kc16*16 + (lane//16)*4 + s, zero in the padding."""
    from cer_mvs_amd import _lib
    lib = _lib.load()
    cout, srcs = 32, [(32, 0), (49, 1)]
    cin = 81
    w = hashed((cout, cin, 3, 3), 7)
    kpad = 32 + 64
    size = lib.cer_conv3x3_packed_size(cout, kpad)
    assert size == (kpad // 16) * 9 * (cout // 16) * 256
    packed = torch.empty(size)
    ch = (ctypes.c_int * 2)(32, 49)
    kind = (ctypes.c_int * 2)(0, 1)
    assert lib.cer_conv3x3_pack_f32(ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(packed.data_ptr()), cout, cin, ch, kind, 2) == 0
    pk = packed.view(kpad // 16, 9, cout // 16, 64, 4)
    kmap = list(range(32)) + list(range(32, 81)) + [-1] * 15
    for kc, tap, nt, lane, s in [(0, 0, 0, 0, 0), (1, 4, 1, 37, 2), (2, 8, 0, 63, 3), (5, 3, 1, 50, 1), (4, 7, 0, 17, 0)]:
        co = nt * 16 + lane % 16
        ci = kmap[kc * 16 + (lane // 16) * 4 + s]
        want = 0.0 if ci < 0 else float(w[co, ci, tap // 3, tap % 3])
        assert float(pk[kc, tap, nt, lane, s]) == want
    assert lib.cer_conv3x3_pack_f32(ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(packed.data_ptr()), cout, cin + 1, ch, kind, 2) == -2


def test_driver_transforms_match_reference_capture(golden, tmp_path):
    from cer_mvs_amd import inference as I
    g = golden("caller")
    im, k = torch.from_numpy(g["images"]), torch.from_numpy(g["intrinsics"])
    k0 = k.clone()
    im2, k2 = I.scale_operation(im, k, 1.5)
    assert torch.equal(k, k0)                                     # out of place
    assert np.allclose(im2.numpy(), g["scaled_images"], rtol=0, atol=1e-4) and np.array_equal(k2.numpy(), g["scaled_intrinsics"])
    im3, k3 = I.crop_operation(im2, k2, 24, 32)
    assert np.array_equal(im3.numpy(), g["cropped_images"]) and np.array_equal(k3.numpy(), g["cropped_intrinsics"])
    depth = I.disp_to_depth(g["disp"])
    assert depth.dtype == np.float32 and np.array_equal(depth, g["depth"])
    p = tmp_path / "d.pfm"
    I.write_pfm(p, depth)
    assert open(p, "rb").read() == g["pfm"].tobytes()
    with pytest.raises(Exception, match="float32"):
        I.write_pfm(p, depth.astype(np.float64))


def test_synthetic_scene_is_reproducible(golden):
    from cer_mvs_amd.synthetic import synthetic_scene, tensor_checksum
    g = golden("e2e_tiny")
    images, poses, intr, scale = synthetic_scene(int(g["H"]), int(g["W"]), int(g["V"]), seed=int(g["scene_seed"]))
    assert tensor_checksum(images) == int(g["images_checksum"])
    assert images.shape == (1, 4, 3, 64, 96) and float(images.min()) >= 0 and float(images.max()) <= 255
    assert float(scale) == 1.0


def test_fusion_host_logic():
    """cer-mvs_amd/fusion.py host side (no GPU): camera chains equal the oracle's matrices, PFM round trip, PLY layout,
    CPU tensors are refused."""
    import io
    from cer_mvs_amd import fusion
    from cer_mvs_amd.inference import write_pfm
    from cer_mvs_amd.synthetic import synthetic_scene
    _, poses, intr, _ = synthetic_scene(32, 48, 3, seed=4)
    K, E = intr[0], poses[0]
    cams = fusion.compose_cams(K[0], E[0], K[1:], E[1:])
    assert cams.shape == (3, fusion.CAM_FLOATS) and cams.dtype == torch.float32
    for s in range(3):
        assert torch.equal(cams[s, :9].view(3, 3), torch.inverse(K[0]))
        assert torch.equal(cams[s, 9:21].view(3, 4), torch.matmul(E[1 + s], torch.inverse(E[0]))[:3])
        assert torch.equal(cams[s, 21:30].view(3, 3), K[1 + s]) and torch.equal(cams[s, 51:60].view(3, 3), K[0])
        assert torch.equal(cams[s, 39:51].view(3, 4), torch.matmul(E[0], torch.inverse(E[1 + s]))[:3])
    with pytest.raises(RuntimeError, match="CUDA"):
        fusion.vote(torch.ones(4, 4), K[0], E[0], torch.ones(2, 4, 4), K[1:3], E[1:3], 4.0, 1300.0)
    with pytest.raises(RuntimeError, match="CUDA"):
        fusion.fuse_depth_maps(torch.ones(3, 4, 4), K[:3], E[:3], [(0, [1, 2])])


def test_pfm_and_ply_io(tmp_path):
    from cer_mvs_amd import fusion
    from cer_mvs_amd.inference import write_pfm
    d = hashed((7, 5), 61, 400.0, 800.0).numpy()
    write_pfm(tmp_path / "a.pfm", d)
    assert np.array_equal(fusion.read_pfm(tmp_path / "a.pfm"), d)
    xyz = hashed((6, 3), 62, -5.0, 5.0).numpy()
    rgb = (hashed((6, 3), 63, 0.0, 255.0).numpy()).astype(np.uint8)
    fusion.write_ply(tmp_path / "p.ply", xyz, rgb)
    raw = open(tmp_path / "p.ply", "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    assert head.startswith(b"ply\nformat binary_little_endian 1.0\nelement vertex 6\nproperty float x\n")
    assert len(body) == 6 * 15
    v = np.frombuffer(body, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("r", "u1"), ("g", "u1"), ("b", "u1")])
    assert np.array_equal(np.stack([v["x"], v["y"], v["z"]], 1), xyz) and np.array_equal(np.stack([v["r"], v["g"], v["b"]], 1), rgb)
    lib_null = ctypes.c_void_p(None)
    from cer_mvs_amd import _lib
    lib = _lib.load()
    fake = ctypes.c_void_p(4096)
    assert lib.cer_geo_consistency_f32(lib_null, fake, fake, 2, 8, 8, 4.0, 1300.0, lib_null, lib_null, lib_null, lib_null, lib_null, lib_null,
                                       lib_null, lib_null, lib_null) == -1
    assert lib.cer_geo_consistency_f32(fake, fake, fake, 11, 8, 8, 4.0, 1300.0, lib_null, lib_null, lib_null, lib_null, lib_null, lib_null,
                                       lib_null, lib_null, lib_null) == -2      # the reference's vote indexes at most 9 masks


def test_s16_shared_scale_bounds():
    """cer_conv3x3_s16_scale (host): the shared product scale keeps the largest scaled weight below 2^14 and refuses weight
    tensors in which a source's largest scaled weight would fall below 2^-3 (its lo halves would go subnormal)."""
    from importlib import import_module
    lib = import_module("cer-mvs_amd._lib").load()
    I3 = ctypes.c_int * 2
    ch, kind, sx = I3(32, 32), I3(0, 0), I3(6, 6)
    g = np.random.default_rng(3)
    w = g.standard_normal((64, 64, 3, 3)).astype(np.float32) * 0.05
    k = lib.cer_conv3x3_s16_scale(w.ctypes.data, 64, 64, ch, kind, sx, 2)
    assert k > -1000
    assert np.abs(w).max() * 2.0 ** (k - 6) < 2 ** 14 <= np.abs(w).max() * 2.0 ** (k + 1 - 6)
    # second source 2^-19 of the first: 2^14 * 2^-19 = 2^-5 < 2^-3 -> refused
    w2 = w.copy()
    w2[:, 32:] *= 2.0 ** -19
    assert lib.cer_conv3x3_s16_scale(w2.ctypes.data, 64, 64, ch, kind, sx, 2) < -1000
    # 2^-15 of the first still fits (2^13 .. 2^14 scaled maximum times 2^-15 >= 2^-2)
    w3 = w.copy()
    w3[:, 32:] *= 2.0 ** -15
    assert lib.cer_conv3x3_s16_scale(w3.ctypes.data, 64, 64, ch, kind, sx, 2) == k
    # an all-zero source is not a reason to refuse
    w4 = w.copy()
    w4[:, 32:] = 0
    assert lib.cer_conv3x3_s16_scale(w4.ctypes.data, 64, 64, ch, kind, sx, 2) == k
    # channel counts must add up to Cin
    assert lib.cer_conv3x3_s16_scale(w.ctypes.data, 64, 64, I3(32, 16), kind, sx, 2) < -1000


def test_fp8_correction_weight_packing():
    """cer_conv3x3_s16_pack(collapsed | 2) (host): per (32-channel tap step, 32 output channels) 4 KiB = f16 hi halves of both
    16-channel halves, then the e4m3 A operand of the fp8 matrix instruction: [lo * 2^5 | hi * 2^-6] per half, bytes 0-15 and 16-31
    of every lane 1 KiB apart.  The library's e4m3 encoder must agree with torch's (round to nearest even, subnormals)."""
    from importlib import import_module
    lib = import_module("cer-mvs_amd._lib").load()
    I1 = ctypes.c_int * 1
    ch, kind, sx = I1(64), I1(2), I1(14)
    Cout, Cin = 64, 64
    g = torch.Generator().manual_seed(11)
    w = (torch.rand(Cout, Cin, 3, 3, generator=g) - 0.5) * torch.logspace(-4, 0, Cin).view(1, Cin, 1, 1)     # small and large weights
    w = w.contiguous()
    log2S = lib.cer_conv3x3_s16_scale(ctypes.c_void_p(w.data_ptr()), Cout, Cin, ch, kind, sx, 1)
    assert log2S > -1000
    size = lib.cer_conv3x3_s16_packed_size(Cout, ch, kind, 1, 2)
    assert size == lib.cer_conv3x3_s16_packed_size(Cout, ch, kind, 1, 0) == 2 * 9 * 2 * 2048
    packed = torch.zeros(size, dtype=torch.float16)
    assert lib.cer_conv3x3_s16_pack(ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(packed.data_ptr()), Cout, Cin, ch, kind, sx, 1, 2, log2S) == 0
    raw = packed.view(torch.uint8).reshape(2 * 9, 2, 4, 64, 16)                  # [chunk step][n-tile][h0 | h1 | q0 | q1][lane][16 B]
    ws = (w.double() * 2.0 ** (log2S - 14)).float()
    hi = ws.half()
    lo = (ws - hi.float()).half()
    lane = torch.arange(64)
    co, kg = lane & 31, lane >> 5
    for c32, tap, nt in ((0, 0, 0), (1, 4, 1), (1, 8, 0), (0, 7, 1)):
        blk = raw[c32 * 9 + tap, nt]
        for hc in range(2):
            cin = c32 * 32 + hc * 16 + kg[:, None] * 8 + torch.arange(8)[None, :]                  # [lane, e]
            cout = (nt * 32 + co)[:, None].expand(64, 8)
            h_ref = hi[cout, cin, tap // 3, tap % 3]
            l_ref = lo[cout, cin, tap // 3, tap % 3]
            assert torch.equal(blk[hc].contiguous().view(torch.float16).reshape(64, 8), h_ref)
            q = blk[2 + hc]                                                                             # [lane][16]: wl * 2^5 | wh * 2^-6
            assert torch.equal(q[:, :8], (l_ref.float() * 32).to(torch.float8_e4m3fn).view(torch.uint8))
            assert torch.equal(q[:, 8:], (h_ref.float() / 64).to(torch.float8_e4m3fn).view(torch.uint8))


def test_arithmetic_forms_of_the_auto_walk():
    """Round 6: the calibration walk's candidates carry the encoders' form as a "+e6" suffix (csrc/enc_pc.hip, FP6 correction terms); it is
    honoured only while enc_precision is "auto", and pinned encoders drop the suffixed candidates from the walk (host logic, no GPU)."""
    import pytest
    from cer_mvs_amd import RAFT
    m = RAFT(test_mode=True)
    assert m.enc_precision == "auto" and m._auto_forms() == RAFT.AUTO_FORMS and RAFT.AUTO_FORMS[-1] == "s16"
    assert RAFT.AUTO_FORMS[0] == "s16f8+e6+c2" and m._enc_f6 and m._cost_x2 and m.update_block.corr_fp8 is True      # the first candidate until a calibration says otherwise
    m._set_form("s16f8+e6")
    assert m._enc_f6 and not m._cost_x2
    m._set_form("s16f8")
    assert not m._enc_f6 and not m._cost_x2 and m.update_block.corr_fp8 is True
    m._set_form("s16")
    assert not m._enc_f6 and m.update_block.corr_fp8 is False
    p = RAFT(test_mode=True, enc_precision="f16x3")
    assert p._auto_forms() == tuple(f for f in RAFT.AUTO_FORMS if "e6" not in f.split("+")) and not p._enc_f6
    p._set_form("s16f8+e6+c2")
    assert not p._enc_f6 and p._cost_x2                                # pinned encoders: their suffix is ignored, the cost volume's is not
    c = RAFT(test_mode=True, cost_precision="x3")
    assert c._auto_forms() == tuple(f for f in RAFT.AUTO_FORMS if "c2" not in f.split("+")) and c._enc_f6 and not c._cost_x2
    assert RAFT(test_mode=True, cost_precision="x2", gru_precision="s16")._cost_x2
    f = RAFT(test_mode=True, enc_precision="f6", gru_precision="s16")
    assert f._enc_f6 and f.update_block.corr_fp8 is False
    f._set_form("s16f8")
    assert f._enc_f6
    pinned = RAFT(test_mode=True, gru_precision="s16f8")
    assert not pinned._enc_f6 and not pinned._cost_x2                  # a pinned update-block form leaves the encoders and the cost volume fp32-class
    with pytest.raises(ValueError):
        RAFT(test_mode=True, enc_precision="fp8")
    with pytest.raises(ValueError):
        RAFT(test_mode=True, cost_precision="f16")


def test_encoder_fp6_weight_packing():
    """cer_enc_conv_pack_f6 (host, round 6): size and plane order of cer_enc_conv_pack - [chunk32][tap][ntile32][k16-step][hi | q][lane][16 B] - with the hi
    planes IDENTICAL to cer_enc_conv_pack's and the two q planes of a tap holding the lane's K block of the FP6 matrix instruction: 32 e2m3 fields, field i at
    bit 6 i = [wl' (8) | wh (8)] of k16-step 0, then of step 1, divided by one power of two per block; dwords 0-3 in step 0's plane, dwords 4-5 | E8M0 byte of the
    scale * 2^-11 | 0 in step 1's.  Decoded here with an independent e2m3 table: every field must be the nearest representable value (ties to even)."""
    import numpy as np
    from importlib import import_module
    lib = import_module("cer-mvs_amd._lib").load()
    Cout, Cin, taps = 64, 64, 9
    g = torch.Generator().manual_seed(17)
    w = ((torch.rand(Cout, Cin, 3, 3, generator=g) - 0.5) * torch.logspace(-3, 0, Cin).view(1, Cin, 1, 1)).contiguous()
    w[5, 32:48] = 0.0                                                             # an all-zero block
    size = lib.cer_enc_conv_packed_size(Cout, Cin, taps)
    p3, p6 = torch.zeros(size, dtype=torch.float16), torch.zeros(size, dtype=torch.float16)
    assert lib.cer_enc_conv_pack(ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(p3.data_ptr()), Cout, Cin, taps) == 0
    assert lib.cer_enc_conv_pack_f6(ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(p6.data_ptr()), Cout, Cin, taps) == 0
    r3 = p3.view(torch.uint8).reshape(Cin // 32, taps, Cout // 32, 2, 2, 64, 16)  # [chunk][tap][ntile][ks][hi | lo][lane][16 B]
    r6 = p6.view(torch.uint8).reshape(Cin // 32, taps, Cout // 32, 2, 2, 64, 16)
    assert torch.equal(r3[:, :, :, :, 0], r6[:, :, :, :, 0])                      # the main term's operands are the same bytes
    mags = np.array([0, .125, .25, .375, .5, .625, .75, .875, 1, 1.125, 1.25, 1.375, 1.5, 1.625, 1.75, 1.875,
                     2, 2.25, 2.5, 2.75, 3, 3.25, 3.5, 3.75, 4, 4.5, 5, 5.5, 6, 6.5, 7, 7.5])
    hi = w.half()
    lo = ((w - hi.float()) * 2048.0).half()
    for kc, tap, nt in ((0, 0, 0), (1, 4, 1), (1, 8, 0), (0, 7, 1), (1, 3, 0)):
        q = np.concatenate([r6[kc, tap, nt, 0, 1].numpy(), r6[kc, tap, nt, 1, 1].numpy()], axis=1).copy().view(np.uint32)      # [lane][8 dwords]
        for lane in range(64):
            co, kg = nt * 32 + (lane & 31), lane >> 5
            bits = int.from_bytes(q[lane, :6].tobytes(), "little")
            fields = [(bits >> (6 * i)) & 63 for i in range(32)]
            dec = np.array([(-1.0 if f & 32 else 1.0) * mags[f & 31] for f in fields])
            sb = int(q[lane, 6])
            assert q[lane, 7] == 0 and 0 < sb < 255
            t = 2.0 ** (sb - 127 + 11)                                            # the block's scale (the byte carries t * 2^-11)
            ref = []
            for ks in range(2):
                ci = kc * 32 + ks * 16 + kg * 8 + np.arange(8)
                ref += [lo[co, ci, tap // 3, tap % 3].double().numpy(), hi[co, ci, tap // 3, tap % 3].double().numpy()]
            ref = np.concatenate(ref)
            mx = np.abs(ref).max()
            if mx == 0:
                assert not np.any(dec)
                continue
            assert 3.875 <= mx / t <= 7.75                                        # the block maximum lands in the top binade (or just under it)
            # nearest representable magnitude, ties to the even code
            x = np.abs(ref) / t
            d = np.abs(x[:, None] - mags[None, :])
            best = d.min(axis=1)
            assert np.all(np.abs(np.abs(dec) - x) <= best + 1e-12)
            assert np.all((np.sign(dec) == np.sign(ref)) | (dec == 0))
