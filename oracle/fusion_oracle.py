"""TEST INFRASTRUCTURE ONLY - CPU (torch, fp32) restatement of the geometric-consistency depth-map fusion that follows
the depth-inference path in the reference: fusion.py:39-106 (reprojection + 2-D bilinear sample
+ the nine-threshold masks) and fusion.py:199-262 (per-view vote, averaged depth, ten-round bisection of the threshold
exponent on the mean mask area).  Pinned to captures of the reference's own `fusion()` run on CPU under IO shims
(tools/gen_golden_fusion.py -> tests/golden/fusion.npz).  Only tests/, __graft_entry__.smoke() and bench tooling may import
this module; the product path (cer-mvs_amd/fusion.py + csrc/fusion.hip) never does."""
import torch
import torch.nn.functional as F


def bilinear_sample(img, x, y):
    """img [B,1,H,W], pixel coordinates x, y [B,H,W] -> [B,1,H,W]; align_corners=True, zeros outside
    (reference: utils/bilinear_sampler.py:32-41)."""
    H, W = img.shape[-2:]
    xg = 2 * x / (W - 1) - 1
    yg = 2 * y / (H - 1) - 1
    return F.grid_sample(img, torch.stack([xg, yg], dim=-1), align_corners=True)


def reproject_with_depth(depth_ref, K_ref, E_ref, depth_src, K_src, E_src):
    """All [B,...] batched over source views (reference: fusion.py:39-83).  Returns
    (depth_reprojected, x_reprojected, y_reprojected, x_src, y_src), each [B,H,W]."""
    B, H, W = depth_ref.shape
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    x = xx.reshape(1, -1).repeat(B, 1)
    y = yy.reshape(1, -1).repeat(B, 1)
    ones = torch.ones_like(x)
    pix = torch.stack((x, y, ones), dim=1) * depth_ref.reshape(B, 1, -1)                 # fusion.py:51
    xyz_ref = torch.matmul(torch.inverse(K_ref), pix)                                    # :50,52
    hom = torch.cat((xyz_ref, ones.unsqueeze(1)), dim=1)
    xyz_src = torch.matmul(torch.matmul(E_src, torch.inverse(E_ref)), hom)[:, :3]        # :55-56
    kx = torch.matmul(K_src, xyz_src)                                                    # :58
    xy_src = kx[:, :2] / kx[:, 2:3]                                                      # :59
    x_src = xy_src[:, 0].reshape(B, H, W).float()
    y_src = xy_src[:, 1].reshape(B, H, W).float()
    sampled = bilinear_sample(depth_src.view(B, 1, H, W), x_src, y_src)                  # :67
    back = torch.cat((xy_src, ones.unsqueeze(1)), dim=1) * sampled.reshape(B, 1, -1)     # :72
    xyz_s = torch.matmul(torch.inverse(K_src), back)                                     # :71
    xyz_r = torch.matmul(torch.matmul(E_ref, torch.inverse(E_src)),
                         torch.cat((xyz_s, ones.unsqueeze(1)), dim=1))[:, :3]             # :74-75
    depth_rep = xyz_r[:, 2].reshape(B, H, W).float()                                     # :77
    kr = torch.matmul(K_ref, xyz_r)
    xy_r = kr[:, :2] / kr[:, 2:3]                                                        # :78-79
    return depth_rep, xy_r[:, 0].reshape(B, H, W).float(), xy_r[:, 1].reshape(B, H, W).float(), x_src, y_src


def check_geometric_consistency(depth_ref, K_ref, E_ref, depth_src, K_src, E_src, thre1=4.4, thre2=1430.0):
    """reference: fusion.py:86-106.  Returns (masks[9], mask, depth_reprojected (zeroed outside the last mask),
    x_src, y_src, relative_depth_diff)."""
    B, H, W = depth_ref.shape
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    depth_rep, xr, yr, xs, ys = reproject_with_depth(depth_ref, K_ref, E_ref, depth_src, K_src, E_src)
    dist = torch.sqrt((xr - xx[None]) ** 2 + (yr - yy[None]) ** 2)                       # :94
    rel = torch.abs(depth_rep - depth_ref) / depth_ref                                   # :97-98
    masks = [torch.logical_and(dist < i / thre1, rel < i / thre2) for i in range(2, 11)]  # :100-103
    mask = masks[-1]
    depth_rep = depth_rep.clone()
    depth_rep[~mask] = 0                                                                 # :104
    return masks, mask, depth_rep, xs, ys, rel


def vote(depth_ref, K_ref, E_ref, depth_src, K_src, E_src, thre1, thre2):
    """One reference view against its S source views (reference: fusion.py:213-241).  depth_ref [H,W], depth_src [S,H,W].
    Returns (geo_mask bool [H,W], depth_est [H,W])."""
    S = depth_src.shape[0]
    n = 1 + S
    d = depth_ref.unsqueeze(0).repeat(S, 1, 1)
    masks, geo, depth_rep, _, _, _ = check_geometric_consistency(d, K_ref.unsqueeze(0).repeat(S, 1, 1), E_ref.unsqueeze(0).repeat(S, 1, 1),
                                                                 depth_src, K_src, E_src, thre1, thre2)
    sums = [masks[i - 2].sum(dim=0).int() for i in range(2, n)]                           # :226-228
    geo_sum = geo.sum(dim=0)                                                             # :230
    geo_mask = geo_sum >= n                                                              # :232
    for i in range(2, n):
        geo_mask = torch.logical_or(geo_mask, sums[i - 2] >= i)                          # :234-235
    depth_est = (depth_rep.sum(dim=0) + depth_ref) / (geo_sum + 1)                        # :236
    return geo_mask, depth_est


def fuse(depths, Ks, Es, pairs, glb=0.25, rounds=10):
    """The bisection loop of fusion.py:199-262 without the file IO: depths [N,H,W], Ks [N,3,3], Es [N,4,4],
    pairs = [(ref index, [source indices])].  Returns (masks [N,H,W] bool, depth_est [N,H,W], final exponent, history)
    of the LAST round (the one the reference writes out)."""
    lo, hi = -2.0, 2.0
    hist = []
    for it in range(rounds):
        thre = (lo + hi) / 2
        est = torch.zeros_like(depths)
        masks = torch.zeros(depths.shape, dtype=torch.bool)
        means = []
        for ref, srcs in pairs:
            m, e = vote(depths[ref], Ks[ref], Es[ref], depths[srcs], Ks[srcs], Es[srcs], 10 ** thre * 4, 10 ** thre * 1300)
            masks[ref], est[ref] = m, e
            means.append(m.float().mean().item())
        mean = sum(means) / len(means)
        hist.append((thre, mean))
        if mean >= glb:
            lo = thre
        else:
            hi = thre
    return masks, est, thre, hist


def backproject(depth, mask, K, E):
    """Masked pixels -> world points [M,3] (reference: fusion.py:262-270, numpy there; fp64 here like np.linalg.inv)."""
    H, W = depth.shape
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    x, y, d = xx[mask].double(), yy[mask].double(), depth[mask].double()
    cam = torch.linalg.inv(K.double()) @ (torch.stack((x, y, torch.ones_like(x))) * d)
    world = (torch.linalg.inv(E.double()) @ torch.cat((cam, torch.ones_like(x)[None]), 0))[:3]
    return world.t()
