"""Host-side part of the epipolar warp (reference: utils/projective_ops.py:16-23).

Only the 4x4 products stay in torch - ``Pij = K_j P_j P_i^-1 K_i^-1`` for the V source views, a few
hundred flops, evaluated in float32 on the host exactly as the reference evaluates it (two batched
inverses, three products); the per-pixel part (coords_grid, contraction, divide, clamp:
projective_ops.py:5-13,26-28, core/corr.py:87-88) runs inside the cost-build kernel."""
import torch


def pij_matrices(poses, intrinsics, ii, jj):
    """poses [N,4,4] world->camera, intrinsics [N,3,3] at feature resolution, ii/jj index lists
    -> float32 CPU tensor [len(jj),4,4]."""
    poses = poses.detach().to("cpu", torch.float32)
    intrinsics = intrinsics.detach().to("cpu", torch.float32)
    ii = torch.as_tensor(ii, dtype=torch.long).cpu()
    jj = torch.as_tensor(jj, dtype=torch.long).cpu()
    Ks = torch.zeros_like(poses)
    Ks[:, :3, :3] = intrinsics
    Ks[:, 3, 3] = 1.0
    return (Ks[jj] @ poses[jj] @ torch.inverse(poses[ii]) @ torch.inverse(Ks[ii])).contiguous()
