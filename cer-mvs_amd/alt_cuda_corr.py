"""Drop-in for the reference's ``alt_cuda_corr`` extension module (alt_cuda_corr/correlation.cpp:51-54):
``forward(fmap1, fmap2, coords, radius) -> [corr]`` and
``backward(fmap1, fmap2, coords, corr_grad, radius) -> [fmap1_grad, fmap2_grad, coords_grad]``.
Same argument checks as CHECK_INPUT (correlation.cpp:19-21): CUDA + contiguous, else RuntimeError;
float32 only.  Unlike the reference the kernels run on torch's current stream."""
from . import ops


def forward(fmap1, fmap2, coords, radius):
    return [ops.alt_corr_forward(fmap1, fmap2, coords, int(radius))]


def backward(fmap1, fmap2, coords, corr_grad, radius):
    return list(ops.alt_corr_backward(fmap1, fmap2, coords, corr_grad, int(radius)))
