"""The oracle's lens undistortion ([UPSTREAM-RECALL] camera_utils.radial_and_tangential_undistort) checked on the CPU
against the forward OPENCV model it inverts, and the ray generator's zero-parameter guard."""
import pytest
import torch

from oracle import nerf_oracle as O

# (k1, k2, k3, k4, p1, p2): a COLMAP OPENCV camera of an `ns-process-data images` scene, a radial-only phone lens, a
# strong barrel with all six terms
LENSES = [(-0.05, 0.02, 0.0, 0.0, 1e-3, -1e-3), (0.12, -0.03, 0.0, 0.0, 0.0, 0.0), (-0.2, 0.06, -0.01, 0.002, 4e-3, 2e-3)]


def _distort(xy, dp):
    k1, k2, k3, k4, p1, p2 = dp
    x, y = xy[..., 0].double(), xy[..., 1].double()
    r = x * x + y * y
    d = 1 + r * (k1 + r * (k2 + r * (k3 + r * k4)))
    return torch.stack([x * d + 2 * p1 * x * y + p2 * (r + 2 * x * x), y * d + 2 * p2 * x * y + p1 * (r + 2 * y * y)], -1)


@pytest.mark.parametrize("dp", LENSES)
def test_undistort_inverts_the_forward_model(dp):
    g = torch.Generator().manual_seed(3)
    coords = (torch.rand(4096, 2, generator=g) * 2 - 1) * torch.tensor([0.9, 0.55])   # a 1080p frame at fx = 1111 spans +-0.86 x +-0.49
    und = O.radial_and_tangential_undistort(coords, torch.tensor(dp, dtype=torch.float32))
    assert und.dtype == torch.float32 and (und - coords).abs().max() > 1e-3
    assert (_distort(und, dp) - coords.double()).abs().max() < 5e-7


def test_constants_are_the_headers():
    import os, re
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "unerf.h")).read()
    assert int(re.search(r"#define UNERF_UNDISTORT_ITERATIONS (\d+)", hdr).group(1)) == O.UNDISTORT_MAX_ITERATIONS
    assert float(re.search(r"#define UNERF_UNDISTORT_EPS ([0-9.e+-]+)f", hdr).group(1)) == O.UNDISTORT_EPS


def test_zero_parameters_leave_the_rays_untouched():
    c2w = torch.eye(4)[:3]
    a = O.generate_rays(c2w, 40.0, 41.0, 16.0, 12.0, 24, 32)
    b = O.generate_rays(c2w, 40.0, 41.0, 16.0, 12.0, 24, 32, distortion=torch.zeros(6))
    c = O.generate_rays(c2w, 40.0, 41.0, 16.0, 12.0, 24, 32, distortion=LENSES[0])
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    assert (a[1] - c[1]).abs().max() > 1e-4 and torch.equal(a[0], c[0])
