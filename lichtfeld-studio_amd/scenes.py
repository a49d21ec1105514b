"""Synthetic scenes of SURVEY.md §8d (the reference has no datasets in-tree and there is no
network): SYN-A = BASELINE.json config 1, SYN-B = config 2, SYN-C = config 4, SYN-D = config 5.

Everything is generated on the CPU with a fixed torch seed so that every build / rank sees the
same bits, then moved to the device by the caller.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import torch


@dataclass
class Scene:
    name: str
    width: int
    height: int
    sh_degree: int
    means: torch.Tensor        # [N,3]
    raw_quats: torch.Tensor    # [N,4] (w,x,y,z), un-normalised
    raw_scales: torch.Tensor   # [N,3] log-scales
    raw_opacities: torch.Tensor  # [N] logits
    sh0: torch.Tensor          # [N,1,3]
    shN: torch.Tensor          # [N,K-1,3]
    viewmats: torch.Tensor     # [V,4,4] world->camera, row-major
    Ks: torch.Tensor           # [V,3,3]
    extra: dict = field(default_factory=dict)

    @property
    def N(self) -> int:
        return self.means.shape[0]

    def to(self, device) -> "Scene":
        kw = {k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in self.__dict__.items()}
        return Scene(**kw)


def _logit(p: torch.Tensor) -> torch.Tensor:
    return torch.log(p / (1 - p))


def look_at_viewmat(eye, target=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0)) -> torch.Tensor:
    """OpenCV-style camera (x right, y down, z forward); returns world->camera [4,4]."""
    eye = torch.tensor(eye, dtype=torch.float64)
    target = torch.tensor(target, dtype=torch.float64)
    up = torch.tensor(up, dtype=torch.float64)
    z = target - eye
    z = z / z.norm()
    x = torch.linalg.cross(z, up)
    x = x / x.norm()
    y = torch.linalg.cross(z, x)
    R = torch.stack([x, y, z], 0)  # rows = camera axes in world coords
    t = -R @ eye
    m = torch.eye(4, dtype=torch.float64)
    m[:3, :3] = R
    m[:3, 3] = t
    return m.to(torch.float32)


def orbit_cameras(n_views: int, radius: float = 10.0, heights=(-1.0, 0.0, 1.0)) -> torch.Tensor:
    mats = []
    for v in range(n_views):
        ang = 2.0 * math.pi * v / n_views
        eye = (radius * math.cos(ang), heights[v % len(heights)], radius * math.sin(ang))
        mats.append(look_at_viewmat(eye))
    return torch.stack(mats, 0)


def syn_a(seed: int = 42, n: int = 10_000, sh_degree: int = 0) -> Scene:
    """config 1: 10k Gaussians, 1 camera, 256x256, SH degree 0 (tests/torch_impl.cpp-sized)."""
    g = torch.Generator().manual_seed(seed)
    means = torch.randn(n, 3, generator=g)
    means[:, 2] = means[:, 2].abs() + 3
    quats = torch.randn(n, 4, generator=g)
    scales = torch.rand(n, 3, generator=g) * 0.05 + 0.01
    opac = torch.rand(n, generator=g) * 0.8 + 0.1
    K_ = (sh_degree + 1) ** 2
    sh = torch.randn(n, K_, 3, generator=g) * 0.5
    viewmats = torch.eye(4).unsqueeze(0)
    Ks = torch.tensor([[[200.0, 0, 128], [0, 200.0, 128], [0, 0, 1]]])
    return Scene("SYN-A", 256, 256, sh_degree, means, quats, scales.log(), _logit(opac), sh[:, :1].contiguous(),
                 sh[:, 1:].contiguous(), viewmats, Ks)


def _syn_box(name, seed, n, width, height, focal, n_views, sh_degree=3) -> Scene:
    g = torch.Generator().manual_seed(seed)
    means = (torch.rand(n, 3, generator=g) * 2 - 1) * 4
    quats = torch.randn(n, 4, generator=g)
    raw_scales = math.log(0.02) + 0.4 * torch.randn(n, 3, generator=g)
    raw_opac = 2 * torch.randn(n, generator=g)
    K_ = (sh_degree + 1) ** 2
    sh0 = 0.5 * torch.randn(n, 1, 3, generator=g)
    shN = 0.1 * torch.randn(n, K_ - 1, 3, generator=g)
    viewmats = orbit_cameras(n_views)
    Ks = torch.tensor([[focal, 0, width / 2], [0, focal, height / 2], [0, 0, 1]], dtype=torch.float32).repeat(n_views, 1, 1)
    return Scene(name, width, height, sh_degree, means, quats, raw_scales, raw_opac, sh0, shN, viewmats, Ks)


def syn_b(seed: int = 42, n: int = 1_000_000, n_views: int = 64) -> Scene:
    """config 2: 1M Gaussians, 1080p, SH degree 3, 64 orbit cameras."""
    return _syn_box("SYN-B", seed, n, 1920, 1080, 1200.0, n_views)


def syn_b_flat(seed: int = 42, n: int = 1_000_000, n_views: int = 64, max_aspect: float = 100.0) -> Scene:
    """SYN-B with the shape statistics of a TRAINED scene (round 5): every Gaussian is a flat disk - one random axis is `aspect` times thinner than SYN-B's draw, aspect
    log-uniform in [1, max_aspect]. SYN-B itself draws its three log-scales from one sigma = 0.4 normal (aspect <= 6 for all but a handful): the regime in which the backward's
    foot-vector form lost the position / rotation gradients before LFS_BWD_REORTH. Same means / colours / opacities / cameras as SYN-B."""
    sc = _syn_box("SYN-B-flat", seed, n, 1920, 1080, 1200.0, n_views)
    g = torch.Generator().manual_seed(seed + 1000)
    thin = torch.randint(0, 3, (n,), generator=g)
    aspect = torch.exp(torch.rand(n, generator=g) * math.log(max_aspect))
    sc.raw_scales[torch.arange(n), thin] -= aspect.log()
    sc.extra["aspect"] = aspect
    return sc


def syn_c(seed: int = 42, n: int = 3_000_000, n_views: int = 64) -> Scene:
    """config 4: 3M Gaussians, 1600x1200, 64 views/step across 8 GPUs."""
    return _syn_box("SYN-C", seed, n, 1600, 1200, 1000.0, n_views)


def syn_d(seed: int = 42, n: int = 2_000_000, n_views: int = 64) -> Scene:
    """config 5: 2M Gaussians (MCMC preset), 1080p."""
    return _syn_box("SYN-D", seed, n, 1920, 1080, 1200.0, n_views)


def target_image(height: int, width: int, seed: int = 43) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    return torch.rand(3, height, width, generator=g)
