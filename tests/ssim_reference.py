"""Plain PyTorch fp32/fp64 reference of the fused SSIM op and the photometric loss (test infrastructure only).
Same definition as the reference kernel (src/training/kernels/ssim.cu): per-channel 11x11 separable Gaussian window
(sigma 1.5), zero padding, SSIM = ((2 mu1 mu2 + C1)(2 s12 + C2)) / ((mu1^2 + mu2^2 + C1)(s1 + s2 + C2)); "valid" crops 5 px."""
import torch

GAUSS = [0.001028380123898387, 0.0075987582094967365, 0.036000773310661316, 0.10936068743467331, 0.21300552785396576,
         0.26601171493530273, 0.21300552785396576, 0.10936068743467331, 0.036000773310661316, 0.0075987582094967365, 0.001028380123898387]


def _blur(x):  # x [B,C,H,W]
    C = x.shape[1]
    g = torch.tensor(GAUSS, dtype=x.dtype, device=x.device)
    kh = g.view(1, 1, 1, 11).repeat(C, 1, 1, 1)
    kv = g.view(1, 1, 11, 1).repeat(C, 1, 1, 1)
    x = torch.nn.functional.conv2d(x, kh, padding=(0, 5), groups=C)
    return torch.nn.functional.conv2d(x, kv, padding=(5, 0), groups=C)


def ssim_map(img1, img2, C1=0.01 ** 2, C2=0.03 ** 2):
    mu1, mu2 = _blur(img1), _blur(img2)
    s1 = _blur(img1 * img1) - mu1 * mu1
    s2 = _blur(img2 * img2) - mu2 * mu2
    s12 = _blur(img1 * img2) - mu1 * mu2
    return ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s1 + s2 + C2))


def fused_ssim(img1, img2, padding="same"):
    m = ssim_map(img1, img2)
    h, w = m.shape[2], m.shape[3]
    if padding == "valid" and h > 10 and w > 10:
        m = m[:, :, 5:h - 5, 5:w - 5]
    return m.mean()


def photometric_loss(rendered, gt, lambda_dssim=0.2):
    l1 = (rendered - gt).abs().mean()
    return (1 - lambda_dssim) * l1 + lambda_dssim * (1 - fused_ssim(rendered, gt, "valid"))
