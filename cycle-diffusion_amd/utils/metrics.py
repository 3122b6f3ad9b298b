"""Image metrics of the reference's evaluators (evaluation/utils.py:13-67; used by evaluation/translate_text.py):
PSNR on [0, 1] images, the 11x11 Gaussian-window SSIM on [0, 255] images ("the same outputs as MATLAB's"), and the
squared L2 distance. Host-side torch code - callers of the hot path, not part of it."""
import torch
import torch.nn.functional as F


def calculate_psnr(img1, img2):
    """img [3, H, W] in [0, 1] -> 10 log10(1 / mse); 100 for identical images (evaluation/utils.py:60-67)."""
    assert img1.shape == img2.shape
    assert (img1 >= 0).all() and (img1 <= 1).all() and (img2 >= 0).all() and (img2 <= 1).all()
    mse = ((img1 - img2) ** 2).mean()
    if mse == 0:
        return torch.tensor(100.0)
    return 10 * torch.log10(1 / mse)


def _gaussian_window(size=11, sigma=1.5, dtype=torch.float64):
    x = torch.arange(size, dtype=dtype) - (size - 1) / 2.0
    g = torch.exp(-(x ** 2) / (2 * sigma ** 2))
    g = g / g.sum()  # cv2.getGaussianKernel normalises to 1
    return torch.outer(g, g)


def ssim(img1, img2):
    """one channel, values in [0, 255]; valid 11x11 Gaussian windows (evaluation/utils.py:37-57: cv2.filter2D then
    [5:-5, 5:-5] = correlation over fully covered windows)."""
    assert img1.shape == img2.shape and img1.dim() == 2
    C1, C2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    a, b = img1.double()[None, None], img2.double()[None, None]
    w = _gaussian_window()[None, None]
    mu1, mu2 = F.conv2d(a, w), F.conv2d(b, w)
    s11 = F.conv2d(a * a, w) - mu1 ** 2
    s22 = F.conv2d(b * b, w) - mu2 ** 2
    s12 = F.conv2d(a * b, w) - mu1 * mu2
    m = ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 ** 2 + mu2 ** 2 + C1) * (s11 + s22 + C2))
    return m.mean()


def calculate_ssim(img1, img2):
    """img [H, W, 3] / [H, W, 1] / [H, W] in [0, 255]: mean SSIM over channels (evaluation/utils.py:13-34)."""
    if img1.shape != img2.shape:
        raise ValueError("Input images must have the same dimensions.")
    if img1.dim() == 2:
        return ssim(img1, img2)
    if img1.dim() == 3 and img1.shape[2] in (1, 3):
        return torch.stack([ssim(img1[:, :, i], img2[:, :, i]) for i in range(img1.shape[2])]).mean()
    raise ValueError("Wrong input image dimensions.")


def calculate_l2(img1, img2):
    """squared L2 distance of [0, 1] images, as accumulated by evaluation/translate_text.py"""
    return ((img1 - img2) ** 2).sum()
