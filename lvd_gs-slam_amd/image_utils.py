"""``psnr`` / ``mse`` / ``mkdir_p`` under the names ``utils/eval_utils_0806.py:27,29`` imports from the absent
``gaussian_splatting.utils.image_utils`` / ``system_utils`` (published 3DGS helpers: per-image mean over all
but the batch dimension, peak 1.0)."""
import os

import torch


def mse(img1, img2):
    return ((img1 - img2) ** 2).view(img1.shape[0], -1).mean(1, keepdim=True)


def psnr(img1, img2):
    return 20 * torch.log10(1.0 / torch.sqrt(mse(img1, img2)))


def mkdir_p(folder_path):
    os.makedirs(folder_path, exist_ok=True)
