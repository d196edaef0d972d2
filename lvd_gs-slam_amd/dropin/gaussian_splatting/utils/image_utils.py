from lvdgs.image_utils import mse, psnr  # noqa: F401
