from lvdgs.loss_utils import l1_dssim_loss, l1_loss, ssim  # noqa: F401
