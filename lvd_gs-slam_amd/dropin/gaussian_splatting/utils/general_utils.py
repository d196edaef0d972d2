from lvdgs.gaussian_model import build_rotation, get_expon_lr_func, inverse_sigmoid  # noqa: F401
