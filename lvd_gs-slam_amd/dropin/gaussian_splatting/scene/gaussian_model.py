from lvdgs.gaussian_model import GaussianModel, build_rotation, get_expon_lr_func, inverse_sigmoid  # noqa: F401
