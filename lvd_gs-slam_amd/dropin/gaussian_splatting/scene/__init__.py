from lvdgs.gaussian_model import GaussianModel  # noqa: F401
