from lvdgs.image_utils import mkdir_p  # noqa: F401
