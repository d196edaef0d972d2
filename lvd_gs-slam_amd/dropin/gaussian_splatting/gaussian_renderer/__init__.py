from lvdgs.gaussian_renderer import render, render_with_custom_resolution  # noqa: F401
