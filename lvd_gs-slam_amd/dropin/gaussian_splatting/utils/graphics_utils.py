from lvdgs.graphics_utils import (focal2fov, fov2focal, getProjectionMatrix, getProjectionMatrix2,  # noqa: F401
                                  getWorld2View2)
