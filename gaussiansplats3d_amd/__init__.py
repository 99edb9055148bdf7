"""gaussiansplats3d_amd — MI355X-native sort-and-rasterize engine behind the GaussianSplats3D seams.

Only the hot path lives here: ``csrc/`` (hand-written gfx950 HIP kernels + the C ABI of include/gsplat_hip.h)
and thin host-side mirrors of the two reference interfaces it replaces (``sort_worker``, ``splat_mesh``).
"""
from . import camera, scenes, util  # noqa: F401
from ._lib import Context, GsError, build, load  # noqa: F401
from .sort_worker import SortWorker, create_sort_worker  # noqa: F401
from .splat_mesh import SplatMesh  # noqa: F401
from .splat_tree import SortScheduler, SplatTree  # noqa: F401
