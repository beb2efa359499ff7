"""
mqslam_amd -- MI355X (gfx950) native hot path of Multiple-Quadrotor-SLAM.

The directory is named `multiple-quadrotor-slam_amd` (not an importable identifier); import it
through the repo-root shim:  `import mqslam_amd`.
"""
from . import _lib                       # noqa: F401
from . import triangulation_c            # noqa: F401
from . import triangulation              # noqa: F401
from . import synthetic                  # noqa: F401
from . import device                     # noqa: F401
from . import bundle_adjustment          # noqa: F401
from . import matching                   # noqa: F401
from . import sharding                   # noqa: F401
from . import camera                     # noqa: F401
from . import ba_io                      # noqa: F401
from . import sparse_ba                  # noqa: F401
from . import pnp                        # noqa: F401
from . import features                   # noqa: F401
from . import slam_replay                # noqa: F401
from . import slam_loop                  # noqa: F401
from . import slam_device                # noqa: F401
from . import slam_frontend              # noqa: F401
from . import triangulation_comparison   # noqa: F401

loaded = _lib.loaded
