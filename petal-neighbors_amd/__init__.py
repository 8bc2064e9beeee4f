"""petal_neighbors_amd -- MI355X-native drop-in for petal-neighbors' exact k-NN hot path.

Mirrors the reference's public surface for that path (reference src/lib.rs:1-16):

    from petal_neighbors_amd import BallTree, ArrayError, distance
    tree = BallTree.euclidean(points)          # src/ball_tree.rs:367
    idx, dist = tree.query(point, k)           # src/ball_tree.rs:102
    idx = tree.query_radius(point, r)          # src/ball_tree.rs:137
    i, d = tree.query_nearest(point)           # src/ball_tree.rs:80

All compute runs in hand-written HIP kernels behind the C ABI in
``include/petal_mi355x.h`` (``libpetal_mi355x.so``); importing this package
without that library raises -- there is no CPU fallback.
"""
from . import _lib
from ._lib import LibraryMissing

_lib.lib()  # fail loudly, at import, if the HIP library is missing

from . import distance  # noqa: E402
from .ball_tree import BallTree  # noqa: E402
from .errors import ArrayError, PetalError  # noqa: E402
from .sharded import ShardedBallTree, ShardedIndex  # noqa: E402
from .vantage_point_tree import VantagePointTree  # noqa: E402

__all__ = ["BallTree", "VantagePointTree", "ShardedBallTree", "ShardedIndex", "ArrayError", "PetalError", "LibraryMissing", "distance"]
