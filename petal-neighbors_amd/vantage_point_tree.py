"""Façade for ``petal_neighbors::VantagePointTree`` (reference src/vantage_point_tree.rs:13-98).

The reference's second index answers only 1-NN (``query_nearest``, :88-98) and returns the same
neighbour as ``BallTree::query_nearest``; on the MI355X path both are the k = 1 case of the same
exact scan, so this class is a thin alias kept for API coverage.  Where several
points are at exactly the nearest distance the reference's choice depends on tree order; this
returns the smallest index.
"""
from __future__ import annotations

from .ball_tree import BallTree
from .distance import Euclidean


class VantagePointTree:
    def __init__(self, tree: BallTree):
        self._tree = tree
        self.points = tree.points
        self.metric = tree.metric

    @classmethod
    def new(cls, points, metric, device: int = 0):
        """``VantagePointTree::new(points, metric)`` (src/vantage_point_tree.rs:51-72): same
        ``ArrayError`` cases as ``BallTree::new``."""
        return cls(BallTree.new(points, metric, device))

    @classmethod
    def euclidean(cls, points, device: int = 0):
        """``VantagePointTree::euclidean(points)`` (src/vantage_point_tree.rs:31-44)."""
        return cls.new(points, Euclidean(), device)

    def query_nearest(self, point):
        """``query_nearest(point) -> (usize, A)`` (src/vantage_point_tree.rs:88-98)."""
        return self._tree.query_nearest(point)
