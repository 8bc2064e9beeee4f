"""Tree introspection API (src/ball_tree.rs:296-353) against the oracle's faithful tree, node by node.

The product's tree (csrc/tree.cpp, built lazily on the host from the index's own copy of the points) and the oracle's
(oracle/oracle_impl.h) are two independent restatements of the reference's build; every node must agree in range,
permutation slice, leaf flag, centroid (bit for bit), radius (bit for bit), and the derived accessors must follow."""
import numpy as np
import pytest

from conftest import uniform

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n,dim", [(1, 3), (2, 2), (3, 1), (7, 4), (8, 4), (100, 3), (1000, 10), (4097, 17)])
def test_every_node_matches_the_oracle_tree(pn, oracle_mod, dtype, n, dim):
    pts = uniform((n, dim), 500 + n + dim, dtype)
    if n >= 8:
        pts[5] = pts[1]  # equal coordinates: the quick-select's tie handling
    tree = pn.BallTree.euclidean(pts)
    ref = oracle_mod.Tree(pts)
    nn = tree.num_nodes()
    assert nn == ref.num_nodes == (1 << n.bit_length()) - 1
    ref_idx = ref.idx
    for i in range(nn):
        node = ref.node(i)
        s, e = node["range"]
        assert np.array_equal(tree.points_of(i), ref_idx[s:e].astype(np.uint64)), i
        assert tree.radius_of(i).tobytes() == dtype(node["radius"]).tobytes(), i
        assert tree._centroid_of(i).tobytes() == node["centroid"].tobytes(), i
        kids = tree.children_of(i)
        assert (kids is None) == node["is_leaf"], i
        if kids is not None:
            assert kids == (2 * i + 1, 2 * i + 2)
    # derived accessors on sampled node pairs
    rng = np.random.default_rng(n)
    m = pn.distance.Euclidean()
    for _ in range(min(200, nn * nn)):
        a, b = int(rng.integers(nn)), int(rng.integers(nn))
        ra, rb = tree.radius_of(a), tree.radius_of(b)
        want = m.distance(tree._centroid_of(a), tree._centroid_of(b)) - ra - rb
        want = dtype(0) if want < 0 else want
        assert tree.node_distance_lower_bound(a, b).tobytes() == dtype(want).tobytes()
        assert tree.compare_nodes(a, b) == int(ra > rb) - int(ra < rb)
    with pytest.raises(IndexError):
        tree.radius_of(nn)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n,dim", [(1, 3), (2, 2), (7, 4), (100, 3), (1000, 10), (4097, 17)])
def test_cosine_tree_nodes_use_the_cosine_metric(pn, oracle_mod, dtype, n, dim):
    """BallTree::new(points, Cosine): Node::init and node_distance_lower_bound call metric.distance
    (src/ball_tree.rs:309, 459), so radius_of / compare_nodes / node_distance_lower_bound are Cosine values; ranges,
    permutation and centroids do not depend on the metric.  Checked node by node against the oracle's tree built with
    the same metric."""
    pts = uniform((n, dim), 900 + n + dim, dtype) - dtype(0.25)  # mixed signs: cosine distances up to 2
    if n >= 8:
        pts[5] = pts[1]
    tree = pn.BallTree.new(pts, pn.distance.Cosine())
    ref = oracle_mod.Tree(pts, metric="cosine")
    euc = oracle_mod.Tree(pts)
    nn = tree.num_nodes()
    assert nn == ref.num_nodes
    ref_idx = ref.idx
    differs = 0
    for i in range(nn):
        node = ref.node(i)
        s, e = node["range"]
        assert np.array_equal(tree.points_of(i), ref_idx[s:e].astype(np.uint64)), i
        assert tree._centroid_of(i).tobytes() == node["centroid"].tobytes(), i
        got, want = tree.radius_of(i), dtype(node["radius"])
        assert got.tobytes() == want.tobytes() or (np.isnan(got) and np.isnan(want)), i
        differs += int(dtype(euc.node(i)["radius"]).tobytes() != want.tobytes())
    if n >= 100:
        assert differs > nn // 4  # the test would not notice a Euclidean tree otherwise
    rng = np.random.default_rng(n)
    m = pn.distance.Cosine()
    for _ in range(min(200, nn * nn)):
        a, b = int(rng.integers(nn)), int(rng.integers(nn))
        ra, rb = tree.radius_of(a), tree.radius_of(b)
        want = m.distance(tree._centroid_of(a), tree._centroid_of(b)) - ra - rb
        want = dtype(0) if want < 0 else want
        got = tree.node_distance_lower_bound(a, b)
        assert got.tobytes() == dtype(want).tobytes() or (np.isnan(got) and np.isnan(want))
        if not (np.isnan(ra) or np.isnan(rb)):
            assert tree.compare_nodes(a, b) == int(ra > rb) - int(ra < rb)


def test_reference_node_init_vector_and_degenerate_build(pn, kats):
    """G11 (src/ball_tree.rs:784-798): root of [[0,1],[0,9],[0,2]] has centroid [0,4], radius 5; and the 8 identical
    points of src/ball_tree.rs:718-740 build (the split degenerates, every radius is 0)."""
    t = pn.BallTree.euclidean(np.array([[0., 1.], [0., 9.], [0., 2.]]))
    assert t._centroid_of(0).tolist() == [0.0, 4.0] and t.radius_of(0) == 5.0
    assert t.num_nodes() == 3 and sorted(t.points_of(0).tolist()) == [0, 1, 2]
    same = pn.BallTree.euclidean(np.ones((8, 2)))
    assert same.num_nodes() == 15
    assert all(same.radius_of(i) == 0.0 for i in range(15))
    assert same.compare_nodes(0, 1) == 0 and same.node_distance_lower_bound(3, 9) == 0.0
    # queries still go through the scan engines, tree or no tree
    i, d = same.query(np.array([1., 2.]), 3)
    assert d.tolist() == [1.0, 1.0, 1.0]


def test_nan_radius_compares_as_none(pn):
    pts = np.array([[0., 0.], [np.nan, 1.], [2., 2.], [3., 3.]])
    t = pn.BallTree.euclidean(pts)
    rs = [t.radius_of(i) for i in range(t.num_nodes())]
    nan_nodes = [i for i, r in enumerate(rs) if np.isnan(r)]
    fin_nodes = [i for i, r in enumerate(rs) if not np.isnan(r)]
    if nan_nodes and fin_nodes:
        assert t.compare_nodes(nan_nodes[0], fin_nodes[0]) is None
