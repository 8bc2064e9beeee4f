// The reference's own unit tests for the hot path (src/ball_tree.rs:623-782, src/distance.rs:129-141),
// restated against the C++ mirror.  Exit code 0 = all passed.  Usage: host_mirror [cpu|gpu]
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "petal_neighbors.hpp"

static int fails = 0;
#define EXPECT(c) do { if (!(c)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); ++fails; } } while (0)

int main(int argc, char **argv) {
    const bool gpu = argc > 1 && std::string(argv[1]) == "gpu";
    using petal::ArrayError;
    using Tree = petal::BallTree<double>;
    // ball_tree_empty (src/ball_tree.rs:623-630)
    try { Tree::euclidean(nullptr, 0, 0); EXPECT(false); }
    catch (const ArrayError &e) { EXPECT(e.kind == ArrayError::Empty && std::string(e.what()) == "array is empty"); }
    // ball_tree_column_base (:632-638): reversed_axes() of a 3x2 array = 2x3 with strides (1, 2)
    const double p3[] = {1., 1., 1., 1.1, 9., 9.};
    try { Tree::euclidean(p3, 2, 3, 1, 2); EXPECT(false); }
    catch (const ArrayError &e) { EXPECT(e.kind == ArrayError::NotContiguous); }
    // metric (host scalar): pairwise golden [[3,4],[0,0]] -> 5
    petal::distance::Euclidean m;
    const double a[] = {3., 4.}, b[] = {0., 0.};
    EXPECT(m.distance(a, b, 2) == 5.0 && m.rdistance(a, b, 2) == 25.0);
    EXPECT(m == petal::distance::Euclidean{});  // ball_tree_metric (:640-647)
    if (!gpu) {
        // no CPU fallback: construction of a valid array must fail loudly without a GPU
        int ndev = 0;
        if (pn_device_count(&ndev) != PN_OK || ndev == 0) {
            try { Tree::euclidean(p3, 3, 2); EXPECT(false); }
            catch (const petal::DeviceError &e) { EXPECT(e.code == PN_ERR_DEVICE); }
        }
        std::printf("%s (cpu part)\n", fails ? "FAILED" : "ok");
        return fails;
    }
    // ball_tree_3 (:649-698)
    Tree tree = Tree::euclidean(p3, 3, 2);
    const double q0[] = {0., 0.};
    auto nn = tree.query_nearest(q0, 2);
    EXPECT(nn.first == 0 && std::fabs(nn.second - std::sqrt(2.0)) <= 2.3e-16);
    auto r0 = tree.query(q0, 2, 0);
    EXPECT(r0.first.empty() && r0.second.empty());
    auto r1 = tree.query(q0, 2, 1);
    EXPECT(r1.first.size() == 1 && r1.first[0] == 0 && r1.second[0] == nn.second);
    auto rad = tree.query_radius(q0, 2, 2.0);
    EXPECT(rad.size() == 2 && rad[0] == 0 && rad[1] == 1);
    const double q20[] = {20., 20.};
    EXPECT(tree.query_radius(q20, 2, 1.0).empty());
    const double q7[] = {7., 7.};
    nn = tree.query_nearest(q7, 2);
    EXPECT(nn.first == 2 && std::fabs(nn.second - std::sqrt(8.0)) <= 2.3e-16);
    // doc-test (:93-101): query([3,3], 2) -> [1, 0]
    const double pd[] = {1., 1., 1., 2., 9., 9.};
    Tree t2 = Tree::euclidean(pd, 3, 2);
    const double q3[] = {3., 3.};
    auto r2 = t2.query(q3, 2, 2);
    EXPECT(r2.first.size() == 2 && r2.first[0] == 1 && r2.first[1] == 0);
    // ball_tree_query_radius (:767-782)
    const double line[] = {0., 2., 3., 4., 6., 8., 10.};
    Tree t3 = Tree::euclidean(line, 7, 1);
    const double q32[] = {3.2};
    auto rr = t3.query_radius(q32, 1, 1.0);
    EXPECT(rr.size() == 2 && rr[0] == 2 && rr[1] == 3);
    // pairwise (src/distance.rs:129-134), f32 on the GPU
    const float x[] = {3.f, 4.f, 0.f, 0.f};
    auto pw = petal::distance::pairwise(x, 2, 2);
    EXPECT(pw[0] == 0.f && pw[1] == 5.f && pw[2] == 5.f && pw[3] == 0.f);
    // f32 tree, k > n
    petal::BallTree<float> tf = petal::BallTree<float>::euclidean(x, 2, 2);
    auto rf = tf.query(x, 2, 9);
    EXPECT(rf.first.size() == 2 && rf.first[0] == 0 && rf.second[0] == 0.f && rf.second[1] == 5.f);
    // ---- VantagePointTree: euclidian (src/vantage_point_tree.rs:220-233)
    const double p6[] = {1., 2., 1.1, 2.2, 0.9, 1.9, 1., 2.1, -2., 3., -2.2, 3.1};
    auto vp = petal::VantagePointTree<double>::euclidean(p6, 6, 2);
    const double qv[] = {0.95, 1.96};
    EXPECT(vp.query_nearest(qv, 2).first == 0);
    try { petal::VantagePointTree<double>::euclidean(nullptr, 0, 2); EXPECT(false); }
    catch (const ArrayError &e) { EXPECT(e.kind == ArrayError::Empty); }
    // ---- tree introspection (src/ball_tree.rs:296-353) on node_init's vector (:784-798): centroid [0,4], radius 5
    const double pn3[] = {0., 1., 0., 9., 0., 2.};
    Tree ti = Tree::euclidean(pn3, 3, 2);
    EXPECT(ti.num_nodes() == 3 && ti.radius_of(0) == 5.0);
    EXPECT(ti.children_of(0).some && ti.children_of(0).left == 1 && ti.children_of(0).right == 2 && !ti.children_of(1).some);
    EXPECT(ti.points_of(0).size() == 3 && ti.points_of(1).size() == 1 && ti.points_of(2).size() == 2);
    EXPECT(ti.compare_nodes(0, 1) == 1 && ti.compare_nodes(1, 1) == 0 && ti.node_distance_lower_bound(0, 0) == 0.0);
    try { ti.radius_of(3); EXPECT(false); }
    catch (const petal::DeviceError &e) { EXPECT(e.code == PN_ERR_INVALID); }
    // ---- BallTree::new(points, Cosine): cosine (src/distance.rs:143-182) neighbours by an exact scan
    const double pc[] = {1., 0., 0., 1., 1., 1., -1., 0.};
    auto tc = petal::BallTree<double, petal::distance::Cosine>::create(pc, 4, 2);
    const double qc[] = {2., 0.1};
    auto rc = tc.query(qc, 2, 4);
    petal::distance::Cosine cm;
    EXPECT(rc.first.size() == 4 && rc.first[0] == 0 && rc.first[1] == 2 && rc.first[3] == 3);
    for (size_t j = 0; j < 4; ++j) EXPECT(rc.second[j] == cm.distance(qc, pc + 2 * rc.first[j], 2));
    // ---- row shards through the ABI: three virtual shards on GPU 0, one RCCL all-gather, merged = unsharded
    {
        std::vector<float> big(3000 * 8), qs(8);
        unsigned s = 12345u;
        for (float &v : big) { s = s * 1664525u + 1013904223u; v = (float)(s >> 8) * (1.0f / 16777216.0f); }
        for (float &v : qs) { s = s * 1664525u + 1013904223u; v = (float)(s >> 8) * (1.0f / 16777216.0f); }
        petal::ShardedBallTree sh(big.data(), 3000, 8, {0, 0, 0});
        petal::BallTree<float> one = petal::BallTree<float>::euclidean(big.data(), 3000, 8);
        auto a1 = sh.query(qs.data(), 8, 7);
        auto a2 = one.query(qs.data(), 8, 7);
        EXPECT(a1.first == a2.first && a1.second == a2.second);
        EXPECT(sh.query_radius(qs.data(), 8, a2.second[3]) == one.query_radius(qs.data(), 8, a2.second[3]));
    }
    {   // the same over an f64 corpus (BallTree<f64, Euclidean>)
        std::vector<double> big(3000 * 8), qs(8);
        unsigned long long s = 98765ull;
        for (double &v : big) { s = s * 6364136223846793005ull + 1442695040888963407ull; v = (double)(s >> 11) * (1.0 / 9007199254740992.0); }
        for (double &v : qs) { s = s * 6364136223846793005ull + 1442695040888963407ull; v = (double)(s >> 11) * (1.0 / 9007199254740992.0); }
        petal::ShardedBallTreeT<double> sh(big.data(), 3000, 8, {0, 0, 0});
        petal::BallTree<double> one = petal::BallTree<double>::euclidean(big.data(), 3000, 8);
        auto a1 = sh.query(qs.data(), 8, 7);
        auto a2 = one.query(qs.data(), 8, 7);
        EXPECT(a1.first == a2.first && a1.second == a2.second);
        EXPECT(sh.query_radius(qs.data(), 8, a2.second[3]) == one.query_radius(qs.data(), 8, a2.second[3]));
    }
    std::printf("%s\n", fails ? "FAILED" : "ok");
    return fails;
}
