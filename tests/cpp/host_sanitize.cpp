// host_sanitize.cpp -- driver of tests/test_product_host_sanitizers.py: the product's host-only units (csrc/tree.cpp,
// csrc/metric.cpp) compiled stand-alone with -fsanitize=address,undefined and run on a corpus read from a file.
//   in : u32 elem_bytes, u32 metric, u64 n, u64 dim, then n * dim elements (row-major)
//   out: u64 num_nodes, idx[n] (u64), then per node {u64 start, u64 end, u64 is_leaf, f64 radius, dim elements centroid};
//        then f64 {euclidean, reuclidean, cosine}(row 0, row n-1), f64 lower_bound(0, num_nodes - 1), i64 compare(0, 1 or 0)
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/petal_mi355x.h"
#include "../../petal-neighbors_amd/csrc/host_tree.h"

int main(int argc, char **argv) {
    if (argc != 3) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 3;
    uint32_t eb = 0, metric = 0;
    uint64_t n = 0, dim = 0;
    if (fread(&eb, 4, 1, f) != 1 || fread(&metric, 4, 1, f) != 1 || fread(&n, 8, 1, f) != 1 || fread(&dim, 8, 1, f) != 1) return 4;
    std::vector<unsigned char> pts((size_t)(n * dim * eb));
    if (!pts.empty() && fread(pts.data(), 1, pts.size(), f) != pts.size()) return 5;
    fclose(f);
    pn::HostTree *t = pn::host_tree_build(pts.data(), (size_t)n, (size_t)dim, (int)eb, (int)metric);
    if (!t) return 6;
    FILE *o = fopen(argv[2], "wb");
    if (!o) return 7;
    const uint64_t nn = pn::host_tree_num_nodes(t);
    fwrite(&nn, 8, 1, o);
    fwrite(pn::host_tree_idx(t), 8, (size_t)n, o);
    for (uint64_t i = 0; i < nn; ++i) {
        uint64_t s = 0, e = 0, leaf = 0;
        int il = 0;
        pn::host_tree_node(t, (size_t)i, &s, &e, &il);
        leaf = (uint64_t)il;
        const double r = pn::host_tree_radius(t, (size_t)i);
        fwrite(&s, 8, 1, o);
        fwrite(&e, 8, 1, o);
        fwrite(&leaf, 8, 1, o);
        fwrite(&r, 8, 1, o);
        fwrite(pn::host_tree_centroid(t, (size_t)i), eb, (size_t)dim, o);
    }
    double m[3] = {0, 0, 0};
    if (n && dim) {
        const unsigned char *a = pts.data(), *b = pts.data() + (size_t)((n - 1) * dim * eb);
        if (eb == 4) {
            m[0] = pn_euclidean_f32((const float *)a, (const float *)b, (size_t)dim);
            m[1] = pn_reuclidean_f32((const float *)a, (const float *)b, (size_t)dim);
            m[2] = pn_cosine_f32((const float *)a, (size_t)dim, (const float *)b, (size_t)dim);
        } else {
            m[0] = pn_euclidean_f64((const double *)a, (const double *)b, (size_t)dim);
            m[1] = pn_reuclidean_f64((const double *)a, (const double *)b, (size_t)dim);
            m[2] = pn_cosine_f64((const double *)a, (size_t)dim, (const double *)b, (size_t)dim);
        }
    }
    fwrite(m, 8, 3, o);
    const double lb = pn::host_tree_lower_bound(t, 0, (size_t)(nn - 1));
    const int64_t cmp = pn::host_tree_compare(t, 0, nn > 1 ? 1 : 0);
    fwrite(&lb, 8, 1, o);
    fwrite(&cmp, 8, 1, o);
    fclose(o);
    pn::host_tree_free(t);
    return 0;
}
