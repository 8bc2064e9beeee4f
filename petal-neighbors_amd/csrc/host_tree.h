// host_tree.h -- interface of tree.cpp (host-only: no HIP header, so that tree.cpp also builds stand-alone with
// -fsanitize=address,undefined for tests/test_product_host_sanitizers.py).
#pragma once
#include <cstddef>
#include <cstdint>

// ---- tree.cpp: the reference's ball tree, built on the host only when its introspection API is used
namespace pn {
struct HostTree;
// metric: 0 Euclidean, 1 Cosine (Node::init and node_distance_lower_bound use the tree's metric, src/ball_tree.rs:309, 459)
HostTree *host_tree_build(const void *pts, size_t n, size_t dim, int elem_bytes, int metric);  // nullptr: out of memory
void host_tree_free(HostTree *h);
size_t host_tree_num_nodes(const HostTree *h);
const uint64_t *host_tree_idx(const HostTree *h);
void host_tree_node(const HostTree *h, size_t node, uint64_t *start, uint64_t *end, int *is_leaf);
double host_tree_radius(const HostTree *h, size_t node);
const void *host_tree_centroid(const HostTree *h, size_t node);
int host_tree_compare(const HostTree *h, size_t x, size_t y);
double host_tree_lower_bound(const HostTree *h, size_t n1, size_t n2);
}  // namespace pn
