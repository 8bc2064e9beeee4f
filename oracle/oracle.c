/*
 * oracle.c -- CPU oracle for the petal-neighbors hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of petal-neighbors 0.18.0 (/root/reference):
 *   src/distance.rs:21-74          Euclidean {distance, rdistance, conversions}, pairwise
 *   src/ball_tree.rs:38-63         BallTree::new  (validation + implicit-tree sizing)
 *   src/ball_tree.rs:80-86,149-196 query_nearest
 *   src/ball_tree.rs:102-121,203-243 query (k-NN, bounded max-heap)
 *   src/ball_tree.rs:137-142,250-294 query_radius
 *   src/ball_tree.rs:377-423       Neighbor ordering (distance only, OrderedFloat)
 *   src/ball_tree.rs:445-481       Node::init / distance bounds
 *   src/ball_tree.rs:504-613       build_subtree / halve_node_indices / max_spread_column
 * Each function in oracle_impl.h cites the lines it follows.
 *
 * PINNING: the reference is Rust and no rustc/cargo exists in this image, so
 * oracle/_ref cannot be built.  The oracle is pinned against every golden
 * vector the reference's own tests and doc-tests hold for this path
 * (tests/golden/reference_kats.json) and against the
 * reference's property test pattern (tree == naive scan).  What those tests
 * do NOT pin -- the index order inside groups of exactly equal distances,
 * which depends on Rust's std BinaryHeap (restated here from memory) -- is
 * "parity unpinned" and is compared as a set by the tests.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product (libpetal_mi355x.so) never does.
 *
 * Build: see Makefile (-O2 -ffp-contract=off, no fast-math, no -march=native).
 */
#include <math.h>
#include <pthread.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

#define T float
#define SFX f32
#define SQRT sqrtf
#define FMAX fmaxf
#define FMIN fminf
#include "oracle_impl.h"
#undef T
#undef SFX
#undef SQRT
#undef FMAX
#undef FMIN

#define T double
#define SFX f64
#define SQRT sqrt
#define FMAX fmax
#define FMIN fmin
#include "oracle_impl.h"
#undef T
#undef SFX
#undef SQRT
#undef FMAX
#undef FMIN

/* Synthetic-input generator shared (by restatement, not by linking) with the
 * device generator in petal-neighbors_amd/csrc: uniform [0,1) f32 with exactly
 * 24 random bits from a counter-based hash. */
static inline uint32_t mix32(uint64_t seed, uint64_t ctr)
{
    uint64_t z = seed * 0x9E3779B97F4A7C15ull + ctr;
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (uint32_t)(z >> 32);
}

void oracle_fill_uniform_f32(float *out, uint64_t count, uint64_t seed, uint64_t first_ctr)
{
    for (uint64_t i = 0; i < count; ++i)
        out[i] = (float)(mix32(seed, first_ctr + i) >> 8) * (1.0f / 16777216.0f);
}

void oracle_fill_uniform_f64(double *out, uint64_t count, uint64_t seed, uint64_t first_ctr)
{
    for (uint64_t i = 0; i < count; ++i)
        out[i] = (double)(mix32(seed, first_ctr + i) >> 8) * (1.0 / 16777216.0);
}
