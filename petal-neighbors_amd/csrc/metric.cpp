// metric.cpp -- scalar Metric<A> for Euclidean (reference src/distance.rs:21-55).
// Host-side BY DESIGN: the reference's Metric::distance is a one-pair scalar
// call; the batched forms (query, pairwise) run on the GPU.  Sequential,
// unfused fold + correctly rounded sqrt: bit-identical to the device kernels.
// Built with -ffp-contract=off (and the pragma below) so no FMA can appear.
#include <cmath>
#include <cstddef>

#include "../../include/petal_mi355x.h"

#pragma STDC FP_CONTRACT OFF

template <typename T>
static inline T fold(const T *a, const T *b, size_t len) {
    T sum = (T)0;
    for (size_t i = 0; i < len; ++i) {
        const T diff = a[i] - b[i];
        const T sq = diff * diff;
        sum = sum + sq;
    }
    return sum;
}

extern "C" float pn_reuclidean_f32(const float *a, const float *b, size_t len) { return fold<float>(a, b, len); }
extern "C" double pn_reuclidean_f64(const double *a, const double *b, size_t len) { return fold<double>(a, b, len); }
extern "C" float pn_euclidean_f32(const float *a, const float *b, size_t len) { return sqrtf(fold<float>(a, b, len)); }
extern "C" double pn_euclidean_f64(const double *a, const double *b, size_t len) { return sqrt(fold<double>(a, b, len)); }
extern "C" float pn_rdistance_to_distance_f32(float d) { return sqrtf(d); }    // src/distance.rs:47-49
extern "C" double pn_rdistance_to_distance_f64(double d) { return sqrt(d); }
extern "C" float pn_distance_to_rdistance_f32(float d) { return d * d; }       // powi(2), src/distance.rs:52-54
extern "C" double pn_distance_to_rdistance_f64(double d) { return d * d; }

// Cosine (reference src/distance.rs:76-122): 1 - dot / (|a| |b|), three sequential iterator sums; the dot product
// zips the vectors (shorter length), each norm runs over its own vector.  rdistance and the conversions are the
// identity in the reference (:109-121), so there is nothing else to export.
template <typename T>
static inline T cosine(const T *a, size_t la, const T *b, size_t lb) {
    const size_t lz = la < lb ? la : lb;
    T dot = (T)0, n1 = (T)0, n2 = (T)0;
    for (size_t i = 0; i < lz; ++i) {
        const T pr = a[i] * b[i];
        dot = dot + pr;
    }
    for (size_t i = 0; i < la; ++i) {
        const T pr = a[i] * a[i];
        n1 = n1 + pr;
    }
    for (size_t i = 0; i < lb; ++i) {
        const T pr = b[i] * b[i];
        n2 = n2 + pr;
    }
    const T den = std::sqrt(n1) * std::sqrt(n2);
    return (T)1 - dot / den;
}
extern "C" float pn_cosine_f32(const float *a, size_t len_a, const float *b, size_t len_b) {
    return cosine<float>(a, len_a, b, len_b);
}
extern "C" double pn_cosine_f64(const double *a, size_t len_a, const double *b, size_t len_b) {
    return cosine<double>(a, len_a, b, len_b);
}
