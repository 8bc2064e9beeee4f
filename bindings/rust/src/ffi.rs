//! UNVERIFIED SOURCE (no rustc in the build image).
//! What `bindgen` emits from `include/petal_mi355x.h` (ABI version 2), trimmed to what `lib.rs` calls.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
pub struct pn_index {
    _private: [u8; 0],
}
#[repr(C)]
pub struct pn_sharded {
    _private: [u8; 0],
}

pub const PN_OK: c_int = 0;
pub const PN_ERR_EMPTY: c_int = 1; // ArrayError::Empty          src/lib.rs:12
pub const PN_ERR_NOT_CONTIGUOUS: c_int = 2; // ArrayError::NotContiguous  src/lib.rs:14
pub const PN_ERR_COMM: c_int = 8;

extern "C" {
    pub fn pn_last_error() -> *const c_char;
    pub fn pn_index_destroy(index: *mut pn_index);
    pub fn pn_free(p: *mut c_void);

    pub fn pn_index_create_f32(points: *const f32, n_rows: usize, n_cols: usize, row_stride: isize, col_stride: isize,
                               device: c_int, out: *mut *mut pn_index) -> c_int;
    pub fn pn_index_create_f64(points: *const f64, n_rows: usize, n_cols: usize, row_stride: isize, col_stride: isize,
                               device: c_int, out: *mut *mut pn_index) -> c_int;
    pub fn pn_index_create_cosine_f32(points: *const f32, n_rows: usize, n_cols: usize, row_stride: isize,
                                      col_stride: isize, device: c_int, out: *mut *mut pn_index) -> c_int;
    pub fn pn_index_create_cosine_f64(points: *const f64, n_rows: usize, n_cols: usize, row_stride: isize,
                                      col_stride: isize, device: c_int, out: *mut *mut pn_index) -> c_int;

    pub fn pn_query_f32(index: *const pn_index, queries: *const f32, nq: usize, q_cols: usize, q_row_stride: isize,
                        k: usize, idx_out: *mut u64, dist_out: *mut f32) -> c_int;
    pub fn pn_query_f64(index: *const pn_index, queries: *const f64, nq: usize, q_cols: usize, q_row_stride: isize,
                        k: usize, idx_out: *mut u64, dist_out: *mut f64) -> c_int;
    pub fn pn_query_radius_f32(index: *const pn_index, queries: *const f32, nq: usize, q_cols: usize,
                               q_row_stride: isize, radius: f32, offsets: *mut u64, idx_out: *mut *mut u64) -> c_int;
    pub fn pn_query_radius_f64(index: *const pn_index, queries: *const f64, nq: usize, q_cols: usize,
                               q_row_stride: isize, radius: f64, offsets: *mut u64, idx_out: *mut *mut u64) -> c_int;

    pub fn pn_pairwise_f32(x: *const f32, n_rows: usize, n_cols: usize, row_stride: isize, device: c_int,
                           out: *mut f32) -> c_int;
    pub fn pn_pairwise_f64(x: *const f64, n_rows: usize, n_cols: usize, row_stride: isize, device: c_int,
                           out: *mut f64) -> c_int;
    pub fn pn_pairwise_cosine_f32(x: *const f32, n_rows: usize, n_cols: usize, row_stride: isize, device: c_int,
                                  out: *mut f32) -> c_int;
    pub fn pn_pairwise_cosine_f64(x: *const f64, n_rows: usize, n_cols: usize, row_stride: isize, device: c_int,
                                  out: *mut f64) -> c_int;
    pub fn pn_euclidean_f32(a: *const f32, b: *const f32, len: usize) -> f32;
    pub fn pn_euclidean_f64(a: *const f64, b: *const f64, len: usize) -> f64;
    pub fn pn_reuclidean_f32(a: *const f32, b: *const f32, len: usize) -> f32;
    pub fn pn_reuclidean_f64(a: *const f64, b: *const f64, len: usize) -> f64;
    pub fn pn_cosine_f32(a: *const f32, len_a: usize, b: *const f32, len_b: usize) -> f32;
    pub fn pn_cosine_f64(a: *const f64, len_a: usize, b: *const f64, len_b: usize) -> f64;

    // tree introspection (src/ball_tree.rs:296-353)
    pub fn pn_tree_num_nodes(index: *const pn_index, out: *mut u64) -> c_int;
    pub fn pn_tree_children_of(index: *const pn_index, node: u64, is_some: *mut c_int, left: *mut u64,
                               right: *mut u64) -> c_int;
    pub fn pn_tree_points_of(index: *const pn_index, node: u64, idx: *mut *const u64, count: *mut u64) -> c_int;
    pub fn pn_tree_radius_of_f32(index: *const pn_index, node: u64, out: *mut f32) -> c_int;
    pub fn pn_tree_radius_of_f64(index: *const pn_index, node: u64, out: *mut f64) -> c_int;
    pub fn pn_tree_compare_nodes(index: *const pn_index, x: u64, y: u64, ordering: *mut c_int) -> c_int;
    pub fn pn_tree_node_distance_lower_bound_f32(index: *const pn_index, n1: u64, n2: u64, out: *mut f32) -> c_int;
    pub fn pn_tree_node_distance_lower_bound_f64(index: *const pn_index, n1: u64, n2: u64, out: *mut f64) -> c_int;

    // row shards over several GPUs, one RCCL all-gather per batch behind the ABI
    pub fn pn_sharded_create_f32(points: *const f32, n_rows: usize, n_cols: usize, row_stride: isize,
                                 col_stride: isize, devices: *const c_int, n_devices: c_int,
                                 out: *mut *mut pn_sharded) -> c_int;
    pub fn pn_sharded_destroy(sharded: *mut pn_sharded);
    pub fn pn_sharded_query_f32(sharded: *const pn_sharded, queries: *const f32, nq: usize, q_cols: usize,
                                q_row_stride: isize, k: usize, idx_out: *mut u64, dist_out: *mut f32) -> c_int;
    pub fn pn_sharded_query_radius_f32(sharded: *const pn_sharded, queries: *const f32, nq: usize, q_cols: usize,
                                       q_row_stride: isize, radius: f32, offsets: *mut u64,
                                       idx_out: *mut *mut u64) -> c_int;
    // the same over an f64 corpus
    pub fn pn_sharded_create_f64(points: *const f64, n_rows: usize, n_cols: usize, row_stride: isize,
                                 col_stride: isize, devices: *const c_int, n_devices: c_int,
                                 out: *mut *mut pn_sharded) -> c_int;
    pub fn pn_sharded_query_f64(sharded: *const pn_sharded, queries: *const f64, nq: usize, q_cols: usize,
                                q_row_stride: isize, k: usize, idx_out: *mut u64, dist_out: *mut f64) -> c_int;
    pub fn pn_sharded_query_radius_f64(sharded: *const pn_sharded, queries: *const f64, nq: usize, q_cols: usize,
                                       q_row_stride: isize, radius: f64, offsets: *mut u64,
                                       idx_out: *mut *mut u64) -> c_int;
}
