//! UNVERIFIED SOURCE (no rustc in the build image): the drop-in a petal-neighbors maintainer would add.
//!
//! Same names, signatures and error behaviour as petal-neighbors 0.18 for the hot path --
//! `BallTree::{new, euclidean, query, query_nearest, query_radius}` + the introspection accessors
//! (src/ball_tree.rs:15-374), `distance::{Euclidean, Cosine, pairwise}` (src/distance.rs), `ArrayError` (src/lib.rs:9-16)
//! -- forwarding to the C ABI of `libpetal_mi355x.so` (`include/petal_mi355x.h`).  The ndarray stays with the caller
//! (`CowArray`, borrowed or owned); the library keeps its own zero-padded copy in HBM.  There is no CPU fallback: a
//! device failure panics with the library's message.
pub mod ffi;

use ndarray::{Array2, ArrayBase, ArrayView1, ArrayView2, CowArray, Data, Ix1, Ix2};
use std::cmp::Ordering;
use std::ffi::CStr;
use std::os::raw::c_int;

/// `ArrayError` (src/lib.rs:9-16)
#[derive(Debug, thiserror::Error)]
pub enum ArrayError {
    #[error("array is empty")]
    Empty,
    #[error("array is not contiguous in memory")]
    NotContiguous,
}

fn last_error() -> String {
    unsafe { CStr::from_ptr(ffi::pn_last_error()) }.to_string_lossy().into_owned()
}
fn ok(rc: c_int) {
    assert_eq!(rc, ffi::PN_OK, "petal_mi355x: {}", last_error());
}

/// The two element types of the reference (`A: FloatCore`), each bound to its half of the ABI.
pub trait Elem: Copy + Default + PartialOrd + 'static {
    #[doc(hidden)]
    unsafe fn create(p: *const Self, n: usize, d: usize, rs: isize, cs: isize, cosine: bool, out: *mut *mut ffi::pn_index) -> c_int;
    #[doc(hidden)]
    unsafe fn query(ix: *const ffi::pn_index, q: *const Self, nq: usize, qc: usize, k: usize, i: *mut u64, d: *mut Self) -> c_int;
    #[doc(hidden)]
    unsafe fn radius(ix: *const ffi::pn_index, q: *const Self, qc: usize, r: Self, off: *mut u64, out: *mut *mut u64) -> c_int;
    #[doc(hidden)]
    unsafe fn radius_of(ix: *const ffi::pn_index, n: u64, out: *mut Self) -> c_int;
    #[doc(hidden)]
    unsafe fn lower_bound(ix: *const ffi::pn_index, a: u64, b: u64, out: *mut Self) -> c_int;
    #[doc(hidden)]
    unsafe fn euclid(a: *const Self, b: *const Self, n: usize, squared: bool) -> Self;
    #[doc(hidden)]
    unsafe fn cosine(a: *const Self, na: usize, b: *const Self, nb: usize) -> Self;
    #[doc(hidden)]
    unsafe fn pairwise(x: *const Self, n: usize, d: usize, rs: isize, cosine: bool, out: *mut Self) -> c_int;
}
macro_rules! impl_elem {
    ($t:ty, $create:ident, $create_cos:ident, $query:ident, $radius:ident, $rad_of:ident, $lb:ident, $eu:ident, $reu:ident,
     $cos:ident, $pw:ident, $pwc:ident) => {
        impl Elem for $t {
            unsafe fn create(p: *const Self, n: usize, d: usize, rs: isize, cs: isize, cosine: bool, out: *mut *mut ffi::pn_index) -> c_int {
                if cosine { ffi::$create_cos(p, n, d, rs, cs, 0, out) } else { ffi::$create(p, n, d, rs, cs, 0, out) }
            }
            unsafe fn query(ix: *const ffi::pn_index, q: *const Self, nq: usize, qc: usize, k: usize, i: *mut u64, d: *mut Self) -> c_int {
                ffi::$query(ix, q, nq, qc, qc as isize, k, i, d)
            }
            unsafe fn radius(ix: *const ffi::pn_index, q: *const Self, qc: usize, r: Self, off: *mut u64, out: *mut *mut u64) -> c_int {
                ffi::$radius(ix, q, 1, qc, qc as isize, r, off, out)
            }
            unsafe fn radius_of(ix: *const ffi::pn_index, n: u64, out: *mut Self) -> c_int { ffi::$rad_of(ix, n, out) }
            unsafe fn lower_bound(ix: *const ffi::pn_index, a: u64, b: u64, out: *mut Self) -> c_int { ffi::$lb(ix, a, b, out) }
            unsafe fn euclid(a: *const Self, b: *const Self, n: usize, squared: bool) -> Self {
                if squared { ffi::$reu(a, b, n) } else { ffi::$eu(a, b, n) }
            }
            unsafe fn cosine(a: *const Self, na: usize, b: *const Self, nb: usize) -> Self { ffi::$cos(a, na, b, nb) }
            unsafe fn pairwise(x: *const Self, n: usize, d: usize, rs: isize, cosine: bool, out: *mut Self) -> c_int {
                if cosine { ffi::$pwc(x, n, d, rs, 0, out) } else { ffi::$pw(x, n, d, rs, 0, out) }
            }
        }
    };
}
impl_elem!(f32, pn_index_create_f32, pn_index_create_cosine_f32, pn_query_f32, pn_query_radius_f32, pn_tree_radius_of_f32,
           pn_tree_node_distance_lower_bound_f32, pn_euclidean_f32, pn_reuclidean_f32, pn_cosine_f32, pn_pairwise_f32,
           pn_pairwise_cosine_f32);
impl_elem!(f64, pn_index_create_f64, pn_index_create_cosine_f64, pn_query_f64, pn_query_radius_f64, pn_tree_radius_of_f64,
           pn_tree_node_distance_lower_bound_f64, pn_euclidean_f64, pn_reuclidean_f64, pn_cosine_f64, pn_pairwise_f64,
           pn_pairwise_cosine_f64);

pub mod distance {
    use super::*;

    /// `trait Metric<A>` (src/distance.rs:9-14)
    pub trait Metric<A> {
        fn distance(&self, x1: &ArrayView1<A>, x2: &ArrayView1<A>) -> A;
        fn rdistance(&self, x1: &ArrayView1<A>, x2: &ArrayView1<A>) -> A;
        fn rdistance_to_distance(&self, d: A) -> A;
        fn distance_to_rdistance(&self, d: A) -> A;
        #[doc(hidden)]
        fn is_cosine(&self) -> bool;
    }
    /// `Euclidean` (src/distance.rs:16-55)
    #[derive(Default, Clone, Debug, Eq, PartialEq)]
    pub struct Euclidean {}
    unsafe impl Sync for Euclidean {}
    /// `Cosine` (src/distance.rs:76-122)
    #[derive(Default, Clone, Debug, Eq, PartialEq)]
    pub struct Cosine {}
    unsafe impl Sync for Cosine {}

    fn pair<'a, A: Elem>(x1: &'a ArrayView1<A>, x2: &'a ArrayView1<A>) -> (Vec<A>, Vec<A>) {
        (x1.iter().copied().collect(), x2.iter().copied().collect()) // any stride -> contiguous
    }
    impl<A: Elem + num_traits::Float> Metric<A> for Euclidean {
        fn distance(&self, x1: &ArrayView1<A>, x2: &ArrayView1<A>) -> A {
            let (a, b) = pair(x1, x2);
            unsafe { A::euclid(a.as_ptr(), b.as_ptr(), a.len().min(b.len()), false) } // zip truncates, src/distance.rs:27
        }
        fn rdistance(&self, x1: &ArrayView1<A>, x2: &ArrayView1<A>) -> A {
            let (a, b) = pair(x1, x2);
            unsafe { A::euclid(a.as_ptr(), b.as_ptr(), a.len().min(b.len()), true) }
        }
        fn rdistance_to_distance(&self, d: A) -> A { d.sqrt() }
        fn distance_to_rdistance(&self, d: A) -> A { d.powi(2) }
        fn is_cosine(&self) -> bool { false }
    }
    impl<A: Elem> Metric<A> for Cosine {
        fn distance(&self, x1: &ArrayView1<A>, x2: &ArrayView1<A>) -> A {
            let (a, b) = pair(x1, x2);
            unsafe { A::cosine(a.as_ptr(), a.len(), b.as_ptr(), b.len()) }
        }
        fn rdistance(&self, x1: &ArrayView1<A>, x2: &ArrayView1<A>) -> A { self.distance(x1, x2) }
        fn rdistance_to_distance(&self, d: A) -> A { d }
        fn distance_to_rdistance(&self, d: A) -> A { d }
        fn is_cosine(&self) -> bool { true }
    }

    /// `pairwise(x, &metric)` (src/distance.rs:58-74) on the GPU
    pub fn pairwise<A: Elem>(x: ArrayView2<A>, metric: &dyn Metric<A>) -> Array2<A> {
        let x = x.as_standard_layout();
        let (n, d) = x.dim();
        let mut out = Array2::<A>::default((n, n));
        ok(unsafe { A::pairwise(x.as_ptr(), n, d, d as isize, metric.is_cosine(), out.as_mut_ptr()) });
        out
    }
}
use distance::Metric;

/// `BallTree<'a, A, M>` (src/ball_tree.rs:15-24).  `idx` / `nodes` are not fields: no tree exists until an
/// introspection accessor is called (then the library builds the reference's tree once, on the host).
pub struct BallTree<'a, A: Elem, M: Metric<A>> {
    pub points: CowArray<'a, A, Ix2>,
    pub metric: M,
    handle: *mut ffi::pn_index,
}
// queries take &self and every call works in its own pooled workspace inside the library
unsafe impl<'a, A: Elem + Sync, M: Metric<A> + Sync> Sync for BallTree<'a, A, M> {}
unsafe impl<'a, A: Elem + Send, M: Metric<A> + Send> Send for BallTree<'a, A, M> {}

impl<'a, A: Elem, M: Metric<A>> BallTree<'a, A, M> {
    /// `BallTree::new` (src/ball_tree.rs:38-63): the validation and the two `ArrayError`s are the library's.
    pub fn new<T: Into<CowArray<'a, A, Ix2>>>(points: T, metric: M) -> Result<Self, ArrayError> {
        let points = points.into();
        let (n, d) = points.dim();
        let s = points.strides(); // element strides
        let mut handle = std::ptr::null_mut();
        let rc = unsafe { A::create(points.as_ptr(), n, d, s[0], s[1], metric.is_cosine(), &mut handle) };
        match rc {
            ffi::PN_OK => Ok(Self { points, metric, handle }),
            ffi::PN_ERR_EMPTY => Err(ArrayError::Empty),
            ffi::PN_ERR_NOT_CONTIGUOUS => Err(ArrayError::NotContiguous),
            _ => panic!("petal_mi355x: {}", last_error()),
        }
    }

    /// `BallTree::query` (src/ball_tree.rs:102-121): one point per call = a batch of one
    pub fn query<S: Data<Elem = A>>(&self, point: &ArrayBase<S, Ix1>, k: usize) -> (Vec<usize>, Vec<A>) {
        let kout = k.min(self.points.nrows());
        let q = point.as_standard_layout();
        let (mut idx, mut dist) = (vec![0u64; kout], vec![A::default(); kout]);
        ok(unsafe { A::query(self.handle, q.as_ptr(), 1, q.len(), k, idx.as_mut_ptr(), dist.as_mut_ptr()) });
        (idx.into_iter().map(|i| i as usize).collect(), dist)
    }
    /// `BallTree::query_nearest` (src/ball_tree.rs:80-86)
    pub fn query_nearest<S: Data<Elem = A>>(&self, point: &ArrayBase<S, Ix1>) -> (usize, A) {
        let (i, d) = self.query(point, 1);
        (i[0], d[0])
    }
    /// `BallTree::query_radius` (src/ball_tree.rs:137-142); ascending indices (the reference's order is unspecified)
    pub fn query_radius<S: Data<Elem = A>>(&self, point: &ArrayBase<S, Ix1>, distance: A) -> Vec<usize> {
        let q = point.as_standard_layout();
        let mut off = [0u64; 2];
        let mut out: *mut u64 = std::ptr::null_mut();
        ok(unsafe { A::radius(self.handle, q.as_ptr(), q.len(), distance, off.as_mut_ptr(), &mut out) });
        let v = unsafe { std::slice::from_raw_parts(out, off[1] as usize) }.iter().map(|&i| i as usize).collect();
        unsafe { ffi::pn_free(out as *mut _) };
        v
    }
    /// extension: the rows of `queries` in ONE call (what the GPU is for); (nq, min(k, n)) indices and distances
    pub fn query_batch(&self, queries: ArrayView2<A>, k: usize) -> (Array2<u64>, Array2<A>) {
        let q = queries.as_standard_layout();
        let (nq, d) = q.dim();
        let kout = k.min(self.points.nrows());
        let (mut idx, mut dist) = (Array2::<u64>::zeros((nq, kout)), Array2::<A>::default((nq, kout)));
        if nq > 0 && kout > 0 {
            ok(unsafe { A::query(self.handle, q.as_ptr(), nq, d, k, idx.as_mut_ptr(), dist.as_mut_ptr()) });
        }
        (idx, dist)
    }

    // ---- introspection (src/ball_tree.rs:296-353); out-of-range nodes panic like the reference
    pub fn num_nodes(&self) -> usize {
        let mut v = 0u64;
        ok(unsafe { ffi::pn_tree_num_nodes(self.handle, &mut v) });
        v as usize
    }
    pub fn num_points(&self) -> usize { self.points.nrows() }
    pub fn children_of(&self, n: usize) -> Option<(usize, usize)> {
        let (mut some, mut l, mut r) = (0 as c_int, 0u64, 0u64);
        ok(unsafe { ffi::pn_tree_children_of(self.handle, n as u64, &mut some, &mut l, &mut r) });
        (some != 0).then_some((l as usize, r as usize))
    }
    /// returns the node's slice of the permutation; `u64` where the reference has `usize` (same width on x86-64)
    pub fn points_of(&self, n: usize) -> &[u64] {
        let (mut p, mut c) = (std::ptr::null(), 0u64);
        ok(unsafe { ffi::pn_tree_points_of(self.handle, n as u64, &mut p, &mut c) });
        unsafe { std::slice::from_raw_parts(p, c as usize) }
    }
    pub fn radius_of(&self, n: usize) -> A {
        let mut v = A::default();
        ok(unsafe { A::radius_of(self.handle, n as u64, &mut v) });
        v
    }
    pub fn compare_nodes(&self, x: usize, y: usize) -> Option<Ordering> {
        let mut o = 0 as c_int;
        ok(unsafe { ffi::pn_tree_compare_nodes(self.handle, x as u64, y as u64, &mut o) });
        match o { -1 => Some(Ordering::Less), 0 => Some(Ordering::Equal), 1 => Some(Ordering::Greater), _ => None }
    }
    pub fn node_distance_lower_bound(&self, n1: usize, n2: usize) -> A {
        let mut v = A::default();
        ok(unsafe { A::lower_bound(self.handle, n1 as u64, n2 as u64, &mut v) });
        v
    }
}
impl<'a, A: Elem + num_traits::Float> BallTree<'a, A, distance::Euclidean> {
    /// `BallTree::euclidean` (src/ball_tree.rs:367-373)
    pub fn euclidean<T: Into<CowArray<'a, A, Ix2>>>(points: T) -> Result<Self, ArrayError> {
        Self::new(points, distance::Euclidean::default())
    }
}
impl<'a, A: Elem, M: Metric<A>> Drop for BallTree<'a, A, M> {
    fn drop(&mut self) { unsafe { ffi::pn_index_destroy(self.handle) } }
}

/// `VantagePointTree` (src/vantage_point_tree.rs:13-98): 1-NN only; same neighbour as `BallTree::query_nearest`
pub struct VantagePointTree<'a, A: Elem + num_traits::Float>(BallTree<'a, A, distance::Euclidean>);
impl<'a, A: Elem + num_traits::Float> VantagePointTree<'a, A> {
    pub fn euclidean<T: Into<CowArray<'a, A, Ix2>>>(points: T) -> Result<Self, ArrayError> {
        BallTree::euclidean(points).map(Self)
    }
    pub fn query_nearest<S: Data<Elem = A>>(&self, point: &ArrayBase<S, Ix1>) -> (usize, A) { self.0.query_nearest(point) }
}

/// Row shards of an f32 corpus over the GPUs `devices` (one process; the all-gather is RCCL inside the library)
pub struct ShardedBallTree {
    handle: *mut ffi::pn_sharded,
    n: usize,
}
impl ShardedBallTree {
    pub fn euclidean(points: ArrayView2<f32>, devices: &[i32]) -> Result<Self, ArrayError> {
        let (n, d) = points.dim();
        let s = points.strides();
        let mut handle = std::ptr::null_mut();
        let rc = unsafe { ffi::pn_sharded_create_f32(points.as_ptr(), n, d, s[0], s[1], devices.as_ptr(), devices.len() as c_int, &mut handle) };
        match rc {
            ffi::PN_OK => Ok(Self { handle, n }),
            ffi::PN_ERR_EMPTY => Err(ArrayError::Empty),
            ffi::PN_ERR_NOT_CONTIGUOUS => Err(ArrayError::NotContiguous),
            _ => panic!("petal_mi355x: {}", last_error()),
        }
    }
    pub fn query_batch(&self, queries: ArrayView2<f32>, k: usize) -> (Array2<u64>, Array2<f32>) {
        let q = queries.as_standard_layout();
        let (nq, d) = q.dim();
        let kout = k.min(self.n);
        let (mut idx, mut dist) = (Array2::<u64>::zeros((nq, kout)), Array2::<f32>::zeros((nq, kout)));
        if nq > 0 && kout > 0 {
            ok(unsafe { ffi::pn_sharded_query_f32(self.handle, q.as_ptr(), nq, d, d as isize, k, idx.as_mut_ptr(), dist.as_mut_ptr()) });
        }
        (idx, dist)
    }
}
impl Drop for ShardedBallTree {
    fn drop(&mut self) { unsafe { ffi::pn_sharded_destroy(self.handle) } }
}
