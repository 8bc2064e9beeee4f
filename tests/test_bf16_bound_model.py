"""CPU model of the bf16 tier's bound (petal-neighbors_amd/csrc/bf16_filter.hip, header): the same constants,
computed with the same rounding directions in numpy, and the EXACT (f64) value of the dot product the matrix core
evaluates.  Checks, without a GPU, the inequality the tier's proof rests on,

    exact_sum + g * sum|terms|  <=  |q - p|^2 - |q|^2_down        (g = 2^-13),

i.e. that a matrix core whose accumulation error stays within g * sum|terms| can never return a value above the
true squared distance.  The GPU tests (tests/test_gpu_bf16.py) check the delivered values themselves.
"""
import numpy as np
import pytest

from conftest import uniform

G = 2.0 ** -13
UP = 1.0 + 2.0 ** -40


def bf16_rne(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    y = r.astype(np.uint32).view(np.float32).reshape(x.shape)
    return np.where(np.abs(x) < 2.0 ** -60, np.float32(0), y)


def f_down(x):  # largest float32 <= x (x >= 0)
    f = x.astype(np.float32)
    return np.where(f.astype(np.float64) > x, np.nextafter(f, np.float32(-np.inf)), f)


def f_up(x):
    f = x.astype(np.float32)
    return np.where(f.astype(np.float64) < x, np.nextafter(f, np.float32(np.inf)), f)


def bf16_trunc(f):  # f >= 0 float32 -> bf16 value (as float32), toward zero
    u = np.ascontiguousarray(f, dtype=np.float32).view(np.uint32) & np.uint32(0xFFFF0000)
    return u.view(np.float32)


def bf16_up(f):  # f >= 0 float32 -> smallest bf16 >= f
    u = np.ascontiguousarray(f, dtype=np.float32).view(np.uint32)
    t = (u >> 16) + ((u & 0xFFFF) != 0)
    return (t.astype(np.uint32) << 16).view(np.float32)


def corpus_columns(p, mu=None):
    """mu: translation vector (f32, per dimension); the tier works with p - mu (f64 subtraction)."""
    p64 = p.astype(np.float64) - (0.0 if mu is None else mu.astype(np.float64)[None, :])
    ph = bf16_rne(p64.astype(np.float32))
    ph64 = ph.astype(np.float64)
    pn = (p64 * p64).sum(1)
    en = ((p64 - ph64) ** 2).sum(1)
    hn = (ph64 * ph64).sum(1)
    rem = pn * (1.0 - G) / UP
    h0 = bf16_trunc(f_down(rem)); rem = rem - h0
    h1 = bf16_trunc(f_down(rem)); rem = rem - h1
    h2 = bf16_trunc(f_down(rem))
    bp = bf16_up(f_up((2.0 * np.sqrt(en) * UP + 2.0 * G * np.sqrt(hn) * UP) * (1.0 + 2.0 * G)))
    dp = bf16_up(f_up(2.0 * np.sqrt(pn) * UP * (1.0 + 2.0 * G)))
    return ph64, np.stack([h0, h1, h2], 1).astype(np.float64), bp.astype(np.float64), dp.astype(np.float64)


def corpus_norm_f32(p, mu=None):
    """layout 2: |p - mu|^2 (1 - g), rounded down to f32 -- the accumulator's initial value"""
    p64 = p.astype(np.float64) - (0.0 if mu is None else mu.astype(np.float64)[None, :])
    return f_down((p64 * p64).sum(1) * (1.0 - G) / UP).astype(np.float64)


def row_constants_f64(p, mu=None):
    """Bp, Dp before their bf16 rounding (bf16_row_stats_kernel): layout 2 uses their maxima over the corpus"""
    p64 = p.astype(np.float64) - (0.0 if mu is None else mu.astype(np.float64)[None, :])
    ph64 = bf16_rne(p64.astype(np.float32)).astype(np.float64)
    pn, en, hn = (p64 * p64).sum(1), ((p64 - ph64) ** 2).sum(1), (ph64 * ph64).sum(1)
    bp = (2.0 * np.sqrt(en) * UP + 2.0 * G * np.sqrt(hn) * UP) * (1.0 + 2.0 * G) * UP
    dp = 2.0 * np.sqrt(pn) * UP * (1.0 + 2.0 * G) * UP
    return bp, dp


def query_columns(q, mu=None):
    q64 = q.astype(np.float64) - (0.0 if mu is None else mu.astype(np.float64)[None, :])
    qh = bf16_rne(q64.astype(np.float32))
    qh64 = qh.astype(np.float64)
    s = (q64 * q64).sum(1)
    aq = bf16_up(f_up(np.sqrt((qh64 * qh64).sum(1)) * UP)).astype(np.float64)
    cq = bf16_up(f_up(np.sqrt(((q64 - qh64) ** 2).sum(1)) * UP)).astype(np.float64)
    return -2.0 * qh64, aq, cq, s / UP


CASES = {
    "uniform": lambda n, d, s: uniform((n, d), s),
    "centered": lambda n, d, s: uniform((n, d), s) - np.float32(0.5),
    "offset1000": lambda n, d, s: uniform((n, d), s) + np.float32(1000.0),
    "scaled1e6": lambda n, d, s: (uniform((n, d), s) - np.float32(0.5)) * np.float32(1e6),
    "tiny1e-20": lambda n, d, s: (uniform((n, d), s) - np.float32(0.5)) * np.float32(1e-20),
    "mixed_scales": lambda n, d, s: ((uniform((n, d), s) - np.float32(0.5))
                                      * (np.float32(10.0) ** (np.arange(d, dtype=np.float32) % 9 - 4))).astype(np.float32),
    "one_hot": lambda n, d, s: np.eye(d, dtype=np.float32)[np.arange(n) % d] * (1 + uniform((n, 1), s)),
}


@pytest.mark.parametrize("name", sorted(CASES))
@pytest.mark.parametrize("dim", [3, 16, 100, 128, 768])
def test_bound_holds_with_the_accumulation_allowance(name, dim):
    p = CASES[name](400, dim, 3)
    q = CASES[name](60, dim, 4)
    mu = p.astype(np.float64).mean(0).astype(np.float32)  # any translation is valid; the library uses the mean
    ph, pieces, bp, dp = corpus_columns(p, mu)
    mq, aq, cq, qn = query_columns(q, mu)
    dot = mq @ ph.T                                   # sum of (-2 q^_k) p^_k, exact enough in f64 (bf16 x bf16 terms)
    exact = pieces.sum(1)[None, :] + dot - aq[:, None] * bp[None, :] - cq[:, None] * dp[None, :]
    mags = pieces.sum(1)[None, :] + np.abs(mq) @ np.abs(ph).T + aq[:, None] * bp[None, :] + cq[:, None] * dp[None, :]
    p64, q64 = p.astype(np.float64), q.astype(np.float64)
    d2 = ((q64[:, None, :] - p64[None, :, :]) ** 2).sum(2)
    slack = (d2 - qn[:, None]) - (exact + G * mags)
    # f64 evaluation noise of this model itself: ~1e-13 relative to the magnitudes
    assert (slack >= -1e-12 * mags).all(), f"{name}/D={dim}: margin violated by {slack.min()}"
    # and the constant handed to the proof is a lower bound of |q - mu|^2
    assert (qn <= ((q64 - mu.astype(np.float64)) ** 2).sum(1)).all()


def test_bound_is_tight_enough_to_be_useful_on_uniform_data():
    p = uniform((2000, 128), 5)
    q = uniform((50, 128), 6)
    mu = p.astype(np.float64).mean(0).astype(np.float32)
    ph, pieces, bp, dp = corpus_columns(p, mu)
    mq, aq, cq, qn = query_columns(q, mu)
    exact = pieces.sum(1)[None, :] + mq @ ph.T - aq[:, None] * bp[None, :] - cq[:, None] * dp[None, :]
    d2 = ((q.astype(np.float64)[:, None, :] - p.astype(np.float64)[None, :, :]) ** 2).sum(2)
    gap = d2 - (exact + qn[:, None])
    assert gap.min() > 0 and gap.max() < 0.2          # squared distances are ~21 +- 2.2 here


@pytest.mark.parametrize("name", sorted(CASES))
@pytest.mark.parametrize("dim", [32, 96, 128])
def test_norm_in_accumulator_layout_bound_holds(name, dim):
    """Layout 2 (bf16_filter.hip, bf16_ci_dim): value = norm_f32 + sum (-2 q^_k) p^_k, and the proof adds back
    |q|^2_down - E(q) with E(q) = |q^|_up Bmax + |eq|_up Dmax.  With the accumulation allowance g (|norm| + sum |products|):
        exact + g * mags - E(q)  <=  |q - p|^2 - |q|^2_down."""
    p = CASES[name](400, dim, 13)
    q = CASES[name](60, dim, 14)
    mu = p.astype(np.float64).mean(0).astype(np.float32)
    ph, _, _, _ = corpus_columns(p, mu)
    cn = corpus_norm_f32(p, mu)
    bp, dp = row_constants_f64(p, mu)
    mq, _, _, qn = query_columns(q, mu)
    q64 = q.astype(np.float64) - mu.astype(np.float64)[None, :]
    qh64 = bf16_rne(q64.astype(np.float32)).astype(np.float64)
    eq = (np.sqrt((qh64 * qh64).sum(1)) * UP * bp.max() + np.sqrt(((q64 - qh64) ** 2).sum(1)) * UP * dp.max()) * UP
    exact = cn[None, :] + mq @ ph.T
    mags = cn[None, :] + np.abs(mq) @ np.abs(ph).T
    p64 = p.astype(np.float64)
    d2 = ((q.astype(np.float64)[:, None, :] - p64[None, :, :]) ** 2).sum(2)
    slack = (d2 - (qn - eq)[:, None]) - (exact + G * mags)
    assert (slack >= -1e-12 * (mags + eq[:, None])).all(), f"{name}/D={dim}: margin violated by {slack.min()}"
