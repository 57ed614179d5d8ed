"""include/photon_det_math.h (host build, via the oracle library) against numpy's libm.

These functions replace atanf/tanf/cos/sin/acosf on the ray path so that oracle and HIP kernels
produce identical bits; here we pin their ACCURACY independently of both."""
import numpy as np
import pytest

SIN, COS, TAN, ATAN, ACOS, ATANF, TANF, ACOSF, COSF = 0, 1, 2, 3, 4, 10, 11, 12, 13


def ulps(y, ref):
    return np.abs(y - ref) / np.spacing(np.abs(ref))


def test_double_functions_within_4_ulp(oracle):
    rng = np.random.default_rng(0)
    t = rng.uniform(0.0, 2 * np.pi, 200_000)
    for fn, ref in ((SIN, np.sin), (COS, np.cos)):
        y = oracle.det_eval(fn, t)
        r = ref(t.astype(np.longdouble)).astype(np.float64)
        assert np.abs(y - r).max() <= 2.3e-16            # absolute: what the callers need
    x = rng.uniform(-1.5, 1.5, 200_000)
    assert ulps(oracle.det_eval(TAN, x), np.tan(x.astype(np.longdouble)).astype(np.float64)).max() <= 4
    x = np.concatenate([rng.uniform(-3, 3, 200_000), 10 * rng.standard_cauchy(100_000)])
    assert ulps(oracle.det_eval(ATAN, x), np.arctan(x.astype(np.longdouble)).astype(np.float64)).max() <= 4
    x = rng.uniform(-1, 1, 200_000)
    assert ulps(oracle.det_eval(ACOS, x), np.arccos(x.astype(np.longdouble)).astype(np.float64)).max() <= 5


@pytest.mark.parametrize("fn,ref,lo,hi", [(ATANF, np.arctan, -0.3, 0.3), (TANF, np.tan, -0.3, 0.3),
                                          (ATANF, np.arctan, -50.0, 50.0), (TANF, np.tan, -1.5, 1.5),
                                          (ACOSF, np.arccos, -1.0, 1.0), (COSF, np.cos, 0.0, 1.5)])
def test_float_functions_are_correctly_rounded(oracle, fn, ref, lo, hi):
    """f32 results = correctly rounded value (<= 1 ulp allowed, and at most 1e-5 of cases off)."""
    rng = np.random.default_rng(fn)
    x = rng.uniform(lo, hi, 300_000).astype(np.float32)
    y = oracle.det_eval(fn, x.astype(np.float64)).astype(np.float32)
    cr = ref(x.astype(np.longdouble)).astype(np.float32)
    off = y != cr
    assert off.mean() <= 1e-5
    assert (np.abs(y.astype(np.float64) - cr) <= np.spacing(np.abs(cr))).all()


def test_special_values(oracle):
    a = oracle.det_eval(ACOS, [1.0, 0.0])
    assert a[0] == 0.0 and a[1] == pytest.approx(np.pi / 2, abs=4e-16)
    assert oracle.det_eval(ACOS, [-1.0])[0] == pytest.approx(np.pi, abs=1e-15)
    assert np.isnan(oracle.det_eval(ACOS, [1.5, -1.0000001, np.nan])).all()
    a = oracle.det_eval(ATAN, [np.inf, -np.inf, 0.0, np.nan])
    assert a[0] == pytest.approx(np.pi / 2) and a[1] == pytest.approx(-np.pi / 2) and a[2] == 0.0 and np.isnan(a[3])
    assert oracle.det_eval(SIN, [0.0])[0] == 0.0 and oracle.det_eval(COS, [0.0])[0] == 1.0


def test_philox_normals_are_standard_normal_and_keyed(oracle):
    """include/photon_philox.h: N(0,1) pairs addressed by (seed, ray, draw, stream)."""
    a = oracle.normal2(seed=123, n=400_000)
    assert abs(a.mean()) < 5e-3 and abs(a.std() - 1.0) < 5e-3
    assert abs(np.mean(a[:, 0] * a[:, 1])) < 5e-3                      # the two variates are uncorrelated
    assert abs(np.mean(a ** 4) - 3.0) < 0.05                           # Gaussian kurtosis
    assert np.array_equal(a, oracle.normal2(seed=123, n=400_000))      # reproducible
    assert not np.array_equal(a[:1000], oracle.normal2(seed=124, n=1000))
    assert not np.array_equal(a[:1000], oracle.normal2(seed=123, n=1000, draw=1))
    assert not np.array_equal(a[:1000], oracle.normal2(seed=123, n=1000, stream=2))
    # Philox4x32-10 known answer (Random123 kat_vectors: counter 0, key 0)
    # first word of philox4x32-10(ctr={0,0,0,0}, key={0,0}) = 0x6627e8d5
    u1 = (0x6627e8d5 + 0.5) / 2 ** 32
    u2 = (0xe169c58d + 0.5) / 2 ** 32
    z = oracle.normal2(seed=0, n=1, draw=0, stream=0)[0]
    r = np.sqrt(-2 * np.log(u1))
    assert np.allclose(z, [r * np.cos(2 * np.pi * u2), r * np.sin(2 * np.pi * u2)], rtol=1e-6)


def test_det_erf_absolute_error_bound(oracle):
    """photon_det_erf (the Gaussian-spot splat's erf, shared by oracle and product since round 5): |error| <= 4e-16 for
    every argument, against 50-digit values -- on a dense random set over [-6.5, 6.5], the interval seams k / 16 and their
    neighbours, the splat's own arguments sqrt(8) (j +- 1/2 - frac) / D, and the special values."""
    import mpmath as mp
    mp.mp.dps = 50
    rng = np.random.default_rng(5)
    seams = np.arange(0, 97) / 16.0
    x = np.concatenate([rng.uniform(-6.5, 6.5, 60_000), seams, np.nextafter(seams, 0), np.nextafter(seams, 10), -seams,
                        (np.sqrt(np.float32(8.0)).astype(np.float64) * (np.arange(-4, 5)[None, :] + 0.5 - rng.random((2000, 1))) / 3.0).ravel(),
                        rng.uniform(-1e-3, 1e-3, 2000)])
    y = oracle.det_erf(x)
    want = np.array([float(mp.erf(mp.mpf(float(v)))) for v in x])          # correctly rounded reference values
    exact_err = np.array([abs(float(mp.mpf(float(a)) - mp.erf(mp.mpf(float(v))))) for a, v in zip(y[:5000], x[:5000])])
    assert np.abs(y - want).max() <= 4e-16 and exact_err.max() <= 4e-16, (np.abs(y - want).max(), exact_err.max())
    assert np.array_equal(oracle.det_erf(-x), -y)                           # odd, exactly (at 0 itself: +-6.9e-18, within the bound)
    s = oracle.det_erf(np.array([6.0, 7.5, 1e300, np.inf, -6.0, -np.inf, 0.0, np.nan]))
    assert np.array_equal(s[:6], [1, 1, 1, 1, -1, -1]) and abs(s[6]) <= 1e-16 and abs(s[7]) == 1.0      # NaN -> +-1 (documented)
    # against glibc's erf: the two agree to a few 1e-16 (what test_erf_form_sensitivity sees on images)
    from scipy.special import erf as sp_erf
    assert np.abs(y - sp_erf(x)).max() <= 6e-16


def test_div_rcp_equals_division(oracle):
    """photon_det_div_rcp(a, b, 1 / b) -- the product's erf splat divides its sixteen erf arguments per ray by the spot
    diameter this way -- is the correctly rounded a / b the oracle (and the reference) compute with a division: no
    mismatch on 8e6 quotients, divisors = the float spot diameters a camera can have (3 px in every shipped case)."""
    rng = np.random.default_rng(9)
    divisors = [3.0, 1.0, 2.5, 0.75, 7.3, 1.9999998807907104, 1e-3, 37.0] + [float(np.float32(v)) for v in rng.uniform(0.5, 20.0, 8)]
    for b in divisors:
        a = np.concatenate([rng.uniform(-15.0, 15.0, 400_000) * b / 3.0, rng.standard_normal(100_000) * 1e-3])
        # the splat's own numerators: (double)sqrtf(8) * ((double)(float)(idx - X) -+ 0.5)
        X = rng.uniform(8.0, 1016.0, 50_000).astype(np.float32)
        idx = (np.floor(X) + rng.integers(-4, 5, X.size)).astype(np.float32)
        a = np.concatenate([a, float(np.sqrt(np.float32(8.0))) * ((idx - X).astype(np.float64) - 0.5)])
        assert oracle.det_div_rcp_mismatches(a, b) == 0, b
