"""Accuracy of the kernels' table-driven fp64 primitives (marl_math.h).

numpy's log/exp/power are accurate to well under 1 ulp on this platform, so they serve as the
reference; errors are reported in ulps of the result."""
import ctypes as C

import numpy as np
import pytest

from common import scenario

pytestmark = pytest.mark.gpu


def probe(eq, op, x, e=0.0):
    import torch
    xd = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64)).cuda()
    yd = torch.empty_like(xd)
    rc = eq._lib.marl_debug_math(eq._ctx, op, C.c_void_p(xd.data_ptr()), C.c_void_p(yd.data_ptr()), xd.numel(), float(e))
    assert rc == 0
    torch.cuda.synchronize()
    return yd.cpu().numpy()


def ulps(got, ref):
    return np.abs(got - ref) / np.spacing(np.abs(ref))


@pytest.fixture(scope="module")
def eq():
    from marlpde_amd.LHeureux_model import LMAHeureuxPorosityDiff
    m = LMAHeureuxPorosityDiff.from_scenario(scenario("default", 64), device=0)
    yield m
    m.close()


def test_log(eq):
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(1e-3, 2.0, 200000), 10 ** rng.uniform(-300, 300, 100000),
                        np.linspace(0.6874, 0.6876, 1001), np.linspace(1.3749, 1.3751, 1001), 1 + np.linspace(-1e-3, 1e-3, 2001)])
    got = probe(eq, 0, x)
    ref = np.log(x)
    far = np.abs(x - 1) > 0.05
    print("log: max ulp error away from 1:", ulps(got[far], ref[far]).max(), " max abs error near 1:", np.max(np.abs(got[~far] - ref[~far])))
    assert ulps(got[far], ref[far]).max() <= 2.0
    assert np.max(np.abs(got[~far] - ref[~far])) <= 2.5e-17       # absolute near log(1) = 0
    sp = np.array([0.0, -0.0, -1.0, np.inf, np.nan, 5e-324, -np.inf])
    with np.errstate(all="ignore"):
        ref = np.log(sp)
    got = probe(eq, 0, sp)
    assert np.array_equal(np.isnan(got), np.isnan(ref)) and np.array_equal(got[~np.isnan(ref)], ref[~np.isnan(ref)])


def test_exp(eq):
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(-40, 40, 200000), rng.uniform(-745, 709, 100000), rng.uniform(-1e-3, 1e-3, 10000)])
    got = probe(eq, 1, x)
    ref = np.exp(x)
    normal = ref > 1e-300
    print("exp: max ulp error:", ulps(got[normal], ref[normal]).max())
    assert ulps(got[normal], ref[normal]).max() <= 1.5
    assert np.all(np.abs(got[~normal] - ref[~normal]) <= 1e-300)
    # contract of the kernels' exp: finite arguments from about -1e7 up to 709 (0 below -745), NaN -> NaN;
    # -inf and overflow are the callers' business (pow_sat patches b = 0; see marl_math.h)
    sp = np.array([-1e6, -1e5, -800.0, np.nan, 0.0])
    got = probe(eq, 1, sp)
    assert list(got[:3]) == [0, 0, 0] and np.isnan(got[3]) and got[4] == 1.0


@pytest.mark.parametrize("e", [2.48, 2.8, 1.0, 0.5])
def test_pow(eq, e):
    rng = np.random.default_rng(2)
    b = np.concatenate([rng.uniform(0, 1.2, 200000), 10 ** rng.uniform(-12, 0, 50000), [0.0, 1.0]])
    got = probe(eq, 2, b, e)
    ref = np.power(b, e)
    # error model: (2 + 3 |e ln b|) ulp of the result: the log's <= 1 ulp error and the rounding of e*log(b)
    # are amplified by |e ln b|; large only where b^e itself is negligible
    bound = (2.0 + 3.0 * np.abs(e * np.log(np.maximum(b, 1e-300)))) * np.spacing(np.abs(ref))
    assert np.all(np.abs(got - ref) <= bound + 1e-300)
    assert got[-2] == 0.0 and got[-1] == 1.0
    assert np.isnan(probe(eq, 2, np.array([np.nan]), e))[0]


def test_reciprocal_and_sigma(eq):
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.uniform(-10, 10, 100000), 10 ** rng.uniform(-100, 100, 10000)])
    assert ulps(probe(eq, 3, x), 1.0 / x).max() <= 1.0
    assert not np.isfinite(probe(eq, 3, np.array([0.0, np.nan, -0.0]))).any()   # non-finite in, non-finite out
    pe = np.concatenate([rng.uniform(-100, 100, 50000), rng.uniform(-0.05, 0.05, 50000), [0.0, 1e-2, -1e-2, 100.0, 150.0, -150.0]])
    got = probe(eq, 4, pe, -4.28)
    a = np.abs(pe)
    with np.errstate(all="ignore"):
        ref = np.where(a < 1e-2, 0.0, np.where(a > 100, np.sign(-4.28), np.cosh(pe) / np.sinh(pe) - 1 / pe))
    print("sigma: max abs error:", np.max(np.abs(got - ref)))
    assert np.max(np.abs(got - ref)) <= 2e-12     # cancellation near |Pe| = 1e-2 on both sides (coth ~ 100)
    # the fused integrators' variant (op 5): power series for |Pe| <= 0.5, the closed form beyond; checked against the closed form
    # in extended precision (whose own cancellation error is eps_80bit / Pe^2 <= 1e-15 relative)
    got5 = probe(eq, 5, pe, -4.28)
    assert np.array_equal(got5[a > 0.5], got[a > 0.5]) and np.all(got5[a < 1e-2] == 0.0)
    m = (a >= 1e-2) & (a <= 0.5)
    xl = pe[m].astype(np.longdouble)
    refl = (np.cosh(xl) / np.sinh(xl) - 1 / xl).astype(np.float64)
    assert np.max(np.abs(got5[m] - refl) / np.abs(refl)) <= (1e-14 if np.finfo(np.longdouble).eps < 1e-18 else 1e-11)
