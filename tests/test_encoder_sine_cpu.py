"""The IPE encoder's own sine / exponential (hos_encode.hip: enc_sin, enc_exp), restated in numpy float32 with the same constants
and the same operation order, against float64: the bound the kernel's comment states (<= 2 ulp over the model's argument range,
2^l * (contracted mean . unit direction) (+ pi/2), l < 12).  CPU only; the GPU tests compare the kernel's features with the oracle."""
import numpy as np

f32 = np.float32


def fma(a, b, c):
    # products of two float32 are exact in float64; one rounding to float32 at the end (as v_fma_f32)
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)


def enc_sin(x):
    x = x.astype(f32)
    kf = np.rint(x * f32(float.fromhex("0x1.45f306p-1"))).astype(f32)
    r = fma(kf, np.full_like(x, -float.fromhex("0x1.921fb6p+0")), x)
    r = fma(kf, np.full_like(x, float.fromhex("0x1.777a5cp-25")), r)
    r = fma(kf, np.full_like(x, float.fromhex("0x1.ee59dap-50")), r)
    k = kf.astype(np.int64)
    z = r * r
    c = lambda v: np.full_like(x, v)
    sp = fma(c(-1.9515295891e-4), z, c(8.3321608736e-3))
    sp = fma(sp, z, c(-1.6666654611e-1))
    sp = fma(sp * z, r, r)
    cp = fma(c(2.443315711809948e-5), z, c(-1.388731625493765e-3))
    cp = fma(cp, z, c(4.166664568298827e-2))
    cp = fma(cp * z, z, fma(c(-0.5), z, c(1.0)))
    v = np.where(k & 1, cp, sp)
    return np.where(k & 2, -v, v).astype(f32)


def enc_exp(x):
    x = x.astype(f32)
    l2e, l2e_lo = f32(float.fromhex("0x1.715476p+0")), f32(float.fromhex("0x1.4ae0cp-26"))
    t = x * l2e
    e = fma(x, np.full_like(x, l2e), -t) + x * l2e_lo
    return (np.exp2(t.astype(np.float64)).astype(f32) * fma(e, np.full_like(x, 0.69314718), np.full_like(x, 1.0))).astype(f32)


def ulps(got, want):
    w32 = want.astype(f32)
    ulp = np.spacing(np.abs(w32)).astype(np.float64)
    return np.abs(got.astype(np.float64) - want) / ulp


def test_enc_sin_within_two_ulp_over_the_model_range():
    rng = np.random.RandomState(7)
    m = (rng.rand(2_000_000).astype(f32) * 2 - 1) * f32(2.2)
    lvl = rng.randint(0, 12, m.shape)
    x = (m * (2.0 ** lvl).astype(f32)).astype(f32)
    x = np.where(rng.rand(*x.shape) < 0.5, x, (x + f32(1.57079637050628662109375)).astype(f32))
    want = np.sin(x.astype(np.float64))
    got = enc_sin(x)
    assert np.abs(got.astype(np.float64) - want).max() < 1.2e-7
    big = np.abs(want) > 1e-3
    assert ulps(got[big], want[big]).max() < 2.0


def test_enc_exp_within_two_and_a_half_ulp():
    rng = np.random.RandomState(8)
    x = -(rng.rand(1_000_000).astype(f32) * f32(87.0))
    x[::3] *= f32(0.01)
    want = np.exp(x.astype(np.float64))
    assert ulps(enc_exp(x), want).max() < 2.5
