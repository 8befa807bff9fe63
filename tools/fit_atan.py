"""Fit the odd minimax polynomial atan(q) = q + q^3 P(q^2) on [0,1] used by atan2_fast (flan_amd/csrc/pv_math.h) and
measure the fp32 evaluation error of the whole atan2 sequence against float64."""
import numpy as np
f32 = np.float32

def hexf(x): return float(f32(x)).hex()

def lawson(u, y, deg, w, iters=60):
    lw = np.ones_like(u)
    for _ in range(iters):
        V = np.vander(u, deg + 1, increasing=True)
        sw = np.sqrt(lw)
        coef = np.linalg.lstsq(V * sw[:, None], y * sw, rcond=None)[0]
        err = np.abs(V @ coef - y) * w
        lw = lw * (err / err.max() + 1e-3); lw /= lw.sum()
    return coef

N = 6000
q = np.cos(np.pi * (np.arange(N) + 0.5) / N) * 0.5 + 0.5
q = q[q > 1e-4]
u = q * q
y = (np.arctan(q) - q) / q ** 3
def fma(a, b, c): return f32(np.float64(a) * np.float64(b) + np.float64(c))

def atan2_fast(yv, xv, coef):
    yv = f32(yv); xv = f32(xv)
    ax, ay = np.abs(xv), np.abs(yv)
    mx, mn = np.maximum(ax, ay), np.minimum(ax, ay)
    with np.errstate(all="ignore"):
        r = f32(1.0) / mx                      # v_rcp_f32 (1 ulp) -- modelled as exact-rounded
        q0 = f32(mn * r); e = fma(-q0, mx, mn); qq = fma(e, r, q0)
    uu = f32(qq * qq)
    p = f32(coef[-1])
    for c in coef[-2::-1]: p = fma(p, uu, f32(c))
    a = fma(f32(qq * uu), p, qq)
    a = np.where(ay > ax, f32(f32(np.pi / 2) - a), a)
    a = np.where(np.signbit(xv), f32(f32(np.pi) - a), a)
    a = np.where(mx == 0, np.where(np.signbit(xv), f32(np.pi), f32(0)), a)
    return np.copysign(a, yv).astype(f32)

rng = np.random.default_rng(1)
n = 4_000_000
xv = (rng.standard_normal(n) * 10 ** rng.uniform(-3, 3, n)).astype(f32)
yv = (rng.standard_normal(n) * 10 ** rng.uniform(-3, 3, n)).astype(f32)
ref = np.arctan2(yv.astype(np.float64), xv.astype(np.float64))
ref32 = ref.astype(f32)
for deg in (6, 7, 8, 9):
    coef = lawson(u, y, deg, w=q ** 3 / np.arctan(q))
    got = atan2_fast(yv, xv, coef)
    err = np.abs(got.astype(np.float64) - ref)
    ulp = err / np.spacing(np.abs(ref32)).astype(np.float64)
    print("deg", deg, "max abs err %.3e  max ulp %.2f  mean ulp %.3f  frac == RN(atan2) %.4f" % (err.max(), ulp.max(), ulp.mean(), np.mean(got == ref32)))
    print("   coef:", ", ".join(hexf(c) for c in coef))
