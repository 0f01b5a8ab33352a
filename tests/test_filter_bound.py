"""The matrix-core filter's margin as a checked inequality, on the CPU.

The render kernel's scan evaluates disc/Rs^2 + m per (ray, sphere) as a dot product of split half-float operands on
the matrix cores (raytracer_amd/csrc/rt_mfma.h) and treats a negative result as "cannot be ruled out".  It never
drops a sphere the ray's line meets as long as

    |value of the split operands' exact dot product - (disc/Rs^2 + m)|  +  (worst rounding of the f32 accumulation)  <  m

for every pair.  This test rebuilds both operand sides bit for bit in numpy (the table as abi_scene.inc builds it:
double -> float -> half, round to nearest even; the ray side as mfma_ray_operand does: float arithmetic, halves
rounded towards zero), evaluates the left side in double and checks the inequality over aimed, grazing, on-surface
and far-origin rays.  The reference has no counterpart: it tests every sphere exactly (src/RayTracer.cpp:77-82,
inc/RTObjects.h:135-149); the GPU tests in test_gpu_filter.py measure the same error on the device's own values."""
import numpy as np
import pytest

from raytracer_amd import scenes
from tests.test_gpu_filter import _rays, _xyzr

MF_SCALE = 1024.0
MF_MARGIN = 2.0 ** -15
MF_MARGIN_FAR = 2.0 ** -20
MF_CULL = 1.002
MF_MAX_BIG = 8
f32 = np.float32


def frame_of(xyzr):
    """Centre, radius and the spheres kept out of the table, as build_mf_table chooses them."""
    n = len(xyzr)
    big = np.zeros(n, bool)

    def bound():
        keep = ~big
        c, r = xyzr[keep, :3], np.abs(xyzr[keep, 3])
        lo, hi = (c - r[:, None]).min(axis=0), (c + r[:, None]).max(axis=0)
        c0 = 0.5 * (lo + hi)
        reach = (np.linalg.norm(c - c0, axis=1) + r).max()
        idx = np.flatnonzero(keep)
        return c0, reach, int(idx[np.argmax(r)])  # (first of the largest, as the strict '>' of the builder picks)

    c0, Rs, far_i = bound()
    n_big = 0
    while n_big < MF_MAX_BIG and n - n_big > 1:
        big[far_i] = True
        c1, R1, far1 = bound()
        if not R1 < 0.5 * Rs:
            big[far_i] = False
            break
        n_big += 1
        c0, Rs, far_i = c1, R1, far1
    return c0, Rs, big


def half_rne(x):
    return np.asarray(x, np.float64).astype(np.float32).astype(np.float16).astype(np.float64)


def half_rtz(x):
    """v_cvt_pkrtz_f16_f32: float -> half, rounded towards zero."""
    x = np.asarray(x, np.float32)
    h = x.astype(np.float16)
    over = np.abs(h.astype(np.float32)) > np.abs(x)
    h = np.where(over, np.nextafter(h, np.float16(0)), h)
    return h.astype(np.float64)


def fma32(a, b, c):
    return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(np.float32)


def dot32(a, b):  # fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)), as rt_device.h's dot()
    return fma32(a[..., 0], b[..., 0], fma32(a[..., 1], b[..., 1], (a[..., 2] * b[..., 2]).astype(np.float32)))


@pytest.mark.parametrize("name,spread", [("random_spheres", 6.0), ("grid30", 12.0), ("grid10k", 40.0)])
def test_split_half_operands_and_f32_accumulation_stay_inside_the_margin(name, spread):
    scene = scenes.by_name(name)
    xyzr = _xyzr(scene)
    c0, Rs, big = frame_of(xyzr)
    c0f = c0.astype(np.float32)
    scale = f32(MF_SCALE) / f32(Rs)
    inv_rs = float(scale) / MF_SCALE
    tab = xyzr[~big]

    # sphere side: nine features, q, and the constant that meets the per-ray term; negated, scaled by MF_SCALE
    x, y, z = ((tab[:, k] - float(c0f[k])) * inv_rs for k in range(3))
    r = tab[:, 3] * inv_rs
    feat = np.stack([x * x, y * y, z * z, 2 * x * y, 2 * x * z, 2 * y * z, 2 * x, 2 * y, 2 * z], axis=1)
    S = -MF_SCALE * feat
    S_hi = half_rne(S)
    S_lo = half_rne(S - S_hi)
    Q = -MF_SCALE * (r * r - (x * x + y * y + z * z))
    Q_hi = half_rne(Q)
    Q_lo = half_rne(Q - Q_hi)

    rng = np.random.default_rng(5)
    org, d = _rays(scene, 600, rng, spread)
    dh = (d * (1.0 / np.sqrt(dot32(d, d).astype(np.float64)))[:, None].astype(np.float32)).astype(np.float32)
    # ray side, in float as the kernel computes it
    oc = ((org - c0f[None, :]).astype(np.float32) * scale).astype(np.float32)
    o2 = fma32(dh, -dot32(oc, dh)[:, None], oc)
    p = fma32(dh, -dot32(o2, dh)[:, None], o2)
    pp = (dot32(p, p) * f32(1.0 / MF_SCALE)).astype(np.float32)
    far = np.abs(oc).max(axis=1)
    t = (fma32(f32(MF_MARGIN_FAR), far, f32(MF_MARGIN * MF_SCALE)) - pp).astype(np.float32)
    live = pp <= f32(MF_CULL * MF_CULL * MF_SCALE)
    ds = (dh * f32(32.0)).astype(np.float32)
    a = np.stack([ds[:, 0] * ds[:, 0], ds[:, 1] * ds[:, 1], ds[:, 2] * ds[:, 2], ds[:, 0] * ds[:, 1], ds[:, 0] * ds[:, 2],
                  ds[:, 1] * ds[:, 2], p[:, 0], p[:, 1], p[:, 2]], axis=1).astype(np.float32)
    a[~live] = 0.0  # a ray whose line passes outside the cloud is given operands that pass for no sphere
    t[~live] = -60000.0
    a_hi = half_rtz(a)
    a_lo = half_rtz((a.astype(np.float64) - a_hi).astype(np.float32))  # the remainder is exact in float
    t_hi = half_rtz(t)
    t_lo = half_rtz((t.astype(np.float64) - t_hi).astype(np.float32))

    # the 31 products of a (ray, sphere) pair, exactly (each is a product of two halves: exact in double)
    terms = np.concatenate([
        (a_hi + a_lo)[:, None, :] * S_hi[None, :, :],              # a_hi S_hi + a_lo S_hi
        a_hi[:, None, :] * S_lo[None, :, :],                       # a_hi S_lo
        (MF_SCALE * (Q_hi + Q_lo))[None, :, None] + 0 * a_hi[:, None, :1],
        (-MF_SCALE * (t_hi + t_lo))[:, None, None] + 0 * S_hi[None, :, :1],
    ], axis=2)
    value = -terms.sum(axis=2) / MF_SCALE ** 2  # = disc/Rs^2 + m, as the matrix cores see it
    # worst case of a float accumulation of the K = 32 products one after the other: half an ulp of the running
    # sum per addition, the running sum bounded by the sum of magnitudes
    split = np.concatenate([np.abs(a_hi[:, None, :] * S_hi[None]), np.abs(a_lo[:, None, :] * S_hi[None]),
                            np.abs(a_hi[:, None, :] * S_lo[None])], axis=2).sum(axis=2)
    mags = split + (MF_SCALE * (np.abs(Q_hi) + np.abs(Q_lo)))[None, :] + (MF_SCALE * (np.abs(t_hi) + np.abs(t_lo)))[:, None]
    acc_bound = 32 * 2.0 ** -24 * mags / MF_SCALE ** 2

    # what it stands for, in double from the same float inputs
    o64, dh64 = org.astype(np.float64), d.astype(np.float64)
    dh64 /= np.linalg.norm(dh64, axis=1, keepdims=True)
    oc64 = o64[:, None, :] - tab[None, :, :3]
    hb = np.einsum("rsk,rk->rs", oc64, dh64)
    disc = (tab[None, :, 3] ** 2 - (np.einsum("rsk,rsk->rs", oc64, oc64) - hb * hb)) / (Rs * Rs)
    oprime = (o64 - c0f.astype(np.float64)[None, :]) / Rs
    m = MF_MARGIN + MF_MARGIN_FAR * np.abs(oprime).max(axis=1)

    err = np.abs(value - (disc + m[:, None]))[live]
    room = np.broadcast_to(m[:, None], disc.shape)[live]
    worst = float(((err + acc_bound[live]) / room).max())
    print(f"{name}: {live.sum()} rays x {len(tab)} rows: representation error up to {float((err / room).max()):.3f} of the "
          f"margin, with the accumulation bound {worst:.3f}; sum of magnitudes up to {float(mags[live].max() / MF_SCALE ** 2):.2f}")
    assert live.sum() > 100
    assert worst < 1.0, "the margin does not cover the split and the accumulation"
    # and therefore: no pair whose line meets the sphere can come out non-negative
    meets = (disc >= 0)[live]
    assert np.all((value - acc_bound)[live][meets] > 0)
