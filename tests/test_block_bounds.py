"""The bounds in front of a filter table too long for LDS, checked on the CPU against the tables the library builds.

A scene of more than 1440 spheres is scanned in two passes (raytracer_amd/csrc/rt_mfma.h, scan_mfma_culled): the
matrix-core filter first runs over one bounding sphere per table block of 32 spheres, in a frame in which the thin axes
of the cloud are stretched, together with a test for bounds that lie behind the ray's origin; then over the blocks whose
bound is wanted by some ray of the wave.  The reference asks every object (SceneManager::hitClosest,
src/RayTracer.cpp:77-82), so the cut is only right if a block that holds a sphere a ray can hit is always wanted by that
ray.  This test takes the tables from rt_filter_tables_host (host code only: the very function an upload calls), decodes
the half-floats the matrix cores will multiply, and checks in double:

  1. every table row is its sphere (the leaf order of the k-d partition loses none and invents none);
  2. every bound holds its 32 spheres, mapped into the bounds' frame, with the growth that covers the exact decision's
     rounding still there after the half-float split;
  3. for thousands of rays (aimed, grazing, starting on surfaces, origins far out): whenever a double evaluation of
     SphereObject::hitDetail (inc/RTObjects.h:135-149) finds a root ahead of the origin, the bound of that sphere's block
     has the ray's line inside it and some point ahead of the origin -- by the margin the filter's arithmetic needs;
  4. the arithmetic itself: the filter value of (ray, bound) as the split operands give it, less the worst rounding of
     the f32 accumulation, is still negative for those pairs; the one-product test for what lies behind, with its
     single half-floats, is off by less than half its margin, towards "ahead" or not."""
import numpy as np
import pytest

from raytracer_amd import scenes
from raytracer_amd.api import filter_tables
from tests.test_filter_bound import dot32, f32, fma32, half_rtz
from tests.test_gpu_filter import _cloud, _rays, _xyzr

MF_SCALE = 1024.0
MF_MARGIN = 2.0 ** -15
MF_MARGIN_FAR = 2.0 ** -20
MF_CULL = 1.002
MF_AHEAD_MARGIN = 2.0 ** -9
MF_AHEAD_CLAMP = 30.0


def _halves(words):
    """uint32 words -> (low half, high half) as doubles."""
    w = np.asarray(words, dtype=np.uint32)
    lo = (w & np.uint32(0xFFFF)).astype(np.uint16).view(np.float16).astype(np.float64)
    hi = (w >> np.uint32(16)).astype(np.uint16).view(np.float16).astype(np.float64)
    return lo, hi


def _matrix_row(i):  # mf_row_of_sphere(): matrix row of a block that holds the block's i-th sphere
    return (i & 3) + 8 * ((i & 15) >> 2) + 4 * (i >> 4)


def _filter_rows(block_words):
    """block_words [blocks, >= 2 k-steps, 64 lanes, 4 words] -> per row (block * 32 + i) the sixteen words of its two k-steps."""
    nb = block_words.shape[0]
    rows = np.zeros((nb * 32, 16), dtype=np.uint32)
    for i in range(32):
        r = _matrix_row(i)
        for step in range(2):
            for h in range(2):
                rows[i::32, 8 * step + 4 * h:8 * step + 4 * h + 4] = block_words[:, step, 32 * h + r, :]
    return rows


def _decode_filter_rows(rows):
    """-> (S_hi [n, 9], S_lo [n, 9], Q_hi, Q_lo, unit): the negated, scaled features as the matrix cores read them."""
    lo, hi = _halves(rows)
    S_hi = lo[:, :9]
    assert np.array_equal(lo[:, :9], hi[:, :9])  # words 0..8 hold the high half twice
    S_lo = np.stack([lo[:, 9], hi[:, 9], lo[:, 10], hi[:, 10], lo[:, 11], hi[:, 11], lo[:, 12], hi[:, 12], lo[:, 13]], axis=1)
    return S_hi, S_lo, hi[:, 13], lo[:, 14], lo[:, 15]


def _centre_radius(S_hi, S_lo, Q_hi, Q_lo):
    """The sphere a filter row stands for, in frame units: features 6..8 are -1024 * 2x, q is -1024 (r^2 - |x|^2)."""
    U = -(S_hi[:, 6:9] + S_lo[:, 6:9]) / (2.0 * MF_SCALE)
    r2 = -(Q_hi + Q_lo) / MF_SCALE + (U * U).sum(axis=1)
    return U, np.sqrt(np.maximum(r2, 0.0))


def _scenes():
    box = (30.0, 30.0, 4.0)
    return {
        "grid64": (lambda: scenes.by_name("grid64"), 30.0),
        "grid10k": (lambda: scenes.by_name("grid10k"), 45.0),
        "cloud": (lambda: _cloud("cloud", 3000, 7, box=box, radii=(0.05, 0.9), extra=[((1.0, 2.0, 0.5), -0.7), ((-3.0, 1.0, 0.0), -1.5)]), 25.0),
        "offset": (lambda: _cloud("offset", 2500, 8, centre=(5000.0, -3000.0, 2000.0), box=box, radii=(0.1, 0.6)), 25.0),
    }


@pytest.mark.parametrize("name", ["grid64", "grid10k", "cloud", "offset"])
def test_a_block_a_ray_can_hit_is_always_wanted_by_that_ray(name):
    make, spread = _scenes()[name]
    scene = make()
    xyzr = _xyzr(scene)
    T = filter_tables(scene.spheres)
    assert T is not None and T["bounds"] is not None, "the scene's table is expected to be too long for LDS"
    n_used, row_index = T["n_used"], T["row_index"]
    c0f = T["center"].astype(np.float64)
    inv_rs = float(T["scale"]) / MF_SCALE           # table frame units per world unit
    cs = T["bound_scale"].astype(np.float64)        # the bounds' frame, x 1024, per axis
    inv = cs / MF_SCALE
    inv_max = float(inv.max())
    Rs = 1.0 / inv_rs

    # 1. the table's rows are the scene's spheres, each once, but for the ones decided outside the table
    in_table = row_index[row_index >= 0]
    assert sorted(in_table.tolist()) == sorted(set(range(len(xyzr))) - set(T["big"]))
    assert np.all(row_index[:len(in_table)] >= 0) and np.all(row_index[len(in_table):] < 0)
    assert n_used == (len(in_table) + 31) // 32
    S_hi, S_lo, Q_hi, Q_lo, _ = _decode_filter_rows(_filter_rows(T["table"])[:len(in_table)])
    U_row, r_row = _centre_radius(S_hi, S_lo, Q_hi, Q_lo)
    want = (xyzr[in_table, :3] - c0f[None, :]) * inv_rs
    assert np.abs(U_row - want).max() < 2.0 ** -20
    assert np.abs(r_row - np.abs(xyzr[in_table, 3]) * inv_rs).max() < 2.0 ** -18 / np.abs(xyzr[in_table, 3] * inv_rs).min() * 2.0 ** -3 + 2.0 ** -18

    # 2. every bound holds its members, in the bounds' frame, with room
    bS_hi, bS_lo, bQ_hi, bQ_lo, _ = _decode_filter_rows(_filter_rows(T["bounds"])[:n_used])
    U, P = _centre_radius(bS_hi, bS_lo, bQ_hi, bQ_lo)
    u = (xyzr[in_table, :3] - c0f[None, :]) * inv[None, :]
    rho = np.abs(xyzr[in_table, 3]) * inv_max
    block_of_row = np.arange(len(in_table)) // 32
    reach = np.linalg.norm(u - U[block_of_row], axis=1) + rho
    need = np.zeros(n_used)
    np.maximum.at(need, block_of_row, reach)
    grow = Rs / 1024.0 * inv_max
    assert np.all(P >= need * (1.0 + 2.0 ** -10) + 0.5 * grow), "a bound lost its growth to rounding"
    # (a corner block's ball reaches a little past the unit ball the members fill: harmless -- a line that far out meets
    # no member -- as long as the numbers stay of the size the filter's error budget assumes, which 4a checks)
    assert np.all(np.linalg.norm(U, axis=1) + P <= 1.5), "the bounds leave the unit ball their frame is sized for"
    print(f"{name}: {len(in_table)} rows in {n_used} blocks, stretch {np.round(cs / cs.min(), 2).tolist()}, bounds' radii "
          f"{P.min() / inv.min():.2f} .. {P.max() / inv.min():.2f} world units along the axes that are not stretched")

    # the third k-step: the rows of the test for what lies behind a ray
    lanes = np.array([_matrix_row(i) for i in range(32)])
    lo0, hi0 = _halves(T["bounds"][:, 2, lanes, 0])           # lanes 0..31: (-1024 Ux, -1024 Uy), (-1024 Uz, -32768 P)
    lo1, hi1 = _halves(T["bounds"][:, 2, lanes, 1])
    U_half = -np.stack([lo0.reshape(-1), hi0.reshape(-1), lo1.reshape(-1)], axis=1)[:n_used] / MF_SCALE
    P_half = -hi1.reshape(-1)[:n_used] / 32768.0
    ones_lo, ones_hi = _halves(T["bounds"][:, 2, 32 + lanes, 0])  # lanes 32..63: the two slots that meet t
    assert np.all(ones_lo.reshape(-1)[:n_used] == -32768.0) and np.all(ones_hi.reshape(-1)[:n_used] == -32768.0)
    assert np.abs(U_half - U).max() <= 2.0 ** -11 and np.all(np.abs(P_half - P) <= 2.0 ** -11 * P)

    # 3. rays
    rng = np.random.default_rng(17)
    org, d = _rays(scene, 700, rng, spread)
    far = rng.random(len(org)) < 0.1  # some origins far out, aimed back at the cloud, and some past it, looking away
    org[far] = (c0f + (org[far] - c0f) * rng.uniform(5.0, 40.0, size=(far.sum(), 1))).astype(np.float32)
    d[far] = ((c0f + rng.normal(size=(far.sum(), 3)) * 0.3 * Rs - org[far]) * rng.choice([1.0, -1.0], size=(far.sum(), 1))).astype(np.float32)
    dh = (d * (1.0 / np.sqrt(dot32(d, d).astype(np.float64)))[:, None].astype(np.float32)).astype(np.float32)
    o64, d64 = org.astype(np.float64), dh.astype(np.float64)
    d64 /= np.linalg.norm(d64, axis=1, keepdims=True)
    tab = xyzr[in_table]
    oc = o64[:, None, :] - tab[None, :, :3]
    hb = np.einsum("rsk,rk->rs", oc, d64)
    c = np.einsum("rsk,rsk->rs", oc, oc) - tab[None, :, 3] ** 2
    disc = hb * hb - c
    # a root ahead of the origin, generously: whatever single precision could still take for one
    hit = (disc >= -1e-7 * tab[None, :, 3] ** 2) & (-hb + np.sqrt(np.maximum(disc, 0.0)) > -1e-6 * np.abs(tab[None, :, 3]))
    assert hit.sum() > 500
    ray_i, row_i = np.nonzero(hit)
    blk = block_of_row[row_i]

    # the ray in the bounds' frame, as the kernel maps it (floats)
    csf = T["bound_scale"]
    dsf = (dh * csf[None, :]).astype(np.float32)
    dc = (dsf * (1.0 / np.sqrt(dot32(dsf, dsf).astype(np.float64)))[:, None].astype(np.float32)).astype(np.float32)
    ocf = ((org - T["center"][None, :]).astype(np.float32) * csf[None, :]).astype(np.float32)  # x 1024
    o_fr = ocf.astype(np.float64) / MF_SCALE
    d_fr = dc.astype(np.float64)
    d_fr /= np.linalg.norm(d_fr, axis=1, keepdims=True)
    rel = U[None, :, :] - o_fr[:, None, :]                       # [rays, bounds, 3]
    along = np.einsum("rbk,rk->rb", rel, d_fr)
    disc_b = P[None, :] ** 2 - (np.einsum("rbk,rbk->rb", rel, rel) - along * along)
    ahead = along + P[None, :]
    m_ray = MF_MARGIN + MF_MARGIN_FAR * np.abs(o_fr).max(axis=1)
    assert np.all(disc_b[ray_i, blk] > 0.0), "a ray hits a sphere whose block's bound its line misses"
    assert np.all(ahead[ray_i, blk] > 0.0), "a ray hits a sphere whose block's bound lies behind its origin"

    # 4a. the filter on the bounds, as the split operands give it (tests/test_filter_bound.py has the same for the table)
    o2 = fma32(dc, -dot32(ocf, dc)[:, None], ocf)
    p = fma32(dc, -dot32(o2, dc)[:, None], o2)
    pp = (dot32(p, p) * f32(1.0 / MF_SCALE)).astype(np.float32)
    t = (fma32(f32(MF_MARGIN_FAR), np.abs(ocf).max(axis=1), f32(MF_MARGIN * MF_SCALE)) - pp).astype(np.float32)
    live = pp <= f32(MF_CULL * MF_CULL * MF_SCALE)
    ds32 = (dc * f32(32.0)).astype(np.float32)
    a = np.stack([ds32[:, 0] * ds32[:, 0], ds32[:, 1] * ds32[:, 1], ds32[:, 2] * ds32[:, 2], ds32[:, 0] * ds32[:, 1],
                  ds32[:, 0] * ds32[:, 2], ds32[:, 1] * ds32[:, 2], p[:, 0], p[:, 1], p[:, 2]], axis=1).astype(np.float32)
    a_hi = half_rtz(a)
    a_lo = half_rtz((a.astype(np.float64) - a_hi).astype(np.float32))
    t_hi = half_rtz(t)
    t_lo = half_rtz((t.astype(np.float64) - t_hi).astype(np.float32))
    assert np.all(live[ray_i]), "a ray that hits a sphere passes outside the bounds' frame"
    A_hi, A_lo, T_hi, T_lo = a_hi[ray_i], a_lo[ray_i], t_hi[ray_i], t_lo[ray_i]
    B_hi, B_lo, BQ = bS_hi[blk], bS_lo[blk], (bQ_hi + bQ_lo)[blk]
    terms = ((A_hi + A_lo) * B_hi).sum(axis=1) + (A_hi * B_lo).sum(axis=1) + MF_SCALE * BQ - MF_SCALE * (T_hi + T_lo)
    mags = (np.abs(A_hi * B_hi) + np.abs(A_lo * B_hi) + np.abs(A_hi * B_lo)).sum(axis=1) + MF_SCALE * np.abs(BQ) + MF_SCALE * (np.abs(T_hi) + np.abs(T_lo))
    value = -terms / MF_SCALE ** 2            # = disc_b + m, as the matrix cores see it (negated there)
    acc_bound = 32 * 2.0 ** -24 * mags / MF_SCALE ** 2
    err = np.abs(value - (disc_b[ray_i, blk] + m_ray[ray_i]))
    print(f"   {len(ray_i)} (ray, sphere) meetings in {len(np.unique(blk))} blocks; bound filter: split error up to "
          f"{float((err / m_ray[ray_i]).max()):.3f} of the margin, with the accumulation bound {float(((err + acc_bound) / m_ray[ray_i]).max()):.3f}")
    assert np.all(err + acc_bound < m_ray[ray_i])
    assert np.all(value - acc_bound > 0.0), "the filter could drop a bound whose block holds a hit"

    # 4b. the test for what lies behind: single half-floats, t split in two
    o_dot_d = dot32((ocf * f32(1.0 / MF_SCALE)).astype(np.float32), dc).astype(np.float64)
    t_a = np.clip(MF_AHEAD_MARGIN - o_dot_d, -MF_AHEAD_CLAMP, MF_AHEAD_CLAMP).astype(np.float32) * f32(32.0)
    ta_hi = half_rtz(t_a)
    ta_lo = half_rtz((t_a.astype(np.float64) - ta_hi).astype(np.float32))
    d_half = half_rtz((dc * f32(MF_SCALE)).astype(np.float32)) / MF_SCALE
    s_kernel = np.einsum("bk,rk->rb", U_half, d_half) + P_half[None, :] + ((ta_hi + ta_lo) / 32.0)[:, None]
    inside = np.abs(o_dot_d) < MF_AHEAD_CLAMP - 1.0
    s_true = ahead + MF_AHEAD_MARGIN
    off = np.abs(s_kernel - s_true)[inside]
    print(f"   behind-test: off by up to {float(off.max()):.2e} frame units (margin {MF_AHEAD_MARGIN:.2e}), "
          f"{100.0 * float((ahead[inside] <= 0).mean()):.1f} % of (ray, bound) pairs lie behind")
    assert off.max() < 0.5 * MF_AHEAD_MARGIN
    assert np.all(s_kernel[ray_i, blk] > 0.0), "the kernel's own arithmetic puts a bound that holds a hit behind the origin"
    # beyond the clamp the sign is still right: an origin that far past the cloud has everything behind it
    past = o_dot_d >= MF_AHEAD_CLAMP
    assert not np.any(hit[past]), "a ray that starts thirty frame radii past the cloud hits something"


def test_short_tables_get_no_bounds_and_tiny_scenes_no_table():
    assert filter_tables(scenes.by_name("small").spheres) is None
    T = filter_tables(scenes.by_name("random_spheres").spheres)
    assert T["bounds"] is None and T["n_bound_blocks"] == 0 and T["n_blocks"] == T["n_used"] == 16
