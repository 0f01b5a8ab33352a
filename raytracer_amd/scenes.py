"""Build-defined scenes for the benchmark configurations.

The reference ships no scenes; these are "Ray Tracing in One Weekend" look-alikes
adapted to the reference camera's conventions (world up = +z unless the view
direction is within 0.999 of it, fixed 1x1 viewport at focal length 1 —
reference src/Camera.cpp:206-221, inc/Camera.h:24-26,48).

Every number is rounded to float32 here, so the HIP library and the CPU oracle
receive bit-identical inputs.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .api import (CAMERA_DTYPE, LIGHT_DTYPE, MATERIAL_DTYPE, PRIM_DTYPE, SPHERE_DTYPE, PRIM_CUBE, PRIM_PLANE,
                  PRIM_POLYGON, PRIM_SPHERE)
from .api import MAT_DIELECTRIC, MAT_EMISSIVE, MAT_LAMBERTIAN, MAT_METAL


@dataclass
class Scene:
    name: str
    spheres: np.ndarray  # SPHERE_DTYPE [n]
    mat_id: np.ndarray  # uint32 [n]
    mats: np.ndarray  # MATERIAL_DTYPE [m]
    camera: np.ndarray  # CAMERA_DTYPE [1]
    lights: np.ndarray = None  # LIGHT_DTYPE [l] (Phong point lights; used with direct lighting only)
    prims: np.ndarray = None  # PRIM_DTYPE [p]: the general primitive list in insertion order
    verts: np.ndarray = None  # float32 [v, 3]: polygon vertices
    needs_general: bool = False  # holds planes / cubes / polygons or selected primitives

    @property
    def n_prims(self) -> int:
        return len(self.prims) if self.prims is not None else 0

    @property
    def n_spheres(self) -> int:
        return len(self.spheres)


def make_camera(center=(0.0, 0.0, 0.0), direction=(0.0, 0.0, 1.0)) -> np.ndarray:
    """Camera::setCenter/setDirection + updateViewPort (reference src/Camera.cpp:206-236)."""
    c = np.asarray(center, dtype=np.float64)
    d = np.asarray(direction, dtype=np.float64)
    d = d / np.sqrt(np.dot(d, d))
    up = np.array([0.0, 0.0, 1.0])
    if abs(np.dot(d, up)) > 0.999:
        up = np.array([0.0, 1.0, 0.0])
    right = np.cross(d, up)
    right = right / np.sqrt(np.dot(right, right))
    down = np.cross(d, right)
    down = down / np.sqrt(np.dot(down, down))
    ul = c + d * 1.0 - right * 1.0 * 0.5 - down * 1.0 * 0.5
    cam = np.zeros(1, dtype=CAMERA_DTYPE)
    cam["center"][0] = c
    cam["upper_left"][0] = ul
    cam["right"][0] = right
    cam["down"][0] = down
    return cam


class _Builder:
    def __init__(self):
        self.sph = []
        self.mid = []
        self.mats = []
        self.lights = []
        self.prims = []  # (type, mat, p[8], first_vertex, n_vertices, selected)
        self.verts = []

    def plane(self, point, normal, mat):
        """PlaneObject(point, normal, material): the constructor normalises the normal
        (reference inc/RTObjects.h:167-172)."""
        n = np.asarray(normal, dtype=np.float64)
        n = n / np.sqrt(np.dot(n, n))
        self.prims.append([PRIM_PLANE, mat, [*map(float, point), *map(float, n), 0.0, 0.0], 0, 0, 0])

    def cube(self, center, half_size, mat):
        """CubeObject(halfSize, material) at `center` (reference inc/RTObjects.h:543-549)."""
        self.prims.append([PRIM_CUBE, mat, [*map(float, center), *map(float, half_size), 0.0, 0.0], 0, 0, 0])

    def polygon(self, vertices, mat):
        """PolygonObject(vertices, material): normal from the first three vertices, centroid = vertex mean
        (computeNormalAndCentroid, reference inc/RTObjects.h:413-442; the mean is scaled in float there)."""
        v = np.asarray(vertices, dtype=np.float32).astype(np.float64)
        c = v.sum(axis=0) * np.float64(np.float32(1.0) / np.float32(len(v)))
        if len(v) >= 3:
            n = np.cross(v[1] - v[0], v[2] - v[0])
            n = n / np.sqrt(np.dot(n, n))
        else:
            n = np.array([0.0, 0.0, 1.0])
        first = len(self.verts)
        self.verts.extend(v.tolist())
        self.prims.append([PRIM_POLYGON, mat, [*n, *c, 0.0, 0.0], first, len(v), 0])

    def select(self, prim_index, flag=True):
        """Primitives::setSelectFlag (reference inc/RTObjects.h:79)."""
        self.prims[prim_index][5] = 1 if flag else 0

    def light(self, position, ambient, diffuse, specular, view_pow):
        self.lights.append((tuple(position), tuple(ambient), tuple(diffuse), tuple(specular), float(view_pow)))

    def mat(self, type_, albedo=(0, 0, 0), fuzz=0.0, ior=1.0, emission=(0, 0, 0)) -> int:
        self.mats.append((type_, tuple(albedo), min(float(fuzz), 1.0), float(ior), tuple(emission)))
        return len(self.mats) - 1

    def lambertian(self, albedo):
        return self.mat(MAT_LAMBERTIAN, albedo)

    def metal(self, albedo, fuzz):
        return self.mat(MAT_METAL, albedo, fuzz=fuzz)

    def dielectric(self, tint, ior):
        return self.mat(MAT_DIELECTRIC, tint, ior=ior)

    def emissive(self, emission):
        return self.mat(MAT_EMISSIVE, emission=emission)

    def sphere(self, center, radius, mat):
        self.sph.append((float(center[0]), float(center[1]), float(center[2]), float(radius)))
        self.mid.append(mat)
        self.prims.append([PRIM_SPHERE, mat, [float(center[0]), float(center[1]), float(center[2]), float(radius),
                                              0.0, 0.0, 0.0, 0.0], 0, 0, 0])

    def build(self, name, camera) -> Scene:
        spheres = np.array(self.sph, dtype=np.float32).reshape(-1, 4).view(SPHERE_DTYPE).reshape(-1)
        mats = np.zeros(len(self.mats), dtype=MATERIAL_DTYPE)
        for i, (t, a, f, ior, e) in enumerate(self.mats):
            mats[i]["type"] = t
            mats[i]["albedo"] = a
            mats[i]["fuzz"] = f
            mats[i]["ior"] = ior
            mats[i]["emission"] = e
            # RTMaterial::diffuse_ is set by RTLambertian's constructor only (reference inc/RTMaterial.h:65)
            mats[i]["diffuse"] = a if t == MAT_LAMBERTIAN else (0, 0, 0)
        lights = np.zeros(len(self.lights), dtype=LIGHT_DTYPE)
        for i, (pos, amb, dif, spec, vp) in enumerate(self.lights):
            lights[i]["position"] = pos
            lights[i]["ambient"] = amb
            lights[i]["diffuse"] = dif
            lights[i]["specular"] = spec
            lights[i]["view_pow"] = vp
        prims = np.zeros(len(self.prims), dtype=PRIM_DTYPE)
        for i, (t, m, pp, fv, nv, sel) in enumerate(self.prims):
            prims[i]["type"] = t
            prims[i]["mat"] = m
            prims[i]["p"] = pp
            prims[i]["first_vertex"] = fv
            prims[i]["n_vertices"] = nv
            prims[i]["selected"] = sel
        verts = np.array(self.verts, dtype=np.float32).reshape(-1, 3)
        general = any(pr[0] != PRIM_SPHERE or pr[5] for pr in self.prims)
        return Scene(name, spheres, np.array(self.mid, dtype=np.uint32), mats, camera, lights, prims, verts, general)


def small_scene(bubble: bool = False, light: bool = False, phong: int = 0) -> Scene:
    """Ground + diffuse / metal / glass spheres in front of the default camera
    (centre 0, direction +z, so world-up falls back to +y)."""
    b = _Builder()
    ground = b.lambertian((0.8, 0.8, 0.0))
    center = b.lambertian((0.1, 0.2, 0.5))
    left = b.metal((0.8, 0.6, 0.2), 0.3)
    right = b.dielectric((1.0, 1.0, 1.0), 1.5)
    b.sphere((0.0, -100.5, 3.0), 100.0, ground)
    b.sphere((0.0, 0.0, 3.0), 0.5, center)
    b.sphere((-1.0, 0.0, 3.0), 0.5, left)
    b.sphere((1.0, 0.0, 3.0), 0.5, right)
    if bubble:  # hollow glass: negative radius flips the normal (reference inc/RTObjects.h:154)
        b.sphere((1.0, 0.0, 3.0), -0.4, right)
    if light:
        b.sphere((0.0, 1.5, 3.0), 0.4, b.emissive((4.0, 4.0, 4.0)))
    if phong >= 1:  # Phong point lights (class Light, reference inc/RTObjects.h:220-321)
        b.light((2.0, 3.0, 1.0), (0.05, 0.05, 0.05), (0.6, 0.6, 0.5), (1.0, 1.0, 1.0), 16.0)
    if phong >= 2:
        b.light((-3.0, 1.0, 4.5), (0.0, 0.0, 0.02), (0.2, 0.3, 0.5), (1.0, 1.0, 1.0), 4.0)
    return b.build("small", make_camera((0, 0, 0), (0, 0, 1)))


def mixed_scene(selected: bool = False, phong: int = 0) -> Scene:
    """Every primitive kind of the reference in one scene: a ground plane, an axis-aligned box, a quad and a
    triangle, and spheres of three materials, in front of the default camera (+z, world-up +y)."""
    b = _Builder()
    ground = b.lambertian((0.6, 0.6, 0.5))
    blue = b.lambertian((0.1, 0.2, 0.5))
    gold = b.metal((0.8, 0.6, 0.2), 0.1)
    glass = b.dielectric((1.0, 1.0, 1.0), 1.5)
    red = b.lambertian((0.7, 0.15, 0.1))
    green = b.lambertian((0.2, 0.6, 0.2))
    b.plane((0.0, -0.5, 0.0), (0.0, 1.0, 0.0), ground)                       # 0
    b.sphere((0.0, 0.0, 3.0), 0.5, blue)                                      # 1
    b.cube((-1.3, -0.1, 3.2), (0.4, 0.4, 0.4), gold)                          # 2
    b.sphere((1.1, 0.0, 2.8), 0.5, glass)                                     # 3
    b.polygon([(-2.0, -0.5, 5.0), (2.0, -0.5, 5.0), (2.0, 1.5, 5.0), (-2.0, 1.5, 5.0)], red)   # 4: back wall quad
    b.polygon([(0.2, 0.6, 2.2), (0.9, 0.9, 2.4), (0.5, 1.3, 2.6)], green)    # 5: floating triangle
    b.cube((0.3, -0.3, 1.8), (0.15, 0.2, 0.15), glass)                        # 6: small glass box
    if selected:
        b.select(1)
        b.select(2)
        b.select(5)
    if phong >= 1:
        b.light((2.0, 3.0, 1.0), (0.05, 0.05, 0.05), (0.6, 0.6, 0.5), (1.0, 1.0, 1.0), 16.0)
    return b.build("mixed", make_camera((0, 0, 0), (0, 0, 1)))


class _Lcg:
    """Tiny deterministic generator for scene layout (independent of the render RNG)."""

    def __init__(self, seed):
        self.s = seed & 0xFFFFFFFFFFFFFFFF

    def u(self) -> float:
        self.s = (self.s * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
        return ((self.s >> 40) & 0xFFFFFF) / float(1 << 24)

    def rng(self, lo, hi) -> float:
        return lo + (hi - lo) * self.u()


def random_spheres_scene(seed: int = 12345, lit: bool = False, floor: bool = False) -> Scene:
    """The RTOW "final scene" turned z-up: RTOW (x, y, z) -> (x, -z, y)."""
    g = _Lcg(seed)
    b = _Builder()
    b.sphere((0.0, 0.0, -1000.0), 1000.0, b.lambertian((0.5, 0.5, 0.5)))
    for a in range(-11, 11):
        for bb in range(-11, 11):
            choose = g.u()
            cx = a + 0.9 * g.u()
            cy = -(bb + 0.9 * g.u())
            cz = 0.2
            if np.sqrt((cx - 4.0) ** 2 + cy ** 2 + (cz - 0.2) ** 2) <= 0.9:
                continue
            if choose < 0.8:
                alb = (g.u() * g.u(), g.u() * g.u(), g.u() * g.u())
                m = b.lambertian(alb)
            elif choose < 0.95:
                alb = (g.rng(0.5, 1), g.rng(0.5, 1), g.rng(0.5, 1))
                m = b.metal(alb, g.rng(0, 0.5))
            else:
                m = b.dielectric((1.0, 1.0, 1.0), 1.5)
            b.sphere((cx, cy, cz), 0.2, m)
    b.sphere((0.0, 0.0, 1.0), 1.0, b.dielectric((1.0, 1.0, 1.0), 1.5))
    b.sphere((-4.0, 0.0, 1.0), 1.0, b.lambertian((0.4, 0.2, 0.1)))
    b.sphere((4.0, 0.0, 1.0), 1.0, b.metal((0.7, 0.6, 0.5), 0.0))
    name = "random_spheres"
    if lit:
        # two Phong point lights above the field: every hit then sends a shadow query per light
        # (Camera::computeDirectLighting, reference src/Camera.cpp:161-178)
        b.light((6.0, -4.0, 8.0), (0.03, 0.03, 0.03), (0.5, 0.5, 0.45), (0.8, 0.8, 0.8), 24.0)
        b.light((-8.0, 5.0, 6.0), (0.02, 0.02, 0.03), (0.25, 0.3, 0.4), (0.5, 0.5, 0.5), 12.0)
        name = "random_spheres_lit"
    if floor:
        # a ground plane instead of relying on the big sphere alone, and one selected sphere: the general kernel
        # with a long sphere list
        b.plane((0.0, 0.0, -0.001), (0.0, 0.0, 1.0), b.lambertian((0.45, 0.5, 0.45)))
        b.cube((2.0, -2.5, 0.5), (0.5, 0.5, 0.5), b.metal((0.8, 0.7, 0.6), 0.05))
        b.select(10)
        name += "_floor"
    return b.build(name, make_camera((13.0, -3.0, 2.0), (-13.0, 3.0, -2.0)))


def halo_overlap_scene() -> Scene:
    """Two SELECTED spheres whose halos interact: the grown shell (1.05 r, reference inc/RTObjects.h:26,96-105)
    of the big sphere A, first in insertion order, is crossed by rays that then meet the small selected
    sphere B in front of it; an unselected sphere and the ground complete the picture."""
    b = _Builder()
    b.sphere((0.0, 0.0, 5.0), 1.0, b.lambertian((0.2, 0.3, 0.7)))        # 0: A
    b.sphere((0.8, 0.0, 3.0), 0.3, b.lambertian((0.7, 0.6, 0.2)))        # 1: B, over A's silhouette
    b.sphere((-0.9, 0.3, 3.4), 0.35, b.metal((0.8, 0.8, 0.8), 0.05))     # 2: unselected, over the other edge
    b.sphere((0.0, -100.5, 3.0), 99.5, b.lambertian((0.5, 0.5, 0.5)))    # 3: ground
    b.select(0)
    b.select(1)
    return b.build("halo_overlap", make_camera((0, 0, 0), (0, 0, 1)))


def grid_scene(nx: int = 100, ny: int = 100, seed: int = 777) -> Scene:
    """nx*ny spheres on a unit-pitch grid (+ ground): 70 % glass, 15 % metal, 15 % diffuse."""
    g = _Lcg(seed)
    b = _Builder()
    b.sphere((0.0, 0.0, -1000.0), 1000.0, b.lambertian((0.5, 0.5, 0.5)))
    for j in range(ny):
        for i in range(nx):
            cx = (i - nx / 2 + 0.5) + g.rng(-0.05, 0.05)
            cy = (j - ny / 2 + 0.5) + g.rng(-0.05, 0.05)
            choose = g.u()
            if choose < 0.70:
                m = b.dielectric((g.rng(0.8, 1), g.rng(0.8, 1), g.rng(0.8, 1)), 1.5)
            elif choose < 0.85:
                m = b.metal((g.rng(0.5, 1), g.rng(0.5, 1), g.rng(0.5, 1)), g.rng(0, 0.3))
            else:
                m = b.lambertian((g.u() * g.u(), g.u() * g.u(), g.u() * g.u()))
            b.sphere((cx, cy, 0.4), 0.4, m)
    ext = 0.5 * max(nx, ny)
    return b.build(f"grid{nx}x{ny}", make_camera((0.9 * ext, -1.1 * ext, 0.6 * ext), (-0.9, 1.1, -0.6)))


def scene_text(scene: Scene) -> str:
    """The scene as a scene file of the C++ host layer (SceneIO.h: the reference's one-line text records
    of materials, primitives and lights), 9 significant digits, so that render_scene and the Camera API
    see the floats this module holds.  Sphere-only scenes and general ones alike."""
    f = lambda v: " ".join(f"{float(x):.9g}" for x in v)
    cam = scene.camera[0]
    c = cam["center"].astype(np.float64)
    d = cam["upper_left"].astype(np.float64) + 0.5 * cam["right"].astype(np.float64) + 0.5 * cam["down"].astype(np.float64) - c
    d = d / np.sqrt(np.dot(d, d))
    out = ["# written by raytracer_amd.scenes.scene_text: " + scene.name, f"Camera {f(c)} {f(d)}"]
    zero = (0.0, 0.0, 0.0)
    for m in scene.mats:
        t = int(m["type"])
        if t == MAT_LAMBERTIAN:
            out.append(f"Material Lambertian {f(m['albedo'])} {f(zero)} {f(m['emission'])}")
        elif t == MAT_METAL:
            out.append(f"Material Metal {f(m['diffuse'])} {f(m['albedo'])} {f(m['emission'])} {float(m['fuzz']):.9g}")
        elif t == MAT_DIELECTRIC:
            out.append(f"Material Dielectric {f(m['diffuse'])} {f(zero)} {f(m['emission'])} {f(m['albedo'])} "
                       f"{float(m['ior']):.9g}")
        else:
            out.append(f"Material Emissive {f(m['diffuse'])} {f(zero)} {f(m['emission'])}")
    if scene.needs_general:
        for pr in scene.prims:
            t, p8, sel = int(pr["type"]), pr["p"], int(pr["selected"])
            if t == PRIM_SPHERE:
                out.append(f"Object {int(pr['mat'])} Sphere {f(p8[:3])} {sel} {float(p8[3]):.9g}")
            elif t == PRIM_PLANE:
                out.append(f"Object {int(pr['mat'])} Plane {f(p8[:3])} {sel} {f(p8[3:6])}")
            elif t == PRIM_CUBE:
                out.append(f"Object {int(pr['mat'])} Cube {f(p8[:3])} {sel} {f(p8[3:6])}")
            else:
                v = scene.verts[int(pr["first_vertex"]):int(pr["first_vertex"]) + int(pr["n_vertices"])]
                out.append(f"Object {int(pr['mat'])} Polygon {f(p8[3:6])} {sel} {len(v)} {f(v.reshape(-1))}")
    else:
        for sp, mi in zip(scene.spheres, scene.mat_id):
            out.append(f"Object {int(mi)} Sphere {f((sp['cx'], sp['cy'], sp['cz']))} 0 {float(sp['r']):.9g}")
    if scene.lights is not None:
        for l in scene.lights:
            out.append(f"Light {f(l['position'])} {f(l['ambient'])} {f(l['diffuse'])} {f(l['specular'])} "
                       f"{float(l['view_pow']):.9g}")
    return "\n".join(out) + "\n"


def by_name(name: str) -> Scene:
    if name == "small":
        return small_scene()
    if name == "small_bubble":
        return small_scene(bubble=True)
    if name == "small_light":
        return small_scene(light=True)
    if name == "small_phong":
        return small_scene(phong=2)
    if name == "mixed":
        return mixed_scene()
    if name == "mixed_selected":
        return mixed_scene(selected=True)
    if name == "mixed_lit":
        return mixed_scene(selected=True, phong=1)
    if name == "random_spheres_lit":
        return random_spheres_scene(lit=True)
    if name == "random_spheres_floor":
        return random_spheres_scene(floor=True)
    if name == "random_spheres_lit_floor":
        return random_spheres_scene(lit=True, floor=True)
    if name == "halo_overlap":
        return halo_overlap_scene()
    if name == "random_spheres":
        return random_spheres_scene()
    if name == "grid10k":
        return grid_scene(100, 100)
    if name.startswith("grid"):
        n = int(name[4:])
        return grid_scene(n, n)
    raise KeyError(name)
