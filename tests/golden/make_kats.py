#!/usr/bin/env python3
"""Writes tests/golden/kats.json: known-answer vectors for the render path, derived by hand
(closed-form geometry and optics written out below), NOT by running the oracle or the GPU code.
The reference ships no vectors of its own, so these are what pins the oracle.

Run:  python tests/golden/make_kats.py
"""
import json
import math
import os

S2 = math.sqrt(2.0)
kats = {}

# ---- SphereObject::hitDetail (reference inc/RTObjects.h:129-159) --------------------------
# each: origin, direction, centre, radius, tmin -> hit?, t, point, unit normal (against the ray), front face
kats["hit"] = [
    dict(name="head-on from outside", o=[0, 0, 0], d=[0, 0, 1], c=[0, 0, 3], r=0.5, tmin=0.001,
         hit=True, t=2.5, p=[0, 0, 2.5], n=[0, 0, -1], front=True),
    dict(name="direction of length 2 halves t", o=[0, 0, 0], d=[0, 0, 2], c=[0, 0, 3], r=0.5, tmin=0.001,
         hit=True, t=1.25, p=[0, 0, 2.5], n=[0, 0, -1], front=True),
    dict(name="from the centre: far root, back face", o=[0, 0, 3], d=[0, 0, 1], c=[0, 0, 3], r=0.5, tmin=0.001,
         hit=True, t=0.5, p=[0, 0, 3.5], n=[0, 0, -1], front=False),
    dict(name="sphere behind the ray", o=[0, 0, 0], d=[0, 0, -1], c=[0, 0, 3], r=0.5, tmin=0.001, hit=False),
    dict(name="clear miss", o=[0, 0, 0], d=[0, 1, 0], c=[0, 0, 3], r=0.5, tmin=0.001, hit=False),
    dict(name="negative radius flips the outward normal", o=[0, 2, 0], d=[0, 0, 1], c=[0, 2, 3], r=-0.5, tmin=0.001,
         hit=True, t=2.5, p=[0, 2, 2.5], n=[0, 0, -1], front=False),
    dict(name="3-4-5 triangle, oblique", o=[0, 0, 0], d=[0.6, 0, 0.8], c=[3, 0, 4], r=1.0, tmin=0.001,
         hit=True, t=4.0, p=[2.4, 0, 3.2], n=[-0.6, 0, -0.8], front=True),
    dict(name="near root equals tmin exactly -> open interval takes the far root",
         o=[0, 0, 0], d=[0, 0, 1], c=[0, 0, 1.5], r=0.5, tmin=1.0,
         hit=True, t=2.0, p=[0, 0, 2.0], n=[0, 0, -1], front=False),
    dict(name="both roots at or below tmin", o=[0, 0, 0], d=[0, 0, 1], c=[0, 0, 1.5], r=0.5, tmin=2.0, hit=False),
    dict(name="off-axis chord: x offset 0.3 of r 0.5 -> half chord 0.4", o=[0.3, 0, 0], d=[0, 0, 1], c=[0, 0, 5], r=0.5,
         tmin=0.001, hit=True, t=4.6, p=[0.3, 0, 4.6], n=[0.6, 0, -0.8], front=True),
]

# ---- SceneManager::hitClosest (reference src/RayTracer.cpp:68-82): ties and order -----------
kats["closest"] = dict(
    spheres=[[0, 0, 3, 0.5], [0, 0, 3, 0.5], [0, 0, 6, 1.0], [0, 0, -4, 1.0]],
    rays=[
        dict(o=[0, 0, 0], d=[0, 0, 1], id=0, t=2.5),      # exact duplicate: strict '<' keeps the lower index
        dict(o=[0, 0, 4], d=[0, 0, 1], id=2, t=1.0),      # starts behind the first two
        dict(o=[0, 0, 0], d=[0, 0, -1], id=3, t=3.0),
        dict(o=[0, 0, 0], d=[1, 0, 0], id=-1, t=0.0),
    ],
)

# ---- RTMaterial::scatter (reference inc/RTMaterial.h) ----------------------------------------
# random_unit(u0, u1) as defined by the oracle header: z = 1-2*u0, phi = 2*pi*u1.
#   u0 = 0.5, u1 = 0    -> (1, 0, 0);   u0 = 0 -> (0, 0, 1);   u0 = 0.5, u1 = 0.25 -> (0, 1, 0)
kats["scatter"] = [
    dict(name="lambertian: normal + unit vector, unnormalised", type=0, albedo=[0.1, 0.2, 0.5], fuzz=0, ior=1,
         d=[0, -1, 0], p=[1, 2, 3], n=[0, 1, 0], front=1, u=[0.5, 0.0, 0.0],
         scattered=1, att=[0.1, 0.2, 0.5], o=[1, 2, 3], dir=[1, 1, 0]),
    dict(name="lambertian: unit vector cancels the normal -> nearZero falls back to the normal", type=0,
         albedo=[0.3, 0.3, 0.3], fuzz=0, ior=1, d=[0, 0, 1], p=[0, 0, 0], n=[0, 0, -1], front=1, u=[0.0, 0.0, 0.0],
         scattered=1, att=[0.3, 0.3, 0.3], o=[0, 0, 0], dir=[0, 0, -1]),
    dict(name="metal, fuzz 0: mirror reflection of a 45 degree ray, normalised", type=1, albedo=[0.8, 0.6, 0.2],
         fuzz=0.0, ior=1, d=[2, -2, 0], p=[0, 0, 0], n=[0, 1, 0], front=1, u=[0.3, 0.7, 0.0],
         scattered=1, att=[0.8, 0.6, 0.2], o=[0, 0, 0], dir=[1 / S2, 1 / S2, 0]),
    dict(name="metal, fuzz 0.5: reflection + 0.5 * unit vector", type=1, albedo=[1, 1, 1], fuzz=0.5, ior=1,
         d=[0, -1, 0], p=[0, 0, 0], n=[0, 1, 0], front=1, u=[0.5, 0.25, 0.0],
         scattered=1, att=[1, 1, 1], o=[0, 0, 0], dir=[0, 1.5, 0]),
    # dielectric, normal incidence, front face: eta = 1/1.5, cos = 1, Schlick r0 = 0.04.
    # u2 = 0.5 > 0.04 -> refracts straight on; origin nudged +1e-4 along the normal.
    dict(name="glass, normal incidence, refracts", type=2, albedo=[1, 1, 1], fuzz=0, ior=1.5,
         d=[0, -1, 0], p=[0, 0, 0], n=[0, 1, 0], front=1, u=[0.0, 0.0, 0.5],
         scattered=1, att=[1, 1, 1], o=[0, 1e-4, 0], dir=[0, -1, 0]),
    # same, u2 = 0.01 < 0.04 -> reflects
    dict(name="glass, normal incidence, reflects", type=2, albedo=[0.9, 1, 1], fuzz=0, ior=1.5,
         d=[0, -1, 0], p=[0, 0, 0], n=[0, 1, 0], front=1, u=[0.0, 0.0, 0.01],
         scattered=1, att=[0.9, 1, 1], o=[0, 1e-4, 0], dir=[0, 1, 0]),
    # 45 degree incidence from air: sin(t) = sin(45)/1.5 = 0.4714045, cos(t) = 0.8819171
    # reflectance(cos 45) = 0.04 + 0.96 * (1 - 0.7071068)^5 = 0.04207 < 0.5 -> refracts
    dict(name="glass, 45 degrees, Snell", type=2, albedo=[1, 1, 1], fuzz=0, ior=1.5,
         d=[1, -1, 0], p=[0, 0, 0], n=[0, 1, 0], front=1, u=[0.0, 0.0, 0.5],
         scattered=1, att=[1, 1, 1], o=[0, 1e-4, 0],
         dir=[(1 / S2) / 1.5, -math.sqrt(1 - ((1 / S2) / 1.5) ** 2), 0]),
    # leaving glass at 60 degrees: eta*sin = 1.5*0.866 > 1 -> total internal reflection whatever u2 is;
    # back face: origin nudged -1e-4 along the (inward-facing) normal
    dict(name="glass, total internal reflection", type=2, albedo=[1, 1, 1], fuzz=0, ior=1.5,
         d=[math.sqrt(3) / 2, -0.5, 0], p=[0, 0, 0], n=[0, 1, 0], front=0, u=[0.0, 0.0, 0.999],
         scattered=1, att=[1, 1, 1], o=[0, -1e-4, 0], dir=[math.sqrt(3) / 2, 0.5, 0]),
    dict(name="emissive absorbs", type=3, albedo=[0, 0, 0], fuzz=0, ior=1, d=[0, -1, 0], p=[0, 0, 0], n=[0, 1, 0],
         front=1, u=[0.1, 0.2, 0.3], scattered=0),
]

# ---- convertRTColor / linearToGamma (reference src/Camera.cpp:11-34) --------------------------
kats["tonemap"] = [
    dict(rgb=[0.0, 1.0, 0.25], rgba=[0, 255, 128, 255]),          # sqrt(0.25)=0.5 -> 128; 1 clamps to 0.999 -> 255
    dict(rgb=[-1.0, 4.0, 0.5], rgba=[0, 255, 181, 255]),          # negative -> 0; sqrt(0.5)*256 = 181.02
    dict(rgb=[0.0625, 0.36, 0.81], rgba=[64, 153, 230, 255]),     # 0.25, 0.6, 0.9 times 256
]

# ---- Camera::updateViewPort (reference src/Camera.cpp:206-221) --------------------------------
# default camera: centre 0, direction +z -> |d.z| > 0.999 so world-up falls back to +y;
#   right = normalize(d x up) = (0,0,1)x(0,1,0) = (-1,0,0); down = normalize(d x right) = (0,-1,0)
#   upper-left = c + d - right/2 - down/2 = (0.5, 0.5, 1)
# z-up camera looking along +x: right = (1,0,0)x(0,0,1) = (0,-1,0); down = (1,0,0)x(0,-1,0) = (0,0,-1)
kats["viewport"] = [
    dict(center=[0, 0, 0], direction=[0, 0, 1], right=[-1, 0, 0], down=[0, -1, 0], upper_left=[0.5, 0.5, 1]),
    dict(center=[1, 2, 3], direction=[5, 0, 0], right=[0, -1, 0], down=[0, 0, -1], upper_left=[2, 2.5, 3.5]),
]

# ---- sky gradient on a miss (reference src/Camera.cpp:155-156), through a depth-1 render of an
# empty scene: the pixel-centre ray of the middle pixel of an odd-sized frame of the default camera
# points along +z exactly (y = 0) -> a = 0.5 -> (0.75, 0.85, 1.0)
kats["sky"] = dict(direction_y=0.0, rgb=[0.75, 0.85, 1.0])

# ---- Light::getDirectLighting through a depth-1 render (reference inc/RTObjects.h:247-257,300-320,
# src/Camera.cpp:161-178).  Camera at (0,1,0) looking straight down at a Lambertian ground sphere
# (centre (0,-100,0), r 100, diffuse (0.8,0.8,0)): the centre pixel's ray hits P = (0,0,0), n = (0,1,0).
# With maxRayDepth 1 the radiance is emitted (0) + LIndirect (attenuation * black) + LDirect.
#   light A at (0,2,0), ambient 0.1, diffuse 0.5, pow 8: toLight = toView = n, so cos = 1, the mirrored
#     light direction is n again -> specular 1:  0.1*0.8 + 0.5*0.8 + 1 = 1.48 (r,g);  blue: 0 + 0 + 1 = 1
#   blocker: a small sphere at (0,1.5,0) r 0.2 sits between P and the light -> shadowed: ambient only
#   light B at (3,0,0) lies in the tangent plane: cos = 0 -> no Lambert term; mirrored direction
#     (-1,0,0) is orthogonal to toView -> no specular: ambient only even when unshadowed
kats["direct_light"] = dict(
    camera=dict(center=[0, 1, 0], direction=[0, -1, 0]),
    ground=dict(c=[0, -100, 0], r=100.0, diffuse=[0.8, 0.8, 0.0]),
    cases=[
        dict(name="overhead light, unshadowed", light=dict(position=[0, 2, 0], ambient=[0.1] * 3, diffuse=[0.5] * 3,
             view_pow=8.0), blocker=None, rgb=[1.48, 1.48, 1.0]),
        dict(name="overhead light, blocked", light=dict(position=[0, 2, 0], ambient=[0.1] * 3, diffuse=[0.5] * 3,
             view_pow=8.0), blocker=dict(c=[0, 1.5, 0], r=0.2), rgb=[0.08, 0.08, 0.0]),
        dict(name="light in the tangent plane", light=dict(position=[3, 0, 0], ambient=[0.25] * 3, diffuse=[0.5] * 3,
             view_pow=2.0), blocker=None, rgb=[0.2, 0.2, 0.0]),
    ],
)

# ---- the other primitive kinds and the selection halo, through hitClosest ------------------------
# (PlaneObject::hit inc/RTObjects.h:174-190, CubeObject::hitBox :586-644, PolygonObject::hitDetail :469-512,
#  hitExpanded :96-105/:370-386/:560-568, SceneManager::hitClosest src/RayTracer.cpp:68-100)
# prim specs: ["sphere", centre, r] | ["plane", point, normal] | ["cube", centre, half] | ["polygon", vertices];
# "selected" lists the selected indices.  Each ray: expanded_state -> id, t, unit normal, front, expanded.
# normal/front follow HitRecord::setFaceNormal: front = dot(d, outward) < 0, normal faces against the ray.
#   plane: "outward" is first turned ALONG the ray (:184-185), so front is always false and the reported
#          normal is the plane normal turned against the ray
#   polygon: outward is turned against the ray (:502-503), so front is always true
SQ = [[-1, -1, 4], [1, -1, 4], [1, 1, 4], [-1, 1, 4]]     # unit normal (2,0,0)x(2,2,0) = +z, centroid (0,0,4)
TRI = [[2, 0, 0], [2, 2, 0], [2, 0, 2]]                   # in the plane x = 2: (0,2,0)x(0,0,2) = +x
H105 = math.sqrt(1.05 ** 2 - 1.02 ** 2)                   # half chord of the r = 1.05 shell at offset 1.02
kats["prims"] = [
    dict(name="plane y = -1", prims=[["plane", [0, -1, 0], [0, 1, 0]]], selected=[], rays=[
        dict(o=[0, 0, 0], d=[0, -1, 0], es=0, id=0, t=1.0, n=[0, 1, 0], front=0, exp=0),     # (-1)/(-1)
        dict(o=[0, -3, 0], d=[0, 2, 0], es=0, id=0, t=1.0, n=[0, -1, 0], front=0, exp=0),    # from below: 2/2
        dict(o=[0, 0, 0], d=[1, 0, 0], es=0, id=-1),                                         # parallel: -1 * inf
        dict(o=[0, 0, 0], d=[0, 1, 0], es=0, id=-1),                                         # behind: t = -1
        dict(o=[0, 1, 0], d=[0.6, -0.8, 0], es=0, id=0, t=2.5, n=[0, 1, 0], front=0, exp=0), # (-2)/(-0.8)
        dict(o=[0, 1, 0], d=[0.6, -0.8, 0], es=1, id=0, t=2.5, n=[0, 1, 0], front=0, exp=0), # halo pass: same hit
    ]),
    dict(name="cube centre (0,0,5), half (1,2,0.5)", prims=[["cube", [0, 0, 5], [1, 2, 0.5]]], selected=[], rays=[
        dict(o=[0, 0, 0], d=[0, 0, 1], es=0, id=0, t=4.5, n=[0, 0, -1], front=1, exp=0),     # near z face
        dict(o=[0, 0, 0], d=[0, 0, 2], es=0, id=0, t=2.25, n=[0, 0, -1], front=1, exp=0),    # |d| = 2 halves t
        dict(o=[0, 0, 5], d=[1, 0, 0], es=0, id=0, t=1.0, n=[-1, 0, 0], front=0, exp=0),     # from inside: exit face
        dict(o=[3, 0, 0], d=[0, 0, 1], es=0, id=-1),                                         # outside the x slab
        dict(o=[0, 0, 0], d=[0, 0, -1], es=0, id=-1),                                        # box behind the ray
        # x slab [5, 7.5], z slab [4/0.6, 5/0.6]: entry through the z face at t = 20/3
        dict(o=[-5, 0, 0.5], d=[0.8, 0, 0.6], es=0, id=0, t=20 / 3, n=[0, 0, -1], front=1, exp=0),
        # x slab [5, 7.5], z slab [1/0.6, 2/0.6] lies before it: miss
        dict(o=[-5, 0, 3.5], d=[0.8, 0, 0.6], es=0, id=-1),
    ]),
    dict(name="square in z = 4", prims=[["polygon", SQ]], selected=[], rays=[
        dict(o=[0, 0, 0], d=[0, 0, 1], es=0, id=0, t=4.0, n=[0, 0, -1], front=1, exp=0),
        dict(o=[0.5, 0.5, 6], d=[0, 0, -1], es=0, id=0, t=2.0, n=[0, 0, 1], front=1, exp=0),   # from behind
        dict(o=[1.5, 0, 0], d=[0, 0, 1], es=0, id=-1),                                         # beside it
        dict(o=[0, 0, 0], d=[1, 0, 0], es=0, id=-1),                                           # parallel
        dict(o=[0, 0, 5], d=[0, 0, 1], es=0, id=-1),                                           # behind the ray
    ]),
    dict(name="triangle in x = 2 (projection drops x)", prims=[["polygon", TRI]], selected=[], rays=[
        dict(o=[0, 0.5, 0.5], d=[1, 0, 0], es=0, id=0, t=2.0, n=[-1, 0, 0], front=1, exp=0),   # y + z = 1 < 2
        dict(o=[0, 1.5, 1.5], d=[1, 0, 0], es=0, id=-1),                                       # y + z = 3 > 2
    ]),
    dict(name="tie between kinds: insertion order decides", prims=[["plane", [0, 0, 4], [0, 0, 1]],
                                                                     ["sphere", [0, 0, 5], 1.0]], selected=[], rays=[
        dict(o=[0, 0, 0], d=[0, 0, 1], es=0, id=0, t=4.0, n=[0, 0, -1], front=0, exp=0),
    ]),
    dict(name="same tie, sphere first", prims=[["sphere", [0, 0, 5], 1.0], ["plane", [0, 0, 4], [0, 0, 1]]],
         selected=[], rays=[
        dict(o=[0, 0, 0], d=[0, 0, 1], es=0, id=0, t=4.0, n=[0, 0, -1], front=1, exp=0),
    ]),
    dict(name="selected sphere in front of a plane", prims=[["sphere", [0, 0, 5], 1.0], ["plane", [0, 0, 10], [0, 0, -1]]],
         selected=[0], rays=[
        dict(o=[0, 0, 0], d=[0, 0, 1], es=1, id=0, t=4.0, n=[0, 0, -1], front=1, exp=0),     # the sphere itself
        # passes the sphere at x = 1.02 but meets its 1.05 shell: outward = (1.02, 0, -h)/1.05
        dict(o=[1.02, 0, 0], d=[0, 0, 1], es=1, id=0, t=5 - H105, n=[1.02 / 1.05, 0, -H105 / 1.05], front=1, exp=1),
        dict(o=[1.02, 0, 0], d=[0, 0, 1], es=0, id=1, t=10.0, n=[0, 0, -1], front=0, exp=0), # no halo pass
        dict(o=[1.06, 0, 0], d=[0, 0, 1], es=1, id=1, t=10.0, n=[0, 0, -1], front=0, exp=0), # past the shell
    ]),
    dict(name="same, nothing selected", prims=[["sphere", [0, 0, 5], 1.0], ["plane", [0, 0, 10], [0, 0, -1]]],
         selected=[], rays=[
        dict(o=[1.02, 0, 0], d=[0, 0, 1], es=1, id=1, t=10.0, n=[0, 0, -1], front=0, exp=0),
    ]),
    dict(name="halo hidden behind a nearer object", prims=[["sphere", [0, 0, 5], 1.0], ["plane", [0, 0, 3], [0, 0, -1]]],
         selected=[0], rays=[
        dict(o=[1.02, 0, 0], d=[0, 0, 1], es=1, id=1, t=3.0, n=[0, 0, -1], front=0, exp=0),
    ]),
    dict(name="selected cube", prims=[["cube", [0, 0, 5], [1, 1, 1]]], selected=[0], rays=[
        dict(o=[1.02, 0, 0], d=[0, 0, 1], es=1, id=0, t=3.95, n=[0, 0, -1], front=1, exp=1), # grown half 1.05
        dict(o=[1.02, 0, 0], d=[0, 0, 1], es=0, id=-1),
        dict(o=[0.5, 0, 0], d=[0, 0, 1], es=1, id=0, t=4.0, n=[0, 0, -1], front=1, exp=0),
    ]),
    dict(name="selected square", prims=[["polygon", SQ]], selected=[0], rays=[
        dict(o=[1.02, 0, 0], d=[0, 0, 1], es=1, id=0, t=4.0, n=[0, 0, -1], front=1, exp=1),  # vertices at +-1.05
        dict(o=[1.02, 0, 0], d=[0, 0, 1], es=0, id=-1),
        dict(o=[1.06, 0, 0], d=[0, 0, 1], es=1, id=-1),
    ]),
]

# ---- random words: regression vector of the generator both sides share ------------------------
# (pcg4d as written in oracle/rt_oracle.c; values produced by a straight Python transcription)
def pcg4d(x, y, z, w):
    M = 0xFFFFFFFF
    x = (x * 1664525 + 1013904223) & M; y = (y * 1664525 + 1013904223) & M
    z = (z * 1664525 + 1013904223) & M; w = (w * 1664525 + 1013904223) & M
    x = (x + y * w) & M; y = (y + z * x) & M; z = (z + x * y) & M; w = (w + y * z) & M
    x ^= x >> 16; y ^= y >> 16; z ^= z >> 16; w ^= w >> 16
    x = (x + y * w) & M; y = (y + z * x) & M; z = (z + x * y) & M; w = (w + y * z) & M
    return [x, y, z, w]

kats["rng"] = [dict(key=list(k), words=pcg4d(*k)) for k in
               [(0, 0, 0, 0), (1, 2, 3, 4), (809999, 499, 50, 12345), (0xFFFFFFFF, 0, 7, 1)]]

out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kats.json")
with open(out, "w") as f:
    json.dump(kats, f, indent=1)
print("wrote", out)
