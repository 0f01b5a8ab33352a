"""The reference-shaped C++ host layer (raytracer_amd/host): Camera / SceneManager / SphereObject /
RTMaterialManager above the C ABI.  The driver binary is built by __graft_entry__.build()."""
import json
import os
import subprocess

import numpy as np
import pytest

from raytracer_amd import scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEMO = os.path.join(ROOT, "raytracer_amd", "host", "tests", "host_demo.bin")
KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kats.json")))


def _need_demo():
    if not os.path.exists(DEMO):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "raytracer_amd", "host"), "-s"])
    return DEMO


def test_camera_pose_matches_kats_and_oracle(oracle64):
    """Camera::setCenter/setDirection + updateViewPort (reference src/Camera.cpp:206-236)."""
    demo = _need_demo()
    cases = [(k["center"], k["direction"]) for k in KATS["viewport"]] + [((13, -3, 2), (-13, 3, -2))]
    for c, d in cases:
        out = subprocess.check_output([demo, "--pose"] + [str(float(x)) for x in list(c) + list(d)], text=True)
        ul, right, down = [np.array([float(x) for x in line.split()]) for line in out.strip().splitlines()]
        cam = oracle64.update_viewport(c, d)
        assert np.allclose(ul, cam["upper_left"][0], atol=1e-6)
        assert np.allclose(right, cam["right"][0], atol=1e-6)
        assert np.allclose(down, cam["down"][0], atol=1e-6)
    for k in KATS["viewport"]:
        out = subprocess.check_output([demo, "--pose"] + [str(float(x)) for x in k["center"] + k["direction"]], text=True)
        ul = np.array([float(x) for x in out.splitlines()[0].split()])
        assert np.allclose(ul, k["upper_left"], atol=1e-12)


def test_unsupported_scenes_are_rejected_not_rendered_differently():
    """Selected objects and a short output buffer raise BackendError before any GPU work
    (there is no CPU fallback to route them to)."""
    out = subprocess.run([_need_demo(), "--unsupported"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "caught 2" in out.stdout


@pytest.mark.gpu
def test_camera_render_matches_oracle(tmp_path, oracle64):
    """Camera::render through the host API == the oracle's render of the same scene and seed."""
    W, H, spp, depth = 160, 90, 128, 10
    raw = str(tmp_path / "img.rgba")
    ppm = str(tmp_path / "img.ppm")
    r = subprocess.run([_need_demo(), raw, str(W), str(H), str(spp), str(depth), ppm], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    img = np.fromfile(raw, np.uint8).reshape(H, W, 4)
    ref, _, rays = oracle64.render(scenes.small_scene(), W, H, spp, depth, seed=21)
    other, _, _ = oracle64.render(scenes.small_scene(), W, H, spp, depth, seed=22)
    d = img[:, :, :3].astype(float) - ref[:, :, :3]
    floor = np.mean((other[:, :, :3].astype(float) - ref[:, :, :3]) ** 2)
    assert np.mean(d * d) <= floor, "host-API render is further from the oracle than another seed is"
    assert np.abs(d.mean(axis=(0, 1))).max() <= 0.25
    assert np.all(img[:, :, 3] == 255)
    assert open(ppm, "rb").read(2) == b"P6"
    got_rays = int(r.stdout.strip().split()[-1])
    assert abs(got_rays - rays) <= 0.004 * rays


@pytest.mark.gpu
def test_camera_render_with_lights_matches_oracle(tmp_path, oracle64):
    """enableLDirect with two Phong lights through the host API (Camera::computeDirectLighting,
    reference src/Camera.cpp:161-178)."""
    W, H, spp, depth = 160, 90, 64, 8
    raw = str(tmp_path / "lit.rgba")
    r = subprocess.run([_need_demo(), raw, str(W), str(H), str(spp), str(depth), "-", "lit"], capture_output=True,
                       text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    img = np.fromfile(raw, np.uint8).reshape(H, W, 4)
    scene = scenes.small_scene(phong=2)
    ref, _, rays = oracle64.render(scene, W, H, spp, depth, seed=21, direct_light=True)
    other, _, _ = oracle64.render(scene, W, H, spp, depth, seed=22, direct_light=True)
    d = img[:, :, :3].astype(float) - ref[:, :, :3]
    assert np.mean(d * d) <= np.mean((other[:, :, :3].astype(float) - ref[:, :, :3]) ** 2)
    assert np.abs(d.mean(axis=(0, 1))).max() <= 0.25
    assert abs(int(r.stdout.strip().split()[-1]) - rays) <= 0.004 * rays
