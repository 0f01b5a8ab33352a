"""CPU sanitizer runs (AddressSanitizer + UndefinedBehaviorSanitizer): the oracle compiled into a driver
that calls every entry point over every primitive / material kind, and the C++ host layer's GPU-free
modes (poses, rejection paths, scene files).  `make asan` builds and runs them; a finding aborts the run."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(directory):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0")
    return subprocess.run(["make", "-C", os.path.join(ROOT, directory), "-s", "asan"], capture_output=True, text=True,
                          env=env, timeout=600)


def test_oracle_under_asan_and_ubsan():
    r = _run("oracle")
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("oracle sanitizer run ok") == 2  # double and float builds
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr


def test_host_layer_under_asan_and_ubsan():
    if not os.path.exists(os.path.join(ROOT, "raytracer_amd", "librt_amd.so")):
        import __graft_entry__ as g

        g.build_hip()
    r = _run(os.path.join("raytracer_amd", "host"))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "host layer sanitizer run ok" in r.stdout
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
