"""Tuning / confidence aid: many random sphere clouds rendered with the matrix-core filter and with the packed filter;
frames, sums and ray counts must be equal (the scenes of tests/test_gpu_filter.py's random-cloud test, more seeds).
usage: [LIT=1] stress_filters.py [first_seed last_seed]      (LIT=1: with a light, i.e. through the general kernel)"""
import os
import sys
import time
sys.path.insert(0, '.')
import importlib.util
spec = importlib.util.spec_from_file_location('tf', 'tests/test_gpu_filter.py')
tf = importlib.util.module_from_spec(spec)
spec.loader.exec_module(tf)


class Env:
    def setenv(self, k, v):
        os.environ[k] = v


first, last = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (100, 400)
different, not_in_view, t0 = 0, 0, time.time()
for seed in range(first, last):
    os.environ.pop("RT_AMD_SPHERE_SOURCE", None)
    same, in_view, what = tf.random_cloud_case(seed, Env().setenv, lit=os.environ.get("LIT") == "1")
    if not same:
        different += 1
        print("seed", seed, "DIFFERENT", what)
    elif not in_view:
        not_in_view += 1  # a comparison that says little (nothing to bounce off), not a failure
print(f"{last - first} scenes in {time.time() - t0:.1f} s: different {different}, equal but with nothing in view {not_in_view}")
sys.exit(1 if different else 0)
