"""Run one case of tests/test_gpu_fuzz.py::test_random_scene_matches_oracle by its seed and print what differs.
usage: python tools/fuzz_case.py <case_seed>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np
import test_gpu_fuzz as tf
seed = int(sys.argv[1])
c = tf._case(np.random.default_rng(1000 + seed))
print(c)
try:
    tf._check_case(c)
    print("PASS")
except AssertionError as e:
    print("FAIL", str(e)[:3000])
