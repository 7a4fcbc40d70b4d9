"""One-off wide sweep of the randomised parity cases of tests/test_gpu_fuzz.py (seeds beyond the ones
the test suite runs).  usage: python tests/tools/fuzz_sweep.py [first_seed] [count]"""
import os, sys, traceback, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import test_gpu_fuzz as F
first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 300
bad = 0
for seed in range(first, first + count):
    for fn in (F.test_random_configuration_matches_oracle, F.test_random_torchgate_matches_oracle):
        try:
            fn(seed)
        except BaseException as e:
            bad += 1
            case = F._case(seed) if fn is F.test_random_configuration_matches_oracle else F._case_T(seed)
            print("FAIL", fn.__name__, seed, type(e).__name__, str(e)[:300].replace("\n", " "), case, flush=True)
print("done", count, "seeds,", bad, "failures")
