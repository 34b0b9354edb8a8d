"""the wide-row slab fuzz of tests/test_gpu_fuzz.py over 500 seeds, every failing configuration printed in full (pytest abbreviates them)"""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import test_gpu_fuzz as t
bad = []
for seed in range(500):
    c = t.draw_wide_slabs(seed)
    try:
        t.check_random_slabs(c)
    except AssertionError as e:
        print("FAIL", seed, {k: v for k, v in c.items()}, str(e)[-60:].replace("\n", " "), flush=True)
        bad.append(seed)
print("bad", bad)
