"""Which configuration of the native KD step departs from the per-launch engine, and in which gradient tensors (developer aid for tests/test_gpu_train_native.py)."""
import itertools, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import test_gpu_train_native as T

cases = [(False, (True, False, True, True), False)] * 3 if len(sys.argv) > 1 else [(True, (True, True, True, True), True), (False, (True, True, True, True), True), (True, (True, False, True, True), True), (True, (True, True, True, True), False), (False, (True, False, True, True), False)]
for share, flags, masking in cases:
    try:
        T.test_native_kd_step_equals_the_per_launch_engine(share, flags, masking)
        print("share %s flags %s masking %s: OK" % (share, flags, masking))
    except AssertionError as e:
        msg = str(e).split("\n")[0]
        print("share %s flags %s masking %s: FAIL %s" % (share, flags, masking, msg[:400]))
