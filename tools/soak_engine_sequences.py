"""Soak: random engine configurations x random call sequences against the oracle (tests/test_engine_sequences_gpu.py::_run) on seeds the suite
does not use; a failure prints the seed and the sequence (add the seed to the test's parametrize list).
    python tools/soak_engine_sequences.py [first_seed] [count] [ops_per_sequence]"""
import os
import sys
import traceback

root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
import test_engine_sequences_gpu as es  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
n_ops = int(sys.argv[3]) if len(sys.argv) > 3 else 8
bad = 0
for s in range(first, first + count):
    try:
        es._run(s, n_ops)
    except Exception as e:
        bad += 1
        print(f"FAILED seed {s}: {type(e).__name__}: {str(e)[:1200]}")
        traceback.print_exc(limit=2)
print(f"soak: seeds {first}..{first + count - 1}, {bad} failures")
sys.exit(1 if bad else 0)
