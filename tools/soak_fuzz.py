"""Soak: the seeded fuzz tests of tests/ on seeds they do not use (a failure prints the seed; add it to the test's parametrize list).
    python tools/soak_fuzz.py [first_seed] [count]"""
import os
import sys
import traceback

root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
import test_fusion_fuzz_gpu as ff  # noqa: E402
import test_gemm_fuzz_gpu as gf  # noqa: E402
import test_gemm_group_fuzz_gpu as gg  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = 0
for s in range(first, first + count):
    for name, fn in (("gemm small", lambda: gf.test_gemm_random_cases(s, False)), ("gemm big", lambda: gf.test_gemm_random_cases(s, True) if s % 6 == 0 else None),
                     ("gemm group", lambda: gg._run(s)), ("fusion", lambda: ff._run(s))):
        try:
            fn()
        except Exception:
            bad += 1
            print(f"FAILED {name} seed {s}")
            traceback.print_exc(limit=3)
print(f"soak: seeds {first}..{first + count - 1}, {bad} failures")
sys.exit(1 if bad else 0)
