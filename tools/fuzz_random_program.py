"""Random programs (examples/random_program.cpp) with random shapes and seeds for a while: hehub itself on the CPU
(oracle/_ref/ref_randprog_cpu, prebuilt where the reference tree is) against the MI355X layer with 1 / 8 lanes, eager / deferred.
    gpurun -- 'python tools/fuzz_random_program.py [seconds=300] [first_seed=1000]'"""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from make_random_program import REF, digest  # noqa: E402
from hehub_amd.build import build_example  # noqa: E402

MODES = {"1 lane": {"HEHUB_AMD_LANES": "1", "HEHUB_AMD_DEFER": "0"}, "8 lanes": {"HEHUB_AMD_LANES": "8", "HEHUB_AMD_DEFER": "0"},
         "deferred": {"HEHUB_AMD_DEFER": "1"}, "deferred, 8 lanes": {"HEHUB_AMD_DEFER": "1", "HEHUB_AMD_LANES": "8"},
         "3 lanes": {"HEHUB_AMD_LANES": "3", "HEHUB_AMD_DEFER": "0"}, "default": {},
         # device ranks (round 6; the ranks share the GPU of a one-GPU box): calls placed by operand residency / round robin, groups per rank
         "2 ranks": {"HEHUB_AMD_DEVICES": "0,0", "HEHUB_AMD_DEFER": "0"}, "3 ranks, recorded": {"HEHUB_AMD_DEVICES": "0,0,0"},
         "8 ranks, 2 lanes": {"HEHUB_AMD_DEVICES": "0,0,0,0,0,0,0,0", "HEHUB_AMD_LANES": "2", "HEHUB_AMD_DEFER": "0"},
         "5 ranks, recorded, 1 lane": {"HEHUB_AMD_DEVICES": "0,0,0,0,0", "HEHUB_AMD_LANES": "1"},
         "2 ranks, recorded, 8 lanes": {"HEHUB_AMD_DEVICES": "0,0", "HEHUB_AMD_LANES": "8"},
         "8 ranks, recorded": {"HEHUB_AMD_DEVICES": "0,0,0,0,0,0,0,0"}}
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
assert os.path.exists(REF), "oracle/_ref/ref_randprog_cpu is built where /root/reference is (make -C oracle ref_randprog)"
binary = build_example("random_program")
rng = random.Random(seed)
t0, cases, bad = time.time(), 0, 0
while time.time() - t0 < seconds:
    logn = rng.choice([8, 10, 11, 12, 12, 13, 14])
    L = rng.randint(2, 7 if logn <= 12 else 5)
    case = (logn, L, rng.randint(2, 24), rng.randint(50, 1500 if logn <= 12 else 300), seed + cases, rng.choice([0, 0, 1]))
    want = digest(REF, case)[0]
    for mode, env in MODES.items():
        for level in ("B",):
            got = digest(binary, case, env)[0]
            if got != want:
                bad += 1
                print("MISMATCH", case, mode, got, want, flush=True)
    a = {mode: digest(binary, case, dict(env, HP_PARITY_LEVEL="A"))[0] for mode, env in list(MODES.items())[:3] + list(MODES.items())[6:8] + list(MODES.items())[10:]}
    if len(set(a.values())) != 1:
        bad += 1
        print("MISMATCH at level A", case, a, flush=True)
    cases += 1
print(f"{cases} random programs x {len(MODES)} modes at level B against hehub on the CPU, x 7 modes at level A against each other: {bad} mismatches "
      f"in {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
