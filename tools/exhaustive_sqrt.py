"""Exhaustive check of the device sqrt (device_math.hpp `sqrt_rn`) against the host's IEEE
sqrtf: every non-negative float bit pattern (0 .. +inf) plus a sample of negatives / NaNs.
Run once on a GPU box after touching sqrt_rn:  python tools/exhaustive_sqrt.py"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from pedoni_amd import abi  # noqa: E402

CHUNK = 1 << 26
t0 = time.time()
bad = 0
for lo in range(0, 0x7F800001, CHUNK):
    hi = min(lo + CHUNK, 0x7F800001)
    x = np.arange(lo, hi, dtype=np.uint32).view(np.float32)
    got = abi.selftest_math(1, x)
    with np.errstate(all="ignore"):
        want = np.sqrt(x)
    neq = got.view(np.uint32) != want.view(np.uint32)
    bad += int(neq.sum())
    if neq.any():
        i = np.flatnonzero(neq)[:4]
        print("mismatch", x[i], got[i], want[i], flush=True)
    print(f"{hi / 0x7F800001:6.1%} checked, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
x = np.concatenate([np.random.default_rng(1).integers(0x80000000, 2**32, 1 << 20, dtype=np.uint64)
                    .astype(np.uint32), [0x80000000, 0x7FC00000, 0x7F800001]]).astype(np.uint32).view(np.float32)
got = abi.selftest_math(1, x)
with np.errstate(all="ignore"):
    want = np.sqrt(x)
ok = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
print("negatives / NaNs:", int((~ok).sum()), "mismatches")
print("TOTAL mismatches:", bad + int((~ok).sum()))
