"""Idle lane-steps of msm_accumulate for uniform digits: slices handed to the lanes in bucket order (today) against slices
ordered by length.  A wave of 64 lanes takes as many steps as its longest slice.
Measured counterpart (profiles/r02e): slices of 64 cost the accumulation +5.1 % in bucket order; the model says 6.1 %."""
import numpy as np

NB, W = 1 << 15, 16


def idle(ksl: int, ordered: bool, log_m: int = 20):
    cnt = np.random.default_rng(1).multinomial(W * ((1 << log_m) + 6), np.full(NB, 1 / NB))
    full, rem = cnt // ksl, cnt % ksl
    if ordered:
        lens = np.concatenate([np.full(full.sum(), ksl), np.sort(rem[rem > 0])[::-1]])
    else:
        reps = full + (rem > 0)
        pos = np.arange(reps.sum()) - np.repeat(np.cumsum(reps) - reps, reps)
        lens = np.where(pos < np.repeat(full, reps), ksl, np.repeat(rem, reps))
    pad = (-len(lens)) % 64
    waves = np.concatenate([lens, np.zeros(pad, dtype=lens.dtype)]).reshape(-1, 64).max(axis=1)
    return len(lens), (waves * 64).sum() / lens.sum() - 1


if __name__ == "__main__":
    for log_m, ksls in ((20, (32, 64, 128)), (18, (8, 16, 32, 64)), (17, (8, 16, 32, 64)), (16, (4, 8, 16, 32, 64))):
        for ksl in ksls:
            (n, a), (_, b) = idle(ksl, False, log_m), idle(ksl, True, log_m)
            print(f"m = 2^{log_m} ksl = {ksl:3d}: {n / NB:6.2f} slices per bucket, idle lane-steps {100 * a:5.1f} % in bucket order, {100 * b:4.1f} % in length order")
