"""Idle lane-steps of msm_accumulate for uniform digits: slices handed to the lanes in bucket order (today) against slices
ordered by length.  A wave of 64 lanes takes as many steps as its longest slice.
Measured counterpart (profiles/r02e): slices of 64 cost the accumulation +5.1 % in bucket order; the model says 6.1 %."""
import numpy as np

NB, M, W = 1 << 15, (1 << 20) + 6, 16
cnt = np.random.default_rng(1).multinomial(W * M, np.full(NB, 1 / NB))


def idle(ksl: int, ordered: bool):
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
    for ksl in (32, 64, 128):
        for ordered in (False, True):
            n, w = idle(ksl, ordered)
            print(f"ksl={ksl:3d} {'length order' if ordered else 'bucket order'}: {n:7d} slices, idle lane-steps {100 * w:5.2f} %")
