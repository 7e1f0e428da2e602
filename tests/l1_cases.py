"""Seeded random L1 (band-kernel) cases shared by oracle, reference-.so and GPU parity tests."""
import numpy as np

BASES = np.frombuffer(b"ACGT", dtype=np.uint8)


def random_case(rng: np.random.Generator, band: int, T: int, with_n=True, plant_repeat=True, q_max=64):
    """One window/target pair with truth_len == T + 2B - 1, random penalties, an indel/SNV-perturbed read."""
    L = T + 2 * band - 1
    truth = BASES[rng.integers(0, 4, L)].copy()
    if plant_repeat and L > 40:
        a = int(rng.integers(5, L - 30))
        n = int(rng.integers(6, 16))
        unit = BASES[rng.integers(0, 4, int(rng.integers(1, 3)))]
        rep = np.tile(unit, n)[:n]
        truth[a:a + n] = rep
    # read: copy of a slice of the window near the centre diagonal, then perturb
    start = int(rng.integers(max(0, band - 6), band + 6))
    src = truth[start:start + T + 20]
    read = list(src[:T + 10])
    n_edits = int(rng.integers(0, 5))
    for _ in range(n_edits):
        kind = rng.random()
        p = int(rng.integers(0, max(1, len(read) - 1)))
        if kind < 0.6:
            read[p] = int(BASES[rng.integers(0, 4)])
        elif kind < 0.8:
            del read[p:p + int(rng.integers(1, 6))]
        else:
            ins = BASES[rng.integers(0, 4, int(rng.integers(1, 6)))]
            read[p:p] = [int(b) for b in ins]
    while len(read) < T:
        read.append(int(BASES[rng.integers(0, 4)]))
    read = np.array(read[:T], dtype=np.uint8)
    if with_n:
        if rng.random() < 0.3:
            truth[int(rng.integers(0, L))] = ord("N")
        if rng.random() < 0.2:
            read[int(rng.integers(0, T))] = ord("N")
    quals = rng.choice(np.array([2, 12, 25, 37, q_max], dtype=np.uint8), size=T, p=[0.05, 0.1, 0.2, 0.6, 0.05])
    gap_open = rng.integers(20, 60, L).astype(np.int8)
    gap_open[rng.random(L) < 0.1] = rng.integers(2, 15)
    gap_extend = rng.integers(1, 6, L).astype(np.int8)
    # SNV mask = haplotype rotated by one (the reference's default masks), priors mostly high with some caps
    mask = np.roll(truth, 1 if rng.random() < 0.5 else -1)
    prior = np.full(L, 125, dtype=np.int8)
    low = rng.random(L) < 0.15
    prior[low] = rng.integers(5, 40, int(low.sum()))
    return dict(band=band, truth=truth.tobytes(), target=read.tobytes(), quals=quals, gap_open=gap_open,
                gap_extend=gap_extend, mask=mask.tobytes(), prior=prior)
