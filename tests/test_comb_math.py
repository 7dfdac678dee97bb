"""The identity behind the comb MSM of csrc/msm_comb.hip, restated with the oracle's plain-Python group arithmetic (no GPU):

    sum_i s_i P_i  =  sum_t 2^t * [ sum_clumps +-T[clump][idx(clump, t)] ]  +  (2^T - 1) * sum_i P_i/2
    T[clump][idx]  =  +P'_top + sum_{j < top} (bit_j(idx) ? +1 : -1) P'_j,      P' = P/2 = ((r+1)/2) P

with the pattern of bit plane t read as signed bits (sigma = 2 b - 1), complemented -- and the entry negated -- when the top base's
sign is minus.  Covers a short last clump, scalars 0 / 1 / r-1, and the plane count T < 254 of the narrow wire-comb classes."""
import random

from oracle import pairing as pr
from oracle.gnark_format import R

G = (1, 2)


def comb_msm(points, scalars, k, T):
    half = (R + 1) // 2
    ph = [pr.g1_mul(p, half) for p in points]
    n = len(points)
    planes = []
    for t in range(T):
        acc = (0, 0)
        for c0 in range(0, n, k):
            kc = min(k, n - c0)
            pat = sum(((scalars[c0 + j] >> t) & 1) << j for j in range(kc))
            top = (pat >> (kc - 1)) & 1
            idx = (pat if top else ~pat) & ((1 << (kc - 1)) - 1)
            entry = ph[c0 + kc - 1]                                   # T[clump][idx]
            for j in range(kc - 1):
                entry = pr.g1_add(entry, ph[c0 + j] if (idx >> j) & 1 else pr.g1_neg(ph[c0 + j]))
            acc = pr.g1_add(acc, entry if top else pr.g1_neg(entry))
        planes.append(acc)
    r = (0, 0)
    for t in range(T - 1, -1, -1):                                     # Horner over the planes
        r = pr.g1_add(pr.g1_add(r, r), planes[t])
    corr = (0, 0)
    for p in ph:
        corr = pr.g1_add(corr, p)
    return pr.g1_add(r, pr.g1_mul(corr, (1 << T) - 1))


def naive(points, scalars):
    r = (0, 0)
    for p, s in zip(points, scalars):
        r = pr.g1_add(r, pr.g1_mul(p, s))
    return r


def test_comb_identity_full_width():
    rng = random.Random(11)
    pts = [pr.g1_mul(G, rng.randrange(1, R)) for _ in range(11)]
    for k in (2, 4, 5):
        sc = [rng.randrange(R) for _ in pts]
        sc[0], sc[1], sc[2] = 0, 1, R - 1
        assert comb_msm(pts, sc, k, 254) == naive(pts, sc)
    assert comb_msm(pts, [0] * len(pts), 4, 254) == (0, 0)           # all planes all-minus: the constant cancels them


def test_comb_identity_narrow_classes():
    rng = random.Random(12)
    pts = [pr.g1_mul(G, rng.randrange(1, R)) for _ in range(7)]
    for T in (8, 32):
        sc = [rng.randrange(1 << T) for _ in pts]
        assert comb_msm(pts, sc, 3, T) == naive(pts, sc)
