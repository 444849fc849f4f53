"""Synthetic planted-partition inputs: Q equal groups with contiguous labels
(matching the `-n` semantics of main.cpp:240-252), cin = cQ/((Q-1)eps+1), cout = eps*cin
(blockmodel.cpp:256-257); per block pair m ~ Poisson(p * pairs) endpoint pairs drawn uniformly,
self-loops dropped, duplicates merged."""
import numpy as np


def group_sizes(N, Q):
    sizes = [N // Q] * Q
    sizes[-1] += N - sum(sizes)
    return sizes


def cin_cout(Q, c, eps):
    cin = c * Q / ((Q - 1) * eps + 1)
    return cin, eps * cin


def planted_partition(N, Q, c, eps, seed):
    """returns (pairs uint32 [m,2] with a<b unique, cin, cout)"""
    rng = np.random.default_rng(seed)
    cin, cout = cin_cout(Q, c, eps)
    sizes = group_sizes(N, Q)
    starts = np.cumsum([0] + sizes)
    chunks = []
    for r in range(Q):
        for s in range(r, Q):
            p = (cin if r == s else cout) / N
            npairs = sizes[r] * (sizes[r] - 1) / 2 if r == s else sizes[r] * sizes[s]
            m = rng.poisson(p * npairs)
            a = rng.integers(starts[r], starts[r + 1], m, dtype=np.int64)
            b = rng.integers(starts[s], starts[s + 1], m, dtype=np.int64)
            lo, hi = np.minimum(a, b), np.maximum(a, b)
            keep = lo != hi
            chunks.append(lo[keep] * N + hi[keep])
    keys = np.unique(np.concatenate(chunks))
    pairs = np.empty((len(keys), 2), dtype=np.uint32)
    pairs[:, 0] = keys // N
    pairs[:, 1] = keys % N
    return pairs, cin, cout


def true_conf(N, Q):
    return np.repeat(np.arange(Q, dtype=np.uint32), group_sizes(N, Q))


def cab_matrix(Q, cin, cout):
    cab = np.full((Q, Q), cout)
    np.fill_diagonal(cab, cin)
    return cab


def dc_sbm_powerlaw(N, Q, mean_degree, eps, seed, tail=1.5, cap_factor=300.0):
    """Degree-corrected planted partition: propensities theta_i with
    P(theta > x) = x^-tail (density exponent tail+1), capped at cap_factor * mean and normalised to
    mean 1; Chung-Lu edges: the pair count of block pair (r, s) is Poisson with mean
    omega_rs * |r||s| / N (omega = cin on the diagonal, cout off it; half of it on the diagonal) and
    endpoints are drawn proportionally to theta inside each block. Returns (pairs, cab for
    --deg_corr_flag 1 with the reference's raw degrees, i.e. omega / mean_degree^2, mean theta check)."""
    rng = np.random.default_rng(seed)
    cin, cout = cin_cout(Q, mean_degree, eps)
    sizes = group_sizes(N, Q)
    starts = np.cumsum([0] + sizes)
    theta = (1.0 - rng.random(N)) ** (-1.0 / tail)
    theta = np.minimum(theta, cap_factor * theta.mean())
    theta /= theta.mean()
    cum = [np.cumsum(theta[starts[r]:starts[r + 1]]) for r in range(Q)]
    chunks = []
    for r in range(Q):
        for s in range(r, Q):
            w = (cin if r == s else cout) / N
            mass = cum[r][-1] * cum[s][-1] * (0.5 if r == s else 1.0)
            m = rng.poisson(w * mass)
            a = starts[r] + np.searchsorted(cum[r], rng.random(m) * cum[r][-1])
            b = starts[s] + np.searchsorted(cum[s], rng.random(m) * cum[s][-1])
            lo, hi = np.minimum(a, b), np.maximum(a, b)
            keep = lo != hi
            chunks.append(lo[keep].astype(np.int64) * N + hi[keep])
    keys = np.unique(np.concatenate(chunks))
    pairs = np.empty((len(keys), 2), dtype=np.uint32)
    pairs[:, 0] = keys // N
    pairs[:, 1] = keys % N
    c_eff = 2.0 * len(pairs) / N
    return pairs, cab_matrix(Q, cin, cout) / (c_eff * c_eff), c_eff
