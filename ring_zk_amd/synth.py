"""Seeded synthetic inputs with the reference's distributions.

  key entries U, message x, scalars g : uniform over [-(q-1)/2, (q-1)/2]  (commit.rs:41,53; params.rs:126;
                                         benches/bench.rs:354-358 — full length N, the worst case)
  r                                   : uniform in [-b, b]                 (commit.rs:101)
  y                                   : trunc(Normal(0, sigma))            (open.rs:88-94, polynomial.rs:28-44)
  d                                   : exactly kappa coefficients +-1     (challenge_space.rs:12-33)

numpy versions feed the parity tests; torch versions build the bench inputs directly in HBM.
"""
from __future__ import annotations

import numpy as np

Q_DEFAULT = 3515337053


def isqrt(x: int) -> int:
    import math

    return math.isqrt(x)


def sigma(b: int, kappa: int, k: int, N: int) -> int:
    return b * (11 * kappa) * isqrt(k * N)  # params.rs:94-98


def key(rng: np.random.Generator, N, n, k, l, q=Q_DEFAULT) -> np.ndarray:
    """[a1;a2] with a1 = [I_n | a1'], a2 = [0 | I_l | a2'] (commit.rs:33-60)."""
    half = (q - 1) // 2
    A = np.zeros((n + l, k, N), dtype=np.int64)
    for i in range(n):
        A[i, i, 0] = 1
        A[i, n:, :] = rng.integers(-half, half + 1, (k - n, N), dtype=np.int64)
    for i in range(l):
        A[n + i, n + i, 0] = 1
        if k - n - l > 0:
            A[n + i, n + l:, :] = rng.integers(-half, half + 1, (k - n - l, N), dtype=np.int64)
    return A


def uniform(rng, shape, q=Q_DEFAULT) -> np.ndarray:
    half = (q - 1) // 2
    return rng.integers(-half, half + 1, shape, dtype=np.int64)


def small(rng, shape, b=1) -> np.ndarray:
    return rng.integers(-b, b + 1, shape, dtype=np.int64)


def gauss(rng, shape, sig) -> np.ndarray:
    return np.trunc(rng.normal(0.0, float(sig), shape)).astype(np.int64)


def challenge(rng, batch_shape, N, kappa) -> np.ndarray:
    kap = min(kappa, N)
    B = int(np.prod(batch_shape, dtype=np.int64)) if len(batch_shape) else 1
    d = np.zeros((B, N), dtype=np.int64)
    for i in range(B):
        pos = rng.choice(N, kap, replace=False)
        d[i, pos] = rng.choice(np.array([-1, 1], dtype=np.int64), kap)
    return d.reshape(tuple(batch_shape) + (N,))


# ---- device-side generators (torch: plumbing for synthetic bench data only) -----------------------------
def t_uniform(gen, shape, device, q=Q_DEFAULT):
    import torch

    half = (q - 1) // 2
    return torch.randint(-half, half + 1, shape, dtype=torch.int64, device=device, generator=gen)


def t_small(gen, shape, device, b=1):
    import torch

    return torch.randint(-b, b + 1, shape, dtype=torch.int64, device=device, generator=gen)


def t_gauss(gen, shape, device, sig):
    import torch

    return torch.trunc(torch.randn(shape, dtype=torch.float64, device=device, generator=gen) * float(sig)).to(torch.int64)


def t_challenge(gen, B, N, kappa, device):
    import torch

    kap = min(kappa, N)
    keys = torch.rand((B, N), device=device, generator=gen)
    pos = keys.argsort(dim=1)[:, :kap]
    signs = torch.randint(0, 2, (B, kap), device=device, generator=gen, dtype=torch.int64) * 2 - 1
    d = torch.zeros((B, N), dtype=torch.int64, device=device)
    d.scatter_(1, pos, signs)
    return d


def t_key(gen, N, n, k, l, device, q=Q_DEFAULT):
    import torch

    half = (q - 1) // 2
    A = torch.zeros((n + l, k, N), dtype=torch.int64, device=device)
    for i in range(n):
        A[i, i, 0] = 1
        A[i, n:, :] = torch.randint(-half, half + 1, (k - n, N), dtype=torch.int64, device=device, generator=gen)
    for i in range(l):
        A[n + i, n + i, 0] = 1
        if k - n - l > 0:
            A[n + i, n + l:, :] = torch.randint(-half, half + 1, (k - n - l, N), dtype=torch.int64, device=device,
                                                generator=gen)
    return A
