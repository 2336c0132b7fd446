"""Synthetic multi-domain CTR data: platform-independent numpy PCG64 streams.

ids: uniform over [0, V) (worst case for the gather: pure HBM) or Zipf(alpha) truncated to V (hot rows);
the domain column draws domains with geometric-ish weights; labels come from a planted teacher so AUC means
something: y ~ Bernoulli(sigmoid(sum_f u_f[x_f] + c_domain - 2.5)), u ~ N(0, teacher_std^2) (0.3: SURVEY 8d).  dtypes follow run.py:198-199 (int32 ids,
int16 labels)."""
import numpy as np


def make_dataset(n_rows, field_dims, n_domain, domain_idx, seed=2000, dist="uniform", alpha=1.05, teacher_seed=2001, teacher_std=0.3):
    rng = np.random.Generator(np.random.PCG64(seed))
    F = len(field_dims)
    X = np.empty((n_rows, F), dtype=np.int32)
    for f, V in enumerate(field_dims):
        if f == domain_idx:
            w = 0.8 ** np.arange(n_domain)
            X[:, f] = rng.choice(n_domain, size=n_rows, p=w / w.sum()).astype(np.int32)
        elif dist == "zipf":
            ranks = np.arange(1, V + 1, dtype=np.float64)
            p = ranks ** (-alpha)
            cdf = np.cumsum(p / p.sum())
            X[:, f] = np.minimum(np.searchsorted(cdf, rng.random(n_rows)), V - 1).astype(np.int32)
        else:
            X[:, f] = rng.integers(0, V, size=n_rows, dtype=np.int64).astype(np.int32)
    trng = np.random.Generator(np.random.PCG64(teacher_seed))
    logit = np.full(n_rows, -2.5)
    for f, V in enumerate(field_dims):
        if V <= 4_000_000:
            u = trng.normal(0.0, teacher_std, size=V)
            logit += u[X[:, f]]
        else:                               # very large vocabularies: hash the id into a 1M-entry teacher table
            u = trng.normal(0.0, teacher_std, size=1_000_003)
            logit += u[(X[:, f].astype(np.int64) * 2654435761) % 1_000_003]
    c = trng.normal(0.0, 0.5, size=n_domain)
    logit += c[X[:, domain_idx]]
    y = (rng.random(n_rows) < 1.0 / (1.0 + np.exp(-logit))).astype(np.int16)
    return X, y
