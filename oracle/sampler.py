"""Oracle: categorical action samplers (TEST INFRASTRUCTURE, see __init__).

1. `sample_numpy_reference` IS the reference: paac.py:34-45 verbatim semantics
   (probs - float32.epsneg, one np.random.multinomial(1, p) per env in index
   order on the caller's legacy MT19937 stream).  numpy is a third-party
   dependency of the reference that is installed here, so numpy itself is the
   sampler oracle.
2. `sample_mt_restated` restates what numpy's legacy multinomial does for n=1
   in terms of raw 53-bit MT19937 doubles; it is what the
   HIP kernel implements and is checked against (1) in
   tests/test_oracle_golden.py (both modes replay the reference capture) and
   tests/test_hip_misc.py::test_sampler_mt_matches_numpy.
3. `philox4x32` / `sample_philox` restate the build's own counter-based
   throughput sampler (no reference counterpart: parity is to this spec only).
"""
import numpy as np

EPSNEG32 = np.finfo(np.float32).epsneg   # 5.9604645e-08, paac.py:42


def sample_numpy_reference(probs_f32, rs):
    """paac.py:34-45 on RandomState `rs` (the reference uses the global np.random)."""
    probs = probs_f32 - EPSNEG32
    return [int(np.nonzero(rs.multinomial(1, p))[0][0]) for p in probs]


def sample_mt_restated(probs_f32, rs):
    """Appendix D.  Consumes doubles from `rs.random_sample()` (same 53-bit
    doubles numpy's binomial inversion draws).  Returns (actions, draws_used)."""
    p32 = (np.asarray(probs_f32, dtype=np.float32) - np.float32(EPSNEG32)).astype(np.float32)
    p = p32.astype(np.float64)
    N, A = p.shape
    actions = []
    used = 0
    for e in range(N):
        remaining = 1.0
        act = A - 1
        for j in range(A - 1):
            pj = p[e, j] / remaining
            if pj != 0.0:                      # random_binomial returns 0 without drawing when p == 0
                u = rs.random_sample()
                used += 1
                if pj <= 0.5:
                    hit = u > 1.0 - pj
                else:
                    q = 1.0 - pj
                    hit = not (u > 1.0 - q)
                if hit:
                    act = j
                    break
            remaining -= p[e, j]
        actions.append(act)
    return actions, used


# ---------------------------------------------------------------------------
# Philox4x32-10 (Salmon et al. 2011), the build's own throughput RNG.
PHILOX_M0 = np.uint64(0xD2511F53)
PHILOX_M1 = np.uint64(0xCD9E8D57)
PHILOX_W0 = np.uint32(0x9E3779B9)
PHILOX_W1 = np.uint32(0xBB67AE85)


def philox4x32(counter, key, rounds=10):
    """counter: [...,4] uint32, key: [...,2] uint32 -> [...,4] uint32."""
    c = np.array(counter, dtype=np.uint32, copy=True)
    k = np.array(key, dtype=np.uint32, copy=True)
    c0, c1, c2, c3 = [c[..., i].astype(np.uint64) for i in range(4)]
    k0 = k[..., 0].astype(np.uint64)
    k1 = k[..., 1].astype(np.uint64)
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(rounds):
        p0 = PHILOX_M0 * c0
        p1 = PHILOX_M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & mask
        hi1, lo1 = p1 >> np.uint64(32), p1 & mask
        c0, c1, c2, c3 = (hi1 ^ c1 ^ k0) & mask, lo1, (hi0 ^ c3 ^ k1) & mask, lo0
        k0 = (k0 + np.uint64(PHILOX_W0)) & mask
        k1 = (k1 + np.uint64(PHILOX_W1)) & mask
    return np.stack([c0, c1, c2, c3], axis=-1).astype(np.uint32)


def philox_uniform(seed, step, env_ids, stream=0):
    """u in [0,1): 24 high bits of word 0 of philox(counter=(env, step_lo, step_hi, stream), key=(seed_lo, seed_hi))."""
    env_ids = np.asarray(env_ids, dtype=np.uint32)
    ctr = np.zeros(env_ids.shape + (4,), dtype=np.uint32)
    ctr[..., 0] = env_ids
    ctr[..., 1] = np.uint32(step & 0xFFFFFFFF)
    ctr[..., 2] = np.uint32((step >> 32) & 0xFFFFFFFF)
    ctr[..., 3] = np.uint32(stream)
    key = np.zeros(env_ids.shape + (2,), dtype=np.uint32)
    key[..., 0] = np.uint32(seed & 0xFFFFFFFF)
    key[..., 1] = np.uint32((seed >> 32) & 0xFFFFFFFF)
    r = philox4x32(ctr, key)
    return (r[..., 0] >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)


def sample_philox(probs_f32, seed, step, env_offset=0):
    """Inverse-CDF on float32 running sums: action = first j with u < sum_{i<=j} p_i, else A-1."""
    p = np.asarray(probs_f32, dtype=np.float32)
    N, A = p.shape
    u = philox_uniform(seed, step, np.arange(N, dtype=np.uint32) + np.uint32(env_offset))
    acts = np.full(N, A - 1, dtype=np.int32)
    done = np.zeros(N, dtype=bool)
    c = np.zeros(N, dtype=np.float32)
    for j in range(A - 1):
        c = (c + p[:, j]).astype(np.float32)
        hit = (~done) & (u < c)
        acts[hit] = j
        done |= hit
    return acts
