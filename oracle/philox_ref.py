"""Oracle: counter-based normal generator (test infrastructure, see oracle/__init__.py).

The reference draws the momentum with `torch.randn_like` on the device generator
(main_sampling.py:692) and the accept uniform with the CPU generator (:720); both
streams depend on batch layout, so they cannot be shard-invariant.  The build's
own generator is Philox4x32-10 (Salmon et al., SC'11) keyed by the run seed and
counted by (element-quad, chain id, draw number, stream tag); this file is its
numpy restatement, used to pin `nhmc_randn_philox` / `nhmc_uniform_philox`.

Layout (must match csrc/rng.hip):
    key     = (seed & 0xffffffff, seed >> 32)
    counter = (quad, chain_id, draw, tag)      tag 0 = momentum normals, 1 = accept uniform
    normals for elements 4*quad .. 4*quad+3 of the chain come from the four output
    words r0..r3:  u = ((r >> 8) + 0.5) * 2^-24,
    z0 = sqrt(-2 ln u(r0)) cos(2 pi u(r1)), z1 = same radius * sin(...),
    z2, z3 likewise from (r2, r3).
    uniform for the accept test of (chain, draw): ((r0 >> 8) + 0.5) * 2^-24 at quad 0, tag 1.
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised over numpy uint32 arrays; returns four uint32 arrays."""
    c0, c1, c2, c3 = (np.asarray(v, dtype=np.uint32) for v in (c0, c1, c2, c3))
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0, k1 = np.uint32(k0), np.uint32(k1)
    for _ in range(10):
        p0 = c0.astype(np.uint64) * M0
        p1 = c2.astype(np.uint64) * M1
        hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & MASK).astype(np.uint32)
        hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & MASK).astype(np.uint32)
        c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
        k0 = np.uint32((int(k0) + int(W0)) & 0xFFFFFFFF)
        k1 = np.uint32((int(k1) + int(W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def _unit(r):
    return ((r >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -24)


def randn_chain(seed, chain_id, draw, n):
    """n standard normals (fp32) for one chain and draw; n must be a multiple of 4."""
    assert n % 4 == 0
    quad = np.arange(n // 4, dtype=np.uint32)
    r0, r1, r2, r3 = philox4x32_10(quad, np.uint32(chain_id), np.uint32(draw), np.uint32(0),
                                   seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    out = np.empty((n // 4, 4), dtype=np.float32)
    two_pi = np.float32(6.283185307179586)
    for col, (ra, rb) in ((0, (r0, r1)), (2, (r2, r3))):
        rad = np.sqrt(np.float32(-2.0) * np.log(_unit(ra)))
        ang = two_pi * _unit(rb)
        out[:, col] = rad * np.cos(ang)
        out[:, col + 1] = rad * np.sin(ang)
    return out.reshape(-1)


def uniform_chain(seed, chain_id, draw):
    r0, _, _, _ = philox4x32_10(np.uint32(0), np.uint32(chain_id), np.uint32(draw), np.uint32(1),
                                seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    return float(_unit(np.asarray(r0).reshape(1))[0])
