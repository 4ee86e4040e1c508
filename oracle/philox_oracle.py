"""TEST INFRASTRUCTURE: numpy restatement of the in-kernel noise of mcmc_update_philox
(step_amd/csrc/mcmc_update.hip).  No reference counterpart: the reference draws torch.randn_like
(sampler/daps.py:192); this generator exists so that a chain's noise does not depend on how chains
are batched or sharded.  Philox4x32-10 as published (Salmon, Moraes, Dror, Shaw: "Parallel random
numbers: as easy as 1, 2, 3", SC'11; the known-answer vectors of the Random123 distribution are
checked in tests/test_philox_oracle.py), then Box-Muller on 24-bit uniforms."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)


def philox4x32_10(ctr, key):
    """ctr: uint32 [..., 4], key: uint32 [..., 2] (broadcastable) -> uint32 [..., 4]."""
    c = [np.asarray(ctr[..., i], dtype=np.uint32) for i in range(4)]
    k0 = np.asarray(key[..., 0], dtype=np.uint32)
    k1 = np.asarray(key[..., 1], dtype=np.uint32)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = M0 * c[0].astype(np.uint64)
            p1 = M1 * c[2].astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), p0.astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), p1.astype(np.uint32)
            c = [hi1 ^ c[1] ^ k0, lo1, hi0 ^ c[3] ^ k1, lo0]
            k0 = (k0 + W0).astype(np.uint32)
            k1 = (k1 + W1).astype(np.uint32)
    return np.stack(np.broadcast_arrays(*c), axis=-1)


def box_muller(a, b):
    u = ((a >> np.uint32(8)).astype(np.float32) + np.float32(1.0)) * np.float32(1.0 / 16777216.0)
    r = np.sqrt(np.float32(-2.0) * np.log(u.astype(np.float64))).astype(np.float32)
    ang = (b >> np.uint32(8)).astype(np.float64) * (2.0 * np.pi / 16777216.0)
    return r * np.cos(ang).astype(np.float32), r * np.sin(ang).astype(np.float32)


def chain_noise(seed, step, chain_len):
    """The chain_len normals that mcmc_update_philox draws for a chain with `seed` at `step`."""
    assert chain_len % 4 == 0
    j = np.arange(chain_len // 4, dtype=np.uint64)
    ctr = np.stack([(j & np.uint64(0xFFFFFFFF)).astype(np.uint32), (j >> np.uint64(32)).astype(np.uint32),
                    np.full(j.shape, step & 0xFFFFFFFF, np.uint32), np.full(j.shape, step >> 32, np.uint32)], -1)
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    key = np.array([seed & 0xFFFFFFFF, seed >> 32], dtype=np.uint32)
    r = philox4x32_10(ctr, key[None, :])
    n0, n1 = box_muller(r[:, 0], r[:, 1])
    n2, n3 = box_muller(r[:, 2], r[:, 3])
    return np.stack([n0, n1, n2, n3], -1).reshape(-1)


def chain_noise_quad(seed, step, chain_len, width):
    """The same draw in the column-quad layout (mcmc_update_philox with quad_w = width, and the
    black-hole adjoint's in-kernel Langevin step): the chain is rows of `width` elements and
    generator call (row // 4) * width + column supplies rows 4 (row // 4) .. + 3 of that column."""
    assert chain_len % (4 * width) == 0
    calls = chain_noise(seed, step, chain_len).reshape(chain_len // (4 * width), width, 4)
    return np.ascontiguousarray(calls.transpose(0, 2, 1)).reshape(-1)
