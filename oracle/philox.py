"""TEST INFRASTRUCTURE — not part of the product path.

Counter-based uniforms (Philox4x32-10, Salmon et al. SC'11, published algorithm)
restated in numpy.  The HIP kernels evaluate the same function in-kernel
(drakes_amd/csrc/drk_rng.h); tests compare the two bit for bit.

The reference draws `torch.rand_like(categorical_probs)` once per reverse step
(/root/reference/drakes_dna/diffusion_gosai_update.py:24-28 and :30-35).  This
function replaces that draw with a pure function of
(seed, step, global sample index, position, class) so that every
implementation (reference under injection, oracle, HIP) sees the same noise.

Counter layout (the build's own convention):
    c0 = position within the sequence
    c1 = global sample index (low 32 bits)
    c2 = reverse-step index
    c3 = 0x44524B53 ("DRKS") ^ (global sample index >> 32)
    key = (seed & 0xffffffff, seed >> 32)
One Philox block (4 x 32 bit) feeds the 5 classes of one position:
    U[v] = (r[v] >> 8) * 2^-24                       v = 0..3
    U[4] = (((r0 & 255) << 16) | ((r1 & 255) << 8) | (r2 & 255)) * 2^-24
so U is in [0, 1) on the 2^-24 grid, like torch.rand for float32.
"""
import numpy as np

_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = 0x9E3779B9
_W1 = 0xBB67AE85
_MASK32 = np.uint64(0xFFFFFFFF)
DOMAIN_TAG = 0x44524B53


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """All arguments broadcastable uint32 arrays; returns 4 uint32 arrays."""
    c0 = np.asarray(c0, dtype=np.uint64)
    c1 = np.asarray(c1, dtype=np.uint64)
    c2 = np.asarray(c2, dtype=np.uint64)
    c3 = np.asarray(c3, dtype=np.uint64)
    k0 = int(k0) & 0xFFFFFFFF
    k1 = int(k1) & 0xFFFFFFFF
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    for _ in range(10):
        p0 = _M0 * c0
        p1 = _M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & _MASK32
        hi1, lo1 = p1 >> np.uint64(32), p1 & _MASK32
        n0 = hi1 ^ c1 ^ np.uint64(k0)
        n2 = hi0 ^ c3 ^ np.uint64(k1)
        c0, c1, c2, c3 = n0, lo1, n2, lo0
        k0 = (k0 + _W0) & 0xFFFFFFFF
        k1 = (k1 + _W1) & 0xFFFFFFFF
    return (c0.astype(np.uint32), c1.astype(np.uint32),
            c2.astype(np.uint32), c3.astype(np.uint32))


def step_uniforms(seed, step, batch, length, sample_offset=0):
    """float32 [batch, length, 5] uniforms for one reverse step."""
    seed = int(seed)
    samp = np.arange(batch, dtype=np.uint64)[:, None] + np.uint64(sample_offset)
    pos = np.arange(length, dtype=np.uint64)[None, :]
    c1 = samp & _MASK32
    c3 = np.uint64(DOMAIN_TAG) ^ (samp >> np.uint64(32))
    r0, r1, r2, r3 = philox4x32_10(pos, c1, np.uint64(step), c3,
                                   seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    u = np.empty((batch, length, 5), dtype=np.float32)
    scale = np.float32(2.0 ** -24)
    u[..., 0] = (r0 >> np.uint32(8)).astype(np.float32) * scale
    u[..., 1] = (r1 >> np.uint32(8)).astype(np.float32) * scale
    u[..., 2] = (r2 >> np.uint32(8)).astype(np.float32) * scale
    u[..., 3] = (r3 >> np.uint32(8)).astype(np.float32) * scale
    tail = ((r0 & np.uint32(255)) << np.uint32(16)) | ((r1 & np.uint32(255)) << np.uint32(8)) | (r2 & np.uint32(255))
    u[..., 4] = tail.astype(np.float32) * scale
    return u


# Known-answer vectors from the Random123 distribution (kat_vectors, philox4x32 10 rounds).
KAT = [
    ((0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x00000000, 0x00000000),
     (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff),
     (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


WIDE_TAG = 0x50524F54   # 'PROT': domain tag of the wide-vocabulary stream (block index in the top byte)


def step_uniforms_wide(seed, step, batch, length, nclass, sample_offset=0):
    """float32 [batch, length, nclass] uniforms for one reverse step of a vocabulary wider than 5 (protein, V = 22):
    class c takes word c % 4 of Philox block c // 4, counter = (position, sample lo, step,
    WIDE_TAG ^ sample hi ^ (block << 24)), key = seed, U = (word >> 8) * 2^-24."""
    seed = int(seed)
    samp = np.arange(batch, dtype=np.uint64)[:, None] + np.uint64(sample_offset)
    pos = np.arange(length, dtype=np.uint64)[None, :]
    c1 = samp & _MASK32
    u = np.empty((batch, length, nclass), dtype=np.float32)
    scale = np.float32(2.0 ** -24)
    for blk in range((nclass + 3) // 4):
        c3 = np.uint64(WIDE_TAG) ^ (samp >> np.uint64(32)) ^ np.uint64(blk << 24)
        r = philox4x32_10(pos, c1, np.uint64(step), c3, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
        for w in range(4):
            c = 4 * blk + w
            if c < nclass:
                u[..., c] = (r[w] >> np.uint32(8)).astype(np.float32) * scale
    return u
