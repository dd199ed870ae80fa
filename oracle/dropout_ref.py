"""TEST INFRASTRUCTURE ONLY (see oracle/README or DESIGN.md section 3): CPU restatement of the training-time dropout mask of
`uds_dropout` (include/uds_hip.h).  The reference applies `keras.layers.Dropout` (`surrogate/emulator.py:199-213,234-235,287-288,
314-318`, `training=fit` at :411,434): inverted dropout, kept elements scaled by 1 / (1 - rate).  Keras draws its mask from
TensorFlow's stateful generator, which cannot be matched bit for bit; parity with the reference is therefore STATISTICAL
(keep fraction, expectation, gradient = the same mask), while the mask of the HIP kernel itself is pinned bit-exactly against
this restatement of Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11; its
known-answer vectors are checked in tests/test_dropout.py).
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(counter, key):
    """counter (..., 4) uint32 words [c0, c1, c2, c3], key (2,) uint32 -> (..., 4) uint32."""
    c = np.asarray(counter, dtype=np.uint64).copy()
    k0, k1 = int(key[0]), int(key[1])
    for _ in range(10):
        p0 = M0 * c[..., 0]
        p1 = M1 * c[..., 2]
        n0 = ((p1 >> np.uint64(32)) ^ c[..., 1] ^ np.uint64(k0)) & MASK32
        n1 = p1 & MASK32
        n2 = ((p0 >> np.uint64(32)) ^ c[..., 3] ^ np.uint64(k1)) & MASK32
        n3 = p0 & MASK32
        c = np.stack([n0, n1, n2, n3], axis=-1)
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return c.astype(np.uint32)


def dropout_mask(n, rate, seed, offset):
    """keep mask (n,) bool of uds_dropout: element i uses counter (offset + i) // 4 (low 64 bits of the 128-bit counter, high 0),
    word (offset + i) % 4, key = seed; kept when word >= ceil(rate * 2^32)."""
    pos = np.arange(n, dtype=np.uint64) + np.uint64(offset)
    ctr = pos >> np.uint64(2)
    counter = np.stack([ctr & MASK32, ctr >> np.uint64(32), np.zeros_like(ctr), np.zeros_like(ctr)], axis=-1)
    words = philox4x32_10(counter, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))
    w = words[np.arange(n), (pos & np.uint64(3)).astype(np.int64)]
    thresh = min(4294967295, int(np.ceil(float(np.float32(rate)) * 4294967296.0)))
    return w >= np.uint32(thresh)


def dropout(x, rate, seed, offset):
    """float64 restatement: x / (1 - rate) where kept (the kernel multiplies by the fp32 value of 1 / (1 - rate))."""
    x = np.asarray(x, dtype=np.float64)
    m = dropout_mask(x.size, rate, seed, offset).reshape(x.shape)
    return np.where(m, x * float(np.float32(1.0) / (np.float32(1.0) - np.float32(rate))), 0.0)
