"""Host restatement of the library's dropout decision (include/franken_hip.h, fk_dropout): numpy uint32 arithmetic, used by the GPU tests to
predict every mask bit for bit.  keep <=> mix32(mix32(mix32(hi ^ seed) + step * 0x85EBCA6B + site) ^ (lo * 0x9E3779B9)) >= p * 2^32."""
import numpy as np

M32 = np.uint64(0xFFFFFFFF)


def mix32(x):
    x = np.asarray(x, dtype=np.uint64) & M32
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & M32
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & M32
    x ^= x >> np.uint64(16)
    return x


def threshold(p):
    t = float(np.float32(p)) * 4294967296.0
    return 4294967295 if t >= 4294967295.0 else (1 if t < 1.0 else int(t))


def keep(seed, step, site, hi, lo, p):
    """hi, lo: broadcastable integer arrays (the two index words) -> boolean keep mask"""
    salt = np.uint64((int(step) * 0x85EBCA6B + int(site)) & 0xFFFFFFFF)
    row = mix32((mix32(np.asarray(hi, dtype=np.uint64) ^ np.uint64(int(seed) & 0xFFFFFFFF)) + salt) & M32)
    bits = mix32(row ^ ((np.asarray(lo, dtype=np.uint64) * np.uint64(0x9E3779B9)) & M32))
    return bits >= np.uint64(threshold(p))


def keep_flat(seed, step, site, n, p):
    """elementwise dropout over n contiguous elements (hi = index >> 32 = 0 below 2^32 elements)"""
    return keep(seed, step, site, 0, np.arange(n, dtype=np.uint64), p)


def keep_attention(seed, step, site, B, H, Nq, Nk, p):
    """[B, H, Nq, Nk]: hi = (b * H + h) * Nq + q, lo = key"""
    hi = np.arange(B * H * Nq, dtype=np.uint64).reshape(B, H, Nq, 1)
    lo = np.arange(Nk, dtype=np.uint64).reshape(1, 1, 1, Nk)
    return keep(seed, step, site, hi, lo, p)
