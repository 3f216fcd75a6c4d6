"""Closed-form deterministic generators shared by the golden-vector script and the tests.

TEST INFRASTRUCTURE ONLY (see oracle/README.md): nothing in the shipped product path
(`mebt_amd/`, the C-ABI library, `bench.py`'s GPU leg) imports this file.

Everything here is exact integer / float64 arithmetic followed by one cast to float32, so the
values are bit-identical on every machine (no libm, no torch RNG stream involved).  The golden
script loads these weights into the *reference* model ("closed-form deterministic
weight generator"), the tests load them into the oracle and into the HIP path.
"""
import zlib
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64_(x, tmp):
    """In-place vectorised splitmix64 finaliser on a uint64 array (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        x += np.uint64(0x9E3779B97F4A7C15)
        np.right_shift(x, np.uint64(30), out=tmp); x ^= tmp; x *= np.uint64(0xBF58476D1CE4E5B9)
        np.right_shift(x, np.uint64(27), out=tmp); x ^= tmp; x *= np.uint64(0x94D049BB133111EB)
        np.right_shift(x, np.uint64(31), out=tmp); x ^= tmp
    return x


def _name_key(name):
    return np.uint64(zlib.crc32(name.encode()) & 0xFFFFFFFF) << np.uint64(32)


_CHUNK = 1 << 20


def hash_u64(name, n, stream=0):
    """h[i] = splitmix64(splitmix64(i ^ key) + key), key from crc32(name) and the stream id.
    Chunked + in-place so that large tensors do not page-fault a dozen temporaries."""
    key = _name_key(name) ^ np.uint64(stream * 0x51ED27 + 1)
    out = np.empty(n, dtype=np.uint64)
    tmp = np.empty(min(n, _CHUNK), dtype=np.uint64)
    for s in range(0, n, _CHUNK):
        e = min(n, s + _CHUNK)
        x = out[s:e]
        x[:] = np.arange(s, e, dtype=np.uint64)
        x ^= key
        _splitmix64_(x, tmp[: e - s])
        with np.errstate(over="ignore"):
            x += key
        _splitmix64_(x, tmp[: e - s])
    return out


def pseudo_normal(name, shape, std=0.02, stream=0):
    """~N(0, std^2): Irwin-Hall sum of four 16-bit uniforms (exact in float64)."""
    n = int(np.prod(shape)) if len(shape) else 1
    h = hash_u64(name, n, stream)
    out = np.empty(n, dtype=np.float32)
    m = np.uint64(0xFFFF)
    c = float(np.sqrt(3.0)) * float(std)
    for s in range(0, n, _CHUNK):
        hc = h[s:s + _CHUNK]
        acc = (hc & m).astype(np.float64)
        for sh in (16, 32, 48):
            acc += ((hc >> np.uint64(sh)) & m).astype(np.float64)
        acc /= 65536.0
        acc -= 2.0
        acc *= c                                     # unit variance * std
        out[s:s + _CHUNK] = acc.astype(np.float32)
    return out.reshape(shape)


def uniform01(name, shape, stream=0):
    """Uniform in (0,1), 32-bit resolution, never 0 or 1."""
    n = int(np.prod(shape)) if len(shape) else 1
    h = hash_u64(name, n, stream)
    return (((h >> np.uint64(32)).astype(np.float64) + 0.5) / 4294967296.0).reshape(shape)


def exp1_noise(name, shape, stream=0):
    """Exp(1) noise (stand-in for `tensor.exponential_()`, reference mebt/transformer.py:837 and
    mebt/mask_sampler.py:182), float32."""
    return (-np.log(uniform01(name, shape, stream))).astype(np.float32)


def randint(name, shape, hi, stream=0):
    n = int(np.prod(shape)) if len(shape) else 1
    return (hash_u64(name, n, stream) % np.uint64(hi)).astype(np.int64).reshape(shape)


def permutation(name, n, stream=0):
    """A permutation of range(n): argsort of hash keys (stable, ties impossible in practice)."""
    return np.argsort(hash_u64(name, n, stream), kind="stable").astype(np.int64)


def param_value(name, shape):
    """Closed-form value of a model parameter, by state-dict name (SURVEY.md §A.2 schema).

    Unlike the reference's init (gpt.py:225-232: bias 0, LN weight 1) every parameter gets a
    non-trivial value so that a kernel that drops a bias or an LN affine term cannot pass."""
    if name.endswith("ln1.weight") or name.endswith("ln2.weight") or name.endswith("ln_f.weight"):
        return (1.0 + pseudo_normal(name, shape, std=0.05)).astype(np.float32)
    if name.endswith(".bias"):
        return pseudo_normal(name, shape, std=0.02)
    return pseudo_normal(name, shape, std=0.02)


def state_dict_numpy(shapes):
    """shapes: {name: shape} -> {name: float32 array}."""
    return {k: param_value(k, tuple(v)) for k, v in shapes.items()}


def exp1_counter(seed, rows, V):
    """CPU twin of the in-kernel Exp(1) generator of the sampler (mebt_amd/csrc/sampler.hip:exp1_counter): element (row, e)
    hashes the 64-bit counter row * V + e with the 64-bit seed (3 multiply / xor-shift rounds on 32 bits), 24 bits -> u in
    (0, 1) -> -log(u).  The uniform bits are reproduced exactly; the logarithm to float32 rounding."""
    idx = np.arange(rows * V, dtype=np.uint64)
    seed = np.uint64(seed)
    M = np.uint64(0xFFFFFFFF)
    lo, hi = idx & M, idx >> np.uint64(32)
    h = ((lo ^ (seed & M)) + ((hi + np.uint64(0x5A3B1E)) & M) * np.uint64(0x632BE5AB) + (seed >> np.uint64(32))) & M
    h = (h * np.uint64(0x9E3779B1)) & M
    h ^= h >> np.uint64(15)
    h = (h * np.uint64(0x85EBCA77)) & M
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0xC2B2AE35)) & M
    h ^= h >> np.uint64(16)
    u = ((h >> np.uint64(8)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)
    return (-np.log(u.astype(np.float64))).astype(np.float32).reshape(rows, V)


def uniform_counter(seed, rows):
    """CPU twin of the per-row uniform of the sampler's inverse-CDF draw (mebt_amd/csrc/sampler.hip: `pair_hash(seed, 0x5A3B1F, row)`
    + one more multiply / xor-shift round, 24 bits -> u in (0, 1)).  Exact."""
    idx = np.arange(rows, dtype=np.uint64)
    seed = np.uint64(seed)
    M = np.uint64(0xFFFFFFFF)
    lo, hi = idx & M, idx >> np.uint64(32)
    h = ((lo ^ (seed & M)) + ((hi + np.uint64(0x5A3B1F)) & M) * np.uint64(0x632BE5AB) + (seed >> np.uint64(32))) & M
    h = (h * np.uint64(0x9E3779B1)) & M
    h ^= h >> np.uint64(15)
    h = (h * np.uint64(0x85EBCA77)) & M
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0xC2B2AE35)) & M
    h ^= h >> np.uint64(16)
    return ((h >> np.uint64(8)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)
