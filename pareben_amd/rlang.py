"""Host-side restatement of the small pieces of R semantics the CrossValidate() path relies on.

* ``RRandom`` -- R's default RNG (Mersenne-Twister, ``set.seed`` scrambling, ``unif_rand``) and
  ``sample()`` for both sampler generations (R >= 3.6 "Rejection", R < 3.6 "Rounding").  R itself is
  not vendored in the reference tree; this follows R's published algorithm (src/main/RNG.c,
  src/main/unique.c / random.c of R 3.5-4.x) and is pinned by the known answers in
  SURVEY.md 8(c): ``set.seed(1); runif(3)`` and ``set.seed(1); sample(10)`` for both samplers.
* ``r_seq_by`` -- ``seq(from, to, by)`` (``from + i*by``; the values parEBEN's grids are made of).
* ``r_sd`` -- ``sd()`` (n-1 denominator).
"""
import math
import numpy as np

_N = 624
_M = 397
_I2_32M1 = 2.328306437080797e-10  # = 1/(2^32 - 1), R's MT_genrand scale


class RRandom:
    """R's Mersenne-Twister stream as seeded by ``set.seed(seed)``."""

    def __init__(self, seed=1, sample_kind="Rejection"):
        if sample_kind not in ("Rejection", "Rounding"):
            raise ValueError("sample_kind must be 'Rejection' (R >= 3.6) or 'Rounding' (R < 3.6)")
        self.sample_kind = sample_kind
        s = np.uint32(seed & 0xFFFFFFFF)
        with np.errstate(over="ignore"):
            # Randomize(): initial scrambling, then one LCG draw per state word (625 words)
            for _ in range(50):
                s = np.uint32(69069) * s + np.uint32(1)
            st = np.empty(_N + 1, dtype=np.uint32)
            for j in range(_N + 1):
                s = np.uint32(69069) * s + np.uint32(1)
                st[j] = s
        # FixupSeeds: dummy[0] = mti = N  => the state vector is regenerated on the first draw
        self.mt = [int(v) for v in st[1:]]
        self.mti = _N

    def _genrand(self):
        mt = self.mt
        if self.mti >= _N:
            for kk in range(_N):
                y = (mt[kk] & 0x80000000) | (mt[(kk + 1) % _N] & 0x7FFFFFFF)
                v = mt[(kk + _M) % _N] ^ (y >> 1)
                if y & 1:
                    v ^= 0x9908B0DF
                mt[kk] = v
            self.mti = 0
        y = mt[self.mti]
        self.mti += 1
        y ^= y >> 11
        y ^= (y << 7) & 0x9D2C5680
        y ^= (y << 15) & 0xEFC60000
        y ^= y >> 18
        return y & 0xFFFFFFFF

    def unif_rand(self):
        v = self._genrand() * 2.3283064365386963e-10  # [0,1)
        # fixup(): keep strictly inside (0,1)
        if v <= 0.0:
            return 0.5 * _I2_32M1
        if 1.0 - v <= 0.0:
            return 1.0 - 0.5 * _I2_32M1
        return v

    def _rbits(self, bits):
        v = 0
        n = 0
        while n <= bits:
            v1 = int(math.floor(self.unif_rand() * 65536))
            v = 65536 * v + v1
            n += 16
        if bits < 64:
            v &= (1 << bits) - 1
        return float(v)

    def unif_index(self, dn):
        if self.sample_kind == "Rounding":
            return math.floor(dn * self.unif_rand())
        if dn <= 0:
            return 0.0
        bits = int(math.ceil(math.log2(dn)))
        while True:
            dv = self._rbits(bits)
            if dv < dn:
                return dv

    def sample(self, x, size=None):
        """``sample(x, size)`` without replacement (R's do_sample); default size = length(x): a random
        permutation.  The first `size` picks do not depend on `size`."""
        x = list(x)
        n = len(x)
        idx = list(range(n))
        out = []
        m = n
        for _ in range(n if size is None else int(size)):
            j = int(self.unif_index(m))
            out.append(x[idx[j]])
            m -= 1
            idx[j] = idx[m]
        return out


def r_seq_by(frm, to, by):
    """R's seq.default(from, to, by): n = floor((to-from)/by + 1e-10), values from + (0..n)*by,
    then clamped so rounding cannot overshoot ``to`` (pmin for by > 0, pmax for by < 0) -- which is
    why the stored real-R grids end in exactly 0.05 although 1 + 19*(-0.05) = 0.04999999999999993."""
    n = int(math.floor((to - frm) / by + 1e-10))
    v = np.array([frm + i * by for i in range(n + 1)], dtype=np.float64)
    return np.minimum(v, to) if by > 0 else np.maximum(v, to)


def r_sd(v):
    v = np.asarray(v, dtype=np.float64)
    return float(np.std(v, ddof=1))
