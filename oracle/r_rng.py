"""Emulation of base R's default RNG (Mersenne-Twister + inversion), test infrastructure only.

Reproduces ``set.seed(s); runif(); rnorm()`` streams so that the three known-answer examples in
the reference's rendered documentation can be regenerated
without an R runtime.  Follows R's documented algorithm (RNG.c: Randomize/FixupSeeds/MT_genrand,
snorm.c: INVERSION):

    set.seed(s):  seed = uint32(s); 50x seed = 69069*seed+1; then 625x the same LCG fills i_seed;
                  i_seed[0] (the MT position) is forced to 624, i_seed[1:625] is the MT state.
    unif_rand():  genrand_int32() * 2.3283064365386963e-10, clamped into (0,1)
    norm_rand():  u = floor(2^27 * unif_rand()) + unif_rand();  qnorm(u / 2^27)   (two uniforms per normal)

qnorm here is scipy.special.ndtri (differs from R's AS241 by <= 1 ulp).
Validated: set.seed(123); runif(3,-.25,.25) = -0.10621124, 0.14415257, -0.04551154.
"""
import numpy as np
from scipy.special import ndtri

_I2_32M1 = 2.328306437080797e-10
_BIG = 134217728.0


class RRng:
    def __init__(self, seed):
        s = np.uint64(np.uint32(seed))
        m = np.uint64(0xFFFFFFFF)
        for _ in range(50):
            s = (np.uint64(69069) * s + np.uint64(1)) & m
        i_seed = np.empty(625, dtype=np.uint32)
        for j in range(625):
            s = (np.uint64(69069) * s + np.uint64(1)) & m
            i_seed[j] = np.uint32(s)
        self._bg = np.random.MT19937()
        st = self._bg.state
        st["state"]["key"] = i_seed[1:].copy()
        st["state"]["pos"] = 624
        self._bg.state = st

    def unif_rand(self, k):
        raw = self._bg.random_raw(int(k)).astype(np.float64)
        v = raw * 2.3283064365386963e-10
        v = np.where(v <= 0.0, 0.5 * _I2_32M1, v)
        v = np.where(1.0 - v <= 0.0, 1.0 - 0.5 * _I2_32M1, v)
        return v

    def runif(self, k, a=0.0, b=1.0):
        return a + (b - a) * self.unif_rand(k)

    def rnorm(self, k, mean=0.0, sd=1.0):
        k = int(k)
        out = np.empty(k, dtype=np.float64)
        chunk = 1 << 20
        for s in range(0, k, chunk):
            m = min(chunk, k - s)
            u = self.unif_rand(2 * m)
            uu = np.floor(_BIG * u[0::2]) + u[1::2]
            out[s:s + m] = ndtri(uu / _BIG)
        return mean + sd * out

    def _unif_index(self, dn):
        """R_unif_index, sample.kind = "Rejection" (the default since R 3.6.0; RNG.c: rbits + rejection)"""
        if dn <= 0:
            return 0
        bits = int(np.ceil(np.log2(dn)))
        while True:
            v = 0
            for _ in range(0, bits + 1, 16):
                v = 65536 * v + int(np.floor(self.unif_rand(1)[0] * 65536))
            if bits < 64:
                v &= (1 << bits) - 1
            if v < dn:
                return v

    def sample(self, x):
        """sample(x): a random permutation of x (do_sample without replacement, k = n)"""
        x = np.asarray(x)
        n = len(x)
        idx = list(range(n))
        out = np.empty(n, dtype=np.int64)
        m = n
        for i in range(n):
            j = self._unif_index(m)
            out[i] = idx[j]
            m -= 1
            idx[j] = idx[m]
        return x[out]
