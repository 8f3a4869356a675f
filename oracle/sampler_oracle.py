"""CPU ORACLE (test infrastructure only): the emcee stretch move, restated.

PARITY UNPINNED.  emcee (pinned 3.1.4, ref: pdm.lock:504-505) is a third-party dependency that is
neither vendored in /root/reference nor installable offline, and the reference holds no test or
golden vector for the sampler.  This file restates the published algorithm (Goodman & Weare 2010,
"Ensemble samplers with affine invariance"; emcee 3.1.x ``EnsembleSampler.sample``,
``moves/red_blue.py:RedBlueMove.propose``, ``moves/stretch.py:StretchMove.get_proposal``) as driven by
the reference's call sites (ref: mcmc.py:83-107, 187-204):

    per step:   move = random.choice(moves, p=weights)          # one uniform draw, single move
                inds = arange(W) % 2 ; random.shuffle(inds)
                for split in (0, 1):
                    s = coords[inds == split] ; c = coords[inds != split]      (current state)
                    zz = ((a - 1) * random.rand(Ns) + 1) ** 2 / a
                    factors = (ndim - 1) * log(zz)
                    rint = random.randint(Nc, size=Ns)
                    q = c[rint] - (c[rint] - s) * zz[:, None]
                    new_lp = log_prob(q)                                       (NaN -> ValueError)
                    accept_i  <=>  factors_i + new_lp_i - lp_i > log(random.rand())   (one draw each)

Two randomness sources feed the same step function:
  * ``EmceeStream``  -- numpy ``RandomState`` (MT19937) consumed in emcee's order; what
    ``gpemu_sampler_step_host_rng`` replays on the device;
  * ``PhiloxStream`` -- the counter-based generator of csrc/k_sampler.hip (Philox4x32-10,
    counter = (index, stream, step_lo, step_hi), key = seed), restated in numpy.
It is validated statistically against analytic Gaussian targets in tests/test_sampler_host.py.
"""
from __future__ import annotations

import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
MASK32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10 block function (Salmon et al. 2011).  uint32 arrays in, 4 out."""
    c0, c1, c2, c3 = (np.asarray(x, dtype=np.uint32) for x in (c0, c1, c2, c3))
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0 = np.uint32(k0)
    k1 = np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            n0 = ((p1 >> np.uint64(32)).astype(np.uint32)) ^ c1 ^ k0
            n1 = (p1 & MASK32).astype(np.uint32)
            n2 = ((p0 >> np.uint64(32)).astype(np.uint32)) ^ c3 ^ k1
            n3 = (p0 & MASK32).astype(np.uint32)
            c0, c1, c2, c3 = n0, n1, n2, n3
            k0 = np.uint32((int(k0) + int(W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def u01(hi, lo):
    v = (hi.astype(np.uint64) << np.uint64(32)) | lo.astype(np.uint64)
    return (v >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


class PhiloxStream:
    """Same draws as rng_step_kernel in csrc/k_sampler.hip."""

    def __init__(self, seed, a=2.0):
        self.k0 = seed & 0xFFFFFFFF
        self.k1 = (seed >> 32) & 0xFFFFFFFF
        self.a = a
        self.step = 0

    def draw(self, W):
        lo, hi = self.step & 0xFFFFFFFF, (self.step >> 32) & 0xFFFFFFFF
        w = np.arange(W, dtype=np.uint32)
        r0 = philox4x32_10(w, 0, lo, hi, self.k0, self.k1)[0]
        keys = (r0.astype(np.uint64) << np.uint64(32)) | w.astype(np.uint64)
        order = np.argsort(keys, kind="stable")
        inds = np.empty(W, dtype=np.int32)
        inds[order] = np.arange(W) & 1
        ns = [(W + 1) // 2, W // 2]
        zz, rint, logu = [], [], []
        for h in range(2):
            i = np.arange(ns[h], dtype=np.uint32)
            r = philox4x32_10(i, 1 + h, lo, hi, self.k0, self.k1)
            r2 = philox4x32_10(i, 3 + h, lo, hi, self.k0, self.k1)
            u = u01(r[0], r[1])
            t = (self.a - 1.0) * u + 1.0
            zz.append(t * t / self.a)
            rint.append(((r[2].astype(np.uint64) * np.uint64(W - ns[h])) >> np.uint64(32)).astype(np.int64))
            with np.errstate(divide="ignore"):
                logu.append(np.log(u01(r2[0], r2[1])))
        self.step += 1
        return inds, zz, rint, logu


class EmceeStream:
    """numpy RandomState consumed in emcee 3.1.x order (see module docstring)."""

    def __init__(self, seed, a=2.0):
        self.random = np.random.RandomState(seed)
        self.a = a

    def draw(self, W):
        rs = self.random
        rs.choice(1, p=[1.0])                       # EnsembleSampler.sample: choice of the move
        inds = np.arange(W) % 2
        rs.shuffle(inds)
        inds = inds.astype(np.int32)
        zz, rint, logu = [], [], []
        for split in range(2):
            ns = int(np.sum(inds == split))
            nc = W - ns
            zz.append(((self.a - 1.0) * rs.rand(ns) + 1) ** 2.0 / self.a)
            rint.append(rs.randint(nc, size=(ns,)).astype(np.int64))
            with np.errstate(divide="ignore"):
                logu.append(np.log(np.array([rs.rand() for _ in range(ns)])))
        return inds, zz, rint, logu


def stretch_step(X, lp, draws, log_prob_fn):
    """One RedBlue stretch step in place.  Returns the boolean accepted mask (W,)."""
    inds, zz, rint, logu = draws
    W, ndim = X.shape
    accepted = np.zeros(W, dtype=bool)
    all_inds = np.arange(W)
    for split in range(2):
        S1 = inds == split
        s = X[S1]
        c = X[~S1]
        z = zz[split]
        factors = (ndim - 1.0) * np.log(z)
        q = c[rint[split]] - (c[rint[split]] - s) * z[:, None]
        new_lp = np.asarray(log_prob_fn(q), dtype=np.float64)
        if np.any(np.isnan(new_lp)):
            raise ValueError("Probability function returned NaN")
        for i, j in enumerate(all_inds[S1]):
            if factors[i] + new_lp[i] - lp[j] > logu[split][i]:
                accepted[j] = True
                X[j] = q[i]
                lp[j] = new_lp[i]
    return accepted


def run(X0, log_prob_fn, stream, steps):
    """Returns chain (steps, W, d), log_prob (steps, W), acceptance counts (W,)."""
    X = np.array(X0, dtype=np.float64)
    lp = np.asarray(log_prob_fn(X), dtype=np.float64)
    W, d = X.shape
    chain = np.empty((steps, W, d))
    lps = np.empty((steps, W))
    nacc = np.zeros(W, dtype=np.int64)
    for t in range(steps):
        nacc += stretch_step(X, lp, stream.draw(W), log_prob_fn)
        chain[t] = X
        lps[t] = lp
    return chain, lps, nacc
