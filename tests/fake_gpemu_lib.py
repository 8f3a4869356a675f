"""TEST DOUBLE for ``libgpemu.so`` -- test infrastructure only, importable from ``tests/`` alone.

A Python object with the C ABI's entry points (``include/gpemu.h``) that the ``gpemu`` host glue calls through
``gpemu._lib.lib()``, every one implemented with the CPU oracle (``oracle/gp_oracle.py``, ``oracle/sampler_oracle.py``).
It takes the same ctypes arguments the real library gets -- raw pointers, sizes, handles -- so everything ABOVE the C ABI
runs unchanged: argument marshalling, the estimator classes, the lock-step L-BFGS-B driver, the drop-in
``emulation`` / ``log_posterior`` / ``mcmc`` modules, the emcee facade, the HDF5 writer.  Used by
``tests/test_integration_rehearsal.py`` to walk the reference's real ``steer_analysis`` through the drop-in modules in the
build container, which has no GPU.  It is NOT a CPU fallback of the product: nothing under ``bayesian-inference_amd/``
imports it, and the product keeps failing loudly without a HIP device (``tests/test_dropin_host.py``).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from oracle import gp_oracle as O
from oracle import sampler_oracle as SO

OK = 0
_keep = {}          # handle id -> python object


def _addr(p):
    if p is None:
        return 0
    if isinstance(p, int):
        return p
    if isinstance(p, C.c_void_p):
        return p.value or 0
    return C.cast(p, C.c_void_p).value or 0


def _arr(p, shape, dtype=np.float64):
    """numpy view of the caller's buffer at pointer ``p`` (None for a null pointer)."""
    a = _addr(p)
    if a == 0:
        return None
    n = int(np.prod(shape)) if len(shape) else 1
    ct = {np.float64: C.c_double, np.int64: C.c_int64, np.int32: C.c_int32, np.uint64: C.c_uint64}[dtype]
    return np.ctypeslib.as_array((ct * n).from_address(a)).reshape(shape)


def _new_handle(out_pp, obj):
    hid = 0x1000 + 16 * (len(_keep) + 1)
    _keep[hid] = obj
    C.cast(out_pp, C.POINTER(C.c_void_p))[0] = C.c_void_p(hid)
    return hid


def _obj(h):
    return _keep[_addr(h)]


class _Fit:
    def __init__(self, X, spec, jitter):
        self.X, self.spec, self.jitter = X, spec, jitter


class _Model:
    def __init__(self, model, cun):
        self.model, self.cun = model, cun
        self.setups = {}        # n_div -> [per chain: block setups]
        self.lik = None


class _Sampler:
    pass


class FakeLib:
    """The entry points of include/gpemu.h used by the host glue, oracle-backed."""

    def __init__(self):
        self._err = b""
        self.calls = {}

    def _count(self, name):
        self.calls[name] = self.calls.get(name, 0) + 1

    # ---- library ---------------------------------------------------------------------------------
    def gpemu_version(self):
        return b"gpemu TEST DOUBLE (oracle-backed, tests only)"

    def gpemu_last_error(self):
        return self._err

    def gpemu_device_count(self):
        return 1

    def gpemu_device_name(self, device, buf, buflen):
        return OK

    # ---- PCA / truncation covariance ---------------------------------------------------------------
    def gpemu_pca_fit(self, device, N, F, Y, nc, smean, sscale, svar, pmean, comp, ev, evr, Ypca, flip, nsweeps):
        self._count("pca_fit")
        Yv = _arr(Y, (N, F))
        mean, scale, var = O.scaler_fit(Yv)
        pca = O.pca_fit((Yv - mean) / scale)
        nc = int(nc) if nc > 0 else min(N, F)
        _arr(smean, (F,))[:] = mean
        _arr(sscale, (F,))[:] = scale
        _arr(svar, (F,))[:] = var
        _arr(pmean, (F,))[:] = ((Yv - mean) / scale).mean(axis=0)
        _arr(comp, (nc, F))[:] = pca["components"][:nc]
        _arr(ev, (nc,))[:] = pca["explained_variance"][:nc]
        _arr(evr, (nc,))[:] = pca["explained_variance_ratio"][:nc]
        _arr(Ypca, (N, nc))[:] = pca["Y_pca"][:, :nc]
        _arr(flip, (nc,), np.int64)[:] = pca["flip_argmax"][:nc]
        _arr(nsweeps, (1,), np.int64)[0] = 0
        return OK

    def gpemu_truncation_cov(self, device, n_comp, F, n_pc, comp, ev, cov_out):
        c, e = _arr(comp, (n_comp, F)), _arr(ev, (n_comp,))
        S_un = c.T[:, n_pc:]
        _arr(cov_out, (F, F))[:] = S_un.dot(np.diag(e[n_pc:]).dot(S_un.T))      # ref: emulation.py:246-249
        return OK

    # ---- fit ------------------------------------------------------------------------------------------
    def gpemu_fit_create(self, out, device, N, d, X, kind, nu, has_const, has_noise, jitter):
        spec = O.KernelSpec(kind=int(kind), nu=float(nu) if kind == O.MATERN else np.inf, has_const=bool(has_const),
                            has_noise=bool(has_noise))
        _new_handle(out, _Fit(_arr(X, (N, d)).copy(), spec, float(jitter)))
        return OK

    def gpemu_fit_destroy(self, h):
        _keep.pop(_addr(h), None)
        return OK

    def _lml(self, f, y, theta, want_grad):
        try:
            lml, grad = O.lml_and_grad(f.X, y, theta, f.spec, f.jitter)
        except np.linalg.LinAlgError:
            return None, None
        return lml, grad

    def gpemu_fit_lml(self, h, y, theta, n_theta, lml_out, grad_out):
        self._count("fit_lml")
        f = _obj(h)
        lml, grad = self._lml(f, _arr(y, (f.X.shape[0],)), _arr(theta, (n_theta,)), _addr(grad_out) != 0)
        if lml is None:
            self._err = b"kernel matrix is not positive definite"
            return 1
        C.cast(lml_out, C.POINTER(C.c_double))[0] = lml
        g = _arr(grad_out, (n_theta,))
        if g is not None:
            g[:] = grad
        return OK

    def gpemu_fit_lml_batch(self, h, n, ys, thetas, n_theta, lml_out, grad_out, info_out):
        self._count("fit_lml_batch")
        f = _obj(h)
        N = f.X.shape[0]
        Y, T = _arr(ys, (n, N)), _arr(thetas, (n, n_theta))
        L, G, I = _arr(lml_out, (n,)), _arr(grad_out, (n, n_theta)), _arr(info_out, (n,), np.int32)
        for i in range(n):
            lml, grad = self._lml(f, Y[i], T[i], G is not None)
            if lml is None:
                L[i] = np.nan
                if G is not None:
                    G[i] = np.nan
                I[i] = 1
            else:
                L[i] = lml
                if G is not None:
                    G[i] = grad
                I[i] = 0
        return OK

    def gpemu_fit_factor(self, h, y, theta, n_theta, L_out, alpha_out, lml_out):
        f = _obj(h)
        N = f.X.shape[0]
        try:
            gp = O.gp_fit_at_theta(f.X, _arr(y, (N,)), _arr(theta, (n_theta,)).copy(), f.spec, f.jitter)
        except np.linalg.LinAlgError:
            self._err = b"kernel matrix is not positive definite"
            return 1
        _arr(L_out, (N, N))[:] = gp.L
        _arr(alpha_out, (N,))[:] = gp.alpha
        lml, _ = O.lml_and_grad(f.X, _arr(y, (N,)), _arr(theta, (n_theta,)), f.spec, f.jitter)
        C.cast(lml_out, C.POINTER(C.c_double))[0] = lml
        return OK

    # ---- model ----------------------------------------------------------------------------------------
    def gpemu_model_create(self, out, device, N, d, F, k, kind, nu, has_const, has_noise, X, ls, constv, noise, alpha, L,
                           comp, smean, sscale, cun):
        spec = O.KernelSpec(kind=int(kind), nu=float(nu) if kind == O.MATERN else np.inf, has_const=bool(has_const),
                            has_noise=bool(has_noise))
        lsv, av, Lv = _arr(ls, (k, d)), _arr(alpha, (k, N)), _arr(L, (k, N, N))
        cv, nv = _arr(constv, (k,)), _arr(noise, (k,))
        gps = [O.GP(ls=lsv[i].copy(), const=float(cv[i]) if cv is not None else 0.0,
                    noise=float(nv[i]) if nv is not None else 0.0, L=Lv[i].copy(), alpha=av[i].copy()) for i in range(k)]
        comps = _arr(comp, (k, F)).copy()
        model = O.GroupModel(X_train=_arr(X, (N, d)).copy(), spec=spec, gps=gps, components=comps,
                             explained_variance=np.zeros(k), scaler_mean=_arr(smean, (F,)).copy(),
                             scaler_scale=_arr(sscale, (F,)).copy(), n_pc=k)
        cu = _arr(cun, (F, F))
        _new_handle(out, _Model(model, np.zeros((F, F)) if cu is None else cu.copy()))
        return OK

    def gpemu_model_destroy(self, h):
        _keep.pop(_addr(h), None)
        return OK

    def gpemu_model_sync(self, h):
        return OK

    def gpemu_gp_predict(self, h, B, X, mean_out, var_out):
        m = _obj(h)
        d, k = m.model.X_train.shape[1], m.model.n_pc
        mu, var = O.gp_predict_all(_arr(X, (B, d)), m.model)
        _arr(mean_out, (B, k))[:] = mu
        _arr(var_out, (B, k))[:] = var
        return OK

    def gpemu_predict_full(self, h, B, X, n_div, cv_out, cov_out):
        self._count("predict_full")
        m = _obj(h)
        d, F = m.model.X_train.shape[1], m.model.components.shape[1]
        out = O.predict_group(_arr(X, (B, d)), m.model, m.cun * (B / float(n_div)))      # predict_group divides by B
        _arr(cv_out, (B, F))[:] = out["central_value"]
        _arr(cov_out, (B, F, F))[:] = out["cov"]
        return OK

    def _setup(self, m, n_div):
        key = float(n_div)
        if key not in m.setups:
            y, yerr, lo, hi, bs = m.lik
            m.setups[key] = [O.lowrank_setup_blocks(m.model, y[c], yerr, bs, n_div=n_div, cov_unexpl=m.cun)
                             for c in range(y.shape[0])]
        return m.setups[key]

    def gpemu_likelihood_setup(self, h, y_exp, y_err, lo, hi, n_div, nblk, block_start):
        return self.gpemu_likelihood_setup_chains(h, 1, y_exp, y_err, lo, hi, n_div, nblk, block_start)

    def gpemu_likelihood_setup_chains(self, h, n_chains, y_exp, y_err, lo, hi, n_div, nblk, block_start):
        self._count("likelihood_setup")
        m = _obj(h)
        d, F = m.model.X_train.shape[1], m.model.components.shape[1]
        bs = [0, F] if nblk == 0 else [int(v) for v in _arr(block_start, (nblk + 1,), np.int64)]
        m.lik = (_arr(y_exp, (n_chains, F)).copy(), _arr(y_err, (F,)).copy(), _arr(lo, (d,)).copy(), _arr(hi, (d,)).copy(), bs)
        m.setups = {}
        m.n_div = float(n_div)
        self._setup(m, n_div)
        return OK

    def _logpost_rows(self, m, X, chain_of_row=None):
        y, yerr, lo, hi, bs = m.lik
        sets = self._setup(m, m.n_div)
        out = np.full(X.shape[0], -np.inf)
        inside = np.all((X > lo) & (X < hi), axis=1)
        if inside.any():
            mu, var = O.gp_predict_all(X[inside], m.model)
            rows = np.flatnonzero(inside)
            for j, r in enumerate(rows):
                c = 0 if chain_of_row is None else chain_of_row[r]
                out[r] = O.loglik_lowrank_blocks(mu[j], var[j], sets[c])
        return out

    def gpemu_logpost(self, h, B, X, out, mode):
        self._count("logpost")
        m = _obj(h)
        _arr(out, (B,))[:] = self._logpost_rows(m, _arr(X, (B, m.model.X_train.shape[1])))
        return OK

    # ---- sampler --------------------------------------------------------------------------------------
    def gpemu_sampler_create(self, out, groups, n_groups, W, a, seed):
        sd = np.array([seed if isinstance(seed, int) else seed.value], dtype=np.uint64)
        return self._sampler_create(out, groups, n_groups, W, a, sd)

    def gpemu_sampler_create_chains(self, out, groups, n_groups, W, a, seeds, n_chains):
        return self._sampler_create(out, groups, n_groups, W, a, _arr(seeds, (n_chains,), np.uint64).copy())

    def _sampler_create(self, out, groups, n_groups, Wc, a, seeds):
        s = _Sampler()
        gp = C.cast(groups, C.POINTER(C.c_void_p))
        s.models = [_keep[gp[i]] for i in range(n_groups)]
        s.Wc, s.a, s.nch = int(Wc), float(a), int(len(seeds))
        s.W = s.Wc * s.nch
        s.d = s.models[0].model.X_train.shape[1]
        s.streams = [SO.PhiloxStream(int(sd), a=float(a)) for sd in seeds]
        s.X = s.lp = None
        s.chain, s.lps = [], []
        s.nacc = np.zeros(s.W, dtype=np.int64)
        s.iterations = 0
        _new_handle(out, s)
        return OK

    def gpemu_sampler_destroy(self, h):
        _keep.pop(_addr(h), None)
        return OK

    def _lp(self, s, c):
        def fn(Xq):
            Xq = np.atleast_2d(Xq)
            tot = np.zeros(Xq.shape[0])
            for m in s.models:
                keep = m.n_div
                m.n_div = 1.0                       # the sampler evaluates one walker per call (n_div = 1)
                tot = tot + self._logpost_rows(m, Xq, None if s.nch == 1 else np.full(Xq.shape[0], c))
                m.n_div = keep
            return tot
        return fn

    def gpemu_sampler_set_state(self, h, X0, logp0):
        s = _obj(h)
        s.X = _arr(X0, (s.W, s.d)).copy()
        lp = _arr(logp0, (s.W,))
        if lp is not None:
            s.lp = lp.copy()
        else:
            s.lp = np.concatenate([self._lp(s, c)(s.X[c * s.Wc:(c + 1) * s.Wc]) for c in range(s.nch)])
        return OK

    def gpemu_sampler_get_state(self, h, X, logp):
        s = _obj(h)
        xv, lv = _arr(X, (s.W, s.d)), _arr(logp, (s.W,))
        if xv is not None:
            xv[:] = s.X
        if lv is not None:
            lv[:] = s.lp
        return OK

    def gpemu_sampler_reset(self, h):
        s = _obj(h)
        s.chain, s.lps = [], []
        s.nacc[:] = 0
        s.iterations = 0
        return OK

    def gpemu_sampler_reserve_chain(self, h, n):
        return OK

    def gpemu_sampler_run(self, h, steps, store):
        self._count("sampler_run")
        s = _obj(h)
        for _ in range(int(steps)):
            for c in range(s.nch):
                sl = slice(c * s.Wc, (c + 1) * s.Wc)
                Xc, lpc = s.X[sl], s.lp[sl]
                try:
                    acc = SO.stretch_step(Xc, lpc, s.streams[c].draw(s.Wc), self._lp(s, c))
                except ValueError:
                    self._err = b"log-probability returned NaN"
                    return 1
                s.nacc[sl] += acc
            s.iterations += 1
            if store:
                s.chain.append(s.X.copy())
                s.lps.append(s.lp.copy())
        return OK

    def gpemu_sampler_get_counts(self, h, nacc, iters, clen):
        s = _obj(h)
        v = _arr(nacc, (s.W,), np.int64)
        if v is not None:
            v[:] = s.nacc
        if _addr(iters):
            C.cast(iters, C.POINTER(C.c_int64))[0] = s.iterations
        if _addr(clen):
            C.cast(clen, C.POINTER(C.c_int64))[0] = len(s.chain)
        return OK

    def gpemu_sampler_get_chain(self, h, first, n, chain_out, lp_out):
        s = _obj(h)
        if n > 0:
            c, l = _arr(chain_out, (n, s.W, s.d)), _arr(lp_out, (n, s.W))
            if c is not None:
                c[:] = np.stack(s.chain[first:first + n])
            if l is not None:
                l[:] = np.stack(s.lps[first:first + n])
        return OK

    def gpemu_sampler_acf(self, h, first, n_steps, w0, nw, lag0, n_lags, f_out):
        s = _obj(h)
        x = np.stack(s.chain[first:first + n_steps])[:, w0:w0 + nw]
        from gpemu.sampler import function_1d
        f = np.zeros((n_steps, s.d))
        with np.errstate(invalid="ignore", divide="ignore"):
            for dd in range(s.d):
                for w in range(nw):
                    f[:, dd] += function_1d(x[:, w, dd])
        f /= nw
        _arr(f_out, (n_lags, s.d))[:] = f[lag0:lag0 + n_lags]
        return OK


def install(monkeypatch=None):
    """Put a FakeLib behind ``gpemu._lib.lib()``.  Returns it (``.calls`` counts what the glue asked for)."""
    from gpemu import _lib
    fake = FakeLib()
    if monkeypatch is not None:
        monkeypatch.setattr(_lib, "lib", lambda: fake)
        monkeypatch.setattr(_lib, "_lib", fake, raising=False)
    else:
        _lib.lib = lambda: fake
        _lib._lib = fake
    return fake
