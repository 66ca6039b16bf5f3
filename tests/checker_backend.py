"""A CPU stand-in for oem_amd.distributed.HipBackend, used ONLY to exercise the row-sharded driver under gloo.

It is an independent restatement of the three local stages (numpy moments about the common shift; the
moments -> (XX, XY, standardisation) algebra of finalize_kernel; the oracle's path on the resulting Gram), so
the test also checks the moment algebra itself against the plain oracle fit on the unsharded data."""
import numpy as np
import torch

from oracle import oracle as orc


def shift_in_effect(s, p):
    """include/oemgpu.h, oemgpu_shift_sums_dev: c = sample mean if any column has mean^2 > 2^8 var, else 0"""
    m = s[:p + 1] / s[p + 1]
    var = np.maximum(s[p + 2:2 * p + 3] / s[p + 1] - m * m, 0.0)
    return m if np.any(m * m > 256.0 * var) else np.zeros(p + 1)


class CheckerBackend:
    def section(self):
        import contextlib
        return contextlib.nullcontext()

    def shift_in_effect(self):
        return self.shifted

    def new_buffer(self, n):
        return torch.zeros(n, dtype=torch.float64)

    def sum_in_order(self, parts, nparts, n, out):
        acc = parts[:n].clone()
        for r in range(1, nparts):
            acc += parts[r * n:(r + 1) * n]
        out.copy_(acc)

    def shift_sums(self, x, n, ld, p, y, out):
        xn, yn = x.numpy(), y.numpy()
        k = min(n, 64)                                   # any common sample works: the shift is only provisional
        out[:p] = torch.from_numpy(xn[:k].sum(0)); out[p] = float(yn[:k].sum()); out[p + 1] = float(k)
        out[p + 2:2 * p + 2] = torch.from_numpy((xn[:k] ** 2).sum(0)); out[2 * p + 2] = float((yn[:k] ** 2).sum())

    passes = 0

    def moments(self, x, n, ld, p, y, sums, out):
        self.passes += 1
        c = np.zeros(p + 1) if sums is None else shift_in_effect(sums.numpy(), p)
        z = np.column_stack([x.numpy() - c[:p], y.numpy() - c[p], np.ones(n)])
        out.copy_(torch.from_numpy((z.T @ z).ravel()))

    def shift_advised(self):
        return self.advised

    def solve(self, mom, sums, p, semantics, standardize, intercept, args, outs=None):
        assert semantics == 0
        M = mom.numpy().reshape(p + 2, p + 2)
        n = M[p + 1, p + 1]
        if sums is None:                                          # include/oemgpu.h, oemgpu_last_shift_advised
            c = np.zeros(p + 1)
            mean = M[p + 1, :p + 1] / n
            var = np.maximum(np.diag(M)[:p + 1] / n - mean * mean, 0.0)
            self.advised = bool(np.any(mean * mean > 256.0 * var))
        else:
            c = shift_in_effect(sums.numpy(), p)
            self.advised = False
        self.shifted = bool(np.any(c != 0.0))
        sh = M[p + 1, :p + 1]
        mu = c + sh / n
        cen = M[:p + 1, :p + 1] - np.outer(sh, sh) / n
        raw = cen + n * np.outer(mu, mu)
        flag = int(bool(standardize)) + 2 * int(bool(intercept))
        G = cen if flag >= 2 else raw
        sx = np.ones(p)
        if flag & 1:
            sx = np.sqrt(np.maximum(np.diag(cen)[:p], 0) / n); sx[sx == 0] = 1.0
        sy = 1.0 if flag == 0 else np.sqrt(cen[p, p] / n)
        xx = G[:p, :p] / np.outer(sx, sx) / n
        xy = G[p, :p] / (sx * sy) / n
        d = orc.eig_max(xx) * 1.005
        lmax = np.abs(xy).max() * sy
        nl = args.nl
        if args.lam is not None:
            lam = args.lam.copy()
        else:
            base = np.exp(np.linspace(np.log(lmax), np.log(lmax * args.c.lambda_min_ratio), nl))
            lam = np.tile(base, (args.npen, 1))
        pens = [orc.PENALTIES[k] for k in args.pen]
        beta, niter = orc.path(xx, xy, d, lam / sy, penalty=pens, tol=args.c.tol, maxit=args.c.maxit,
                               alpha=args.c.alpha, gamma=args.c.gamma, tau=args.c.tau, penalty_factor=args.pf)
        args.outputs(p + 1)
        for k in range(args.npen):
            for i in range(nl):
                b = beta[k, i].copy()
                if flag & 1:
                    b = b / sx
                if flag:
                    b = b * sy
                args.beta[k, i, 1:] = b
                args.beta[k, i, 0] = (mu[p] - b @ mu[:p]) if flag & 2 else 0.0
        args.lam_out[:] = lam; args.niter[:] = niter; args.loss[:] = 1e99; args.d.value = d

    # ---- xval.oem over row shards (oem_amd.distributed.xval_oem_sharded): an independent restatement of the three phases from
    # the per-fold Gram matrices of Z = [X | y | 1] (ref src/oem_xval_dense.h:358-484, 733-853; src/oem_xval_dense.cpp:343-461),
    # element-wise penalties, no observation weights
    def to_device(self, a):
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64).copy())

    def to_host(self, t):
        return t.numpy().copy()

    def xval_moments_len(self, p, nfolds, weighted):
        assert not weighted
        return nfolds * (p + 2) * (p + 2)

    def xval_fold_moments(self, x, n, ld, p, y, w, foldid, nfolds, args, out):
        assert w is None
        xn, yn, fid = x.numpy(), y.numpy(), np.asarray(foldid)
        self._xv = (xn.copy(), yn.copy(), fid.copy())
        cnt = np.zeros(nfolds, dtype=np.int64)
        m = (p + 2) * (p + 2)
        for k in range(nfolds):
            sel = fid == k + 1
            z = np.column_stack([xn[sel], yn[sel], np.ones(int(sel.sum()))])
            out[k * m:(k + 1) * m] = torch.from_numpy((z.T @ z).ravel())
            cnt[k] = int(sel.sum())
        return cnt

    def xval_solve_folds(self, moments, fold_n_total, n_local, p, nfolds, weighted, standardize, intercept, args):
        K, nl, npen = nfolds, args.nl, args.npen
        M = moments.numpy()[:K * (p + 2) * (p + 2)].reshape(K, p + 2, p + 2)
        off = 1 if intercept else 0
        q = p + off
        pens = [orc.PENALTIES[k] for k in args.pen]
        pf = np.concatenate([np.zeros(off), args.pf])
        args.outputs(p + 1)
        self._bf = np.zeros((K, npen, nl, p + 1))
        lam = None
        for ff in range(K + 1):
            S = sum(M[k] for k in range(K) if k + 1 != ff)
            nobs = S[p + 1, p + 1]
            assert nobs == sum(int(fold_n_total[k]) for k in range(K) if k + 1 != ff)
            cs = np.diag(S)[:p] / (nobs - 1.0)
            cs[cs == 0.0] = 1.0
            ci = 1.0 / np.sqrt(cs)
            XX = np.zeros((q, q)); XY = np.zeros(q)
            XX[off:, off:] = S[:p, :p]; XY[off:] = S[p, :p]
            if intercept:
                XX[0, 0] = nobs; XX[0, 1:] = XX[1:, 0] = S[p + 1, :p]; XY[0] = S[p + 1, p]
            if standardize:
                sc = np.concatenate([np.ones(off), ci])
                XX = XX * np.outer(sc, sc); XY = XY * sc
            XX /= nobs; XY /= nobs
            d = orc.eig_max(XX) * 1.005
            if ff == 0:
                lmax = np.abs(XY[off:]).max()                     # lambda_zero without the intercept slot (ref :1025-1032)
                if args.lam is not None:
                    lam = args.lam.copy()
                else:
                    lam = np.tile(np.exp(np.linspace(np.log(lmax), np.log(lmax * args.c.lambda_min_ratio), nl)), (npen, 1))
                    for k, name in enumerate(pens):
                        if ".net" in name:
                            lam[k] = lam[k] / args.c.alpha
                args.lam_out[:] = lam; args.loss[:] = 1e99; args.d.value = d
            beta, niter = orc.path(XX, XY, d, lam, penalty=pens, tol=args.c.tol, maxit=args.c.maxit, alpha=args.c.alpha,
                                   gamma=args.c.gamma, tau=args.c.tau, penalty_factor=pf)
            for k in range(npen):
                for i in range(nl):
                    b = beta[k, i]
                    full = np.concatenate([[b[0] if intercept else 0.0], b[off:] * (ci if standardize else 1.0)])
                    if ff == 0:
                        args.beta[k, i] = full
                    else:
                        self._bf[ff - 1, k, i] = full
            if ff == 0:
                args.niter[:] = niter

    def xval_cv_triples(self, n_local, p, nfolds, weighted, type_measure, args):
        xn, yn, fid = self._xv
        t = np.zeros((args.npen, args.nl, 3))
        for k in range(args.npen):
            for i in range(args.nl):
                b = self._bf[fid - 1, k, i]                                   # [n_local, p + 1]: the fit that left the row's fold out
                res = yn - (b[:, 0] + np.einsum("ij,ij->i", xn, b[:, 1:]))
                v = np.abs(res) if type_measure == 1 else res * res
                m = v.mean()
                t[k, i] = (len(v), m, ((v - m) ** 2).sum())
        return t

    def xval_merge(self, triples, args):
        t = np.asarray(triples)
        cvm = np.zeros((args.npen, args.nl)); cvsd = np.zeros((args.npen, args.nl))
        for k in range(args.npen):
            for i in range(args.nl):
                n = t[:, k, i, 0].sum()
                mean = (t[:, k, i, 0] * t[:, k, i, 1]).sum() / n
                m2 = (t[:, k, i, 2] + t[:, k, i, 0] * (t[:, k, i, 1] - mean) ** 2).sum()
                cvm[k, i] = mean; cvsd[k, i] = np.sqrt(m2 / (n - 1.0)) / np.sqrt(n)
        return cvm, cvsd
