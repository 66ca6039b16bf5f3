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
