#!/usr/bin/env python3
"""numpy prototype of the divide & conquer tridiagonal eigensolver that the device path implements
(tools only: validates the formulas against LAPACK before they are written as HIP kernels).

Cuppen's method as organised in LAPACK dstedc/dlaed0-4 (published algorithm; restated, not copied):
leaves by QL, then a binary tree of rank-one merges  D + rho z z^T  with
  * deflation (tiny z_i, or two close d's rotated together),
  * secular equation roots with the origin shifted to the nearest pole (delta_i = (d_i - d_K) - tau),
    safeguarded two-pole rational iteration + bisection,
  * Gu/Eisenstat recomputation of z so that the computed vectors are numerically orthogonal.
"""
import numpy as np

EPS = np.finfo(float).eps


ITERS = []


def secular_root(j, d, w2, rho):
    """Root j of 1 + rho*sum(w2/(d - lam)) in (d[j], d[j+1]) (last: (d[k-1], d[k-1]+rho*sum(w2))).
    Returns (K, tau): lam = d[K] + tau, and delta = (d - d[K]) - tau is what callers must use.
    Iteration: psi (poles <= j) and phi (poles > j) are each replaced by a + b/(pole - t) matching value and
    slope at the current point (Bunch-Nielsen-Sorensen / Li rational model, quadratically convergent), the
    resulting quadratic is solved for the root inside the bracket; bisection safeguards it."""
    k = len(d)
    if j < k - 1:
        gap = d[j + 1] - d[j]
        mid = 0.5 * gap
        dl = (d - d[j]) - mid
        fmid = 1.0 + rho * np.sum(w2 / dl)
        if fmid > 0:      # root in the left half: origin j, tau in (0, mid]
            K, lo, hi = j, 0.0, mid
        else:             # origin j+1, tau in [-mid, 0)
            K, lo, hi = j + 1, -mid, 0.0
    else:
        K = k - 1
        lo, hi = 0.0, rho * np.sum(w2)
    D = d - d[K]
    tau = 0.5 * (lo + hi)
    left = np.arange(k) <= j
    for it in range(100):
        delta = D - tau
        t1 = w2 / delta
        t2 = t1 / delta
        psi, dpsi = rho * np.sum(t1[left]), rho * np.sum(t2[left])
        phi, dphi = rho * np.sum(t1[~left]), rho * np.sum(t2[~left])
        g = 1.0 + psi + phi
        err = 8.0 * EPS * (1.0 + abs(psi) + abs(phi)) + EPS * abs(g)
        if abs(g) <= err:
            break
        if g > 0:
            hi = tau
        else:
            lo = tau
        new = None
        if it < 40:
            dj = delta[j]
            bpsi = dpsi * dj * dj
            apsi = psi - dpsi * dj
            if j < k - 1:
                dj1 = delta[j + 1]
                bphi = dphi * dj1 * dj1
                aphi = phi - dphi * dj1
                c = 1.0 + apsi + aphi
                Dj, Dj1 = D[j], D[j + 1]
                # c (Dj - t)(Dj1 - t) + bpsi (Dj1 - t) + bphi (Dj - t) = 0
                A = c
                Bq = -(c * (Dj + Dj1) + bpsi + bphi)
                Cq = c * Dj * Dj1 + bpsi * Dj1 + bphi * Dj
                cands = []
                if A == 0:
                    if Bq != 0:
                        cands = [-Cq / Bq]
                else:
                    disc = Bq * Bq - 4 * A * Cq
                    if disc >= 0:
                        q = -0.5 * (Bq + np.copysign(np.sqrt(disc), Bq))
                        cands = [q / A] + ([Cq / q] if q != 0 else [])
                for t in cands:
                    if lo < t < hi:
                        new = t
                        break
            else:
                c = 1.0 + apsi
                if c != 0:
                    t = D[j] + bpsi / (-c) if False else D[j] - bpsi / c
                    # c + bpsi/(Dj - t) = 0  ->  t = Dj + bpsi/c
                    t = D[j] + bpsi / c
                    if lo < t < hi:
                        new = t
        if new is None or new == tau:
            new = 0.5 * (lo + hi)
            if not (lo < new < hi):
                break
        tau = new
    ITERS.append(it + 1)
    return K, tau


def merge(d1, Q1, d2, Q2, rho_in):
    """Eigen-decomposition of diag(T1', T2') + rho_in * v v^T, v = e_{n1} + e_{n1+1}-style tear."""
    n1, n2 = len(d1), len(d2)
    n = n1 + n2
    z = np.concatenate([Q1[-1, :], Q2[0, :] * (1.0 if rho_in >= 0 else -1.0)]) / np.sqrt(2.0)
    rho = abs(2.0 * rho_in)
    d = np.concatenate([d1, d2])
    Q = np.zeros((n, n))
    Q[:n1, :n1] = Q1
    Q[n1:, n1:] = Q2
    perm = np.argsort(d, kind="stable")
    d = d[perm]; z = z[perm]; Q = Q[:, perm]
    tol = 8.0 * EPS * max(np.abs(d).max(), np.abs(z).max())
    nondef = []
    deflated = []
    if rho * np.abs(z).max() <= tol:
        deflated = list(range(n))
    else:
        pj = None
        for j in range(n):
            if rho * abs(z[j]) <= tol:
                deflated.append(j)
                continue
            if pj is None:
                pj = j
                continue
            s, c = z[pj], z[j]
            tau = np.hypot(c, s)
            t = d[j] - d[pj]
            c /= tau; s = -s / tau
            if abs(t * c * s) <= tol:
                z[j] = tau; z[pj] = 0.0
                qp, qj = Q[:, pj].copy(), Q[:, j].copy()
                Q[:, pj] = c * qp + s * qj
                Q[:, j] = -s * qp + c * qj
                dp, dj = d[pj], d[j]
                d[pj] = dp * c * c + dj * s * s
                d[j] = dp * s * s + dj * c * c
                deflated.append(pj)
                pj = j
            else:
                nondef.append(pj)
                pj = j
        if pj is not None:
            nondef.append(pj)
    k = len(nondef)
    lam = np.zeros(k)
    Qn = np.zeros((n, k))
    if k > 0:
        dl = d[nondef]; w = z[nondef]
        w2 = w * w
        DELTA = np.zeros((k, k))  # DELTA[i, j] = dl[i] - lam[j]
        for j in range(k):
            K, tau = secular_root(j, dl, w2, rho)
            DELTA[:, j] = (dl - dl[K]) - tau
            lam[j] = dl[K] + tau
        # Gu/Eisenstat: zhat_i^2 = prod_j (lam_j - d_i) / prod_{j!=i} (d_j - d_i) / rho
        zh = np.zeros(k)
        for i in range(k):
            p = DELTA[i, i]
            for j in range(k):
                if j != i:
                    p *= DELTA[i, j] / (dl[i] - dl[j])
            zh[i] = np.copysign(np.sqrt(-p), w[i])
        S = zh[:, None] / DELTA
        S /= np.linalg.norm(S, axis=0)[None, :]
        Qn = Q[:, nondef] @ S
    dd = d[deflated]
    allv = np.concatenate([lam, dd])
    allQ = np.concatenate([Qn, Q[:, deflated]], axis=1)
    o = np.argsort(allv, kind="stable")
    return allv[o], allQ[:, o]


def leaf(d, e):
    T = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
    return np.linalg.eigh(T)


def dc(d, e, leaf_size=8):
    n = len(d)
    if n <= leaf_size:
        return leaf(d, e)
    n1 = n // 2
    rho = e[n1 - 1]
    d1 = d[:n1].copy(); d2 = d[n1:].copy()
    d1[-1] -= abs(rho); d2[0] -= abs(rho)
    w1, Q1 = dc(d1, e[:n1 - 1], leaf_size)
    w2, Q2 = dc(d2, e[n1:], leaf_size)
    return merge(w1, Q1, w2, Q2, rho)


def check(d, e, name):
    from scipy.linalg import eigh_tridiagonal
    n = len(d)
    w, Q = dc(d.copy(), e.copy())
    wr = eigh_tridiagonal(d, e, eigvals_only=True)
    T = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
    nrm = max(np.abs(wr).max(), 1e-300)
    print("%-28s n=%4d  |w-w_ref|/|w|=%.1e  orth=%.1e  resid=%.1e" % (
        name, n, np.abs(w - wr).max() / nrm, np.abs(Q.T @ Q - np.eye(n)).max(), np.abs(T @ Q - Q * w).max() / nrm))


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    for n in (5, 16, 33, 100, 257):
        check(rng.standard_normal(n), rng.standard_normal(n - 1), "random")
    n = 200
    check(np.ones(n) * 2, -np.ones(n - 1), "1-2-1 laplacian")
    check(np.abs(np.arange(n) - n // 2).astype(float), np.ones(n - 1), "wilkinson")
    dg = np.concatenate([np.abs(np.arange(21) - 10).astype(float)] * 5)
    eg = np.ones(len(dg) - 1); eg[20::21] = 1e-8
    check(dg, eg, "glued wilkinson")
    check(np.ones(n), 1e-9 * rng.standard_normal(n - 1), "clustered (tiny e)")
    e0 = rng.standard_normal(n - 1); e0[::7] = 0.0
    check(rng.standard_normal(n), e0, "zeros in e")
    check(10.0 ** (-np.arange(n) / 10.0), 10.0 ** (-np.arange(n - 1) / 10.0) * 0.5, "graded")
    check(rng.standard_normal(n) * 1e150, rng.standard_normal(n - 1) * 1e150, "huge scale")
    it = np.array(ITERS)
    print("secular iterations: mean %.1f median %d p99 %d max %d over %d roots" % (it.mean(), np.median(it), np.percentile(it, 99), it.max(), len(it)))
