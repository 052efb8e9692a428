"""Dev tool (CPU): NumPy emulation of the kernel's projected-Newton box QP (step_body.h: qp_regs) on QPs collected by
tools/qp_collect.py -- to study how rule changes move the number of Newton solves before touching the kernel."""
import sys, numpy as np
d = np.load(sys.argv[1])
H_, f_, W_, IT, U_ = d["H"], d["f"], d["warm"], d["iters"], d["U"]
lb, ub = -2.0, 2.0

nfix=[0]
def pn(H, f, x0, eact_rel=1e-8, predict=False, max_it=200, eps_rule=None, ls="armijo"):
    N = len(f)
    tol, tight, slack = 1e-9, 1e-12, 1e-14
    eact = eact_rel * (ub - lb)
    gs = np.abs(f) + 2 * np.abs(H).sum(1) * max(abs(lb), abs(ub))
    x = np.clip(x0, lb, ub); hx = H @ x; J0 = x @ (hx + f)
    it = polish = 0; nback = 0; nf = 0; S = np.zeros(N, bool); nsw = 0
    while True:
        g = 2 * hx + f
        e = eact
        if eps_rule is not None:
            w = np.abs(x - np.clip(x - g / np.diag(2 * H), lb, ub)).max()
            e = max(eact, min(eps_rule * (ub - lb), w))
        atl, atu = x <= lb + e, x >= ub - e
        inI = (atl & (g > 0)) | (atu & (g < 0))
        atl0, atu0 = x <= lb + eact, x >= ub - eact
        inI0 = (atl0 & (g > 0)) | (atu0 & (g < 0))
        viol = np.where(inI0, 0.0, np.abs(g)); res = np.abs(x - np.clip(x - g, lb, ub)); xs = np.maximum(np.abs(x), 1)
        bad = ~((viol <= tol * gs) | (res <= tol * xs)); loose = viol > tight * gs
        if not bad.any() and it > 0:
            if not loose.any() or polish >= 2: return x, it, nback, nf, nsw
            polish += 1
        if it >= max_it: return x, it, nback, nf, nsw
        F = ~inI
        nsw += int((S ^ F).sum()); S = F.copy()
        p = np.where(g > 0, lb, np.where(g < 0, ub, x)) - x
        if F.any(): p[F] = -np.linalg.solve(2 * H[np.ix_(F, F)], g[F])
        accepted = False
        if predict and F.any():
            # active-set prediction: free variables the Newton point carries beyond a bound are fixed there and the
            # minimiser over the remaining face is recomputed (one tableau sweep + one mat-vec per round instead of a whole
            # iteration); the result is taken only if it lowers the true cost, else the projected Armijo step below
            F2 = F.copy(); xa = x + p; z0 = x.copy()
            for _ in range(N):
                out = F2 & ((xa > ub) | (xa < lb))
                if not out.any(): break
                nfix[0] += 1 if predict != "all" else out.sum(); nf += 1
                if predict == "ratio":  # the bound that is met first on the way from z to the Newton point
                    step = xa - z0
                    with np.errstate(divide="ignore", invalid="ignore"):
                        al = np.where(out, np.where(step > 0, (ub - z0) / step, (lb - z0) / step), np.inf)
                    j = np.argmin(al); out = np.zeros(N, bool); out[j] = True
                if predict == "lane":  # the offender on the lowest owner lane: variable k sits on lane (k & 7) * 8 + (k >> 3)
                    ks = np.where(out)[0]; j = ks[np.argmin((ks & 7) * 8 + (ks >> 3))]; out = np.zeros(N, bool); out[j] = True
                if predict == "first":  # the earliest offender of the horizon
                    j = np.where(out)[0][0]; out = np.zeros(N, bool); out[j] = True
                if predict == "one":   # only the worst offender per round
                    viol_amt = np.where(out, np.maximum(xa - ub, lb - xa), -1)
                    j = np.argmax(viol_amt); out = np.zeros(N, bool); out[j] = True
                xa = np.where(out, np.where(xa > ub, ub, lb), xa)
                F2 = F2 & ~out
                nsw += int(out.sum()); S = F2.copy()
                rest = ~F2
                xa[rest] = np.clip(xa[rest], lb, ub)
                if F2.any():
                    rhs = -(f[F2] + 2 * H[np.ix_(F2, rest)] @ xa[rest])
                    xa[F2] = np.linalg.solve(2 * H[np.ix_(F2, F2)], rhs)
            xa = np.clip(xa, lb, ub); hxa = H @ xa; Ja = xa @ (hxa + f)
            if Ja <= J0: accepted = True
        if not accepted:
            alpha = 1.0
            while True:
                xa = np.clip(x + alpha * p, lb, ub); hxa = H @ xa; Ja = xa @ (hxa + f)
                pdec = np.where(F, alpha * (-g * p), g * (x - xa)).sum()
                mag = max(abs(J0), abs(Ja))
                if J0 - Ja >= 1e-4 * pdec - slack * mag or alpha < 1e-10: break
                alpha *= 0.25; nback += 1
        x, hx, J0 = xa, hxa, Ja
        it += 1

def run(name, start='warm', **kw):
    its, errs, nb, costs, sweeps = [], [], 0, [], []
    S, Bk = IT.shape
    for s in range(S):
        for b in range(0, Bk, 2):
            x, it, nback, nf, nsw = pn(H_[s, b], f_[s, b], W_[s, b] if start == 'warm' else (np.zeros(20) if start == 'cold' else np.clip(-np.linalg.solve(2 * H_[s, b], f_[s, b]), lb, ub)), **kw)
            its.append(it); costs.append(it + 0.27 * nf + 0.08 * nsw + 0.3 * nback); sweeps.append(nsw); errs.append(np.abs(x - U_[s, b]).max()); nb += nback
    its = np.array(its); costs = np.array(costs)
    print('   sweeps/solve %.1f;' % np.mean(sweeps), end=''); print(' cost (it + 0.27 fix + 0.08 sweep + .3 backtrack): mean %.2f  E[max16] %.2f  p99 %.2f max %.2f' % (costs.mean(), costs.reshape(S, -1, 16).max(2).mean(), np.percentile(costs, 99), costs.max()))
    mx = its.reshape(S, -1, 16).max(2).mean()
    print("  (fixes %d)" % nfix[0], end=""); nfix[0] = 0
    print("%-28s mean %.2f  P(>=3) %.3f P(>=5) %.3f P(>=10) %.4f max %d  E[max16] %.2f  backtracks/solve %.2f  max|x - x_gpu| %.1e" % (
        name, its.mean(), (its >= 3).mean(), (its >= 5).mean(), (its >= 10).mean(), its.max(), mx, nb / its.sum(), max(errs)))
    return its

g = IT[:, ::2].ravel()
print("GPU                          mean %.2f  P(>=3) %.3f P(>=5) %.3f P(>=10) %.4f max %d  E[max16] %.2f" % (
    g.mean(), (g >= 3).mean(), (g >= 5).mean(), (g >= 10).mean(), g.max(), IT[:, ::2].reshape(IT.shape[0], -1, 16).max(2).mean()))
run("emulation (kernel rules)")
run("prediction (worst offender)", predict="one")
run("prediction (lowest lane)", predict="lane")
run("prediction (earliest)", predict="first")
