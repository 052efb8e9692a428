#!/usr/bin/env python3
"""Golden vectors for the DARE / LQR terminal ingredients: the reference's own
``solve_DARE`` and ``dlqr`` (duffing.py:583-613) are taken out of the script's syntax tree -- the script itself is
not run -- and evaluated on (a) the reference's own call (offline A, B of the golden loop, Q = 10 I, R = 0.01,
duffing.py:667-671), (b) online-updated models of the golden loop, (c) seeded random models at L = 20.
Runs only in the build container (needs /root/reference); stores DATA only (inputs and the computed P, K).

    python tests/golden/make_golden_dare.py
"""
import ast
import os

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def reference_functions():
    tree = ast.parse(open(os.path.join(REF, "duffing.py")).read())
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("solve_DARE", "dlqr")]
    assert len(keep) == 2
    ns = {"np": np}
    exec(compile(ast.Module(body=keep, type_ignores=[]), "duffing.py", "exec"), ns)
    return ns["solve_DARE"], ns["dlqr"]


def main():
    solve_DARE, dlqr = reference_functions()
    g = np.load(os.path.join(HERE, "duffing_loop.npz"))
    cases = []
    L = g["A0"].shape[0]
    cases.append((g["A0"], g["B0"].reshape(L, 1), 10.0 * np.eye(L), 0.01))          # the reference's own call
    for i in (5, 60, 125):                                                          # online-updated models
        cases.append((g["loop_Ap"][i], g["loop_Bp"][i].reshape(L, 1), 10.0 * np.eye(L), 0.01))
    qd = np.zeros(L); qd[:2] = 100.0                                                # Koopman_update.m:279 Q_Lift
    cases.append((g["A0"], g["B0"].reshape(L, 1), np.diag(qd), 0.01))
    rng = np.random.RandomState(11)
    for L2 in (20, 20, 32):
        A = rng.randn(L2, L2)
        A *= 0.9 / np.abs(np.linalg.eigvals(A)).max()
        cases.append((A, rng.randn(L2, 1), 10.0 * np.eye(L2), 0.01))
    out = {"n": np.array(len(cases))}
    for k, (A, B, Q, R) in enumerate(cases):
        P = solve_DARE(A, B, Q, R)
        K = dlqr(A, B, Q, R)
        out.update({"A%d" % k: A, "B%d" % k: B, "Q%d" % k: Q, "R%d" % k: np.array(R), "P%d" % k: np.array(P), "K%d" % k: np.array(K)})
        print(k, A.shape, "max|P| %.4g" % np.abs(P).max(), "K", np.ravel(K)[:3])
    np.savez_compressed(os.path.join(HERE, "dare.npz"), **out)


if __name__ == "__main__":
    main()
