"""Dev tool (CPU): what-if model of the roll-out kernel's scheduling inside one workgroup of 16 trajectories, fed with
measured per-step Newton-solve counts (tools/iters_matrix.py).  Processor sharing per SIMD; compares the lock-step
lift (barrier per step) with dynamically formed groups of four."""
import sys, numpy as np
it = np.load(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/iters_matrix.npz")["iters"].astype(float)
RATE = {0: 0.0, 1: 1.0, 2: 0.85, 3: 0.73, 4: 0.645}
def body_alone(i, base=17.0, per=1.25): return base + per * (max(i, 1) - 1)

def sim_lockstep(itw, lift=8.3, base=17.0):
    # itw: steps x 16
    t = 0.0
    for k in range(itw.shape[0]):
        t += lift
        rem = np.array([body_alone(i, base) for i in itw[k]])
        act = np.ones(16, bool)
        while act.any():
            rates = np.zeros(16)
            for s in range(4):
                idx = [w for w in range(s, 16, 4) if act[w]]
                for w in idx: rates[w] = RATE[len(idx)]
            dt = min(rem[w] / rates[w] for w in range(16) if act[w])
            rem -= rates * dt; t += dt
            act &= rem > 1e-9
    return t

def sim_quads(itw, lift_work=2.0, lift_lat=2.0, base=17.0, G=4):
    S = itw.shape[0]
    k = np.zeros(16, int)            # next step index of each wave
    state = ["lift_wait"] * 16       # lift_wait, lift, lat, body, done
    rem = np.zeros(16); t = 0.0
    waiting = list(range(16))
    lat_end = {}
    def form():
        nonlocal waiting
        active = sum(1 for s in state if s != "done")
        while len(waiting) >= min(G, max(1, active)) and waiting:
            q = waiting[:G]; waiting = waiting[G:]
            for w in q: state[w] = "lift"; rem[w] = lift_work
    form()
    while any(s != "done" for s in state):
        rates = np.zeros(16)
        for s in range(4):
            idx = [w for w in range(s, 16, 4) if state[w] in ("lift", "body")]
            for w in idx: rates[w] = RATE[len(idx)]
        cands = [rem[w] / rates[w] for w in range(16) if rates[w] > 0]
        cands += [lat_end[w] - t for w in lat_end]
        dt = min(cands)
        rem -= rates * dt; t += dt
        for w in range(16):
            if state[w] == "lift" and rem[w] <= 1e-9:
                state[w] = "lat"; lat_end[w] = t + lift_lat
            elif state[w] == "body" and rem[w] <= 1e-9:
                k[w] += 1
                if k[w] >= S: state[w] = "done"
                else: state[w] = "lift_wait"; waiting.append(w)
        for w in list(lat_end):
            if lat_end[w] <= t + 1e-9:
                del lat_end[w]; state[w] = "body"; rem[w] = body_alone(itw[k[w], w], base)
        form()
    return t

for (a, b, name) in ((5, 25, "driver window (steps 5..25)"), (20, 220, "default window (steps 20..220)")):
    ls, qd, q2 = [], [], []
    for wg in range(0, 256, 4):
        itw = it[a:b, wg * 16:(wg + 1) * 16]
        ls.append(sim_lockstep(itw)); qd.append(sim_quads(itw)); q2.append(sim_quads(itw, G=8, lift_work=2.0))
    n = b - a
    print("%s: lockstep mean %.1f max %.1f us/step | quads mean %.1f max %.1f | octets mean %.1f max %.1f" % (
        name, np.mean(ls) / n, np.max(ls) / n, np.mean(qd) / n, np.max(qd) / n, np.mean(q2) / n, np.max(q2) / n))
