"""Schedule simulator for a lagged row-parallel kernel: cell (r,c) is computed at iteration r + lag(c) by lane c & 63.
Finds a lag function (non-decreasing, unit steps) for bands with wide rows and checks: one cell per lane per iteration,
spread of lags among the active lanes, iteration count."""
import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from oracle import mzoracle as mo

def indel_band(rng, rate, mean_len=3.0):
    M = int(rng.integers(900, 1101))
    centre = np.zeros(M + 1, dtype=np.int64); c = 0; i = 1
    while i <= M:
        u = rng.random()
        if u < rate / 2 and i > 1:
            for _ in range(min(int(rng.geometric(1.0 / mean_len)), M - i + 1)):
                centre[i] = c; i += 1
            continue
        if u < rate: c += int(rng.geometric(1.0 / mean_len))
        c += 1; centre[i] = c; i += 1
    N = int(max(c, 11))
    LB = np.minimum(centre, N).astype(np.int32); RB = LB.copy(); LB[0] = 0; RB[M] = N
    return mo.smooth(LB, RB, M, N, 30) + (M, N)

def col_ranges(LB, RB, M, N):
    tlo = np.searchsorted(RB, np.arange(N + 1), side='left')          # first r with RB[r] >= c
    thi = np.searchsorted(LB, np.arange(N + 1), side='right') - 1      # last r with LB[r] <= c
    return tlo, thi

def find_lag(LB, RB, M, N):
    tlo, thi = col_ranges(LB, RB, M, N)
    lag = np.zeros(N + 1, dtype=np.int64)
    for c in range(64, N + 1):
        need = lag[c - 64] + max(0, thi[c - 64] - tlo[c] + 1)        # lane free again: column c-64 done before column c starts
        lag[c] = max(lag[c - 1], need)
    # unit steps: ramp up ahead of jumps (raise columns to the left), then re-propagate the lane constraint
    for _ in range(8):
        changed = False
        for c in range(N, 0, -1):
            if lag[c - 1] < lag[c] - 1: lag[c - 1] = lag[c] - 1; changed = True
        for c in range(64, N + 1):
            need = max(lag[c - 1], lag[c - 64] + max(0, thi[c - 64] - tlo[c] + 1))
            if lag[c] < need: lag[c] = need; changed = True
        if not changed: break
    return lag, tlo, thi

def check(LB, RB, M, N, lag, tlo, thi):
    # lane occupancy intervals
    ok = True; spread = 0
    busy_from = tlo + lag; busy_to = thi + lag
    for c in range(64, N + 1):
        if busy_from[c] <= busy_to[c - 64]: ok = False
    if (np.diff(lag) < 0).any() or (np.diff(lag) > 1).any(): ok = False
    # spread of lags among columns of a row and its neighbours (active window)
    for r in range(0, M + 1, 7):
        spread = max(spread, int(lag[RB[r]] - lag[LB[r]]))
    iters = int(M + lag[N])
    return ok, spread, iters

rng = np.random.default_rng(11)
for rate in (2, 10, 30):
    res = []
    for _ in range(60):
        LB, RB, M, N = indel_band(rng, rate / 1000.0)
        wide = int((RB - LB > 62).sum())
        lag, tlo, thi = find_lag(LB, RB, M, N)
        ok, spread, iters = check(LB, RB, M, N, lag, tlo, thi)
        res.append((ok, spread, iters / M, wide / (M + 1), int(lag[N])))
    res = np.array(res, dtype=float)
    print(f"{rate} events/1000: schedule ok {res[:,0].mean():.2f}, max lag spread in a row {res[:,1].max():.0f} (mean {res[:,1].mean():.1f}), "
          f"iterations / rows {res[:,2].mean():.3f}, wide rows {res[:,3].mean():.2f}, final lag {res[:,4].mean():.1f}")

# ---- the coarser form: one lag per 64-column period (at most three periods in flight, segment heads always on lane 0)
def period_lag(LB, RB, M, N):
    tlo, thi = col_ranges(LB, RB, M, N)
    nper = (N >> 6) + 1
    lam = np.zeros(nper, dtype=np.int64); step = np.zeros(nper, dtype=np.int64)
    for k in range(nper - 1):
        c = np.arange(64 * k, min(64 * k + 64, N + 1 - 64))
        d = int((thi[c] - tlo[c + 64] + 1).max()) if len(c) else 1
        step[k + 1] = max(1, d); lam[k + 1] = lam[k] + step[k + 1]
    return lam, step

print("per-period lags:")
for rate in (2, 10, 30, 60):
    res = []
    for _ in range(80):
        LB, RB, M, N = indel_band(rng, rate / 1000.0)
        lam, step = period_lag(LB, RB, M, N)
        width = int((RB - LB + 1).max())
        cells = int((RB - LB + 1).sum())
        T = M + int(lam[-1])
        elig = step.max() <= 31 and width <= 127 and RB[0] <= 63
        res.append((elig, step.max(), width, T / M, cells / (64.0 * T), cells / (64.0 * M)))
    res = np.array(res, dtype=float)
    print(f"{rate} events/1000: eligible {res[:,0].mean():.2f}, largest step {res[:,1].max():.0f} (mean of max {res[:,1].mean():.1f}), widest row {res[:,2].max():.0f}, "
          f"iterations / rows {res[:,3].mean():.3f}, lanes busy {res[:,4].mean():.3f} (cells / 64 rows {res[:,5].mean():.3f})")

print("per-period lags with a bubble at each re-arm (step >= depth + 1), two consecutive steps within 32:")
for rate in (2, 10, 30, 60):
    res = []
    for _ in range(200):
        LB, RB, M, N = indel_band(rng, rate / 1000.0)
        tlo, thi = col_ranges(LB, RB, M, N)
        nper = (N >> 6) + 1
        step = np.ones(nper, dtype=np.int64)
        for k in range(nper - 1):
            c = np.arange(64 * k, min(64 * k + 64, N + 1 - 64))
            if len(c): step[k + 1] = max(1, int((thi[c] - tlo[c + 64] + 2).max()))
        two = int((step[1:] + step[:-1]).max()) if nper > 1 else 0
        T = M + int(step[1:].sum())
        cells = int((RB - LB + 1).sum())
        res.append((two <= 32 and step.max() <= 31 and (RB - LB).max() <= 126 and RB[0] <= 63, two, T / M, cells / (64.0 * T)))
    res = np.array(res, dtype=float)
    print(f"{rate} events/1000: eligible {res[:,0].mean():.3f}, two steps max {res[:,1].max():.0f} mean {res[:,1].mean():.1f}, iterations / rows {res[:,2].mean():.3f}, lanes busy {res[:,3].mean():.3f}")
