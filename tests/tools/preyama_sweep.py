"""Differential sweep of mz_preyama_batch() (block text in, block text out; one-stage and two-stage merges on the device)
against the oracle's restatement of pre_yama: random block pairs, 1..10 rows, dash-heavy, short and long overlaps, radii
5..60.    python tests/tools/preyama_sweep.py <seed0> <seed1>"""
import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import inputs
import multiz_amd as mz
from oracle import mzoracle as mo
from test_preyama import _prejob, _assemble, same_block
mz.api.init(0)
tot = bad = nulls = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(424_000 + seed)
    jobs, meta = [], []
    while len(jobs) < 300:
        n1, n2 = int(rng.integers(1, 11)), int(rng.integers(2, 11))
        a1, a2, beg, end = inputs.random_block_pair(rng, n1, n2, int(rng.integers(40, 1200)))
        if end - beg < 3:
            continue
        R = int(rng.choice([5, 15, 30, 60]))
        v = int(rng.integers(0, 2))
        if v == 0 and n1 < 2:
            continue
        if rng.random() < 0.15:                                   # a row of nothing but dashes under the overlap
            blk = a1 if rng.random() < 0.5 else a2
            k = int(rng.integers(1, len(blk.rows))) if len(blk.rows) > 1 else 0
            if k:
                blk.rows[k].text = "-" * blk.textSize
        try:
            want, _ = mo.pre_yama(a1, a2, beg, end, R, v)
        except (RuntimeError, IndexError):
            continue
        j, cb1, cb2 = _prejob(a1, a2, beg, end, R)
        jobs.append(j + (v,)); meta.append((a1, cb1, a2, cb2, want, v))
    res = mz.preyama_batch(jobs)
    for i, (r, (a1, cb1, a2, cb2, want, v)) in enumerate(zip(res, meta)):
        tot += 1
        nulls += want is None
        got = _assemble(r, a1, cb1, a2, cb2) if r["status"] == 0 else "status %d stage %d" % (r["status"], r["stage"])
        if not (r["status"] == 0 and same_block(got, want)):
            bad += 1
            print("BAD seed", seed, "job", i, "v", v, "rows", len(a1.rows), len(a2.rows), "null", r["null_code"], got if isinstance(got, str) else "", flush=True)
print("merges", tot, "NULL results", nulls, "bad", bad)
