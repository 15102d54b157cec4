"""Where do the waves of a launch of few long pairs run, and when?  (HISTORY 9.12 / section 10: the serial C5 DP takes 25.4 or 30.2 ms.)
    python tests/tools/c5_where.py [pairs, default 1000] [columns, default 100000]            (on the GPU box)
MZ_DP_STAMP (include/mz_amd.h) makes k_dp_row_lat leave, per pair, the HW_ID / XCC_ID of its wave and s_memrealtime (100 MHz) at its start
and end.  Serial launches (mz_dev_run) and pipelined ones (mz_dev_run_async): per launch its span, the waves' own lifetimes, and how the
waves were dealt over the SIMDs -- SIMDs holding 0 / 1 / 2 / 3+ waves of the launch -- with the lifetimes of waves alone on their SIMD and
of waves that share one."""
import sys, numpy as np
sys.path.insert(0, '.')
import multiz_amd as mz
from multiz_amd import synth, api
import torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
cols = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
mz.api.init(0)
batch = synth.make_batch(n, 2, 2, cols * 95 // 100, cols * 105 // 100, 30)
db = mz.DevBatch(batch)
MZ_DP_STAMP = 0x20000
def report(d, label):
    torch.cuda.synchronize()
    a = d._view(d.c.scanAux, 3 * n, np.int64).reshape(n, 3)
    hw, t0, t1 = a[:, 0], a[:, 1], a[:, 2]
    simd, cu, sh, se, xcc = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7, (hw >> 32) & 15
    key = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
    life = (t1 - t0) / 1e5                                     # ms
    uniq, inv, cnt = np.unique(key, return_inverse=True, return_counts=True)
    per = cnt[inv]
    cus = len(np.unique(key >> 2))
    hist = np.bincount(cnt, minlength=5)
    s = f"{label}: span {(t1.max() - t0.min()) / 1e5:6.2f} ms, starts within {(t0.max() - t0.min()) / 1e5:5.2f} ms; {cus} CUs, SIMDs with 1 / 2 / 3 / 4+ waves: {hist[1]} / {hist[2]} / {hist[3]} / {hist[4:].sum()}"
    for k in (1, 2, 3):
        if (per == k).any():
            s += f"; waves {k} to a SIMD live {life[per == k].mean():5.2f} ms (max {life[per == k].max():5.2f})"
    print(s, flush=True)
db.c.dp_hint |= MZ_DP_STAMP
for i in range(6):
    ms = db.run(timed=True)
    report(db, f"serial    {i} (DP {ms[1]:6.2f} ms by HIP events)")
ring = [db] + [db.alternate() for _ in range(2)]
for d in ring:
    d.c.dp_hint |= MZ_DP_STAMP
import os
for i in range(6):
    ring[i % 3].run_async()
    db.wait(); torch.cuda.synchronize()
    report(ring[i % 3], f"pipelined {i} (one at a time)")
for rnd in range(2):
    for i in range(3):
        ring[i].run_async()
    db.wait(); torch.cuda.synchronize()
    for i in range(3):
        report(ring[i], f"pipelined, three in flight, round {rnd} batch {i}")
