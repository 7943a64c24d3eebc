#!/usr/bin/env python3
"""Time rzk_open_commit_batch_dev alone for a key shape (shared-operand path study)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ring_zk_amd import Context, synth

N, n, k, l, B = (int(v) for v in sys.argv[1:6])
ctx = Context(N, n, k, l)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
ctx.load_key(synth.t_key(g, N, n, k, l, dev))
x = synth.t_uniform(g, (B, l, N), dev); r = synth.t_small(g, (B, k, N), dev); y = synth.t_gauss(g, (B, k, N), dev, ctx.sigma)
for _ in range(2): ctx.open_commit(x, r, y)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): c, t, ok = ctx.open_commit(x, r, y)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print(f"N={N} ({n},{k},{l}) B={B} share_min={os.environ.get('RZK_SLOT_SHARE_MIN','default')}: open_commit {dt*1e3:.3f} ms  ({B/dt:.4g} commits/s) ok={int(ok.sum())}")
