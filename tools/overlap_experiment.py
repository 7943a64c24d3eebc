#!/usr/bin/env python3
"""Experiment: run the OpenProof cycle of one 4096-proof batch as S independent sub-batches on S HIP streams,
so that the HBM-bound response rows of one sub-batch overlap the VALU-bound commit / verify rows of another."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ring_zk_amd import Context, synth  # noqa: E402

N, n, k, l, B = 1024, 1, 3, 1, 4096
dev = torch.device("cuda", 0)
steps, warm = 300, 150


def run(S):
    ctxs = [Context(N, n, k, l, device=0) for _ in range(S)]
    gk = torch.Generator(device=dev)
    gk.manual_seed(1234)
    A = synth.t_key(gk, N, n, k, l, dev)
    for c in ctxs:
        c.load_key(A)
    g = torch.Generator(device=dev)
    g.manual_seed(1000)
    sig = ctxs[0].sigma
    d = synth.t_challenge(g, B, N, ctxs[0].kappa, dev)
    x = synth.t_uniform(g, (B, l, N), dev)
    r = synth.t_small(g, (B, k, N), dev)
    y = synth.t_gauss(g, (B, k, N), dev, sig)
    streams = [torch.cuda.Stream() for _ in range(S)]
    parts = [slice(i * B // S, (i + 1) * B // S) for i in range(S)]
    torch.cuda.synchronize()

    def step():
        accs = []
        for slot in range(3):
            for i in range(S):
                # staggered order: odd sub-batches run response before commit (the two are independent), so that an
                # HBM-bound kernel and a VALU-bound one are in flight together from the first launch on
                ph = slot if not (STAGGER and i % 2 and slot < 2) else 1 - slot
                with torch.cuda.stream(streams[i]):
                    p = parts[i]
                    if ph == 0:
                        step.c[i], step.t[i], _ = ctxs[i].open_commit(x[p], r[p], y[p])
                    elif ph == 1:
                        step.z[i] = ctxs[i].open_response(y[p], r[p], d[p])
                    else:
                        accs.append(ctxs[i].open_verify(step.z[i], step.t[i], step.c[i], d[p]))
        return accs

    step.c, step.t, step.z = [None] * S, [None] * S, [None] * S
    for _ in range(warm):
        accs = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        accs = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    ok = sum(int(a.sum()) for a in accs)
    print(f"streams={S}: {dt * 1e6:8.1f} us per {B}-proof cycle  {B / dt / 1e6:7.3f} M proofs/s  accepted {ok}")


STAGGER = False
args = [a for a in sys.argv[1:] if a != "--stagger"]
for STAGGER in ([False, True] if "--stagger" in sys.argv else [False]):
    print("stagger", STAGGER)
    for S in [int(a) for a in args] or [1, 2, 3, 4]:
        run(S)
