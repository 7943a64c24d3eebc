#!/usr/bin/env python3
"""Experiment: the OpenProof cycle of one batch captured once into a HIP graph (S sub-batches on S streams inside the
capture) and replayed, against the plain call sequence.  usage: tools/graph_experiment.py [N] [B] [S ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ring_zk_amd import Context, synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
SS = [int(a) for a in sys.argv[3:]] or [1, 2]
n, k, l = 1, 3, 1
dev = torch.device("cuda", 0)
steps, warm = 300, 150


def run(S, graph, stagger=False):
    ctxs = [Context(N, n, k, l, device=0) for _ in range(S)]
    gk = torch.Generator(device=dev)
    gk.manual_seed(1234)
    A = synth.t_key(gk, N, n, k, l, dev)
    for c in ctxs:
        c.load_key(A)
    g = torch.Generator(device=dev)
    g.manual_seed(1000)
    d = synth.t_challenge(g, B, N, ctxs[0].kappa, dev)
    x = synth.t_uniform(g, (B, l, N), dev)
    r = synth.t_small(g, (B, k, N), dev)
    y = synth.t_gauss(g, (B, k, N), dev, ctxs[0].sigma)
    parts = [slice(i * B // S, (i + 1) * B // S) for i in range(S)]
    side = [torch.cuda.Stream() for _ in range(S)]
    torch.cuda.synchronize()

    def cycle(i):
        p = parts[i]
        if stagger and i % 2:
            z = ctxs[i].open_response(y[p], r[p], d[p])
            c, t, _ = ctxs[i].open_commit(x[p], r[p], y[p])
        else:
            c, t, _ = ctxs[i].open_commit(x[p], r[p], y[p])
            z = ctxs[i].open_response(y[p], r[p], d[p])
        return ctxs[i].open_verify(z, t, c, d[p])

    def step():
        if S == 1:
            return [cycle(0)]
        cur = torch.cuda.current_stream()
        accs = []
        for i in range(S):
            side[i].wait_stream(cur)
            with torch.cuda.stream(side[i]):
                accs.append(cycle(i))
        for i in range(S):
            cur.wait_stream(side[i])
        return accs

    main = torch.cuda.Stream()
    with torch.cuda.stream(main):
        for _ in range(3):
            accs = step()
        torch.cuda.synchronize()
        if graph:
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=main):
                accs = step()
            run_step = gr.replay
        else:
            run_step = step
        for _ in range(warm):
            run_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            run_step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
    ok = sum(int(a.sum()) for a in accs)
    print(f"N={N} B={B} streams={S} graph={graph} stagger={stagger}: {dt * 1e6:8.1f} us per cycle  {B / dt / 1e6:7.3f} M proofs/s  accepted {ok}", flush=True)


def run_fork(graph, prio=False):
    """One batch; commit and response (independent) on two streams, verify after both.  prio: commit / verify on a
    high-priority stream, response on a normal one, so that response workgroups only take the slots commit's retiring
    waves leave."""
    ca, cb = Context(N, n, k, l, device=0), Context(N, n, k, l, device=0)
    gk = torch.Generator(device=dev)
    gk.manual_seed(1234)
    A = synth.t_key(gk, N, n, k, l, dev)
    ca.load_key(A)
    cb.load_key(A)
    g = torch.Generator(device=dev)
    g.manual_seed(1000)
    d = synth.t_challenge(g, B, N, ca.kappa, dev)
    x = synth.t_uniform(g, (B, l, N), dev)
    r = synth.t_small(g, (B, k, N), dev)
    y = synth.t_gauss(g, (B, k, N), dev, ca.sigma)
    side = torch.cuda.Stream(priority=0)
    main = torch.cuda.Stream(priority=-1 if prio else 0)

    def step():
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            z = cb.open_response(y, r, d)
        c, t, _ = ca.open_commit(x, r, y)
        cur.wait_stream(side)
        return ca.open_verify(z, t, c, d)

    with torch.cuda.stream(main):
        for _ in range(3):
            acc = step()
        torch.cuda.synchronize()
        run_step = step
        if graph:
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=main):
                acc = step()
            run_step = gr.replay
        for _ in range(warm):
            run_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            run_step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
    print(f"N={N} B={B} fork commit||response graph={graph} prio={prio}: {dt * 1e6:8.1f} us per cycle  {B / dt / 1e6:7.3f} M proofs/s  accepted {int(acc.sum())}", flush=True)


run_fork(False)
run_fork(False, True)
run_fork(True, True)
for S in SS:
    run(S, False)
    run(S, True)
    if S > 1:
        run(S, True, True)
