#!/usr/bin/env python3
"""bench.py — headline benchmark of the ring-zk MI355X backend.

Metric (BASELINE.json): proofs/sec for the full OpenProof cycle commit -> challenge -> response ->
verify at N = 1024, (n,k,l) = (1,3,1), batch = 4096 independent proofs per GPU, plus the achieved
HBM GB/s of the kernels against the chip's roofline.

A "step" is one pass of the hot path over one batch: rzk_open_commit_batch_dev,
rzk_open_response_batch_dev, rzk_open_verify_batch_dev on inputs already resident in HBM.  The
random inputs (r, y and the challenge d) are pre-sampled like the data itself: the reference draws them with the
host RNG and does no ring arithmetic in generate_challenge (src/prove/open.rs:143-158), so no
sampler is inside the timed region (config.inputs says so; the device samplers are timed by tools/bench_phases.py).

  python bench.py --gpus N --steps K --warmup W
N > 1 is launched by the driver through torch.distributed.run (one rank per GPU, RCCL only for the
barrier / key broadcast / result reduction: proofs are independent, the batch is simply split, weak scaling).

Other BASELINE configurations: --workload linear|sum --N --shape n,k,l --summands V --batch B [--chunk C].
With --chunk the batch is processed in sub-batches of C proofs whose inputs are re-drawn by the device samplers for
every chunk INSIDE the timed region (config 5: 4096 proofs x ~45 MB of live data do not fit in HBM at once).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
BUTTERFLY_PEAK = 4.7e12      # measured chip-wide rate of the 32-bit Montgomery butterfly at 4 waves per SIMD
                             # (tools/microbench/valu_rates.cpp, profiles/r01_valu_rates.txt): the VALU ceiling


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--N", type=int, default=1024)
    ap.add_argument("--batch", type=int, default=4096, help="proofs per GPU and step")
    ap.add_argument("--chunk", type=int, default=0, help="process the batch in sub-batches of this many proofs, "
                    "inputs re-drawn per chunk by the device samplers inside the timed region (0 = whole batch resident)")
    ap.add_argument("--workload", choices=["open", "linear", "sum"], default="open",
                    help="open = BASELINE metric config; linear / sum = the other BASELINE configs")
    ap.add_argument("--shape", type=str, default="1,3,1", help="n,k,l")
    ap.add_argument("--summands", type=int, default=8, help="V for --workload sum")
    ap.add_argument("--ramp", type=int, default=-1,
                    help="untimed steps run once before the warmup: the GPU needs ~20 ms of load to reach steady "
                         "clocks (measured: 337 us/step right after start-up vs 288 us/step once warm); -1 = adaptive: "
                         "groups of untimed steps until two consecutive groups agree to 1 %% (0.25 s .. 3 s of work)")
    ap.add_argument("--broadcast-key", type=int, default=1,
                    help="N > 1: rank 0 generates the key and broadcasts the [a1;a2] slab (RCCL) instead of every rank "
                         "regenerating it from the seed")
    ap.add_argument("--extra-steps", type=int, default=-1,
                    help="steps of each of the two secondary timed regions (randomness drawn inside the step; trusted-producer "
                         "mode): -1 = min(steps, 100), 0 = skip them")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=25.0, help="target length of the CPU baseline sample")
    return ap.parse_args()


# ---- accounting ---------------------------------------------------------------------------------------------------
def cycle_polys(workload, n, k, l, V):
    """Polynomials a cycle moves at the boundary, key resident: everything a phase takes in or hands out."""
    if workload == "open":      # 7+3, 7+3, 6 = 26 at (1,3,1)
        return {"commit": (l + 2 * k) + (2 * n + l), "response": (2 * k + 1) + k, "verify": k + n + l + 1}   # z, t, c1, d
    if workload == "linear":
        return {"commit": (1 + l + 4 * k) + (4 * n + 3 * l), "response": (4 * k + 1) + 2 * k,
                "verify": 2 * k + 2 * (n + l) + 1 + 2 * n + l + 1}
    return {"commit": (V + V * l + 2 * V * k + 2 * k) + ((V + 1) * (n + l) + (V + 1) * n + l),
            "response": (2 * V * k + 2 * k + 1) + (V + 1) * k,
            "verify": (V + 1) * k + (V + 1) * (n + l) + V + (V + 1) * n + l + 1}


def ring_multiplies(workload, n, k, l, V):
    """Ring multiplies of the reference per cycle (Mat::dot multiplies identity / zero blocks too)."""
    if workload == "open":
        return (2 * n + l) * k + k + (n * k + n)
    if workload == "linear":
        return (2 * l + 4 * n * k + 4 * l * k) + 2 * k + (2 * n * k + 2 * n + 2 * l * k + 3 * l)
    return (2 * V * l + (V + 1) * k * (2 * n + 2 * l)) + (V + 1) * k + ((V + 1) * (n * k + n + l * k) + 2 * V * l + l)


def min_transform_units(workload, n, k, l, V, rot):
    """Transform units (one length-N transform over one 30-bit prime) a cycle needs at least with this design: every
    distinct operand transformed once per prime and program, every output row transformed back once per prime.
    Key products with ternary / Gaussian operands need 2 primes, products of two full-range polynomials 3, the
    response rows and the challenge products of the verifiers none when they run as rotations (rot: N <= 1024).
    a1' has ca = k-n random columns, a2' has cb = k-n-l (a subset of the same operands)."""
    ca, cb = k - n, k - n - l
    commit_one = 2 * (ca + n + l) + 2 * (ca + n)           # c = [a1;a2].r + [0;x] and t = a1.y
    rel_one = 2 * (ca + n) + (0 if rot else 2 * (n + 1))   # a1.z - c1 (.) d - t == 0
    resp_one = 0 if rot else 1 + 2 * k                     # z = y + r (.) d  (one prime)
    a2_one = 2 * (cb + l)                                  # a2 . v as a program of its own
    if workload == "open":
        return commit_one + resp_one + rel_one
    if workload == "linear":
        commit = 3 * (2 * l + 1) + 2 * commit_one + 2 * l + 3 * (2 * l + 1 + cb)   # gx; two commits with t, t'; a2.y; u
        verify = 2 * rel_one + 2 * l + 3 * (2 * l + 1) + 3 * (2 * l + 1 + cb) + (0 if rot else 3 * (l + 1))
        return commit + 2 * resp_one + verify
    vec = 3 * (V * l + V + l)                              # sum_i x_i (.) g_i over l rows
    vec_key = 3 * (V * l + V + cb + l)                     # sum_i w_i (.) g_i - a2 . v'
    commit = vec + (V + 1) * commit_one + V * a2_one + vec_key
    verify = (V + 1) * rel_one + V * a2_one + vec + vec_key + (0 if rot else 3 * (l + 1))
    return commit + (V + 1) * resp_one + verify


# ---- CPU baseline ---------------------------------------------------------------------------------------------------
def cpu_ntt(N, p, psi, seconds=2.0):
    """The same transform (Cooley-Tukey, table of psi powers in bit-reversed order, one 30-bit prime) on the host cores
    (oracle/rzk_oracle.c, rzko_ntt_forward_batch): BASELINE.md §3.2's "same-algorithm CPU" figure next to the kernel's."""
    import numpy as np

    from oracle import oracle as O

    threads = O.hw_threads()
    psi_n = psi   # primitive 2N-th root of unity mod p (rzk_ntt_psi)
    assert O.powmod(psi_n, N, p) == p - 1
    rng = np.random.default_rng(7)
    cnt = 64 * threads
    a = rng.integers(0, p, (cnt, N), dtype=np.uint32)
    t0 = time.perf_counter()
    O.ntt_forward_batch(a, p, psi_n, threads, inplace=True)
    dt = time.perf_counter() - t0
    cnt = int(max(cnt, min(cnt * seconds / max(dt, 1e-6), 1 << 20)))
    a = rng.integers(0, p, (cnt, N), dtype=np.uint32)
    t0 = time.perf_counter()
    O.ntt_forward_batch(a, p, psi_n, threads, inplace=True)
    dt = time.perf_counter() - t0
    return {"value": cnt * 2 * N * 4 / dt / 1e9, "unit": "GB/s", "polys_per_s": cnt / dt, "cores": threads, "kind": "port",
            "sample": f"{cnt} residue polynomials of {N} coefficients in place, {dt:.2f} s"}


def cpu_baseline(workload, N, n, k, l, V, seconds):
    """CPU restatement (oracle, schoolbook multiply, literal Mat::dot) timed on the host cores."""
    import numpy as np

    from oracle import oracle as O
    from ring_zk_amd import synth

    P = O.Params(N=N, n=n, k=k, l=l)
    threads = O.hw_threads()
    rng = np.random.default_rng(99)
    A = synth.key(rng, N, n, k, l)
    if workload == "open":
        def sample(B):
            x = synth.uniform(rng, (B, l, N))
            r = synth.small(rng, (B, k, N))
            y = synth.gauss(rng, (B, k, N), P.sigma)
            d = synth.challenge(rng, (B,), N, P.kappa)
            t0 = time.perf_counter()
            acc = O.open_cycle_batch(P, A, x, r, y, d, threads)
            dt = time.perf_counter() - t0
            assert acc == B, f"CPU baseline: {acc}/{B} proofs accepted"
            return dt

        probe = max(threads, 8)
        dt = sample(probe)
        rate = probe / dt
        B = int(max(probe, min(rate * seconds, 65536)))
        B -= B % threads or 0
        B = max(B, threads)
        dt = sample(B)
        how = "OpenMP over proofs"
    else:
        # Linear / Sum: single proofs one after the other; at N >= 1024 the oracle spreads every ring multiply over the
        # host threads (rzko_poly_mul), below that it runs on one core
        def one():
            g = synth.uniform(rng, (max(V, 1), N))
            xs = synth.uniform(rng, (max(V, 1), l, N))
            rs, rp = synth.small(rng, (max(V, 1), k, N)), synth.small(rng, (k, N))
            ys, yp = synth.gauss(rng, (max(V, 1), k, N), P.sigma), synth.gauss(rng, (k, N), P.sigma)
            d = synth.challenge(rng, (1,), N, P.kappa)[0]
            t0 = time.perf_counter()
            if workload == "linear":
                c, cp, t, tp, u, ok = O.linear_commit(P, A, g[0], xs[0], rs[0], rp, ys[0], yp)
                z, zp = O.linear_response(P, ys[0], yp, rs[0], rp, d)
                acc = O.linear_verify(P, A, z, zp, c, cp, g[0], t, tp, u, d)
            else:
                cs, cp, ts, tp, u, ok = O.sum_commit(P, A, g, xs, rs, rp, ys, yp)
                zs, zp = O.sum_response(P, ys, yp, rs, rp, d)
                acc = O.sum_verify(P, A, zs, zp, cs, cp, g, ts, tp, u, d)
            dt = time.perf_counter() - t0
            assert acc == 1, "CPU baseline: proof rejected"
            return dt

        dt1 = one()
        B = int(max(1, min(seconds / max(dt1, 1e-6), 4096)))
        dt = sum(one() for _ in range(B))
        how = "proofs one after the other, ring multiplies spread over the host threads" if N >= 1024 else "one core"
        if N < 1024:
            threads = 1
    return {
        "value": B / dt,
        "unit": "proofs/s",
        "cores": threads,
        "kind": "port",
        "sample": f"{B} {workload.capitalize()}Proof cycles (commit+response+verify), N={N}, (n,k,l)=({n},{k},{l})"
                  + (f", V={V}" if workload == "sum" else "")
                  + f", schoolbook CPU restatement (oracle/rzk_oracle.c), {how}, {dt:.1f} s",
    }


def self_launch(ngpus):
    """`python bench.py --gpus N` (N > 1) started WITHOUT torch.distributed.run: this process becomes a plain parent that
    starts the N ranks as a fresh child (`python -m torch.distributed.run ... bench.py <same args>`), relays the
    child's output (rank 0 prints the JSON line) and exits with the child's status.  The parent never imports torch
    or touches a HIP device: a process that has initialised the GPU must not start GPU children on this pool."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ngpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this pool (RCCL needs it)
    env["RZK_BENCH_SELF_LAUNCHED"] = "1"
    print(f"bench.py: --gpus {ngpus} without a launcher: starting {ngpus} ranks through torch.distributed.run "
          f"(127.0.0.1:{port})", file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if os.environ.get("RZK_BENCH_JOIN_ONLY"):
        # CPU-tier rehearsal of the launch path (tests/test_bench_launch.py): every rank joins the process group,
        # takes part in one reduction and leaves — no GPU needed, nothing measured
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(os.environ.get("RZK_BENCH_BACKEND", "gloo"), rank=rank, world_size=world)
        from ring_zk_amd import shard

        el, tot, per_rank = shard.reduce_result(dist, 1.0 + rank, rank + 1, torch.device("cpu"), gather=True)
        if rank == 0:
            print(json.dumps({"joined": world, "max_elapsed": el, "sum": tot, "per_rank": per_rank,
                              "self_launched": bool(os.environ.get("RZK_BENCH_SELF_LAUNCHED"))}), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    # RZK_BENCH_BACKEND=gloo: rehearsal of the multi-rank path on a box with fewer GPUs than ranks (ranks
    # share devices, reductions run on CPU tensors).  The real multi-GPU run uses nccl (= RCCL over xGMI).
    backend = os.environ.get("RZK_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    dev_index = local_rank if backend == "nccl" else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from ring_zk_amd import Context, shard, synth

    N = args.N
    n, k, l = (int(v) for v in args.shape.split(","))
    V = args.summands if args.workload == "sum" else 1
    B = args.batch
    chunk = args.chunk if 0 < args.chunk < B else 0
    Bc = chunk or B                      # proofs resident at a time
    nchunks = (B + Bc - 1) // Bc
    assert B % Bc == 0, "--chunk must divide --batch"
    # --ramp -1 (default): adaptive, see ramp_until_steady below
    ramp = args.ramp if args.ramp >= 0 else 0
    dev = torch.device("cuda", dev_index)
    red_dev = dev if backend == "nccl" else torch.device("cpu")   # where the result reduction runs
    ctx = Context(N, n, k, l, device=dev_index)
    sig = ctx.sigma

    # ---- commitment key: same on every rank.  Rank 0 draws it and broadcasts the dense [a1;a2] slab (48 KiB at
    # (1,3,1) N=1024, 4.25 MiB at config 5: SURVEY §8e); --broadcast-key 0 regenerates it from the seed per rank.
    gk = torch.Generator(device=dev)
    gk.manual_seed(1234)
    A = synth.t_key(gk, N, n, k, l, dev)
    if world > 1 and args.broadcast_key:
        A = shard.broadcast_key(dist, A if rank == 0 else torch.zeros_like(A), red_dev)
    ctx.load_key(A)
    # per-proof inputs from the library's device-side samplers (counter-based, seed recorded in the output):
    # the reference's distributions: x, g uniform over Z_q; r uniform in [-b, b]; y = (i64) N(0, sigma);
    # d with kappa coefficients +-1
    seed = 1000 + rank
    half = (ctx.q - 1) // 2

    def draw(chunk_index):
        sid = iter(range(64 * chunk_index, 64 * chunk_index + 64))
        uni = lambda *lead: ctx.sample_uniform(seed, next(sid), half, lead)
        small = lambda *lead: ctx.sample_uniform(seed, next(sid), ctx.b, lead)
        gauss = lambda *lead: ctx.sample_gauss(seed, next(sid), float(sig), lead)
        d = ctx.sample_challenge(seed, next(sid), (Bc,))
        if args.workload == "open":
            return dict(d=d, x=uni(Bc, l), r=small(Bc, k), y=gauss(Bc, k))
        if args.workload == "linear":
            return dict(d=d, g=uni(Bc), x=uni(Bc, l), r=small(Bc, k), rp=small(Bc, k), y=gauss(Bc, k), yp=gauss(Bc, k))
        return dict(d=d, gs=uni(Bc, V), xs=uni(Bc, V, l), rs=small(Bc, V, k), rp=small(Bc, k), ys=gauss(Bc, V, k),
                    yp=gauss(Bc, k))

    def phases(I):
        if args.workload == "open":
            c, t, ok = ctx.open_commit(I["x"], I["r"], I["y"])
            yield "commit"
            z = ctx.open_response(I["y"], I["r"], I["d"])
            yield "response"
            acc = ctx.open_verify(z, t, c, I["d"])
            yield "verify"
        elif args.workload == "linear":
            c, cp, t, tp, u, ok = ctx.linear_commit(I["g"], I["x"], I["r"], I["rp"], I["y"], I["yp"])
            yield "commit"
            z, zp = ctx.linear_response(I["y"], I["yp"], I["r"], I["rp"], I["d"])
            yield "response"
            acc = ctx.linear_verify(z, zp, c, cp, I["g"], t, tp, u, I["d"])
            yield "verify"
            ok = (ok == 3).to(torch.uint8)
        else:
            cs, cp, ts, tp, u, ok = ctx.sum_commit(I["gs"], I["xs"], I["rs"], I["rp"], I["ys"], I["yp"])
            yield "commit"
            zs, zp = ctx.sum_response(I["ys"], I["yp"], I["rs"], I["rp"], I["d"])
            yield "response"
            acc = ctx.sum_verify(zs, zp, cs, cp, I["gs"], ts, tp, u, I["d"])
            yield "verify"
        phases.result = (ok, acc)

    resident = None if chunk else draw(0)
    torch.cuda.synchronize()
    accepted = [0, 0]   # accept / commit-ok counts of the last step

    sid_fresh = [10_000]

    def redraw_randomness(I):
        """The draws the reference's benches time with the phases (benches/bench.rs:41-47: commit() with its r loop,
        commit.rs:98-107; the prover's Gaussian y, open.rs:88-94; generate_challenge's d, open.rs:143-158): fresh
        r, y (and r', y' ...) and d from the device samplers, the message x / g stays."""
        J = dict(I)
        sid = sid_fresh[0]
        sid_fresh[0] += 16
        for i, key in enumerate(sorted(I)):
            if key in ("r", "rp", "rs"):
                J[key] = ctx.sample_uniform(seed, sid + i, ctx.b, I[key].shape[:-1])
            elif key in ("y", "yp", "ys"):
                J[key] = ctx.sample_gauss(seed, sid + i, float(sig), I[key].shape[:-1])
            elif key == "d":
                J[key] = ctx.sample_challenge(seed, sid + i, I[key].shape[:-1])
        return J

    def step(marks=None, count=False, sample=False):
        """One pass over the batch.  marks: list collecting launch-count marks per phase (profiled steps);
        count: also add up the verdict flags of every chunk (done for the last timed step only: two small torch
        reductions per chunk that are not part of the path); sample: draw the step's randomness inside the step."""
        tot_ok = tot_acc = None
        for ci in range(nchunks):
            I = resident if resident is not None else draw(ci)
            if sample and resident is not None:
                I = redraw_randomness(I)
            if marks is not None:
                m = [ctx.prof_count()]
                for _name in phases(I):
                    m.append(ctx.prof_count())
                marks.append(m)
            else:
                for _name in phases(I):
                    pass
            if count:
                ok, acc = phases.result
                tot_ok = ok.sum() if tot_ok is None else tot_ok + ok.sum()
                tot_acc = acc.sum() if tot_acc is None else tot_acc + acc.sum()
        return tot_ok, tot_acc

    def barrier():
        shard.barrier(dist, dev)

    # HIP events (created without the system-scope fence) bracket the row-kernel launches of every 8th timed
    # step, on the stream the kernels run on; they are read back after the timed region.  Bracketing every
    # launch costs 4 % of throughput (a ~5 us bubble per event pair) without changing the kernel durations
    # (equal to rocprofv3's either way); every 8th step keeps that below 1 %.
    # RZK_BENCH_PROF=0 times the loop without any event and profiles in a separate pass.
    prof_live = rank == 0 and os.environ.get("RZK_BENCH_PROF", "1") != "0"
    prof_every = 8
    # the untimed steps also size the library's event pool (hipEventCreate is slow): as many profiled steps as the
    # timed region will hold
    nprof_steps = (args.steps + prof_every - 1) // prof_every if prof_live else 0
    # (1) everything the timed region will use for the first time — the library's event pool (hipEventCreate), torch's
    # reduction kernels and .item() of the verdict count — runs before the ramp: kernels launched within ~50 steps after
    # the first profiled steps were measured 15-20 % slow (commit 157 us against 133 us), whatever came before them
    for i in range(nprof_steps):
        ctx.prof_enable(True)
        step()
        ctx.prof_enable(False)
    w_ok, w_acc = step(count=True)
    assert int(w_ok.item()) >= 0 and int(w_acc.item()) >= 0
    # (2) ramp
    ramp_info = None
    if args.ramp < 0:
        # The first process on a fresh box runs its first few hundred steps 15-20 % slow (157 us commit launches against
        # 133 us; measured: 11.8 M proofs/s with 100 untimed steps, 13.7 M with 400, 13.9 M with 1500), whatever the
        # length of the timed region.  Untimed groups of steps run until two consecutive groups agree to 1 % (at least
        # 0.25 s of work, at most 3 s), so that a short timed region (--steps 20) measures the steady state too.
        torch.cuda.synchronize()
        g0 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        one = time.perf_counter() - g0
        group = max(1, min(200, int(0.05 / max(one, 1e-5))))   # ~50 ms of work per group
        times, total = [], one
        while total < 3.0:
            torch.cuda.synchronize()
            g0 = time.perf_counter()
            for _ in range(group):
                step()
            torch.cuda.synchronize()
            times.append((time.perf_counter() - g0) / group)
            total += times[-1] * group
            steady = len(times) >= 3 and abs(times[-1] - times[-2]) <= 0.01 * times[-2] and \
                abs(times[-2] - times[-3]) <= 0.01 * times[-3]
            if steady and total >= 0.25:
                break
        ramp_info = {"groups": len(times), "steps_per_group": group, "first_ms": (times[0] if times else one) * 1e3,
                     "last_ms": (times[-1] if times else one) * 1e3}
    # (3) the plain warmup steps the caller asked for
    for _ in range(ramp + args.warmup):
        step()
    barrier()
    marks = []   # (launch count before the chunk, after commit, after response, after verify)
    if prof_live:
        ctx.prof_reset()
    t0 = time.perf_counter()
    for i in range(args.steps):
        final = i + 1 == args.steps
        if prof_live and i % prof_every == 0:
            ctx.prof_enable(True)
            res = step(marks, count=final)
            ctx.prof_enable(False)
        else:
            res = step(count=final)
        if final:
            ok_s, acc_s = res
    barrier()
    elapsed = time.perf_counter() - t0

    accepted, ok_cnt = int(acc_s.item()), int(ok_s.item())
    # (RZK_BENCH_DIAG=1: timing-only runs of the diagnostic library builds of DESIGN.md §6, whose results are wrong on purpose)
    diag = os.environ.get("RZK_BENCH_DIAG", "0") == "1"
    assert diag or (accepted == B and ok_cnt == B), f"rank {rank}: {accepted}/{B} accepted, {ok_cnt}/{B} commit-ok"

    elapsed, tot_acc, per_rank = shard.reduce_result(dist, elapsed, accepted, red_dev, gather=True)
    proofs = B * world * args.steps
    value = proofs / elapsed

    # ---- secondary timed regions (same bracket: barrier + synchronize on both sides, max over ranks) ----------------
    # (a) the step's randomness drawn INSIDE the step, as the reference's criterion benches do; (b) trusted-producer
    # mode (rzk_ctx_trust_device_outputs): the cycle's operands are the library's own device outputs, the per-
    # coefficient range test is skipped.  `value` stays the checked, pre-sampled number.
    extra = args.extra_steps if args.extra_steps >= 0 else min(args.steps, 100)
    secondary = {}
    if extra > 0:
        def timed(nsteps, **kw):
            for _ in range(max(2, nsteps // 10)):
                step(**kw)
            barrier()
            t1 = time.perf_counter()
            for i in range(nsteps):
                res = step(count=(i + 1 == nsteps), **kw)
            barrier()
            el = time.perf_counter() - t1
            ok2, acc2 = res
            return el, int(acc2.item()), int(ok2.item())

        if resident is not None:
            el, acc2, ok2 = timed(extra, sample=True)
            # fresh r can fail check_commit_constraint only with negligible probability at b = 1; every proof whose
            # commit was ok must verify
            assert acc2 == ok2, f"rank {rank}: sampled cycle: {acc2} accepted of {ok2} commit-ok"
            el, acc_sum, _pr = shard.reduce_result(dist, el, acc2, red_dev, gather=True)
            secondary["value_with_sampling"] = B * world * extra / el
            secondary["with_sampling"] = {"steps": extra, "ms_per_step": el / extra * 1e3, "accepted_last_step": acc_sum,
                                          "drawn_in_step": "r, y (prover randomness) and d (challenge) by rzk_sample_*_dev; x resident"}
        ctx.trust_device_outputs(True)
        el, acc2, ok2 = timed(extra)
        ctx.trust_device_outputs(False)
        assert acc2 == B and ok2 == B
        el, _a, _pr = shard.reduce_result(dist, el, acc2, red_dev, gather=True)
        secondary["value_trusted"] = B * world * extra / el
        secondary["trusted"] = {"steps": extra, "ms_per_step": el / extra * 1e3,
                                "mode": "rzk_ctx_trust_device_outputs(1): canonical test of loaded coefficients skipped"}

    roofline = None
    ntt = None
    units = None
    if rank == 0:
        if not prof_live:   # same measurement in a separate pass
            ctx.prof_enable(True)
            ctx.prof_reset()
            for _ in range(max(3, min(args.steps, 10))):
                step(marks)
        durs = ctx.prof_read_all()
        kinfo = ctx.prof_read_kernels()
        ctx.prof_enable(False)
        launches = len(durs)
        # per kernel (template instance): launches, time and algorithmic bytes as the library reports them per launch
        per_kernel = {}
        for (kname, kbytes), dur in zip(kinfo, durs):
            e = per_kernel.setdefault(kname, {"launches": 0, "us": 0.0, "bytes": 0})
            e["launches"] += 1
            e["us"] += dur
            e["bytes"] += kbytes
        nprof = max(len(marks), 1)
        # per phase: the durations of its row-program launches, averaged over the profiled chunks
        names = ("commit", "response", "verify")
        phase_launch_us = {name: [0.0] * (marks[0][j + 1] - marks[0][j]) for j, name in enumerate(names)} if marks else {}
        for m in marks:
            for j, name in enumerate(names):
                for li, dur in enumerate(durs[m[j]:m[j + 1]]):
                    phase_launch_us[name][li] += dur / nprof
        phase_us = {name: sum(v) for name, v in phase_launch_us.items()}
        polys = cycle_polys(args.workload, n, k, l, V)
        # algorithmic bytes at the boundary, key resident: every polynomial a phase takes in or hands
        # out, 8*N bytes each (Open at (1,3,1): 10 + 10 + 6 = 26 polynomials)
        phase_bytes = {name: polys[name] * 8 * N * Bc for name in names}
        cycle_bytes = sum(phase_bytes.values())
        cycle_us = sum(phase_us.values())
        # the dominant kernel = the kernel (template instance) with the largest share of the profiled time; `frac` is
        # ITS algorithmic bytes over ITS time (both summed over its launches: the library reports, per launch, which
        # kernel ran and 8 N x (distinct polynomials read + rows stored) x batch entries)
        dom_name = max(per_kernel, key=lambda nm: per_kernel[nm]["us"]) if per_kernel else None
        dom = per_kernel.get(dom_name, {"launches": 1, "us": 1.0, "bytes": 0})
        dom_phase = max(names, key=lambda nm: phase_us.get(nm, 0.0))
        dom_time = dom["us"] / max(dom["launches"], 1)
        dom_bytes = dom["bytes"] / max(dom["launches"], 1)
        achieved = dom["bytes"] / (dom["us"] * 1e-6) / 1e9
        total_prof_us = sum(e["us"] for e in per_kernel.values()) or 1.0
        kernels_out = {nm: {"launches_per_step": e["launches"] / nprof, "avg_launch_us": e["us"] / e["launches"],
                            "share_of_kernel_time": e["us"] / total_prof_us,
                            "algorithmic_bytes_per_launch": e["bytes"] / e["launches"],
                            "frac_of_hbm_peak": e["bytes"] / (e["us"] * 1e-6) / 1e9 / HBM_PEAK_GBS}
                       for nm, e in sorted(per_kernel.items(), key=lambda kv: -kv[1]["us"])}
        traffic, traffic_src = None, None
        # HBM bytes per launch of the dominant kernel: from the committed PMC passes of this same configuration
        # (profiles/r03_<config>_pmc_traffic.json, tools/collect_evidence.sh), matched by workload key and kernel name
        import glob
        key = f"{args.workload}-{N}-{args.shape}"
        for pmc_path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r03_*pmc_traffic.json"))):
            try:
                pj = json.load(open(pmc_path))
            except Exception:
                continue
            ent = pj.get("kernels", {}).get(dom_name) if pj.get("workload") == key else None
            if ent and ent.get("hbm_bytes_per_launch"):
                traffic = ent.get("hbm_bytes_per_launch")
                traffic_src = f"imported from profiles/{os.path.basename(pmc_path)} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes " \
                              "of this same command, corrected as the guide prescribes; not measured in this run)"
                break
        roofline = {
            "kernel": dom_name,
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "traffic_source": traffic_src,
            "avg_launch_us": dom_time,
            "algorithmic_bytes_per_launch": dom_bytes,
            "scope": "the dominant kernel's own launches (algorithmic bytes and HIP-event durations of exactly those)",
            "kernels": kernels_out,
            "phase_us": phase_us,
            "phase_frac": {nm: (phase_bytes[nm] / (phase_us[nm] * 1e-6) / 1e9) / HBM_PEAK_GBS if phase_us.get(nm) else None
                           for nm in names},
            "cycle": {"us": cycle_us, "algorithmic_bytes": cycle_bytes,
                      "achieved": cycle_bytes / (cycle_us * 1e-6) / 1e9 if cycle_us else None,
                      "frac": cycle_bytes / (cycle_us * 1e-6) / 1e9 / HBM_PEAK_GBS if cycle_us else None},
            "launches_timed": int(launches),
        }
        rot = True   # challenge products as rotations at every ring degree (N = 2048: two-wavefront teams, round 3)
        upp = min_transform_units(args.workload, n, k, l, V, rot)
        bpu = (N // 2) * (N.bit_length() - 1)
        brate = upp * bpu * value / max(world, 1)
        units = {"ring_multiplies_per_proof_reference": ring_multiplies(args.workload, n, k, l, V),
                 "transform_units_per_proof_min": upp, "butterflies_per_unit": bpu,
                 "butterflies_per_s_per_gpu": brate, "butterfly_peak": BUTTERFLY_PEAK,
                 "frac_of_butterfly_peak": brate / BUTTERFLY_PEAK,
                 "note": "minimum transform units of the design (operands once per prime and program, rows once per prime; "
                         "rotations instead of transforms for challenge products) x (N/2) log2 N butterflies, "
                         "against the measured chip-wide butterfly rate (integer-VALU ceiling)"}
        # stand-alone batched forward NTT (one residue polynomial = 2*N*4 algorithmic bytes)
        # 65536 x 4 KiB in + the same out = 512 MiB per launch: beyond the 256 MiB Infinity Cache
        if N >= 512:
            cnt = 16 * 4096
            xin = torch.randint(0, ctx.ntt_prime(0), (cnt, N), dtype=torch.int32, device=dev)
            xout = torch.empty_like(xin)
            ntt_us = ctx.bench_ntt_forward(0, xin, xout, 20)
            ntt_gbs = cnt * 2 * N * 4 / (ntt_us * 1e-6) / 1e9
            ntt = {"kernel": f"ntt_fwd_kernel<{N.bit_length() - 1}>", "polys": cnt, "avg_launch_us": ntt_us, "achieved": ntt_gbs,
                   "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ntt_gbs / HBM_PEAK_GBS}
            if not args.no_cpu_baseline and world == 1:
                ntt["cpu_same_algorithm"] = cpu_ntt(N, ctx.ntt_prime(0), ctx.ntt_psi(0))

    cpu = None
    cpu_reason = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.workload, N, n, k, l, V, args.cpu_seconds)
    elif world > 1:
        cpu_reason = "timed on rank 0 at N=1 only (the host cores are shared by all ranks of a multi-GPU run)"
    else:
        cpu_reason = "--no-cpu-baseline"

    if rank == 0:
        out = {
            "metric": "proofs/sec (commit+challenge+response+verify), N=1024, batch=4096; NTT GB/s vs HBM peak",
            "value": value,
            **{k_: v_ for k_, v_ in secondary.items() if k_.startswith("value_")},
            "unit": "proofs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ramp": ramp_info if ramp_info is not None else {"steps": ramp},
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic" if not diag else "synthetic; DIAGNOSTIC run (RZK_BENCH_DIAG=1): results not checked, not a measurement of the product",
            "config": {
                "workload": f"{args.workload.capitalize()}Proof cycle, N={N}, (n,k,l)=({n},{k},{l}), kappa=36, "
                            + (f"V={V} summands, " if args.workload == "sum" else "") + f"batch={B} proofs per GPU"
                            + (f", processed in {nchunks} chunks of {Bc} proofs" if chunk else ""),
                "arithmetic": "u32 residues of up to three 30-bit NTT primes, exact CRT to the centred residue mod q; "
                              "int64 coefficients at the boundary, every one tested for the centred range while it is loaded",
                "inputs": f"device-side samplers (Philox4x32-10), seed {1000}+rank: x uniform over Z_q, r in [-b,b], "
                          "y=(i64)N(0,sigma), d with kappa +-1; "
                          + ("re-drawn per chunk INSIDE the timed region (the whole batch does not fit in HBM)" if chunk else
                             "pre-sampled and resident in HBM before the timed region: no sampler (r, y, challenge d) is timed"),
                "challenge": "pre-sampled like every random input (generate_challenge does no ring arithmetic, open.rs:143-158)",
                "parallelism": f"batch split over {world} GPU(s), no data-path collective; key "
                               + ("broadcast from rank 0" if world > 1 and args.broadcast_key else "generated from the same seed on every rank"),
                "accepted": tot_acc,
                "accepted_per_rank": per_rank,
                "secondary": {k_: v_ for k_, v_ in secondary.items() if not k_.startswith("value_")},
                "values": "value = every coefficient range-checked, randomness pre-sampled (the headline); value_with_sampling = "
                          "r, y and d drawn by the device samplers inside every timed step (what benches/bench.rs times); "
                          "value_trusted = trusted-producer mode, randomness pre-sampled",
            },
            "roofline": roofline,
            "units": units,
            "ntt_roofline": ntt,
            "cpu_baseline": cpu,
            "cpu_baseline_skipped": cpu_reason,
        }
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
