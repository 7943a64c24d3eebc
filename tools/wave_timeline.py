#!/usr/bin/env python3
"""Wave timeline of the Open commit launch (diagnostic, GPU box): builds a library with -DRZK_STAMPS=1, runs the
commit phase, reads the per-wave (start, end) s_memrealtime stamps (100 MHz) back and prints how the wave lifetimes
sit inside the launch: start spread, end spread, busy fraction of the wave slots."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib = os.path.join(ROOT, "gpurun_out", "librzk_stamps.so")
os.environ["RZK_LIB"] = lib   # read when ring_zk_amd._lib is imported
os.makedirs(os.path.dirname(lib), exist_ok=True)
from ring_zk_amd import build  # noqa: E402

build.build_library(out=lib, defines=["RZK_STAMPS=1"] + sys.argv[1:])
import torch  # noqa: E402

from ring_zk_amd import Context  # noqa: E402

N, B = 1024, 4096
ctx = Context(N, 1, 3, 1)
ctx.generate_key(1)
half = (ctx.q - 1) // 2
x = ctx.sample_uniform(1, 0, half, (B, 1))
r = ctx.sample_uniform(1, 1, 1, (B, 3))
y = ctx.sample_gauss(1, 2, float(ctx.sigma), (B, 3))
d = ctx.sample_challenge(1, 3, (B,))
what = os.environ.get("RZK_TIMELINE_OP", "open_commit")
a_, b_ = ctx.sample_uniform(1, 7, half, (B,)), ctx.sample_uniform(1, 8, half, (B,))
for _ in range(300):
    if what == "polymul":
        o = ctx.polymul(a_, b_)
    elif what == "matvec1":
        o = ctx.matvec(0, y)
    elif what == "open_verify":
        if _ == 0:
            c, t, ok = ctx.open_commit(x, r, y)
            z = ctx.open_response(y, r, d)
        acc = ctx.open_verify(z, t, c, d)
    else:
        c, t, ok = ctx.open_commit(x, r, y)
torch.cuda.synchronize()
tot = C.c_size_t(0)
ctx._L.rzk_debug_read_scratch(ctx._h, None, 0, C.byref(tot))
buf = np.empty(tot.value // 4, dtype=np.uint32)
ctx._L.rzk_debug_read_scratch(ctx._h, C.c_void_p(buf.ctypes.data), tot.value, None)
stride = 6 * N + 16
lines = buf.reshape(-1, stride)[:, 6 * N:6 * N + 16]
lines = lines[:B]   # one wave per proof: blocks 0..1023
t0 = lines[:, 0].astype(np.uint64) | (lines[:, 1].astype(np.uint64) << 32)
t1 = lines[:, 2].astype(np.uint64) | (lines[:, 3].astype(np.uint64) << 32)
base = t0.min()
s = (t0 - base).astype(np.float64) / 100.0   # us
e = (t1 - base).astype(np.float64) / 100.0
print("waves", len(s), "launch span us", e.max())
print("start us: min %.2f p50 %.2f p99 %.2f max %.2f" % (s.min(), np.median(s), np.percentile(s, 99), s.max()))
print("end   us: min %.2f p10 %.2f p50 %.2f p90 %.2f max %.2f" % (e.min(), np.percentile(e, 10), np.median(e), np.percentile(e, 90), e.max()))
life = e - s
print("life  us: min %.2f p50 %.2f max %.2f ; slot busy fraction %.3f" % (life.min(), np.median(life), life.max(), life.sum() / (len(s) * e.max())))
cyc = lines[:, 8].astype(np.float64)
print("shader clock over the wave lifetimes: median %.3f GHz (min %.3f max %.3f)" % (np.median(cyc / life) / 1e3, (cyc / life).min() / 1e3, (cyc / life).max() / 1e3))
sec = lines[:, 9:15].astype(np.float64)
tot = cyc.mean()
print("wall cycles per wave: lifetime %.0f ; load_lift %.0f (%.0f%%)  wave_fwd %.0f (%.0f%%)  mac %.0f (%.0f%%)  inverse+fold %.0f (%.0f%%)  finish_row %.0f (%.0f%%)  rotation terms %.0f (%.0f%%)" % (
    tot, *sum(([sec[:, i].mean(), 100 * sec[:, i].mean() / tot] for i in range(6)), [])))
hw = lines[:, 4]
xcc = lines[:, 5] & 0xf
cu = (hw >> 8) & 0xf
se = (hw >> 13) & 0x7 if False else (hw >> 13) & 0x3
simd = (hw >> 4) & 0x3
key = xcc.astype(np.int64) * 10000 + ((hw >> 8) & 0xff).astype(np.int64)
uniq, cnt = np.unique(key, return_counts=True)
print("distinct (xcc, cu/sh/se id) %d ; waves per CU: min %d max %d" % (len(uniq), cnt.min(), cnt.max()))
# end time per XCC
for xc in range(8):
    m = xcc == xc
    if m.any():
        print("xcc", xc, "waves", int(m.sum()), "end p50 %.1f max %.1f" % (np.median(e[m]), e[m].max()))
# per SIMD: the end times of its waves, sorted (a few examples and the mean profile)
sk = key * 4 + simd.astype(np.int64)
prof = []
shown = 0
for k in np.unique(sk):
    m = sk == k
    ends = np.sort(e[m])
    if len(ends) == 4:
        prof.append(ends)
    if shown < 6:
        print("simd", int(k), "waves", int(m.sum()), "ends", np.round(ends, 1).tolist())
        shown += 1
if prof:
    print("mean sorted end times over", len(prof), "SIMDs with 4 waves:", np.round(np.mean(prof, axis=0), 1).tolist())
