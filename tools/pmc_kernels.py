#!/usr/bin/env python3
"""HBM traffic per kernel from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; csv) of one bench.py configuration,
as /opt/skills/guides/MI355X_MICROARCH.md §HBM prescribes: separate passes (TCC slots); values are KiB; on gfx950
FETCH_SIZE reports half of the bytes of a wide coalesced streaming read, so it is calibrated on a kernel of the SAME
run whose byte count is known exactly (ntt_fwd_kernel: polys x N x 4 bytes read; bench.py runs it at the end of every
configuration); WRITE_SIZE needs no correction (checked on the same kernel).

usage: tools/pmc_kernels.py <fetch_dir> <write_dir> <bench.json> <workload-key> [out.json]
bench.json: the JSON line of the same configuration (for the algorithmic bytes per launch the library reports).
"""
import collections
import csv
import glob
import json
import re
import sys


def norm(name):
    name = name.split("(")[0].replace("void ", "").replace("rzk::", "")
    name = name.replace(", WaveTeam>", ">")
    return name


def load(d, counter):
    path = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and "rzk::" in r["Kernel_Name"]:
            agg[norm(r["Kernel_Name"])].append(float(r["Counter_Value"]) * 1024.0)
    return {k: sum(v) / len(v) for k, v in agg.items()}, {k: len(v) for k, v in agg.items()}


def main():
    fetch, nf = load(sys.argv[1], "FETCH_SIZE")
    write, _ = load(sys.argv[2], "WRITE_SIZE")
    bench = None
    for line in open(sys.argv[3]):
        if line.startswith("{"):
            bench = json.loads(line)
    ntt = [k for k in fetch if k.startswith("ntt_fwd_kernel")]
    N = int(re.search(r"N=(\d+)", bench["config"]["workload"]).group(1))
    polys = bench["ntt_roofline"]["polys"]
    cal = (polys * N * 4) / fetch[ntt[0]]
    wr_check = write[ntt[0]] / (polys * N * 4)
    kernels = {}
    alg = bench["roofline"]["kernels"]
    for k in sorted(fetch, key=lambda k: -fetch[k] * nf[k]):
        if k.startswith(("ntt_", "sample_", "fill_", "key_")):
            continue
        e = {"launches_profiled": nf[k], "fetch_bytes_raw": fetch[k], "fetch_bytes_corrected": fetch[k] * cal,
             "write_bytes": write.get(k, 0.0), "hbm_bytes_per_launch": fetch[k] * cal + write.get(k, 0.0)}
        a = alg.get(k)
        if a:
            e["algorithmic_bytes_per_launch"] = a["algorithmic_bytes_per_launch"]
            e["traffic_over_algorithmic"] = e["hbm_bytes_per_launch"] / a["algorithmic_bytes_per_launch"]
            e["avg_launch_us_unprofiled"] = a["avg_launch_us"]
            e["hbm_side_TBps"] = e["hbm_bytes_per_launch"] / (a["avg_launch_us"] * 1e-6) / 1e12
        kernels[k] = e
    out = {"workload": sys.argv[4], "config": bench["config"]["workload"],
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) on bench.py --steps 3",
           "fetch_calibration": cal, "fetch_calibration_from": ntt[0] + f" ({polys} polys x {N} x 4 B read)",
           "write_size_over_known_bytes": wr_check, "kernels": kernels}
    text = json.dumps(out, indent=1)
    print(text)
    if len(sys.argv) > 5:
        open(sys.argv[5], "w").write(text + "\n")


if __name__ == "__main__":
    main()
