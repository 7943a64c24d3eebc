#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc csv output (counter_collection + kernel_trace) per kernel: mean counter value
per launch (summed over the dispatch's dimensions as rocprofv3 reports them) and mean duration."""
import collections
import csv
import glob
import json
import sys


def short(name):
    return name.split("(")[0].replace("void ", "").replace("rzk::", "")


out = collections.defaultdict(dict)
for d in sys.argv[1:]:
    cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    if not cc:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(cc[0])):
        if "rzk::" not in r["Kernel_Name"]:
            continue
        key = f'{short(r["Kernel_Name"])} grid={r["Grid_Size"]}'
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for key, ctrs in agg.items():
        for c, vals in ctrs.items():
            out[key][c] = sum(vals) / len(vals)
    if kt:
        dur = collections.defaultdict(list)
        for r in csv.DictReader(open(kt[0])):
            if "rzk::" in r["Kernel_Name"]:
                g = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
                dur[f'{short(r["Kernel_Name"])} grid={g}'].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        for key, v in dur.items():
            if key in out:
                out[key].setdefault("dur_us", sum(v) / len(v))
print(json.dumps(out, indent=1))
