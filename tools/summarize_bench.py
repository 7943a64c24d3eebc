#!/usr/bin/env python3
import json, sys
for line in open(sys.argv[1]):
    line = line.strip()
    if line.startswith("##"):
        print(line)
    elif line.startswith("{"):
        j = json.loads(line)
        r = j["roofline"]
        print("   %.4g proofs/s  %.3f ms/step  row avg %.1f us  alg %.0f GB/s frac %.3f  phases %s" % (
            j["value"], j["ms_per_step"], r["avg_launch_us"], r["achieved"], r["frac"],
            {k: round(v, 1) for k, v in r["phase_us"].items()}))
    elif line:
        print(line[:300])
