#!/usr/bin/env python3
"""Bank conflicts of the wave NTT's LDS transpositions (ring_zk_amd/csrc/rzk_core.h, Geo / LdsMap).

Model (MI355X_MICROARCH.md, "LDS"): ds_read_b32 / ds_write_b32 serve a wave64 access as two groups of 32 lanes;
within a group, lanes hitting the same bank ((byte address / 4) mod 32) at DIFFERENT words serialise.  For each
ring degree and each of the three register layouts the script prints the worst multiplicity over all register
indices, for the padded addressing  word(j) = j + (j >> 5)  the kernels use and for the unpadded  word(j) = j.

usage: tools/lds_conflicts.py [pad_shift ...]      (default pad shift 5)
"""
import sys


def layouts(logn):
    N, LE = 1 << logn, logn - 6
    E, LOSH = 1 << LE, 6 - LE
    hi_low = logn <= 10

    def p2_hi(lane):
        return (lane & (E - 1)) if hi_low else (lane >> LOSH)

    def p2_lo(lane):
        return (lane >> LE) if hi_low else (lane & ((1 << LOSH) - 1))

    return N, E, {
        "phase 1 (e*64 + lane)": lambda lane, r: r * 64 + lane,
        "phase 2 (hi*64 + (r<<LOSH) + lo)": lambda lane, r: p2_hi(lane) * 64 + (r << LOSH) + p2_lo(lane),
        "phase 3 (lane*E + c)": lambda lane, r: lane * E + r,
    }


def worst(E, index, word):
    w = 1
    for r in range(E):
        for group in (range(0, 32), range(32, 64)):
            banks = {}
            for lane in group:
                a = word(index(lane, r))
                banks.setdefault(a % 32, set()).add(a)
            w = max(w, max(len(v) for v in banks.values()))
    return w


def main():
    shifts = [int(a) for a in sys.argv[1:]] or [5]
    for logn in (9, 10, 11):
        N, E, pats = layouts(logn)
        print(f"N = {N} (E = {E} coefficients per lane)")
        for name, index in pats.items():
            cols = [f"unpadded {worst(E, index, lambda j: j)}-way"]
            for sh in shifts:
                cols.append(f"pad j>>{sh}: {worst(E, index, lambda j, sh=sh: j + (j >> sh))}-way")
            print(f"   {name:36s} " + "   ".join(cols))


if __name__ == "__main__":
    main()
