#!/usr/bin/env python3
"""Static check of the generated gfx950 code: a translation unit whose kernels issue matrix instructions (v_mfma_*) must not contain
packed fp32 arithmetic (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32).  Round 4 found a v_pk_fma_f32 with cross operand selection losing
its update in a gather wave that shares a SIMD with an MFMA wave; the library is therefore compiled with
-target-feature -packed-fp32-ops for those units (csrc/Makefile: NOPK) and this check keeps it that way.

    python tools/packed_fp32_check.py /tmp/lsdm_asm_check/*-gfx950.s        (make -C lsdm_amd/csrc check-packed)

Exit status 1 and a per-kernel listing when a unit mixes the two."""
import re
import sys

PK = re.compile(r"^\s*v_pk_(mul|fma|add)_f32\b")
MFMA = re.compile(r"^\s*v_mfma_")
LABEL = re.compile(r"^([A-Za-z_][\w$.]*):")


def scan(path):
    """-> (number of mfma instructions, {kernel or function label: packed fp32 instructions})"""
    mfma, pk, cur = 0, {}, "?"
    with open(path) as fh:
        for line in fh:
            m = LABEL.match(line)
            if m and not m.group(1).startswith(".L"):
                cur = m.group(1)
            if MFMA.match(line):
                mfma += 1
            elif PK.match(line):
                pk[cur] = pk.get(cur, 0) + 1
    return mfma, pk


def main(paths):
    bad = 0
    for p in paths:
        mfma, pk = scan(p)
        n = sum(pk.values())
        if mfma and n:
            bad += 1
            print("%s: %d v_mfma next to %d packed fp32 instructions" % (p, mfma, n))
            for k, v in sorted(pk.items(), key=lambda kv: -kv[1])[:12]:
                print("    %6d  %s" % (v, k))
    print("%d unit(s) mix matrix instructions with packed fp32 arithmetic" % bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
