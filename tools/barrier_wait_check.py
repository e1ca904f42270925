"""Static check of the compiled kernels (hipcc --save-temps .s files): every s_barrier must be preceded, on EVERY path, by
`s_waitcnt lgkmcnt(0)` after the wave's last LDS store -- hipcc (ROCm 7.2, gfx950) was seen to drop that wait at a loop-header barrier
(a wave-specialised kernel of round 4; common.hpp::wg_sync).  Walks the control-flow graph backwards from each barrier.
    python tools/barrier_wait_check.py /tmp/allasm/*.s"""
import re
import sys

LDS_STORE = re.compile(r"^\s*(ds_write|ds_store|ds_add|ds_max|ds_min|ds_or|ds_and|ds_cmpst|ds_wrxchg|ds_append)")
WAIT0 = re.compile(r"^\s*s_waitcnt\b.*lgkmcnt\(0\)")
BR = re.compile(r"^\s*(s_cbranch_\w+|s_branch)\s+(\.\w+)")


def check(path):
    lines = open(path).read().split("\n")
    out = []
    kernel, kstart = None, 0
    kernels = []
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            kernel, kstart = m.group(1), i
        if l.startswith(".Lfunc_end") and kernel:
            kernels.append((kernel, kstart, i))
            kernel = None
    for name, a, b in kernels:
        body = lines[a:b]
        label_at = {}
        for i, l in enumerate(body):
            m = re.match(r"^(\.\w+):", l)
            if m:
                label_at[m.group(1)] = i
        # predecessors of instruction index i: i-1 (unless i-1 is an unconditional branch / endpgm) + branches targeting a label at i
        preds = {}
        for i, l in enumerate(body):
            m = BR.match(l)
            if m and m.group(2) in label_at:
                preds.setdefault(label_at[m.group(2)], []).append(i)
        def fallthrough_ok(i):        # may control fall from line i to i+1?
            s = body[i].strip()
            return not (s.startswith("s_branch") or s.startswith("s_endpgm") or s.startswith("s_setpc"))
        for i, l in enumerate(body):
            if not re.match(r"^\s*s_barrier", l):
                continue
            seen, stack, bad = set(), [i - 1], None
            while stack and bad is None:
                j = stack.pop()
                while j >= 0:
                    if j in seen:
                        break
                    seen.add(j)
                    s = body[j]
                    if WAIT0.match(s):
                        break
                    if LDS_STORE.match(s):
                        bad = j
                        break
                    for pj in preds.get(j, []):
                        stack.append(pj)
                    if j == 0 or not fallthrough_ok(j - 1):
                        break
                    j -= 1
            if bad is not None:
                out.append((name, a + i + 1, a + bad + 1, body[bad].strip()))
    return out


if __name__ == "__main__":
    tot = 0
    for f in sys.argv[1:]:
        for name, bl, sl, ins in check(f):
            tot += 1
            print("%s: %s: s_barrier at line %d is reachable from the LDS store at line %d (%s) with no s_waitcnt lgkmcnt(0) between" % (f.split("/")[-1], name[:60], bl, sl, ins))
    print("%d unguarded barrier(s)" % tot)
    sys.exit(1 if tot else 0)
