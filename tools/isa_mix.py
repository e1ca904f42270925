"""Static instruction mix of every kernel in a HIP source of lsdm_amd/csrc (hipcc -S for gfx950; no GPU needed).

    python tools/isa_mix.py encoder_u.hip [name-filter]

Per kernel: MFMA, VALU (by opcode), LDS, VMEM and SALU instruction counts of the whole body (loops counted once) -- the
quantity the dense kernels are bound by next to the matrix pipe."""
import collections
import os
import re
import subprocess
import sys

root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lsdm_amd", "csrc")
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = "/tmp/isa_mix_%s.s" % os.path.basename(src)
extra = ["-ffp-contract=off"] if src == "geometry.hip" else []
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only", src, "-o", out] + extra,
               cwd=root, check=True, stderr=subprocess.DEVNULL)
s = open(out).read()
for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)s_endpgm', s, re.S | re.M):
    name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
    if flt not in name:
        continue
    ops, valu = collections.Counter(), collections.Counter()
    for line in m.group(2).split('\n'):
        t = line.strip().split()
        if not t or t[0][0] in '.;/' or t[0].endswith(':'):
            continue
        op = t[0]
        if op.startswith('v_mfma'): ops['mfma'] += 1
        elif op.startswith('v_'): ops['valu'] += 1; valu[op.replace('_e32', '').replace('_e64', '')] += 1
        elif op.startswith('ds_'): ops['lds'] += 1
        elif op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): ops['vmem'] += 1
        elif op.startswith('s_'): ops['salu'] += 1
    print("%s\n   %s\n   valu: %s" % (name[:150], dict(ops), ", ".join("%s %d" % kv for kv in valu.most_common(12))))
