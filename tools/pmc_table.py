"""Formats the per-kernel PMC means of tools/summarize_rocprof.py (pmc) as a table of derived quantities.

Units: SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are summed over waves in quad-cycles (MI355X_MICROARCH.md);
SQ_VALU_MFMA_BUSY_CYCLES in cycles summed over SIMDs; FETCH_SIZE / WRITE_SIZE in KiB (FETCH x2 = gfx950 correction
for 16-byte coalesced reads, an upper bound for mixed-width kernels)."""
import json
import sys

d = json.load(open(sys.argv[1]))
g = lambda k, c: d[k].get(c, {}).get("mean_per_launch", 0.0)
rows = []
for k in d:
    if not any(t in k for t in ("kernel", "Kernel")) or k.startswith("void at::"):
        continue
    gui = g(k, "GRBM_GUI_ACTIVE")
    wc = g(k, "SQ_WAVE_CYCLES")
    if wc == 0:
        continue
    rows.append((gui, k, wc))
rows.sort(reverse=True)
print("| kernel | GUI cycles | waves | wait_any % | wait_inst % | active % | valu % | lds % | vmem % | MFMA util % | VALU insts/wave | LDS insts/wave | VMEM rd/wave | LDS conflict % | FETCH MiB (x2) | WRITE MiB |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
for gui, k, wc in rows:
    w = max(g(k, "SQ_WAVES"), 1.0)
    pct = lambda c: 100.0 * g(k, c) / wc
    mfma = 100.0 * g(k, "SQ_VALU_MFMA_BUSY_CYCLES") / (1024.0 * gui / 8.0) if gui else 0.0
    idx = g(k, "SQ_LDS_IDX_ACTIVE")
    conf = 100.0 * g(k, "SQ_LDS_BANK_CONFLICT") / idx if idx else 0.0
    print("| %s | %.3g | %d | %.0f | %.0f | %.0f | %.0f | %.0f | %.0f | %.1f | %.0f | %.0f | %.0f | %.0f | %.1f (%.1f) | %.1f |" % (
        k[:60], gui, w, pct("SQ_WAIT_ANY"), pct("SQ_WAIT_INST_ANY"), pct("SQ_ACTIVE_INST_ANY"), pct("SQ_ACTIVE_INST_VALU"),
        pct("SQ_ACTIVE_INST_LDS"), pct("SQ_ACTIVE_INST_VMEM"), mfma, g(k, "SQ_INSTS_VALU") / w, g(k, "SQ_INSTS_LDS") / w,
        g(k, "SQ_INSTS_VMEM_RD") / w, conf, g(k, "FETCH_SIZE") / 1024.0, 2 * g(k, "FETCH_SIZE") / 1024.0, g(k, "WRITE_SIZE") / 1024.0))
