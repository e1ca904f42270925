"""profiles/traffic.json from the PMC pass summaries (tools/pmc_passes.sh -> gpurun_out/pmc_<tag>.json).

    python tools/make_traffic.py bf16x3=gpurun_out/pmc_x3.json f32=gpurun_out/pmc_f32.json ...

Per precision and bench kernel name: HBM-side bytes per launch from FETCH_SIZE / WRITE_SIZE (KiB in the raw files).
`fetch_bytes_x2` applies the gfx950 correction for 16-byte coalesced reads (MI355X_MICROARCH.md, HBM section: FETCH_SIZE
reports half of a wide streaming read); these kernels mix access widths, so the truth lies between the two figures."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SA = {"3": "sa_kernel<3,32,32,64>", "64": "sa_kernel<64,64,64,128>", "128": "sa_kernel<128,128,128,256>", "256": "sa_kernel<256,256,256,512>"}
FP = {("256", "512"): "fp_kernel<fp4>", ("128", "256"): "fp_kernel<fp3>", ("64", "256"): "fp_kernel<fp2>", ("0", "128"): "fp_kernel<fp1+head>"}


def bench_name(k):
    m = re.search(r"sa_(?:[hrsu]_)?kernel<(\d+),", k)
    if m:
        return SA.get(m.group(1))
    m = re.search(r"fp_(?:h_)?kernel<(\d+), (\d+),", k)
    if m:
        return FP.get((m.group(1), m.group(2)))
    m = re.search(r"fp_pre_kernel<(\d+),", k)
    if m:
        return {"64": "fp_kernel<fp2>", "128": "fp_kernel<fp3>"}.get(m.group(1))
    if "fps_chain" in k:
        return "fps_chain_kernel"
    for n in ("fps_chain_kernel", "posa_kernel", "cond_kernel", "scene_attn_kernel"):
        if n in k:
            return n
    if "denoise" in k:
        return "denoise_kernel"
    if "upsample12" in k:
        return "upsample12_kernel"
    return None


out = {}
for arg in sys.argv[1:]:
    prec, path = arg.split("=")
    d = json.load(open(path))
    out[prec] = {}
    for k, cs in d.items():
        n = bench_name(k)
        if not n or "FETCH_SIZE" not in cs:
            continue
        f = cs["FETCH_SIZE"]["mean_per_launch"] * 1024.0
        w = cs.get("WRITE_SIZE", {}).get("mean_per_launch", 0.0) * 1024.0
        out[prec][n] = {"fetch_bytes": f, "fetch_bytes_x2": 2 * f, "write_bytes": w, "unit": "bytes per launch (B=64)"}
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1)[:1500])
