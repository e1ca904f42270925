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
# bench stage -> the launches it consists of in the bf16x3 schedule (substring match on the rocprof kernel name); a stage's traffic
# is the SUM over its launches (both are inside the HIP-event interval bench.py reports for the stage)
STAGES = {
    "sa_kernel<3,32,32,64>": ["sa_r_kernel<3, 32, 32, 64", "sa_h_kernel<3, 32, 32, 64", "sa_kernel<3, 32, 32, 64"],
    "sa_kernel<64,64,64,128>": ["sa_ur_kernel<64, 64, 128", "sa_r_kernel<64, 64, 64, 128", "sa_h_kernel<64, 64, 64, 128"],
    "sa_kernel<128,128,128,256>": ["lin_u_kernel<128, 128, true>", "sa_p_kernel<128, 128, 256", "sa_pair_kernel", "sa_u2_kernel<128, 128, 256", "sa_u_kernel<128, 128, 256", "sa_s_kernel<128",
                                   "sa_h_kernel<128, 128, 128, 256"],
    "sa_kernel<256,256,256,512>": ["lin_u_kernel<256, 256, true>", "sa_w_kernel", "sa_x_kernel", "sa_h_kernel<256, 256, 256, 512"],
    "fp_kernel<fp4>": ["lin_u_kernel<512, 256, false>", "fp_pre_kernel<256, 256, 256", "fp_h_kernel<256, 512"],
    "fp_kernel<fp3>": ["fp_pre_kernel<128, 256, 256", "fp_h_kernel<128, 256"],
    "fp_kernel<fp2>": ["fp_d_kernel", "fp_ws_kernel", "fp_rw_kernel", "fp_pre_kernel<64, 256, 128", "fp_ps_kernel", "fp_h_kernel<64, 256"],
    "fp_kernel<fp1+head>": ["fp_c_kernel", "fp_s_kernel", "fp_h_kernel<0, 128"],
    "fps_chain_kernel": ["fps_chain"],
    "denoise_kernel": ["denoise_r_kernel", "denoise_h_kernel", "denoise_kernel"],
    "upsample12_kernel": ["upsample12"],
    "cond_kernel": ["cond_kernel"],
    "scene_attn_kernel": ["scene_attn_kernel"],
    "posa_kernel": ["posa_stage_kernel", "posa_kernel"],
}

out = {}
for arg in sys.argv[1:]:
    prec, path = arg.split("=")
    d = json.load(open(path))
    out[prec] = {}
    for stage, pats in STAGES.items():
        f = w = 0.0
        parts = []
        for k, cs in d.items():
            if "FETCH_SIZE" not in cs or not any(pt in k for pt in pats):
                continue
            f += cs["FETCH_SIZE"]["mean_per_launch"] * 1024.0
            w += cs.get("WRITE_SIZE", {}).get("mean_per_launch", 0.0) * 1024.0
            parts.append(k.replace("void ", "").split("(")[0][:60])
        if parts:
            out[prec][stage] = {"fetch_bytes": f, "fetch_bytes_x2": 2 * f, "write_bytes": w, "unit": "bytes per launch (B=64)", "launches": parts}
    out[prec]["_source"] = os.path.basename(path)
    out[prec]["_kernels"] = sorted({p for st in out[prec].values() if isinstance(st, dict) for p in st["launches"]})
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1)[:1800])
