"""Fold the rocprofv3 output of tools/pmc_round.sh into profiles/: per kernel CLASS of the XL/2 step the average PMC values
per launch (MFMA busy cycles, wave cycles, waits, GRBM_GUI_ACTIVE, FETCH_SIZE x 2 per the guide's gfx950 correction,
WRITE_SIZE) -> profiles/<tag>_pmc.json (read by bench.py for `traffic`), and the kernel-stats CSVs copied next to it."""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r2"
OUT = os.path.join(ROOT, "gpurun_out")
PROF = os.path.join(ROOT, "profiles")
B = 8
M = B * 16 * 256
D, HM = 1152, 4608


def classify(name):
    n = name.replace(" ", "")
    if "gemm_pps_kernel<256,0," in n:
        return "gemm_qkv"
    if "gemm_pps_kernel<256,1," in n:
        return "gemm_fc1"
    if "gemm_pps_kernel<192,2," in n:
        return "gemm_fc2" if n.split(">(")[0].endswith(",1") else "gemm_proj"
    if "gemm_pwr_kernel<2," in n or "gemm_pw_kernel<2," in n:   # 12-wave kernels: <EPI, DT, TAG[, LO, ABL]> (round 6 added LO / ABL)
        args = n.split("_kernel<")[1].split(">(")[0].split(",")
        return "gemm_fc2" if args[2] == "1" else "gemm_proj"
    if "qkv_attn_kernel<" in n:   # fused QKV projection + attention: <HD, DT, MODE, FLAGS>
        return "qkv_attn_temporal" if n.split("qkv_attn_kernel<")[1].split(",")[2] == "1" else "qkv_attn_spatial"
    if "attn_full_kernel" in n:
        return "attn_spatial"
    if "attn_small_kernel" in n:
        return "attn_temporal"
    if "ln_modulate_kernel" in n:
        return "ln_modulate"
    if "conv3x3_kernel" in n or "conv3x3_pp_kernel" in n:
        return "vae_conv3x3"
    if "gn_partial_kernel" in n:
        return "vae_groupnorm_stats"
    if "gn_apply_kernel" in n:
        return "vae_groupnorm_apply"
    return None


ALG = {  # algorithmic bytes per launch: operands read once + outputs written once
    "gemm_qkv": M * D * 2 + 3 * D * D * 2 + M * 3 * D * 2,
    "gemm_proj": M * D * 2 + D * D * 2 + 2 * M * D * 4,
    "gemm_fc1": M * D * 2 + HM * D * 2 + M * HM * 2,
    "gemm_fc2": M * HM * 2 + D * HM * 2 + 2 * M * D * 4,
    "attn_spatial": M * 4 * D * 2, "attn_temporal": M * 4 * D * 2, "ln_modulate": M * D * 6,
    # fused kernel: xn read once, W_qkv once, the attention output written once (q / k / v never reach HBM)
    "qkv_attn_spatial": M * D * 2 + 3 * D * D * 2 + M * D * 2, "qkv_attn_temporal": M * D * 2 + 3 * D * D * 2 + M * D * 2,
}
acc = defaultdict(lambda: defaultdict(list))
names = {}
for d in sorted(glob.glob(os.path.join(OUT, f"{TAG}_pmc*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                c = classify(row["Kernel_Name"])
                if c is None:
                    continue
                names[c] = row["Kernel_Name"]
                acc[c][row["Counter_Name"]].append((int(row["Dispatch_Id"]), float(row["Counter_Value"]),
                                                    int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
res = {}
VAE_ALG = None
try:
    sys.path.insert(0, ROOT)
    import bench as _bench
    _fl, _gs, _ga = _bench.vae_decoder_work(32)
    VAE_ALG = {"vae_conv3x3": ("flops", 16 * _fl), "vae_groupnorm_stats": ("bytes", 16 * _gs), "vae_groupnorm_apply": ("bytes", 16 * _ga)}
except Exception as exc:   # torch-free environments: the VAE rows are skipped
    print("vae rows skipped:", exc)
for c in [k for k in acc if k.startswith("vae_")]:
    ctrs = acc.pop(c)
    rec = {"kernel": names[c], "unit": "one 16-frame decode (second of two; sums over its launches)", "counters_sum_per_decode": {}, "launches_per_decode": {}}
    for k, vals in ctrs.items():
        vals.sort()
        use = vals[len(vals) // 2:]
        rec["counters_sum_per_decode"][k] = sum(v for _, v, _ in use)
        rec["launches_per_decode"][k] = len(use)
        if k == "GRBM_GUI_ACTIVE":
            rec["ns_per_decode_in_pmc_pass"] = sum(t for _, _, t in use)
    cs = rec["counters_sum_per_decode"]
    if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
        rec["hbm_bytes_per_decode"] = int(2 * cs["FETCH_SIZE"] * 1024 + cs["WRITE_SIZE"] * 1024)
        if VAE_ALG and VAE_ALG[c][0] == "bytes":
            rec["algorithmic_bytes_per_decode"] = VAE_ALG[c][1]
            rec["traffic_over_algorithmic"] = round(rec["hbm_bytes_per_decode"] / VAE_ALG[c][1], 3)
    if VAE_ALG and VAE_ALG[c][0] == "flops":
        rec["algorithmic_flops_per_decode"] = VAE_ALG[c][1]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in cs and cs.get("GRBM_GUI_ACTIVE", 0) > 0:
        rec["mfma_busy_frac_of_simd_cycles"] = round(cs["SQ_VALU_MFMA_BUSY_CYCLES"] / (cs["GRBM_GUI_ACTIVE"] / 8 * 1024), 4)
    if cs.get("SQ_WAVE_CYCLES", 0) > 0 and "SQ_WAIT_ANY" in cs:
        rec["SQ_WAIT_ANY_frac_of_wave_cycles"] = round(cs["SQ_WAIT_ANY"] / cs["SQ_WAVE_CYCLES"], 4)
    res[f"{c}:frames=16"] = rec
for c, ctrs in acc.items():
    rec = {"kernel": names[c], "M": M, "algorithmic_bytes_per_launch": ALG[c], "counters_avg_per_launch": {}, "launches_sampled": {}}
    for k, vals in ctrs.items():
        vals.sort()
        skip = len(vals) // 2 if len(vals) >= 4 else 0      # first forward = warm-up (the workload runs >= 2)
        use = vals[skip:]
        rec["counters_avg_per_launch"][k] = sum(v for _, v, _ in use) / len(use)
        rec["launches_sampled"][k] = len(use)
        if k == "GRBM_GUI_ACTIVE":
            rec["avg_ns_in_pmc_pass"] = sum(t for _, _, t in use) / len(use)
    ca = rec["counters_avg_per_launch"]
    if "FETCH_SIZE" in ca and "WRITE_SIZE" in ca:
        rec["FETCH_SIZE_KB_raw"], rec["WRITE_SIZE_KB_raw"] = ca["FETCH_SIZE"], ca["WRITE_SIZE"]
        rec["hbm_bytes_per_launch"] = int(2 * ca["FETCH_SIZE"] * 1024 + ca["WRITE_SIZE"] * 1024)
        rec["traffic_over_algorithmic"] = round(rec["hbm_bytes_per_launch"] / ALG[c], 3)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in ca and "GRBM_GUI_ACTIVE" in ca and ca["GRBM_GUI_ACTIVE"] > 0:
        # SQ_VALU_MFMA_BUSY_CYCLES is summed over the chip's 1024 SIMDs (16 cycles per v_mfma_f32_16x16x32: it equals
        # 16 x the launch's MFMA count, checked against 2 M N K / 16384); GRBM_GUI_ACTIVE is summed over the 8 XCDs
        # (value / launch time = 8 x the shader clock), so elapsed cycles per SIMD = GRBM_GUI_ACTIVE / 8
        rec["mfma_busy_frac_of_simd_cycles"] = round(ca["SQ_VALU_MFMA_BUSY_CYCLES"] / (ca["GRBM_GUI_ACTIVE"] / 8 * 1024), 4)
        # (rounds 2-4 also printed GRBM_GUI_ACTIVE / 8 / launch time as a "shader clock": for launches of tens of microseconds the
        #  counter covers more than the kernel's own interval and the quotient came out above the 2.4 GHz maximum -- dropped; the clock
        #  of a launch is what rocm-smi samples beside it, bench.py: power_check)
    if "SQ_WAVE_CYCLES" in ca and ca["SQ_WAVE_CYCLES"] > 0:
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if k in ca:
                rec[k + "_frac_of_wave_cycles"] = round(ca[k] / ca["SQ_WAVE_CYCLES"], 4)
    rec["note"] = ("rocprofv3 --pmc passes over tools/pmc_workload.py (XL/2 forward, B = 8, in-model cache state), separate passes for "
                   "the SQ / FETCH_SIZE / WRITE_SIZE groups; FETCH_SIZE doubled (MI355X_MICROARCH.md: gfx950 reports half the bytes "
                   "of 16 B/lane streams); FETCH/WRITE count L2 fabric requests, Infinity-Cache hits included")
    res[f"{c}:M={M}"] = rec
os.makedirs(PROF, exist_ok=True)
with open(os.path.join(PROF, f"{TAG}_pmc.json"), "w") as f:
    json.dump(res, f, indent=1)
print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk not in ("note", "kernel")} for k, v in res.items()}, indent=1))
for sub, dst in ((f"{TAG}_stats_bench", f"{TAG}_kernel_stats_bench_B8_steps20.csv"), (f"{TAG}_stats_fwd_vae", f"{TAG}_kernel_stats_forward_B8_plus_vae_decode.csv")):
    hits = glob.glob(os.path.join(OUT, sub, "**", "*kernel_stats.csv"), recursive=True)
    if hits:
        shutil.copy(hits[0], os.path.join(PROF, dst))
        print("copied", hits[0], "->", dst)
bj = os.path.join(OUT, f"{TAG}_stats_bench.json")
if os.path.exists(bj) and os.path.getsize(bj) > 0:
    shutil.copy(bj, os.path.join(PROF, f"{TAG}_bench_under_rocprof_B8_steps20.json"))
