"""Copy the judged profile summaries from gpurun_out/ (scratch) into profiles/ (tracked).
    python tools/summarize_profiles.py r01
Produces profiles/<round>_bench_kernel_stats.csv (rocprofv3 --kernel-trace --stats of `bench.py`),
profiles/<round>_bench_line.json (the JSON line printed under the profiler) and
profiles/<round>_pmc_summary.md (derived metrics from the separate --pmc passes)."""
import collections
import csv
import hashlib
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
os.makedirs(P, exist_ok=True)
if os.path.exists(os.path.join(G, "prof", "trace_kernel_stats.csv")):
    shutil.copy(os.path.join(G, "prof", "trace_kernel_stats.csv"), os.path.join(P, f"{tag}_bench_kernel_stats.csv"))
    shutil.copy(os.path.join(G, "prof_bench.json"), os.path.join(P, f"{tag}_bench_line_under_profiler.json"))
if os.path.exists(os.path.join(G, "bench.json")):
    shutil.copy(os.path.join(G, "bench.json"), os.path.join(P, f"{tag}_bench_line.json"))
if os.path.exists(os.path.join(G, "tune_conv.json")):
    shutil.copy(os.path.join(G, "tune_conv.json"), os.path.join(P, f"{tag}_conv_layer_sweep.json"))
if os.path.exists(os.path.join(G, "box.txt")):
    shutil.copy(os.path.join(G, "box.txt"), os.path.join(P, f"{tag}_box.txt"))

# everything else the round's DESIGN.md / README cite, copied under the round's tag when the GPU run produced it
for src, dst in (("bench_432x768.json", "bench_line_432x768.json"), ("bench_bf16.json", "bench_line_bf16_corr.json"),
                 ("bench_torchrun1.json", "bench_line_torchrun_1rank.json"), ("conv_algo_sweep.txt", "conv_algo_sweep.txt"),
                 ("busy_probe.txt", "busy_probe.txt"), ("corr_roofline_probe.txt", "corr_roofline_probe.txt"),
                 ("refs_chain_probe.txt", "refs_chain_probe.txt"), ("training_side_probe.txt", "training_side_probe.txt"),
                 ("tail_probe.txt", "tail_probe.txt"), ("test_report.txt", "test_report_parity.txt"),
                 (os.path.join("prof_bf16", "trace_kernel_stats.csv"), "bench_bf16_kernel_stats.csv"),
                 (os.path.join("corrprof", "t_kernel_stats.csv"), "corr_probe_kernel_stats.csv"),
                 (os.path.join("tailprof", "t_kernel_stats.csv"), "tail_kernel_stats.csv"),
                 ("tail_insitu_probe.txt", "tail_insitu_probe.txt"), ("gemm_lib_probe.txt", "gemm_lib_probe.txt"),
                 ("frame_timeline.txt", "frame_timeline.txt")):
    if os.path.exists(os.path.join(G, src)) and os.path.getmtime(os.path.join(G, src)) > float(os.environ.get("SUMMARIZE_NEWER_THAN", "0")):
        shutil.copy(os.path.join(G, src), os.path.join(P, f"{tag}_{dst}"))


def load(path):
    return list(csv.DictReader(open(path))) if os.path.exists(path) else []


def table(prefix, title, note):
    """Markdown table of the derived per-kernel metrics of the passes gpurun_out/<prefix>1..4 (None if the passes are absent)."""
    dur = collections.defaultdict(list)
    for r in load(os.path.join(G, prefix + "1", "p_kernel_trace.csv")):
        dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for pm in (prefix + "1", prefix + "2", prefix + "3", prefix + "4"):
        for r in load(os.path.join(G, pm, "p_counter_collection.csv")):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    lines = [title, "", note, "",
             "| kernel | us | clock GHz | waves | MFMA util | per-wave kcycles: alive / own-MFMA / active / wait-inst / wait-any | VALU/wave | LDS bank-conflict cyc | HBM read MB (2xFETCH) | HBM write MB |",
             "|---|---|---|---|---|---|---|---|---|---|"]
    n0 = len(lines)
    for k, v in agg.items():
        if "conv_mfma" not in k and "corr_fwd" not in k and "conv_sk_kernel" not in k and "conv_wino" not in k and "conv_ws_kernel" not in k:
            continue
        c = {n: sum(x) / len(x) for n, x in v.items()}
        if "GRBM_GUI_ACTIVE" not in c:
            continue
        us = sum(dur[k]) / max(len(dur[k]), 1)
        cyc = c["GRBM_GUI_ACTIVE"] / 8
        w = c["SQ_WAVES"]
        lines.append("| `%s` | %.0f | %.2f | %d | %.1f%% | %.0f / %.0f / %.0f / %.0f / %.0f | %.0f | %.0f | %.1f | %.1f |" % (
            k.replace("void ", "").replace("(ConvKArgs)", "").replace("(CorrArgs)", "").replace("(ConvSkArgs)", "").replace("(ConvWinoArgs)", "").replace("(ConvWsArgs)", ""), us, cyc / us / 1e3, w,
            100 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc), c["SQ_WAVE_CYCLES"] * 4 / w / 1e3,
            c["SQ_VALU_MFMA_BUSY_CYCLES"] / w / 1e3, c["SQ_ACTIVE_INST_ANY"] * 4 / w / 1e3,
            c["SQ_WAIT_INST_ANY"] * 4 / w / 1e3, c.get("SQ_WAIT_ANY", 0) * 4 / w / 1e3, c.get("SQ_INSTS_VALU", 0) / w,
            c.get("SQ_LDS_BANK_CONFLICT", 0), 2 * c.get("FETCH_SIZE", 0) / 1024, c.get("WRITE_SIZE", 0) / 1024))
    return (lines if len(lines) > n0 else None), agg


lines, agg = table("pmc", "# PMC summary (" + tag + ")",
                   "Source: `rocprofv3 --kernel-trace --pmc ...` in four separate passes over `tools/prof_kernels.py` "
                   "(corr at P=5184, representative conv layers); counters averaged over the launches of each kernel.\n"
                   "`GRBM_GUI_ACTIVE` is summed over the 8 XCDs (divide by 8 for cycles); `SQ_WAVE_CYCLES`/`SQ_WAIT_*`/"
                   "`SQ_ACTIVE_INST_ANY` count quad-cycles (x4). `FETCH_SIZE`/`WRITE_SIZE` are KiB; per "
                   "MI355X_MICROARCH.md the gfx950 FETCH_SIZE counts 64 B per 128-B request, so read bytes = 2 x FETCH_SIZE.\n"
                   "MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles).")
lines_r, _ = table("pmcr", "## The same counters at batch 4 under the batch-aware plan (multi-reference pass)",
                   "Source: the same passes over `DVC_PROF_R=4 tools/prof_kernels.py`: ColorVidNet's layer shapes with 4 images per "
                   "launch, planned as a batch (`DVC_CONV_BATCH_PLAN`, ClipColorizer.set_exemplars); a kernel name averages over the "
                   "layers that share its instantiation.")
if lines:
    open(os.path.join(P, f"{tag}_pmc_summary.md"), "w").write("\n".join(lines + ([""] + lines_r if lines_r else [])) + "\n")
elif lines_r:
    open(os.path.join(P, f"{tag}_pmc_summary.md"), "w").write("\n".join(["# PMC summary (" + tag + ")", ""] + lines_r) + "\n")
lines = (lines or []) + (lines_r or [])
# the correlation kernel's HBM traffic per launch, the number bench.py's roofline.traffic cites (file + hash)
for k, v in agg.items():
    if "corr_fwd_kernel" in k and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        rd = 2.0 * 1024 * sum(v["FETCH_SIZE"]) / len(v["FETCH_SIZE"])
        wr = 1024.0 * sum(v["WRITE_SIZE"]) / len(v["WRITE_SIZE"])
        rec = {"kernel": k.replace("void ", "").replace("(CorrArgs)", ""), "P": 5184, "bytes_per_launch": round(rd + wr),
               "read_bytes": round(rd), "write_bytes": round(wr), "compulsory_bytes": 10760000,
               "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over tools/prof_kernels.py; "
                         "read = 2 x FETCH_SIZE (gfx950: 64 B tallied per 128-B request, MI355X_MICROARCH.md), averaged over the "
                         "launches of the kernel",
               "measured_on": f"{tag} (profiles/{tag}_pmc_summary.md)",
               # bench.py refuses this file once csrc/corr.hip no longer is the source the PMC passes ran (r04 review, item 5)
               # (the hash taken ON THE GPU BOX next to the passes, tools/gpu_final.sh; of the local source as a fall-back)
               "corr_hip_sha256_16": (open(os.path.join(G, "corr_hip_sha.txt")).read().strip()[:16]
                                      if os.path.exists(os.path.join(G, "corr_hip_sha.txt")) else
                                      hashlib.sha256(open(os.path.join(ROOT, "deep-exemplar-based-video-colorization_amd", "csrc",
                                                                       "corr.hip"), "rb").read()).hexdigest()[:16])}
        json.dump(rec, open(os.path.join(P, "corr_traffic.json"), "w"))
        print("corr traffic:", rec["bytes_per_launch"], "bytes per launch")
print("\n".join(lines[-8:]))
print("profiles/:", sorted(os.listdir(P)))
