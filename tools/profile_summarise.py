"""Turns gpurun_out/prof_<tag>/ (tools/profile_round.sh) into the committed summaries under profiles/<tag>/ and
profiles/traffic.json.   usage: python tools/profile_summarise.py r01"""
import csv, glob, json, os, shutil, subprocess, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src, dst = os.path.join(ROOT, "gpurun_out", "prof_" + tag), os.path.join(ROOT, "profiles", tag)
os.makedirs(dst, exist_ok=True)

def first(pattern):
    hits = glob.glob(os.path.join(src, pattern), recursive=True)
    return hits[0] if hits else None

for name, out in (("trace_bench", "kernel_stats_bench_n1.csv"), ("trace_paths", "kernel_stats_secondary_paths.csv"),
                  ("trace_pm", "kernel_stats_prod_marginals_c2.csv"),
                  ("trace_pm_bx", "kernel_stats_prod_marginals_c2_bf16x3.csv"), ("trace_perm", "kernel_stats_mmd_perm_c4.csv"),
                  ("trace_bf16x3", "kernel_stats_bf16x3_comparison.csv"), ("trace_bwd", "kernel_stats_backward.csv"),
                  ("trace_iw_c5_iw", "kernel_stats_iw_c5_reduction_only.csv"),
                  ("trace_iw_c5_dens", "kernel_stats_iw_c5_with_densities.csv"),
                  ("trace_iw_c5_z", "kernel_stats_iw_c5_with_z.csv"), ("trace_wide", "kernel_stats_wide_latents.csv")):
    f = first(f"{name}/**/*kernel_stats.csv")
    if f:
        shutil.copy(f, os.path.join(dst, out))
texts = []
for log in ("paths.log", "pm.log", "pm_bx.log", "perm.log", "bf16x3.log", "bwd.log", "iw.log", "iw_c5_iw.log", "iw_c5_dens.log",
            "iw_c5_z.log", "small.log", "train_tail.log", "wide.log"):
    p = os.path.join(src, log)
    if os.path.exists(p):
        texts += [f"## {log}"] + [l.rstrip() for l in open(p) if not l.startswith(("W2", "E2", "I2")) and "amdgpu.ids" not in l]
open(os.path.join(dst, "secondary_paths.txt"), "w").write("\n".join(texts) + "\n")

pmc_json = os.path.join(dst, "pmc_counters_bench_n1.json")
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), pmc_json] +
                      [os.path.join(src, d) for d in ("pmc_fetch", "pmc_write", "pmc_sq")], stdout=subprocess.DEVNULL)
pmc = json.load(open(pmc_json))
kern = next(k for k in pmc if "pair_lse_mfma_kernel<32, 2, 2, false, 4" in k)
c = {k: v["mean_per_launch"] for k, v in pmc[kern].items()}
cycles = c["GRBM_GUI_ACTIVE"] / 8.0                          # summed over the 8 XCDs
traffic = {
    "pair_lse_mfma_kernel_hbm_bytes_per_launch": (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024,
    "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python3 bench.py --steps 10 "
           "--warmup 2 --no-cpu-baseline` (tools/profile_round.sh); bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 "
           "(FETCH_SIZE doubled per MI355X_MICROARCH.md HBM section: gfx950 reports half of a wide coalesced read "
           "stream); mean over launches",
    "FETCH_SIZE_KB": c["FETCH_SIZE"], "WRITE_SIZE_KB": c["WRITE_SIZE"],
    "algorithmic_min_bytes": 34865152,
    "mfma_util_pct_from_SQ_VALU_MFMA_BUSY_CYCLES": 100.0 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cycles * 1024),
    "kernel": kern.replace("void ", ""), "gpu_cycles_per_launch": cycles,
}
json.dump(traffic, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)

# matrix-core backward: counters per launch of the two sweeps
if os.path.isdir(os.path.join(src, "pmc_bwd_a")) and os.path.isdir(os.path.join(src, "pmc_bwd_b")):
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), os.path.join(dst, "pmc_bwd.json"),
                           os.path.join(src, "pmc_bwd_a"), os.path.join(src, "pmc_bwd_b")], stdout=subprocess.DEVNULL)

# split-bf16 pair kernel: MFMA utilisation by cycles and the clock it ran at
kt, cc = first("pmc_bf16x3/**/*kernel_trace.csv"), first("pmc_bf16x3/**/*counter_collection.csv")
if kt and cc:
    dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt))
           if "pair_lse_mfma_kernel" in r["Kernel_Name"]}
    ctr = defaultdict(lambda: defaultdict(float))
    for r in csv.DictReader(open(cc)):
        ctr[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    rows = [(dur[d], ctr[d]) for d in dur if dur[d] > 1e6 and d in ctr]
    if rows:
        n = len(rows)
        ns = sum(r[0] for r in rows) / n
        cyc = sum(r[1]["GRBM_GUI_ACTIVE"] for r in rows) / n / 8.0
        busy = sum(r[1]["SQ_VALU_MFMA_BUSY_CYCLES"] for r in rows) / n
        json.dump({"kernel": "pair_lse_mfma_kernel<32,2,2,false,4,false,BX=true> (split-bf16), N=M=65536, D=32",
                   "launches": n, "mean_ns": ns, "gpu_cycles_per_launch": cyc, "clock_GHz": cyc / ns,
                   "mfma_util_pct_by_cycles": 100.0 * busy / (cyc * 1024),
                   "mfma_instructions_per_launch": sum(r[1]["SQ_INSTS_MFMA"] for r in rows) / n},
                  open(os.path.join(dst, "pmc_bf16x3_pair_kernel.json"), "w"), indent=1)
# the stats CSV averages ALL launches of the dominant kernel (clock-ramp and warm-up steps included); the timed
# region of bench.py is the last `steps` of them: their average is what bench.py's roofline.kernel_ms_avg must match
kt = first("trace_bench/**/*kernel_trace.csv")
log = os.path.join(src, "trace_bench.log")
if kt and os.path.exists(log):
    line = next((l for l in open(log) if l.startswith('{"metric"')), None)
    rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
                  for r in csv.DictReader(open(kt)) if "pair_lse_mfma_kernel<32, 2, 2, false, 4, false, false>" in r["Kernel_Name"])
    if line and rows:
        b = json.loads(line)
        k = int(b["steps"] * b["roofline"]["launches_per_step"])
        json.dump({"kernel": "pair_lse_mfma_kernel<32,2,2,false,4>", "launches_in_trace": len(rows),
                   "mean_ms_all_launches": sum(d for _, d in rows) / len(rows) / 1e6,
                   "timed_region_launches": k, "mean_ms_timed_region": sum(d for _, d in rows[-k:]) / k / 1e6,
                   "bench_py_kernel_ms_avg_same_run": b["roofline"]["kernel_ms_avg"],
                   "bench_py_ms_per_step_same_run": b["ms_per_step"]},
                  open(os.path.join(dst, "timed_region_bench_n1.json"), "w"), indent=1)
print(json.dumps(traffic, indent=1))
