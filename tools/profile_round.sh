#!/bin/bash
# Collects the rocprofv3 evidence behind bench.py's roofline object and DESIGN.md's tables.
# Run on the GPU box from the repo root:   bash tools/profile_round.sh r01
# Every pass is a run of its own (counters are never combined with API traces; at most 8 SQ / 4 TCC counters per
# pass); outputs land under gpurun_out/prof_<tag>/ and are summarised into profiles/<tag>/ by
# tools/profile_summarise.py (run afterwards, on the CPU).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r01}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-side-runs"
step() { echo "== $1" | tee -a "$OUT/progress.log"; }

step "kernel trace of bench.py"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_bench" -o p -- $BENCH > "$OUT/trace_bench.log" 2>&1 || exit 1
step "PMC FETCH_SIZE"
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o p -- $BENCH > "$OUT/pmc_fetch.log" 2>&1 || exit 1
step "PMC WRITE_SIZE"
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o p -- $BENCH > "$OUT/pmc_write.log" 2>&1 || exit 1
step "PMC SQ"
timeout -s KILL 150 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_sq" -o p -- $BENCH > "$OUT/pmc_sq.log" 2>&1 || exit 1
step "kernel trace of the secondary paths"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_paths" -o p -- python3 $R/tools/bench_paths.py > "$OUT/paths.log" 2>&1 || exit 1
step "kernel trace of prod-marginals (C2)"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_pm" -o p -- python3 $R/tools/bench_pm.py 65536,65536,32 4096,4096,32 > "$OUT/pm.log" 2>&1 || exit 1
step "kernel trace of prod-marginals (C2), experimental bf16x3 path"
LVAE_PATH=3 timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_pm_bx" -o p -- python3 $R/tools/bench_pm.py 65536,65536,32 > "$OUT/pm_bx.log" 2>&1 || exit 1
step "kernel trace of the permutation test (C4)"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_perm" -o p -- python3 $R/tools/bench_perm.py 262144,262144,32,100 > "$OUT/perm.log" 2>&1 || exit 1
step "kernel trace of the split-bf16 comparison"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_bf16x3" -o p -- python3 $R/tools/bench_bf16x3.py > "$OUT/bf16x3.log" 2>&1 || exit 1
step "PMC SQ + clock of the split-bf16 pair kernel"
timeout -s KILL 150 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_bf16x3" -o p -- python3 $R/tools/run_path.py 3 65536 32 20 > "$OUT/pmc_bf16x3.log" 2>&1 || exit 1
step "kernel trace of the matrix-core backward"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_bwd" -o p -- python3 $R/tools/bench_bwd.py 16384,16384,32 16384,16384,64 16384,16384,128 65536,65536,32 65536,65536,64 42068,42068,64 8192,8192,64 > "$OUT/bwd.log" 2>&1 || exit 1
step "PMC of the matrix-core backward (two passes)"
LVAE_BWD_PATH=mfma timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_bwd_a" -o p -- python3 $R/tools/bench_bwd.py 16384,16384,32 > "$OUT/pmc_bwd_a.log" 2>&1 || exit 1
LVAE_BWD_PATH=mfma timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$OUT/pmc_bwd_b" -o p -- python3 $R/tools/bench_bwd.py 16384,16384,32 > "$OUT/pmc_bwd_b.log" 2>&1 || exit 1
step "kernel trace of the IW call at C5, one form per run (the reduction alone, with the two density matrices, with z)"
for form in iw dens z; do
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_iw_c5_$form" -o p -- python3 $R/tools/bench_iw.py 16384,1000,32 only=$form > "$OUT/iw_c5_$form.log" 2>&1 || exit 1
done
step "IW, small problems, training tail (no trace)"
timeout -k 10 200 python3 $R/tools/bench_iw.py > "$OUT/iw.log" 2>&1 || exit 1
timeout -k 10 200 python3 $R/tools/bench_small.py > "$OUT/small.log" 2>&1 || exit 1
timeout -k 10 200 python3 $R/tools/bench_train_tail.py > "$OUT/train_tail.log" 2>&1 || exit 1
step "kernel trace of the wide-latent paths"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_wide" -o p -- python3 $R/tools/bench_wide.py > "$OUT/wide.log" 2>&1 || exit 1
step "done"
