#!/bin/bash
# PMC passes of the matrix-core backward (tools/bench_bwd.py) -> gpurun_out/<tag>/pmc_bwd*.  Usage: bash tools/pmc_bwd.sh <tag> [M,N,D]
# Each counter set is a run of its own (no trace domains next to --pmc); summarised by tools/pmc_summary.py.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r03}
SHAPE=${2:-16384,16384,32}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export LVAE_BWD_PATH=mfma
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_bwd_a" -o p -- python3 $R/tools/bench_bwd.py $SHAPE > "$OUT/pmc_bwd_a.log" 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$OUT/pmc_bwd_b" -o p -- python3 $R/tools/bench_bwd.py $SHAPE > "$OUT/pmc_bwd_b.log" 2>&1 || exit 1
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_bwd1" -o p -- python3 $R/tools/bench_bwd.py $SHAPE > "$OUT/trace_bwd1.log" 2>&1 || exit 1
python3 $R/tools/pmc_summary.py "$OUT/pmc_bwd.json" "$OUT/pmc_bwd_a" "$OUT/pmc_bwd_b" > "$OUT/pmc_bwd_summary.txt" 2>&1
cp $(find "$OUT/trace_bwd1" -name "*kernel_stats.csv" | head -1) "$OUT/kernel_stats_bwd1.csv"
grep -h "pair_bwd" "$OUT/kernel_stats_bwd1.csv" | cut -c1-200
