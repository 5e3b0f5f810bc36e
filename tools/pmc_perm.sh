#!/bin/bash
# PMC passes of the C4 permutation kernel (LVAE_PATH_AUTO) -> gpurun_out/<tag>/pmc_perm*.  Usage: bash tools/pmc_perm.sh <tag> [n]
# Each counter set is a run of its own (no trace domains next to --pmc); summarised by tools/pmc_summary.py.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r03}
N=${2:-131072}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_perm_a" -o p -- python3 $R/tools/bench_perm.py $N,$N,32,100 > "$OUT/pmc_perm_a.log" 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d "$OUT/pmc_perm_b" -o p -- python3 $R/tools/bench_perm.py $N,$N,32,100 > "$OUT/pmc_perm_b.log" 2>&1 || exit 1
python3 $R/tools/pmc_summary.py "$OUT/pmc_perm.json" "$OUT/pmc_perm_a" "$OUT/pmc_perm_b" > "$OUT/pmc_perm_summary.txt" 2>&1
grep -h "PERM\|true, 5, true\|perm" "$OUT/pmc_perm_summary.txt" | head -5
