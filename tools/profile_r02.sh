#!/bin/bash
# Round-2 rocprofv3 evidence for the secondary kernels: PMC passes on the permutation
# kernel, the 15-bandwidth kernel-sum kernel and the C3 rank share.  One counter set per run, never with API traces.
# Run on the GPU box from the repo root:   bash tools/profile_r02.sh [tag] [bwd]     (bwd: only the backward passes)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r02}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
step() { echo "== $1" | tee -a "$OUT/progress.log"; }
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY GRBM_GUI_ACTIVE"
SQ2="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD"
pass() {  # name, counters, program...
    local name=$1 ctr=$2; shift 2
    step "$name"
    timeout -s KILL 170 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$OUT/$name" -o p -- "$@" > "$OUT/$name.log" 2>&1 || { echo "FAILED $name" | tee -a "$OUT/progress.log"; exit 1; }
}
if [ "${2:-all}" != "bwd" ]; then
for W in perm tts; do
    N=131072; [ $W = tts ] && N=65536
    pass ${W}_sq   "$SQ"          python3 $R/tools/run_mmd.py $W $N 32 2
    pass ${W}_sq2  "$SQ2"         python3 $R/tools/run_mmd.py $W $N 32 2
    pass ${W}_tcc  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"   python3 $R/tools/run_mmd.py $W $N 32 2
    pass ${W}_fetch "FETCH_SIZE"  python3 $R/tools/run_mmd.py $W $N 32 2
done
pass c3_fetch "FETCH_SIZE"  python3 $R/tools/bench_c3_share.py
pass c3_write "WRITE_SIZE"  python3 $R/tools/bench_c3_share.py
pass c3_tcc   "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" python3 $R/tools/bench_c3_share.py
pass c3_sq    "$SQ"         python3 $R/tools/bench_c3_share.py
step "mmd + c3 done"
fi
# matrix-core backward (16,384^2, D = 32): matrix-pipe and vector busy, waits
if [ "${2:-}" = "bwd" ] || [ -z "${2:-}" ]; then
    pass bwd_sq  "$SQ"  python3 $R/tools/bench_bwd.py 16384,16384,32
    pass bwd_sq2 "$SQ2" python3 $R/tools/bench_bwd.py 16384,16384,32
    step "bwd done"
fi
