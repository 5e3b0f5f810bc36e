set -o pipefail
mkdir -p gpurun_out/r3c
run() { name=$1; shift; timeout -k 10 1000 "$@" > gpurun_out/r3c/$name.log 2>&1; echo "$name rc=$? $(tail -1 gpurun_out/r3c/$name.log | cut -c1-120)"; }
run fwd python -u tests/fuzz_log_qz.py 1001 1000
run fwd_big python -u tests/fuzz_log_qz.py 1002 80 big
run bwd python -u tests/fuzz_bwd.py 2000 1003
run iw python -u tests/fuzz_iw.py 3000 1004
run tail python -u tests/fuzz_tail.py 600 1005
run api python -u tests/fuzz_api.py 400 1006
