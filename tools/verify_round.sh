set -o pipefail
mkdir -p gpurun_out/r3v
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r3v/gpu_tests.log 2>&1 || { tail -20 gpurun_out/r3v/gpu_tests.log; exit 1; }
tail -3 gpurun_out/r3v/gpu_tests.log
timeout -k 10 200 python -u tests/fuzz_log_qz.py 777 150 > gpurun_out/r3v/fuzz_fwd.log 2>&1 || { tail -20 gpurun_out/r3v/fuzz_fwd.log; exit 1; }
tail -2 gpurun_out/r3v/fuzz_fwd.log
timeout -k 10 200 python -u tests/fuzz_log_qz.py 778 24 big > gpurun_out/r3v/fuzz_fwd_big.log 2>&1 || { tail -20 gpurun_out/r3v/fuzz_fwd_big.log; exit 1; }
tail -2 gpurun_out/r3v/fuzz_fwd_big.log
timeout -k 10 300 python -u tests/fuzz_bwd.py 500 200 > gpurun_out/r3v/fuzz_bwd.log 2>&1 || { tail -20 gpurun_out/r3v/fuzz_bwd.log; exit 1; }
tail -2 gpurun_out/r3v/fuzz_bwd.log
timeout -k 10 300 python -u tests/fuzz_iw.py 400 5 > gpurun_out/r3v/fuzz_iw.log 2>&1 || { tail -20 gpurun_out/r3v/fuzz_iw.log; exit 1; }
tail -1 gpurun_out/r3v/fuzz_iw.log
timeout -k 10 300 python -u tests/fuzz_tail.py 150 3 > gpurun_out/r3v/fuzz_tail.log 2>&1 || { tail -20 gpurun_out/r3v/fuzz_tail.log; exit 1; }
tail -1 gpurun_out/r3v/fuzz_tail.log
timeout -k 10 600 python -u tests/fuzz_api.py 100 9 > gpurun_out/r3v/fuzz_api.log 2>&1 || { tail -20 gpurun_out/r3v/fuzz_api.log; exit 1; }
tail -1 gpurun_out/r3v/fuzz_api.log
LVAE_BENCH_REHEARSAL=1 timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/r3v/bench_torchrun2.log 2>&1 || { tail -20 gpurun_out/r3v/bench_torchrun2.log; exit 1; }
grep '^{"metric"' gpurun_out/r3v/bench_torchrun2.log | cut -c1-300
timeout -k 10 300 python bench.py > gpurun_out/r3v/bench.log 2>&1 || { tail -20 gpurun_out/r3v/bench.log; exit 1; }
grep '^{"metric"' gpurun_out/r3v/bench.log | cut -c1-400
