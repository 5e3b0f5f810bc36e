import sys, os
sys.path.insert(0, ".")
sys.path.insert(0, "tools")
from quick_bench import run
for tw in sys.argv[1:]:
    os.environ["LVAE_TARGET_WAVES"] = tw
    print("target waves", tw)
    run(65536, 65536, 32, 1, iters=7)
    run(65536, 65536, 64, 1, iters=5)
