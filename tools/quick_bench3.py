import sys, os, subprocess
for lib in sys.argv[1:]:
    env = dict(os.environ, LVAE_LIB=os.path.abspath(lib), LVAE_TARGET_WAVES="2048")
    print("==", lib, flush=True)
    subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tools'); from quick_bench import run; run(65536,65536,32,1,iters=7); run(65536,65536,64,1,iters=5)"], env=env)
