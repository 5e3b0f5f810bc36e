"""Register / spill report of every kernel in the library: compiles each .hip device-only for gfx950 and reads the
code-object metadata (.vgpr_count, .agpr_count, .vgpr_spill_count, .private_segment_fixed_size).
usage: python tools/spill_report.py [file.hip ...]     exit status 1 if any kernel spills vector registers."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "language-transformer-vae_amd", "csrc")
UNROLL = {"pair_lse_mfma.hip", "pair_lse_bf16x3.hip"}

def report(path):
    name = os.path.basename(path)
    with tempfile.TemporaryDirectory() as td:
        co = os.path.join(td, "k.co")
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "--no-gpu-bundle-output", "-c", path, "-o", co]
        if name in UNROLL:
            cmd[1:1] = ["-mllvm", "-unroll-threshold=100000"]
        cmd += [a for a in os.environ.get("SPILL_REPORT_FLAGS", "").split() if a]
        cmd[cmd.index(path)] = os.path.abspath(path); subprocess.run(cmd, check=True, cwd=SRC)
        txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
    rows = []
    for e in re.split(r"\n\s+- \.agpr_count:", txt)[1:]:
        g = lambda k: re.search(r"\.%s:\s+(\S+)" % k, e).group(1)
        sym = g("name")
        dem = subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.strip()
        dem = re.sub(r"\(.*", "", dem).replace("void ", "")
        rows.append((dem, int(re.match(r"\s*(\d+)", e).group(1)), int(g("vgpr_count")), int(g("vgpr_spill_count")),
                     int(g("private_segment_fixed_size")), int(g("group_segment_fixed_size"))))
    return rows

def main():
    files = sys.argv[1:] or sorted(os.path.join(SRC, f) for f in os.listdir(SRC) if f.endswith(".hip"))
    bad = 0
    for f in files:
        for dem, ag, vg, sp, scr, lds in sorted(report(f)):
            flag = "  <-- SPILLS" if sp else ""
            bad += sp > 0
            print(f"{os.path.basename(f):22s} {dem:72s} vgpr {vg:3d} agpr {ag:3d} spill {sp:4d} scratch {scr:5d} lds {lds:6d}{flag}")
    return 1 if bad else 0

if __name__ == "__main__":
    sys.exit(main())
