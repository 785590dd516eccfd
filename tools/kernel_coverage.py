"""Which compiled kernel instantiations does a GPU test run launch?
  1. on the GPU box:  rocprofv3 --kernel-trace -d gpurun_out/testtrace -o t -- python3 -m pytest tests -m gpu -q -p no:cacheprovider
                      python3 tools/kernel_coverage.py names gpurun_out/testtrace/t_results.db > gpurun_out/test_kernels.txt
  2. here (hipcc):    python3 tools/kernel_coverage.py compare gpurun_out/test_kernels.txt
     compiles every csrc/*.hip with -Rpass-analysis=kernel-resource-usage (which names every kernel instantiation), demangles, and
     lists the instantiations that the trace never saw."""
import glob, os, re, sqlite3, subprocess, sys

def norm(s):
    s = re.sub(r"^void ", "", s.strip())
    return s.split("(")[0].replace(" ", "")

if sys.argv[1] == "names":
    con = sqlite3.connect(sys.argv[2])
    print("\n".join(sorted({n for (n,) in con.execute("select distinct name from kernels")})))
else:
    here = os.path.dirname(os.path.abspath(__file__))
    csrc = os.path.join(here, "..", "vision_amd", "csrc")
    flags = "-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math".split()
    mangled = set()
    for f in sorted(glob.glob(os.path.join(csrc, "*.hip"))):
        r = subprocess.run(["/opt/rocm/bin/hipcc"] + flags + ["-Rpass-analysis=kernel-resource-usage", "-c", f, "-o", "/dev/null"],
                           capture_output=True, text=True, cwd=csrc)
        mangled |= set(re.findall(r"Function Name: (\S+)", r.stderr))
    dem = subprocess.run(["c++filt"], input="\n".join(sorted(mangled)), capture_output=True, text=True).stdout.splitlines()
    compiled = {norm(l) for l in dem if l.strip()}
    launched = {norm(l) for l in open(sys.argv[2]) if l.strip()}
    ours = {c for c in compiled if c.startswith("k_")}
    print(f"{len(ours)} compiled kernel instantiations, {len(ours & launched)} launched by the traced run")
    for c in sorted(ours - launched):
        print("  not launched:", c)
