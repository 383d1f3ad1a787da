"""Developer tool: registers / scratch / occupancy of the kernels of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage).
    python tools/kres.py stage [filter ...]"""
import os, re, subprocess, sys
here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "deeptreeattention_amd", "csrc")
f = sys.argv[1]
filt = sys.argv[2:]
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Wno-unused-result",
                    "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(here, f + ".hip"), "-o", "/tmp/%s.dev.o" % f],
                   capture_output=True, text=True)
txt = r.stderr
for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = b.split("\n")[0].strip()
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
    if filt and not any(k in dn for k in filt):
        continue
    g = lambda k: (re.search(k + r": (\d+)", b) or [None, "?"])[1]
    print(dn[:110], "| VGPR", g("VGPRs"), "AGPR", g("AGPRs"), "spill", g("VGPRs Spill") if "VGPRs Spill" in b else g("VGPR Spill"), "scratch", g(r"ScratchSize \[bytes/lane\]"),
          "occ", g(r"Occupancy \[waves/SIMD\]"), "LDS", g(r"LDS Size \[bytes/block\]"))
