"""Registers, scratch, LDS and occupancy of every kernel of a HIP source, from the compiler's resource-usage remarks
(development aid; runs without a GPU).  Usage: kernel_regs.py [source.hip] [substring filter]"""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "raytracer-rs_amd", "csrc", "rt_kernels.hip")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
       "-Rpass-analysis=kernel-resource-usage", "-c", "-o", "/dev/null"] + (["-x", "hip"] if src.endswith(".cpp") else []) + [src]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = {}
for line in out.splitlines():
    m = re.search(r"remark: [^:]*:\d+:\d+: +(.*?)(?: \[-Rpass|$)", line) or re.search(r"remark: +(.*?)(?: \[-Rpass|$)", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": subprocess.run(["c++filt", t.split(":", 1)[1].strip()], capture_output=True, text=True).stdout.strip().split("(")[0]}
    elif ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
        if k.strip().startswith("LDS Size") and flt in cur.get("name", ""):
            print(f"{cur['name'][:70]:70s} vgpr {cur.get('VGPRs','?'):>4s} agpr {cur.get('AGPRs','?'):>3s} sgpr {cur.get('SGPRs','?'):>4s} "
                  f"scratch {cur.get('ScratchSize [bytes/lane]','?'):>4s} occ {cur.get('Occupancy [waves/SIMD]','?'):>2s} lds {cur.get('LDS Size [bytes/block]','?')}")
