"""Developer aid: table of kernel resource usage (hipcc -Rpass-analysis=kernel-resource-usage) for kernels.hip.
usage: python tools/resources.py [extra -D flags ...] [--filter substr]"""
import re, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
flt = None
if "--filter" in args:
    i = args.index("--filter"); flt = args[i + 1]; del args[i:i + 2]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "--offload-arch=gfx950",
       "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-gpu-flush-denormals-to-zero", "-Rpass-analysis=kernel-resource-usage",
       "-c", os.path.join(ROOT, "pathtracing_amd/csrc/kernels.hip"), "-o", "/dev/null"] + args
err = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None; rows = {}
for line in err.splitlines():
    m = re.search(r"remark: (?:\s*)([A-Za-z ]+?)(?: \[[^\]]*\])?: (.+?) \[-Rpass", line)
    if not m: continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
        cur = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void ptrt::", "")
        rows[cur] = {}
    elif cur: rows[cur][k] = v
print(f"{'kernel':44s} {'VGPR':>5s} {'SGPR':>5s} {'vSpill':>6s} {'sSpill':>6s} {'scratch':>7s} {'occ':>4s} {'LDS':>6s}")
for k, r in rows.items():
    if flt and flt not in k: continue
    print(f"{k:44s} {r.get('VGPRs','?'):>5s} {r.get('TotalSGPRs','?'):>5s} {r.get('VGPRs Spill','?'):>6s} {r.get('SGPRs Spill','?'):>6s} "
          f"{r.get('ScratchSize','?'):>7s} {r.get('Occupancy','?'):>4s} {r.get('LDS Size','?'):>6s}")
