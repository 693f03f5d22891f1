"""Kernel resource usage of one HIP source: python tools/kres.py pl-inertial-slam_amd/csrc/plba_marg.hip [name-filter]
(hipcc -Rpass-analysis=kernel-resource-usage, one line per kernel: VGPRs, AGPRs, scratch bytes per lane, spills, LDS, occupancy)."""
import os, re, subprocess, sys
src = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-I", "include", "-I", "pl-inertial-slam_amd/csrc",
       "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + os.environ.get("PLBA_EXTRA_FLAGS", "").split()
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None; rows = []
for line in out.splitlines():
    m = re.search(r"remark: [^:]*:\d+:\d+:\s+(.*?) \[-Rpass", line) or re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}; rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1); cur[k.strip()] = v.strip()
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "").replace("plba::", "")
    name = re.sub(r"^void ", "", name); name = name.split("(")[0]
    if flt and flt not in name: continue
    print("%-60s vgpr %4s agpr %4s scratch %5s spill(s/v) %s/%s lds %7s occ %s" % (name[-60:], r.get("VGPRs"), r.get("AGPRs"), r.get("ScratchSize [bytes/lane]"),
          r.get("SGPRs Spill"), r.get("VGPRs Spill"), r.get("LDS Size [bytes/block]"), r.get("Occupancy [waves/SIMD]")))
