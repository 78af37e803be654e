"""Print register / scratch / LDS usage of kernels matching a regex (hipcc -Rpass-analysis)."""
import re, subprocess, sys
pat = re.compile(sys.argv[1] if len(sys.argv) > 1 else ".")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-std=c++17",
       "-I/root/repo/include", "-c", "/root/repo/tehmm_amd/csrc/tehmm_hip.hip", "-o", "/tmp/_res.o",
       "-Rpass-analysis=kernel-resource-usage"] + sys.argv[2:]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in err.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = cur.split("(")[0].replace("void tehmm::", "")
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[bytes/\w+\])?: (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
for k, v in rows.items():
    if pat.search(k):
        print("%-44s VGPR %3d AGPR %3d vspill %3d sspill %3d scratch %4d occ %s" % (
            k, v.get("VGPRs", -1), v.get("AGPRs", -1), v.get("VGPRs Spill", -1), v.get("SGPRs Spill", -1),
            v.get("ScratchSize", -1), v.get("Occupancy", "?")))
