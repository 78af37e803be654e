"""Register / scratch / LDS usage of kernels from a saved `hipcc -Rpass-analysis=kernel-resource-usage` log.
usage: python tools/resparse.py remarks.txt [regex]"""
import re
import subprocess
import sys

pat = re.compile(sys.argv[2] if len(sys.argv) > 2 else ".")
cur = None
rows = {}
for line in open(sys.argv[1]):
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = cur.split("(")[0].replace("void tehmm::", "").replace("tehmm::", "")
        rows[cur] = {}
        continue
    m = re.search(r":\d+:\d+:(?: remark:)?\s+([A-Za-z ]+?)(?: \[[\w/]+\])?: (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
for k, v in rows.items():
    if pat.search(k):
        print("%-46s VGPR %3d AGPR %3d SGPR %3d vspill %3d sspill %3d scratch %4d occ %s LDS %s" % (
            k[:46], v.get("VGPRs", -1), v.get("AGPRs", -1), v.get("TotalSGPRs", -1), v.get("VGPRs Spill", -1),
            v.get("SGPRs Spill", -1), v.get("ScratchSize", -1), v.get("Occupancy", "?"), v.get("LDS Size", "?")))
