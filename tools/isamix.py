"""Instruction mix of kernels matching a regex in a hipcc -S dump (tools/isamix.py REGEX [asmfile])."""
import re, subprocess, sys
from collections import Counter
asm = sys.argv[2] if len(sys.argv) > 2 else "/tmp/lane.s"
if len(sys.argv) <= 2:
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17",
                    "-I/root/repo/include", "-S", "--cuda-device-only", "/root/repo/tehmm_amd/csrc/tehmm_hip.hip",
                    "-o", asm], check=True, stderr=subprocess.DEVNULL)
s = open(asm).read()
pat = re.compile(sys.argv[1])
for m in re.finditer(r'^(_Z\w+):[^\n]*\n', s, re.M):
    name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0]
    if not pat.search(name):
        continue
    st = m.end(); en = s.index('s_endpgm', st)
    body = s[st:en]
    c = Counter(l.split()[0] for l in body.splitlines()
                if l.strip() and not l.strip().startswith((';', '.')) and not l.strip().endswith(':'))
    print(name, sum(c.values()))
    print('   ', c.most_common(24))
