"""Builds libtehmm_hip.so (hipcc, gfx950) in-tree next to the sources.

-ffp-contract=off is required for bit-exact Viterbi (the reference rounds the multiply and the
add of the segment-ratio term separately; hipcc's default would contract them into an FMA).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libtehmm_hip.so")
SOURCES = [os.path.join(CSRC, "tehmm_hip.hip")]
DEPS = SOURCES + sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + [
    os.path.join(os.path.dirname(HERE), "include", "tehmm_hip.h")]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


STAMP = LIB + ".flags"      # the flags the in-tree library was built with (a development build is rebuilt)


def _built_flags():
    try:
        with open(STAMP) as f:
            return f.read().strip()
    except OSError:
        return None


def needs_build(flags=""):
    if not os.path.exists(LIB) or _built_flags() != flags:
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in DEPS if os.path.exists(d))


def build(force=False, verbose=False, dev_nt=None):
    """dev_nt (or TEHMM_DEV_NT in the environment): development build for ONE padded state count
    (fast to compile; every other N then fails with TEHMM_ERR_UNSUPPORTED-like silence -- never ship it)."""
    dev_nt = dev_nt or os.environ.get("TEHMM_DEV_NT")
    flags = "-DTEHMM_DEV_NT=%d" % int(dev_nt) if dev_nt else ""
    extra = os.environ.get("TEHMM_EXTRA_FLAGS", "").split()      # diagnostics builds (e.g. -DTEHMM_CHAIN_PROF)
    flags = " ".join(([flags] if flags else []) + extra)
    if not force and not needs_build(flags):
        return LIB
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared",
           "-std=c++17", "-Wall", "-Wno-unused-function", "-Wno-unused-lambda-capture", "-o", LIB] + SOURCES
    if flags:
        cmd[1:1] = flags.split()
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    if os.path.exists(STAMP):
        os.remove(STAMP)
    subprocess.check_call(cmd)
    with open(STAMP, "w") as f:
        f.write(flags + "\n")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
