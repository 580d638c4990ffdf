"""Build libspectrobot_hip.so for gfx950 with hipcc (in-tree, no JIT cache)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "lib", "libspectrobot_hip.so")
SOURCES = ["sr_api.hip", "sr_kernels.hip"]
DEPS = SOURCES + ["sr_device.hpp", "sr_kernels.hpp", "tips2003_tables.inc", "../../include/spectrobot_hip.h"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-fno-fast-math", "-fgpu-rdc" if False else "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]


def up_to_date():
    if not os.path.exists(OUT):
        return False
    t = os.path.getmtime(OUT)
    return all(os.path.getmtime(os.path.join(CSRC, d)) <= t for d in DEPS) and \
        os.path.getmtime(os.path.abspath(__file__)) <= t


def build(force=False, verbose=False, extra=(), out=None):
    """out: build a variant (extra flags, e.g. -DSR_KTHETA=5) to another path, for
    SPECTROBOT_HIP_LIB=<path>; the default library is never overwritten by a variant."""
    if out is None and extra:
        raise ValueError("a build with extra flags needs its own output path (out=...)")
    if out is None and not force and up_to_date():
        return OUT
    OUT_ = OUT if out is None else os.path.abspath(out)
    return _compile(OUT_, verbose, extra)


def _compile(OUT, verbose, extra):
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    tmp = "%s.tmp.%d" % (OUT, os.getpid())  # other processes never see a half-written library
    cmd = [HIPCC] + FLAGS + list(extra) + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", tmp]
    if verbose:
        print(" ".join(cmd).replace(tmp, OUT))
    try:
        subprocess.check_call(cmd)
        os.replace(tmp, OUT)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return OUT


def wait_until_built(timeout=600.0):
    """For the ranks that do not build: wait for the building rank's library."""
    import time
    t0 = time.time()
    while not up_to_date():
        if time.time() - t0 > timeout:
            raise RuntimeError("libspectrobot_hip.so was not built within %.0f s" % timeout)
        time.sleep(0.5)
    return OUT


if __name__ == "__main__":
    # python build.py [--force] [--out PATH -DSR_KTHETA=5 ...]
    argv = sys.argv[1:]
    out = None
    if "--out" in argv:
        i = argv.index("--out")
        out = argv[i + 1]
        del argv[i:i + 2]
    print(build(force="--force" in argv, verbose=True,
                extra=[a for a in argv if a.startswith("-") and a != "--force"], out=out))
