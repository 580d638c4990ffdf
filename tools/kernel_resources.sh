#!/bin/bash
# Per-kernel register / LDS / scratch use: compiles sr_kernels.hip device-only (same flags as
# spectrobot_amd/build.py) and reads the code object's metadata.  usage: tools/kernel_resources.sh [pattern]
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
trap 'rm -rf "$tmp"' EXIT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fno-gpu-rdc \
  --cuda-device-only -c "$root/spectrobot_amd/csrc/sr_kernels.hip" -o "$tmp/k.co" ${SR_EXTRA_FLAGS:-}
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input="$tmp/k.co" \
  --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output="$tmp/k.elf"
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$tmp/k.elf" | python3 -c '
import sys, re, subprocess
pat = sys.argv[1] if len(sys.argv) > 1 else ""
txt = sys.stdin.read()
for blk in txt.split("- .agpr_count")[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s*(\S+)", blk) or [None, "?"])[1]
    name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"^void sr::", "", name).split("(")[0]
    if pat in name:
        print("%-58s vgpr %4s sgpr %4s lds %6s scratch %5s" % (name[:58], g("vgpr_count"), g("sgpr_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size")))
' "${1:-}"
