#!/bin/bash
# VGPR / scratch / LDS / code size of every kernel of one source file (compiles it for gfx950 with
# the Makefile's flags and prints the compiler's resource remarks).
#   tools/kernel_info.sh finenv_cashpenalty.hip [extra -D flags]
set -e
cd "$(dirname "$0")/../finrl_amd/csrc"
src=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I../../include -I. "$@" \
    -c -o /dev/null "$src" -Rpass-analysis=kernel-resource-usage 2>&1 |
  python3 -c '
import sys, re
name = None; d = {}
for ln in sys.stdin:
    m = re.search(r"remark:\s*(.*?)\s*\[-Rpass", ln)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        name = t.split(":", 1)[1].strip(); d = {}
    for k, lab in (("VGPRs:", "vgpr"), ("AGPRs:", "agpr"), ("ScratchSize [bytes/lane]:", "scratch"),
                   ("LDS Size [bytes/block]:", "lds"), ("Occupancy [waves/SIMD]:", "occ"),
                   ("SGPRs Spill:", "sspill")):
        if t.startswith(k): d[lab] = t.split(":", 1)[1].strip()
    if t.startswith("LDS Size") and name:
        print("%-84s vgpr %3s agpr %3s scratch %4s sgpr-spill %3s lds %6s occ %s" % (name[:84], d.get("vgpr"), d.get("agpr"), d.get("scratch"), d.get("sspill"), d.get("lds"), d.get("occ")))
'
# code sizes (bytes) of the same kernels: compile to an object, unbundle the gfx950 code object
tmp=$(mktemp -d); trap 'rm -rf $tmp' EXIT
B=/opt/rocm/lib/llvm/bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I../../include -I. "$@" \
    -c -o $tmp/x.o "$src" >/dev/null 2>&1
$B/llvm-objcopy --dump-section .hip_fatbin=$tmp/fb.bin $tmp/x.o
$B/clang-offload-bundler --unbundle --type=o --input=$tmp/fb.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$tmp/co.elf
$B/llvm-readelf -s $tmp/co.elf | awk '$4=="FUNC" && !seen[$8]++ {printf "  code %6d B  %s\n", $3, substr($8,1,100)}'
