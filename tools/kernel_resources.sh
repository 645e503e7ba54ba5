#!/bin/bash
# Registers, LDS and occupancy of every kernel of sphx_resident.hip as compiled for gfx950 (no GPU needed):
#   [SPHX_EXTRA_FLAGS=-D...] tools/kernel_resources.sh [filter-regex] > table
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
S=/tmp/sphx_resident_$$.s
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only $SPHX_EXTRA_FLAGS -S -o $S "$ROOT/sph-poiseuille-flow_amd/csrc/sphx_resident.hip" || exit 1
python3 - "$S" "${1:-.}" <<'PY'
import re, subprocess, sys
txt = open(sys.argv[1]).read()
pat = re.compile(sys.argv[2])
for m in re.finditer(r"^\s*\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S | re.M):
    name, body = m.group(1), m.group(2)
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
    if not pat.search(dem):
        continue
    g = lambda k: (re.search(r"\.amdhsa_" + k + r"\s+(\d+)", body) or [0, "?"])[1]
    # the human-readable block the compiler prints behind each kernel
    tail = txt[m.end():m.end() + 3000]
    occ = (re.search(r"; Occupancy:\s*(\d+)", tail) or [0, "?"])[1]
    vg = (re.search(r"; NumVgprs:\s*(\d+)", tail) or [0, "?"])[1]
    ag = (re.search(r"; NumAgprs:\s*(\d+)", tail) or [0, "?"])[1]
    sg = (re.search(r"; TotalNumSgprs:\s*(\d+)", tail) or [0, "?"])[1]
    sc = (re.search(r"; ScratchSize:\s*(\d+)", tail) or [0, "?"])[1]
    print(f"{dem:90s} vgpr {vg:>3} agpr {ag:>3} sgpr {sg:>3} lds {g('group_segment_fixed_size'):>6} scratch {sc:>4} occupancy {occ}")
PY
rm -f $S
