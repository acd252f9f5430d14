#!/bin/bash
# usage: tools/kres.sh <file.hip> [grep-pattern] [extra hipcc flags...] -- compact per-kernel VGPR / scratch / occupancy table
f=$1; pat=${2:-.}; shift; shift
cd /root/repo/rcppml_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -Rpass-analysis=kernel-resource-usage -c $f -o /tmp/kres_$$.o 2>&1 \
 | awk '/Function Name:/ {n=$(NF-1)} / VGPRs:/ {v=$(NF-1)} /ScratchSize/ {s=$(NF-1)} /Occupancy/ {o=$(NF-1); print n, "v=" v, "scr=" s, "occ=" o}' | grep -E "$pat"
rm -f /tmp/kres_$$.o
