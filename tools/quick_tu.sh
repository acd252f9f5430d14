#!/bin/bash
# usage: tools/quick_tu.sh <out.so name under rcppml_amd/lib> <tu.hip> [extra flags...] -- recompile ONE translation unit with the Makefile's flags
# and relink the library from the objects already in csrc/build (a header edit otherwise rebuilds every unit: ~6 min)
out=$1; tu=$2; shift; shift
cd /root/repo/rcppml_amd/csrc
extra=""
case $tu in ops_cd_f32.hip|ops_cd_f64.hip|ops_irls.hip|ops_cv.hip) extra="-mllvm -amdgpu-mfma-vgpr-form=1";; ops_cd_lmf.hip) extra="-mllvm -amdgpu-mfma-vgpr-form=1 -fno-slp-vectorize";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function -Wno-pass-failed $extra "$@" -c $tu -o build/${tu%.hip}.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/*.o -ldl -o ../lib/$out
