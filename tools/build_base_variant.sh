#!/bin/bash
# The library as of a git revision next to the working tree's build, for tools/ab_libs.sh:  tools/build_base_variant.sh <name> [rev = HEAD]
# -> voicemap_amd/lib/libvoicemap_hip_<name>.so built from <rev>'s voicemap_amd/csrc and include/.
set -e
NAME=$1; REV=${2:-HEAD}
R=$(cd $(dirname $0)/.. && pwd); T=$(mktemp -d)
mkdir -p $T/voicemap_amd $T/include
git -C $R archive $REV voicemap_amd/csrc include | tar -x -C $T
cd $T/voicemap_amd/csrc
for f in *.hip; do
  EXTRA=""; [ "$f" = "conv1_fused.hip" ] && EXTRA="-mllvm -amdgpu-mfma-vgpr-form=1"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc $EXTRA -c $f -o ${f%.hip}.o 2>/dev/null &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/voicemap_amd/lib/libvoicemap_hip_$NAME.so *.o
rm -rf $T; echo built $R/voicemap_amd/lib/libvoicemap_hip_$NAME.so from $REV
