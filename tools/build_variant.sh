#!/bin/bash
# An experiment build of the library next to the product one:  tools/build_variant.sh <name> "<extra hipcc flags>" [file.hip ...]
# -> voicemap_amd/lib/libvoicemap_hip_<name>.so (select it with VOICEMAP_HIP_LIB=...).  Only the listed sources (default: conv_gemm.hip)
# are recompiled with the extra flags; everything else comes from the product build.  Not part of the product.
set -e
NAME=$1; FLAGS=$2; shift 2 || true
FILES=${@:-conv_gemm.hip}
R=$(cd $(dirname $0)/.. && pwd); T=$(mktemp -d)
python -m voicemap_amd.build >/dev/null
cp $R/voicemap_amd/build/*.o $T/
for f in $FILES; do
  EXTRA=""; [ "$f" = "conv1_fused.hip" ] && EXTRA="-mllvm -amdgpu-mfma-vgpr-form=1"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc $EXTRA $FLAGS -c $R/voicemap_amd/csrc/$f -o $T/${f%.hip}.o 2>/dev/null
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/voicemap_amd/lib/libvoicemap_hip_$NAME.so $T/*.o
rm -rf $T; echo built $R/voicemap_amd/lib/libvoicemap_hip_$NAME.so
