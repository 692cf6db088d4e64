#!/bin/bash
# libvoicemap_hip_prof.so: the library with -DVM_EXPERIMENT_PROFILE (conv_nt2r_kernel writes per-wave s_memtime intervals; read them
# with tools/probe/nt2r_prof.py / nt3_prof.py under VOICEMAP_HIP_LIB=.../libvoicemap_hip_prof.so).  Not part of the product build.
# VM_PROF_EXTRA=-DVM_EXPERIMENT_PROFILE_EPI: raw stamps of the epilogue's phases instead (nt3_prof.py with EPI=1).
set -e
R=$(cd $(dirname $0)/.. && pwd); T=$(mktemp -d)
python -m voicemap_amd.build >/dev/null
cp $R/voicemap_amd/build/*.o $T/
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -DVM_EXPERIMENT_PROFILE $VM_PROF_EXTRA -c $R/voicemap_amd/csrc/conv_gemm.hip -o $T/conv_gemm.o 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/voicemap_amd/lib/libvoicemap_hip_prof.so $T/*.o
rm -rf $T; echo built $R/voicemap_amd/lib/libvoicemap_hip_prof.so
