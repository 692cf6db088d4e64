# conv_nt3_kernel with parts switched off (tools/build_variant.sh abl<bits> -DVM_NT3_ABL=<bits> conv_gemm.hip; results wrong by design):
# 1 no in-loop weight loads, 2 no in-loop input DMA, 4 no K-loop MFMAs, 8 no epilogue.  Launch times alone, same box.
for v in default abl1 abl2 abl4 abl8 abl12; do L=""; [ $v != default ] && L=$PWD/voicemap_amd/lib/libvoicemap_hip_$v.so; echo "== $v"; NOCHECK=1 VOICEMAP_HIP_LIB=$L timeout 120 python tools/probe/nt3_launch_times.py 2>&1 | tail -6; done
