# us per conv_nt3 launch for experiment builds of the library (tools/build_variant.sh <name> ...): VARIANTS="default prio1 ..." bash tools/probe/nt3_variants.sh
for rep in 1 2; do for v in ${VARIANTS:-default}; do L=""; [ $v != default ] && L=$PWD/voicemap_amd/lib/libvoicemap_hip_$v.so; echo "== $v"; VOICEMAP_HIP_LIB=$L timeout 120 python tools/probe/nt3_launch_times.py 2>&1 | tail -6; done; done
