# round 5: cycle stamps inside k_sg_tconv<256, 256> (build: tools/build_variant.sh sgprof rgn_sg_kernels.hip -DRGN_SG_PROF)
cp regennet_amd/libregennet_hip.so /tmp/lib_keep.so
cp build/lib_sgprof.so regennet_amd/libregennet_hip.so
python tools/sg_stamps.py 2>&1 | tail -45
cp /tmp/lib_keep.so regennet_amd/libregennet_hip.so
