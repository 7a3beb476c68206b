# GPU: profile by deletion of conv3x3_wres16_kernel (flavour libraries built by tools/flavour.sh w<name> wres16.hip "-DDBN_WRES_DBG=<bits>")
for f in "" ${WRES_AB:-wmfma wall wdma}; do
  if [ -z "$f" ]; then lib=db_text_minimal_amd/libdbnet_hip.so; else lib=db_text_minimal_amd/libdbnet_hip_$f.so; fi
  echo "== ${f:-product}"
  DBN_LIB_PATH=$PWD/$lib python tools/wres_probe.py 2>&1 | grep -v amdgpu | grep -E "${WRES_AB_SHAPES:-cfg5 (layer1|head|layer2)}" | sed 's/| patch.*//'
done
