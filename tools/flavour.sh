#!/bin/bash
# Build an A/B library that differs from the product library in ONE translation unit:
#   tools/flavour.sh <name> <file.hip> "<extra defines>"   ->  db_text_minimal_amd/libdbnet_hip_<name>.so  (select with DBN_LIB_PATH)
# (make FLAVOUR=... rebuilds every unit; this reuses the product objects of the other units.)
set -e
cd "$(dirname "$0")/../db_text_minimal_amd/csrc"
name=$1; unit=$2; extra=$3
make -s -j8 > /dev/null
mkdir -p obj_$name
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off $extra -c $unit -o obj_$name/${unit%.hip}.o
objs=""
for f in conv conv_f32 convt_f32 winograd_f32 winograd_wgrad_f32 conv_x3 conv_b16 wgrad wgrad_f32 wgrad_b16 pack pointwise head_loss postproc deform stem16 convt16; do
  if [ "$f.hip" == "$unit" ]; then objs="$objs obj_$name/$f.o"; else objs="$objs ./$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libdbnet_hip_$name.so $objs
echo built ../libdbnet_hip_$name.so
