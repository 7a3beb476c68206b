#!/bin/bash
# Build an A/B library that differs from the product library in ONE translation unit:
#   tools/flavour.sh <name> <file.hip> "<extra defines>"   ->  db_text_minimal_amd/libdbnet_hip_<name>.so  (select with DBN_LIB_PATH)
# (make FLAVOUR=... rebuilds every unit; this reuses the product objects of the other units.)  The unit list and the compile flags are
# the Makefile's (make print-srcs / print-cxxflags): a new .hip file or a changed flag there is picked up here.
set -e
cd "$(dirname "$0")/../db_text_minimal_amd/csrc"
name=$1; unit=$2; extra=$3
make -s -j8 > /dev/null
srcs=$(make -s print-srcs)
flags=$(make -s print-cxxflags)
case " $srcs " in *" $unit "*) ;; *) echo "flavour.sh: $unit is not one of the Makefile's units: $srcs" >&2; exit 2;; esac
mkdir -p obj_$name
/opt/rocm/bin/hipcc $flags $extra -c $unit -o obj_$name/${unit%.hip}.o
objs=""
for f in $srcs; do
  if [ "$f" == "$unit" ]; then objs="$objs obj_$name/${f%.hip}.o"; else objs="$objs ./${f%.hip}.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libdbnet_hip_$name.so $objs
echo built ../libdbnet_hip_$name.so
