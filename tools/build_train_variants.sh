#!/bin/bash
# A/B builds of csrc/train.hip (training kernels): only train.hip is recompiled per variant, the other objects are the
# production ones.  Select with CASMVS_LIB_PATH=casmvsnet_pl_amd/libtr_<name>.so.
#   usage: tools/build_train_variants.sh name1:"-DFLAG ..." name2:"..."
cd "$(dirname "$0")/.." || exit 1
python -c "from casmvsnet_pl_amd.build import build_library; build_library()" || exit 1
PKG=casmvsnet_pl_amd
mkdir -p $PKG/build_tr
pids=()
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC $flags -Iinclude -I$PKG/csrc \
      -c $PKG/csrc/train.hip -o $PKG/build_tr/train_$name.o &&
    objs=$(ls $PKG/build/*.o | grep -v "/train.o") &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $PKG/build_tr/train_$name.o -o $PKG/libtr_$name.so &&
    echo "built libtr_$name.so ($flags)" ) &
  pids+=($!)
done
rc=0; for p in "${pids[@]}"; do wait $p || rc=1; done; exit $rc
