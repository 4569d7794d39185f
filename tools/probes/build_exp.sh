#!/bin/bash
# An EXPERIMENTS build of the library beside the shipped one: databend_amd/libdbhip_exp.so (git-ignored; select it with DBHIP_LIBRARY).
# Only the named translation units are rebuilt with -DDBHIP_EXPERIMENTS, the rest is linked from the shipped build's objects.
#   bash tools/probes/build_exp.sh k_parquet_dev [k_groupby ...]
set -e
cd "$(dirname "$0")/../../databend_amd/csrc"
make -s all > /dev/null
mkdir -p build_exp
OBJS=""
for f in *.hip; do
  n=${f%.hip}; o=build/$n.o
  for x in "$@"; do
    if [ "$x" = "$n" ]; then
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-bitwise-instead-of-logical -I../../include -DDBHIP_EXPERIMENTS $EXP_DEFS -c $f -o build_exp/$n.o
      o=build_exp/$n.o
    fi
  done
  OBJS="$OBJS $o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libdbhip_exp.so $OBJS -ldl -Wl,--version-script=build/exports.map
echo built ../libdbhip_exp.so
