#!/bin/bash
# build an experiment variant of the library: profiles/build_variant.sh <name> <extra hipcc flags...>
set -e
name=$1; shift
cd "$(dirname "$0")/../deacon-server_amd/csrc"
out=../lib/variants; mkdir -p $out ../build/var_$name
for f in api.hip collective.hip index_table.hip pack.hip plan.hip scan.hip index_file.cpp; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value -Wno-unused-result -x hip "$@" -c $f -o ../build/var_$name/${f%.*}.o &
done
g++ -O3 -std=c++17 -fPIC -c host_pack.cpp -o ../build/var_$name/host_pack.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libdeacon_hip_$name.so ../build/var_$name/*.o -ldl
echo built $out/libdeacon_hip_$name.so
