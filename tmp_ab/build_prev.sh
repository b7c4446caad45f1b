#!/bin/bash
# builds lib/variants/libdeacon_hip_${SCANV:-prev}.so: the current tree with the previous scan.hip / plan.hip
set -e
cd "$(dirname "$0")/../deacon-server_amd/csrc"
out=../lib/variants; mkdir -p $out ../build/var_${SCANV:-prev}
cp ../../tmp_ab/scan_${SCANV:-prev}.hip ./scan_prev_tmp.hip; cp ./plan.hip ./plan_prev_tmp.hip
for f in api.hip index_table.hip pack.hip index_file.cpp; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value -Wno-unused-result -x hip -c $f -o ../build/var_${SCANV:-prev}/${f%.*}.o &
done
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value -Wno-unused-result -x hip -c scan_prev_tmp.hip -o ../build/var_${SCANV:-prev}/scan.o &
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value -Wno-unused-result -x hip -c plan_prev_tmp.hip -o ../build/var_${SCANV:-prev}/plan.o &
g++ -O3 -std=c++17 -fPIC -c host_pack.cpp -o ../build/var_${SCANV:-prev}/host_pack.o &
wait
rm -f scan_prev_tmp.hip plan_prev_tmp.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libdeacon_hip_${SCANV:-prev}.so ../build/var_${SCANV:-prev}/*.o
echo built prev
