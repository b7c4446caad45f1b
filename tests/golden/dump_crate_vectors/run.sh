#!/bin/sh
# One command for anyone with cargo (this repository's build image has none): build the dumper against the real crates and
# the reference's own library, write tests/golden/crate_vectors.json, and run the tests that consume it.
#   sh tests/golden/dump_crate_vectors/run.sh            (from anywhere; -m gpu too when an MI355X is there)
set -e
here=$(cd "$(dirname "$0")" && pwd)
cd "$here"
cargo run --release > ../crate_vectors.json.tmp
mv ../crate_vectors.json.tmp ../crate_vectors.json
cd "$here/../../.."
python -m pytest tests/test_crate_vectors.py -q -m "not gpu"
if python -c "import deacon_server_amd as d, ctypes; n=ctypes.c_int(); import sys; sys.exit(0 if d._native.lib().dcn_device_count(ctypes.byref(n))==0 and n.value>0 else 1)" 2>/dev/null; then
  python -m pytest tests/test_crate_vectors.py -q -m gpu
fi
