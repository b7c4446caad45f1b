"""A9 host side: the index file codec of csrc/index_file.cpp (threaded bincode-2 varint writer, reader), checked
byte for byte against the format's definition by tests/cpp/index_file_test.cpp.  No GPU needed."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_index_file_codec(tmp_path):
    exe = tmp_path / "index_file_test"
    csrc = os.path.join(ROOT, "deacon-server_amd", "csrc")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-x", "hip", "--offload-arch=gfx950", "-fno-gpu-rdc",
                           "-Wno-unused-value", "-I", csrc, "-I", os.path.join(ROOT, "include"),
                           os.path.join(csrc, "index_file.cpp"), os.path.join(ROOT, "tests", "cpp", "index_file_test.cpp"),
                           "-o", str(exe), "-lpthread"])
    # (a regular file is written by all threads at their own offsets; DCN_INDEX_WRITE_SERIAL=1 is the one-writer form that
    # pipes and /dev/stdout take)
    for threads, extra in (("1", {}), ("3", {}), ("8", {}), ("3", {"DCN_INDEX_WRITE_SERIAL": "1"})):
        out = subprocess.run([str(exe), str(tmp_path / "t.idx")], capture_output=True, env=dict(os.environ, DCN_HOST_THREADS=threads, **extra))
        assert out.returncode == 0, out.stderr.decode()
        assert b"index file codec ok" in out.stdout
