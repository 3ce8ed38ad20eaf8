#!/bin/bash
# Dev aid (CPU only): the host-side C++ of the library -- bucket reader, SAM formatter, index builder --
# compiled with AddressSanitizer + UBSan into throw-away libraries under /tmp and driven by the CPU tests of those
# parts (GPU sanitizers are not available on the pool; the kernels go through the host SIMT interpreter instead, which
# tests/emu/build.sh can build with -fsanitize as well).
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=${TMPDIR:-/tmp}/ema_asan_$$
mkdir -p "$out"
flags="-O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -std=c++17 -fPIC -shared"
g++ $flags -I"$root/include" -I"$root/ema_amd/csrc" -o "$out/libhost.so" "$root"/ema_amd/csrc/host_ingest.cpp "$root"/ema_amd/csrc/host_sam.cpp -lpthread
g++ $flags -fopenmp -o "$out/libindex.so" "$root"/ema_amd/csrc/index_build.cpp
cat > "$out/run.py" <<PY
import sys, ctypes
sys.path.insert(0, "$root"); sys.path.insert(0, "$root/tests")
from ema_amd import engine, index
engine._lib = ctypes.CDLL("$out/libhost.so")
L = ctypes.CDLL("$out/libindex.so")
L.ema_index_build.argtypes = [ctypes.c_char_p, ctypes.c_int]; L.ema_index_build.restype = ctypes.c_int
index._lib = L
import pytest
sys.exit(pytest.main(["-x", "-q", "-p", "no:cacheprovider", "$root/tests/test_ingest.py", "$root/tests/test_sam_format.py",
                      "$root/tests/test_index_build.py"]))
PY
cd "$out"
ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)" python3 run.py
rm -rf "$out"
