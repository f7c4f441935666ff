#!/bin/bash
# The CPU test suite against AddressSanitizer + UndefinedBehaviorSanitizer builds of the host front end
# (parser, KD builder, flattener, BMP writer) and of the oracle.  Host code only: the device library is not
# involved in that part; a second part instruments the host code of the device library.  The in-tree libraries
# are put back when the run ends, however it ends.
# usage: tools/sanitize_host.sh            (CPU only, about three minutes)
set -u
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
flags="-std=c++17 -O1 -g -ffp-contract=off -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer -shared"
(cd $root/raytracer_amd/host && g++ $flags -o $tmp/librt_host.so scene.cpp parser.cpp bitmap.cpp host_api.cpp) || exit 1
g++ $flags -pthread -o $tmp/liboracle.so $root/oracle/rt_oracle.cpp || exit 1
cp $root/raytracer_amd/librt_host.so $tmp/librt_host.orig && cp $root/oracle/liboracle.so $tmp/liboracle.orig || exit 1
restore() { cp $tmp/librt_host.orig $root/raytracer_amd/librt_host.so; cp $tmp/liboracle.orig $root/oracle/liboracle.so; rm -rf $tmp; }
trap restore EXIT
cp $tmp/librt_host.so $root/raytracer_amd/librt_host.so && cp $tmp/liboracle.so $root/oracle/liboracle.so || exit 1
cd $root
LD_PRELOAD=$(g++ -print-file-name=libasan.so):$(g++ -print-file-name=libubsan.so) \
ASAN_OPTIONS=detect_leaks=0:halt_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 \
  python3 -m pytest tests -q -m "not gpu" -p no:cacheprovider --deselect tests/test_abi.py > $tmp/out.txt 2>&1
rc=$?
tail -3 $tmp/out.txt
n=$(grep -c "runtime error\|ERROR: AddressSanitizer" $tmp/out.txt)
echo "sanitizer reports: $n"
[ $rc -eq 0 ] && [ "$n" -eq 0 ] || exit 1
# The host side of the device library (scene validation, BVH builder, node copies and cones, the invariant
# checker) the same way: host code instrumented, device code as always; the tests that need no GPU.
cp $tmp/librt_host.orig $root/raytracer_amd/librt_host.so && cp $tmp/liboracle.orig $root/oracle/liboracle.so || exit 1
cp $root/raytracer_amd/librt_amd.so $tmp/librt_amd.orig || exit 1
restore2() { cp $tmp/librt_amd.orig $root/raytracer_amd/librt_amd.so; restore; }
trap restore2 EXIT
(cd $root/raytracer_amd/csrc && hipcc --offload-arch=gfx950 -std=c++17 -O1 -g -ffp-contract=off -fPIC -shared -Wno-unused-value \
   -Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -Xarch_host -fno-omit-frame-pointer \
   -o $tmp/librt_amd.so rt_kernels.hip) || exit 1
cp $tmp/librt_amd.so $root/raytracer_amd/librt_amd.so || exit 1
LD_PRELOAD=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1) \
ASAN_OPTIONS=detect_leaks=0:halt_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 \
  python3 -m pytest tests/test_bvh_invariants.py tests/test_abi.py -q -p no:cacheprovider > $tmp/out_amd.txt 2>&1
rc=$?
tail -3 $tmp/out_amd.txt
n=$(grep -c "runtime error\|ERROR: AddressSanitizer" $tmp/out_amd.txt)
echo "sanitizer reports (device library's host code): $n"
[ $rc -eq 0 ] && [ "$n" -eq 0 ]
