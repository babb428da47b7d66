#!/bin/bash
# Host-side sanitizer runs (no GPU): ASan + UBSan over the C++ host layer's self-test (against stubs of the C ABI) and
# ThreadSanitizer over the multi-threaded BVH builder; ASan + UBSan over the wide regrouping of a tree and its 64-byte quantised form (bvh_wide.cpp).  usage: tools/sanitize/run.sh   (from the repo root)
set -e
OUT=${TMPDIR:-/tmp}/rvpt_sanitize
mkdir -p $OUT
g++ -std=c++17 -g -fsanitize=address,undefined -o $OUT/selftest_asan rvpt_amd/host/host_selftest.cpp rvpt_amd/host/rvpt_host.cpp \
    rvpt_amd/csrc/bvh_builder.cpp tools/sanitize/abi_stubs.cpp -Iinclude -lpthread
$OUT/selftest_asan $OUT
g++ -std=c++17 -g -fsanitize=thread -o $OUT/bvh_tsan tools/sanitize/build_bvh_threads.cpp rvpt_amd/csrc/bvh_builder.cpp -Iinclude -lpthread
RVPT_BVH_THREADS=8 $OUT/bvh_tsan
g++ -std=c++17 -g -fsanitize=address,undefined -DRVPT_HIP_LAB=1 -o $OUT/wide_asan tools/sanitize/wide_form.cpp rvpt_amd/csrc/bvh_wide.cpp rvpt_amd/csrc/bvh_builder.cpp -Iinclude -lpthread
$OUT/wide_asan
echo "sanitizers: clean"
