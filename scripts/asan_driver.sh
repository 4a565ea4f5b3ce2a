#!/bin/bash
# Builds the C-ABI library, the host library and examples/drone_animation.c with the address and undefined-behaviour sanitizers on the HOST side
# only (device code is compiled as usual) into build/asan/, for a run on a GPU box:
#     bash scripts/asan_driver.sh            (here, no GPU needed)
#     ASAN_OPTIONS=detect_leaks=0:quarantine_size_mb=0 ./build/asan/drone_animation raytracer_amd/assets out.webp 4      (on the box;
#         with the default quarantine the sanitizer recycles released device memory from inside the HSA runtime's exit handler and aborts)
#     ASAN_OPTIONS=detect_leaks=0 ./build/asan/abi_check                                             (on the box)
# The sanitizer runtime is the one of the ROCm clang, so all three pieces are compiled with it.
set -eu
cd "$(dirname "$0")/.."
mkdir -p build/asan
CL=/opt/rocm/lib/llvm/bin
# the C-ABI library through its own Makefile (one object list), host side sanitized, objects kept out of the product's directory
make -C raytracer_amd/csrc BUILD=../../build/asan OPT="-O1 -g" EXTRA="-Xarch_host -fsanitize=address,undefined"
cd build/asan
rm -f ./*.o
$CL/clang++ -O1 -g -std=c++17 -fPIC -ffp-contract=off -fsanitize=address,undefined -shared -I../../include -o libraytracer_host.so \
    ../../raytracer_amd/host/raytracer_host.cpp ../../raytracer_amd/host/raytracer_math.cpp ../../raytracer_amd/host/webp_anim.cpp \
    -pthread -L. -lrt_mi355 -ldl -lm -Wl,-rpath,'$ORIGIN'
$CL/clang -std=c11 -O1 -g -fsanitize=address,undefined -D_POSIX_C_SOURCE=200809L -D_DEFAULT_SOURCE -I../../raytracer_amd/host -o drone_animation \
    ../../examples/drone_animation.c -L. -lraytracer_host -Wl,-rpath,'$ORIGIN' -lm
# a walk over the whole C ABI (tests/host_emul/abi_check.cpp), same sanitizer
$CL/clang++ -O1 -g -std=c++17 -fsanitize=address,undefined -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I../../include -o abi_check \
    ../../tests/host_emul/abi_check.cpp -L. -lrt_mi355 -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,'$ORIGIN' -Wl,-rpath,/opt/rocm/lib
ls -la
