# Top-level build: product libraries, oracle (test infrastructure) and the test-only host emulation.
#   make            everything
#   make product    librt_mi355.so (HIP, gfx950) + libraytracer_host.so (reference host API)
#   make examples examples/drone_animation (a C driver on the host API)
#   make oracle     oracle/liboracle.so and, where /root/reference exists, oracle/_ref/libref.so
CXX ?= g++

all: product oracle emul examples tools

product:
	$(MAKE) -C raytracer_amd/csrc
	$(MAKE) -C raytracer_amd/host

oracle:
	$(MAKE) -C oracle

emul: tests/host_emul/libemul.so tests/host_emul/abi_check

# a C++ program that walks the whole C ABI on a GPU (run by tests/test_gpu_edges.py)
tests/host_emul/abi_check: tests/host_emul/abi_check.cpp include/rt_mi355.h product
	$(CXX) -O2 -std=c++17 -Wall -Wextra -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude -o $@ tests/host_emul/abi_check.cpp \
	    -L raytracer_amd/csrc -lrt_mi355 -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,'$$ORIGIN/../../raytracer_amd/csrc' -Wl,-rpath,/opt/rocm/lib

tests/host_emul/libemul.so: tests/host_emul/emul.cpp raytracer_amd/csrc/rt_pixel.h raytracer_amd/csrc/rt_sampled.h raytracer_amd/csrc/rt_instanced.h raytracer_amd/csrc/rt_bicubic.h raytracer_amd/csrc/rt_pack.h raytracer_amd/csrc/rt_exact.h raytracer_amd/csrc/rt_layout.h include/rt_mi355.h
	$(CXX) -O2 -std=c++17 -ffp-contract=off -fopenmp -fPIC -shared -Wall -Wextra -o $@ tests/host_emul/emul.cpp -lm

# C programs on the reference's host API, compiled as C11 against the product library
examples: examples/drone_animation

examples/drone_animation: examples/drone_animation.c raytracer_amd/host/raytracer_host.h product
	$(CC) -std=c11 -O2 -Wall -Wextra -D_POSIX_C_SOURCE=200809L -D_DEFAULT_SOURCE -I raytracer_amd/host -o $@ examples/drone_animation.c \
	    -L raytracer_amd/host -lraytracer_host -Wl,-rpath,'$$ORIGIN/../raytracer_amd/host' -lm

# measurement aids that run on the GPU box (built here, travel with the tree)
tools: scripts/valu_issue_bench

scripts/valu_issue_bench: scripts/valu_issue_bench.hip
	/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o $@ scripts/valu_issue_bench.hip

clean:
	$(MAKE) -C raytracer_amd/csrc clean
	$(MAKE) -C raytracer_amd/host clean
	$(MAKE) -C oracle clean
	rm -f tests/host_emul/libemul.so tests/host_emul/abi_check examples/drone_animation scripts/valu_issue_bench

.PHONY: all product oracle emul examples tools clean
