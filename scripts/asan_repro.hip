// Stand-alone check for the sanitizer abort seen at process exit with several frames in flight: no code of this
// repository, only what that path adds over the one-frame path -- N non-blocking streams, an event each, a device
// buffer each, a kernel and an asynchronous copy into page-locked host memory per stream, everything destroyed
// before main returns.  Built like scripts/asan_driver.sh builds the library (host-side sanitizers only):
//   hipcc -O1 -g --offload-arch=gfx950 -Xarch_host -fsanitize=address,undefined scripts/asan_repro.hip -o build/asan/asan_repro
//   ASAN_OPTIONS=detect_leaks=0 ./build/asan/asan_repro 4 [rounds] [MB of device + page-locked memory allocated and freed before exit]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__global__ void fill(unsigned char* p, size_t n, unsigned char v) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

#define OK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { printf("%s: %s\n", #call, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 4;
    const int rounds = argc > 2 ? atoi(argv[2]) : 24;
    const int churn_mb = argc > 3 ? atoi(argv[3]) : 0;      // device and page-locked memory allocated and freed before exit
    const size_t bytes = 720 * 540 * 3;
    hipStream_t s[8]; hipEvent_t ev[8]; unsigned char* d[8]; unsigned char* h = nullptr;
    if (n < 1 || n > 8) return 2;
    OK(hipHostMalloc((void**)&h, bytes * (size_t)rounds, hipHostMallocPortable));
    for (int i = 0; i < n; i++) {
        OK(hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking));
        OK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
        OK(hipMalloc((void**)&d[i], bytes));
    }
    for (int r = 0; r < rounds; r++) {
        const int i = r % n;
        if (r >= n) OK(hipEventSynchronize(ev[i]));
        hipLaunchKernelGGL(fill, dim3((unsigned)((bytes + 255) / 256)), dim3(256), 0, s[i], d[i], bytes, (unsigned char)r);
        OK(hipGetLastError());
        OK(hipMemcpyAsync(h + bytes * (size_t)r, d[i], bytes, hipMemcpyDeviceToHost, s[i]));
        OK(hipEventRecord(ev[i], s[i]));
    }
    for (int i = 0; i < n; i++) OK(hipStreamSynchronize(s[i]));
    int bad = 0;
    for (int r = 0; r < rounds; r++) bad += h[bytes * (size_t)r + 12345] != (unsigned char)r;
    for (int i = 0; i < n; i++) { OK(hipStreamDestroy(s[i])); OK(hipEventDestroy(ev[i])); OK(hipFree(d[i])); }
    OK(hipHostFree(h));
    for (int mb = 0; mb < churn_mb; mb += 32) {
        void* p = nullptr; void* q = nullptr;
        OK(hipMalloc(&p, (size_t)16 << 20)); OK(hipHostMalloc(&q, (size_t)16 << 20, hipHostMallocPortable));
        OK(hipFree(p)); OK(hipHostFree(q));
    }
    printf("asan_repro: %d streams, %d rounds, %d wrong frames, every handle destroyed\n", n, rounds, bad);
    return bad != 0;
}
