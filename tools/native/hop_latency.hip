// Cross-stream dependency latency on one GPU: stream A runs a ~20 us kernel K1, stream B a kernel K2 that must start after K1.
// Three ways to say so: (1) hipEventRecord(A) + hipStreamWaitEvent(B); (2) hipStreamWriteValue32(A) + hipStreamWaitValue32(B) on signal
// memory; (3) same stream (no hop).  K1 stamps the 100 MHz wall clock when it ends, K2 when it starts: gap = K2.start - K1.end.
// Build: hipcc --offload-arch=gfx950 -O2 tools/native/hop_latency.hip -o gpurun_out/hop_latency ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k1(long long* stamp, int spin) {
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) { }
    if (threadIdx.x == 0 && blockIdx.x == 0) stamp[0] = wall_clock64();
}
__global__ void k2(long long* stamp) {
    if (threadIdx.x == 0 && blockIdx.x == 0) stamp[1] = wall_clock64();
}
int main() {
    hipStream_t A, B;
    CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
    long long* stamp; CK(hipHostMalloc(&stamp, 64, hipHostMallocMapped));
    hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    unsigned* sig = nullptr;
    hipError_t se = hipExtMallocWithFlags((void**)&sig, 64, hipMallocSignalMemory);
    printf("signal memory: %s\n", se == hipSuccess ? "ok" : hipGetErrorString(se));
    if (se != hipSuccess) {      // fall back to plain device memory, then to pinned host memory
        (void)hipGetLastError();
        se = hipMalloc((void**)&sig, 64);
        printf("plain device memory for the wait value: %s\n", se == hipSuccess ? "ok" : hipGetErrorString(se));
    }
    if (se == hipSuccess) CK(hipMemset(sig, 0, 64));
    const int spin = 2000;       // 20 us
    for (int mode = 0; mode < 3; ++mode) {
        if (mode == 1 && se != hipSuccess) continue;
        std::vector<double> gaps;
        for (int it = 0; it < 60; ++it) {
            stamp[0] = stamp[1] = 0;
            if (mode == 0) {
                hipLaunchKernelGGL(k1, dim3(64), dim3(256), 0, A, stamp, spin);
                CK(hipEventRecord(ev, A));
                CK(hipStreamWaitEvent(B, ev, 0));
                hipLaunchKernelGGL(k2, dim3(64), dim3(256), 0, B, stamp);
            } else if (mode == 1) {
                hipLaunchKernelGGL(k1, dim3(64), dim3(256), 0, A, stamp, spin);
                hipError_t e1 = hipStreamWriteValue32(A, sig, (unsigned)(it + 1), 0);
                hipError_t e2 = hipStreamWaitValue32(B, sig, (unsigned)(it + 1), hipStreamWaitValueGte, 0xffffffffu);
                if (e1 != hipSuccess || e2 != hipSuccess) { printf("stream value ops: %s / %s\n", hipGetErrorString(e1), hipGetErrorString(e2)); (void)hipGetLastError(); hipStreamSynchronize(A); break; }
                hipLaunchKernelGGL(k2, dim3(64), dim3(256), 0, B, stamp);
            } else {
                hipLaunchKernelGGL(k1, dim3(64), dim3(256), 0, A, stamp, spin);
                hipLaunchKernelGGL(k2, dim3(64), dim3(256), 0, A, stamp);
            }
            CK(hipStreamSynchronize(A)); CK(hipStreamSynchronize(B));
            if (it >= 10) gaps.push_back((stamp[1] - stamp[0]) / 100.0);
        }
        std::sort(gaps.begin(), gaps.end());
        const char* names[3] = {"event record + stream wait event", "stream write value + stream wait value (signal memory)", "same stream"};
        printf("%-56s gap K1.end -> K2.start: median %6.1f us, p10 %6.1f, p90 %6.1f\n", names[mode], gaps[gaps.size() / 2],
               gaps[gaps.size() / 10], gaps[gaps.size() * 9 / 10]);
    }
    return 0;
}
