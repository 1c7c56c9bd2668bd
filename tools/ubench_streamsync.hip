// Cost of a cross-stream dependency on the RECORDING stream (gfx950, ROCm 7): a chain of short kernels on stream A, each followed by a
// hand-over to stream B that runs a short kernel of its own.  Modes: 0 no hand-over (B free-running), 1 hipEventRecord(A) +
// hipStreamWaitEvent(B), 2 hipStreamWriteValue32(A) + hipStreamWaitValue32(B) on signal memory, 3 as 1 but one hand-over per 4 kernels.
// Build: hipcc --offload-arch=gfx950 -O2 tools/ubench_streamsync.hip -o /tmp/ubench_streamsync
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#include <vector>
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(err_)); exit(1); } } while (0)

__global__ void spin_kernel(float* p, int iters) {
    float v = p[threadIdx.x + blockIdx.x * blockDim.x];
    for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
    p[threadIdx.x + blockIdx.x * blockDim.x] = v;
}

int main(int argc, char** argv) {
    const int N = 400, iters = argc > 1 ? atoi(argv[1]) : 2000;
    float *pa, *pb;
    CK(hipMalloc(&pa, 256 * 256 * 4));
    CK(hipMalloc(&pb, 256 * 256 * 4));
    CK(hipMemset(pa, 0, 256 * 256 * 4));
    CK(hipMemset(pb, 0, 256 * 256 * 4));
    hipStream_t A, B;
    CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    uint32_t* sig = nullptr;
    if (can) {
        CK(hipExtMallocWithFlags((void**)&sig, 8, hipMallocSignalMemory));
        CK(hipMemset(sig, 0, 8));
    }
    printf("stream wait-value supported: %d; spin iterations %d\n", can, iters);
    std::vector<hipEvent_t> ev(N);
    for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    hipEvent_t t0, t1;
    CK(hipEventCreate(&t0));
    CK(hipEventCreate(&t1));
    uint32_t seq = 0;
    for (int mode = 0; mode < 4; ++mode) {
        if (mode == 2 && !can) continue;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipDeviceSynchronize());
            auto h0 = std::chrono::steady_clock::now();
            CK(hipEventRecord(t0, A));
            for (int i = 0; i < N; ++i) {
                hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, A, pa, iters);
                const bool hand = mode == 1 || mode == 2 || (mode == 3 && (i & 3) == 3);
                if (hand && mode != 2) {
                    CK(hipEventRecord(ev[i], A));
                    CK(hipStreamWaitEvent(B, ev[i], 0));
                } else if (hand) {
                    ++seq;
                    CK(hipStreamWriteValue32(A, sig, seq, 0));
                    CK(hipStreamWaitValue32(B, sig, seq, hipStreamWaitValueGte, 0xffffffffu));
                }
                hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, B, pb, iters);
            }
            CK(hipEventRecord(t1, A));
            auto h1 = std::chrono::steady_clock::now();
            CK(hipStreamSynchronize(A));
            CK(hipStreamSynchronize(B));
            auto h2 = std::chrono::steady_clock::now();
            float ms = 0;
            CK(hipEventElapsedTime(&ms, t0, t1));
            printf("mode %d: stream A %.2f us per kernel, host enqueue %.2f us per iteration, all done after %.2f us per iteration\n", mode,
                   ms * 1e3 / N, std::chrono::duration<double, std::micro>(h1 - h0).count() / N,
                   std::chrono::duration<double, std::micro>(h2 - h0).count() / N);
        }
    }
    return 0;
}
