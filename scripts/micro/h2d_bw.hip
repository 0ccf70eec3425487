// micro-benchmark: host -> device bandwidth of this box for the shapes csv_batch_upload uses
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/h2d_bw scripts/micro/h2d_bw.hip && /tmp/h2d_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void k_pull(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

int main()
{
    const size_t N = 64ull << 20;
    void *pin, *pin2, *dev, *pageable;
    CK(hipHostMalloc(&pin, N, hipHostMallocDefault));
    CK(hipHostMalloc(&pin2, N, hipHostMallocNonCoherent));
    pageable = malloc(N);
    memset(pin, 1, N); memset(pin2, 1, N); memset(pageable, 1, N);
    CK(hipMalloc(&dev, N));
    hipStream_t st[8];
    for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    auto run = [&](const char* name, void* src, int nstreams, int nchunks) {
        double best = 1e9;
        for (int rep = 0; rep < 6; rep++) {
            CK(hipDeviceSynchronize());
            const double t0 = now();
            const size_t chunk = N / nchunks;
            for (int c = 0; c < nchunks; c++)
                CK(hipMemcpyAsync((char*)dev + c * chunk, (char*)src + c * chunk, chunk, hipMemcpyHostToDevice, st[c % nstreams]));
            for (int s = 0; s < nstreams; s++) CK(hipStreamSynchronize(st[s]));
            const double dt = now() - t0;
            if (dt < best) best = dt;
        }
        printf("%-44s %2d streams %2d chunks: %7.3f ms  %6.1f GB/s\n", name, nstreams, nchunks, best * 1e3, N / best / 1e9);
    };
    run("pinned (default)", pin, 1, 1);
    run("pinned (default)", pin, 1, 4);
    run("pinned (default)", pin, 2, 4);
    run("pinned (default)", pin, 4, 4);
    run("pinned (default)", pin, 4, 16);
    run("pinned (default)", pin, 8, 16);
    run("pinned (non-coherent)", pin2, 1, 1);
    run("pinned (non-coherent)", pin2, 4, 4);
    run("pageable", pageable, 1, 1);
    run("pageable", pageable, 4, 4);
    // register pageable
    { const double t0 = now(); CK(hipHostRegister(pageable, N, hipHostRegisterDefault)); printf("hipHostRegister(64 MiB): %.3f ms\n", (now() - t0) * 1e3); }
    run("registered", pageable, 1, 1);
    run("registered", pageable, 4, 4);
    // kernel pull from mapped pinned memory
    for (int grid : {64, 256, 1024}) {
        double best = 1e9;
        void* dsrc; CK(hipHostGetDevicePointer(&dsrc, pin, 0));
        for (int rep = 0; rep < 6; rep++) {
            CK(hipDeviceSynchronize());
            const double t0 = now();
            hipLaunchKernelGGL(k_pull, dim3(grid), dim3(256), 0, st[0], (const uint4*)dsrc, (uint4*)dev, N / 16);
            CK(hipStreamSynchronize(st[0]));
            const double dt = now() - t0;
            if (dt < best) best = dt;
        }
        printf("kernel pull from mapped pinned memory, grid %4d: %7.3f ms  %6.1f GB/s\n", grid, best * 1e3, N / best / 1e9);
    }
    // D2H
    {
        double best = 1e9;
        for (int rep = 0; rep < 6; rep++) { CK(hipDeviceSynchronize()); const double t0 = now(); CK(hipMemcpyAsync(pin, dev, 8 << 20, hipMemcpyDeviceToHost, st[0])); CK(hipStreamSynchronize(st[0])); const double dt = now() - t0; if (dt < best) best = dt; }
        printf("D2H 8 MiB pinned: %.3f ms %.1f GB/s\n", best * 1e3, (8 << 20) / best / 1e9);
        best = 1e9;
        for (int rep = 0; rep < 6; rep++) { CK(hipDeviceSynchronize()); const double t0 = now(); CK(hipMemcpyAsync(pin, dev, 64, hipMemcpyDeviceToHost, st[0])); CK(hipStreamSynchronize(st[0])); const double dt = now() - t0; if (dt < best) best = dt; }
        printf("D2H 64 B round trip: %.1f us\n", best * 1e6);
    }
    return 0;
}
