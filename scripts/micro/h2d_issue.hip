// How long does the HOST spend inside hipMemcpyAsync (page-locked source -> device) as a function of size and alignment?
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/h2d_issue scripts/micro/h2d_issue.hip && /tmp/h2d_issue
// (r06: the 16-bit gap column of a 0.5 M-signature batch - 0.96 MB - made csv_cluster_batch's upload 1.1 ms slower on the host)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t cap = 64u << 20;
    char *h = nullptr, *d = nullptr;
    hipHostMalloc((void**)&h, cap, hipHostMallocDefault);
    hipMalloc((void**)&d, cap);
    for (size_t i = 0; i < cap; i += 4096) h[i] = 1;
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    const size_t sizes[] = {64u << 10, 256u << 10, 512u << 10, 959584, 1u << 20, (1u << 20) + 2, 1919168, 4u << 20, 5563704, 16u << 20};
    for (int align = 0; align < 3; align++) {
        const size_t off = align == 0 ? 0 : (align == 1 ? 2 : 4);
        for (size_t s : sizes) {
            double best_issue = 1e9, best_total = 1e9;
            for (int rep = 0; rep < 6; rep++) {
                hipStreamSynchronize(st);
                const double t0 = now();
                hipMemcpyAsync(d + off, h + off, s, hipMemcpyHostToDevice, st);
                const double t1 = now();
                hipStreamSynchronize(st);
                const double t2 = now();
                if (t1 - t0 < best_issue) best_issue = t1 - t0;
                if (t2 - t0 < best_total) best_total = t2 - t0;
            }
            printf("offset %zu  size %9zu B: issue %8.1f us, until done %8.1f us (%.1f GB/s)\n", off, s, best_issue, best_total, s / best_total / 1e3);
        }
    }
    return 0;
}
