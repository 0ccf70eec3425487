// PMC calibration: one micro-kernel per access pattern the clustering pipeline uses, each with a byte count the host
// computes exactly (at 32 / 64 / 128-byte granularity), run under rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in
// separate passes (scripts/micro/pmc_calib.sh).  rocprof_traffic.py divides a pipeline kernel's counter by the factor
// of its pattern.  (MI355X_MICROARCH.md, HBM: "Other access widths and WRITE_SIZE are uncalibrated: calibrate on a known
// byte count in your own access pattern".)
//
//   hipcc --offload-arch=gfx950 -O2 -o build/pmc_calib scripts/micro/pmc_calib.hip && build/pmc_calib > known.json
//
// Every measured launch is preceded by cal_flush (a 1 GiB read + write sweep), so that it starts with cold L2s and the
// counters see every byte the pattern moves; the patterns are sized like the cfg3 batch (2.8 M rows, ~25 k clusters of
// ~20 rows) so that the line-granular behaviour is the pipeline's.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <set>
#include <vector>
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(err_)); exit(1); } } while (0)

__global__ void cal_flush(uint4* p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { uint4 v = p[i]; v.x += 1; p[i] = v; }
}
// ---- reads (the result is folded into one word per wavefront so that nothing is optimised away; 4 B per 64 lanes written)
__device__ __forceinline__ void sink(int* out, int v)
{
    for (int d = 32; d; d >>= 1) v += __shfl_xor(v, d);
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = v;
}
__global__ void cal_read_stream16(const int4* __restrict__ p, size_t n, int* out)          // 16 B per lane, coalesced
{
    int acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const int4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
    sink(out, acc);
}
__global__ void cal_read_stream4(const int* __restrict__ p, size_t n, int* out)            // 4 B per lane, coalesced
{
    int acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    sink(out, acc);
}
__global__ void cal_read_stride32(const int4* __restrict__ p, size_t n16, int* out)        // k_chain_count: a lane owns 8 consecutive int32 = two 16-byte loads, 32 B apart
{
    int acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; 2 * i + 1 < n16; i += (size_t)gridDim.x * blockDim.x) {
        const int4 v = p[2 * i], w = p[2 * i + 1]; acc += v.x + v.y + v.z + v.w + w.x + w.y + w.z + w.w;
    }
    sink(out, acc);
}
// k_refine_indel_wave: a sub-wave of 32 lanes reads one cluster's rows (4 B per lane) from FOUR columns; the list entry first
__global__ void cal_read_runs4(const int4* __restrict__ list, int nlist, const int* c0, const int* c1, const int* c2, const int* c3, int* out)
{
    const int lane = threadIdx.x & 63, sl = lane & 31;
    int acc = 0;
    for (int u = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; 2 * u < nlist; u += (gridDim.x * blockDim.x) >> 6) {
        const int q = 2 * u + (lane >> 5);
        if (q < nlist) { const int4 e = list[q]; if (sl < e.w) acc += c0[e.z + sl] + c1[e.z + sl] + c2[e.z + sl] + c3[e.z + sl]; }
    }
    sink(out, acc);
}
// ---- writes
__global__ void cal_write_stream4(int* p, size_t n)                                        // 4 B per lane, coalesced (k_chain_ids)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (int)i;
}
__global__ void cal_write_stream16(int4* p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_int4((int)i, 1, 2, 3);
}
// one 64-byte record (four 16-byte stores of ONE lane) per cluster, at rec[slot[q]]: scattered (r03 t_rec[first w]) or dense (t_rec0[item])
__global__ void cal_write_rec64(const int4* __restrict__ list, int nlist, int4* rec, int dense)
{
    const int lane = threadIdx.x & 63, sl = lane & 31;
    for (int u = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; 2 * u < nlist; u += (gridDim.x * blockDim.x) >> 6) {
        const int q = 2 * u + (lane >> 5);
        if (q < nlist && sl == 0) {
            const int4 e = list[q];
            int4* r = rec + 4 * (size_t)(dense ? q : e.z);
            r[0] = e; r[1] = e; r[2] = e; r[3] = e;
        }
    }
}
// 8 B per lane into the cluster's own row range (sup_tmp[first w + i]): runs of ~20 x 8 B at 8-byte alignment
__global__ void cal_write_runs8(const int4* __restrict__ list, int nlist, int2* dst)
{
    const int lane = threadIdx.x & 63, sl = lane & 31;
    for (int u = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; 2 * u < nlist; u += (gridDim.x * blockDim.x) >> 6) {
        const int q = 2 * u + (lane >> 5);
        if (q < nlist) { const int4 e = list[q]; if (sl < e.w) dst[e.z + sl] = make_int2(e.z, sl); }
    }
}
// 8 B per item, item-indexed (item_cnt[j]): one lane per cluster stores, neighbouring items by different wavefronts
__global__ void cal_write_item8(const int4* __restrict__ list, int nlist, long long* dst)
{
    const int lane = threadIdx.x & 63, sl = lane & 31;
    for (int u = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; 2 * u < nlist; u += (gridDim.x * blockDim.x) >> 6) {
        const int q = 2 * u + (lane >> 5);
        if (q < nlist && sl == 0) dst[q] = q;
    }
}

static size_t lines(const std::vector<std::pair<size_t, size_t>>& spans, size_t gran)      // distinct gran-byte blocks touched by [begin, end) byte spans
{
    std::set<size_t> s;
    for (auto& sp : spans) for (size_t b = sp.first / gran; b <= (sp.second - 1) / gran; b++) s.insert(b);
    return s.size() * gran;
}

int main()
{
    const size_t W = 2781852;                 // rows of the cfg3 batch
    const int NCL = 24682;                    // gated clusters
    const size_t FL = 1ull << 30;
    void *flush, *c[4], *out, *dst, *rec, *list_d, *item;
    CK(hipMalloc(&flush, FL));
    for (auto& p : c) { CK(hipMalloc(&p, (W + 64) * 4)); CK(hipMemset(p, 1, (W + 64) * 4)); }
    CK(hipMalloc(&out, 1 << 20)); CK(hipMalloc(&dst, (W + 64) * 8)); CK(hipMalloc(&rec, (W + 64) * 64)); CK(hipMalloc(&item, (size_t)NCL * 8 + 64));
    // clusters: NCL runs of 10..32 rows at random places, ascending, non-overlapping (like gated clusters among singletons)
    std::vector<int> e(4 * (size_t)NCL);
    srand(12345);
    size_t w = 0; const size_t gap = W / NCL;
    std::vector<std::pair<size_t, size_t>> sp4, sp8, sp64s, sp64d, spl, spi;
    size_t rows = 0;
    for (int q = 0; q < NCL; q++) {
        const int m = 10 + rand() % 23;
        const size_t s = w + rand() % (gap - 33);
        e[4 * q] = q; e[4 * q + 1] = 0; e[4 * q + 2] = (int)s; e[4 * q + 3] = m;
        sp4.push_back({s * 4, (s + m) * 4}); sp8.push_back({s * 8, (s + m) * 8});
        sp64s.push_back({s * 64, s * 64 + 64}); sp64d.push_back({(size_t)q * 64, (size_t)q * 64 + 64});
        spl.push_back({(size_t)q * 16, (size_t)q * 16 + 16}); spi.push_back({(size_t)q * 8, (size_t)q * 8 + 8});
        rows += m; w += gap;
    }
    CK(hipMalloc(&list_d, e.size() * 4)); CK(hipMemcpy(list_d, e.data(), e.size() * 4, hipMemcpyHostToDevice));
    const int G = 2048, T = 256;
    auto fl = [&]() { hipLaunchKernelGGL(cal_flush, dim3(4096), dim3(256), 0, 0, (uint4*)flush, FL / 16); };
    const int REP = 5;
    for (int r = 0; r < REP; r++) {
        fl(); hipLaunchKernelGGL(cal_read_stream16, dim3(G), dim3(T), 0, 0, (const int4*)c[0], W / 4, (int*)out);
        fl(); hipLaunchKernelGGL(cal_read_stream4, dim3(G), dim3(T), 0, 0, (const int*)c[1], W, (int*)out);
        fl(); hipLaunchKernelGGL(cal_read_stride32, dim3(G), dim3(T), 0, 0, (const int4*)c[2], W / 4, (int*)out);
        fl(); hipLaunchKernelGGL(cal_read_runs4, dim3(1536), dim3(T), 0, 0, (const int4*)list_d, NCL, (const int*)c[0], (const int*)c[1], (const int*)c[2], (const int*)c[3], (int*)out);
        fl(); hipLaunchKernelGGL(cal_write_stream4, dim3(G), dim3(T), 0, 0, (int*)c[3], W);
        fl(); hipLaunchKernelGGL(cal_write_stream16, dim3(G), dim3(T), 0, 0, (int4*)dst, W / 2);
        fl(); hipLaunchKernelGGL(cal_write_rec64, dim3(1536), dim3(T), 0, 0, (const int4*)list_d, NCL, (int4*)rec, 0);
        fl(); hipLaunchKernelGGL(cal_write_rec64, dim3(1536), dim3(T), 0, 0, (const int4*)list_d, NCL, (int4*)rec, 1);
        fl(); hipLaunchKernelGGL(cal_write_runs8, dim3(1536), dim3(T), 0, 0, (const int4*)list_d, NCL, (int2*)dst);
        fl(); hipLaunchKernelGGL(cal_write_item8, dim3(1536), dim3(T), 0, 0, (const int4*)list_d, NCL, (long long*)item);
    }
    CK(hipDeviceSynchronize());
    // the byte counts each pattern moves, exactly and at the three plausible request granularities.  "launch" = the order of
    // the kernel's launches inside one repetition for kernels that run twice (cal_write_rec64: scattered first, dense second)
    const size_t nw = (size_t)G * T / 64 * 4, nw2 = (size_t)1536 * T / 64 * 4;
    printf("{\n");
    printf(" \"cal_read_stream16\": {\"read_exact\": %zu, \"write_exact\": %zu},\n", W / 4 * 16, nw);
    printf(" \"cal_read_stream4\": {\"read_exact\": %zu, \"write_exact\": %zu},\n", W * 4, nw);
    printf(" \"cal_read_stride32\": {\"read_exact\": %zu, \"write_exact\": %zu},\n", W / 8 * 32, nw);
    printf(" \"cal_read_runs4\": {\"read_exact\": %zu, \"read_32\": %zu, \"read_64\": %zu, \"read_128\": %zu, \"write_exact\": %zu, \"clusters\": %d, \"rows\": %zu},\n",
           rows * 16 + (size_t)NCL * 16, 4 * lines(sp4, 32) + lines(spl, 32), 4 * lines(sp4, 64) + lines(spl, 64), 4 * lines(sp4, 128) + lines(spl, 128), nw2, NCL, rows);
    printf(" \"cal_write_stream4\": {\"write_exact\": %zu},\n", W * 4);
    printf(" \"cal_write_stream16\": {\"write_exact\": %zu},\n", W / 2 * 16);
    printf(" \"cal_write_rec64\": {\"launch0_scattered\": {\"write_exact\": %zu, \"write_64\": %zu, \"write_128\": %zu}, \"launch1_dense\": {\"write_exact\": %zu, \"write_64\": %zu, \"write_128\": %zu}, \"read_exact\": %zu},\n",
           (size_t)NCL * 64, lines(sp64s, 64), lines(sp64s, 128), (size_t)NCL * 64, lines(sp64d, 64), lines(sp64d, 128), (size_t)NCL * 16);
    printf(" \"cal_write_runs8\": {\"write_exact\": %zu, \"write_32\": %zu, \"write_64\": %zu, \"write_128\": %zu, \"read_exact\": %zu},\n", rows * 8, lines(sp8, 32), lines(sp8, 64), lines(sp8, 128), (size_t)NCL * 16);
    printf(" \"cal_write_item8\": {\"write_exact\": %zu, \"write_64\": %zu, \"write_128\": %zu, \"read_exact\": %zu},\n", (size_t)NCL * 8, lines(spi, 64), lines(spi, 128), (size_t)NCL * 16);
    printf(" \"cal_flush\": {\"read_exact\": %zu, \"write_exact\": %zu}\n}\n", FL, FL);
    return 0;
}
