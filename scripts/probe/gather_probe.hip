// FETCH_SIZE calibration on GATHER patterns: rocprofv3's FETCH_SIZE on gfx950 was calibrated on k_axpby's
// 8-byte-per-lane streaming reads (reported / true = ~0.5); the assembly kernel reads its element records in 8..24-byte pieces, one
// 128-byte record per lane, and its lists as coalesced streams. This standalone program issues reads of KNOWN line counts:
//   k_stream8    lane i reads double i                       (coalesced: the pattern the factor was calibrated on)
//   k_gather8    lane i reads ONE double of line perm[i]     (one 8-byte piece per 128-byte line, lines in random order)
//   k_strided24  lane i reads 3 doubles of line perm[i]      (24 bytes per 128-byte line)
//   k_record128  lane i reads all 16 doubles of line perm[i] in 8-byte loads (what a lane does with its element record)
//   k_record128s the same with lines in sequential order     (records of consecutive elements)
// Every kernel touches nLines distinct 128-byte lines of a buffer far larger than the 256 MiB memory-side cache, once.
//   rocprofv3 --pmc FETCH_SIZE -- scripts/probe/bin/gather_probe     (counter per kernel / nLines = bytes counted per line touched)
//   hipcc --offload-arch=gfx950 -O3 scripts/probe/gather_probe.hip -o scripts/probe/bin/gather_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <random>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) k_stream8(const double *__restrict__ a, int64_t n, double *out) {
    double s = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) s += a[i];
    if (s == 12345.678) out[0] = s;
}
template <int PIECES>
__global__ void __launch_bounds__(256) k_lines(const double *__restrict__ a, const int32_t *__restrict__ perm, int64_t nLines, double *out) {
    double s = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nLines; i += (int64_t)gridDim.x * 256) {
        const double *g = a + (int64_t)(perm ? perm[i] : (int32_t)i) * 16;
#pragma unroll
        for (int k = 0; k < PIECES; ++k) s += g[k];
    }
    if (s == 12345.678) out[0] = s;
}

int main() {
    const int64_t nLines = (int64_t)1 << 24;            // 16 M lines x 128 B = 2 GiB
    double *a, *out;
    int32_t *perm;
    CK(hipMalloc(&a, nLines * 128)); CK(hipMalloc(&out, 64)); CK(hipMalloc(&perm, nLines * 4));
    CK(hipMemset(a, 0, nLines * 128));
    std::vector<int32_t> h(nLines);
    std::iota(h.begin(), h.end(), 0);
    std::mt19937_64 rng(0);
    std::shuffle(h.begin(), h.end(), rng);
    CK(hipMemcpy(perm, h.data(), nLines * 4, hipMemcpyHostToDevice));
    const int grid = 256 * 32;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timed = [&](const char *name, auto launch, double bytesTrue) {
        launch();                      // warm-up (also a profiled dispatch: the counters are averaged per kernel name)
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-14s %8.3f ms   %7.1f GB/s on the lines touched (%.0f MB)\n", name, ms, bytesTrue / ms / 1e6, bytesTrue / 1e6);
    };
    timed("k_stream8", [&] { hipLaunchKernelGGL(k_stream8, dim3(grid), dim3(256), 0, 0, a, nLines * 16, out); }, nLines * 128.0);
    timed("k_gather8", [&] { hipLaunchKernelGGL(k_lines<1>, dim3(grid), dim3(256), 0, 0, a, perm, nLines, out); }, nLines * 128.0);
    timed("k_strided24", [&] { hipLaunchKernelGGL(k_lines<3>, dim3(grid), dim3(256), 0, 0, a, perm, nLines, out); }, nLines * 128.0);
    timed("k_record128", [&] { hipLaunchKernelGGL(k_lines<16>, dim3(grid), dim3(256), 0, 0, a, perm, nLines, out); }, nLines * 128.0);
    timed("k_record128s", [&] { hipLaunchKernelGGL(k_lines<16>, dim3(grid), dim3(256), 0, 0, a, (const int32_t *)nullptr, nLines, out); }, nLines * 128.0);
    printf("{\"n_lines\": %lld, \"line_bytes\": 128}\n", (long long)nLines);
    return 0;
}
