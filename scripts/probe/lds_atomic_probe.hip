// LDS atomic adds, FP32 against FP64, on one MI355X: the FP32-arithmetic experiment on k_mf_cluster lost 0.6 ms to its 30 ds_add_f32 per
// element where the shipped kernel's 30 ds_add_f64 cost next to nothing. This probe isolates the instruction: 256-lane workgroups, every lane adds to ROWS x 3
// accumulators of a 12 KB table (the cluster kernel's pattern: random rows, three adjacent components), conflict-free and random addressing.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics scripts/probe/lds_atomic_probe.hip -o scripts/probe/bin/lds_atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <class T, bool RANDOM>
__global__ void __launch_bounds__(256) k_add(int reps, const int *__restrict__ rows, T *out) {
    __shared__ T acc[516 * 3];
    for (int t = threadIdx.x; t < 516 * 3; t += 256) acc[t] = 0;
    __syncthreads();
    int li[10];
    for (int j = 0; j < 10; ++j) li[j] = RANDOM ? rows[(blockIdx.x * 256 + threadIdx.x) * 10 + j] : (int)((threadIdx.x * 2 + j * 37) % 516);
    T v = (T)(threadIdx.x + 1);
    for (int r = 0; r < reps; ++r)
#pragma unroll
        for (int j = 0; j < 10; ++j)
#pragma unroll
            for (int d = 0; d < 3; ++d) unsafeAtomicAdd(&acc[li[j] * 3 + d], v);
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = acc[7];
}
template <class T, bool RANDOM> static int run(const char *name, const int *dRows) {
    const int grid = 256 * 8, reps = 200;
    T *out; CK(hipMalloc(&out, grid * sizeof(T)));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k_add<T, RANDOM>), dim3(grid), dim3(256), 0, 0, 2, dRows, out);
    CK(hipEventRecord(a, 0));
    hipLaunchKernelGGL((k_add<T, RANDOM>), dim3(grid), dim3(256), 0, 0, reps, dRows, out);
    CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    const double ops = (double)grid * 256 * reps * 30;
    printf("%-34s %8.3f ms for %.2e lane-adds = %7.1f G adds/s (%.2f adds per CU and clock at 2.4 GHz, 256 CUs)\n", name, ms, ops, ops / ms * 1e-6, ops / (ms * 1e-3) / 256 / 2.4e9);
    CK(hipFree(out));
    return 0;
}
int main() {
    CK(hipSetDevice(0));
    std::vector<int> rows((size_t)256 * 8 * 256 * 10);
    unsigned s = 12345;
    for (auto &r : rows) { s = s * 1664525u + 1013904223u; r = (int)((s >> 8) % 516); }
    int *dRows; CK(hipMalloc(&dRows, rows.size() * sizeof(int)));
    CK(hipMemcpy(dRows, rows.data(), rows.size() * sizeof(int), hipMemcpyHostToDevice));
    if (run<double, false>("ds_add_f64, strided rows", dRows)) return 1;
    if (run<float, false>("ds_add_f32, strided rows", dRows)) return 1;
    if (run<double, true>("ds_add_f64, random rows", dRows)) return 1;
    if (run<float, true>("ds_add_f32, random rows", dRows)) return 1;
    return 0;
}
