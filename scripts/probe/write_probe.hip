// Store-pattern probe for the assembly kernel's write-out (standalone; hipcc --offload-arch=gfx950 -O3).
// K has the tiled layout [tile of 64 blocks][9][64] doubles; a chunk is 256 consecutive block slots (18 KB).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef double dv2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int64_t tiled(int64_t s, int c) { return ((s >> 6) * 9 + c) * 64 + (s & 63); }

template <int MODE>   // 0: nt, 1: plain
__device__ __forceinline__ void st(dv2 *p, dv2 v) {
    if (MODE == 0) __builtin_nontemporal_store(v, p); else *p = v;
}

// grid-stride fill, 16 B per lane
template <int MODE> __global__ void __launch_bounds__(256) k_fill(double *v, int64_t n2) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (int64_t)gridDim.x * 256) st<MODE>((dv2 *)v + i, dv2{1.0, 2.0});
}
// one workgroup per chunk, chunk start offset by `shift` slots (unaligned chunks like the real ones)
template <int MODE, bool LDS> __global__ void __launch_bounds__(256, 8) k_chunk(double *v, int64_t nChunk, int shift) {
    extern __shared__ double acc[];
    const int64_t chunk = blockIdx.x;
    const int64_t s0 = chunk * 256 + shift * 2;
    if (LDS) {
        for (int t = threadIdx.x; t < 256; t += 256)
            for (int c = 0; c < 9; ++c) acc[c * 258 + t] = 1.0;
        __syncthreads();
    }
    for (int p = threadIdx.x; p < 128; p += 256) {
        const int64_t s = s0 + 2 * p;
        for (int c = 0; c < 9; ++c) {
            dv2 w = LDS ? dv2{acc[c * 258 + 2 * p], acc[c * 258 + 2 * p + 1]} : dv2{1.0, 2.0};
            st<MODE>((dv2 *)&v[tiled(s, c)], w);
        }
    }
}
// same with all 256 lanes storing: lane -> (pair p = t & 127, component half): each lane 4-5 components
template <int MODE> __global__ void __launch_bounds__(256, 8) k_chunk256(double *v, int64_t nChunk, int shift) {
    const int64_t s0 = (int64_t)blockIdx.x * 256 + shift * 2;
    const int p = threadIdx.x & 127, h = threadIdx.x >> 7;
    const int64_t s = s0 + 2 * p;
    for (int c = h; c < 9; c += 2) st<MODE>((dv2 *)&v[tiled(s, c)], dv2{1.0, 2.0});
}
// persistent: grid of G workgroups walking chunks b, b + G, ...
template <int MODE> __global__ void __launch_bounds__(256, 8) k_persist(double *v, int64_t nChunk, int shift) {
    for (int64_t chunk = blockIdx.x; chunk < nChunk; chunk += gridDim.x) {
        const int64_t s0 = chunk * 256 + shift * 2;
        for (int p = threadIdx.x; p < 128; p += 256) {
            const int64_t s = s0 + 2 * p;
            for (int c = 0; c < 9; ++c) st<MODE>((dv2 *)&v[tiled(s, c)], dv2{1.0, 2.0});
        }
    }
}

template <class F> float timeit(F f, int reps = 10) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}

int main() {
    const int64_t nnzb = 201600000;                 // config 3: 14.5 GB
    const int64_t nChunk = nnzb / 256 - 2;
    const size_t bytes = (size_t)((nnzb + 63) / 64 + 2) * 9 * 64 * 8;
    double *v; CK(hipMalloc(&v, bytes));
    const double GB = (double)nChunk * 256 * 72 / 1e9;
    auto rep = [&](const char *name, float ms) { printf("%-44s %.3f ms  %.2f TB/s\n", name, ms, GB / ms); fflush(stdout); };
    rep("fill nt grid-stride (16384 WGs)", timeit([&] { hipLaunchKernelGGL(k_fill<0>, dim3(16384), dim3(256), 0, 0, v, nChunk * 256 * 9 / 2); }));
    rep("fill plain grid-stride (16384 WGs)", timeit([&] { hipLaunchKernelGGL(k_fill<1>, dim3(16384), dim3(256), 0, 0, v, nChunk * 256 * 9 / 2); }));
    rep("fill nt one WG per 4 KB", timeit([&] { hipLaunchKernelGGL(k_fill<0>, dim3((unsigned)(nChunk * 256 * 9 / 2 / 256)), dim3(256), 0, 0, v, nChunk * 256 * 9 / 2); }));
    for (int shift : {0, 5}) {
        printf("-- chunk start shift %d slots\n", shift * 2);
        rep("chunk/WG nt, 128 lanes x 9", timeit([&] { hipLaunchKernelGGL((k_chunk<0, false>), dim3((unsigned)nChunk), dim3(256), 0, 0, v, nChunk, shift); }));
        rep("chunk/WG plain, 128 lanes x 9", timeit([&] { hipLaunchKernelGGL((k_chunk<1, false>), dim3((unsigned)nChunk), dim3(256), 0, 0, v, nChunk, shift); }));
        rep("chunk/WG nt + LDS (18 KB, zero, barrier)", timeit([&] { hipLaunchKernelGGL((k_chunk<0, true>), dim3((unsigned)nChunk), dim3(256), 258 * 72, 0, v, nChunk, shift); }));
        rep("chunk/WG nt, 256 lanes x 4.5", timeit([&] { hipLaunchKernelGGL(k_chunk256<0>, dim3((unsigned)nChunk), dim3(256), 0, 0, v, nChunk, shift); }));
        rep("chunk/WG plain, 256 lanes x 4.5", timeit([&] { hipLaunchKernelGGL(k_chunk256<1>, dim3((unsigned)nChunk), dim3(256), 0, 0, v, nChunk, shift); }));
        rep("persistent nt (2048 WGs)", timeit([&] { hipLaunchKernelGGL(k_persist<0>, dim3(2048), dim3(256), 0, 0, v, nChunk, shift); }));
        rep("persistent nt (8192 WGs)", timeit([&] { hipLaunchKernelGGL(k_persist<0>, dim3(8192), dim3(256), 0, 0, v, nChunk, shift); }));
    }
    return 0;
}
