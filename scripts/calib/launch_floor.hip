// Launch floor of back-to-back kernels on gfx950 by workgroup shape, LDS size and kernel-argument size (development aid).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

struct Big { float v[225]; };   // ~900 bytes, like FrameParams

template <int LDS>
__global__ __launch_bounds__(1024) void k_small(uint32_t* out, uint32_t x)
{
    extern __shared__ uint8_t dyn[];
    if (x == 0xdeadbeefu) { dyn[threadIdx.x] = 1; out[0] = dyn[5]; }
}
__global__ __launch_bounds__(1024) void k_big(uint32_t* out, uint32_t x, const Big b)
{
    extern __shared__ uint8_t dyn[];
    if (x == 0xdeadbeefu) { dyn[threadIdx.x] = 1; out[0] = dyn[5] + (uint32_t)b.v[threadIdx.x % 225]; }
}
__global__ __launch_bounds__(1024) void k_big_used(uint32_t* out, uint32_t x, const Big b)
{
    extern __shared__ uint8_t dyn[];
    float s = 0;
    for (int i = 0; i < 225; ++i) s += b.v[i];       // touches every kernel argument (scalar loads)
    if (x == 0xdeadbeefu || s == 12345.0f) { dyn[threadIdx.x] = 1; out[0] = dyn[5]; }
}

int main()
{
    uint32_t* d = nullptr;
    (void)hipMalloc(&d, 64);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipFuncSetAttribute((const void*)k_small<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)k_big, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)k_big_used, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    Big b; for (int i = 0; i < 225; ++i) b.v[i] = (float)i;
    const int N = 300;
    auto run = [&](const char* name, int grid, int threads, int lds, int kind) {
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(e0, 0);
            for (int i = 0; i < N; ++i) {
                if (kind == 0) hipLaunchKernelGGL(k_small<0>, dim3(grid), dim3(threads), lds, 0, d, 1u);
                else if (kind == 1) hipLaunchKernelGGL(k_big, dim3(grid), dim3(threads), lds, 0, d, 1u, b);
                else hipLaunchKernelGGL(k_big_used, dim3(grid), dim3(threads), lds, 0, d, 1u, b);
            }
            (void)hipEventRecord(e1, 0);
            (void)hipDeviceSynchronize();
            float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep == 1) std::printf("%-28s grid %4d x %4d threads, LDS %6d B: %.2f us per launch\n", name, grid, threads, lds, 1e3 * ms / N);
        }
    };
    run("small args", 256, 256, 0, 0);
    run("small args", 256, 1024, 0, 0);
    run("small args", 256, 1024, 64 * 1024, 0);
    run("small args", 256, 1024, 139 * 1024, 0);
    run("small args", 256, 512, 139 * 1024, 0);
    run("900-byte args (unused)", 256, 1024, 139 * 1024, 1);
    run("900-byte args (all read)", 256, 1024, 139 * 1024, 2);
    run("900-byte args (all read)", 256, 1024, 0, 2);
    run("small args", 1024, 256, 0, 0);
    run("small args", 8160, 256, 0, 0);
    return 0;
}
