// Calibration of rocprofv3's FETCH_SIZE for BYTE GATHERS on gfx950 (MI355X_MICROARCH.md, "HBM": "other access widths
// are uncalibrated: calibrate on a known byte count in your own access pattern").  Three launches over a 2 GiB buffer
// (8x the Infinity Cache), each lane loading ONE byte:
//   stride128: lane i reads byte 128*i            -> every lane its own 128-byte line   (16.8 M lines)
//   stride64:  lane i reads byte 64*i             -> every lane its own 64-byte sector  (33.5 M sectors, 2 per line)
//   stride4k:  lane i reads byte 4096*i + 64*(i%64)  -> scattered, one sector per 4 KiB page (0.5 M sectors)
// Build and run on the GPU box:  hipcc --offload-arch=gfx950 -O2 fetch_calib.hip -o fetch_calib && ./fetch_calib
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__global__ void gather_bytes(const uint8_t* __restrict__ buf, uint32_t* __restrict__ sink, uint64_t stride, uint64_t extra_mod, uint64_t n)
{
    const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t off = i * stride + (extra_mod ? 64ull * (i % extra_mod) : 0ull);
    const uint32_t v = buf[off];
    if (v == 0x5au) sink[0] = v;      // never true for the zero-filled buffer: keeps the load alive
}

int main()
{
    const uint64_t bytes = 2ull << 30;
    uint8_t* buf = nullptr;
    uint32_t* sink = nullptr;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { std::printf("hipMalloc failed\n"); return 1; }
    (void)hipMemset(buf, 0, bytes);
    (void)hipMemset(sink, 0, 64);
    (void)hipDeviceSynchronize();
    struct Case { const char* name; uint64_t stride, extra_mod; } cases[3] = {{"stride128", 128, 0}, {"stride64", 64, 0}, {"stride4k", 4096, 64}};
    for (const Case& c : cases) {
        const uint64_t n = bytes / c.stride;
        hipLaunchKernelGGL(gather_bytes, dim3(static_cast<uint32_t>((n + 255) / 256)), dim3(256), 0, 0, buf, sink, c.stride, c.extra_mod, n);
        (void)hipDeviceSynchronize();
        std::printf("%s: %llu lanes, one byte each\n", c.name, static_cast<unsigned long long>(n));
    }
    (void)hipFree(buf); (void)hipFree(sink);
    return 0;
}
