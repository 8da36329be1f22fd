// The step after the path: compute output -> presentation target (gfx950).
//
// Reference: src/render_pipeline.rs:88-130 draws a full-screen quad whose fragment shader (shaders/render.wgsl:39-43) is
//     uv = frag_coord.xy / vec2<f32>(textureDimensions(input_texture));  return textureSample(input_texture, input_sampler, uv);
// with a Linear / ClampToEdge sampler (src/gpu_resources/texture.rs:84-101) and BlendState::REPLACE into an rgba8unorm
// target.  frag_coord is the TARGET pixel centre (x + 0.5, y + 0.5) and the divisor is the INPUT size: the pass maps
// pixels 1:1 (it does not rescale) -- identical sizes copy the frame, a larger target repeats the edge texels, a smaller
// one crops -- and the bilinear weights are whatever f32 rounding leaves of "exactly on a texel centre".
// EXACT arithmetic (plain IEEE f32, no fma; the library is built with -ffp-contract=off): the oracle restates the same
// operations (oracle/volym_oracle.c vo_blit) and the two agree bit for bit.
#pragma once

#include <hip/hip_runtime.h>

namespace volym {

__device__ __forceinline__ void blit_axis(float frag, float fn, int n, int& i0, int& i1, float& w)
{
    const float u = frag / fn;                  // wgsl:41
    const float x = u * fn - 0.5f;              // Vulkan linear filter: texel space, centre convention
    const float fl = __builtin_floorf(x);
    w = x - fl;
    const int i = static_cast<int>(fl);
    i0 = min(max(i, 0), n - 1);                 // ClampToEdge
    i1 = min(max(i + 1, 0), n - 1);
}

__global__ __launch_bounds__(256) void volym_blit_kernel(const uint32_t* __restrict__ src, uint32_t in_w, uint32_t in_h,
                                                         uint32_t* __restrict__ dst, uint32_t out_w, uint32_t out_h)
{
    const uint32_t x = blockIdx.x * 64u + threadIdx.x, y = blockIdx.y * 4u + threadIdx.y;
    if (x >= out_w || y >= out_h) return;
    int x0, x1, y0, y1;
    float wx, wy;
    blit_axis(static_cast<float>(x) + 0.5f, static_cast<float>(in_w), static_cast<int>(in_w), x0, x1, wx);
    blit_axis(static_cast<float>(y) + 0.5f, static_cast<float>(in_h), static_cast<int>(in_h), y0, y1, wy);
    const uint32_t t00 = src[static_cast<size_t>(y0) * in_w + x0], t10 = src[static_cast<size_t>(y0) * in_w + x1];
    const uint32_t t01 = src[static_cast<size_t>(y1) * in_w + x0], t11 = src[static_cast<size_t>(y1) * in_w + x1];
    uint32_t out = 0;
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) {
        const float a = static_cast<float>((t00 >> (8 * ch)) & 255u) / 255.0f, b = static_cast<float>((t10 >> (8 * ch)) & 255u) / 255.0f;
        const float c = static_cast<float>((t01 >> (8 * ch)) & 255u) / 255.0f, d = static_cast<float>((t11 >> (8 * ch)) & 255u) / 255.0f;
        const float top = a * (1.0f - wx) + b * wx, bot = c * (1.0f - wx) + d * wx;      // x first, then y
        const float v = top * (1.0f - wy) + bot * wy;
        // rgba8unorm store: clamp, scale, round to nearest
        uint32_t q = !(v > 0.0f) ? 0u : (v >= 1.0f ? 255u : static_cast<uint32_t>(__builtin_floorf(v * 255.0f + 0.5f)));
        out |= q << (8 * ch);
    }
    dst[static_cast<size_t>(y) * out_w + x] = out;
}

}  // namespace volym
