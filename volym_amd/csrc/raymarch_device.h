// Device-side building blocks of the ray-march (gfx950 only).
//
// Two kinds of arithmetic live here and must not be mixed up:
//   EXACT   -- everything that feeds a discrete decision (voxel index, rho >= threshold,
//              importance tests, the alpha < 0.95 exit, the step state machine).  Plain IEEE
//              f32 in the order the WGSL states it; the file is compiled with
//              -ffp-contract=off so no mul+add is fused behind our back.
//   COLOUR  -- shading that only moves the output continuously (normalisations, Blinn-Phong,
//              colour accumulation).  Free to use v_rsq_f32 and explicit fma; the error budget
//              is 1e-4 per channel (BASELINE.json), the observed error ~1e-6.
//
// "wgsl" = /root/reference/shaders/importance_driven_volume_rendering.wgsl.
#pragma once

#include <hip/hip_runtime.h>

#include "wgsl_math.h"

namespace volym {

enum : uint32_t {
    F_CONE = 1u << 0,          // use_cone_importance_check
    F_IMP_COLORING = 1u << 1,  // use_importance_coloring
    F_OPACITY = 1u << 2,       // use_opacity
    F_IMP_RENDERING = 1u << 3, // use_importance_rendering
    F_GAUSSIAN = 1u << 4,      // use_gaussian_smoothing
    F_LINEAR = 1u << 5,        // VOLYM_FILTER_LINEAR
    F_WRITE_F32 = 1u << 6,
    F_RASTER = 1u << 7,        // world == 1: store straight into the W x H raster
};

struct FrameParams {
    float ivp[16];   // inverse_view_proj, column-major
    float eye[3];
    float thr;       // density_threshold
    float base_step; // raymarching_step_size
    float min_step;  // base_step * 0.25
    float alpha_y;   // min_step * 100: exponent of the opacity correction (wgsl:314)
    uint32_t flags;
    uint32_t ahead_steps;
    uint32_t W, H;
    uint32_t nx, ny, nz;
    uint32_t tiles_x, n_tiles;       // 16x16 tiles of the whole frame
    uint32_t rank, world, n_local;   // tile k is ours when k % world == rank; local index k / world
    uint32_t thr_byte;               // smallest b with b/255 >= thr (256 when none)
    uint32_t tf_n;
    uint32_t mc_n;                   // macro cells per axis
    uint32_t xcd_bands;              // block -> tile remap granularity (0 = identity)
    float gauss_w[5];
    float cone_cos[8], cone_sin[8];
    // ---- exact culling (VARIANT 2; see raymarch_pq.h) ----
    uint32_t cull;           // CULL_* bits
    float hull[2][8][4];     // [0] projected unit cube, [1] projected AABB of the occupied macro cells:
                             // up to 8 edges (a, b, c, valid) in pixel units, |(a,b)| = 1, inside: a*x + b*y + c >= 0
    float aabb_lo[3], aabb_hi[3];   // AABB of the occupied macro cells, already grown by its safety margin
    float imp_lo[3], imp_hi[3];     // positions whose nearest importance texel can be >= 128 (look-ahead probes): the AABB of those
                                    // texels in texture coordinates, open-ended where it touches the border (ClampToEdge), with its
                                    // margin; lo > hi: no such texel
    uint32_t dev;            // timing experiments (VOLYM_DEV_SWITCHES): 1 = drop queued samples unshaded, 2 = never leap in dp items
    uint32_t mask_t8x;       // 8x8 tiles per row of tile_mask
    const uint32_t* tile_mask;   // CULL_TILE_MASK: one bit per 8x8 pixel tile of the frame: set when the projection of some occupied macro
                                 // cell covers a pixel of it (volym_tile_mask_kernel, once per view); a clear bit: no ray of the tile can
                                 // meet anything dense
    uint32_t* tile_mask_spare;   // the mask buffer this view does not use: every launch zeroes it for the next view's mask kernel
    uint32_t mask_words;
    float rcp_w, rcp_h;          // RN(1 / W), RN(1 / H): make_ray's pixel quotients (raymarch_device.h div_pixel)
    float setup_lo;              // 2^-40 when make_ray may share reciprocals between its divisions, +inf when it must not (volym_update)
    uint32_t rect[4];            // variant 3: the screen rectangle {x0, y0, x1, y1} (pixels, whole 64x32 superblocks, x1/y1 may pass the frame) outside of
                                 // which no ray can meet anything dense; x1 <= x0: no such pixel at all
};

#ifndef VOLYM_DEV_SWITCHES
#define VOLYM_DEV_SWITCHES 0     // make DEV=1 compiles the FrameParams::dev timing experiments in (scripts/ablate.py --dev)
#endif
// The distance field is read from LDS: 32^3 cells of 4 bits are the 16 KB it has there.  A finer grid (64^3, read from global memory
// through L1 / L2) was measured slower and exists in the development build only: in the product kernels the test
// folds away (it cost the common instantiation 2 % as a run-time branch in the leap look-up).
#define VOLYM_DF_IN_LDS(fp) (!VOLYM_DEV_SWITCHES || (fp).mc_n <= 32u)
// LDS bytes of the packed 4-bit distance field for the largest macro grid (32^3 cells)
#define VOLYM_DF_LDS_BYTES 16384

enum : uint32_t {
    CULL_CUBE_HULL = 1u << 0,   // hull[0] is usable (every cube corner in front of the eye)
    CULL_OBJ_HULL = 1u << 1,    // hull[1] is usable
    CULL_AABB = 1u << 2,        // aabb_lo/hi are valid: no sample outside it can reach the threshold
    CULL_NOTHING_DENSE = 1u << 3,   // no macro cell can reach the threshold at all
    CULL_TILE_MASK = 1u << 4,   // tile_mask is valid for this view
};

// Per-(transfer function, parameters) tables, built on the host with the same wgsl_math.h
// recipe and staged into LDS by every workgroup.
struct FrameTables {
    float4 tf_tab[256];   // nearest, unsmoothed: rgb = TF(b/255), w = 1 - pow(1 - A, alpha_y)
    float4 lut_f[256];    // decoded LUT texels (continuous-rho modes)
    float ic_alpha[256];  // importance colouring: 1 - pow(1 - i/255, alpha_y)
    float rho[256];       // b / 255
};

struct Counters {
    unsigned long long n_vol, n_imp, n_steps, n_dense, n_hit;
};

struct V3 {
    float x, y, z;
};
__device__ __forceinline__ V3 v3(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ V3 operator/(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
// EXACT: (x*x' + y*y') + z*z'
__device__ __forceinline__ float dot_exact(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b)
{
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ float length_exact(V3 a) { return __builtin_sqrtf(dot_exact(a, a)); }
// EXACT WGSL normalize: v / length(v) (IEEE sqrt and divisions)
__device__ __forceinline__ V3 normalize_exact(V3 a) { return a / length_exact(a); }
// COLOUR: fused dot and hardware reciprocal square root
__device__ __forceinline__ float dot_fast(V3 a, V3 b)
{
    return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x));
}

// Volume layout in HBM, chosen per volume on the host (volym_set_volume):
//   linear   x + nx*(y + ny*z): two 24-bit multiply-adds per address.  Best while the volume is cache-resident (a
//            256^3 march is bound by instruction issue, not by L2 misses: bricks cost ~9 more integer instructions per
//            address and gained nothing, 38.1 vs 36.6 us);
//   bricked  4x4x4 bricks of 64 bytes (brick index x fastest, then y, then z; inside a brick x fastest): the 8x8-pixel
//            footprint of a wave and the +-y/+-z gradient taps fall into the same 64-byte sectors.  Beyond the L2s the
//            march is bound by sector fetches: 512^3 at 1080p 54.3 -> 39.7 us, 1024^3 at 4K 238 -> 121 us (and with
//            the importance volume beside it, BASELINE configs[4] on one GPU, 584 -> 284 us).
__host__ __device__ inline uint32_t brick_count(uint32_t n) { return (n + 3u) >> 2; }

template <bool BRICK>
struct GridT {
    static constexpr bool bricked = BRICK;
    const uint8_t* __restrict__ vol;
    const uint8_t* __restrict__ imp;
    int nx, ny, nz;
    uint32_t bx, bxy;         // bricks per row, bricks per slab (bricked) or nx, nx*ny (linear)
    float fnx, fny, fnz;      // (float)n
    float hix, hiy, hiz;      // (float)(n - 1)
};
typedef GridT<false> Grid;

__host__ __device__ inline uint32_t layout_offset(bool bricked, uint32_t bx, uint32_t bxy, uint32_t ix, uint32_t iy, uint32_t iz)
{
    if (bricked) return (((iz >> 2) * bxy + (iy >> 2) * bx + (ix >> 2)) << 6) | ((iz & 3u) << 4) | ((iy & 3u) << 2) | (ix & 3u);
    return ix + bx * iy + bxy * iz;
}
__host__ __device__ inline uint32_t layout_bx(bool bricked, uint32_t nx) { return bricked ? brick_count(nx) : nx; }
__host__ __device__ inline uint32_t layout_bxy(bool bricked, uint32_t nx, uint32_t ny) { return bricked ? brick_count(nx) * brick_count(ny) : nx * ny; }

template <class G>
__device__ __forceinline__ void grid_init(G& g, const uint8_t* vol, const uint8_t* imp, uint32_t nx, uint32_t ny, uint32_t nz)
{
    g.vol = vol; g.imp = imp;
    g.nx = static_cast<int>(nx); g.ny = static_cast<int>(ny); g.nz = static_cast<int>(nz);
    g.bx = layout_bx(G::bricked, nx); g.bxy = layout_bxy(G::bricked, nx, ny);
    g.fnx = static_cast<float>(nx); g.fny = static_cast<float>(ny); g.fnz = static_cast<float>(nz);
    g.hix = static_cast<float>(nx - 1u); g.hiy = static_cast<float>(ny - 1u); g.hiz = static_cast<float>(nz - 1u);
}

// nearest filter + ClampToEdge: i = clamp(floor(u * n), 0, n - 1)  (EXACT)
// v_cvt_flr_i32_f32 is floor-and-convert in one instruction (saturating, NaN -> 0), the clamp then happens on
// integers (v_med3_i32): the same value as clamping the floored float and converting, for every input
// (|x| >= 2^31 saturates and clamps to an end, NaN gives 0 on both routes), in 3 instructions instead of 6-7.
__device__ __forceinline__ int floor_to_int(float x)
{
    int r;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}

// clamp(i, 0, hi) for hi >= 0 as one median-of-three
__device__ __forceinline__ int clamp_texel(int i, int hi)
{
    int r;
    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(i), "v"(hi));
    return r;
}

// a * b + c on the low 24 bits of a and b (full rate; the compiler's own choice for this pattern was the
// quarter-rate 64-bit v_mad_u64_u32, whose register pair also tied independent loads together)
__device__ __forceinline__ uint32_t mad_u24(uint32_t a, uint32_t b_uniform, uint32_t c)
{
    uint32_t r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b_uniform), "v"(c));
    return r;
}

__device__ __forceinline__ int texel_nearest(float u, float fn, float hi)
{
    return clamp_texel(floor_to_int(u * fn), static_cast<int>(hi));
}

template <class G>
__device__ __forceinline__ uint32_t voxel_offset(const G& g, int ix, int iy, int iz)
{
    if constexpr (G::bricked) {
        const uint32_t x = static_cast<uint32_t>(ix), y = static_cast<uint32_t>(iy), z = static_cast<uint32_t>(iz);
        const uint32_t brick = mad_u24(z >> 2, g.bxy, mad_u24(y >> 2, g.bx, x >> 2));      // < 2^24 bricks: volumes below 2^30 voxels
        return (brick << 6) | ((z & 3u) << 4) | ((y & 3u) << 2) | (x & 3u);
    } else {
        // x + nx*(y + ny*z) with two full-rate 24-bit multiplies: ny*z + y <= 4096*4095 + 4095 < 2^24 for every allowed size
        return mad_u24(mad_u24(static_cast<uint32_t>(iz), static_cast<uint32_t>(g.ny), static_cast<uint32_t>(iy)), static_cast<uint32_t>(g.nx), static_cast<uint32_t>(ix));
    }
}

template <class G>
__device__ __forceinline__ uint32_t nearest_offset(const G& g, V3 p)
{
    return voxel_offset(g, texel_nearest(p.x, g.fnx, g.hix), texel_nearest(p.y, g.fny, g.hiy),
                        texel_nearest(p.z, g.fnz, g.hiz));
}

// linear filter + ClampToEdge along one axis: x = u*n - 0.5, i0 = floor(x), w = x - i0  (EXACT)
__device__ __forceinline__ void texel_linear(float u, float fn, int n, int& i0, int& i1, float& w)
{
    const float x = u * fn - 0.5f;
    float fl = __builtin_floorf(x);
    w = x - fl;
    fl = __builtin_fminf(__builtin_fmaxf(fl, -2.0f), fn);
    const int i = static_cast<int>(fl);
    i0 = min(max(i, 0), n - 1);
    i1 = min(max(i + 1, 0), n - 1);
}

// trilinear density in [0,1]  (EXACT: x, then y, then z; a*(1-w) + b*w)
struct __attribute__((packed, aligned(1))) UnalignedU16 { uint16_t v; };
__device__ __forceinline__ uint32_t load_voxel_pair(const uint8_t* p)
{
    return reinterpret_cast<const UnalignedU16*>(p)->v;
}

template <class G>
__device__ __forceinline__ float fetch_linear(const G& g, const float* __restrict__ s_rho, V3 p)
{
    int x0, x1, y0, y1, z0, z1;
    float fx, fy, fz;
    texel_linear(p.x, g.fnx, g.nx, x0, x1, fx);
    texel_linear(p.y, g.fny, g.ny, y0, y1, fy);
    texel_linear(p.z, g.fnz, g.nz, z0, z1, fz);
    float t000, t100, t010, t110, t001, t101, t011, t111;
    if constexpr (G::bricked) {
    t000 = s_rho[g.vol[voxel_offset(g, x0, y0, z0)]];
    t100 = s_rho[g.vol[voxel_offset(g, x1, y0, z0)]];
    t010 = s_rho[g.vol[voxel_offset(g, x0, y1, z0)]];
    t110 = s_rho[g.vol[voxel_offset(g, x1, y1, z0)]];
    t001 = s_rho[g.vol[voxel_offset(g, x0, y0, z1)]];
    t101 = s_rho[g.vol[voxel_offset(g, x1, y0, z1)]];
    t011 = s_rho[g.vol[voxel_offset(g, x0, y1, z1)]];
    t111 = s_rho[g.vol[voxel_offset(g, x1, y1, z1)]];
    } else {
    // x1 is x0 + 1, or x0 itself at either edge of the row: the two texels of a row are one (unaligned) 16-bit gather
    // instead of two byte gathers.  The volume is allocated with 16 bytes to spare.
    const bool same_x = x1 == x0;
    const uint32_t p00 = load_voxel_pair(g.vol + voxel_offset(g, x0, y0, z0)), p10 = load_voxel_pair(g.vol + voxel_offset(g, x0, y1, z0));
    const uint32_t p01 = load_voxel_pair(g.vol + voxel_offset(g, x0, y0, z1)), p11 = load_voxel_pair(g.vol + voxel_offset(g, x0, y1, z1));
    t000 = s_rho[p00 & 255u], t100 = s_rho[same_x ? (p00 & 255u) : (p00 >> 8)];
    t010 = s_rho[p10 & 255u], t110 = s_rho[same_x ? (p10 & 255u) : (p10 >> 8)];
    t001 = s_rho[p01 & 255u], t101 = s_rho[same_x ? (p01 & 255u) : (p01 >> 8)];
    t011 = s_rho[p11 & 255u], t111 = s_rho[same_x ? (p11 & 255u) : (p11 >> 8)];
    }
    const float c00 = t000 * (1.0f - fx) + t100 * fx;
    const float c10 = t010 * (1.0f - fx) + t110 * fx;
    const float c01 = t001 * (1.0f - fx) + t101 * fx;
    const float c11 = t011 * (1.0f - fx) + t111 * fx;
    const float c0 = c00 * (1.0f - fy) + c10 * fy;
    const float c1 = c01 * (1.0f - fy) + c11 * fy;
    return c0 * (1.0f - fz) + c1 * fz;
}

template <class G>
__device__ __forceinline__ float sample_density(const G& g, const float* __restrict__ s_rho, bool linear, V3 p)
{
    if (linear) return fetch_linear(g, s_rho, p);
    return s_rho[g.vol[nearest_offset(g, p)]];
}

__device__ __forceinline__ bool outside01(V3 p)
{
    return (p.x < 0.0f) | (p.y < 0.0f) | (p.z < 0.0f) | (p.x > 1.0f) | (p.y > 1.0f) | (p.z > 1.0f);
}

// wgsl:52-75, 5 taps along the ray, out-of-volume taps skipped  (EXACT)
template <bool COUNT, class G>
__device__ __forceinline__ float sample_density_smoothed(const G& g, const float* __restrict__ s_rho,
                                                         bool linear, const FrameParams& fp, V3 pos, V3 dir,
                                                         uint32_t& n_vol)
{
    float sum = 0.0f, wsum = 0.0f;
#pragma unroll
    for (int i = -2; i <= 2; ++i) {
        const float offset = static_cast<float>(i) * 0.005f;
        const V3 sp = pos + dir * offset;
        if (outside01(sp)) continue;
        const float w = fp.gauss_w[i + 2];
        const float s = sample_density(g, s_rho, linear, sp);
        if (COUNT) n_vol++;
        sum += s * w;
        wsum += w;
    }
    return sum / wsum;
}

// transfer function, Linear/ClampToEdge, continuous rho  (EXACT in alpha, colour follows)
__device__ __forceinline__ float4 sample_tf(const float4* __restrict__ s_lut, uint32_t tf_n, float u)
{
    int i0, i1;
    float w;
    texel_linear(u, static_cast<float>(tf_n), static_cast<int>(tf_n), i0, i1, w);
    const float4 a = s_lut[i0], b = s_lut[i1];
    const float iw = 1.0f - w;
    return make_float4(a.x * iw + b.x * w, a.y * iw + b.y * w, a.z * iw + b.z * w, a.w * iw + b.w * w);
}

// Look-ahead probes (wgsl:94-160).  The probe positions follow pos += dir*step whatever the fetched values
// are, so the importance bytes are gathered PQ_PROBE_BATCH at a time (one memory round trip per batch
// instead of one per probe) and then examined in order: same answer, and the reference-fetch count stops at
// the probe on which the shader would have returned.
#define VOLYM_PROBE_BATCH 4

// wgsl:141-160  (EXACT)
template <bool COUNT, class G>
__device__ __forceinline__ bool ahead_straight(const G& g, const FrameParams& fp, V3 cur, V3 dir, float t_exit,
                                               uint32_t& n_imp)
{
    V3 pos = cur;
    const int n = static_cast<int>(fp.ahead_steps);
    const float step = (t_exit - length_exact(cur)) / static_cast<float>(n);
    for (int i = 0; i < n; i += VOLYM_PROBE_BATCH) {
        uint32_t ib[VOLYM_PROBE_BATCH];
#pragma unroll
        for (int j = 0; j < VOLYM_PROBE_BATCH; ++j) {
            pos = pos + dir * step;
            ib[j] = (i + j < n) ? g.imp[nearest_offset(g, pos)] : 0u;
        }
#pragma unroll
        for (int j = 0; j < VOLYM_PROBE_BATCH; ++j) {
            if (i + j < n) {
                if (COUNT) n_imp++;
                if (ib[j] >= 128u) return true;   // i/255 >= 0.5  <=>  i >= 128
            }
        }
    }
    return false;
}

// Can the look-ahead of the sample at p0 meet an important voxel at all?  Its probes sit at p0 + i * (d * step), i = 1..N,
// step = (t_exit - |p0|) / N (wgsl:111, :144): on the segment from p0 to p0 + d * (t_exit - |p0|), up to the rounding of N
// accumulated additions; the cone's eight directions deviate from d by at most 0.2 in length (wgsl:100-113), so their
// segments end within 0.2 |L| of that end point.  If the bounding box of all that misses the box of the positions that map
// to important texels (FrameParams::imp_lo/hi), every probe reads an importance < 128 and the shader's loop returns false:
// the probes need not be walked.  Conservative by construction (any NaN compares false: not rejected); EXACT results.
__device__ __forceinline__ bool ahead_cannot_hit(const FrameParams& fp, V3 p0, V3 d, float t_exit, bool cone)
{
    const float len = __builtin_sqrtf(__builtin_fmaf(p0.x, p0.x, __builtin_fmaf(p0.y, p0.y, p0.z * p0.z)));
    const float L = t_exit - len;
    const float aL = __builtin_fabsf(L);
    const V3 pe = v3(__builtin_fmaf(d.x, L, p0.x), __builtin_fmaf(d.y, L, p0.y), __builtin_fmaf(d.z, L, p0.z));
    // rounding of the accumulated probe positions (|coordinates| <= len + |L| + 1) and of L itself, plus the cone's spread
    const float slop = (len + aL + 1.0f) * (static_cast<float>(fp.ahead_steps) + 8.0f) * 2.4e-7f + 1.0e-6f + (cone ? 0.21f * aL : 0.0f);
    const bool miss = (__builtin_fmaxf(p0.x, pe.x) + slop < fp.imp_lo[0]) | (__builtin_fminf(p0.x, pe.x) - slop > fp.imp_hi[0]) |
                      (__builtin_fmaxf(p0.y, pe.y) + slop < fp.imp_lo[1]) | (__builtin_fminf(p0.y, pe.y) - slop > fp.imp_hi[1]) |
                      (__builtin_fmaxf(p0.z, pe.z) + slop < fp.imp_lo[2]) | (__builtin_fminf(p0.z, pe.z) - slop > fp.imp_hi[2]);
    // a ray along the y axis has no `right` vector (wgsl:99: normalize of a zero cross product): leave those to the probes
    const bool degenerate = cone && d.x == 0.0f && d.z == 0.0f;
    // (a second stage -- the segment itself against the box by slab distances -- was measured: it rejects little more on the
    // reference's benchmark scene and costs 8 % on its straight rows; removed)
    return miss && !degenerate;
}

// Can the probe at `p` read an important texel?  (p inside FrameParams::imp_lo/hi; NaN: yes)
__device__ __forceinline__ bool probe_may_hit(const FrameParams& fp, V3 p)
{
    return !((p.x < fp.imp_lo[0]) | (p.x > fp.imp_hi[0]) | (p.y < fp.imp_lo[1]) | (p.y > fp.imp_hi[1]) | (p.z < fp.imp_lo[2]) | (p.z > fp.imp_hi[2]));
}

// The straight look-ahead of up to K samples per lane, spread over the wave (uninstrumented launches): the samples that
// need one are numbered across the wave (sample 0 of all lanes first, then sample 1, ...), their owners post
// (lane, sample) in a 256-byte LDS mailbox, and lane l of round r walks the chain of candidate 64 r + l -- every lane
// has a chain to walk as long as candidates remain, whoever owns them.  The sample's ray travels by cross-lane reads, the
// answer by ballot.  Same positions and f32 operations per chain as ahead_straight; a chain's early exit only ever saved
// fetches.  Every lane of the wave must call this.
template <int K, class G>
__device__ __forceinline__ void ahead_straight_wave(const G& g, const FrameParams& fp, const bool (&need_in)[K], const float (&ts)[K], V3 o, V3 dir,
                                                    float t_exit, uint32_t lane, uint8_t* mail, bool (&found)[K], uint32_t* rounds_out = nullptr)
{
    static_assert(K <= 4, "two bits for the sample index");
    bool need[K];
    {
        bool any = false;
#pragma unroll
        for (int k = 0; k < K; ++k) { need[k] = need_in[k]; any = any || need[k]; found[k] = false; }
        if (__ballot(any) == 0ull) return;
        // samples whose probes cannot reach an important voxel are answered here (false)
#pragma unroll
        for (int k = 0; k < K; ++k) need[k] = need[k] && !ahead_cannot_hit(fp, o + dir * ts[k], dir, t_exit, false);
    }
    uint32_t my_idx[K];
    uint32_t total = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const unsigned long long m = __ballot(need[k]);
        my_idx[k] = total + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u));
        total += static_cast<uint32_t>(__popcll(m));
        found[k] = false;
    }
    if (total == 0u) return;
    if (rounds_out) *rounds_out += (total + 63u) >> 6;          // rounds of 64 chains walked (a tile's counted cost)
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < K; ++k)
        if (need[k]) mail[my_idx[k]] = static_cast<uint8_t>(lane | (static_cast<uint32_t>(k) << 6));
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int n = static_cast<int>(fp.ahead_steps);
    for (uint32_t r = 0; r * 64u < total; ++r) {
        const uint32_t j = r * 64u + lane;
        const bool job = j < total;
        const uint32_t code = mail[job ? j : 0u];
        const int owner = static_cast<int>(code & 63u);
        const uint32_t kk = code >> 6;
        const V3 o0 = v3(__shfl(o.x, owner, 64), __shfl(o.y, owner, 64), __shfl(o.z, owner, 64));
        const V3 d0 = v3(__shfl(dir.x, owner, 64), __shfl(dir.y, owner, 64), __shfl(dir.z, owner, 64));
        const float t_exit0 = __shfl(t_exit, owner, 64);
        float t0 = __shfl(ts[0], owner, 64);
#pragma unroll
        for (int k = 1; k < K; ++k) {
            const float tk = __shfl(ts[k], owner, 64);
            t0 = kk == static_cast<uint32_t>(k) ? tk : t0;
        }
        const V3 p0 = o0 + d0 * t0;                                                // the sample position, as its owner computes it (wgsl:251)
        const float step = (t_exit0 - length_exact(p0)) / static_cast<float>(n);  // wgsl:145-147
        const V3 ds = d0 * step;
        V3 pos = p0;
        bool live = job, hit = false;
        for (int i = 0; i < n; i += VOLYM_PROBE_BATCH) {
            // the positions are the reference's (accumulated additions); a batch in which no live chain is where an important texel
            // can be read (outside the box of those texels: the byte is < 128 whatever it is) is not fetched
            V3 pb[VOLYM_PROBE_BATCH];
            bool inside = false;
#pragma unroll
            for (int b = 0; b < VOLYM_PROBE_BATCH; ++b) {
                pos = pos + ds;
                pb[b] = pos;
                inside = inside || probe_may_hit(fp, pos);
            }
            if (__ballot(live && inside) == 0ull) continue;
            uint32_t ib[VOLYM_PROBE_BATCH];
#pragma unroll
            for (int b = 0; b < VOLYM_PROBE_BATCH; ++b) ib[b] = g.imp[nearest_offset(g, pb[b])];   // clamped offset: safe wherever pos is
#pragma unroll
            for (int b = 0; b < VOLYM_PROBE_BATCH; ++b)
                if (live && i + b < n && ib[b] >= 128u) { hit = true; live = false; }   // i/255 >= 0.5  <=>  i >= 128
            if (__ballot(live) == 0ull) break;
        }
        const unsigned long long hits = __ballot(hit);
#pragma unroll
        for (int k = 0; k < K; ++k)
            if (need[k] && (my_idx[k] >> 6) == r) found[k] = ((hits >> (my_idx[k] & 63u)) & 1ull) != 0ull;
    }
}

// The cone look-ahead of one sample per lane, spread over the wave (uninstrumented launches): lane 8c + d of a round
// walks cone direction d of the c-th sample that needs a look-ahead, so the 8 chains of a sample run side by side and the
// lanes whose sample needs none work for the others.  Same positions and f32 operations per direction as ahead_cone; a
// direction's early exit only ever saved fetches.  `cone_xo`/`cone_yo`: fp.cone_cos/sin[lane & 7] * 0.2, selected once per
// kernel.  Every lane of the wave must call this (ballots and cross-lane reads inside).
template <class G>
__device__ __forceinline__ bool ahead_cone_wave(const G& g, const FrameParams& fp, bool need_in, V3 start, V3 dir, float t_exit, uint32_t lane,
                                                float cone_xo, float cone_yo)
{
    if (__ballot(need_in) == 0ull) return false;
    const bool need = need_in && !ahead_cannot_hit(fp, start, dir, t_exit, true);   // cannot reach an important voxel: false, unwalked
    const unsigned long long mask = __ballot(need);
    if (mask == 0ull) return false;
    const int n = static_cast<int>(fp.ahead_steps);
    const uint32_t n_cand = static_cast<uint32_t>(__popcll(mask));
    const uint32_t my_rank = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mask >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mask), 0u));
    const float my_step = (t_exit - length_exact(start)) / static_cast<float>(n);
    bool found = false;
    unsigned long long rest = mask;
    for (uint32_t r = 0; r * 8u < n_cand; ++r) {
        // the lanes of the next (up to) 8 samples, 6 bits each (wave-uniform)
        unsigned long long owners = 0ull;
#pragma unroll
        for (uint32_t i = 0; i < 8u; ++i) {
            if (rest != 0ull) {
                owners |= static_cast<unsigned long long>(__builtin_ctzll(rest)) << (6u * i);
                rest &= rest - 1ull;
            }
        }
        const uint32_t c = lane >> 3;
        const bool job = r * 8u + c < n_cand;
        const int owner = static_cast<int>((owners >> (6u * c)) & 63ull);
        const V3 p0 = v3(__shfl(start.x, owner, 64), __shfl(start.y, owner, 64), __shfl(start.z, owner, 64));
        const V3 d0 = v3(__shfl(dir.x, owner, 64), __shfl(dir.y, owner, 64), __shfl(dir.z, owner, 64));
        const float step = __shfl(my_step, owner, 64);
        const V3 right = normalize_exact(cross(d0, v3(0.0f, 1.0f, 0.0f)));       // wgsl:99-113, as ahead_cone
        const V3 new_up = cross(d0, right);
        const V3 sd = normalize_exact((d0 + right * cone_xo) + new_up * cone_yo);
        V3 pos = p0;
        bool left = !job, hit = false;                                             // left: this direction has left [0,1]^3 (wgsl:122-124)
        for (int i = 0; i < n; i += VOLYM_PROBE_BATCH) {
            uint32_t ib[VOLYM_PROBE_BATCH];
            bool out[VOLYM_PROBE_BATCH];
#pragma unroll
            for (int j = 0; j < VOLYM_PROBE_BATCH; ++j) {
                pos = pos + sd * step;
                out[j] = outside01(pos);
                ib[j] = g.imp[nearest_offset(g, pos)];                            // clamped offset: safe wherever pos is
            }
            // (skipping the batches no direction can read an important texel in, as the straight walk does, was measured here: the
            // eight directions of a sample seldom agree, and the tests cost 4-14 % on the cone rows)
#pragma unroll
            for (int j = 0; j < VOLYM_PROBE_BATCH; ++j) {
                if (!left && !hit && i + j < n) {
                    if (out[j]) left = true;
                    else if (ib[j] >= 128u) hit = true;                           // i/255 >= 0.5  <=>  i >= 128
                }
            }
            // a sample is decided as soon as ONE of its eight directions has met an important voxel (wgsl:108-139 returns
            // there): its other seven lanes stop walking
            const unsigned long long hit_now = __ballot(hit);
            if (((hit_now >> (lane & 56u)) & 0xffull) != 0ull) left = true;
            if (__ballot(!left && !hit) == 0ull) break;
        }
        const unsigned long long hits = __ballot(hit);
        if (need && (my_rank >> 3) == r) found = ((hits >> (8u * (my_rank & 7u))) & 0xffull) != 0ull;
    }
    return found;
}

// wgsl:94-139  (EXACT)
template <bool COUNT, class G>
__device__ __forceinline__ bool ahead_cone(const G& g, const FrameParams& fp, V3 cur, V3 dir, float t_exit,
                                           uint32_t& n_imp)
{
    const int n = static_cast<int>(fp.ahead_steps);
    const float step = (t_exit - length_exact(cur)) / static_cast<float>(n);
    const V3 right = normalize_exact(cross(dir, v3(0.0f, 1.0f, 0.0f)));
    const V3 new_up = cross(dir, right);
    for (int c = 0; c < 8; ++c) {
        const float xo = fp.cone_cos[c] * 0.2f;
        const float yo = fp.cone_sin[c] * 0.2f;
        const V3 sd = normalize_exact((dir + right * xo) + new_up * yo);
        V3 pos = cur;
        bool left = false;                       // this direction has left [0,1]^3 (wgsl:122-124 break)
        for (int i = 0; i < n && !left; i += VOLYM_PROBE_BATCH) {
            uint32_t ib[VOLYM_PROBE_BATCH];
            bool out[VOLYM_PROBE_BATCH];
#pragma unroll
            for (int j = 0; j < VOLYM_PROBE_BATCH; ++j) {
                pos = pos + sd * step;
                out[j] = outside01(pos);
                ib[j] = (i + j < n && !out[j]) ? g.imp[nearest_offset(g, pos)] : 0u;
            }
#pragma unroll
            for (int j = 0; j < VOLYM_PROBE_BATCH; ++j) {
                if (!left && i + j < n) {
                    if (out[j]) {
                        left = true;
                    } else {
                        if (COUNT) n_imp++;
                        if (ib[j] >= 128u) return true;
                    }
                }
            }
        }
    }
    return false;
}

// wgsl:190-211 given the six gradient taps  (COLOUR)
__device__ __forceinline__ V3 blinn_phong(V3 color, V3 grad, V3 pos, V3 eye)
{
    const float g2 = dot_fast(grad, grad);
    if (!(g2 > 0.0f)) return color;   // zero gradient: normalize gives NaN, length(NaN) > 0 is false
    const float ginv = __builtin_amdgcn_rsqf(g2);
    const V3 n = grad * ginv;
    const float il = 0.57735026919f;   // normalize(1,1,1)
    const V3 e = eye - pos;
    const V3 E = e * __builtin_amdgcn_rsqf(dot_fast(e, e));
    const V3 h = v3(E.x + il, E.y + il, E.z + il);
    const V3 Hh = h * __builtin_amdgcn_rsqf(dot_fast(h, h));
    const float diffuse = __builtin_fmaxf(0.0f, (n.x + n.y + n.z) * il);
    float s = __builtin_fmaxf(0.0f, dot_fast(Hh, n));
    const float s2 = s * s, s4 = s2 * s2, s8 = s4 * s4, s16 = s8 * s8;
    const float spec = s16 * s8;       // ^24
    const float kd = __builtin_fmaf(0.7f, diffuse, 0.2f);
    const float ks = 0.4f * spec;
    return v3(__builtin_fmaf(color.x, kd, ks), __builtin_fmaf(color.y, kd, ks), __builtin_fmaf(color.z, kd, ks));
}

// t += base, repeated until t >= t_stop, without the repetitions: the exact f32 recurrence of the empty-space steps once
// the step size has reached `base` (wgsl:263-274 with cur == base).  Inside one binade t = m * u (u = ulp, m a 24-bit
// integer) and fl(t + base) = (m + K) * u with the same K = round(base / u) at every step, unless base / u lies exactly
// halfway between two integers (ties go to the even mantissa, which alternates).  So n steps inside a binade are one
// integer multiply-add; the step that crosses into the next binade, ties, and tiny t are single real additions.
__device__ __forceinline__ void replay_saturated(float& t, float t_stop, float base)
{
    while (t < t_stop) {
        const uint32_t tb = __float_as_uint(t);
        const uint32_t ex = tb >> 23;                                    // biased exponent (t > 0)
        const float inv_u = __uint_as_float((277u - ex) << 23);          // 2^(150 - ex) = 1 / ulp(t)
        const float q = base * inv_u;                                    // exact: a power-of-two scaling
        const float kf = __builtin_rintf(q);
        if (ex < 100u || !(q < 4194304.0f) || __builtin_fabsf(q - kf) == 0.5f || !(kf >= 1.0f)) {   // tiny t, huge or tiny step, tie
            t += base;
            continue;
        }
        const uint32_t m = (tb & 0x7fffffu) | 0x800000u;
        const uint32_t K = static_cast<uint32_t>(kf);
        const float xs = t_stop * inv_u;                                 // exact scaling; >= 2^24: not in this binade
        const uint32_t lim = xs < 16777216.0f ? static_cast<uint32_t>(__builtin_ceilf(xs)) : 0x1000000u;   // first mantissa >= t_stop
        const uint32_t need = lim - m;                                   // >= 1 because t < t_stop
        uint32_t n = static_cast<uint32_t>(static_cast<float>(need) * __builtin_amdgcn_rcpf(kf));
        if (n * K < need) n++;                                           // n = ceil(need / K): quotients are small, one fix-up suffices
        if (n * K < need) n++;
        if (m + n * K >= 0x1000000u) {
            // the n-th step leaves the binade: n - 1 steps in closed form, then one real addition
            const uint32_t m2 = m + (n - 1u) * K;
            t = __uint_as_float((tb & 0xff800000u) | (m2 & 0x7fffffu));
            t += base;
        } else {
            const uint32_t m2 = m + n * K;
            t = __uint_as_float((tb & 0xff800000u) | (m2 & 0x7fffffu)); // >= t_stop
        }
    }
}

// The same shading with the half vector given.  Every sample of a ray sees the eye in the same direction: pos = eye + d*t, so
// E = normalize(eye - pos) = -d and Hh = normalize(E + L) is a constant of the RAY (wgsl:199-205 recompute it per sample;
// the values differ by the rounding of pos, ~1e-7).  COLOUR arithmetic.
__device__ __forceinline__ V3 ray_half_vector(V3 d)
{
    const float il = 0.57735026919f;   // normalize(1,1,1)
    const V3 h = v3(il - d.x, il - d.y, il - d.z);
    return h * __builtin_amdgcn_rsqf(dot_fast(h, h));
}
__device__ __forceinline__ V3 blinn_phong_h(V3 color, V3 grad, V3 Hh)
{
    const float g2 = dot_fast(grad, grad);
    if (!(g2 > 0.0f)) return color;   // zero gradient: normalize gives NaN, length(NaN) > 0 is false
    const float ginv = __builtin_amdgcn_rsqf(g2);
    const V3 n = grad * ginv;
    const float il = 0.57735026919f;
    const float diffuse = __builtin_fmaxf(0.0f, (n.x + n.y + n.z) * il);
    float s = __builtin_fmaxf(0.0f, dot_fast(Hh, n));
    const float s2 = s * s, s4 = s2 * s2, s8 = s4 * s4, s16 = s8 * s8;
    const float spec = s16 * s8;       // ^24
    const float kd = __builtin_fmaf(0.7f, diffuse, 0.2f);
    const float ks = 0.4f * spec;
    return v3(__builtin_fmaf(color.x, kd, ks), __builtin_fmaf(color.y, kd, ks), __builtin_fmaf(color.z, kd, ks));
}

// rgba8unorm store: clamp, scale, round to nearest
__device__ __forceinline__ uint32_t to_unorm8(float v)
{
    if (!(v > 0.0f)) return 0u;
    if (v >= 1.0f) return 255u;
    return static_cast<uint32_t>(__builtin_floorf(v * 255.0f + 0.5f));
}

__device__ __forceinline__ uint32_t pack_rgba8(float r, float g, float b, float a)
{
    return to_unorm8(r) | (to_unorm8(g) << 8) | (to_unorm8(b) << 16) | (to_unorm8(a) << 24);
}

// ---- ray set-up (wgsl:221-241, EXACT) ----
// The 14 divisions of the set-up share 5 denominators (W, H, wp.w, |world - eye|, and each direction component for its slab pair).
// hipcc expands a binary32 division into: v_div_scale (denominator), v_div_scale (numerator), v_rcp, two fma that refine the
// reciprocal, a multiply and three fma that form and correct the quotient twice (the last one is v_div_fmas), v_div_fixup.  When
// both operands are normal numbers of moderate size -- here: magnitudes in (2^-40, 2^40), the exact conditions are those of
// v_div_scale_f32 in the ISA manual: |exponent difference| < 96, neither 1/den nor num/den denormal, biased exponent(num) > 23 --
// the two v_div_scale return their operand unchanged and clear VCC, v_div_fmas is a plain fma and v_div_fixup returns the
// quotient it is given.  What is left are the eight instructions below, and the first three depend on the denominator only: the
// quotients rcp_refined / div_by produce are, instruction for instruction, those of the `/` operator.  make_ray computes the set-up
// with them speculatively, checks the ranges on the values themselves, and a wave in which any lane is outside them recomputes
// the set-up with plain divisions (wave-uniform branch; in practice: rays exactly parallel to a cube face).
// tests/test_setup_division.py: the sequence against IEEE division on the CPU (exhaustive for the pixel quotients);
// volym_selftest_ray_setup + tests/test_gpu_parity.py: both forms on the device, every bit of every ray of a frame.
__device__ __forceinline__ float rcp_refined(float d)
{
    const float r = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}
__device__ __forceinline__ float div_by(float n, float d, float r)
{
    float q = n * r;
    float e = __builtin_fmaf(-d, q, n);
    q = __builtin_fmaf(e, r, q);
    e = __builtin_fmaf(-d, q, n);
    return __builtin_fmaf(e, r, q);
}
// gx / W for integers 0 <= gx < W <= 16384 with r = RN(1 / W) from the host: one correction step gives the correctly rounded
// quotient for every such pair (exhaustive: tests/test_setup_division.py, 134 M pairs)
__device__ __forceinline__ float div_pixel(float g, float w, float r)
{
    const float q = g * r;
    return __builtin_fmaf(__builtin_fmaf(-w, q, g), r, q);
}

struct Ray {
    V3 o, d;
    float t_entry, t_exit;
    bool hit;
};

__device__ __forceinline__ void ray_slabs(Ray& r, float t1x, float t2x, float t1y, float t2y, float t1z, float t2z)
{
    const float entry = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(t1x, t2x), __builtin_fminf(t1y, t2y)),
                                        __builtin_fminf(t1z, t2z));
    const float exit_ = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(t1x, t2x), __builtin_fmaxf(t1y, t2y)),
                                        __builtin_fmaxf(t1z, t2z));
    r.t_entry = __builtin_fmaxf(entry, 0.0f);
    r.t_exit = __builtin_fmaxf(exit_, 0.0f);
    r.hit = !(r.t_exit <= r.t_entry);
}

// the shader's arithmetic, operation for operation
__device__ __forceinline__ Ray make_ray_ieee(const FrameParams& fp, uint32_t gx, uint32_t gy)
{
    Ray r;
    const float scx = static_cast<float>(gx) / static_cast<float>(fp.W);
    const float scy = static_cast<float>(gy) / static_cast<float>(fp.H);
    const float ndx = scx * 2.0f - 1.0f;
    const float ndy = 1.0f - scy * 2.0f;
    float wp[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
        wp[k] = ((fp.ivp[k] * ndx + fp.ivp[4 + k] * ndy) + fp.ivp[8 + k] * 0.0f) + fp.ivp[12 + k] * 1.0f;
    r.o = v3(fp.eye[0], fp.eye[1], fp.eye[2]);
    const V3 world = v3(wp[0] / wp[3], wp[1] / wp[3], wp[2] / wp[3]);
    r.d = normalize_exact(world - r.o);
    ray_slabs(r, (0.0f - r.o.x) / r.d.x, (1.0f - r.o.x) / r.d.x, (0.0f - r.o.y) / r.d.y, (1.0f - r.o.y) / r.d.y,
              (0.0f - r.o.z) / r.d.z, (1.0f - r.o.z) / r.d.z);
    return r;
}

// the same values through shared reciprocals; ok: every operand was inside the range in which that is the same arithmetic
__device__ __forceinline__ Ray make_ray_shared(const FrameParams& fp, uint32_t gx, uint32_t gy, bool& ok)
{
    Ray r;
    const float scx = div_pixel(static_cast<float>(gx), static_cast<float>(fp.W), fp.rcp_w);
    const float scy = div_pixel(static_cast<float>(gy), static_cast<float>(fp.H), fp.rcp_h);
    const float ndx = scx * 2.0f - 1.0f;
    const float ndy = 1.0f - scy * 2.0f;
    float wp[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
        wp[k] = ((fp.ivp[k] * ndx + fp.ivp[4 + k] * ndy) + fp.ivp[8 + k] * 0.0f) + fp.ivp[12 + k] * 1.0f;
    r.o = v3(fp.eye[0], fp.eye[1], fp.eye[2]);
    const float rw = rcp_refined(wp[3]);
    const V3 a = v3(div_by(wp[0], wp[3], rw), div_by(wp[1], wp[3], rw), div_by(wp[2], wp[3], rw)) - r.o;
    const float len = length_exact(a);
    const float rl = rcp_refined(len);
    r.d = v3(div_by(a.x, len, rl), div_by(a.y, len, rl), div_by(a.z, len, rl));
    const float rx = rcp_refined(r.d.x), ry = rcp_refined(r.d.y), rz = rcp_refined(r.d.z);
    ray_slabs(r, div_by(0.0f - r.o.x, r.d.x, rx), div_by(1.0f - r.o.x, r.d.x, rx), div_by(0.0f - r.o.y, r.d.y, ry),
              div_by(1.0f - r.o.y, r.d.y, ry), div_by(0.0f - r.o.z, r.d.z, rz), div_by(1.0f - r.o.z, r.d.z, rz));
    // Ranges.  wp (4 values) and len inside (setup_lo, 2^40): setup_lo is 2^-40, or +inf when the host could not vouch for its part
    // (W, H <= 16384; |ivp| < 2^60, so that wp is never NaN: fmin / fmax would drop one; the six slab numerators inside
    // [2^-40, 2^40]).  Direction components above 2^-30 in magnitude: with len in range that bounds the numerators of the
    // normalisation from below (a zero or tiny component gives a quotient below 2^-60 whatever the sequence does with it), their
    // upper bound is len itself; and it is the range of the slab denominators.
    const float lo = __builtin_fminf(__builtin_fminf(__builtin_fminf(__builtin_fabsf(wp[0]), __builtin_fabsf(wp[1])), __builtin_fabsf(wp[2])),
                                     __builtin_fminf(__builtin_fabsf(wp[3]), len));
    const float hi = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(wp[0]), __builtin_fabsf(wp[1])), __builtin_fabsf(wp[2])),
                                     __builtin_fmaxf(__builtin_fabsf(wp[3]), len));
    const float dlo = __builtin_fminf(__builtin_fminf(__builtin_fabsf(r.d.x), __builtin_fabsf(r.d.y)), __builtin_fabsf(r.d.z));
    ok = lo > fp.setup_lo && hi < 0x1p+40f && dlo > 0x1p-30f;
    return r;
}

// SHARED = false: the plain divisions only -- the instantiations that already keep state in scratch (importance rendering, continuous
// rho) lose more to the fast form's extra live values than they gain (same-box A/B: importance look-ahead 60.3 -> 63.3 us, smoothing
// and cone +0.7 %, against -0.5 % on the common instantiation): they keep the shader's form
template <bool SHARED = true>
__device__ __forceinline__ Ray make_ray(const FrameParams& fp, uint32_t gx, uint32_t gy)
{
    if constexpr (!SHARED) return make_ray_ieee(fp, gx, gy);
    bool ok;
    const Ray r = make_ray_shared(fp, gx, gy, ok);
    if (__builtin_expect(__ballot(!ok) == 0ull, 1)) return r;
    return make_ray_ieee(fp, gx, gy);
}

}  // namespace volym
