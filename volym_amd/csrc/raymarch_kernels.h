// The ray-march kernels (gfx950).  One wave64 = one 8x8 pixel tile, one 256-thread
// workgroup = one 16x16 screen tile (the reference's workgroup shape, wgsl:213).
//
//   VARIANT 0  "direct": every fetch the reference shader makes is issued (BASELINE config 2).
//   VARIANT 1  "macro-cell": a (mc_n)^3 grid of per-cell density maxima lets a ray replay the
//              step state machine through provably-empty cells without touching the volume
//              (BASELINE config 3); the float arithmetic on t / cur_step is replayed exactly,
//              so pixels are identical to VARIANT 0.
//   COUNT      instrumented launch that counts the reference's fetches (volym_stats_pass).
#pragma once

#include "raymarch_device.h"

namespace volym {

// block -> tile.  Workgroups are dealt round-robin over the 8 XCDs (block b -> XCD b % 8), each
// with its own 4 MiB L2; giving each XCD contiguous screen bands keeps its slice of the volume
// L2-resident.  bands == 0: identity.
__device__ __forceinline__ uint32_t block_to_local_tile(uint32_t b, uint32_t n_local, uint32_t bands)
{
    if (bands == 0u) return b;
    // n_local tiles are split into 8*bands chunks; XCD x owns chunks x, x+8, x+16, ...
    const uint32_t xcd = b & 7u;
    const uint32_t i = b >> 3;                      // i-th block of this XCD
    const uint32_t chunks = 8u * bands;
    const uint32_t per_chunk = (n_local + chunks - 1u) / chunks;
    const uint32_t band = i / per_chunk;            // which of this XCD's chunks
    const uint32_t within = i - band * per_chunk;
    return (band * 8u + xcd) * per_chunk + within;  // may be >= n_local: caller guards
}

// One dense sample: classification (wgsl:276-304), gradient + Blinn-Phong (wgsl:181-211) and
// compositing (wgsl:313-325).  Returns 0 when the march goes on (t already advanced), 1 on the
// first-hit break (wgsl:319-323).
struct MarchCtx {
    const float4* s_tf_tab;
    const float4* s_lut;
    const float* s_ic_alpha;
    const float* s_rho;
    uint32_t flags;
    bool linear, table_mode;
};

template <bool COUNT, class G>
__device__ __forceinline__ int dense_sample(const MarchCtx& m, const G& g, const FrameParams& fp, const Ray& ray, V3 pos,
                                            int ix, int iy, int iz, uint32_t off, uint32_t b, float rho, float& t, float cur,
                                            V3& acc, float& acc_a, uint32_t& n_vol, uint32_t& n_imp)
{
    float cr, cg, cb, alpha_step;
    bool use_alpha = (m.flags & F_OPACITY) != 0u;
    if (m.flags & F_IMP_COLORING) {                              // wgsl:83-92
        const uint32_t ib = g.imp[off];
        const float im = m.s_rho[ib];
        cr = __builtin_fminf(im * 1.5f, 1.0f);
        cg = (1.0f - im) * 1.2f;
        cb = 0.2f;
        alpha_step = m.s_ic_alpha[ib];
        use_alpha = true;
    } else {
        if (m.flags & F_IMP_RENDERING) {                         // wgsl:283-295
            const uint32_t ib = g.imp[off];
            const bool ahead = (m.flags & F_CONE) ? ahead_cone<COUNT>(g, fp, pos, ray.d, ray.t_exit, n_imp)
                                                  : ahead_straight<COUNT>(g, fp, pos, ray.d, ray.t_exit, n_imp);
            if (ib < 255u && ahead) { t += cur; return 0; }      // importance < 1.0 && ahead
        }
        if (m.table_mode) {
            const float4 ca = m.s_tf_tab[b];
            cr = ca.x; cg = ca.y; cb = ca.z; alpha_step = ca.w;
        } else {
            const float4 ca = sample_tf(m.s_lut, fp.tf_n, rho);  // wgsl:297-303
            cr = ca.x; cg = ca.y; cb = ca.z;
            alpha_step = 1.0f - wgsl_pow(1.0f - ca.w, fp.alpha_y);   // wgsl:314
        }
    }

    // ---- gradient taps (wgsl:181-188) ----
    V3 grad;
    const float o = 0.01f;
    if (m.table_mode) {
        const int ixp = texel_nearest(pos.x + o, g.fnx, g.hix), ixm = texel_nearest(pos.x - o, g.fnx, g.hix);
        const int iyp = texel_nearest(pos.y + o, g.fny, g.hiy), iym = texel_nearest(pos.y - o, g.fny, g.hiy);
        const int izp = texel_nearest(pos.z + o, g.fnz, g.hiz), izm = texel_nearest(pos.z - o, g.fnz, g.hiz);
        const uint32_t bxp = g.vol[voxel_offset(g, ixp, iy, iz)], bxm = g.vol[voxel_offset(g, ixm, iy, iz)];
        const uint32_t byp = g.vol[voxel_offset(g, ix, iyp, iz)], bym = g.vol[voxel_offset(g, ix, iym, iz)];
        const uint32_t bzp = g.vol[voxel_offset(g, ix, iy, izp)], bzm = g.vol[voxel_offset(g, ix, iy, izm)];
        grad = v3(m.s_rho[bxp] - m.s_rho[bxm], m.s_rho[byp] - m.s_rho[bym], m.s_rho[bzp] - m.s_rho[bzm]);
    } else {
        grad = v3(sample_density(g, m.s_rho, m.linear, v3(pos.x + o, pos.y, pos.z)) -
                      sample_density(g, m.s_rho, m.linear, v3(pos.x - o, pos.y, pos.z)),
                  sample_density(g, m.s_rho, m.linear, v3(pos.x, pos.y + o, pos.z)) -
                      sample_density(g, m.s_rho, m.linear, v3(pos.x, pos.y - o, pos.z)),
                  sample_density(g, m.s_rho, m.linear, v3(pos.x, pos.y, pos.z + o)) -
                      sample_density(g, m.s_rho, m.linear, v3(pos.x, pos.y, pos.z - o)));
    }
    if (COUNT) n_vol += 6;
    // the common 1/(2*0.01) factor cancels in normalize()
    const V3 shaded = blinn_phong(v3(cr, cg, cb), grad, pos, ray.o);   // wgsl:306-311

    if (use_alpha) {                                             // wgsl:313-318
        const float w = (1.0f - acc_a) * alpha_step;
        acc = v3(__builtin_fmaf(shaded.x, w, acc.x), __builtin_fmaf(shaded.y, w, acc.y), __builtin_fmaf(shaded.z, w, acc.z));
        acc_a += w;
        t += cur;                                                // wgsl:325
        return 0;
    }
    acc = shaded;                                                // wgsl:319-323
    acc_a = 1.0f;
    return 1;
}


template <int VARIANT, bool COUNT, bool TRACE, bool BRICK = false>
__global__ __launch_bounds__(256) void volym_raymarch_kernel(
    const uint8_t* __restrict__ vol, const uint8_t* __restrict__ imp, const FrameTables* __restrict__ tables,
    const uint8_t* __restrict__ df4, uint32_t* __restrict__ out_shard, uint32_t* __restrict__ out_raster,
    float4* __restrict__ out_f32, Counters* __restrict__ counters, uint4* __restrict__ trace, const FrameParams fp)
{
    // development aid (TRACE launches only): per-wave start/end on the 100 MHz realtime counter
    unsigned long long trace_t0 = 0;
    uint32_t trace_iters = 0, trace_dense = 0;
    if (TRACE) trace_t0 = __builtin_amdgcn_s_memrealtime();
    __shared__ float4 s_tf_tab[256];
    __shared__ float4 s_lut[256];
    __shared__ float s_ic_alpha[256];
    __shared__ float s_rho[256];
    __shared__ __attribute__((aligned(16))) uint8_t s_df[VARIANT == 1 ? VOLYM_DF_LDS_BYTES : 16];

    const uint32_t flags = fp.flags;
    MarchCtx m;
    m.s_tf_tab = s_tf_tab; m.s_lut = s_lut; m.s_ic_alpha = s_ic_alpha; m.s_rho = s_rho;
    m.flags = flags;
    m.linear = (flags & F_LINEAR) != 0u;
    const bool gauss = (flags & F_GAUSSIAN) != 0u;
    m.table_mode = !m.linear && !gauss;      // rho is one of 256 values
    const bool use_df = VARIANT == 1 && m.table_mode;

    {
        const uint32_t i = threadIdx.x;
        s_tf_tab[i] = tables->tf_tab[i];
        s_rho[i] = tables->rho[i];
        if (!m.table_mode) s_lut[i] = tables->lut_f[i];
        if (flags & F_IMP_COLORING) s_ic_alpha[i] = tables->ic_alpha[i];
        if (use_df) {
            const uint32_t n16 = VOLYM_DF_IN_LDS(fp) ? (fp.mc_n * fp.mc_n * fp.mc_n / 2u + 15u) / 16u : 0u;   // 16-byte pieces (a finer grid stays in global memory)
            const uint4* src = reinterpret_cast<const uint4*>(df4);
            uint4* dst = reinterpret_cast<uint4*>(s_df);
            for (uint32_t k = i; k < n16; k += 256u) dst[k] = src[k];
        }
    }
    __syncthreads();

    const uint32_t local_tile = block_to_local_tile(blockIdx.x, fp.n_local, fp.xcd_bands);
    if (local_tile >= fp.n_local) return;
    const uint32_t tile = local_tile * fp.world + fp.rank;
    const uint32_t tx = tile % fp.tiles_x, ty = tile / fp.tiles_x;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t px = ((wave & 1u) << 3) | (lane & 7u);
    const uint32_t py = ((wave >> 1) << 3) | (lane >> 3);
    const uint32_t gx = tx * 16u + px, gy = ty * 16u + py;
    const bool in_frame = gx < fp.W && gy < fp.H;   // wgsl:217-219

    GridT<BRICK> g;
    grid_init(g, vol, imp, fp.nx, fp.ny, fp.nz);

    uint32_t n_vol = 0, n_imp = 0, n_steps = 0, n_dense = 0, n_hit = 0;
    float out_r = 0.0f, out_g = 0.0f, out_b = 0.0f, out_a = 1.0f;   // miss: (0,0,0,1) wgsl:239

    if (in_frame) {
        const Ray ray = make_ray(fp, gx, gy);
        if (ray.hit) {
            if (COUNT) n_hit = 1;
            const float base = fp.base_step, min_step = fp.min_step, thr = fp.thr;
            float t = ray.t_entry, cur = base;
            V3 acc = v3(0.0f, 0.0f, 0.0f);
            float acc_a = 0.0f;

            // distance-field leaps: conservative arithmetic only (never decides a pixel by itself)
            const float mcf = static_cast<float>(fp.mc_n), inv_mc = 1.0f / mcf;
            const float idx_ = 1.0f / ray.d.x, idy_ = 1.0f / ray.d.y, idz_ = 1.0f / ray.d.z;
            const float nox = -ray.o.x * idx_, noy = -ray.o.y * idy_, noz = -ray.o.z * idz_;

            while (t < ray.t_exit && acc_a < 0.95f) {             // wgsl:250
                if (TRACE) trace_iters++;
                const V3 pos = ray.o + ray.d * t;                 // wgsl:251
                if (use_df) {
                    // Every sample whose position lies in a macro cell with distance value D >= 1 sees
                    // rho < threshold: the (2D-1)^3 cells around it hold no voxel >= thr_byte.  Such a
                    // step is (wgsl:263-274) cur = min(base, cur*1.5); t += cur -- replayed here in the
                    // same f32 arithmetic, without the fetches.
                    const float cxf = __builtin_floorf(pos.x * mcf), cyf = __builtin_floorf(pos.y * mcf),
                                czf = __builtin_floorf(pos.z * mcf);
                    const int cx = static_cast<int>(cxf), cy = static_cast<int>(cyf), cz = static_cast<int>(czf);
                    uint32_t D = 0;
                    if (static_cast<uint32_t>(cx | cy | cz) < fp.mc_n) {
                        const uint32_t ci = static_cast<uint32_t>(cx) + fp.mc_n * (static_cast<uint32_t>(cy) + fp.mc_n * static_cast<uint32_t>(cz));
                        D = (static_cast<uint32_t>(VOLYM_DF_IN_LDS(fp) ? s_df[ci >> 1] : df4[ci >> 1]) >> ((ci & 1u) * 4u)) & 15u;
                    }
                    if (D != 0u) {
                        // box of empty cells [c-R, c+R+1]/mc_n shrunk by eps on every face, R = D-1
                        const float eps = 4.0e-5f;
                        const float a = static_cast<float>(D - 1u) * inv_mc - eps;
                        const float lx = __builtin_fmaf(cxf, inv_mc, -a), hx = __builtin_fmaf(cxf, inv_mc, a + inv_mc);
                        const float ly = __builtin_fmaf(cyf, inv_mc, -a), hy = __builtin_fmaf(cyf, inv_mc, a + inv_mc);
                        const float lz = __builtin_fmaf(czf, inv_mc, -a), hz = __builtin_fmaf(czf, inv_mc, a + inv_mc);
                        const float ex = __builtin_fmaxf(__builtin_fmaf(lx, idx_, nox), __builtin_fmaf(hx, idx_, nox));
                        const float ey = __builtin_fmaxf(__builtin_fmaf(ly, idy_, noy), __builtin_fmaf(hy, idy_, noy));
                        const float ez = __builtin_fmaxf(__builtin_fmaf(lz, idz_, noz), __builtin_fmaf(hz, idz_, noz));
                        float te = __builtin_fminf(__builtin_fminf(ex, ey), ez);   // NaN (0*inf) drops out
                        te = te - 2.0e-5f * __builtin_fabsf(te);                    // rounding slack
                        const bool inside = pos.x > lx && pos.x < hx && pos.y > ly && pos.y < hy && pos.z > lz && pos.z < hz;
                        const float t_stop = __builtin_fminf(te, ray.t_exit);
                        if (inside && t < t_stop) {
                            do {
                                if (COUNT) { n_steps++; n_vol++; n_imp++; }
                                cur = __builtin_fminf(base, cur * 1.5f);
                                t += cur;
                            } while (t < t_stop);
                            continue;
                        }
                    }
                }
                if (COUNT) n_steps++;

                // ---- density (wgsl:253-259) and the step state machine (wgsl:263-274) ----
                int ix = 0, iy = 0, iz = 0;
                uint32_t off = 0, b = 0;
                float rho = 0.0f;
                bool dense;
                if (m.table_mode) {
                    ix = texel_nearest(pos.x, g.fnx, g.hix);
                    iy = texel_nearest(pos.y, g.fny, g.hiy);
                    iz = texel_nearest(pos.z, g.fnz, g.hiz);
                    off = voxel_offset(g, ix, iy, iz);
                    b = vol[off];
                    if (COUNT) n_vol++;
                    dense = b >= fp.thr_byte;                     // <=> b/255 >= thr
                } else {
                    if (gauss) rho = sample_density_smoothed<COUNT>(g, s_rho, m.linear, fp, pos, ray.d, n_vol);
                    else { rho = sample_density(g, s_rho, m.linear, pos); if (COUNT) n_vol++; }
                    dense = rho >= thr;
                    off = nearest_offset(g, pos);                 // importance texel (always nearest)
                }
                if (COUNT) n_imp++;                               // wgsl:260 fetches it every step
                cur = dense ? min_step : __builtin_fminf(base, cur * 1.5f);
                if (!dense) { t += cur; continue; }
                if (COUNT) n_dense++;
                if (TRACE) trace_dense++;
                if (dense_sample<COUNT>(m, g, fp, ray, pos, ix, iy, iz, off, b, rho, t, cur, acc, acc_a, n_vol, n_imp)) break;
            }
            out_r = acc.x; out_g = acc.y; out_b = acc.z; out_a = acc_a;
        }

        // ---- store (wgsl:328-329; rgba8unorm) ----
        const uint32_t packed = pack_rgba8(out_r, out_g, out_b, out_a);
        if (flags & F_RASTER) {
            const size_t o = static_cast<size_t>(gy) * fp.W + gx;
            out_raster[o] = packed;
            if (flags & F_WRITE_F32) out_f32[o] = make_float4(out_r, out_g, out_b, out_a);
        } else {
            out_shard[static_cast<size_t>(local_tile) * 256u + threadIdx.x] = packed;
        }
    } else if (!(flags & F_RASTER)) {
        out_shard[static_cast<size_t>(local_tile) * 256u + threadIdx.x] = 0u;
    }

    if (TRACE) {
        uint32_t it = trace_iters, dn = trace_dense;
        for (int s = 32; s > 0; s >>= 1) { it = max(it, __shfl_xor(it, s, 64)); dn = max(dn, __shfl_xor(dn, s, 64)); }
        const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
        if (lane == 0)
            trace[(static_cast<size_t>(blockIdx.x) << 2) | wave] =
                make_uint4(static_cast<uint32_t>(trace_t0), static_cast<uint32_t>(t1 - trace_t0), it, dn);
    }
    if (COUNT) {
        unsigned long long v[5] = {n_vol, n_imp, n_steps, n_dense, n_hit};
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            unsigned long long x = v[k];
            for (int s = 32; s > 0; s >>= 1) x += __shfl_xor(x, s, 64);
            v[k] = x;
        }
        if (lane == 0) {
            atomicAdd(&counters->n_vol, v[0]);
            atomicAdd(&counters->n_imp, v[1]);
            atomicAdd(&counters->n_steps, v[2]);
            atomicAdd(&counters->n_dense, v[3]);
            atomicAdd(&counters->n_hit, v[4]);
        }
    }
}

// Chebyshev distance (in cells, capped at 15) from every macro cell to the nearest cell that may
// hold a voxel >= thr_byte; 0 for such cells.  Cells outside the grid count as empty.  Output packed
// two cells per byte (low nibble = even cell).  Also emits the AABB of the occupied cells.
//
// One workgroup, bit-parallel: a row (y, z) of the grid is one 64-bit word (bit x), thread r owns the rows r, r + 1024, ...
// (n <= 64: at most four).  mask_0 = occupied, mask_k = mask_{k-1} dilated by one cell along x, y and z (a 3x3x3 box); the
// masks are nested, so D(c) = #{k in 0..14 : c not in mask_k}, accumulated in four bit planes.
constexpr uint32_t VOLYM_DF_MAX_N = 64;
template <int RPT>     // rows per thread: 1 for grids up to 32^3, 4 up to 64^3
__global__ __launch_bounds__(1024) void volym_distance_field_kernel(const uint8_t* __restrict__ mc_max, uint8_t* __restrict__ df4,
                                                                    int* __restrict__ aabb, uint32_t mc_n, uint32_t thr_byte)
{
    typedef unsigned long long u64;
    __shared__ u64 rows[RPT * 1024];
    __shared__ u64 tmp[RPT * 1024];
    __shared__ int s_box[6];
    const uint32_t n = mc_n, n_rows = n * n;
    if (threadIdx.x < 6u) s_box[threadIdx.x] = (threadIdx.x < 3u) ? static_cast<int>(n) : -1;
    u64 m[RPT], p0[RPT], p1[RPT], p2[RPT], p3[RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const uint32_t r = threadIdx.x + 1024u * i;
        m[i] = 0; p0[i] = p1[i] = p2[i] = p3[i] = 0;
        if (r < n_rows) {
            const uint32_t y = r % n, z = r / n;
            for (uint32_t x = 0; x < n; ++x)
                if (mc_max[x + n * (y + n * z)] >= thr_byte) m[i] |= 1ull << x;
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const uint32_t r = threadIdx.x + 1024u * i;
        if (r < n_rows && m[i]) {
            const uint32_t y = r % n, z = r / n;
            atomicMin(&s_box[0], __builtin_ctzll(m[i]));
            atomicMax(&s_box[3], 63 - __builtin_clzll(m[i]));
            atomicMin(&s_box[1], static_cast<int>(y)); atomicMax(&s_box[4], static_cast<int>(y));
            atomicMin(&s_box[2], static_cast<int>(z)); atomicMax(&s_box[5], static_cast<int>(z));
        }
    }
    const u64 row_mask = n >= 64u ? ~0ull : ((1ull << n) - 1ull);
    for (int k = 0; k < 15; ++k) {
        // count += !mask_k  (bit-sliced ripple add of a one-bit addend)
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            u64 carry = ~m[i] & row_mask;
            u64 t0 = p0[i] & carry; p0[i] ^= carry; carry = t0;
            t0 = p1[i] & carry; p1[i] ^= carry; carry = t0;
            t0 = p2[i] & carry; p2[i] ^= carry; carry = t0;
            p3[i] ^= carry;
        }
        if (k == 14) break;
        // dilate: x in-register, then y and z through LDS
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const uint32_t r = threadIdx.x + 1024u * i;
            m[i] = (m[i] | (m[i] << 1) | (m[i] >> 1)) & row_mask;
            if (r < n_rows) rows[r] = m[i];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const uint32_t r = threadIdx.x + 1024u * i;
            if (r < n_rows) {
                const uint32_t y = r % n;
                u64 v = m[i];
                if (y > 0u) v |= rows[r - 1u];
                if (y + 1u < n) v |= rows[r + 1u];
                tmp[r] = v;
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const uint32_t r = threadIdx.x + 1024u * i;
            if (r < n_rows) {
                const uint32_t z = r / n;
                u64 v = tmp[r];
                if (z > 0u) v |= tmp[r - n];
                if (z + 1u < n) v |= tmp[r + n];
                m[i] = v;
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const uint32_t r = threadIdx.x + 1024u * i;
        if (r < n_rows) {
            const uint32_t y = r % n, z = r / n;
            uint8_t* out = df4 + (static_cast<size_t>(n) * (y + n * z)) / 2u;
            for (uint32_t x = 0; x < n; x += 2u) {
                const uint32_t lo = static_cast<uint32_t>((p0[i] >> x) & 1ull) | (static_cast<uint32_t>((p1[i] >> x) & 1ull) << 1) | (static_cast<uint32_t>((p2[i] >> x) & 1ull) << 2) | (static_cast<uint32_t>((p3[i] >> x) & 1ull) << 3);
                const uint32_t x1 = x + 1u;
                const uint32_t hi = static_cast<uint32_t>((p0[i] >> x1) & 1ull) | (static_cast<uint32_t>((p1[i] >> x1) & 1ull) << 1) | (static_cast<uint32_t>((p2[i] >> x1) & 1ull) << 2) | (static_cast<uint32_t>((p3[i] >> x1) & 1ull) << 3);
                out[x / 2u] = static_cast<uint8_t>(lo | (hi << 4));
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < 6u) aabb[threadIdx.x] = s_box[threadIdx.x];
}

// Which 8x8 pixel tiles can see anything dense?  Every macro cell that may hold a voxel >= thr_byte (the criterion of the
// distance field and of the AABB) is projected -- its box grown by `margin` in texture space, the bounding rectangle of its
// eight corners grown by 1.5 pixels, as the hulls are -- and the tiles with a pixel inside get their bit.  A ray meets a dense
// sample only inside such a cell, and the ray of pixel (gx, gy) passes through the point (gx, gy) of the projection
// (wgsl:221-229), so a tile without its bit is constant.  One thread per cell, bits set with atomicOr into a mask that the
// march kernel keeps cleared (it zeroes the buffer it is not reading, see volym_raymarch_pq_kernel); a bit left over from an
// earlier view only costs the march of a tile, never a pixel.  Launched once per view, in stream order (raymarch.hip
// ensure_frame_resources).  The caller guarantees w > 0 for every corner (the AABB of the occupied cells projected with all
// its corners in front of the eye: CULL_OBJ_HULL); a cell with a corner that is not sets every bit.
struct ClipMatrix { float m[16]; };   // world -> clip, column-major
constexpr uint32_t VOLYM_TILE_MASK_MAX_WORDS = 1u << 20;     // 32 M tiles of 8x8 pixels
// volym_selftest_ray_setup: one pixel per thread (64 x 4 pixels per workgroup, a wave = 64 pixels of a row)
__global__ __launch_bounds__(256) void volym_ray_setup_selftest_kernel(FrameParams fp, unsigned long long* __restrict__ out)
{
    const uint32_t gx = blockIdx.x * 64u + (threadIdx.x & 63u), gy = blockIdx.y * 4u + (threadIdx.x >> 6);
    if (gx >= fp.W || gy >= fp.H) return;
    bool ok;
    const Ray s = make_ray_shared(fp, gx, gy, ok);
    const bool fell_back = __ballot(!ok) != 0ull;
    const Ray p = make_ray(fp, gx, gy);               // what the march kernels call
    const Ray q = make_ray_ieee(fp, gx, gy);
    auto same = [](float a, float b) { return __float_as_uint(a) == __float_as_uint(b); };
    const bool eq = same(p.d.x, q.d.x) && same(p.d.y, q.d.y) && same(p.d.z, q.d.z) && same(p.t_entry, q.t_entry) &&
                    same(p.t_exit, q.t_exit) && p.hit == q.hit && same(p.o.x, q.o.x) && same(p.o.y, q.o.y) && same(p.o.z, q.o.z);
    // a lane inside the ranges must agree on its own, whatever the rest of its wave did
    const bool eq_s = !ok || (same(s.d.x, q.d.x) && same(s.d.y, q.d.y) && same(s.d.z, q.d.z) && same(s.t_entry, q.t_entry) &&
                              same(s.t_exit, q.t_exit) && s.hit == q.hit);
    const unsigned long long bad = __ballot(!(eq && eq_s)), all = __ballot(true);
    if ((threadIdx.x & 63u) == static_cast<uint32_t>(__ffsll(static_cast<long long>(all)) - 1)) {
        if (bad) atomicAdd(&out[0], static_cast<unsigned long long>(__popcll(bad)));
        if (fell_back) atomicAdd(&out[1], static_cast<unsigned long long>(__popcll(all)));
        atomicAdd(&out[2], static_cast<unsigned long long>(__popcll(all)));
    }
}

__global__ __launch_bounds__(256) void volym_tile_mask_kernel(const uint8_t* __restrict__ mc_max, uint32_t mc_n, uint32_t thr_byte, ClipMatrix M,
                                                              float margin, uint32_t W, uint32_t H, uint32_t t8x, uint32_t n_words,
                                                              uint32_t* __restrict__ out)
{
    const uint32_t cell = blockIdx.x * 256u + threadIdx.x;
    const uint32_t cells = mc_n * mc_n * mc_n;
    if (cell >= cells || mc_max[cell] < thr_byte) return;
    const float inv = 1.0f / static_cast<float>(mc_n);
    const float fw = static_cast<float>(W), fh = static_cast<float>(H);
    const uint32_t cx = cell % mc_n, cy = (cell / mc_n) % mc_n, cz = cell / (mc_n * mc_n);
    const float lo[3] = {static_cast<float>(cx) * inv - margin, static_cast<float>(cy) * inv - margin, static_cast<float>(cz) * inv - margin};
    const float hi[3] = {static_cast<float>(cx + 1u) * inv + margin, static_cast<float>(cy + 1u) * inv + margin, static_cast<float>(cz + 1u) * inv + margin};
    float px0 = 3.0e38f, px1 = -3.0e38f, py0 = 3.0e38f, py1 = -3.0e38f;
    bool bad = false;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float x = (k & 1) ? hi[0] : lo[0], y = (k & 2) ? hi[1] : lo[1], z = (k & 4) ? hi[2] : lo[2];
        const float qx = M.m[0] * x + M.m[4] * y + M.m[8] * z + M.m[12];
        const float qy = M.m[1] * x + M.m[5] * y + M.m[9] * z + M.m[13];
        const float qw = M.m[3] * x + M.m[7] * y + M.m[11] * z + M.m[15];
        if (!(qw > 0.0f)) bad = true;
        const float iw = 1.0f / qw;
        const float sx = (qx * iw + 1.0f) * 0.5f * fw, sy = (1.0f - qy * iw) * 0.5f * fh;
        if (!(sx == sx) || !(sy == sy)) bad = true;
        px0 = fminf(px0, sx); px1 = fmaxf(px1, sx); py0 = fminf(py0, sy); py1 = fmaxf(py1, sy);
    }
    if (bad) {                                   // cannot happen under CULL_OBJ_HULL; be safe: everything is marched
        for (uint32_t i = 0; i < n_words; ++i) atomicOr(&out[i], 0xffffffffu);
        return;
    }
    // pixels (integer points) inside the grown rectangle; a relative 1e-5 for the f32 projection
    const float gx = 1.5f + 1.0e-5f * fw, gy = 1.5f + 1.0e-5f * fh;
    const float fx0 = fmaxf(ceilf(px0 - gx), 0.0f), fx1 = fminf(floorf(px1 + gx), fw - 1.0f);
    const float fy0 = fmaxf(ceilf(py0 - gy), 0.0f), fy1 = fminf(floorf(py1 + gy), fh - 1.0f);
    if (!(fx0 <= fx1) || !(fy0 <= fy1)) return;          // off screen
    const uint32_t tx0 = static_cast<uint32_t>(fx0) >> 3, tx1 = static_cast<uint32_t>(fx1) >> 3;
    const uint32_t ty0 = static_cast<uint32_t>(fy0) >> 3, ty1 = static_cast<uint32_t>(fy1) >> 3;
    for (uint32_t ty = ty0; ty <= ty1; ++ty) {
        // the bits tx0..tx1 of row ty, word by word
        uint32_t bit = ty * t8x + tx0;
        const uint32_t last = ty * t8x + tx1;
        while (bit <= last) {
            const uint32_t word = bit >> 5, first_in = bit & 31u;
            const uint32_t end_in = (last >> 5) == word ? (last & 31u) : 31u;
            const uint32_t m = (end_in == 31u ? 0xffffffffu : ((1u << (end_in + 1u)) - 1u)) & ~((1u << first_in) - 1u);
            if (word < n_words) atomicOr(&out[word], m);     // (no value comes back: nothing waits)
            bit = (word + 1u) << 5;
        }
    }
}

// The same mask, aggregated in LDS first: one workgroup per block of 8 x 8 x 4 macro cells (a compact piece of the volume: its
// cells project onto neighbouring tiles), the bits ORed into a copy of the mask in LDS, and only the words that are not zero go
// to global memory -- a few thousand device-scope atomics per view instead of one per (cell, tile row, word): ~13 us -> ~3 us
// at 1920 x 1080.  Same rectangles, same bits.  n_words * 4 bytes of dynamic LDS (the caller falls back to the kernel above
// for frames whose mask does not fit).
__global__ __launch_bounds__(256) void volym_tile_mask_lds_kernel(const uint8_t* __restrict__ mc_max, uint32_t mc_n, uint32_t thr_byte, ClipMatrix M,
                                                                  float margin, uint32_t W, uint32_t H, uint32_t t8x, uint32_t n_words,
                                                                  uint32_t* __restrict__ out)
{
    extern __shared__ uint32_t s_mask[];
    for (uint32_t i = threadIdx.x; i < n_words; i += 256u) s_mask[i] = 0u;
    __syncthreads();
    // block -> cell: blocks of 8 x 8 x 4 cells, x fastest
    const uint32_t nbx = (mc_n + 7u) / 8u, nby = (mc_n + 7u) / 8u;
    const uint32_t bx = blockIdx.x % nbx, by = (blockIdx.x / nbx) % nby, bz = blockIdx.x / (nbx * nby);
    const uint32_t cx = bx * 8u + (threadIdx.x & 7u), cy = by * 8u + ((threadIdx.x >> 3) & 7u), cz = bz * 4u + (threadIdx.x >> 6);
    bool all_bits = false;
    if (cx < mc_n && cy < mc_n && cz < mc_n && mc_max[cx + mc_n * (cy + mc_n * cz)] >= thr_byte) {
        const float inv = 1.0f / static_cast<float>(mc_n);
        const float fw = static_cast<float>(W), fh = static_cast<float>(H);
        const float lo[3] = {static_cast<float>(cx) * inv - margin, static_cast<float>(cy) * inv - margin, static_cast<float>(cz) * inv - margin};
        const float hi[3] = {static_cast<float>(cx + 1u) * inv + margin, static_cast<float>(cy + 1u) * inv + margin, static_cast<float>(cz + 1u) * inv + margin};
        float px0 = 3.0e38f, px1 = -3.0e38f, py0 = 3.0e38f, py1 = -3.0e38f;
        bool bad = false;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float x = (k & 1) ? hi[0] : lo[0], y = (k & 2) ? hi[1] : lo[1], z = (k & 4) ? hi[2] : lo[2];
            const float qx = M.m[0] * x + M.m[4] * y + M.m[8] * z + M.m[12];
            const float qy = M.m[1] * x + M.m[5] * y + M.m[9] * z + M.m[13];
            const float qw = M.m[3] * x + M.m[7] * y + M.m[11] * z + M.m[15];
            if (!(qw > 0.0f)) bad = true;
            const float iw = 1.0f / qw;
            const float sx = (qx * iw + 1.0f) * 0.5f * fw, sy = (1.0f - qy * iw) * 0.5f * fh;
            if (!(sx == sx) || !(sy == sy)) bad = true;
            px0 = fminf(px0, sx); px1 = fmaxf(px1, sx); py0 = fminf(py0, sy); py1 = fmaxf(py1, sy);
        }
        if (bad) {
            all_bits = true;                         // cannot happen under CULL_OBJ_HULL; be safe: everything is marched
        } else {
            const float gx = 1.5f + 1.0e-5f * fw, gy = 1.5f + 1.0e-5f * fh;
            const float fx0 = fmaxf(ceilf(px0 - gx), 0.0f), fx1 = fminf(floorf(px1 + gx), fw - 1.0f);
            const float fy0 = fmaxf(ceilf(py0 - gy), 0.0f), fy1 = fminf(floorf(py1 + gy), fh - 1.0f);
            if (fx0 <= fx1 && fy0 <= fy1) {
                const uint32_t tx0 = static_cast<uint32_t>(fx0) >> 3, tx1 = static_cast<uint32_t>(fx1) >> 3;
                const uint32_t ty0 = static_cast<uint32_t>(fy0) >> 3, ty1 = static_cast<uint32_t>(fy1) >> 3;
                for (uint32_t ty = ty0; ty <= ty1; ++ty) {
                    uint32_t bit = ty * t8x + tx0;
                    const uint32_t last = ty * t8x + tx1;
                    while (bit <= last) {
                        const uint32_t word = bit >> 5, first_in = bit & 31u;
                        const uint32_t end_in = (last >> 5) == word ? (last & 31u) : 31u;
                        const uint32_t m = (end_in == 31u ? 0xffffffffu : ((1u << (end_in + 1u)) - 1u)) & ~((1u << first_in) - 1u);
                        if (word < n_words) atomicOr(&s_mask[word], m);
                        bit = (word + 1u) << 5;
                    }
                }
            }
        }
    }
    const bool any_all = __syncthreads_or(all_bits ? 1 : 0) != 0;
    for (uint32_t i = threadIdx.x; i < n_words; i += 256u) {
        const uint32_t v = any_all ? 0xffffffffu : s_mask[i];
        if (v) atomicOr(&out[i], v);
    }
}

// per-cell maxima of the density volume: cell (cx,cy,cz) of the mc_n^3 grid covers the voxels a
// nearest-filter sample with pos in [c/mc_n, (c+1)/mc_n) can select, i.e. floor(pos*n) for those pos.
__global__ __launch_bounds__(256) void volym_macrocell_kernel(const uint8_t* __restrict__ vol, uint8_t* __restrict__ mc_max,
                                                              uint32_t nx, uint32_t ny, uint32_t nz, uint32_t mc_n, uint32_t bricked)
{
    const uint32_t cell = blockIdx.x;
    const uint32_t cx = cell % mc_n, cy = (cell / mc_n) % mc_n, cz = cell / (mc_n * mc_n);
    // voxel range [lo, hi) per axis, one voxel of slack on both sides (float rounding of pos*n)
    auto lo = [&](uint32_t c, uint32_t n) { uint32_t v = static_cast<uint32_t>((static_cast<uint64_t>(c) * n) / mc_n); return v > 0u ? v - 1u : 0u; };
    auto hi = [&](uint32_t c, uint32_t n) { uint32_t v = static_cast<uint32_t>((static_cast<uint64_t>(c + 1u) * n + mc_n - 1u) / mc_n) + 1u; return v < n ? v : n; };
    const uint32_t x0 = lo(cx, nx), x1 = hi(cx, nx), y0 = lo(cy, ny), y1 = hi(cy, ny), z0 = lo(cz, nz), z1 = hi(cz, nz);
    const uint32_t wx = x1 - x0, wy = y1 - y0, wz = z1 - z0;
    const uint32_t total = wx * wy * wz;
    uint32_t m = 0;
    for (uint32_t i = threadIdx.x; i < total; i += 256u) {
        const uint32_t x = x0 + i % wx, y = y0 + (i / wx) % wy, z = z0 + i / (wx * wy);
        const uint32_t v = vol[layout_offset(bricked != 0u, layout_bx(bricked != 0u, nx), layout_bxy(bricked != 0u, nx, ny), x, y, z)];
        m = v > m ? v : m;
    }
    for (int s = 32; s > 0; s >>= 1) { const uint32_t o = __shfl_xor(m, s, 64); m = o > m ? o : m; }
    __shared__ uint32_t s_m[4];
    if ((threadIdx.x & 63u) == 0u) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t r = s_m[0];
        for (int w = 1; w < 4; ++w) r = s_m[w] > r ? s_m[w] : r;
        mc_max[cell] = static_cast<uint8_t>(r);
    }
}

// linear (x fastest) staging copy -> 4x4x4 bricks; one thread per voxel of the padded grid
__global__ __launch_bounds__(256) void volym_rebrick_kernel(const uint8_t* __restrict__ linear, uint8_t* __restrict__ bricked,
                                                            uint32_t nx, uint32_t ny, uint32_t nz)
{
    const uint32_t bx = brick_count(nx), by = brick_count(ny), bz = brick_count(nz);
    const uint64_t total = static_cast<uint64_t>(bx) * by * bz * 64u;
    const uint64_t o = static_cast<uint64_t>(blockIdx.x) * 256u + threadIdx.x;
    if (o >= total) return;
    const uint32_t brick = static_cast<uint32_t>(o >> 6), in = static_cast<uint32_t>(o & 63u);
    const uint32_t x = (brick % bx) * 4u + (in & 3u), y = ((brick / bx) % by) * 4u + ((in >> 2) & 3u), z = (brick / (bx * by)) * 4u + (in >> 4);
    uint8_t v = 0;
    if (x < nx && y < ny && z < nz) v = linear[static_cast<size_t>(x) + static_cast<size_t>(nx) * (y + static_cast<size_t>(ny) * z)];
    bricked[o] = v;
}

// root side of the image gather: world shards of 16x16 tiles -> W x H raster
__global__ __launch_bounds__(256) void volym_assemble_kernel(const uint32_t* __restrict__ gathered, uint32_t* __restrict__ raster,
                                                             uint32_t W, uint32_t H, uint32_t tiles_x, uint32_t n_tiles,
                                                             uint32_t world, uint32_t shard_tiles)
{
    const uint32_t tile = blockIdx.x;
    if (tile >= n_tiles) return;
    const uint32_t rank = tile % world, local = tile / world;
    const uint32_t tx = tile % tiles_x, ty = tile / tiles_x;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t px = ((wave & 1u) << 3) | (lane & 7u);
    const uint32_t py = ((wave >> 1) << 3) | (lane >> 3);
    const uint32_t gx = tx * 16u + px, gy = ty * 16u + py;
    if (gx < W && gy < H)
        raster[static_cast<size_t>(gy) * W + gx] =
            gathered[(static_cast<size_t>(rank) * shard_tiles + local) * 256u + threadIdx.x];
}

// ---- packed shards: most tiles of a frame are constant, and a gather moves what is not ---------------------------------
// A packed shard is [header: one uint2 per local tile, padded to 1 KiB][1 KiB tiles].  header[t] = {slot, 0} for a tile
// stored at tiles[slot], or {PACK_CONSTANT, value} for a tile whose 256 pixels all equal `value` (not stored).  Slots are
// handed out by an atomic counter: their order varies from launch to launch, the header says where each tile went.
constexpr uint32_t PACK_CONSTANT = 0xffffffffu;
__host__ __device__ inline size_t pack_header_bytes(uint32_t shard_tiles) { return (static_cast<size_t>(shard_tiles) * 8u + 1023u) & ~static_cast<size_t>(1023u); }

// one wave per local tile; counters[parity] counts the slots of this launch, counters[parity ^ 1] is zeroed for the next
// launch on the same stream, counters[2] collects an overflow flag (a tile that found no slot)
__global__ __launch_bounds__(64) void volym_pack_shard_kernel(const uint32_t* __restrict__ shard, uint8_t* __restrict__ packed, uint32_t n_local,
                                                              uint32_t shard_tiles, uint32_t max_slots, uint32_t* __restrict__ counters, uint32_t parity)
{
    const uint32_t t = blockIdx.x, lane = threadIdx.x;
    if (t == 0u && lane == 0u) counters[parity ^ 1u] = 0u;
    if (t >= n_local) return;
    const uint4 px = reinterpret_cast<const uint4*>(shard + static_cast<size_t>(t) * 256u)[lane];
    const uint32_t v0 = __builtin_amdgcn_readfirstlane(px.x);
    const bool same = px.x == v0 && px.y == v0 && px.z == v0 && px.w == v0;
    uint2* header = reinterpret_cast<uint2*>(packed);
    if (__ballot(same) == ~0ull) {
        if (lane == 0u) header[t] = make_uint2(PACK_CONSTANT, v0);
        return;
    }
    uint32_t slot = 0;
    if (lane == 0u) slot = atomicAdd(&counters[parity], 1u);
    slot = __builtin_amdgcn_readfirstlane(slot);
    if (slot >= max_slots) {                      // no room: the caller sized the buffer for another frame
        if (lane == 0u) { counters[2] = 1u; header[t] = make_uint2(PACK_CONSTANT, 0u); }
        return;
    }
    if (lane == 0u) header[t] = make_uint2(slot, 0u);
    reinterpret_cast<uint4*>(packed + pack_header_bytes(shard_tiles) + static_cast<size_t>(slot) * 1024u)[lane] = px;
}

// root side: world packed shards, `stride` bytes apart, -> W x H raster (same pixel mapping as volym_assemble_kernel)
__global__ __launch_bounds__(256) void volym_assemble_packed_kernel(const uint8_t* __restrict__ gathered, size_t stride, uint32_t* __restrict__ raster,
                                                                    uint32_t W, uint32_t H, uint32_t tiles_x, uint32_t n_tiles, uint32_t world,
                                                                    uint32_t shard_tiles)
{
    const uint32_t tile = blockIdx.x;
    if (tile >= n_tiles) return;
    const uint32_t rank = tile % world, local = tile / world;
    const uint8_t* base = gathered + static_cast<size_t>(rank) * stride;
    const uint2 h = reinterpret_cast<const uint2*>(base)[local];
    const uint32_t tx = tile % tiles_x, ty = tile / tiles_x;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t gx = tx * 16u + (((wave & 1u) << 3) | (lane & 7u)), gy = ty * 16u + (((wave >> 1) << 3) | (lane >> 3));
    if (gx < W && gy < H) {
        const uint32_t v = h.x == PACK_CONSTANT ? h.y
                                                : reinterpret_cast<const uint32_t*>(base + pack_header_bytes(shard_tiles))[static_cast<size_t>(h.x) * 256u + threadIdx.x];
        raster[static_cast<size_t>(gy) * W + gx] = v;
    }
}

}  // namespace volym
