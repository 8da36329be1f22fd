// VARIANT 3 of the ray-march: a ray pool per workgroup (gfx950).
//
// Why (DESIGN.md "Kernel v3", profiles/r02_*, profiles/r03_*): variant 2 hands every wave one 8x8 pixel tile and keeps its 64
// rays in registers until the last of them has finished.  The frame then depends on a work list dealt from the measured costs
// of earlier frames (a view nobody measured ran at half speed: no static assignment of TILES to 256 workgroups balances a frame
// whose cost sits in ~2 000 of 32 400 tiles), the 64 rays of a tile are in different phases (every iteration issued the leap
// code, the sampling code and the shading code for all of them), and the tiles whose rays are all long had to be split by the
// host into depth-parallel quarters.
//
// Here the scheduling unit is the RAY and the phase it is in, and the frame is dealt to the workgroups PIXEL BLOCK by pixel
// block:
//   * inside the screen rectangle of the occupied macro cells' AABB the frame is a lattice of 4x2-pixel blocks; every
//     superblock of 16x16 blocks (64x32 pixels) gives each of the 256 workgroups one block.  Every workgroup so renders a
//     regular 1/256 sample of the image: the same share of every structure on screen, whatever the scene and the view
//     (measured on the bench frame: heaviest workgroup 1.05-1.10 x the mean, against 1.4-1.7 x for any dealing of 8x8 tiles
//     that does not know their costs).  No global ticket, no atomics on global memory, nothing learned.  Outside that
//     rectangle every pixel is constant: 16x16 tiles, dealt round robin, filled with 16-byte stores;
//   * ray state lives in LDS slots of the workgroup (48 bytes: direction, t, t_end, step, alpha, pixel, colour);
//   * job kinds, each run by any wave of the workgroup on 64 items that are all in the same phase:
//       FILL      a 16x16 tile outside the rectangle (constant, or a cube hit test per pixel on the cube's silhouette);
//       CLASSIFY  64 lattice pixels against the projected hulls and the tile mask (per pixel, same margins as variant 2's tile
//                 classification): constants are stored, the rest go to the pixel list;
//       SETUP     64 listed pixels: ray generation (wgsl:221-241), AABB clip; the rays that survive get a slot and go to the
//                 approach list, the others are stored;
//       APPROACH  rays outside a dense run (wgsl:263-274 with rho < threshold): leaps through provably empty macro
//                 cells in closed form, then K speculative non-dense samples; no shading code.  A ray that meets its
//                 first dense sample moves to the dense list WITHOUT accepting it; a ray that reaches t_end is stored;
//       DENSE     rays inside a dense run: K samples at the fixed minimum step, the six gradient taps of all of them
//                 in flight together with the class bytes, every lane shades its own accepted samples in the
//                 reference's order (wgsl:297-323: plain f32 accumulation, no queue, no atomics); alpha >= 0.95 or
//                 t_end stores the pixel, a non-dense sample sends the ray back to the approach list;
//   * lists are multi-producer / multi-consumer rings in LDS (reserve with one atomic add, claim with one
//     compare-and-swap, entries carry their own "written" flag); producers never wait, so every wait in the kernel
//     is a consumer waiting for a producer that is a few instructions from done (and every wait is bounded: a wait that
//     runs out sets an error word that the blocking host calls report).
// Every accepted sample is the reference's: same f32 operations on the control path, in the same order per ray.
//
// Handles the common instantiation (nearest filter, no smoothing, opacity on, no importance mode); everything else runs
// variant 2.
#pragma once

#include <type_traits>

#include "raymarch_pq.h"

namespace volym {

constexpr int PL_WAVES = 12;
constexpr uint32_t PL_SLOTS = 2048;         // ray slots per workgroup
constexpr uint32_t PL_RING = 2048;          // entries per ray ring (power of two, >= PL_SLOTS)
constexpr uint32_t PL_PRING = 4096;         // entries of the pixel ring (power of two)
constexpr uint32_t PL_P_ROOM = PL_PRING - 64u * (PL_WAVES + 1u);   // classify only while the pixel ring holds no more than this
constexpr int PL_K = 4;                     // speculative samples per round
constexpr int PL_A_ROUNDS = 3;              // approach rounds per visit
constexpr uint32_t PL_SPIN_LIMIT = 1u << 22;
// the lattice: blocks of PL_BW x PL_BH pixels, superblocks of 16 x 16 blocks
constexpr uint32_t PL_BW = 4, PL_BH = 2;
constexpr uint32_t PL_SBW = 16u * PL_BW, PL_SBH = 16u * PL_BH;      // 64 x 32 pixels (multiples of 16: the rectangle is made of whole tiles)

struct PoolCtl {
    uint32_t headA, tailA, headD, tailD;
    uint32_t headF, tailF, headP, tailP;
    int32_t credits;          // free slots nobody has reserved
    uint32_t c_ticket;        // classify jobs handed out
    uint32_t f_ticket;        // fill jobs handed out
    uint32_t error;
};

enum : uint32_t { PL_ERR_SPIN = 1u, PL_ERR_WATCHDOG = 2u, PL_ERR_CLAIM = 4u, PL_ERR_BOUNDS = 8u };

// ---- list primitives (wave-uniform results) ----
__device__ __forceinline__ void pl_fence()
{
    // LDS executes a wave's instructions in order: a compiler barrier is all a publish / claim needs (a workgroup-scope
    // fence would also wait for the global loads in flight)
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
}

__device__ __forceinline__ uint32_t pl_ld(const uint32_t* p) { return *reinterpret_cast<const volatile uint32_t*>(p); }

__device__ __forceinline__ uint32_t pl_claim(uint32_t* head, uint32_t* tail, uint32_t want, uint32_t lane, uint32_t& base)
{
    uint32_t h = 0, n = 0;
    if (lane == 0u) {
        for (;;) {
            h = pl_ld(head);
            const uint32_t t = pl_ld(tail);
            n = min(want, t - h);
            if (n == 0u) break;
            if (atomicCAS(head, h, h + n) == h) break;
        }
    }
    base = __builtin_amdgcn_readfirstlane(h);
    return __builtin_amdgcn_readfirstlane(n);
}

// entry `pos` of a ring: wait until its producer has written it, take it, leave the place empty
template <class T>
__device__ __forceinline__ uint32_t pl_take(T* ring, uint32_t pos, uint32_t mask, PoolCtl* ctl)
{
    volatile T* e = ring + (pos & mask);
    uint32_t v = *e, spins = 0;
    while (v == 0u) {
        __builtin_amdgcn_s_sleep(1);
        v = *e;
        if (++spins > PL_SPIN_LIMIT) { atomicOr(&ctl->error, PL_ERR_SPIN); return 1u; }     // (a valid value: the frame is reported broken, nothing is indexed out of range)
    }
    *e = 0;
    return v;
}

// push `value` (non-zero) of the lanes flagged by pred
template <class T>
__device__ __forceinline__ void pl_push(uint32_t* tail, T* ring, uint32_t mask, bool pred, uint32_t value, uint32_t lane)
{
    const unsigned long long m = __ballot(pred);
    if (m == 0ull) return;
    pl_fence();
    uint32_t base = 0;
    if (lane == 0u) base = atomicAdd(tail, static_cast<uint32_t>(__popcll(m)));
    base = __builtin_amdgcn_readfirstlane(base);
    if (pred) ring[(base + lane_rank_in_mask(m)) & mask] = static_cast<T>(value);
}

// The pixel (x, y) against the two projected hulls, with variant 2's margins (classify_tile with a rectangle of extent 0).
__device__ __forceinline__ uint32_t classify_pixel(const FrameParams& fp, float x, float y, bool masked_out)
{
    const float margin = 1.5f;
    bool out_cube = false, in_cube = true, out_obj = false;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float vc = __builtin_fmaf(fp.hull[0][e][0], x, __builtin_fmaf(fp.hull[0][e][1], y, fp.hull[0][e][2]));
        const bool valid_c = fp.hull[0][e][3] > 0.5f;
        out_cube = out_cube || (valid_c && vc < -margin);
        in_cube = in_cube && (!valid_c || vc > margin);
        const float vo = __builtin_fmaf(fp.hull[1][e][0], x, __builtin_fmaf(fp.hull[1][e][1], y, fp.hull[1][e][2]));
        out_obj = out_obj || (fp.hull[1][e][3] > 0.5f && vo < -margin);
    }
    const bool cube_ok = (fp.cull & CULL_CUBE_HULL) != 0u, obj_ok = (fp.cull & CULL_OBJ_HULL) != 0u;
    out_cube = cube_ok && out_cube;
    in_cube = cube_ok && in_cube;
    out_obj = masked_out || (fp.cull & CULL_NOTHING_DENSE) != 0u || (obj_ok && out_obj);
    if (out_cube) return TILE_FILL_MISS;
    if (out_obj) return in_cube ? TILE_FILL_EMPTY : TILE_HIT_TEST;
    return TILE_MARCH;
}

template <bool BRICK>
__global__ __launch_bounds__(PL_WAVES * 64) void volym_raymarch_pool_kernel(
    const uint8_t* __restrict__ vol, const FrameTables* __restrict__ tables, const uint8_t* __restrict__ df4,
    uint32_t* __restrict__ g_sync, uint32_t* __restrict__ out_shard, uint32_t* __restrict__ out_raster, float4* __restrict__ out_f32,
    uint32_t* __restrict__ dbg, const FrameParams fp)
{
    __shared__ float4 s_tf[256];
    __shared__ __attribute__((aligned(16))) uint8_t s_df[VOLYM_DF_LDS_BYTES];
    __shared__ float4 s_r0[PL_SLOTS];                  // {d.x, d.y, d.z, t}
    __shared__ float4 s_r1[PL_SLOTS];                  // {t_end, cur, alpha, pixel}
    __shared__ uint32_t s_rc[PL_SLOTS][3];             // accumulated colour, 4.28 fixed point
    __shared__ uint16_t s_ringA[PL_RING], s_ringD[PL_RING], s_ringF[PL_RING];     // slot number + 1
    __shared__ uint32_t s_ringP[PL_PRING];             // pixels waiting for their ray: x | y << 15 | 1 << 30 | hit test only << 31
    __shared__ __attribute__((aligned(16))) PoolCtl s_ctl;

    constexpr uint32_t THREADS = PL_WAVES * 64u;
    const uint32_t flags = (fp.flags & ~(F_IMP_COLORING | F_IMP_RENDERING | F_CONE | F_LINEAR | F_GAUSSIAN)) | F_OPACITY;
    if (fp.tile_mask_spare)
        for (uint32_t wrd = blockIdx.x * THREADS + threadIdx.x; wrd < fp.mask_words; wrd += gridDim.x * THREADS) fp.tile_mask_spare[wrd] = 0u;
    {
        const uint32_t i = threadIdx.x;
        if (i < 256u) s_tf[i] = tables->tf_tab[i];
        const uint32_t n16 = VOLYM_DF_IN_LDS(fp) ? (fp.mc_n * fp.mc_n * fp.mc_n / 2u + 15u) / 16u : 0u;      // a finer grid is read from global memory (L1 / L2)
        const uint4* src = reinterpret_cast<const uint4*>(df4);
        uint4* dst = reinterpret_cast<uint4*>(s_df);
        for (uint32_t k = i; k < n16; k += THREADS) dst[k] = src[k];
        for (uint32_t k = i; k < PL_RING; k += THREADS) {
            s_ringA[k] = 0; s_ringD[k] = 0;
            s_ringF[k] = k < PL_SLOTS ? static_cast<uint16_t>(k + 1u) : static_cast<uint16_t>(0);
        }
        for (uint32_t k = i; k < PL_PRING; k += THREADS) s_ringP[k] = 0;
        if (i == 0u) {
            s_ctl.headA = 0; s_ctl.tailA = 0; s_ctl.headD = 0; s_ctl.tailD = 0; s_ctl.headP = 0; s_ctl.tailP = 0;
            s_ctl.headF = 0; s_ctl.tailF = PL_SLOTS; s_ctl.credits = static_cast<int32_t>(PL_SLOTS);
            s_ctl.c_ticket = 0; s_ctl.f_ticket = 0; s_ctl.error = 0;
        }
    }
    __syncthreads();

    const uint32_t lane = threadIdx.x & 63u;
    GridT<BRICK> g;
    grid_init(g, vol, vol, fp.nx, fp.ny, fp.nz);
    const float base = fp.base_step, min_step = fp.min_step;
    const float cur_after_run = __builtin_fminf(base, min_step * 1.5f);     // wgsl:263-269: the step after a dense run's last sample
    const float mcf = static_cast<float>(fp.mc_n), inv_mc = 1.0f / mcf;
    const V3 o = v3(fp.eye[0], fp.eye[1], fp.eye[2]);
    const bool culling = fp.cull != 0u;
    const HullEdge hull_edge = load_hull_edge(fp, lane);
    uint32_t* const out_px = (flags & F_RASTER) ? out_raster : out_shard;
    const bool write_f32 = (flags & F_RASTER) && (flags & F_WRITE_F32);
    PoolCtl* const ctl = &s_ctl;
    // every pixel store goes through here: an index outside the output (a bug, never an input) is reported, not written
    const uint32_t out_limit = (flags & F_RASTER) ? fp.W * fp.H : fp.n_local * 256u;
    auto store_px = [&](uint32_t pix, uint32_t packed, float a) __attribute__((always_inline)) {
        if (pix < out_limit) {
            out_px[pix] = packed;
            if (write_f32) out_f32[pix] = make_float4(0.0f, 0.0f, 0.0f, a);
        } else {
            atomicOr(&ctl->error, PL_ERR_BOUNDS);
        }
    };

    // ---- the share of this workgroup ----
    // rectangle of the lattice (pixels, whole superblocks; empty when x1 <= x0), its superblocks, the lattice cells of this
    // workgroup (cell c of 256 belongs to workgroup c % G) and the classify jobs of 64 pixels = 8 (superblock, cell) pairs
    const uint32_t G = gridDim.x, wg = blockIdx.x;
    const uint32_t rx0 = fp.rect[0], ry0 = fp.rect[1], rx1 = fp.rect[2], ry1 = fp.rect[3];
    const uint32_t rw = rx1 > rx0 ? (rx1 - rx0) / PL_SBW : 0u, rh = ry1 > ry0 ? (ry1 - ry0) / PL_SBH : 0u;
    const uint32_t n_sb = rw * rh;
    const uint32_t n_cell = wg < 256u ? (256u - wg + G - 1u) / G : 0u;
    const uint32_t n_pairs = n_sb * n_cell;
    const uint32_t n_cjobs = (n_pairs + 7u) / 8u;
    const uint32_t n_fjobs = fp.n_local > wg ? (fp.n_local - wg + G - 1u) / G : 0u;      // local 16x16 tiles wg, wg + G, ...
    const float inv_rw = rw ? 1.0f / static_cast<float>(rw) : 0.0f;

    // output index of pixel (gx, gy); false: not ours to store (outside the frame in raster mode, another rank's tile)
    auto pixel_index = [&](uint32_t gx, uint32_t gy, uint32_t& pix) __attribute__((always_inline)) -> bool {
        if (flags & F_RASTER) { pix = gy * fp.W + gx; return gx < fp.W && gy < fp.H; }
        const uint32_t tile = (gy >> 4) * fp.tiles_x + (gx >> 4);
        const uint32_t lt = tile / fp.world;
        pix = lt * 256u + ((((gy >> 3) & 1u) * 2u + ((gx >> 3) & 1u)) << 6) + ((gy & 7u) << 3) + (gx & 7u);
        return tile - lt * fp.world == fp.rank && (gx >> 4) < fp.tiles_x && tile < fp.n_tiles;
    };

    // ---- a finished ray: rgba8unorm store (wgsl:328-329), slot back to the free ring ----
    auto finalize = [&](bool pred, uint32_t id, float r, float gc, float b, float a, uint32_t pix) __attribute__((always_inline)) {
        const unsigned long long m = __ballot(pred);
        if (m == 0ull) return;
        if (pred) {
            if (pix < out_limit) {
                out_px[pix] = pack_rgba8(r, gc, b, a);
                if (write_f32) out_f32[pix] = make_float4(r, gc, b, a);
            } else {
                atomicOr(&ctl->error, PL_ERR_BOUNDS);
            }
        }
        pl_push(&ctl->tailF, s_ringF, PL_RING - 1u, pred, id + 1u, lane);
        pl_fence();
        if (lane == 0u) atomicAdd(&ctl->credits, static_cast<int32_t>(__popcll(m)));
    };

    // ---- constant 16x16 tile ----
    auto fill16 = [&](uint32_t lt, uint32_t tx16, uint32_t ty16, uint32_t cls) __attribute__((always_inline)) {
        const uint32_t packed = cls == TILE_FILL_MISS ? 0xff000000u : 0u;       // (0,0,0,1) wgsl:239 / (0,0,0,0) wgsl:328
        if (flags & F_RASTER) {
            const uint32_t gx4 = tx16 * 16u + (lane & 3u) * 4u, gy4 = ty16 * 16u + (lane >> 2);
            if (gy4 < fp.H && (fp.W & 3u) == 0u && (reinterpret_cast<uintptr_t>(out_raster) & 15u) == 0u && gx4 < fp.W && !write_f32) {
                *reinterpret_cast<uint4*>(out_raster + static_cast<size_t>(gy4) * fp.W + gx4) = make_uint4(packed, packed, packed, packed);
            } else if (gy4 < fp.H) {
                for (uint32_t i = 0; i < 4u; ++i)
                    if (gx4 + i < fp.W) {
                        out_raster[static_cast<size_t>(gy4) * fp.W + gx4 + i] = packed;
                        if (write_f32) out_f32[static_cast<size_t>(gy4) * fp.W + gx4 + i] = make_float4(0.0f, 0.0f, 0.0f, cls == TILE_FILL_MISS ? 1.0f : 0.0f);
                    }
            }
        } else {
            for (uint32_t i = 0; i < 4u; ++i) {
                const uint32_t sl = lane * 4u + i, sub4 = sl >> 6, in = sl & 63u;
                const uint32_t gxx = tx16 * 16u + ((sub4 & 1u) << 3) + (in & 7u), gyy = ty16 * 16u + ((sub4 >> 1) << 3) + (in >> 3);
                out_shard[static_cast<size_t>(lt) * 256u + sl] = (gxx < fp.W && gyy < fp.H) ? packed : 0u;
            }
        }
    };

    // development timeline (dbg != nullptr): per wave {start, end (10 ns ticks), jobs by kind, idle turns, rays visited, ticks by kind}
#if VOLYM_DEV_SWITCHES
    uint32_t dj[7] = {0, 0, 0, 0, 0, 0, 0}, dt[7] = {0, 0, 0, 0, 0, 0, 0}, d_idle = 0, d_raysA = 0, d_raysD = 0, d_vL[3] = {0, 0, 0}, d_rL[3] = {0, 0, 0};
    const uint32_t d_t0 = dbg ? static_cast<uint32_t>(__builtin_amdgcn_s_memrealtime()) : 0u;
    uint32_t d_mark = d_t0, d_kind = 0;
#define PL_DBG_END() do { if (dbg) { const uint32_t now_ = static_cast<uint32_t>(__builtin_amdgcn_s_memrealtime()); dt[d_kind] += now_ - d_mark; d_mark = now_; } } while (0)
#define PL_DBG(x) x
#else
#define PL_DBG_END() do { } while (0)
#define PL_DBG(x)
#endif
    uint32_t watchdog = 0;
    uint32_t idle_turns = 0;
    for (;;) {
        PL_DBG_END();
        if (++watchdog > (1u << 21)) { if (lane == 0u) atomicOr(&ctl->error, PL_ERR_WATCHDOG); break; }
        // ---- scheduler: what is there to do? (racy snapshot; every job re-checks what it claims; heads before tails) ----
        // one LDS access for the whole control block (lane i reads word i), then cross-lane reads: a turn of the scheduler costs one
        // round trip, not ten.  (A tail read a moment before its head can lag it: a negative count is an empty list.)
        const uint32_t cw = pl_ld(reinterpret_cast<const uint32_t*>(ctl) + min(lane, static_cast<uint32_t>(sizeof(PoolCtl) / 4u - 1u)));
        auto cword = [&](const uint32_t* field) __attribute__((always_inline)) {
            return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(cw), static_cast<int>(field - reinterpret_cast<const uint32_t*>(ctl))));
        };
        auto count = [&](const uint32_t* head, const uint32_t* tail) __attribute__((always_inline)) {
            const int32_t n = static_cast<int32_t>(cword(tail) - cword(head));
            return n > 0 ? static_cast<uint32_t>(n) : 0u;
        };
        const uint32_t nA = count(&ctl->headA, &ctl->tailA), nD = count(&ctl->headD, &ctl->tailD), nP = count(&ctl->headP, &ctl->tailP);
        const int32_t credits = static_cast<int32_t>(cword(reinterpret_cast<const uint32_t*>(&ctl->credits)));
        const uint32_t c_tk = cword(&ctl->c_ticket), f_tk = cword(&ctl->f_ticket);
        enum : uint32_t { J_NONE, J_FILL, J_CLASSIFY, J_SETUP, J_A, J_D };
        uint32_t job = J_NONE;
        if (nD >= 64u) job = J_D;
        else if (nA >= 64u) job = J_A;
        else if (nP >= 64u && credits >= 64) job = J_SETUP;
        else if (c_tk < n_cjobs && nP <= PL_P_ROOM) job = J_CLASSIFY;
        else if (f_tk < n_fjobs) job = J_FILL;
        // short lists: a wave that has found nothing to do for a few turns takes what is there (the longest-waiting wave first:
        // the others see an empty list and go on waiting); a short dense list gets more lanes per ray (below)
        else if (nD >= 33u || (nD != 0u && idle_turns >= 3u)) job = J_D;
        else if (nA >= 48u || (nA != 0u && idle_turns >= 4u)) job = J_A;
        else if (nP != 0u && credits >= 64 && idle_turns >= 4u) job = J_SETUP;
        if (job == J_NONE) {
            // nothing to claim.  Every ray alive holds a slot and every set-up job reserves its slots before it takes its pixels: all
            // slots free, all tickets gone and the pixel list empty means that only a classify job still in flight could add work --
            // its wave is then the one that does it (a wave that leaves early costs parallelism, never pixels)
            if (c_tk >= n_cjobs && f_tk >= n_fjobs && nP == 0u && credits == static_cast<int32_t>(PL_SLOTS)) break;
            __builtin_amdgcn_s_sleep(2);
            idle_turns++;
            PL_DBG(d_idle++; d_kind = 0;)
            continue;
        }
        PL_DBG(d_kind = job; dj[job]++;)
        idle_turns = 0;

        if (job == J_FILL) {
            // ---- a 16x16 tile of this shard outside the lattice rectangle: no ray of it can meet anything dense ----
            uint32_t tk0 = 0;
            if (lane == 0u) tk0 = atomicAdd(&ctl->f_ticket, 4u);
            tk0 = __builtin_amdgcn_readfirstlane(tk0);
            for (uint32_t tk = tk0; tk < min(tk0 + 4u, n_fjobs); ++tk) {
                const uint32_t lt = wg + G * tk;
                const uint32_t tile = lt * fp.world + fp.rank;
                const uint32_t tx16 = tile % fp.tiles_x, ty16 = tile / fp.tiles_x;
                if (tx16 * 16u >= rx0 && tx16 * 16u < rx1 && ty16 * 16u >= ry0 && ty16 * 16u < ry1) continue;      // the lattice's
                const uint32_t cls = culling ? classify_tile(fp, hull_edge, lane, static_cast<float>(tx16 * 16u), static_cast<float>(ty16 * 16u), 15.0f, true) : TILE_HIT_TEST;
                if (cls >= TILE_FILL_EMPTY) { fill16(lt, tx16, ty16, cls); continue; }
                // on the silhouette of the cube (or no usable hull): the cube hit test per pixel (wgsl:236-241), nothing to march
                for (uint32_t sub = 0; sub < 4u; ++sub) {
                    const uint32_t gx = tx16 * 16u + ((sub & 1u) << 3) + (lane & 7u), gy = ty16 * 16u + ((sub >> 1) << 3) + (lane >> 3);
                    const bool in_frame = gx < fp.W && gy < fp.H;
                    bool hit = false;
                    if (in_frame) hit = make_ray(fp, gx, gy).hit;
                    const uint32_t pix = (flags & F_RASTER) ? gy * fp.W + gx : lt * 256u + sub * 64u + lane;
                    if (in_frame || !(flags & F_RASTER)) store_px(pix, !in_frame ? 0u : hit ? 0u : 0xff000000u, hit ? 0.0f : 1.0f);
                }
            }
            continue;
        }

        if (job == J_CLASSIFY) {
            // ---- 64 lattice pixels: 8 blocks of this workgroup's cells in 8 consecutive superblocks ----
            uint32_t tk = 0;
            if (lane == 0u) tk = atomicAdd(&ctl->c_ticket, 1u);
            tk = __builtin_amdgcn_readfirstlane(tk);
            if (tk >= n_cjobs) continue;
            const uint32_t pair = tk * 8u + (lane >> 3);
            const uint32_t ci = n_cell > 1u ? pair % n_cell : 0u, sb = n_cell > 1u ? pair / n_cell : pair;
            const uint32_t cell = wg + G * ci;                                   // lattice cell 0..255 inside the superblock
            uint32_t sby = static_cast<uint32_t>(static_cast<float>(sb) * inv_rw);   // sb / rw (sb < 2^22: one correction suffices)
            if (sby * rw > sb) sby--;
            if ((sby + 1u) * rw <= sb) sby++;
            const uint32_t sbx = sb - sby * rw;
            const uint32_t gx = rx0 + sbx * PL_SBW + (cell & 15u) * PL_BW + (lane & 3u);
            const uint32_t gy = ry0 + sby * PL_SBH + (cell >> 4) * PL_BH + ((lane >> 2) & 1u);
            uint32_t pix = 0;
            const bool ours = pair < n_pairs && pixel_index(gx, gy, pix);
            const bool in_frame = gx < fp.W && gy < fp.H;
            uint32_t cls = TILE_MARCH;
            if (culling && ours && in_frame) {
                bool masked = false;
                if (fp.cull & CULL_TILE_MASK) {
                    const uint32_t bit = (gy >> 3) * fp.mask_t8x + (gx >> 3);
                    masked = ((fp.tile_mask[bit >> 5] >> (bit & 31u)) & 1u) == 0u;
                }
                cls = classify_pixel(fp, static_cast<float>(gx), static_cast<float>(gy), masked);
            }
            if (ours && (!in_frame || cls >= TILE_FILL_EMPTY)) store_px(pix, !in_frame ? 0u : cls == TILE_FILL_MISS ? 0xff000000u : 0u, cls == TILE_FILL_MISS ? 1.0f : 0.0f);
            const bool keep = ours && in_frame && cls < TILE_FILL_EMPTY;
            pl_push(&ctl->tailP, s_ringP, PL_PRING - 1u, keep, gx | (gy << 15) | (1u << 30) | (cls == TILE_HIT_TEST ? 1u << 31 : 0u), lane);
            continue;
        }

        if (job == J_SETUP) {
            // ---- 64 listed pixels: reserve 64 slots' worth of credit, then the pixels ----
            int32_t had = 0;
            if (lane == 0u) had = atomicSub(&ctl->credits, 64);
            if (__builtin_amdgcn_readfirstlane(had) < 64) { if (lane == 0u) atomicAdd(&ctl->credits, 64); __builtin_amdgcn_s_sleep(2); continue; }
            uint32_t pb = 0;
            const uint32_t np = pl_claim(&ctl->headP, &ctl->tailP, 64u, lane, pb);
            if (np == 0u) { if (lane == 0u) atomicAdd(&ctl->credits, 64); continue; }
            const bool have = lane < np;
            uint32_t code = 0;
            if (have) code = pl_take(s_ringP, pb + lane, PL_PRING - 1u, ctl);
            const uint32_t gx = code & 0x7fffu, gy = (code >> 15) & 0x7fffu;
            const bool hit_only = (code >> 31) != 0u;
            uint32_t pix = 0;
            (void)pixel_index(gx, gy, pix);
            Ray ray;
            ray.o = o; ray.d = v3(0.0f, 0.0f, 0.0f); ray.t_entry = 0.0f; ray.t_exit = 0.0f; ray.hit = false;
            if (have) ray = make_ray(fp, gx, gy);
            bool active = have && ray.hit;
            const float miss_a = active ? 0.0f : 1.0f;                // miss: (0,0,0,1) wgsl:239
            float t = active ? ray.t_entry : 0.0f;
            if (hit_only) active = false;                             // hit rays outside the object's hull see nothing dense: (0,0,0,0)
            float t_end = ray.t_exit;
            if (culling && (fp.cull & CULL_AABB) && active) {
                // slab test against the AABB of the occupied macro cells (conservative arithmetic, as variant 2)
                const float rx = __builtin_amdgcn_rcpf(ray.d.x), ry = __builtin_amdgcn_rcpf(ray.d.y), rz = __builtin_amdgcn_rcpf(ray.d.z);
                const float ax0 = (fp.aabb_lo[0] - ray.o.x) * rx, ax1 = (fp.aabb_hi[0] - ray.o.x) * rx;
                const float ay0 = (fp.aabb_lo[1] - ray.o.y) * ry, ay1 = (fp.aabb_hi[1] - ray.o.y) * ry;
                const float az0 = (fp.aabb_lo[2] - ray.o.z) * rz, az1 = (fp.aabb_hi[2] - ray.o.z) * rz;
                float tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(ax0, ax1), __builtin_fminf(ay0, ay1)), __builtin_fminf(az0, az1));
                float tf = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(ax0, ax1), __builtin_fmaxf(ay0, ay1)), __builtin_fmaxf(az0, az1));
                tn = tn - 2.0e-5f * __builtin_fabsf(tn) - 1.0e-6f;
                tf = tf + 2.0e-5f * __builtin_fabsf(tf) + 1.0e-6f;
                const bool outside_static = (ray.d.x == 0.0f && (ray.o.x < fp.aabb_lo[0] || ray.o.x > fp.aabb_hi[0])) ||
                                            (ray.d.y == 0.0f && (ray.o.y < fp.aabb_lo[1] || ray.o.y > fp.aabb_hi[1])) ||
                                            (ray.d.z == 0.0f && (ray.o.z < fp.aabb_lo[2] || ray.o.z > fp.aabb_hi[2]));
                if (outside_static || !(tn <= tf)) {
                    active = false;
                } else {
                    t_end = __builtin_fminf(t_end, tf);
                    replay_saturated(t, __builtin_fminf(tn, t_end), base);       // the empty steps in front of the AABB, cur == base throughout
                }
            }
            if ((fp.cull & CULL_NOTHING_DENSE) && culling) active = false;
            active = active && t < t_end;                              // wgsl:250
            if (have && !active) store_px(pix, miss_a == 1.0f ? 0xff000000u : 0u, miss_a);
            const unsigned long long live = __ballot(active);
            const uint32_t n_live = static_cast<uint32_t>(__popcll(live));
            if (n_live != 0u) {
                uint32_t fbase = 0;
                const uint32_t gotF = pl_claim(&ctl->headF, &ctl->tailF, n_live, lane, fbase);
                if (gotF != n_live) { if (lane == 0u) atomicOr(&ctl->error, PL_ERR_CLAIM); }
                uint32_t id = 0;
                const uint32_t rank = lane_rank_in_mask(live);
                const bool mine = active && rank < gotF;
                if (mine) {
                    id = pl_take(s_ringF, fbase + rank, PL_RING - 1u, ctl) - 1u;
                    s_r0[id] = make_float4(ray.d.x, ray.d.y, ray.d.z, t);
                    s_r1[id] = make_float4(t_end, base, 0.0f, __uint_as_float(pix));
                    s_rc[id][0] = 0u; s_rc[id][1] = 0u; s_rc[id][2] = 0u;
                }
                pl_push(&ctl->tailA, s_ringA, PL_RING - 1u, mine, id + 1u, lane);
            }
            pl_fence();
            if (lane == 0u && n_live != 64u) atomicAdd(&ctl->credits, static_cast<int32_t>(64u - n_live));
            continue;
        }

        if (job == J_A) {
            // =============================== APPROACH: 64 rays outside a dense run ===============================
            uint32_t hb = 0;
            const uint32_t n = pl_claim(&ctl->headA, &ctl->tailA, 64u, lane, hb);
            if (n == 0u) continue;
            PL_DBG(d_raysA += n;)
            const bool valid = lane < n;
            uint32_t id = 0;
            if (valid) id = pl_take(s_ringA, hb + lane, PL_RING - 1u, ctl) - 1u;
            const float4 r0 = s_r0[id];
            const float4 r1 = s_r1[id];
            const V3 d = v3(r0.x, r0.y, r0.z);
            float t = r0.w, cur = r1.y;
            const float t_end = r1.x;
            const float idx_ = __builtin_amdgcn_rcpf(d.x), idy_ = __builtin_amdgcn_rcpf(d.y), idz_ = __builtin_amdgcn_rcpf(d.z);
            const float nox = -o.x * idx_, noy = -o.y * idy_, noz = -o.z * idz_;
            bool active = valid, ended = false, to_dense = false;
            // one leap through the box of empty macro cells around the ray's position (exact replay of the skipped steps,
            // wgsl:263-274 with rho < threshold); true if the lane leapt
            auto try_leap = [&](bool want) __attribute__((always_inline)) -> bool {
                const V3 pos = o + d * t;
                const float cxf = __builtin_floorf(pos.x * mcf), cyf = __builtin_floorf(pos.y * mcf), czf = __builtin_floorf(pos.z * mcf);
                const int cx = static_cast<int>(cxf), cy = static_cast<int>(cyf), cz = static_cast<int>(czf);
                const bool in_range = static_cast<uint32_t>(cx | cy | cz) < fp.mc_n;
                const uint32_t ci = in_range ? mad_u24(mad_u24(static_cast<uint32_t>(cz), fp.mc_n, static_cast<uint32_t>(cy)), fp.mc_n, static_cast<uint32_t>(cx)) : 0u;
                uint32_t D = (static_cast<uint32_t>(VOLYM_DF_IN_LDS(fp) ? s_df[ci >> 1] : df4[ci >> 1]) >> ((ci & 1u) * 4u)) & 15u;
                if (!(want && in_range)) D = 0u;
                if (D >= PQ_MIN_LEAP_D) {
                    const float eps = 4.0e-5f;
                    const float a = static_cast<float>(D - 1u) * inv_mc - eps;
                    const float lx = __builtin_fmaf(cxf, inv_mc, -a), hx = __builtin_fmaf(cxf, inv_mc, a + inv_mc);
                    const float ly = __builtin_fmaf(cyf, inv_mc, -a), hy = __builtin_fmaf(cyf, inv_mc, a + inv_mc);
                    const float lz = __builtin_fmaf(czf, inv_mc, -a), hz = __builtin_fmaf(czf, inv_mc, a + inv_mc);
                    const float ex = __builtin_fmaxf(__builtin_fmaf(lx, idx_, nox), __builtin_fmaf(hx, idx_, nox));
                    const float ey = __builtin_fmaxf(__builtin_fmaf(ly, idy_, noy), __builtin_fmaf(hy, idy_, noy));
                    const float ez = __builtin_fmaxf(__builtin_fmaf(lz, idz_, noz), __builtin_fmaf(hz, idz_, noz));
                    float te = __builtin_fminf(__builtin_fminf(ex, ey), ez);
                    te = te - 2.0e-5f * __builtin_fabsf(te);
                    const float t_stop = __builtin_fminf(te, t_end);
                    while (cur < base && t < t_stop) {          // at most four steps until the step size is back at `base`
                        cur = __builtin_fminf(base, cur * 1.5f);
                        t += cur;
                    }
                    replay_saturated(t, t_stop, base);
                    return true;
                }
                return false;
            };
#pragma unroll 1
            for (int round = 0; round < PL_A_ROUNDS; ++round) {
                if (__ballot(active) == 0ull) break;
                // ---- leaps: a ray in open space usually finds a second box behind the first ----
                const bool l1 = try_leap(active);
                if (__ballot(l1) != 0ull) {
                    if (l1 && !(t < t_end)) { active = false; ended = true; }
                    const bool l2 = try_leap(active && l1);
                    if (l2 && !(t < t_end)) { active = false; ended = true; }
                }
                // ---- K samples under the prediction "not dense": positions follow from the step rule alone ----
                float ts[PL_K + 1], cp[PL_K + 1];
                ts[0] = t; cp[0] = cur;
#pragma unroll
                for (int k = 0; k < PL_K; ++k) {
                    cp[k + 1] = __builtin_fminf(base, cp[k] * 1.5f);              // wgsl:266-268
                    ts[k + 1] = ts[k] + cp[k + 1];                                 // wgsl:272
                }
                uint32_t bs[PL_K];
#pragma unroll
                for (int k = 0; k < PL_K; ++k) bs[k] = vol[nearest_offset(g, o + d * ts[k])];    // wgsl:251-258; clamped offsets: no guard
                // the march accepts the leading samples that are inside [t, t_end) and not dense (wgsl:250, :271-274)
                bool ok = active;
                float t_new = ts[0], cur_new = cp[0];
#pragma unroll
                for (int k = 0; k < PL_K; ++k) {
                    ok = ok && ts[k] < t_end && bs[k] < fp.thr_byte;
                    t_new = ok ? ts[k + 1] : t_new;
                    cur_new = ok ? cp[k + 1] : cur_new;
                }
                if (active) {
                    t = t_new; cur = cur_new;
                    if (!(t < t_end)) { ended = true; active = false; }            // wgsl:250
                    else if (!ok) { to_dense = true; active = false; }             // the sample at t is dense: the dense job takes it
                }
            }
            // ---- state back to the slot, the ray to its next list ----
            if (valid && !ended) {
                s_r0[id].w = t;
                s_r1[id].y = cur;
            }
            finalize(valid && ended, id, static_cast<float>(s_rc[id][0]) * PQ_FIX_INV, static_cast<float>(s_rc[id][1]) * PQ_FIX_INV, static_cast<float>(s_rc[id][2]) * PQ_FIX_INV, r1.z, __float_as_uint(r1.w));
            pl_push(&ctl->tailD, s_ringD, PL_RING - 1u, valid && to_dense, id + 1u, lane);
            pl_push(&ctl->tailA, s_ringA, PL_RING - 1u, valid && !ended && !to_dense, id + 1u, lane);
            continue;
        }

        {
            // =============================== DENSE: rays inside a dense run ===============================
            // 64 rays, one lane each -- or, when fewer are waiting, 32 rays on two lanes each or 16 rays on four: the lanes of a ray
            // take its next samples alternately (sample s on lane s % L), so that a short list advances its rays 2 or 4 times as far
            // per visit (the chains of dependent visits are what a workgroup ends on).  The acceptance runs replicated on the
            // lanes of a ray, on the same values in the same order; colour is summed in 4.28 fixed point (integer adds commute:
            // the pixel does not depend on how many lanes its ray had, i.e. on the scheduling).
            const uint32_t lanes_per_ray = nD >= 33u ? 1u : nD >= 17u ? 2u : 4u;
            auto dense_visit = [&](auto LC) __attribute__((always_inline)) {
                constexpr int L = decltype(LC)::value;
                constexpr int LOG = L == 1 ? 0 : L == 2 ? 1 : 2;
                uint32_t hb = 0;
                const uint32_t n = pl_claim(&ctl->headD, &ctl->tailD, 64u / L, lane, hb);
                if (n == 0u) return;
                PL_DBG(d_raysD += n; d_vL[LOG]++; d_rL[LOG] += n;)
                const uint32_t sub = lane & (L - 1u), r = lane >> LOG;
                const bool valid = r < n;
                uint32_t id = 0;
                if (valid) id = pl_take(s_ringD, hb + r, PL_RING - 1u, ctl) - 1u;     // (the lanes of a ray read, then clear, the same entry)
                const float4 r0 = s_r0[id];
                const float4 r1 = s_r1[id];
                const V3 d = v3(r0.x, r0.y, r0.z);
                const float t_end = r1.x;
                float acc_a = r1.z;
                const V3 Hh = ray_half_vector(d);                   // wgsl:199-205: a constant of the ray (COLOUR)
                // this lane's samples: s = k * L + sub, at t + s * min_step by repeated addition (wgsl:264, :325 inside a dense run)
                float ts[PL_K];
                {
                    float tt = r0.w;
#pragma unroll
                    for (int q = 1; q < L; ++q) tt = sub >= static_cast<uint32_t>(q) ? tt + min_step : tt;
#pragma unroll
                    for (int k = 0; k < PL_K; ++k) {
                        ts[k] = tt;
#pragma unroll
                        for (int q = 0; q < L; ++q) tt += min_step;
                    }
                }
                // class byte and the six gradient taps (wgsl:181-188) of all K samples in flight together
                uint32_t bs[PL_K];
                int gxd[PL_K], gyd[PL_K], gzd[PL_K];
#pragma unroll
                for (int k = 0; k < PL_K; ++k) {
                    const V3 pos = o + d * ts[k];                                   // wgsl:251
                    const int ix = texel_nearest(pos.x, g.fnx, g.hix), iy = texel_nearest(pos.y, g.fny, g.hiy), iz = texel_nearest(pos.z, g.fnz, g.hiz);
                    const float of = 0.01f;
                    const int ixp = texel_nearest(pos.x + of, g.fnx, g.hix), ixm = texel_nearest(pos.x - of, g.fnx, g.hix);
                    const int iyp = texel_nearest(pos.y + of, g.fny, g.hiy), iym = texel_nearest(pos.y - of, g.fny, g.hiy);
                    const int izp = texel_nearest(pos.z + of, g.fnz, g.hiz), izm = texel_nearest(pos.z - of, g.fnz, g.hiz);
                    bs[k] = vol[voxel_offset(g, ix, iy, iz)];
                    const int bxp = vol[voxel_offset(g, ixp, iy, iz)], bxm = vol[voxel_offset(g, ixm, iy, iz)];
                    const int byp = vol[voxel_offset(g, ix, iyp, iz)], bym = vol[voxel_offset(g, ix, iym, iz)];
                    const int bzp = vol[voxel_offset(g, ix, iy, izp)], bzm = vol[voxel_offset(g, ix, iy, izm)];
                    gxd[k] = bxp - bxm; gyd[k] = byp - bym; gzd[k] = bzp - bzm;   // b/255 differences up to the common factor (cancels in normalize)
                }
                // ---- acceptance, sample by sample in the reference's order (wgsl:250, :263-274, :313-318); identical on the lanes of a ray ----
                float4 tf_my[PL_K];
#pragma unroll
                for (int k = 0; k < PL_K; ++k) tf_my[k] = s_tf[bs[k]];              // TF colour and 1 - pow(1 - A, cur * 100) (wgsl:297-314)
                bool run = valid;           // still inside the run: every sample so far was accepted and dense
                bool ended = false, to_approach = false;
                float t = r0.w, cur = min_step, tq = r0.w;        // tq: position of the sample the acceptance is looking at
                bool emit_my[PL_K];
                float w_my[PL_K];
#pragma unroll
                for (int k = 0; k < PL_K; ++k) {
                    emit_my[k] = false; w_my[k] = 0.0f;
                    const unsigned long long dmask = __ballot(bs[k] >= fp.thr_byte);   // <=> b/255 >= thr
                    const uint32_t dbits = static_cast<uint32_t>(dmask >> (lane & ~(L - 1u))) & ((1u << L) - 1u);
                    const int a_bits = __float_as_int(tf_my[k].w);
#pragma unroll
                    for (int q = 0; q < L; ++q) {
                        float alpha_q;
                        if (L == 1) alpha_q = tf_my[k].w;
                        else if (L == 2) alpha_q = __int_as_float(q == 0 ? __builtin_amdgcn_mov_dpp(a_bits, 0xA0, 0xf, 0xf, true) : __builtin_amdgcn_mov_dpp(a_bits, 0xF5, 0xf, 0xf, true));
                        else alpha_q = __int_as_float(q == 0 ? __builtin_amdgcn_mov_dpp(a_bits, 0x00, 0xf, 0xf, true) : q == 1 ? __builtin_amdgcn_mov_dpp(a_bits, 0x55, 0xf, 0xf, true)
                                                      : q == 2 ? __builtin_amdgcn_mov_dpp(a_bits, 0xAA, 0xf, 0xf, true) : __builtin_amdgcn_mov_dpp(a_bits, 0xFF, 0xf, 0xf, true));
                        const bool go = run && tq < t_end && acc_a < 0.95f;             // wgsl:250
                        if (run && !go) ended = true;
                        const bool dense = ((dbits >> q) & 1u) != 0u;
                        const bool emit = go && dense;
                        const float w = (1.0f - acc_a) * alpha_q;                        // wgsl:315
                        if (emit) acc_a += w;                                            // wgsl:317
                        if (go && !dense) {                                              // the run is over: this sample only advances (wgsl:266-274)
                            to_approach = true;
                            cur = cur_after_run;
                            t = tq + cur_after_run;
                        }
                        if (sub == static_cast<uint32_t>(q)) { emit_my[k] = emit; w_my[k] = w; }
                        run = emit;
                        tq += min_step;
                    }
                }
                if (run) {                                                               // all accepted: the next sample would be at tq
                    t = tq;
                    if (!(t < t_end && acc_a < 0.95f)) { ended = true; run = false; }    // wgsl:250, one visit early
                }
                if (to_approach && !(t < t_end)) { to_approach = false; ended = true; }  // (alpha < 0.95 still holds)
                // ---- shading of this lane's accepted samples (COLOUR), 4.28 fixed point ----
                uint32_t fr = 0, fg = 0, fb = 0;
#pragma unroll
                for (int k = 0; k < PL_K; ++k) {
                    if (__ballot(emit_my[k]) != 0ull) {
                        const V3 shaded = blinn_phong_h(v3(tf_my[k].x, tf_my[k].y, tf_my[k].z), v3(static_cast<float>(gxd[k]), static_cast<float>(gyd[k]), static_cast<float>(gzd[k])), Hh);
                        if (emit_my[k]) {                                                // wgsl:316
                            fr += static_cast<uint32_t>(__builtin_fmaf(shaded.x * w_my[k], PQ_FIX_SCALE, 0.5f));
                            fg += static_cast<uint32_t>(__builtin_fmaf(shaded.y * w_my[k], PQ_FIX_SCALE, 0.5f));
                            fb += static_cast<uint32_t>(__builtin_fmaf(shaded.z * w_my[k], PQ_FIX_SCALE, 0.5f));
                        }
                    }
                }
                if (L >= 2) {
                    fr += static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(fr), 0xB1, 0xf, 0xf, true));    // quad_perm(1,0,3,2)
                    fg += static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(fg), 0xB1, 0xf, 0xf, true));
                    fb += static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(fb), 0xB1, 0xf, 0xf, true));
                }
                if (L >= 4) {
                    fr += static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(fr), 0x4E, 0xf, 0xf, true));    // quad_perm(2,3,0,1)
                    fg += static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(fg), 0x4E, 0xf, 0xf, true));
                    fb += static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(fb), 0x4E, 0xf, 0xf, true));
                }
                const uint32_t cr = s_rc[id][0] + fr, cg = s_rc[id][1] + fg, cb = s_rc[id][2] + fb;
                const bool writer = valid && sub == 0u;
                if (writer && !ended) {
                    s_r0[id].w = t;
                    s_r1[id].y = cur;
                    s_r1[id].z = acc_a;
                    s_rc[id][0] = cr; s_rc[id][1] = cg; s_rc[id][2] = cb;
                }
                finalize(writer && ended, id, static_cast<float>(cr) * PQ_FIX_INV, static_cast<float>(cg) * PQ_FIX_INV, static_cast<float>(cb) * PQ_FIX_INV, acc_a, __float_as_uint(r1.w));
                pl_push(&ctl->tailA, s_ringA, PL_RING - 1u, writer && to_approach, id + 1u, lane);
                pl_push(&ctl->tailD, s_ringD, PL_RING - 1u, writer && run, id + 1u, lane);
            };
            if (lanes_per_ray == 1u) dense_visit(std::integral_constant<int, 1>());
            else if (lanes_per_ray == 2u) dense_visit(std::integral_constant<int, 2>());
            else dense_visit(std::integral_constant<int, 4>());
            continue;
        }
    }
#if VOLYM_DEV_SWITCHES
    if (dbg) {
        PL_DBG_END();
        if (lane == 0u) {
            uint32_t* r = dbg + (static_cast<size_t>(blockIdx.x) * PL_WAVES + (threadIdx.x >> 6)) * 24u;
            r[0] = d_t0; r[1] = d_mark; r[2] = d_idle; r[3] = d_raysA; r[4] = d_raysD; r[5] = watchdog;
            for (int k = 0; k < 6; ++k) { r[6 + k] = dj[k]; r[12 + k] = dt[k]; }
            for (int k = 0; k < 3; ++k) { r[18 + k] = d_vL[k]; r[21 + k] = d_rL[k]; }
        }
    }
#endif
#undef PL_DBG_END
#undef PL_DBG
    // g_sync[2]: error bits of the frame (a bounded wait that ran out: a bug, never an input)
    __syncthreads();
    if (threadIdx.x == 0u) {
        const uint32_t err = pl_ld(&ctl->error);
        if (err) atomicOr(&g_sync[2], err);
    }
}

}  // namespace volym
