// VARIANT 3 of the ray-march: a ray pool per workgroup (gfx950).
//
// Why (DESIGN.md "Kernel v3", profiles/r02_*): variant 2 hands every wave one 8x8 pixel tile and keeps its 64 rays in
// registers until the last of them has finished.  The frame then depends on a work list dealt from the measured costs of
// earlier frames (a view nobody measured ran at half speed), the 64 rays of a tile are in different phases (every iteration
// issued the leap code, the sampling code and the shading code for all of them), and the tiles whose rays are all long had
// to be split by the host into depth-parallel quarters.
//
// Here the scheduling unit is the RAY and the phase it is in:
//   * ray state lives in LDS slots of the workgroup (48 bytes: direction, t, t_end, step, alpha, pixel, colour);
//   * three job kinds, each run by any wave on 64 rays that are all in the same phase:
//       SETUP     one 8x8 pixel tile: classification against the hulls / tile mask, ray generation (wgsl:221-241), AABB
//                 clip; the rays that survive get a slot and go to the approach list;
//       APPROACH  rays outside a dense run (wgsl:263-274 with rho < threshold): leaps through provably empty macro
//                 cells in closed form, then K speculative non-dense samples; no shading code.  A ray that meets its
//                 first dense sample moves to the dense list WITHOUT accepting it; a ray that reaches t_end is stored;
//       DENSE     rays inside a dense run: K samples at the fixed minimum step, the six gradient taps of all of them
//                 in flight together with the class bytes, every lane shades its own accepted samples in the
//                 reference's order (wgsl:297-323: plain f32 accumulation, no queue, no atomics); alpha >= 0.95 or
//                 t_end stores the pixel, a non-dense sample sends the ray back to the approach list;
//   * lists are multi-producer / multi-consumer rings of slot numbers in LDS (reserve with one atomic add, claim with
//     one compare-and-swap, entries carry their own "written" flag); producers never wait, so every wait in the kernel
//     is a consumer waiting for a producer that is a few instructions from done;
//   * tiles come from ONE global ticket per frame (workgroups claim a few 16x16 entries at a time from the centre-first
//     list): no learned schedule, the first frame of a view runs like the hundredth.
// Every accepted sample is the reference's: same f32 operations on the control path, in the same order per ray.
//
// Handles the common instantiation (nearest filter, no smoothing, opacity on, no importance mode); everything else runs
// variant 2.
#pragma once

#include "raymarch_pq.h"

namespace volym {

constexpr int PL_WAVES = 16;
constexpr uint32_t PL_SLOTS = 2048;         // ray slots per workgroup
constexpr uint32_t PL_RING = 2048;          // entries per ring (power of two, >= PL_SLOTS)
constexpr uint32_t PL_ENT_RING = 32;        // 16x16 entries a workgroup holds (power of two)
constexpr uint32_t PL_REFILL = 8;           // entries per global ticket
constexpr uint32_t PL_REFILLS_IN_FLIGHT = 2; // waves of a workgroup that may refill at the same time
constexpr uint32_t PL_ENT_LOW = 3;          // refill when fewer whole entries than this are unclaimed
constexpr int PL_K = 4;                     // speculative samples per round
constexpr int PL_A_ROUNDS = 2;              // approach rounds per visit
constexpr uint32_t PL_SPIN_LIMIT = 1u << 22;

struct PoolCtl {
    uint32_t headA, tailA, headD, tailD;
    uint32_t headF, tailF;
    int32_t credits;          // free slots nobody has reserved
    uint32_t sub_ticket;      // 8x8 sub-tiles handed out (4 per entry)
    uint32_t ent_tail;        // entries reserved by refill jobs (an entry is readable once its sequence word says so)
    uint32_t refill_busy;     // refill jobs in flight
    uint32_t exhausted, error;
};

enum : uint32_t { PL_ERR_SPIN = 1u, PL_ERR_WATCHDOG = 2u, PL_ERR_CLAIM = 4u };

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
__device__ __forceinline__ uint32_t pl_take(uint16_t* ring, uint32_t pos, PoolCtl* ctl)
{
    volatile uint16_t* e = ring + (pos & (PL_RING - 1u));
    uint32_t v = *e, spins = 0;
    while (v == 0u) {
        __builtin_amdgcn_s_sleep(1);
        v = *e;
        if (++spins > PL_SPIN_LIMIT) { atomicOr(&ctl->error, PL_ERR_SPIN); return 0u; }
    }
    *e = 0;
    return v - 1u;
}

__device__ __forceinline__ void pl_push(uint32_t* tail, uint16_t* ring, bool pred, uint32_t id, uint32_t lane)
{
    const unsigned long long m = __ballot(pred);
    if (m == 0ull) return;
    pl_fence();
    uint32_t base = 0;
    if (lane == 0u) base = atomicAdd(tail, static_cast<uint32_t>(__popcll(m)));
    base = __builtin_amdgcn_readfirstlane(base);
    if (pred) ring[(base + lane_rank_in_mask(m)) & (PL_RING - 1u)] = static_cast<uint16_t>(id + 1u);
}

template <bool BRICK>
__global__ __launch_bounds__(PL_WAVES * 64) void volym_raymarch_pool_kernel(
    const uint8_t* __restrict__ vol, const FrameTables* __restrict__ tables, const uint8_t* __restrict__ df4,
    const uint2* __restrict__ order, uint32_t n_entries, uint32_t* __restrict__ g_sync, uint32_t* __restrict__ out_shard, uint32_t* __restrict__ out_raster, float4* __restrict__ out_f32,
    uint32_t* __restrict__ dbg, const FrameParams fp)
{
    __shared__ float4 s_tf[256];
    __shared__ __attribute__((aligned(16))) uint8_t s_df[VOLYM_DF_LDS_BYTES];
    __shared__ float4 s_r0[PL_SLOTS];                  // {d.x, d.y, d.z, t}
    __shared__ float4 s_r1[PL_SLOTS];                  // {t_end, cur, alpha, pixel}
    __shared__ float s_rc[PL_SLOTS][3];                // accumulated colour
    __shared__ uint16_t s_ringA[PL_RING], s_ringD[PL_RING], s_ringF[PL_RING];
    __shared__ uint2 s_ent[PL_ENT_RING];
    __shared__ uint32_t s_ent_seq[PL_ENT_RING];       // entry number + 1 once s_ent holds that entry
    __shared__ __attribute__((aligned(16))) PoolCtl s_ctl;

    constexpr uint32_t THREADS = PL_WAVES * 64u;
    const uint32_t flags = (fp.flags & ~(F_IMP_COLORING | F_IMP_RENDERING | F_CONE | F_LINEAR | F_GAUSSIAN)) | F_OPACITY;
    if (fp.tile_mask_spare)
        for (uint32_t wrd = blockIdx.x * THREADS + threadIdx.x; wrd < fp.mask_words; wrd += gridDim.x * THREADS) fp.tile_mask_spare[wrd] = 0u;
    {
        const uint32_t i = threadIdx.x;
        if (i < 256u) s_tf[i] = tables->tf_tab[i];
        const uint32_t n16 = (fp.mc_n * fp.mc_n * fp.mc_n / 2u + 15u) / 16u;
        const uint4* src = reinterpret_cast<const uint4*>(df4);
        uint4* dst = reinterpret_cast<uint4*>(s_df);
        for (uint32_t k = i; k < n16; k += THREADS) dst[k] = src[k];
        for (uint32_t k = i; k < PL_RING; k += THREADS) {
            s_ringA[k] = 0; s_ringD[k] = 0;
            s_ringF[k] = k < PL_SLOTS ? static_cast<uint16_t>(k + 1u) : static_cast<uint16_t>(0);
        }
        if (i == 0u) {
            s_ctl.headA = 0; s_ctl.tailA = 0; s_ctl.headD = 0; s_ctl.tailD = 0;
            s_ctl.headF = 0; s_ctl.tailF = PL_SLOTS; s_ctl.credits = static_cast<int32_t>(PL_SLOTS);
            for (uint32_t k = 0; k < PL_ENT_RING; ++k) s_ent_seq[k] = 0;
            s_ctl.sub_ticket = 0; s_ctl.ent_tail = 0; s_ctl.refill_busy = 0; s_ctl.exhausted = 0; s_ctl.error = 0;
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

    // ---- a finished ray: rgba8unorm store (wgsl:328-329), slot back to the free ring ----
    auto finalize = [&](bool pred, uint32_t id, float r, float gc, float b, float a, uint32_t pix) __attribute__((always_inline)) {
        const unsigned long long m = __ballot(pred);
        if (m == 0ull) return;
        if (pred) {
            out_px[pix] = pack_rgba8(r, gc, b, a);
            if (write_f32) out_f32[pix] = make_float4(r, gc, b, a);
        }
        pl_push(&ctl->tailF, s_ringF, pred, id, lane);
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
    uint32_t dj[5] = {0, 0, 0, 0, 0}, dt[5] = {0, 0, 0, 0, 0}, d_idle = 0, d_raysA = 0, d_raysD = 0;
    const uint32_t d_t0 = dbg ? static_cast<uint32_t>(__builtin_amdgcn_s_memrealtime()) : 0u;
    uint32_t d_mark = d_t0, d_kind = 0;
#define PL_DBG_END() do { if (dbg) { const uint32_t now_ = static_cast<uint32_t>(__builtin_amdgcn_s_memrealtime()); dt[d_kind] += now_ - d_mark; d_mark = now_; } } while (0)
    uint32_t watchdog = 0;
    for (;;) {
        PL_DBG_END();
        if (++watchdog > (1u << 21)) { if (lane == 0u) atomicOr(&ctl->error, PL_ERR_WATCHDOG); break; }
        // ---- scheduler: what is there to do? (racy snapshot; every job re-checks what it claims) ----
        // (heads before tails: a tail never lags its head)
        const uint32_t hA = pl_ld(&ctl->headA), hD = pl_ld(&ctl->headD);
        const uint32_t nA = __builtin_amdgcn_readfirstlane(pl_ld(&ctl->tailA) - hA), nD = __builtin_amdgcn_readfirstlane(pl_ld(&ctl->tailD) - hD);
        const int32_t credits = static_cast<int32_t>(__builtin_amdgcn_readfirstlane(pl_ld(reinterpret_cast<const uint32_t*>(&ctl->credits))));
        const uint32_t subs = __builtin_amdgcn_readfirstlane(pl_ld(&ctl->sub_ticket));
        const uint32_t ents = __builtin_amdgcn_readfirstlane(pl_ld(&ctl->ent_tail));
        const uint32_t busy = __builtin_amdgcn_readfirstlane(pl_ld(&ctl->refill_busy));
        const uint32_t exhausted = __builtin_amdgcn_readfirstlane(pl_ld(&ctl->exhausted));
        const uint32_t subs_left = ents * 4u - min(subs, ents * 4u);
        const bool want_refill = !exhausted && busy < PL_REFILLS_IN_FLIGHT && subs_left < PL_ENT_LOW * 4u;
        const bool can_setup = subs_left != 0u && credits >= 64;
        enum : uint32_t { J_NONE, J_REFILL, J_SETUP, J_A, J_D };
        uint32_t job = J_NONE;
        if (nD >= 64u) job = J_D;
        else if (nA >= 64u) job = J_A;
        else if (want_refill) job = J_REFILL;
        else if (can_setup) job = J_SETUP;
        else if (nD != 0u) job = J_D;
        else if (nA != 0u) job = J_A;
        if (job == J_NONE) {
            if (exhausted && !busy && subs_left == 0u && credits == static_cast<int32_t>(PL_SLOTS)) break;   // nothing left anywhere
            __builtin_amdgcn_s_sleep(8);
            d_idle++; d_kind = 0;
            continue;
        }
        d_kind = job; dj[job]++;

        if (job == J_REFILL) {
            // claim the next few 16x16 entries of the frame; constant ones are stored here, the others become sub-tile work.
            // At most PL_REFILLS_IN_FLIGHT waves do this at a time, and only while fewer than PL_ENT_LOW entries are unclaimed:
            // the ring of PL_ENT_RING entries never wraps onto an entry that still has sub-tiles to hand out.
            uint32_t before = PL_REFILLS_IN_FLIGHT;
            if (lane == 0u) before = atomicAdd(&ctl->refill_busy, 1u);
            bool go_on = __builtin_amdgcn_readfirstlane(before) < PL_REFILLS_IN_FLIGHT;
            if (go_on) {
                // the snapshot above is older than the count: look again
                const uint32_t ents_now = __builtin_amdgcn_readfirstlane(pl_ld(&ctl->ent_tail));
                const uint32_t subs_now = __builtin_amdgcn_readfirstlane(pl_ld(&ctl->sub_ticket));
                go_on = pl_ld(&ctl->exhausted) == 0u && ents_now * 4u - min(subs_now, ents_now * 4u) < PL_ENT_LOW * 4u;
            }
            if (!go_on) { if (lane == 0u) atomicSub(&ctl->refill_busy, 1u); continue; }
            uint32_t tk = 0;
            if (lane == 0u) tk = atomicAdd(&g_sync[0], PL_REFILL);
            tk = __builtin_amdgcn_readfirstlane(tk);
            uint32_t pushed = 0, keep_lt = 0, keep_xy = 0;     // lane i keeps the i-th entry that has to be marched
            bool none_left = false;
            for (uint32_t i = 0; i < PL_REFILL; ++i) {
                const uint32_t e = tk + i;
                if (e >= n_entries) { none_left = true; break; }
                const uint2 ent = order[e];
                const uint32_t lt = __builtin_amdgcn_readfirstlane(ent.x), txy = __builtin_amdgcn_readfirstlane(ent.y);
                const uint32_t tx16 = txy & 0xffffu, ty16 = txy >> 16;
                uint32_t cls = TILE_MARCH;
                if (culling) {
                    bool masked16 = false;
                    if (fp.cull & CULL_TILE_MASK)
                        masked16 = !(tile_mask_bit(fp, tx16 * 2u, ty16 * 2u) || tile_mask_bit(fp, tx16 * 2u + 1u, ty16 * 2u) ||
                                     tile_mask_bit(fp, tx16 * 2u, ty16 * 2u + 1u) || tile_mask_bit(fp, tx16 * 2u + 1u, ty16 * 2u + 1u));
                    cls = classify_tile(fp, hull_edge, lane, static_cast<float>(tx16 * 16u), static_cast<float>(ty16 * 16u), 15.0f, masked16);
                }
                if (cls >= TILE_FILL_EMPTY) {
                    fill16(lt, tx16, ty16, cls);
                } else {
                    if (lane == pushed) { keep_lt = lt; keep_xy = txy; }
                    pushed++;
                }
            }
            if (pushed != 0u) {
                uint32_t pos = 0;
                if (lane == 0u) pos = atomicAdd(&ctl->ent_tail, pushed);
                pos = __builtin_amdgcn_readfirstlane(pos);
                if (lane < pushed) {
                    const uint32_t e = pos + lane;
                    s_ent[e & (PL_ENT_RING - 1u)] = make_uint2(keep_lt, keep_xy);
                    pl_fence();
                    *reinterpret_cast<volatile uint32_t*>(&s_ent_seq[e & (PL_ENT_RING - 1u)]) = e + 1u;
                }
            }
            pl_fence();
            if (lane == 0u) {
                if (none_left) *reinterpret_cast<volatile uint32_t*>(&ctl->exhausted) = 1u;
                pl_fence();
                atomicSub(&ctl->refill_busy, 1u);
            }
            continue;
        }

        if (job == J_SETUP) {
            // ---- one 8x8 pixel tile: reserve 64 slots' worth of credit, then a sub-tile ----
            int32_t had = 0;
            if (lane == 0u) had = atomicSub(&ctl->credits, 64);
            if (__builtin_amdgcn_readfirstlane(had) < 64) { if (lane == 0u) atomicAdd(&ctl->credits, 64); __builtin_amdgcn_s_sleep(2); continue; }
            uint32_t got = 0, s_tk = 0;
            uint2 ent = make_uint2(0u, 0u);
            if (lane == 0u) {
                for (;;) {
                    s_tk = pl_ld(&ctl->sub_ticket);
                    const uint32_t et = pl_ld(&ctl->ent_tail);
                    if ((s_tk >> 2) >= et) break;
                    if (pl_ld(&s_ent_seq[(s_tk >> 2) & (PL_ENT_RING - 1u)]) != (s_tk >> 2) + 1u) break;     // reserved, not written yet: later
                    const uint2* ep = &s_ent[(s_tk >> 2) & (PL_ENT_RING - 1u)];                              // read before the claim: valid if the claim succeeds
                    ent = make_uint2(pl_ld(&ep->x), pl_ld(&ep->y));
                    if (atomicCAS(&ctl->sub_ticket, s_tk, s_tk + 1u) == s_tk) { got = 1; break; }
                }
            }
            if (__builtin_amdgcn_readfirstlane(got) == 0u) { if (lane == 0u) atomicAdd(&ctl->credits, 64); continue; }
            const uint32_t sub = __builtin_amdgcn_readfirstlane(s_tk) & 3u;
            const uint32_t local_tile = __builtin_amdgcn_readfirstlane(ent.x), txy = __builtin_amdgcn_readfirstlane(ent.y);
            const uint32_t tx = txy & 0xffffu, ty = txy >> 16;
            const uint32_t px_in_sub = lane & 7u, py_in_sub = lane >> 3;
            const uint32_t gx = tx * 16u + ((sub & 1u) << 3) + px_in_sub, gy = ty * 16u + ((sub >> 1) << 3) + py_in_sub;
            const bool in_frame = gx < fp.W && gy < fp.H;             // wgsl:217-219
            const uint32_t pix = (flags & F_RASTER) ? gy * fp.W + gx : local_tile * 256u + sub * 64u + lane;
            const bool store_here = in_frame || !(flags & F_RASTER);  // shard layout: pixels outside the frame are zero
            uint32_t tclass = TILE_MARCH;
            if (culling) {
                const bool masked8 = (fp.cull & CULL_TILE_MASK) != 0u && !tile_mask_bit(fp, tx * 2u + (sub & 1u), ty * 2u + (sub >> 1));
                tclass = classify_tile(fp, hull_edge, lane, static_cast<float>(tx * 16u + ((sub & 1u) << 3)), static_cast<float>(ty * 16u + ((sub >> 1) << 3)), 7.0f, masked8);
            }
            if (tclass >= TILE_FILL_EMPTY) {
                if (store_here) {
                    out_px[pix] = !in_frame ? 0u : tclass == TILE_FILL_MISS ? 0xff000000u : 0u;
                    if (write_f32) out_f32[pix] = make_float4(0.0f, 0.0f, 0.0f, tclass == TILE_FILL_MISS ? 1.0f : 0.0f);
                }
                if (lane == 0u) atomicAdd(&ctl->credits, 64);
                continue;
            }
            Ray ray;
            ray.o = o; ray.d = v3(0.0f, 0.0f, 0.0f); ray.t_entry = 0.0f; ray.t_exit = 0.0f; ray.hit = false;
            if (in_frame) ray = make_ray(fp, gx, gy);
            bool active = in_frame && ray.hit;
            const float miss_a = active ? 0.0f : 1.0f;                // miss: (0,0,0,1) wgsl:239
            float t = active ? ray.t_entry : 0.0f;
            if (culling && tclass == TILE_HIT_TEST) active = false;  // hit rays of this tile see nothing dense: (0,0,0,0)
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
            if (!active && store_here) {
                out_px[pix] = in_frame ? (miss_a == 1.0f ? 0xff000000u : 0u) : 0u;
                if (write_f32) out_f32[pix] = make_float4(0.0f, 0.0f, 0.0f, miss_a);
            }
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
                    id = pl_take(s_ringF, fbase + rank, ctl);
                    s_r0[id] = make_float4(ray.d.x, ray.d.y, ray.d.z, t);
                    s_r1[id] = make_float4(t_end, base, 0.0f, __uint_as_float(pix));
                    s_rc[id][0] = 0.0f; s_rc[id][1] = 0.0f; s_rc[id][2] = 0.0f;
                }
                pl_push(&ctl->tailA, s_ringA, mine, id, lane);
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
            d_raysA += n;
            const bool valid = lane < n;
            uint32_t id = 0;
            if (valid) id = pl_take(s_ringA, hb + lane, ctl);
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
                uint32_t D = (static_cast<uint32_t>(s_df[ci >> 1]) >> ((ci & 1u) * 4u)) & 15u;
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
            finalize(valid && ended, id, s_rc[id][0], s_rc[id][1], s_rc[id][2], r1.z, __float_as_uint(r1.w));
            pl_push(&ctl->tailD, s_ringD, valid && to_dense, id, lane);
            pl_push(&ctl->tailA, s_ringA, valid && !ended && !to_dense, id, lane);
            continue;
        }

        {
            // =============================== DENSE: 64 rays inside a dense run ===============================
            uint32_t hb = 0;
            const uint32_t n = pl_claim(&ctl->headD, &ctl->tailD, 64u, lane, hb);
            if (n == 0u) continue;
            d_raysD += n;
            const bool valid = lane < n;
            uint32_t id = 0;
            if (valid) id = pl_take(s_ringD, hb + lane, ctl);
            const float4 r0 = s_r0[id];
            const float4 r1 = s_r1[id];
            const V3 d = v3(r0.x, r0.y, r0.z);
            const float t_end = r1.x;
            float acc_a = r1.z;
            float cr = s_rc[id][0], cg = s_rc[id][1], cb = s_rc[id][2];
            const V3 Hh = ray_half_vector(d);                       // wgsl:199-205: a constant of the ray (COLOUR)
            float ts[PL_K + 1];
            ts[0] = r0.w;
#pragma unroll
            for (int k = 0; k < PL_K; ++k) ts[k + 1] = ts[k] + min_step;            // wgsl:264, :325 inside a dense run
            // class byte and the six gradient taps (wgsl:181-188) of all K samples in flight together
            uint32_t bs[PL_K];
            int gxd[PL_K], gyd[PL_K], gzd[PL_K];
#pragma unroll
            for (int k = 0; k < PL_K; ++k) {
                const V3 pos = o + d * ts[k];                                       // wgsl:251
                const int ix = texel_nearest(pos.x, g.fnx, g.hix), iy = texel_nearest(pos.y, g.fny, g.hiy), iz = texel_nearest(pos.z, g.fnz, g.hiz);
                const float of = 0.01f;
                const int ixp = texel_nearest(pos.x + of, g.fnx, g.hix), ixm = texel_nearest(pos.x - of, g.fnx, g.hix);
                const int iyp = texel_nearest(pos.y + of, g.fny, g.hiy), iym = texel_nearest(pos.y - of, g.fny, g.hiy);
                const int izp = texel_nearest(pos.z + of, g.fnz, g.hiz), izm = texel_nearest(pos.z - of, g.fnz, g.hiz);
                bs[k] = vol[voxel_offset(g, ix, iy, iz)];
                const int bxp = vol[voxel_offset(g, ixp, iy, iz)], bxm = vol[voxel_offset(g, ixm, iy, iz)];
                const int byp = vol[voxel_offset(g, ix, iyp, iz)], bym = vol[voxel_offset(g, ix, iym, iz)];
                const int bzp = vol[voxel_offset(g, ix, iy, izp)], bzm = vol[voxel_offset(g, ix, iy, izm)];
                gxd[k] = bxp - bxm; gyd[k] = byp - bym; gzd[k] = bzp - bzm;       // b/255 differences up to the common factor (cancels in normalize)
            }
            bool run = valid;           // still inside the run: every sample so far was accepted and dense
            bool ended = false, to_approach = false;
            float t = ts[0], cur = min_step;
#pragma unroll
            for (int k = 0; k < PL_K; ++k) {
                const bool go = run && ts[k] < t_end && acc_a < 0.95f;              // wgsl:250
                if (run && !go) { ended = true; t = ts[k]; }
                const bool dense = bs[k] >= fp.thr_byte;                             // <=> b/255 >= thr
                const bool emit = go && dense;
                if (__ballot(emit) != 0ull) {
                    const float4 ca = s_tf[bs[k]];                                   // TF colour and 1 - pow(1 - A, cur * 100) (wgsl:297-314)
                    const float w = (1.0f - acc_a) * ca.w;                           // wgsl:315
                    const V3 shaded = blinn_phong_h(v3(ca.x, ca.y, ca.z), v3(static_cast<float>(gxd[k]), static_cast<float>(gyd[k]), static_cast<float>(gzd[k])), Hh);
                    if (emit) {
                        cr = __builtin_fmaf(shaded.x, w, cr); cg = __builtin_fmaf(shaded.y, w, cg); cb = __builtin_fmaf(shaded.z, w, cb);   // wgsl:316 (COLOUR)
                        acc_a += w;                                                  // wgsl:317
                    }
                }
                if (go && !dense) {                                                  // the run is over: this sample only advances (wgsl:266-274)
                    to_approach = true;
                    cur = cur_after_run;
                    t = ts[k] + cur_after_run;
                }
                run = emit;
            }
            if (run) {                                                               // all K accepted: the next sample would be at ts[K]
                t = ts[PL_K];
                if (!(t < t_end && acc_a < 0.95f)) { ended = true; run = false; }    // wgsl:250, one visit early
            }
            if (to_approach && !(t < t_end)) { to_approach = false; ended = true; }  // (alpha < 0.95 still holds)
            if (valid && !ended) {
                s_r0[id].w = t;
                s_r1[id].y = cur;
                s_r1[id].z = acc_a;
                s_rc[id][0] = cr; s_rc[id][1] = cg; s_rc[id][2] = cb;
            }
            finalize(valid && ended, id, cr, cg, cb, acc_a, __float_as_uint(r1.w));
            pl_push(&ctl->tailA, s_ringA, valid && to_approach, id, lane);
            pl_push(&ctl->tailD, s_ringD, valid && run, id, lane);
            continue;
        }
    }
    if (dbg) {
        PL_DBG_END();
        if (lane == 0u) {
            uint32_t* r = dbg + (static_cast<size_t>(blockIdx.x) * PL_WAVES + (threadIdx.x >> 6)) * 16u;
            r[0] = d_t0; r[1] = d_mark; r[2] = dj[1]; r[3] = dj[2]; r[4] = dj[3]; r[5] = dj[4]; r[6] = d_idle; r[7] = d_raysA; r[8] = d_raysD;
            r[9] = dt[0]; r[10] = dt[1]; r[11] = dt[2]; r[12] = dt[3]; r[13] = dt[4]; r[14] = watchdog; r[15] = 0;
        }
    }
#undef PL_DBG_END
    // g_sync: {frame ticket, workgroups done, error bits}.  The last workgroup to finish rewinds the ticket for the next launch
    // (stream order; also what a replayed HIP graph needs: nothing outside the kernel resets anything).
    __syncthreads();
    if (threadIdx.x == 0u) {
        const uint32_t err = pl_ld(&ctl->error);
        if (err) atomicOr(&g_sync[2], err);
        __threadfence();
        if (atomicAdd(&g_sync[1], 1u) == gridDim.x - 1u) {
            atomicExch(&g_sync[0], 0u);
            atomicExch(&g_sync[1], 0u);
        }
    }
}

}  // namespace volym
