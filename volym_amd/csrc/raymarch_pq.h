// VARIANT 2 of the ray-march: persistent workgroups + per-wave shading queue (gfx950).
//
// Why this shape (profiles/ and DESIGN.md "Kernel v2"): with one pixel per lane and a monolithic
// loop, the kernel time was the latency of its longest rays -- ~80 dependent iterations of
// {byte gather -> classify -> 6 gradient gathers -> Blinn-Phong -> composite}, executed by waves in
// which a handful of lanes were still alive.  Three observations remove that chain:
//
//  1. CONTROL needs one byte per sample.  Positions, the step state machine, rho >= threshold, the
//     opacity alpha_step (a table of the byte), the alpha < 0.95 exit: none of it needs the gradient
//     or the shading.  So a lane runs only the control path, and every dense sample it accepts is
//     pushed into a per-wave LDS queue as {pos, weight w = (1-alpha)*alpha_step, class byte, owner}.
//  2. COLOUR is a sum of independent terms w_k * shade(pos_k).  Whenever 64 samples are queued the
//     whole wave shades them, one sample per lane, whoever owns them, and adds the result to the
//     owner's accumulator in LDS.  The accumulators are 4.28 fixed point and are updated with integer
//     atomics, so the sum is independent of the order of arrival: deterministic and identical on
//     every GPU/sharding.  (|error| <= 4e-9 per term; the budget is 1e-4.)
//  3. The control path itself is speculated K samples at a time: predict that the next K samples
//     have the class (dense / not dense) of the last one, which fixes their positions through the
//     step state machine; issue the K byte gathers together; then accept samples in order with the
//     REAL classes and stop at the first misprediction.  Accepted samples are bit-identical to the
//     sequential march (same f32 operations in the same order).
//
// Work distribution: persistent workgroups of PQ_WAVES waves, one per CU (it owns the CU's LDS).  Every workgroup reads a
// list of items -- workgroup b the entries b, b+G, b+2G, ... of `order` -- and its waves draw them through a ticket counter
// in LDS: dynamic balance inside the workgroup, no global atomics, nothing to reset between launches.  The first frame of
// a context runs the host's centre-first list (the orbit camera targets the volume centre, src/camera.rs:23); a launch can be
// asked to record a counted cost per list entry, from which a feedback thread on the host deals the next list (raymarch.hip,
// "cost feedback"): most expensive entries first, the most expensive tiles as depth-parallel quarter items (bit 31),
// constant 16x16 tiles as super fill items (bit 30).  Entries are {item code, x | y << 16 of the entry's 16x16 tile}.
//
// Item kinds and their loops (all bit-identical to the sequential march):
//   * 8x8 tile, one lane per ray, K speculative samples per iteration ("classic");
//   * 4x4 quarter tile, four lanes per ray ("depth-parallel"): for the long chains of dependent samples a frame ends on;
//   * fill items: tiles the hulls prove constant.
// Template parameters: TABLE (nearest filter without smoothing: density is a table of the byte), COUNT (instrumented),
// TRACE (development), IMP (general flag handling) / IR (importance rendering on the specialised paths), BRICK (layout).
//
// Exact culling (results unchanged, DESIGN.md "Culling"):
//   * no sample outside the AABB of the occupied macro cells can reach the threshold, so a ray is
//     replayed (t += cur in f32, no fetches) up to the AABB entry and abandoned at its exit; a ray that
//     misses the AABB is final at (0,0,0,0);
//   * a wave tile whose 8x8 pixel rectangle lies outside the projected cube stores (0,0,0,1) and one
//     inside the cube but outside the projected AABB stores (0,0,0,0) without any per-ray work.  Both
//     tests use hulls projected on the host and a 1.5 pixel safety margin; tiles that straddle an
//     edge take the exact per-ray path.
// The instrumented (COUNT) launch disables culling: it counts what the reference would fetch.
#pragma once

#include <type_traits>

#include "raymarch_device.h"

namespace volym {

constexpr int PQ_WAVES = 16;             // waves per workgroup of the common instantiation (see WAVES below)
constexpr int PQ_WAVES_WIDE = 12;        // ... of the instantiations that need more than 128 VGPRs
constexpr uint32_t PQ_MIN_LEAP_D = 2;    // smallest distance-field value worth a leap
constexpr int PQ_DP_DEPTH = 2;           // depth-parallel items: samples per lane and iteration (4 lanes per ray)
constexpr uint32_t PQ_NO_ITEM = 0xffffffffu;   // padding of the work list
constexpr int PQ_ITEMS_LDS = 512;       // work-list entries staged in LDS per workgroup (the rest stay in global memory)
// ring entries per wave: < 64 left over from the last iteration + 64 per speculative sample of this one.  One workgroup
// per CU owns the whole 160 KB of LDS, so the queue is sized for the shading to run at ONE point of the loop.
constexpr int pq_qcap(int k) { return 64 + 64 * k; }
constexpr float PQ_FIX_SCALE = 268435456.0f;        // 2^28
constexpr float PQ_FIX_INV = 1.0f / 268435456.0f;

__device__ __forceinline__ uint32_t lane_rank_in_mask(unsigned long long mask)
{
    return __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mask >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mask), 0u));
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

enum : uint32_t { TILE_MARCH = 0, TILE_HIT_TEST = 1, TILE_FILL_EMPTY = 2, TILE_FILL_MISS = 3 };

// Classify the pixel rectangle [x0, x0+extent] x [y0, y0+extent] against the two projected hulls (wave-uniform
// result; lane l evaluates hull l>>5, edge (l>>2)&7, corner l&3).
// The lane's edge (a, b, c, valid) is fetched once per kernel (HullEdge): indexing the kernel arguments by lane is a
// memory gather, and a fill tile is little else.
struct HullEdge { float a, b, c; bool valid; };
__device__ __forceinline__ HullEdge load_hull_edge(const FrameParams& fp, uint32_t lane)
{
    // one gather from the kernel-argument segment per kernel
    const uint32_t h = lane >> 5, e = (lane >> 2) & 7u;
    HullEdge r;
    r.a = fp.hull[h][e][0]; r.b = fp.hull[h][e][1]; r.c = fp.hull[h][e][2]; r.valid = fp.hull[h][e][3] > 0.5f;
    return r;
}
__device__ __forceinline__ uint32_t classify_tile(const FrameParams& fp, const HullEdge& edge, uint32_t lane, float x0, float y0, float extent = 7.0f,
                                                   bool masked_out = false)
{
    const uint32_t k = lane & 3u;
    const float cx = (k & 1u) ? x0 + extent : x0, cy = (k & 2u) ? y0 + extent : y0;
    const float a = edge.a, b = edge.b, c = edge.c;
    const bool valid = edge.valid;
    const float v = __builtin_fmaf(a, cx, __builtin_fmaf(b, cy, c));
    const float margin = 1.5f;
    const unsigned long long mo = __ballot(valid && v < -margin);     // corner strictly outside edge
    const unsigned long long mi = __ballot(!valid || v > margin);     // corner strictly inside edge
    auto outside = [&](uint32_t bits) { return ((bits & (bits >> 1) & (bits >> 2) & (bits >> 3)) & 0x11111111u) != 0u; };
    const bool cube_ok = (fp.cull & CULL_CUBE_HULL) != 0u, obj_ok = (fp.cull & CULL_OBJ_HULL) != 0u;
    const bool out_cube = cube_ok && outside(static_cast<uint32_t>(mo));
    const bool in_cube = cube_ok && static_cast<uint32_t>(mi) == 0xffffffffu;
    const bool out_obj = masked_out || (fp.cull & CULL_NOTHING_DENSE) != 0u || (obj_ok && outside(static_cast<uint32_t>(mo >> 32)));
    if (out_cube) return TILE_FILL_MISS;
    if (out_obj) return in_cube ? TILE_FILL_EMPTY : TILE_HIT_TEST;
    return TILE_MARCH;
}
// tile_mask bit of the 8x8 tile (t8x, t8y) (wave-uniform)
__device__ __forceinline__ bool tile_mask_bit(const FrameParams& fp, uint32_t t8x, uint32_t t8y)
{
    const uint32_t bit = t8y * fp.mask_t8x + t8x;
    return ((fp.tile_mask[bit >> 5] >> (bit & 31u)) & 1u) != 0u;
}
// WAVES: waves per workgroup.  16 (four per SIMD, a budget of 128 VGPRs) for the common instantiation, which fits; the
// importance / continuous-rho instantiations need ~150 registers and run 12 waves (three per SIMD, 168 VGPRs) instead of
// spilling 64-100 bytes per lane to scratch (profiles/r02_kernel_resources.txt).
// CJ (with IR): the look-ahead's walks as jobs shared by the waves of the workgroup ("cone jobs" below): 1 the cone look-ahead (a
// record on 8 lanes, one per direction), 2 the straight look-ahead (a record on one lane)
// LB (development build only, north_star "LDS-staged voxel bricks"): the K speculative voxel fetches of an iteration go through a
// per-wave cache of 4x4x4 bricks in LDS, filled with one coalesced 64-byte read per brick (16 lanes x 4 bytes) -- the A/B of
// profiles/r03_lds_bricks_ab.txt.  Bricked layout, common instantiation only.
template <bool TABLE, bool COUNT, bool TRACE = false, int KSPEC = 1, bool IMP = true, bool BRICK = false, bool IR = false, int WAVES = PQ_WAVES, int CJ = 0, bool LB = false>
__global__ __launch_bounds__(WAVES * 64) void volym_raymarch_pq_kernel(
    const uint8_t* __restrict__ vol, const uint8_t* __restrict__ imp, const FrameTables* __restrict__ tables,
    const uint8_t* __restrict__ df4, const uint2* __restrict__ order, uint32_t n_items, uint16_t* __restrict__ cost,
    uint32_t* __restrict__ out_shard, uint32_t* __restrict__ out_raster, float4* __restrict__ out_f32,
    Counters* __restrict__ counters, uint4* __restrict__ trace, const FrameParams fp)
{
    constexpr int K = TABLE ? KSPEC : 2;     // speculation depth (continuous-rho modes: 2 -- each sample is 5-8 gathers)
    constexpr int PQ_QCAP = pq_qcap(K);
    static_assert(K <= 4, "the shading queue of K > 4 does not fit the LDS");
    static_assert(CJ == 0 || IR, "look-ahead jobs belong to the importance-rendering instantiation");
    unsigned long long trace_t0 = 0;
    uint32_t trace_iters = 0, trace_flushes = 0, trace_tiles = 0, trace_marched = 0, trace_lanes = 0, trace_accepted = 0, trace_dp_iters = 0;
    unsigned long long tm_leap = 0, tm_samp = 0, tm_flush = 0, tm_mark = 0;
#define PQ_TICK() (TRACE ? __builtin_amdgcn_s_memtime() : 0ull)
    if (TRACE) trace_t0 = __builtin_amdgcn_s_memrealtime();

    __shared__ float4 s_tf_tab[256];
    __shared__ float4 s_lut[TABLE ? 1 : 256];
    __shared__ float s_ic_alpha[256];
    __shared__ float s_rho[256];
    __shared__ __attribute__((aligned(16))) uint8_t s_df[VOLYM_DF_LDS_BYTES];
    constexpr int THREADS = WAVES * 64;
    __shared__ float4 s_q[WAVES][PQ_QCAP];            // {pos, w}: one 16-byte LDS access per record
    __shared__ uint32_t s_qm[WAVES][PQ_QCAP];
    __shared__ float s_qr[TABLE ? 1 : WAVES][TABLE ? 1 : PQ_QCAP];
    __shared__ uint32_t s_acc[WAVES][3][64];
    __shared__ float4 s_hh[WAVES][64];                  // per ray of the wave's tile: the Blinn-Phong half vector (a constant of the ray)
    __shared__ uint8_t s_mail[(IMP || IR) ? WAVES : 1][(IMP || IR) ? 256 : 1];   // look-ahead candidates of a wave (ahead_straight_wave); cone jobs: the verdicts
    uint8_t (*const s_cres)[(IMP || IR) ? 256 : 1] = s_mail;      // per wave, [k][lane]: 0 pending, 1 nothing important ahead, 2 important ahead (a frame uses one of the two look-aheads)
    // IR: the cone look-ahead's walks, shared by the waves of the workgroup (below, "cone jobs")
    constexpr uint32_t CJ_LPR = CJ == 2 ? 1u : 8u;                   // lanes per record
    constexpr uint32_t CJ_RPS = 64u / CJ_LPR;                        // records per serve
    constexpr uint32_t CJ_CAP = CJ == 2 ? 64u : 32u;      // (a window: a wave with more samples to ask about serves jobs until there is room)
    static_assert(!LB || (TABLE && BRICK && !IMP && !IR && !COUNT && K == 4), "LDS brick staging: bricked layout, common instantiation");
    constexpr uint32_t LB_SLOTS = 8u;                                // bricks per wave (8 x 64 B x 16 waves = 8 KB, taken from the staged work list)
    __shared__ uint32_t s_lb[LB ? WAVES : 1][LB ? LB_SLOTS * 16u : 1u];
    constexpr int ITEMS_LDS = LB ? PQ_ITEMS_LDS / 4 : CJ == 2 ? PQ_ITEMS_LDS / 2 : PQ_ITEMS_LDS;   // (the straight jobs' ring takes the LDS of half the staged list)
    __shared__ float4 s_cj[CJ ? 2 * CJ_CAP : 1];        // {start.xyz, step} {dir.xyz, owner wave | k << 4 | lane << 6}
    __shared__ uint32_t s_cj_flag[CJ ? CJ_CAP : 1];     // 0: free, 2: being written, 1: written and not yet taken
    __shared__ uint32_t s_cj_ctl[4];                    // head, tail, waves that may still submit
    __shared__ uint32_t s_next_ticket;
    __shared__ uint2 s_items[ITEMS_LDS];                // this workgroup's work list (entries b, b+G, ...): {item code, tile x | tile y << 16}

    // IMP = false also pins the opacity flag of the common case so that its tests fold away; every other combination
    // runs the IMP = true instantiation
    // IR (with IMP = false): the same specialised paths with importance rendering on (opacity on, no importance colouring)
    static_assert(!(IMP && IR), "IR specialises the IMP = false paths");
    const uint32_t flags = IMP ? fp.flags : IR ? ((fp.flags & ~F_IMP_COLORING) | F_OPACITY | F_IMP_RENDERING)
                                               : ((fp.flags & ~(F_IMP_COLORING | F_IMP_RENDERING | F_CONE)) | F_OPACITY);
    const bool linear = (flags & F_LINEAR) != 0u;
    const bool gauss = (flags & F_GAUSSIAN) != 0u;
    // IMP = false: instantiation for frames with both importance modes off (the reference's "Base" rows): the
    // look-ahead and importance-colouring code, its registers and its scalar state are compiled out
    const bool imp_coloring = IMP && (flags & F_IMP_COLORING) != 0u;
    const bool imp_rendering = (IMP || IR) && (flags & F_IMP_RENDERING) != 0u;
    const bool need_imp = imp_coloring || imp_rendering;

    // keep the tile-mask buffer that this view does not read zeroed: the next view's volym_tile_mask_kernel ORs its bits into it
    // (a few words per workgroup; whether it has happened never decides a pixel, only whether an empty tile is marched)
    if (fp.tile_mask_spare)
        for (uint32_t wrd = blockIdx.x * (WAVES * 64u) + threadIdx.x; wrd < fp.mask_words; wrd += gridDim.x * (WAVES * 64u)) fp.tile_mask_spare[wrd] = 0u;
    if (VOLYM_DEV_SWITCHES && (fp.dev & 16u)) return;                    // launch + dispatch only
    // A launch whose costs are captured also reports when it ran: behind the costs, the end time of every wave and the start
    // time of every workgroup (100 MHz counter).  The host evens out what the counted costs mispredict (raymarch.hip, trim_list).
    uint32_t* const wg_time = reinterpret_cast<uint32_t*>(cost + ((n_items + 1u) & ~1u));
    if (cost && threadIdx.x == 0u) wg_time[gridDim.x * WAVES + blockIdx.x] = static_cast<uint32_t>(__builtin_amdgcn_s_memrealtime());
    {
        const uint32_t i = threadIdx.x;
        if (i == 0u) s_next_ticket = 0u;
        if (CJ) {
            for (uint32_t k = i; k < CJ_CAP; k += THREADS) s_cj_flag[k] = 0u;
            if (i == 0u) { s_cj_ctl[0] = 0u; s_cj_ctl[1] = 0u; s_cj_ctl[2] = WAVES; }
        }
        for (uint32_t k = i; k < static_cast<uint32_t>(ITEMS_LDS); k += THREADS) {
            const size_t gi = blockIdx.x + static_cast<size_t>(gridDim.x) * k;
            if (gi < n_items) s_items[k] = order[gi];
        }
        if (i < 256u) {
            s_tf_tab[i] = tables->tf_tab[i];
            s_rho[i] = tables->rho[i];
            s_ic_alpha[i] = tables->ic_alpha[i];
            if (!TABLE) s_lut[i] = tables->lut_f[i];
        }
        {
            const uint32_t n16 = VOLYM_DF_IN_LDS(fp) ? (fp.mc_n * fp.mc_n * fp.mc_n / 2u + 15u) / 16u : 0u;      // a finer grid is read from global memory (L1 / L2)
            const uint4* src = reinterpret_cast<const uint4*>(df4);
            uint4* dst = reinterpret_cast<uint4*>(s_df);
            if (!(VOLYM_DEV_SWITCHES && (fp.dev & 8u)))
                for (uint32_t k = i; k < n16; k += THREADS) dst[k] = src[k];
        }
    }
    __syncthreads();
    if (VOLYM_DEV_SWITCHES && (fp.dev & 4u)) return;                     // ... + staging

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    GridT<BRICK> g;
    grid_init(g, vol, imp, fp.nx, fp.ny, fp.nz);

    float4* const q4 = s_q[wave];
    uint32_t* const qm = s_qm[wave];
    float* const qr = s_qr[TABLE ? 0 : wave];
    uint32_t* const acc_r = s_acc[wave][0];
    uint32_t* const acc_g = s_acc[wave][1];
    uint32_t* const acc_b = s_acc[wave][2];
    float4* const hh = s_hh[wave];
    uint8_t* const mail = s_mail[(IMP || IR) ? wave : 0];

    // LB: lane s < LB_SLOTS holds the id of the brick staged in place s of this wave's cache (the volume does not change during a
    // launch: the cache lives as long as the wave), lb_next the place the next miss looks at first
    uint32_t lb_tag = 0xffffffffu, lb_next = 0u;
    uint32_t n_vol = 0, n_imp = 0, n_steps = 0, n_dense = 0, n_hit = 0;
    const float base = fp.base_step, min_step = fp.min_step, thr = fp.thr;
    const float mcf = static_cast<float>(fp.mc_n), inv_mc = 1.0f / mcf;

    // ---- work distribution: this workgroup owns items b, b+G, ...; its waves draw them in order ----
    const uint32_t n_mine = n_items > blockIdx.x ? (n_items - blockIdx.x + gridDim.x - 1) / gridDim.x : 0u;
    auto grab = [&]() __attribute__((always_inline)) -> uint32_t {
        uint32_t ticket = 0;
        if (lane == 0) ticket = atomicAdd(&s_next_ticket, 1u);
        return __builtin_amdgcn_readfirstlane(ticket);
    };
    const bool culling = !COUNT && fp.cull != 0u;
    const HullEdge hull_edge = load_hull_edge(fp, lane);

    // ---- cone jobs (IR, use_cone_importance_check): the 8 x N probes of a sample's look-ahead (wgsl:94-139) are the bulk of such a
    // frame, and they sit in the few tiles that show unimportant matter in front of important matter -- one wave per tile walked
    // them while the other waves of its workgroup had long finished (r02: 298 us per frame).  So the walks are jobs in a ring in LDS:
    // the wave that needs them writes {start, step, direction, who asked} records, ANY wave of the workgroup takes 8 records at a
    // time (8 samples x 8 directions on its 64 lanes, as ahead_cone_wave does), and writes one result byte per sample.  The asking
    // wave serves jobs itself while it waits, and a wave that has run out of tiles serves until every wave has: no wave ever
    // waits for anything but a job that some wave -- if need be itself -- is free to run.  Same positions and f32 operations per
    // direction as ahead_cone; every wait is bounded.
    uint32_t cj_count = 0;          // samples this wave has asked about (wave-uniform): part of a tile's counted cost
    uint32_t la_rounds = 0;         // rounds of 64 straight look-ahead chains this wave has walked (wave-uniform): likewise
    auto cj_serve = [&]() __attribute__((always_inline)) -> bool {
        uint32_t h = 0, n = 0;
        if (lane == 0u) {
            for (;;) {
                h = *reinterpret_cast<volatile uint32_t*>(&s_cj_ctl[0]);
                const uint32_t tl = *reinterpret_cast<volatile uint32_t*>(&s_cj_ctl[1]);
                n = min(CJ_RPS, tl - h);
                if (n == 0u || atomicCAS(&s_cj_ctl[0], h, h + n) == h) break;
            }
        }
        h = __builtin_amdgcn_readfirstlane(h); n = __builtin_amdgcn_readfirstlane(n);
        if (n == 0u) return false;
        if (VOLYM_DEV_SWITCHES && (fp.dev & 512u) && lane == 0u) atomicAdd(&counters->n_imp, static_cast<unsigned long long>(n));      // debug: records taken
        const uint32_t c = lane / CJ_LPR;
        const uint32_t glead = lane & ~(CJ_LPR - 1u);                  // first lane of this record's group
        const bool job = c < n;
        const uint32_t idx = (h + c) & (CJ_CAP - 1u);
        // Take a record out of the place: 1 written -> 3 taken (exclusive: places are reserved a lap apart, two servers can be
        // looking at one place), copy it, free the place at once (-> 0).  A group of eight lanes that has its record does not hold
        // the place while its siblings wait for theirs: nobody holds one thing and waits for another, so no circle of waits.
        bool job_ok = false;
        float4 r0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), r1 = r0;
        {
            bool have = !job;
            uint32_t spins = 0;
            while (!have) {
                uint32_t st = 0;
                if (lane == glead) st = atomicCAS(&s_cj_flag[idx], 1u, 3u) == 1u ? 1u : 0u;
                if (CJ_LPR > 1u) st = __shfl(st, static_cast<int>(glead), 64);
                if (st != 0u) {
                    r0 = s_cj[2u * idx]; r1 = s_cj[2u * idx + 1u];
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                    if (lane == glead) *reinterpret_cast<volatile uint32_t*>(&s_cj_flag[idx]) = 0u;
                    have = true; job_ok = true;
                } else {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins >= (1u << 20)) {                 // cannot happen; never hang
                        have = true;
                        if (VOLYM_DEV_SWITCHES && (fp.dev & 512u) && lane == glead) atomicAdd(&counters->n_hit, 1ull);   // debug: a record that never came
                    }
                }
            }
        }
        const V3 p0 = v3(r0.x, r0.y, r0.z), d0 = v3(r1.x, r1.y, r1.z);
        const float step = r0.w;
        const uint32_t meta = __float_as_uint(r1.w);
        const int nprobe = static_cast<int>(fp.ahead_steps);
        bool hit = false;
        if constexpr (CJ != 2) {
            const float cone_xo = fp.cone_cos[lane & 7u] * 0.2f, cone_yo = fp.cone_sin[lane & 7u] * 0.2f;
            const V3 right = normalize_exact(cross(d0, v3(0.0f, 1.0f, 0.0f)));       // wgsl:99-113, as ahead_cone
            const V3 new_up = cross(d0, right);
            const V3 sd = normalize_exact((d0 + right * cone_xo) + new_up * cone_yo);
            V3 pos = p0;
            bool left = !job_ok;                                                       // left: this direction has left [0,1]^3 (wgsl:122-124)
            for (int i = 0; i < nprobe; i += VOLYM_PROBE_BATCH) {
                uint32_t ib[VOLYM_PROBE_BATCH];
                bool out[VOLYM_PROBE_BATCH];
#pragma unroll
                for (int j = 0; j < VOLYM_PROBE_BATCH; ++j) {
                    pos = pos + sd * step;
                    out[j] = outside01(pos);
                    ib[j] = imp[nearest_offset(g, pos)];                              // clamped offset: safe wherever pos is
                }
#pragma unroll
                for (int j = 0; j < VOLYM_PROBE_BATCH; ++j) {
                    if (!left && !hit && i + j < nprobe) {
                        if (out[j]) left = true;
                        else if (ib[j] >= 128u) hit = true;                           // i/255 >= 0.5  <=>  i >= 128
                    }
                }
                const unsigned long long hit_now = __ballot(hit);                     // one direction's hit decides the sample (wgsl:108-139 returns there)
                if (((hit_now >> (lane & 56u)) & 0xffull) != 0ull) left = true;
                if (__ballot(!left && !hit) == 0ull) break;
            }
        } else {
            // the straight look-ahead (wgsl:141-160): one chain per lane, the positions the shader's accumulated additions, the bytes
            // gathered VOLYM_PROBE_BATCH at a time and examined in order (as ahead_straight_wave walks a chain)
            const V3 ds = d0 * step;
            V3 pos = p0;
            bool live = job_ok;
            for (int i = 0; i < nprobe; i += VOLYM_PROBE_BATCH) {
                V3 pb[VOLYM_PROBE_BATCH];
                bool inside = false;
#pragma unroll
                for (int b = 0; b < VOLYM_PROBE_BATCH; ++b) {
                    pos = pos + ds;
                    pb[b] = pos;
                    inside = inside || probe_may_hit(fp, pos);
                }
                if (__ballot(live && inside) == 0ull) continue;             // no live chain where an important texel can be read
                uint32_t ib[VOLYM_PROBE_BATCH];
#pragma unroll
                for (int b = 0; b < VOLYM_PROBE_BATCH; ++b) ib[b] = imp[nearest_offset(g, pb[b])];   // clamped offset: safe wherever pos is
#pragma unroll
                for (int b = 0; b < VOLYM_PROBE_BATCH; ++b)
                    if (live && i + b < nprobe && ib[b] >= 128u) { hit = true; live = false; }   // i/255 >= 0.5  <=>  i >= 128
                if (__ballot(live) == 0ull) break;
            }
        }
        const unsigned long long hits = __ballot(hit);
        const bool verdict = CJ == 2 ? hit : ((hits >> (lane & 56u)) & 0xffull) != 0ull;
        if (job_ok && lane == glead)
            *reinterpret_cast<volatile uint8_t*>(&s_cres[meta & 15u][((meta >> 4) & 3u) * 64u + ((meta >> 6) & 63u)]) = verdict ? 2 : 1;
        return true;
    };
    // the look-aheads of up to NK samples per lane (sample k at o + d * tsk[k]): submit, serve while waiting, read the verdicts.
    // One loop, one place that serves: a turn either submits the next k, or finds every verdict in, or walks somebody's job.
    auto cj_lookahead = [&](const auto& need_in, const auto& tsk, V3 org, V3 dir, float t_exit, auto& found) __attribute__((always_inline)) {
        constexpr int NK = static_cast<int>(sizeof(tsk) / sizeof(tsk[0]));
        uint32_t need_bits = 0;
        bool any = false;
#pragma unroll
        for (int k = 0; k < NK; ++k) { found[k] = false; any = any || need_in[k]; }
        if (__ballot(any) == 0ull) return;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const bool nd = need_in[k] && !ahead_cannot_hit(fp, org + dir * tsk[k], dir, t_exit, CJ == 1);   // cannot reach an important voxel: false, unwalked
            need_bits |= nd ? 1u << k : 0u;
            if (nd) *reinterpret_cast<volatile uint8_t*>(&s_cres[wave][k * 64 + lane]) = 0;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        uint32_t k = 0, todo_bits = need_bits;                                    // todo: not yet in the ring
        for (uint32_t turns = 0; turns < (1u << 16); ++turns) {
            if (k < static_cast<uint32_t>(NK)) {
                const bool mine = ((todo_bits >> k) & 1u) != 0u;
                const unsigned long long m = __ballot(mine);
                if (m == 0ull) { k++; continue; }
                const uint32_t n = static_cast<uint32_t>(__popcll(m));
                uint32_t got = 0, tl = 0;
                if (lane == 0u) {                                                 // as many of the n records as there is room for (serving makes room)
                    const uint32_t hd = *reinterpret_cast<volatile uint32_t*>(&s_cj_ctl[0]);
                    tl = *reinterpret_cast<volatile uint32_t*>(&s_cj_ctl[1]);
                    const uint32_t room = CJ_CAP - min(CJ_CAP, tl - hd);
                    got = min(n, room);
                    if (got != 0u && atomicCAS(&s_cj_ctl[1], tl, tl + got) != tl) got = 0;
                }
                got = __builtin_amdgcn_readfirstlane(got);
                cj_count += got;
                const uint32_t first_pos = __builtin_amdgcn_readfirstlane(tl);       // (read here, where lane 0 is active)
                if (got != 0u) {
                    const uint32_t rank = lane_rank_in_mask(m);
                    if (mine && rank < got) {
                        float tk = tsk[0];
#pragma unroll
                        for (int q = 1; q < NK; ++q) tk = k == static_cast<uint32_t>(q) ? tsk[q] : tk;
                        const V3 start = org + dir * tk;                          // the sample position, as its owner computes it (wgsl:251)
                        const uint32_t idx = (first_pos + rank) & (CJ_CAP - 1u);
                        const float step = (t_exit - length_exact(start)) / static_cast<float>(static_cast<int>(fp.ahead_steps));   // wgsl:111
                        // Take the place (0 free -> 2 being written -> 1 written): a place can be reserved a lap apart by two waves while its
                        // last taker still reads it, and only one of them may write next.  Every lane writes in the very turn it gets its
                        // place -- a lane that waited for its siblings' places would close a circle of waits (a server holds back a place
                        // until all eight of its records are there, one of which is this wave's).
                        bool written = false;
                        uint32_t spins = 0;
                        while (!written) {
                            if (atomicCAS(&s_cj_flag[idx], 0u, 2u) == 0u) {
                                s_cj[2u * idx] = make_float4(start.x, start.y, start.z, step);
                                s_cj[2u * idx + 1u] = make_float4(dir.x, dir.y, dir.z, __uint_as_float(wave | (k << 4) | (lane << 6)));
                                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                                *reinterpret_cast<volatile uint32_t*>(&s_cj_flag[idx]) = 1u;
                                written = true;
                            } else {
                                __builtin_amdgcn_s_sleep(1);
                                if (++spins >= (1u << 20)) {                      // cannot happen (see above); never hang: the verdict is "nothing ahead"
                                    *reinterpret_cast<volatile uint8_t*>(&s_cres[wave][k * 64 + lane]) = 1;
                                    written = true;
                                    if (VOLYM_DEV_SWITCHES && (fp.dev & 512u)) atomicAdd(&counters->n_dense, 1ull);
                                }
                            }
                        }
                        todo_bits &= ~(1u << k);
                        if (VOLYM_DEV_SWITCHES && (fp.dev & 512u)) atomicAdd(&counters->n_vol, 1ull);   // debug: records written
                    }
                    if (got == n) { k++; continue; }                              // (else: the rest of this k after a job or two)
                }
            } else {
                bool pending = false;
#pragma unroll
                for (int q = 0; q < NK; ++q) pending = pending || (((need_bits >> q) & 1u) != 0u && *reinterpret_cast<volatile uint8_t*>(&s_cres[wave][q * 64 + lane]) == 0);
                if (__ballot(pending) == 0ull) break;
            }
            if (!cj_serve()) __builtin_amdgcn_s_sleep(1);
            if (VOLYM_DEV_SWITCHES && (fp.dev & 512u) && turns + 1u == (1u << 16) && lane == 0u) atomicAdd(&counters->n_steps, 1ull);       // debug: a look-ahead that gave up
        }
#pragma unroll
        for (int q = 0; q < NK; ++q) found[q] = ((need_bits >> q) & 1u) != 0u && *reinterpret_cast<volatile uint8_t*>(&s_cres[wave][q * 64 + lane]) == 2;
    };
    // this lane's cone direction (lane & 7) for the wave-wide cone look-ahead, computed where it is used (two gathers
    // from the kernel-argument segment and two multiplies per call: nothing kept live across the march)
    for (uint32_t ticket = grab(); ticket < n_mine; ticket = grab()) {
      {
        // item = local_tile*4 + sub (an 8x8 wave tile, one lane per ray), or, for tiles the cost feedback
        // found expensive, bit 31 | (that id << 2) | quarter: a 4x4 quarter tile marched DEPTH-PARALLEL,
        // four lanes per ray, lane k of a quad taking the k-th speculative sample (see "dp" below)
        const size_t list_pos = blockIdx.x + static_cast<size_t>(gridDim.x) * ticket;     // this entry's place in the work list
        const uint2 entry = ticket < static_cast<uint32_t>(ITEMS_LDS) ? s_items[ticket] : order[list_pos];
        const uint32_t raw_p = __builtin_amdgcn_readfirstlane(entry.x);
        if (raw_p == PQ_NO_ITEM) continue;
        // the 16x16 tile's position, worked out by the host when it dealt the list (no integer division here)
        const uint32_t entry_xy = __builtin_amdgcn_readfirstlane(entry.y);
        const uint32_t tx = entry_xy & 0xffffu, ty = entry_xy >> 16;
        // counted cost of this list entry, reported by list position (one writer per entry: no atomics, nothing to reset); the
        // host's feedback thread maps positions back to tiles (raymarch.hip, "cost feedback")
        uint32_t entry_cost = 0;
        // bits 28-29: issue priority the host derived from the measured cost.  The frame ends with its longest chains of
        // dependent samples; a wave that carries one gets the SIMD's issue slots first, the cheap items fill the gaps.
        switch ((raw_p >> 28) & 3u) {
            case 0: __builtin_amdgcn_s_setprio(0); break;
            case 1: __builtin_amdgcn_s_setprio(1); break;
            case 2: __builtin_amdgcn_s_setprio(2); break;
            default: __builtin_amdgcn_s_setprio(3); break;
        }
        const uint32_t raw0 = raw_p & ~0x30000000u;
        // bit 30 (bit 31 clear): a whole 16x16 tile that was constant in the frame the costs were measured on.  One
        // classification of the 16x16 rectangle and, if it still says "constant", 16-byte stores; otherwise its four
        // sub-tiles are processed here one after the other.
        const bool is_super = (raw0 >> 30) == 1u;
        uint32_t n_sub = 1;
        if (is_super) {
            const uint32_t lt = raw0 & 0x3fffffffu;
            const uint32_t tx16 = tx, ty16 = ty;
            bool masked16 = false;
            if (culling && (fp.cull & CULL_TILE_MASK))
                masked16 = !(tile_mask_bit(fp, tx16 * 2u, ty16 * 2u) || tile_mask_bit(fp, tx16 * 2u + 1u, ty16 * 2u) ||
                             tile_mask_bit(fp, tx16 * 2u, ty16 * 2u + 1u) || tile_mask_bit(fp, tx16 * 2u + 1u, ty16 * 2u + 1u));
            const uint32_t cls = culling ? classify_tile(fp, hull_edge, lane, static_cast<float>(tx16 * 16u), static_cast<float>(ty16 * 16u), 15.0f, masked16) : TILE_MARCH;
            if (cls >= TILE_FILL_EMPTY) {
                const uint32_t packed = cls == TILE_FILL_MISS ? 0xff000000u : 0u;
                if (flags & F_RASTER) {
                    const uint32_t gx4 = tx16 * 16u + (lane & 3u) * 4u, gy4 = ty16 * 16u + (lane >> 2);
                    if (gy4 < fp.H && (fp.W & 3u) == 0u && (reinterpret_cast<uintptr_t>(out_raster) & 15u) == 0u && gx4 < fp.W && !(flags & F_WRITE_F32)) {
                        // rows are 16-byte aligned: one store per lane
                        *reinterpret_cast<uint4*>(out_raster + static_cast<size_t>(gy4) * fp.W + gx4) = make_uint4(packed, packed, packed, packed);
                    } else if (gy4 < fp.H) {
                        for (uint32_t i = 0; i < 4u; ++i)
                            if (gx4 + i < fp.W) {
                                out_raster[static_cast<size_t>(gy4) * fp.W + gx4 + i] = packed;
                                if (flags & F_WRITE_F32) out_f32[static_cast<size_t>(gy4) * fp.W + gx4 + i] = make_float4(0.0f, 0.0f, 0.0f, cls == TILE_FILL_MISS ? 1.0f : 0.0f);
                            }
                    }
                } else {
                    // shard layout: sub-tile major, 64 pixels each; pixels outside the frame are zero
                    for (uint32_t i = 0; i < 4u; ++i) {
                        const uint32_t sl = lane * 4u + i, sub4 = sl >> 6, in = sl & 63u;
                        const uint32_t gxx = tx16 * 16u + ((sub4 & 1u) << 3) + (in & 7u), gyy = ty16 * 16u + ((sub4 >> 1) << 3) + (in >> 3);
                        out_shard[static_cast<size_t>(lt) * 256u + sl] = (gxx < fp.W && gyy < fp.H) ? packed : 0u;
                    }
                }
                if (cost && lane == 0) cost[list_pos] = 0;
                continue;
            }
            n_sub = 4;
        }
      for (uint32_t sub_iter = 0; sub_iter < n_sub; ++sub_iter) {
        const uint32_t raw = is_super ? ((raw0 & 0x3fffffffu) * 4u + sub_iter) : raw0;
        const bool is_quarter = (raw >> 31) != 0u;
        const bool dp = is_quarter && !COUNT;
        const uint32_t item = is_quarter ? ((raw & 0x7fffffffu) >> 2) : raw;
        const uint32_t quarter = raw & 3u;
        if (is_quarter && !dp && quarter != 0u) continue;   // instrumented launch: quarter 0 stands for the whole tile
        const uint32_t local_tile = item >> 2, sub = item & 3u;
        const uint32_t px_in_sub = dp ? (((quarter & 1u) << 2) + ((lane >> 2) & 3u)) : (lane & 7u);
        const uint32_t py_in_sub = dp ? (((quarter >> 1) << 2) + (lane >> 4)) : (lane >> 3);
        const uint32_t sub_lane = py_in_sub * 8u + px_in_sub;       // position inside the 8x8 sub-tile
        const uint32_t own = dp ? (lane & ~3u) : lane;              // lane whose accumulators this ray uses
        const uint32_t gx = tx * 16u + ((sub & 1u) << 3) + px_in_sub;
        const uint32_t gy = ty * 16u + ((sub >> 1) << 3) + py_in_sub;
        const bool in_frame = gx < fp.W && gy < fp.H;   // wgsl:217-219

        if (TRACE) { trace_tiles++; tm_mark = PQ_TICK(); }
        uint32_t tile_iters = 0, tile_flushes = 0, tile_trips = 0;   // deterministic cost of this tile, fed back to the scheduler
        const uint32_t cj_at_start = cj_count, la_at_start = la_rounds;
        uint32_t trace_ray_iters = 0;                                // TRACE: iterations this lane's ray was active in
        uint32_t tclass = TILE_MARCH;
        if (culling && !dp) {
            const bool masked8 = (fp.cull & CULL_TILE_MASK) != 0u && !tile_mask_bit(fp, tx * 2u + (sub & 1u), ty * 2u + (sub >> 1));
            tclass = classify_tile(fp, hull_edge, lane, static_cast<float>(tx * 16u + ((sub & 1u) << 3)), static_cast<float>(ty * 16u + ((sub >> 1) << 3)), 7.0f, masked8);
            if (tclass >= TILE_FILL_EMPTY) {            // no ray of this tile can differ from the constant
                const uint32_t packed = tclass == TILE_FILL_MISS ? 0xff000000u : 0u;     // (0,0,0,1) wgsl:239 / (0,0,0,0) wgsl:328
                if (flags & F_RASTER) {
                    if (in_frame) {
                        const size_t o = static_cast<size_t>(gy) * fp.W + gx;
                        out_raster[o] = packed;
                        if (flags & F_WRITE_F32) out_f32[o] = make_float4(0.0f, 0.0f, 0.0f, tclass == TILE_FILL_MISS ? 1.0f : 0.0f);
                    }
                } else {
                    out_shard[static_cast<size_t>(local_tile) * 256u + sub * 64u + sub_lane] = in_frame ? packed : 0u;
                }
                continue;                                    // a constant tile costs (next to) nothing
            }
        }
        if (TRACE) trace_marched++;

        acc_r[lane] = 0u; acc_g[lane] = 0u; acc_b[lane] = 0u;
        uint32_t q_head = 0, q_count = 0;               // wave-uniform ring state

        Ray ray;
        ray.o = v3(0.0f, 0.0f, 0.0f); ray.d = v3(0.0f, 0.0f, 0.0f); ray.t_entry = 0.0f; ray.t_exit = 0.0f;
        ray.hit = false;
        if (in_frame) ray = make_ray<TABLE && !IMP && !IR>(fp, gx, gy);     // shared reciprocals where the registers allow (raymarch_device.h)
        {
            const V3 hvec = ray_half_vector(ray.d);
            hh[lane] = make_float4(hvec.x, hvec.y, hvec.z, 0.0f);     // (a depth-parallel quad: four equal entries, owner = the first)
        }
        bool active = in_frame && ray.hit;
        if (COUNT && active) n_hit++;
        float t = active ? ray.t_entry : 0.0f, cur = base, acc_a = active ? 0.0f : 1.0f;   // miss: (0,0,0,1) wgsl:239
        bool last_dense = false;
        if (culling && tclass == TILE_HIT_TEST) active = false;     // hit rays of this tile see nothing dense: (0,0,0,0)
        float t_end = ray.t_exit;                        // samples at t >= t_end cannot be dense
        if (culling && (fp.cull & CULL_AABB) && active) {
            // slab test against the AABB of the occupied macro cells (conservative arithmetic)
            // v_rcp_f32 (1 ulp) instead of a division: the slab distances carry a 2e-5 margin
            const float rx = __builtin_amdgcn_rcpf(ray.d.x), ry = __builtin_amdgcn_rcpf(ray.d.y), rz = __builtin_amdgcn_rcpf(ray.d.z);
            const float ax0 = (fp.aabb_lo[0] - ray.o.x) * rx, ax1 = (fp.aabb_hi[0] - ray.o.x) * rx;
            const float ay0 = (fp.aabb_lo[1] - ray.o.y) * ry, ay1 = (fp.aabb_hi[1] - ray.o.y) * ry;
            const float az0 = (fp.aabb_lo[2] - ray.o.z) * rz, az1 = (fp.aabb_hi[2] - ray.o.z) * rz;
            float tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(ax0, ax1), __builtin_fminf(ay0, ay1)), __builtin_fminf(az0, az1));
            float tf = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(ax0, ax1), __builtin_fmaxf(ay0, ay1)), __builtin_fmaxf(az0, az1));
            tn = tn - 2.0e-5f * __builtin_fabsf(tn) - 1.0e-6f;
            tf = tf + 2.0e-5f * __builtin_fabsf(tf) + 1.0e-6f;
            // a zero direction component makes its slab pair NaN/inf: fmin/fmax drop NaN, so test the origin there
            const bool outside_static = (ray.d.x == 0.0f && (ray.o.x < fp.aabb_lo[0] || ray.o.x > fp.aabb_hi[0])) ||
                                        (ray.d.y == 0.0f && (ray.o.y < fp.aabb_lo[1] || ray.o.y > fp.aabb_hi[1])) ||
                                        (ray.d.z == 0.0f && (ray.o.z < fp.aabb_lo[2] || ray.o.z > fp.aabb_hi[2]));
            if (outside_static || !(tn <= tf)) {
                active = false;                          // never inside the AABB: nothing dense on this ray
            } else {
                t_end = __builtin_fminf(t_end, tf);
                // replay of the empty steps in front of the AABB (wgsl:263-274), cur == base throughout
                const float t_first = __builtin_fminf(tn, t_end);
                replay_saturated(t, t_first, base);
            }
        }
        const float idx_ = __builtin_amdgcn_rcpf(ray.d.x), idy_ = __builtin_amdgcn_rcpf(ray.d.y), idz_ = __builtin_amdgcn_rcpf(ray.d.z);   // leap exits carry a margin too
        const float nox = -ray.o.x * idx_, noy = -ray.o.y * idy_, noz = -ray.o.z * idz_;

        // ---- shade up to 64 queued samples, one per lane (COLOUR arithmetic) ----
        auto flush = [&](uint32_t n) __attribute__((always_inline)) {
            unsigned long long fl0 = 0;
            tile_flushes++;
            if (TRACE) { trace_flushes++; fl0 = PQ_TICK(); }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lane < n && !(VOLYM_DEV_SWITCHES && (fp.dev & 1u))) {
                const uint32_t e0 = q_head + lane, e = min(e0, e0 - PQ_QCAP);      // ring wrap (unsigned)
                const float4 rec = q4[e];
                const V3 pos = v3(rec.x, rec.y, rec.z);
                const float w = rec.w;
                const uint32_t meta = qm[e];
                const uint32_t owner = meta & 63u, b = (meta >> 8) & 255u, ib = (meta >> 16) & 255u;
                V3 color;
                if (imp_coloring) {                                   // wgsl:83-92
                    const float im = s_rho[ib];
                    color = v3(__builtin_fminf(im * 1.5f, 1.0f), (1.0f - im) * 1.2f, 0.2f);
                } else if (TABLE) {
                    const float4 ca = s_tf_tab[b];
                    color = v3(ca.x, ca.y, ca.z);
                } else {
                    const float4 ca = sample_tf(s_lut, fp.tf_n, qr[e]); // wgsl:297-303
                    color = v3(ca.x, ca.y, ca.z);
                }
                // gradient taps (wgsl:181-188); the common 1/(2*0.01) factor cancels in normalize()
                V3 grad;
                const float o = 0.01f;
                if (TABLE) {
                    const int ix = texel_nearest(pos.x, g.fnx, g.hix), iy = texel_nearest(pos.y, g.fny, g.hiy),
                              iz = texel_nearest(pos.z, g.fnz, g.hiz);
                    // the +o / -o taps of an axis as one packed add and one packed multiply (same IEEE operations per component)
                    const f32x2 pm = {o, -o};
                    const f32x2 uxs = (pos.x + pm) * g.fnx, uys = (pos.y + pm) * g.fny, uzs = (pos.z + pm) * g.fnz;
                    const int ixp = clamp_texel(floor_to_int(uxs.x), static_cast<int>(g.hix)), ixm = clamp_texel(floor_to_int(uxs.y), static_cast<int>(g.hix));
                    const int iyp = clamp_texel(floor_to_int(uys.x), static_cast<int>(g.hiy)), iym = clamp_texel(floor_to_int(uys.y), static_cast<int>(g.hiy));
                    const int izp = clamp_texel(floor_to_int(uzs.x), static_cast<int>(g.hiz)), izm = clamp_texel(floor_to_int(uzs.y), static_cast<int>(g.hiz));
                    const int bxp = vol[voxel_offset(g, ixp, iy, iz)], bxm = vol[voxel_offset(g, ixm, iy, iz)];
                    const int byp = vol[voxel_offset(g, ix, iyp, iz)], bym = vol[voxel_offset(g, ix, iym, iz)];
                    const int bzp = vol[voxel_offset(g, ix, iy, izp)], bzm = vol[voxel_offset(g, ix, iy, izm)];
                    // b/255 differences up to the common factor 1/255
                    grad = v3(static_cast<float>(bxp - bxm), static_cast<float>(byp - bym), static_cast<float>(bzp - bzm));
                } else {
                    grad = v3(sample_density(g, s_rho, linear, v3(pos.x + o, pos.y, pos.z)) -
                                  sample_density(g, s_rho, linear, v3(pos.x - o, pos.y, pos.z)),
                              sample_density(g, s_rho, linear, v3(pos.x, pos.y + o, pos.z)) -
                                  sample_density(g, s_rho, linear, v3(pos.x, pos.y - o, pos.z)),
                              sample_density(g, s_rho, linear, v3(pos.x, pos.y, pos.z + o)) -
                                  sample_density(g, s_rho, linear, v3(pos.x, pos.y, pos.z - o)));
                }
                const float4 hv = hh[owner];
                const V3 shaded = blinn_phong_h(color, grad, v3(hv.x, hv.y, hv.z));   // wgsl:190-211, half vector per ray
                // w * shaded in 4.28 fixed point; integer adds commute, so arrival order is irrelevant
                atomicAdd(&acc_r[owner], static_cast<uint32_t>(__builtin_fmaf(shaded.x * w, PQ_FIX_SCALE, 0.5f)));
                atomicAdd(&acc_g[owner], static_cast<uint32_t>(__builtin_fmaf(shaded.y * w, PQ_FIX_SCALE, 0.5f)));
                atomicAdd(&acc_b[owner], static_cast<uint32_t>(__builtin_fmaf(shaded.z * w, PQ_FIX_SCALE, 0.5f)));
            }
            q_head += n;
            if (q_head >= PQ_QCAP) q_head -= PQ_QCAP;
            q_count -= n;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (TRACE) tm_flush += PQ_TICK() - fl0;
        };

        // ---- push the samples flagged by `emit` (any subset of lanes), shade when 64 are waiting ----
        auto append = [&](bool emit, V3 p, float w, uint32_t meta, float rho) __attribute__((always_inline)) {
            if (VOLYM_DEV_SWITCHES && (fp.dev & 128u)) return;              // timing experiment: nothing is queued or shaded
            const unsigned long long mask = __ballot(emit);
            if (mask == 0ull) return;
            if (emit) {
                const uint32_t e0 = q_head + q_count + lane_rank_in_mask(mask), e = min(e0, e0 - PQ_QCAP);
                q4[e] = make_float4(p.x, p.y, p.z, w); qm[e] = meta;
                if (!TABLE) qr[e] = rho;
            }
            q_count += static_cast<uint32_t>(__popcll(mask));
            if (!TABLE && q_count >= 64u) flush(64u);      // continuous-rho modes shade as soon as a batch is full (see drain)
        };
        // Shading happens at one point of an iteration: right after the byte gathers of the next samples were issued,
        // so that the latency of those gathers and the shading of the previous samples overlap (the long chains of
        // dependent samples are what a frame ends on).
        // (a two-batch form -- the twelve gradient gathers of 128 queued samples in flight together -- was built and measured for the
        // tiles that queue several batches per iteration: 32.7 us against 32.3 without it; removed)
        auto drain = [&]() __attribute__((always_inline)) { while (q_count >= 64u) flush(64u); };

        if (VOLYM_DEV_SWITCHES && (fp.dev & 256u)) active = false;          // timing experiment: set-up and store only
        if (TRACE) tm_mark = PQ_TICK();
        // ---- at most ONE leap per lane and iteration through provably empty macro cells (both march paths) ----
        // Looking for a leap costs ~100 instructions and an LDS round trip per iteration.  A ray that has just sampled
        // its way through a whole batch of non-dense samples (or has not sampled yet) is in open space and may find one; a
        // ray whose last batch met the object is within a cell of it, where the distance field says "sample".
        bool leap_ok = true;
        auto leap_phase = [&]() __attribute__((always_inline)) {
            // at most ONE leap per lane and iteration through provably empty macro cells (see
            // raymarch_kernels.h VARIANT 1).  Cells with distance value < PQ_MIN_LEAP_D are simply sampled:
            // a one-cell leap replays ~3 steps, which the K-wide speculation below covers in the same
            // iteration without the ~100 instructions and the LDS round trip of a leap. ----
            active = active && t < t_end && acc_a < 0.95f;            // wgsl:250 (t_end: nothing dense beyond)
            const bool want_leap = active && !last_dense && leap_ok;
            if (__ballot(want_leap) != 0ull) {                          // inside a dense run nobody can leap
                // the cell lookup runs on every lane (clamped index, no branch); only D decides
                const V3 pos = ray.o + ray.d * t;
                const float cxf = __builtin_floorf(pos.x * mcf), cyf = __builtin_floorf(pos.y * mcf), czf = __builtin_floorf(pos.z * mcf);
                const int cx = static_cast<int>(cxf), cy = static_cast<int>(cyf), cz = static_cast<int>(czf);
                const bool in_range = static_cast<uint32_t>(cx | cy | cz) < fp.mc_n;
                const uint32_t ci = in_range ? mad_u24(mad_u24(static_cast<uint32_t>(cz), fp.mc_n, static_cast<uint32_t>(cy)), fp.mc_n, static_cast<uint32_t>(cx)) : 0u;
                uint32_t D = (static_cast<uint32_t>(VOLYM_DF_IN_LDS(fp) ? s_df[ci >> 1] : df4[ci >> 1]) >> ((ci & 1u) * 4u)) & 15u;
                if (!(want_leap && in_range)) D = 0u;
                if (D >= PQ_MIN_LEAP_D) {
                    // smoothing taps sit up to 2*0.005 along the ray from the sample (wgsl:53-60): keep them inside too
                    const float eps = gauss ? 4.0e-5f + 0.0101f : 4.0e-5f;
                    const float a = static_cast<float>(D - 1u) * inv_mc - eps;
                    const float lx = __builtin_fmaf(cxf, inv_mc, -a), hx = __builtin_fmaf(cxf, inv_mc, a + inv_mc);
                    const float ly = __builtin_fmaf(cyf, inv_mc, -a), hy = __builtin_fmaf(cyf, inv_mc, a + inv_mc);
                    const float lz = __builtin_fmaf(czf, inv_mc, -a), hz = __builtin_fmaf(czf, inv_mc, a + inv_mc);
                    const float ex = __builtin_fmaxf(__builtin_fmaf(lx, idx_, nox), __builtin_fmaf(hx, idx_, nox));
                    const float ey = __builtin_fmaxf(__builtin_fmaf(ly, idy_, noy), __builtin_fmaf(hy, idy_, noy));
                    const float ez = __builtin_fmaxf(__builtin_fmaf(lz, idz_, noz), __builtin_fmaf(hz, idz_, noz));
                    float te = __builtin_fminf(__builtin_fminf(ex, ey), ez);
                    te = te - 2.0e-5f * __builtin_fabsf(te);
                    // D >= 2: the cell of pos lies at least one whole cell inside the box, no `inside` test needed
                    const float t_stop = __builtin_fminf(te, t_end);
                    if (!COUNT) {
                        // replay of empty steps, wgsl:263-274: at most four until the step size is back at `base`, the rest in closed form
                        while (cur < base && t < t_stop) {
                            cur = __builtin_fminf(base, cur * 1.5f);
                            t += cur;
                        }
                        tile_trips++;
                        replay_saturated(t, t_stop, base);
                    }
                    while (COUNT && t < t_stop) {                 // the instrumented launch counts every replayed step
                        if (COUNT) {
                            n_steps++; n_imp++;
                            if (gauss) {                          // the shader fetches the taps that lie inside [0,1]^3 (wgsl:58-66)
                                const V3 q = ray.o + ray.d * t;
                                for (int i = -2; i <= 2; ++i) if (!outside01(q + ray.d * (static_cast<float>(i) * 0.005f))) n_vol++;
                            } else {
                                n_vol++;
                            }
                        }
                        cur = __builtin_fminf(base, cur * 1.5f);
                        t += cur;
                    }
                    if (!(t < t_end)) active = false;             // wgsl:250
                }
            }

        };

        if (dp) {
            // ================= depth-parallel march of a 4x4 quarter tile =================
            // Lanes 4r..4r+3 hold the same ray r; lane k of the quad takes the k-th of the four speculative
            // samples.  All state (t, cur, alpha, last_dense) is replicated in the quad and updated by the
            // same arithmetic in every lane, so the accepted samples are those of the sequential march.
            const uint32_t kq = lane & 3u, qsh = lane & 60u;
            if constexpr (!IMP) {
                // Pinned flags (opacity on, no importance mode): PQ_DP_DEPTH samples per lane, i.e. 4 * PQ_DP_DEPTH
                // speculative samples per ray and iteration, sample s on lane s & 3.  These items are the frame's longest
                // chains of dependent samples (grazing rays that stay within a cell of a surface, or inside it): what
                // matters for them is samples per iteration, and a long run of one class is exactly what they consist of.
                constexpr int J = PQ_DP_DEPTH;
                bool try_leap = true;
                active = active && t < t_end;
                while (__ballot(active) != 0ull) {
                    tile_iters++;
                    if (TRACE) { trace_iters++; trace_dp_iters++; trace_lanes += static_cast<uint32_t>(__popcll(__ballot(active))); tm_mark = PQ_TICK(); }
                    // a leap is worth looking for only where the march just ran through a whole batch of non-dense samples
                    // (or has not sampled yet); `active` is up to date either way
                    if (try_leap && !(VOLYM_DEV_SWITCHES && (fp.dev & 2u))) leap_phase();
                    if (TRACE) { const unsigned long long now = PQ_TICK(); tm_leap += now - tm_mark; tm_mark = now; }
                    // predicted march under "class stays last_dense" (wgsl:263-274): position of sample s and the step
                    // size in force before it; this lane fetches samples kq, kq + 4, ...
                    constexpr int N = 4 * J;
                    float ts[N], cb[N], my_t[J];
                    {
                        float tt = t, cc = cur;
#pragma unroll
                        for (int sidx = 0; sidx < N; ++sidx) {
                            ts[sidx] = tt; cb[sidx] = cc;
                            if ((sidx & 3) == 0) my_t[sidx >> 2] = tt;
                            else my_t[sidx >> 2] = kq == static_cast<uint32_t>(sidx & 3) ? tt : my_t[sidx >> 2];
                            cc = last_dense ? min_step : __builtin_fminf(base, cc * 1.5f);
                            tt += cc;
                        }
                    }
                    V3 my_pos[J];
                    uint32_t my_b[J], my_ib[J];
#pragma unroll
                    for (int j = 0; j < J; ++j) {
                        my_ib[j] = 255u;
                        my_pos[j] = ray.o + ray.d * my_t[j];                 // wgsl:251
                        const uint32_t off_j = nearest_offset(g, my_pos[j]);  // clamped offset: no guard needed
                        my_b[j] = vol[off_j];
                        if (IR) my_ib[j] = imp[off_j];
                    }
                    drain();
                    uint32_t cm = 0;                                         // class bits of the N samples (same in the quad)
                    float a4[J][4];
#pragma unroll
                    for (int j = 0; j < J; ++j) {
                        cm |= (static_cast<uint32_t>(__ballot(my_b[j] >= fp.thr_byte) >> qsh) & 15u) << (4 * j);   // <=> b/255 >= thr
                        const int a_bits = __float_as_int(s_tf_tab[my_b[j]].w);
                        a4[j][0] = __int_as_float(__builtin_amdgcn_mov_dpp(a_bits, 0x00, 0xf, 0xf, true));
                        a4[j][1] = __int_as_float(__builtin_amdgcn_mov_dpp(a_bits, 0x55, 0xf, 0xf, true));
                        a4[j][2] = __int_as_float(__builtin_amdgcn_mov_dpp(a_bits, 0xaa, 0xf, 0xf, true));
                        a4[j][3] = __int_as_float(__builtin_amdgcn_mov_dpp(a_bits, 0xff, 0xf, 0xf, true));
                    }
                    // importance rendering (wgsl:283-295): every lane looks ahead for its own samples; the verdicts travel like
                    // the classes.  sm: samples that are dense but suppressed -- they advance the march and emit nothing.
                    uint32_t sm = 0;
                    if (IR) {
                        bool need[J], ahead[J];
#pragma unroll
                        for (int j = 0; j < J; ++j) { need[j] = active && my_b[j] >= fp.thr_byte && my_ib[j] < 255u; ahead[j] = false; }
                        if constexpr (CJ != 0) {
                            cj_lookahead(need, my_t, ray.o, ray.d, ray.t_exit, ahead);
                        } else if (flags & F_CONE) {
#pragma unroll
                            for (int j = 0; j < J; ++j)
                                ahead[j] = ahead_cone_wave(g, fp, need[j], my_pos[j], ray.d, ray.t_exit, lane, fp.cone_cos[lane & 7u] * 0.2f, fp.cone_sin[lane & 7u] * 0.2f);
                        } else {
                            // (the straight look-ahead through the job ring was measured: 104 us against 66 -- one chain is one lane's work
                            // here, eight lanes' there, and the ring is a window of 32 samples)
                            bool any_need = false;
#pragma unroll
                            for (int j = 0; j < J; ++j) any_need = any_need || need[j];
                            if (__ballot(any_need) != 0ull) ahead_straight_wave<J>(g, fp, need, my_t, ray.o, ray.d, ray.t_exit, lane, mail, ahead, &la_rounds);
                        }
#pragma unroll
                        for (int j = 0; j < J; ++j) sm |= (static_cast<uint32_t>(__ballot(need[j] && ahead[j]) >> qsh) & 15u) << (4 * j);
                    }
                    // How many samples does the sequential march accept?  It stops (wgsl:250) at the first sample with
                    // t >= t_end or alpha >= 0.95, and the predicted positions hold up to and including the first sample
                    // whose class differs from the prediction.  t and alpha grow monotonically, so the three limits are
                    // counts.  Alpha is the reference's own recurrence (wgsl:313-318), applied to the dense samples.
                    uint32_t n_t = 0, n_a = 0;
                    float a_run = acc_a, a_after[N], my_w[J];
#pragma unroll
                    for (int sidx = 0; sidx < N; ++sidx) {
                        n_t += ts[sidx] < t_end ? 1u : 0u;
                        n_a += a_run < 0.95f ? 1u : 0u;
                        const float w = (1.0f - a_run) * a4[sidx >> 2][sidx & 3];
                        if ((sidx & 3) == 0) my_w[sidx >> 2] = w;
                        else my_w[sidx >> 2] = kq == static_cast<uint32_t>(sidx & 3) ? w : my_w[sidx >> 2];
                        a_run = (((cm & ~sm) >> sidx) & 1u) ? a_run + w : a_run;
                        a_after[sidx] = a_run;
                    }
                    const uint32_t full = (1u << N) - 1u;
                    const uint32_t mism = cm ^ (last_dense ? full : 0u);
                    const uint32_t n_c = min(static_cast<uint32_t>(__builtin_ctz(mism | (1u << N))) + 1u, static_cast<uint32_t>(N));
                    const uint32_t n_acc = min(min(n_t, n_a), n_c);          // >= 1 on an active lane
                    const uint32_t em = active ? (cm & ~sm & ((1u << n_acc) - 1u)) : 0u;   // accepted dense samples emit, unless suppressed
                    bool my_emit[J];
#pragma unroll
                    for (int j = 0; j < J; ++j) my_emit[j] = ((em >> (4 * j + kq)) & 1u) != 0u;
                    if (TRACE && kq == 0u && active) trace_accepted += n_acc;
                    // state after the last accepted sample
                    const uint32_t last = n_acc - 1u;
                    float t_sel = ts[0], cb_sel = cb[0], a_sel = a_after[0];
#pragma unroll
                    for (int sidx = 1; sidx < N; ++sidx) {
                        const bool ge = last >= static_cast<uint32_t>(sidx);
                        t_sel = ge ? ts[sidx] : t_sel; cb_sel = ge ? cb[sidx] : cb_sel; a_sel = ge ? a_after[sidx] : a_sel;
                    }
                    const bool dl = ((cm >> last) & 1u) != 0u;
                    const bool last_dense_before = last_dense;
                    const float cur_new = dl ? min_step : __builtin_fminf(base, cb_sel * 1.5f);    // wgsl:263-269 with the real class
                    if (active) {
                        cur = cur_new;
                        t = t_sel + cur_new;                                  // wgsl:272, :325
                        acc_a = a_sel;
                        last_dense = dl;
                    }
                    leap_ok = !dl && !last_dense_before && n_acc == static_cast<uint32_t>(N);
                    try_leap = __ballot(active && leap_ok) != 0ull;
                    active = active && t < t_end && acc_a < 0.95f;
#pragma unroll
                    for (int j = 0; j < J; ++j) append(my_emit[j], my_pos[j], my_w[j], own | (my_b[j] << 8), 0.0f);
                    if (TRACE) tm_samp += PQ_TICK() - tm_mark;
                }
            } else {
            const bool use_alpha_dp = imp_coloring || (flags & F_OPACITY) != 0u;
            while (__ballot(active) != 0ull) {
                tile_iters++;
                if (TRACE) { trace_iters++; trace_dp_iters++; trace_lanes += static_cast<uint32_t>(__popcll(__ballot(active))); tm_mark = PQ_TICK(); }
                leap_phase();
                if (TRACE) { const unsigned long long now = PQ_TICK(); tm_leap += now - tm_mark; tm_mark = now; }
                // positions of the four speculative samples under "class stays last_dense" (wgsl:263-274)
                float ts4[4], cs4[4];
                {
                    float tt = t, cc = cur;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        ts4[j] = tt;
                        cc = last_dense ? min_step : __builtin_fminf(base, cc * 1.5f);
                        cs4[j] = cc;
                        tt += cc;
                    }
                }
                const float tk = kq == 0u ? ts4[0] : (kq == 1u ? ts4[1] : (kq == 2u ? ts4[2] : ts4[3]));
                const V3 pos = ray.o + ray.d * tk;                    // wgsl:251
                const uint32_t off = nearest_offset(g, pos);
                uint32_t b = 0, ib = 0;
                float rho_k = 0.0f;
                if (TABLE) b = vol[off];                               // clamped offset: no guard needed
                if (active && need_imp) ib = imp[off];
                if (!TABLE && active) {                                // wgsl:253-259
                    uint32_t cnt = 0;
                    rho_k = gauss ? sample_density_smoothed<false>(g, s_rho, linear, fp, pos, ray.d, cnt) : sample_density(g, s_rho, linear, pos);
                }
                if (TABLE) drain();                                    // continuous-rho modes shade inside append (registers)
                const bool dense_k = TABLE ? b >= fp.thr_byte : rho_k >= thr;   // b/255 >= thr; a NaN density is not dense
                float a_k;
                if (imp_coloring) a_k = s_ic_alpha[ib];
                else if (TABLE) a_k = s_tf_tab[b].w;
                else a_k = dense_k ? 1.0f - wgsl_pow(1.0f - sample_tf(s_lut, fp.tf_n, rho_k).w, fp.alpha_y) : 0.0f;   // wgsl:314
                const uint32_t quad_d = static_cast<uint32_t>(__ballot(dense_k) >> qsh) & 15u;
                const uint32_t quad_c = static_cast<uint32_t>(__ballot(tk < t_end) >> qsh) & 15u;
                // importance rendering (wgsl:283-295): every lane looks ahead for its own sample -- one chain per lane, or the
                // 8 cone chains of 8 samples per round -- and the verdicts travel like the classes.  A sample that is not
                // accepted in the end only cost fetches.
                uint32_t quad_s = 0u;
                if (imp_rendering && !imp_coloring) {
                    const bool need_k = active && dense_k && ib < 255u;
                    bool ahead_k;
                    if (flags & F_CONE) {
                        ahead_k = ahead_cone_wave(g, fp, need_k, pos, ray.d, ray.t_exit, lane, fp.cone_cos[lane & 7u] * 0.2f, fp.cone_sin[lane & 7u] * 0.2f);
                    } else {
                        const float tks[1] = {tk};
                        const bool needs[1] = {need_k};
                        bool founds[1];
                        ahead_straight_wave<1>(g, fp, needs, tks, ray.o, ray.d, ray.t_exit, lane, mail, founds);
                        ahead_k = founds[0];
                    }
                    quad_s = static_cast<uint32_t>(__ballot(need_k && ahead_k) >> qsh) & 15u;
                }
                const int a_bits = __float_as_int(a_k);
                const float a4[4] = {__int_as_float(__builtin_amdgcn_mov_dpp(a_bits, 0x00, 0xf, 0xf, true)),
                                     __int_as_float(__builtin_amdgcn_mov_dpp(a_bits, 0x55, 0xf, 0xf, true)),
                                     __int_as_float(__builtin_amdgcn_mov_dpp(a_bits, 0xaa, 0xf, 0xf, true)),
                                     __int_as_float(__builtin_amdgcn_mov_dpp(a_bits, 0xff, 0xf, 0xf, true))};
                // accept samples in order with their real classes; identical in the four lanes
                float alpha = acc_a, my_w = 0.0f;
                bool my_emit = false, done = false, finished = false;
                int last = -1;
                if (active) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (!done) {
                            if (!(((quad_c >> j) & 1u) != 0u && alpha < 0.95f)) {      // wgsl:250
                                done = true; finished = true;
                            } else {
                                const bool dj = ((quad_d >> j) & 1u) != 0u;
                                last = j;
                                if (dj && ((quad_s >> j) & 1u) == 0u) {               // a suppressed sample only advances (wgsl:290-293)
                                    if (use_alpha_dp) {                               // wgsl:313-318
                                        const float w = (1.0f - alpha) * a4[j];
                                        if (static_cast<uint32_t>(j) == kq) { my_w = w; my_emit = true; }
                                        alpha += w;
                                    } else {                                          // wgsl:319-323
                                        if (static_cast<uint32_t>(j) == kq) { my_w = 1.0f; my_emit = true; }
                                        alpha = 1.0f;
                                        done = true; finished = true;
                                    }
                                }
                                if (dj != last_dense) done = true;                    // later positions are off
                            }
                        }
                    }
                    if (TRACE && kq == 0u) trace_accepted += static_cast<uint32_t>(last + 1);
                    if (last >= 0) {
                        const bool dl = ((quad_d >> last) & 1u) != 0u;
                        const float cur_before = last == 0 ? cur : (last == 1 ? cs4[0] : (last == 2 ? cs4[1] : cs4[2]));
                        const float t_last = last == 0 ? ts4[0] : (last == 1 ? ts4[1] : (last == 2 ? ts4[2] : ts4[3]));
                        cur = dl ? min_step : __builtin_fminf(base, cur_before * 1.5f);   // wgsl:263-269 with the real class
                        if (!(finished && !use_alpha_dp && dl)) t = t_last + cur;         // wgsl:272, :325 (not after the first-hit break)
                        last_dense = dl;
                    }
                    acc_a = alpha;
                    if (finished) active = false;
                }
                append(my_emit, pos, my_w, own | (b << 8) | (ib << 16), rho_k);
                if (TRACE) tm_samp += PQ_TICK() - tm_mark;
            }
            }
        } else {
        while (__ballot(active) != 0ull) {
            tile_iters++;
            if (TRACE) { trace_iters++; trace_lanes += static_cast<uint32_t>(__popcll(__ballot(active))); tm_mark = PQ_TICK(); if (active) trace_ray_iters++; }
            leap_phase();
            if (TRACE) { const unsigned long long now = PQ_TICK(); tm_leap += now - tm_mark; tm_mark = now; }
            // ---- 2. K speculative samples: positions under the prediction "class stays last_dense" ----
            float ts[K];
            uint32_t offs[K], bs[K], ibs[K];
            float rhos[K];
            uint32_t taps[K];                                       // reference fetches of sample k (instrumented launch)
            {
                float tt = t, cc = cur;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    ts[k] = tt;
                    cc = last_dense ? min_step : __builtin_fminf(base, cc * 1.5f);
                    tt += cc;
                }
                if constexpr (K % 2 == 0) {
                    // two samples per instruction: o + d * t and the texel scale as packed f32 multiplies and adds (the same
                    // IEEE operations per component, nothing fused)
#pragma unroll
                    for (int k = 0; k < K; k += 2) {
                        const f32x2 tp = {ts[k], ts[k + 1]};
                        const f32x2 ux = (ray.o.x + ray.d.x * tp) * g.fnx, uy = (ray.o.y + ray.d.y * tp) * g.fny, uz = (ray.o.z + ray.d.z * tp) * g.fnz;   // wgsl:251
                        offs[k] = voxel_offset(g, clamp_texel(floor_to_int(ux.x), static_cast<int>(g.hix)), clamp_texel(floor_to_int(uy.x), static_cast<int>(g.hiy)),
                                               clamp_texel(floor_to_int(uz.x), static_cast<int>(g.hiz)));
                        offs[k + 1] = voxel_offset(g, clamp_texel(floor_to_int(ux.y), static_cast<int>(g.hix)), clamp_texel(floor_to_int(uy.y), static_cast<int>(g.hiy)),
                                                   clamp_texel(floor_to_int(uz.y), static_cast<int>(g.hiz)));
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < K; ++k) offs[k] = nearest_offset(g, ray.o + ray.d * ts[k]);   // wgsl:251
                }
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    bs[k] = 0; ibs[k] = 0; rhos[k] = 0.0f; taps[k] = 1;
                    // the clamped texel selection keeps every offset inside the volume, whatever t is: the byte gathers
                    // need no guard (a finished lane re-reads its last texels; nothing uses them)
                    if (TABLE && !LB) bs[k] = vol[offs[k]];
                    if (IR) ibs[k] = imp[offs[k]];
                    if (active) {
                        if (!IR && need_imp) ibs[k] = imp[offs[k]];
                        if (!TABLE) {                                   // wgsl:253-259, all K densities in flight together
                            const V3 p = ray.o + ray.d * ts[k];
                            uint32_t cnt = 0;
                            if (gauss) rhos[k] = sample_density_smoothed<true>(g, s_rho, linear, fp, p, ray.d, cnt);
                            else { rhos[k] = sample_density(g, s_rho, linear, p); cnt = 1; }
                            taps[k] = cnt;
                        }
                    }
                }
            }

            // LB: which bricks do the K x 64 sample positions of this iteration touch?  One round per distinct brick of a sample
            // slot: a brick already staged is a tag compare; a new one takes a place nobody referenced in this iteration and is
            // fetched by 16 lanes (64 contiguous bytes), four bricks per load instruction, at most eight per iteration; lanes whose
            // brick found no place gather their byte from global memory as the product kernel does.
            uint32_t lb_slots = 0xffffffffu;                        // place of sample k in byte k (0xff: from global memory)
            uint32_t lb_ld[2] = {0u, 0u}, lb_dst = 0xffffu;         // the dwords this lane fetched, and the places they go to
            if constexpr (LB) {
                const uint32_t* const vol32 = reinterpret_cast<const uint32_t*>(vol);
                const unsigned long long act = __ballot(active);
                uint32_t used = 0u, n_new = 0u, lb_b[2] = {0u, 0u};
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const uint32_t bk = offs[k] >> 6;
                    unsigned long long todo = act;
                    while (todo != 0ull) {
                        const uint32_t b = __builtin_amdgcn_readlane(bk, __builtin_ctzll(todo));
                        const unsigned long long same = __ballot(bk == b) & todo;
                        todo &= ~same;
                        const uint32_t hit = static_cast<uint32_t>(__ballot(lb_tag == b)) & ((1u << LB_SLOTS) - 1u);
                        uint32_t sl;
                        if (hit != 0u) {
                            sl = static_cast<uint32_t>(__builtin_ctz(hit));
                            if (VOLYM_DEV_SWITCHES && (fp.dev & 512u) && lane == 0u) atomicAdd(&counters->n_hit, 1ull);
                        } else {
                            const uint32_t free_places = ~used & ((1u << LB_SLOTS) - 1u);
                            if (free_places == 0u || n_new == 8u) {
                                if (VOLYM_DEV_SWITCHES && (fp.dev & 512u) && lane == 0u) atomicAdd(&counters->n_dense, 1ull);
                                continue;                                   // no place: these lanes gather
                            }
                            const uint32_t from_next = free_places & ~((1u << lb_next) - 1u);
                            sl = static_cast<uint32_t>(__builtin_ctz(from_next != 0u ? from_next : free_places));
                            lb_next = (sl + 1u) & (LB_SLOTS - 1u);
                            if (lane == sl) lb_tag = b;
                            if ((lane >> 4) == (n_new & 3u)) {              // the 16 lanes that fetch this brick
                                if (n_new < 4u) { lb_b[0] = b; lb_dst = (lb_dst & 0xff00u) | sl; }
                                else { lb_b[1] = b; lb_dst = (lb_dst & 0x00ffu) | (sl << 8); }
                            }
                            n_new++;
                            if (VOLYM_DEV_SWITCHES && (fp.dev & 512u) && lane == 0u) atomicAdd(&counters->n_imp, 1ull);
                        }
                        used |= 1u << sl;
                        if ((same >> lane) & 1ull) lb_slots = (lb_slots & ~(0xffu << (8 * k))) | (sl << (8 * k));
                    }
                }
                if (VOLYM_DEV_SWITCHES && (fp.dev & 512u) && lane == 0u) atomicAdd(&counters->n_vol, 1ull);      // iterations
                if ((lb_dst & 0xffu) != 0xffu) lb_ld[0] = vol32[lb_b[0] * 16u + (lane & 15u)];
                if ((lb_dst >> 8) != 0xffu) lb_ld[1] = vol32[lb_b[1] * 16u + (lane & 15u)];
#pragma unroll
                for (int k = 0; k < K; ++k)
                    if (active && ((lb_slots >> (8 * k)) & 0xffu) == 0xffu) bs[k] = vol[offs[k]];
            }

            // table mode: shade here, while the byte gathers are in flight; the continuous-rho modes shade at the end of
            // the iteration instead, where the K densities are no longer live (their shading alone needs ~100 registers)
            if (TABLE) drain();
            if constexpr (LB) {
                // the fetched bricks into their places, then every staged sample's byte out of them (one wave: LDS accesses of a
                // wave complete in order)
                uint32_t* const lbw = s_lb[wave];
                if ((lb_dst & 0xffu) != 0xffu) lbw[(lb_dst & 0xffu) * 16u + (lane & 15u)] = lb_ld[0];
                if ((lb_dst >> 8) != 0xffu) lbw[(lb_dst >> 8) * 16u + (lane & 15u)] = lb_ld[1];
                const uint8_t* const lbb = reinterpret_cast<const uint8_t*>(lbw);
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const uint32_t sl = (lb_slots >> (8 * k)) & 0xffu;
                    if (sl != 0xffu) bs[k] = lbb[sl * 64u + (offs[k] & 63u)];
                }
            }

            // ---- 3. accept samples in order with their real classes ----
            if constexpr (TABLE && !COUNT && !IMP) {
                // Straight-line form for the pinned flags (opacity on, no importance mode): the same decisions and the
                // same f32 operations as the general form below, as selects instead of branches, and the K opacity
                // look-ups issued together.
                float a_tab[K];
#pragma unroll
                for (int k = 0; k < K; ++k) a_tab[k] = s_tf_tab[bs[k]].w;
                const bool predicted = last_dense;
                // importance rendering (wgsl:283-295): the look-ahead of every sample that can still be accepted as dense,
                // side by side (straight) or spread over the wave (cone); a suppressed sample only advances
                bool supp[K];
#pragma unroll
                for (int k = 0; k < K; ++k) supp[k] = false;
                if (IR) {
                    bool need[K];
                    bool chain = active;
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        const bool dense_k = bs[k] >= fp.thr_byte;
                        need[k] = chain && dense_k && ibs[k] < 255u;
                        chain = chain && dense_k == predicted;
                    }
                    if constexpr (CJ != 0) {
                        cj_lookahead(need, ts, ray.o, ray.d, ray.t_exit, supp);
                    } else if (flags & F_CONE) {
#pragma unroll
                        for (int k = 0; k < K; ++k)
                            supp[k] = ahead_cone_wave(g, fp, need[k], ray.o + ray.d * ts[k], ray.d, ray.t_exit, lane, fp.cone_cos[lane & 7u] * 0.2f, fp.cone_sin[lane & 7u] * 0.2f);
                    } else {
                        bool any_need = false;
#pragma unroll
                        for (int k = 0; k < K; ++k) any_need = any_need || need[k];
                        if (__ballot(any_need) != 0ull) ahead_straight_wave<K>(g, fp, need, ts, ray.o, ray.d, ray.t_exit, lane, mail, supp, &la_rounds);
                    }
#pragma unroll
                    for (int k = 0; k < K; ++k) supp[k] = supp[k] && need[k];
                }
                bool valid = active;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const bool go = valid && t < t_end && acc_a < 0.95f;             // wgsl:250
                    if (TRACE && go) trace_accepted++;
                    const bool dense = bs[k] >= fp.thr_byte;                          // <=> b/255 >= thr
                    const bool emit = go && dense && !supp[k];
                    const float w = (1.0f - acc_a) * a_tab[k];                        // wgsl:313-318
                    const V3 pos = ray.o + ray.d * t;                                 // == ps[k] while go
                    acc_a = emit ? acc_a + w : acc_a;
                    const float cur_next = dense ? min_step : __builtin_fminf(base, cur * 1.5f);   // wgsl:263-269
                    cur = go ? cur_next : cur;
                    t = go ? t + cur_next : t;                                        // wgsl:272, :325
                    last_dense = go ? dense : last_dense;
                    valid = go && dense == predicted;                                 // later speculative positions are off
                    if (k == K - 1) leap_ok = go && !dense && !predicted;
                    append(emit, pos, w, own | (bs[k] << 8), 0.0f);
                }
                active = active && t < t_end && acc_a < 0.95f;                        // wgsl:250, one iteration early
                if (TRACE) tm_samp += PQ_TICK() - tm_mark;
                continue;
            }
            // straight look-ahead of all K samples side by side (uninstrumented launches; the instrumented one counts the
            // probes the reference executes, sample by sample, below)
            bool ahead_pre[K];
            constexpr bool PRE_AHEAD = IMP && !COUNT && K > 1;
            const bool pre_ahead = PRE_AHEAD && imp_rendering && !imp_coloring;
            if (PRE_AHEAD && pre_ahead && (flags & F_CONE)) {
                // cone look-ahead: one phase per speculative sample, the 8 directions of 8 samples at a time on the 64 lanes
                bool chain = active;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const bool dense_k = TABLE ? bs[k] >= fp.thr_byte : rhos[k] >= thr;
                    const bool need_k = chain && dense_k && ibs[k] < 255u;
                    chain = chain && dense_k == last_dense;
                    ahead_pre[k] = ahead_cone_wave(g, fp, need_k, ray.o + ray.d * ts[k], ray.d, ray.t_exit, lane, fp.cone_cos[lane & 7u] * 0.2f, fp.cone_sin[lane & 7u] * 0.2f);
                }
            } else if (PRE_AHEAD && pre_ahead) {
                bool need[K];
                bool chain = active;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const bool dense_k = TABLE ? bs[k] >= fp.thr_byte : rhos[k] >= thr;
                    need[k] = chain && dense_k && ibs[k] < 255u;
                    chain = chain && dense_k == last_dense;
                }
                ahead_straight_wave<K>(g, fp, need, ts, ray.o, ray.d, ray.t_exit, lane, mail, ahead_pre);
            }
            bool valid = active;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                bool emit = false;
                V3 pos = v3(0.0f, 0.0f, 0.0f);
                float w = 0.0f;
                if (valid && !(t < t_end && acc_a < 0.95f)) { active = false; valid = false; }   // wgsl:250
                if (valid) {
                    if (TRACE) trace_accepted++;
                    if (COUNT) { n_steps++; n_imp++; }            // wgsl:260 fetches importance every step
                    pos = ray.o + ray.d * t;                      // t == ts[k] bit for bit while valid
                    bool dense;
                    if (TABLE) {
                        if (COUNT) n_vol++;
                        dense = bs[k] >= fp.thr_byte;             // <=> b/255 >= thr
                    } else {
                        if (COUNT) n_vol += taps[k];
                        dense = rhos[k] >= thr;
                    }
                    const bool predicted = last_dense;
                    cur = dense ? min_step : __builtin_fminf(base, cur * 1.5f);   // wgsl:263-269
                    last_dense = dense;
                    if (dense != predicted) valid = false;        // later speculative positions are off
                    bool advance = true;
                    if (dense) {
                        if (COUNT) n_dense++;
                        bool use_alpha = (flags & F_OPACITY) != 0u;
                        float alpha_step;
                        bool suppressed = false;
                        if (imp_coloring) {
                            alpha_step = s_ic_alpha[ibs[k]];
                            use_alpha = true;                     // wgsl:279-281
                        } else {
                            if (imp_rendering) {                  // wgsl:283-295
                                bool ahead;
                                if (PRE_AHEAD && pre_ahead) ahead = ahead_pre[k];
                                else ahead = (flags & F_CONE) ? ahead_cone<COUNT>(g, fp, pos, ray.d, ray.t_exit, n_imp)
                                                              : ahead_straight<COUNT>(g, fp, pos, ray.d, ray.t_exit, n_imp);
                                suppressed = ibs[k] < 255u && ahead;
                            }
                            if (TABLE) alpha_step = s_tf_tab[bs[k]].w;
                            else alpha_step = 1.0f - wgsl_pow(1.0f - sample_tf(s_lut, fp.tf_n, rhos[k]).w, fp.alpha_y);   // wgsl:314
                        }
                        if (!suppressed) {
                            if (COUNT) n_vol += 6;                // the six gradient taps, wgsl:181-188
                            emit = true;
                            if (use_alpha) {                      // wgsl:313-318
                                w = (1.0f - acc_a) * alpha_step;
                                acc_a += w;
                            } else {                              // wgsl:319-323: first hit, break
                                w = 1.0f;
                                acc_a = 1.0f;
                                advance = false;
                                active = false;
                                valid = false;
                            }
                        }
                    }
                    if (advance) t += cur;                        // wgsl:272, :292, :325
                }
                append(emit, pos, w, own | (bs[k] << 8) | (ibs[k] << 16), rhos[k]);
            }
            if (TRACE) tm_samp += PQ_TICK() - tm_mark;
        }
        }
        drain();
        if (q_count) flush(q_count);

        // ---- store (wgsl:328-329; rgba8unorm) ----
        if (in_frame && (!dp || (lane & 3u) == 0u)) {
            const float out_r = static_cast<float>(acc_r[own]) * PQ_FIX_INV, out_g = static_cast<float>(acc_g[own]) * PQ_FIX_INV,
                        out_b = static_cast<float>(acc_b[own]) * PQ_FIX_INV;
            const uint32_t packed = pack_rgba8(out_r, out_g, out_b, acc_a);
            if (flags & F_RASTER) {
                const size_t o = static_cast<size_t>(gy) * fp.W + gx;
                out_raster[o] = packed;
                // (traced launch: the f32 frame carries the ray's and the tile's iteration counts instead, scripts/ray_lengths.py)
                if (flags & F_WRITE_F32) out_f32[o] = TRACE ? make_float4(static_cast<float>(trace_ray_iters), static_cast<float>(tile_iters), out_b, acc_a) : make_float4(out_r, out_g, out_b, acc_a);
            } else {
                out_shard[static_cast<size_t>(local_tile) * 256u + sub * 64u + sub_lane] = packed;
            }
        } else if (!in_frame && !(flags & F_RASTER) && (!dp || (lane & 3u) == 0u)) {
            out_shard[static_cast<size_t>(local_tile) * 256u + sub * 64u + sub_lane] = 0u;
        }
        // cost of the tile as the scheduler sees it, in units of ~56 instructions: ray set-up, loop iterations, shading
        // batches, replayed steps (counted, not timed: the work list must not depend on the weather)
        // a depth-parallel iteration takes 8 samples per ray where the classic loop takes 4: its iterations count double, so
        // that a quarter's number estimates what its tile would cost as one item (the host takes the maximum of the four)
#ifndef VOLYM_COST_FLUSH
#define VOLYM_COST_FLUSH 2u
#endif
// a round of 64 straight look-ahead chains in cost units: measured over bonsai / teapot at 1080p and configs[4] (1024^3 + labels at
// 4K): 0 leaves the teapot at 111 us (77 with any weight from 3 to 36) and the bonsai at 68 (64); 9 costs configs[4] 6 % (156 us
// against 147: its frame is bound by throughput, and every tile the extra cost pushes over the split threshold adds work)
#ifndef VOLYM_COST_LA
#define VOLYM_COST_LA 3u
#endif
#ifndef VOLYM_COST_CJ
#define VOLYM_COST_CJ 14u
#endif
#ifndef VOLYM_COST_FLUSH_DP
#define VOLYM_COST_FLUSH_DP 2u
#endif
        entry_cost += dp ? tile_iters * 10u + tile_flushes * VOLYM_COST_FLUSH_DP + tile_trips / 7u : 5u + tile_iters * 5u + tile_flushes * VOLYM_COST_FLUSH + tile_trips / 7u;
        if (IR) entry_cost += (la_rounds - la_at_start) * VOLYM_COST_LA;           // a round of 64 chains of N probes
        if (CJ == 1) entry_cost += (cj_count - cj_at_start) * VOLYM_COST_CJ / 8u;   // a cone job of 8 samples is ~14 units of whichever wave walks it
        if (CJ == 2) entry_cost += (cj_count - cj_at_start) * VOLYM_COST_LA / 64u;  // 64 chains: one round
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
      if (cost && lane == 0) cost[list_pos] = static_cast<uint16_t>(min(65535u, entry_cost));
      }
    }

    if (CJ) {
        // out of tiles: walk the other waves' cone jobs until every wave of the workgroup is out of tiles
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        if (lane == 0u) atomicSub(&s_cj_ctl[2], 1u);
        for (uint32_t turns = 0; turns < (1u << 20); ++turns) {
            if (cj_serve()) continue;
            if (*reinterpret_cast<volatile uint32_t*>(&s_cj_ctl[2]) == 0u) break;
            __builtin_amdgcn_s_sleep(4);
        }
    }
    if (cost && lane == 0) wg_time[blockIdx.x * WAVES + wave] = static_cast<uint32_t>(__builtin_amdgcn_s_memrealtime());
    if (TRACE) for (int sft = 32; sft > 0; sft >>= 1) trace_accepted += __shfl_xor(trace_accepted, sft, 64);
    if (TRACE && lane == 0) {
        const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
        const size_t rec = (static_cast<size_t>(blockIdx.x) * WAVES + wave) * 2u;
        trace[rec] = make_uint4(static_cast<uint32_t>(trace_t0), static_cast<uint32_t>(t1 - trace_t0), trace_iters | (trace_tiles << 16), min(trace_flushes, 0xfffu) | (min(trace_marched, 15u) << 12) | (min(trace_dp_iters, 0xffffu) << 16));
        trace[rec + 1] = make_uint4(static_cast<uint32_t>(tm_leap >> 4), static_cast<uint32_t>((tm_samp - tm_flush) >> 4), static_cast<uint32_t>(tm_flush >> 4),
                                    min(trace_lanes, 0xffffu) | (min(trace_accepted, 0xffffu) << 16));   // lanes active at the loop top, samples accepted
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

}  // namespace volym
