// libvolym_hip.so: context + C ABI (include/volym_hip.h) over the gfx950 kernels.
// Replaces the reference's gpu_context.rs / gpu_resources/* / demos/pipeline.rs for the
// ray-march path; citations are file:line under /root/reference/.
//
// Threading contract: volym_update and volym_compute_pass only enqueue -- no hipMalloc, no
// hipFree, no stream synchronisation on their path.  Everything that allocates or waits lives in the set-up calls
// (volym_create, volym_set_*, volym_set_option, volym_set_shard) and in the explicitly blocking ones (volym_sync,
// volym_settle, volym_read_*, volym_stats_pass, volym_time_*).  The cost feedback of the work lists runs on a thread
// of its own (below, "cost feedback").
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <chrono>
#include <cstring>
#include <queue>
#include <string>
#include <vector>

#include "context.hpp"
#include "raymarch_kernels.h"
#include "raymarch_pq.h"
#include "raymarch_pool.h"
#include "blit.h"

// the common instantiation lives in raymarch_common.hip (its own scheduler flag)
namespace volym {
extern template __global__ void volym_raymarch_pq_kernel<true, false, false, 4, false, false, false, PQ_WAVES, 0, false>(
    const uint8_t* __restrict__, const uint8_t* __restrict__, const FrameTables* __restrict__, const uint8_t* __restrict__, const uint2* __restrict__, uint32_t,
    uint16_t* __restrict__, uint32_t* __restrict__, uint32_t* __restrict__, float4* __restrict__, Counters* __restrict__, uint4* __restrict__, const FrameParams);
}  // namespace volym

using namespace volym;

static_assert(sizeof(volym_camera_uniforms) == 208, "CameraUniforms is 208 bytes (src/gpu_resources/camera.rs:56-64)");
static_assert(sizeof(volym_parameter_uniforms) == 32, "ParameterUniforms is 32 bytes (src/gpu_resources/parameters.rs:55-66)");

// cos/sin of (s/8) * 2 * 3.14159 for s = 0..7 (wgsl:99-103), f32
static const float k_cone_cos[8] = {0x1p+0f, 0x1.6a09f6p-1f, 0x1.54442ep-20f, -0x1.6a09bap-1f,
                                    -0x1p+0f, -0x1.6a0a32p-1f, -0x1.fe6644p-19f, 0x1.6a097ep-1f};
static const float k_cone_sin[8] = {0x0p+0f, 0x1.6a09d8p-1f, 0x1p+0f, 0x1.6a0a14p-1f,
                                    0x1.54442ep-19f, -0x1.6a099cp-1f, -0x1p+0f, -0x1.6a0a5p-1f};

static thread_local std::string g_create_error;

int volym::ctx_fail(volym_ctx* c, int code, const std::string& msg)
{
    if (c) c->err = msg; else g_create_error = msg;
    return code;
}
static int fail(volym_ctx* c, int code, const std::string& msg) { return ctx_fail(c, code, msg); }
#define HIPCHK(ctx, expr) VOLYM_HIPCHK(ctx, expr)

static void recompute_shard(volym_ctx* c)
{
    c->n_local = c->n_tiles > c->rank ? (c->n_tiles - c->rank + c->world - 1) / c->world : 0;
    c->shard_tiles = (c->n_tiles + c->world - 1) / c->world;   // equal-sized shards, padded
}

static uint32_t max_grid(const volym_ctx* c) { return static_cast<uint32_t>(c->n_cus) * c->wgs_per_cu; }

// Does a plain frame with these flags run the ray pool (variant 3, raymarch_pool.h)?  The common instantiation only: nearest
// filter, no smoothing, opacity on, no importance mode; every other flag set runs variant 2.
static bool frame_uses_pool(const volym_ctx* c, uint32_t flags)
{
    return c->kernel_variant == 3 && !(flags & (F_LINEAR | F_GAUSSIAN | F_IMP_COLORING | F_IMP_RENDERING)) && (flags & F_OPACITY);
}

// ---- host-side table construction (EXACT arithmetic, same recipe as the device) --------------
static void host_texel_linear(float u, int n, int& i0, int& i1, float& w)
{
    const float x = u * static_cast<float>(n) - 0.5f;
    float fl = std::floor(x);
    w = x - fl;
    if (!(fl >= -2.0f)) fl = -2.0f;
    if (fl > static_cast<float>(n)) fl = static_cast<float>(n);
    const int i = static_cast<int>(fl);
    i0 = i < 0 ? 0 : (i > n - 1 ? n - 1 : i);
    i1 = i + 1 < 0 ? 0 : (i + 1 > n - 1 ? n - 1 : i + 1);
}

static void build_tables(volym_ctx* c, FrameTables& t, float alpha_y)
{
    const int n = static_cast<int>(c->tf_n);
    for (int b = 0; b < 256; ++b) {
        t.rho[b] = static_cast<float>(b) / 255.0f;
        const uint8_t* q = c->lut + 4 * (b < n ? b : n - 1);
        t.lut_f[b] = make_float4(static_cast<float>(q[0]) / 255.0f, static_cast<float>(q[1]) / 255.0f,
                                 static_cast<float>(q[2]) / 255.0f, static_cast<float>(q[3]) / 255.0f);
    }
    for (int b = 0; b < 256; ++b) {
        int i0, i1;
        float w;
        host_texel_linear(t.rho[b], n, i0, i1, w);   // wgsl:297-302: rho is the coordinate
        const float4 a = t.lut_f[i0], bb = t.lut_f[i1];
        const float iw = 1.0f - w;
        const float A = a.w * iw + bb.w * w;
        t.tf_tab[b] = make_float4(a.x * iw + bb.x * w, a.y * iw + bb.y * w, a.z * iw + bb.z * w,
                                  1.0f - wgsl_pow(1.0f - A, alpha_y));   // wgsl:314
        t.ic_alpha[b] = 1.0f - wgsl_pow(1.0f - t.rho[b], alpha_y);       // wgsl:83-84, :314
    }
}// ---- exact culling inputs (raymarch_pq.h): hulls of the projected unit cube and of the projected AABB of the
// occupied macro cells, in pixel coordinates, plus the AABB itself.  Double precision on the host; the kernel
// applies a 1.5 pixel margin, far above the f32 noise of the per-pixel ray set-up it stands in for. ----
namespace {
struct P2 { double x, y; };

bool invert4d(const double m[16], double inv[16])
{
    double a[4][8];
    for (int r = 0; r < 4; ++r)
        for (int col = 0; col < 4; ++col) { a[r][col] = m[col * 4 + r]; a[r][4 + col] = r == col ? 1.0 : 0.0; }
    for (int i = 0; i < 4; ++i) {
        int piv = i;
        for (int r = i + 1; r < 4; ++r) if (std::fabs(a[r][i]) > std::fabs(a[piv][i])) piv = r;
        if (std::fabs(a[piv][i]) < 1e-300) return false;
        if (piv != i) for (int k = 0; k < 8; ++k) std::swap(a[i][k], a[piv][k]);
        const double d = a[i][i];
        for (int k = 0; k < 8; ++k) a[i][k] /= d;
        for (int r = 0; r < 4; ++r) if (r != i) { const double f = a[r][i]; if (f != 0.0) for (int k = 0; k < 8; ++k) a[r][k] -= f * a[i][k]; }
    }
    for (int r = 0; r < 4; ++r) for (int col = 0; col < 4; ++col) inv[col * 4 + r] = a[r][4 + col];
    return true;
}

// convex hull (monotone chain) of <= 8 points -> edges (a, b, c, 1) with |(a,b)| = 1, inside >= 0
bool hull_edges(const P2* pts, int n, float out[8][4])
{
    std::vector<P2> p(pts, pts + n);
    std::sort(p.begin(), p.end(), [](const P2& u, const P2& v) { return u.x < v.x || (u.x == v.x && u.y < v.y); });
    auto cross = [](const P2& o, const P2& u, const P2& v) { return (u.x - o.x) * (v.y - o.y) - (u.y - o.y) * (v.x - o.x); };
    std::vector<P2> h(2 * p.size());
    int k = 0;
    for (size_t i = 0; i < p.size(); ++i) { while (k >= 2 && cross(h[k - 2], h[k - 1], p[i]) <= 0) --k; h[k++] = p[i]; }
    for (size_t i = p.size() - 1, t = k + 1; i > 0; --i) { while (k >= static_cast<int>(t) && cross(h[k - 2], h[k - 1], p[i - 1]) <= 0) --k; h[k++] = p[i - 1]; }
    const int m = k - 1;                          // closed polygon, last == first
    for (int e = 0; e < 8; ++e) out[e][0] = out[e][1] = out[e][2] = out[e][3] = 0.0f;
    if (m < 3 || m > 8) return false;
    double cx = 0, cy = 0;
    for (int i = 0; i < m; ++i) { cx += h[i].x; cy += h[i].y; }
    cx /= m; cy /= m;
    for (int i = 0; i < m; ++i) {
        const P2 &u = h[i], &v = h[(i + 1) % m];
        double a = -(v.y - u.y), b = v.x - u.x;
        const double len = std::sqrt(a * a + b * b);
        if (len < 1e-9) return false;
        a /= len; b /= len;
        double cc = -(a * u.x + b * u.y);
        if (a * cx + b * cy + cc < 0) { a = -a; b = -b; cc = -cc; }
        out[i][0] = static_cast<float>(a); out[i][1] = static_cast<float>(b); out[i][2] = static_cast<float>(cc); out[i][3] = 1.0f;
    }
    return true;
}
}  // namespace

static void compute_culling(volym_ctx* c)
{
    FrameParams& fp = c->fp;
    fp.cull = 0;
    std::memset(fp.hull, 0, sizeof fp.hull);
    // variant 3's lattice rectangle: the whole frame unless the hull of the occupied cells says less (below)
    fp.rect[0] = 0; fp.rect[1] = 0;
    fp.rect[2] = (c->W + PL_SBW - 1u) / PL_SBW * PL_SBW; fp.rect[3] = (c->H + PL_SBH - 1u) / PL_SBH * PL_SBH;
    c->hull_dirty = false;
    c->mask_wanted = false;
    if (!c->culling) return;
    // AABB of the occupied cells (per threshold byte, computed when the volume was set), grown by what a sample may
    // reach beyond its own position
    const int* h_aabb = c->aabb_tab[std::min(c->thr_byte_cull, 256u)];
    const bool none = h_aabb[3] < h_aabb[0];
    if (none) fp.cull |= CULL_NOTHING_DENSE;
    double lo[3] = {0, 0, 0}, hi[3] = {1, 1, 1};
    const double margin = 1.0e-4 + ((fp.flags & F_GAUSSIAN) ? 0.0101 : 0.0);   // smoothing taps sit up to 2*0.005 along the ray (wgsl:53-60)
    if (!none) {
        for (int i = 0; i < 3; ++i) {
            lo[i] = static_cast<double>(h_aabb[i]) / c->mc_n - margin;
            hi[i] = static_cast<double>(h_aabb[3 + i] + 1) / c->mc_n + margin;
            fp.aabb_lo[i] = static_cast<float>(lo[i]);
            fp.aabb_hi[i] = static_cast<float>(hi[i]);
        }
        fp.cull |= CULL_AABB;
    }
    // world -> clip as the exact inverse of the matrix the rays are generated from
    double ivp[16], M[16];
    for (int i = 0; i < 16; ++i) ivp[i] = (&c->cam_copy.inverse_view_proj[0][0])[i];
    if (!invert4d(ivp, M)) return;
    auto clip = [&](double x, double y, double z, double out[4]) {
        for (int r = 0; r < 4; ++r) out[r] = M[0 * 4 + r] * x + M[1 * 4 + r] * y + M[2 * 4 + r] * z + M[3 * 4 + r];
    };
    // the eye must be the centre of projection of that matrix (w == 0), otherwise the hulls say nothing about the rays
    double ce[4];
    clip(fp.eye[0], fp.eye[1], fp.eye[2], ce);
    const double scale = std::fabs(M[3]) + std::fabs(M[7]) + std::fabs(M[11]) + std::fabs(M[15]);
    if (!(std::fabs(ce[3]) <= 1e-4 * scale)) return;
    double bb[4] = {0, 0, 0, 0};      // bounding rectangle of the last projected box
    auto project_box = [&](const double blo[3], const double bhi[3], float out[8][4]) -> bool {
        P2 pts[8];
        for (int k = 0; k < 8; ++k) {
            double q[4];
            clip((k & 1) ? bhi[0] : blo[0], (k & 2) ? bhi[1] : blo[1], (k & 4) ? bhi[2] : blo[2], q);
            if (!(q[3] > 1e-3 * scale)) return false;       // a corner at or behind the eye plane: no hull
            pts[k].x = (q[0] / q[3] + 1.0) * 0.5 * c->W;     // wgsl:221-229 inverted: pixel = (ndc + 1)/2 * W
            pts[k].y = (1.0 - q[1] / q[3]) * 0.5 * c->H;
            if (!std::isfinite(pts[k].x) || !std::isfinite(pts[k].y)) return false;
            bb[0] = k ? std::min(bb[0], pts[k].x) : pts[k].x; bb[1] = k ? std::min(bb[1], pts[k].y) : pts[k].y;
            bb[2] = k ? std::max(bb[2], pts[k].x) : pts[k].x; bb[3] = k ? std::max(bb[3], pts[k].y) : pts[k].y;
        }
        return hull_edges(pts, 8, out);
    };
    const double c0[3] = {0, 0, 0}, c1[3] = {1, 1, 1};
    if (project_box(c0, c1, fp.hull[0])) fp.cull |= CULL_CUBE_HULL;
    if (!none && project_box(lo, hi, fp.hull[1])) {
        fp.cull |= CULL_OBJ_HULL;
        // every pixel outside the hull by more than its 1.5 pixel margin is constant: so is every pixel outside the hull's bounding
        // rectangle grown by 3 pixels, rounded outwards to whole superblocks
        const double x0 = std::max(0.0, std::floor((bb[0] - 3.0) / PL_SBW) * PL_SBW), y0 = std::max(0.0, std::floor((bb[1] - 3.0) / PL_SBH) * PL_SBH);
        const double x1 = std::min(static_cast<double>(fp.rect[2]), std::ceil((bb[2] + 3.0) / PL_SBW) * PL_SBW), y1 = std::min(static_cast<double>(fp.rect[3]), std::ceil((bb[3] + 3.0) / PL_SBH) * PL_SBH);
        if (x1 > x0 && y1 > y0) { fp.rect[0] = static_cast<uint32_t>(x0); fp.rect[1] = static_cast<uint32_t>(y0); fp.rect[2] = static_cast<uint32_t>(x1); fp.rect[3] = static_cast<uint32_t>(y1); }
        else { fp.rect[0] = fp.rect[1] = fp.rect[2] = fp.rect[3] = 0; }       // the object is off screen
    }
    if (none) { fp.rect[0] = fp.rect[1] = fp.rect[2] = fp.rect[3] = 0; }
    // the per-tile mask of the occupied cells' projections (volym_tile_mask_kernel): its cells lie inside the AABB whose
    // corners were all found in front of the eye, so their corners are too
    fp.mask_t8x = c->tiles_x * 2u;
    c->mask_margin = static_cast<float>(margin);
    for (int i = 0; i < 16; ++i) c->mask_clip[i] = static_cast<float>(M[i]);
    c->mask_wanted = (fp.cull & CULL_OBJ_HULL) != 0u && c->tile_mask && c->tile_mask_words != 0u;
}
// ================================================================================================================
// Work lists and their cost feedback (variant 2).
//
// The kernel's persistent workgroups read a list of items; the frame time is set by how well that list balances the few
// hundred tiles whose rays take ~10x the average number of dependent samples.  Only a rendered frame knows which they
// are, so a launch can be asked to report a counted cost per list entry ("capture"): the costs are copied to pinned host
// memory on a second stream, and a feedback THREAD -- never the caller -- turns them into the next list: most expensive
// entries first and dealt longest-processing-time first to the workgroups, the most expensive tiles split into four
// depth-parallel quarter items, constant 16x16 tiles merged into super-fill items.  The caller's volym_compute_pass only
// looks at an atomic flag: when a new list is ready it switches to it between two launches.  Lists are scheduling only:
// every list renders the same pixels, so a list measured on a neighbouring view is a good list for this one, and a
// moving camera simply keeps the feedback running (one capture in flight at a time).
// ================================================================================================================

// geometric list: the 8x8-pixel wave tiles of this rank's 16x16 tiles (item = local_tile*4 + sub), by Chebyshev
// distance of the tile centre from the screen centre.  The orbit camera always targets the volume centre
// (src/camera.rs:23), so the long rays are the central ones: they start first.
static void build_geometric(volym_ctx* c)
{
    std::vector<std::pair<uint32_t, uint32_t>> keyed;
    keyed.reserve(static_cast<size_t>(c->n_local) * 4);
    for (uint32_t lt = 0; lt < c->n_local; ++lt) {
        const uint32_t tile = lt * c->world + c->rank;
        const uint32_t tx = tile % c->tiles_x, ty = tile / c->tiles_x;
        for (uint32_t sub = 0; sub < 4; ++sub) {
            const int x0 = static_cast<int>(tx * 16u + (sub & 1u) * 8u), y0 = static_cast<int>(ty * 16u + (sub >> 1) * 8u);
            if (x0 >= static_cast<int>(c->W) || y0 >= static_cast<int>(c->H)) {
                if (c->world == 1) continue;          // wholly outside the frame: nothing to store in raster mode
            }
            const int dx = std::abs(2 * x0 + 8 - static_cast<int>(c->W)), dy = std::abs(2 * y0 + 8 - static_cast<int>(c->H));
            // rings of 16 pixels; inside a ring a hash decides, so that a workgroup (which takes every G-th item)
            // does not sit at the same angular position on every ring
            const uint32_t item = lt * 4u + sub;
            uint32_t h = item * 0x9E3779B1u;
            h ^= h >> 15; h *= 0x85EBCA77u; h ^= h >> 13;
            keyed.emplace_back((static_cast<uint32_t>(std::max(dx, dy)) / 32u) << 20 | (h & 0xfffffu), item);
        }
    }
    std::sort(keyed.begin(), keyed.end());
    c->geometric.resize(keyed.size());
    for (size_t i = 0; i < keyed.size(); ++i) c->geometric[i] = keyed[i].second;
}

// Deal `item_cost` into a list (feedback thread; also the caller's thread inside blocking set-up calls).
static void deal_list(const volym_ctx* c, const volym_ctx::FbJob& job, const std::vector<uint16_t>& measured_cost, std::vector<uint8_t>& item_is_dp,
                      const std::vector<uint32_t>& geometric, uint32_t n_local, WorkList& out)
{
    const uint32_t waves = job.waves;
    // The costs were measured on an earlier frame; when the camera moves, what was expensive there is expensive a tile or two
    // further on here.  A maximum filter over the neighbouring 8x8 items (radius job.dilate) makes the list hold for a
    // while: the price is a few tiles split or started early that did not need it.
    std::vector<uint16_t> item_cost(measured_cost);
    // Has the camera moved since the captured frame?  Then the list will be read on yet another view: dilate the costs and
    // keep split tiles split (hysteresis).  A view that stands still gets exactly what its own costs say -- but only costs
    // MEASURED on whole 8x8 entries say it well (a split tile reports an estimate).  So when the captured list held split
    // tiles, the first deal for a standing view is a measuring list without any split, and the deal after it is final.
    const bool moving = c->view_serial.load(std::memory_order_relaxed) != job.view_serial;
    const bool measuring = !moving && job.captured_has_dp && job.dp_min_cost < 0;
    const int dilate = job.dilate >= 0 ? job.dilate : (moving ? 1 : 0);
    if (dilate > 0) {
        const uint32_t gw = c->tiles_x * 2u, gh = c->tiles_y * 2u;
        std::vector<uint16_t> grid(static_cast<size_t>(gw) * gh, 0), tmp(static_cast<size_t>(gw) * gh, 0);
        auto cell_of = [&](uint32_t item) {
            const uint32_t tile = (item >> 2) * c->world + c->rank, sub = item & 3u;
            return static_cast<size_t>((tile / c->tiles_x) * 2u + (sub >> 1)) * gw + (tile % c->tiles_x) * 2u + (sub & 1u);
        };
        for (uint32_t item : geometric) grid[cell_of(item)] = measured_cost[item];
        const int r = dilate;
        for (uint32_t y = 0; y < gh; ++y)
            for (uint32_t x = 0; x < gw; ++x) {
                uint16_t m = 0;
                for (int d = -r; d <= r; ++d) { const int xx = static_cast<int>(x) + d; if (xx >= 0 && xx < static_cast<int>(gw)) m = std::max(m, grid[static_cast<size_t>(y) * gw + xx]); }
                tmp[static_cast<size_t>(y) * gw + x] = m;
            }
        for (uint32_t y = 0; y < gh; ++y)
            for (uint32_t x = 0; x < gw; ++x) {
                uint16_t m = 0;
                for (int d = -r; d <= r; ++d) { const int yy = static_cast<int>(y) + d; if (yy >= 0 && yy < static_cast<int>(gh)) m = std::max(m, tmp[static_cast<size_t>(yy) * gw + x]); }
                grid[static_cast<size_t>(y) * gw + x] = m;
            }
        for (uint32_t item : geometric) item_cost[item] = grid[cell_of(item)];
    }
    // Tiles above the threshold are split into four 4x4 quarter tiles marched depth-parallel (raymarch_pq.h): their cost is
    // a long chain of dependent samples, which four lanes per ray walk ~4x faster, on four waves.  Which tiles?  Those that
    // would keep one wave busy for more than ~1.5x a wave's fair share of the frame (sum of costs / resident waves): below
    // that they hide in the bulk and splitting only adds work.  A tile that is split stays split until its estimated cost
    // falls below 0.7x the threshold (its cost is an estimate while it is split).
    uint64_t total_cost = 0;
    for (uint32_t item : geometric) total_cost += item_cost[item];
    const uint32_t resident_waves = std::max(1u, job.max_grid * waves);
    // Measured over four scenes (profiles/r03_dp_scene_sweep.txt: bonsai, teapot, a dense ball, thin vessels; 1080p, where the
    // split matters -- at 3840x2160 every setting gives the same frame time): the common instantiation wants 1.7-1.9x (bonsai
    // 33.9 us at 1.9x against 34.3 with r02's rule, teapot 44.9 against 51.9, ball 54.1 against 60.9; the vessels do not
    // care); r02's 1.5x with an absolute floor of 104 units was the optimum of the bonsai alone and cost the other scenes
    // 10-17 %.  The look-ahead instantiations keep 1.5x, the continuous-rho modes 1.2x (their classic loop speculates only
    // two samples deep, a depth-parallel item four), as measured in r01 / r02.  A list dealt for a moving camera is read on later
    // views: there the lower threshold (more tiles split than the captured view needed) is the better one (turntable at 0.25
    // degrees per frame: 53.4 us at 1.5x, 57.0 at 1.9x).
    const uint64_t tenths = job.dp_min_cost < -1 ? static_cast<uint64_t>(-job.dp_min_cost) : (job.continuous ? 12u : (job.plain && !moving) ? 19u : 15u);
    const uint64_t floor_cost = job.dp_floor;                                   // 64 units: a tile below that is never worth four waves
    const uint32_t adaptive = static_cast<uint32_t>(std::max<uint64_t>(floor_cost, tenths * total_cost / (10u * resident_waves) + 16));
    const uint32_t dp_thr = job.dp_min_cost < 0 ? adaptive : static_cast<uint32_t>(job.dp_min_cost);
#if VOLYM_DEV_SWITCHES
    if (std::getenv("VOLYM_TRIM_LOG"))
        std::fprintf(stderr, "deal: total cost %llu, fair share %llu, floor %llu, split threshold %u (moving %d measuring %d)\n", static_cast<unsigned long long>(total_cost),
                     static_cast<unsigned long long>(total_cost / resident_waves), static_cast<unsigned long long>(floor_cost), dp_thr, moving ? 1 : 0, measuring ? 1 : 0);
#endif
    const bool dp_ok = job.dp_min_cost != 0 && !measuring;
    std::vector<std::pair<uint32_t, uint32_t>> keyed;      // (cost share, entry)
    keyed.reserve(geometric.size() * 2);
    bool has_dp = false;
    // 16x16 tiles whose four sub-tiles were all constant become one "super" fill item (bit 30)
    std::vector<uint8_t> all_fill(n_local, 1), seen(n_local, 0), cnt(n_local, 0);
    for (uint32_t item : geometric) { if (item_cost[item] != 0) all_fill[item >> 2] = 0; cnt[item >> 2]++; }
    for (uint32_t lt = 0; lt < n_local; ++lt) if (cnt[lt] != 4) all_fill[lt] = 0;   // sub-tiles outside the frame are not listed
    for (uint32_t item : geometric) {
        const uint32_t k = item_cost[item];
        if (job.super_fill && all_fill[item >> 2]) {
            if (!seen[item >> 2]) { seen[item >> 2] = 1; keyed.emplace_back(0u, 0x40000000u | (item >> 2)); }
            item_is_dp[item] = 0;
            continue;
        }
        if (job.dev_drop_tenths && static_cast<uint64_t>(k) * 10u * resident_waves >= static_cast<uint64_t>(job.dev_drop_tenths) * total_cost) { item_is_dp[item] = 0; continue; }   // dev: what if the longest tiles were not there?
        const bool split = dp_ok && (k >= dp_thr || (moving && item_is_dp[item] && job.dp_min_cost < 0 && 10u * k >= 7u * dp_thr));
        has_dp = has_dp || split;
        item_is_dp[item] = split ? 1 : 0;
        if (split)
            for (uint32_t qd = 0; qd < 4; ++qd) keyed.emplace_back((k * job.dp_share_pct + 99u) / 100u, 0x80000000u | (item << 2) | qd);
        else
            keyed.emplace_back(k, item);
    }
    {
        // stable counting sort by decreasing cost share (shares are small integers): the feedback thread's latency is what
        // a moving camera sees as the age of its list
        uint32_t kmax = 0;
        for (const auto& kv : keyed) kmax = std::max(kmax, kv.first);
        std::vector<uint32_t> start(static_cast<size_t>(kmax) + 2u, 0);
        for (const auto& kv : keyed) start[kmax - kv.first + 1u]++;
        for (size_t i = 1; i < start.size(); ++i) start[i] += start[i - 1];
        std::vector<std::pair<uint32_t, uint32_t>> sorted(keyed.size());
        for (const auto& kv : keyed) sorted[start[kmax - kv.first]++] = kv;
        keyed.swap(sorted);
    }
    if (job.only_quarters) {     // dev experiment: how long do the depth-parallel items take with the machine to themselves?
        std::vector<std::pair<uint32_t, uint32_t>> q;
        for (const auto& kv : keyed) if (kv.second >> 31) q.push_back(kv);
        keyed.swap(q);
    }
    // issue priority (bits 28-29) from the entry's cost relative to a wave's fair share of the frame
    const uint64_t fair = std::max<uint64_t>(1, total_cost / resident_waves);
    const bool prio_ok = job.prio_tenths[0] > 0 && static_cast<uint64_t>(n_local) * 16u < (1u << 28);
    const uint32_t n_keyed = static_cast<uint32_t>(keyed.size());
    const uint32_t G = std::max(1u, std::min((n_keyed + waves - 1) / waves, job.max_grid));
    // Workgroup b reads entries b, b + G, ... of the list.  The entries, in order of decreasing cost, are dealt in
    // boustrophedon order over the workgroups (0..G-1, G-1..0, ...): the sums differ by about one entry of the current size,
    // as with a longest-processing-time heap, in one pass (the feedback thread's latency is the age of a moving camera's
    // list).  Rounds are list rows: entry i sits in row i / G.
    const size_t rows = (static_cast<size_t>(n_keyed) + G - 1) / G;
    out.entries.assign(static_cast<size_t>(G) * rows, PQ_NO_ITEM);
    out.shares.assign(static_cast<size_t>(G) * rows, 0);
    for (uint32_t i = 0; i < n_keyed; ++i) {
        const std::pair<uint32_t, uint32_t>& kv = keyed[i];
        uint32_t prio = 0;
        if (prio_ok && kv.first) {
            const uint64_t k10 = static_cast<uint64_t>(kv.first) * 10u;
            prio = k10 >= job.prio_tenths[2] * fair ? 3u : k10 >= job.prio_tenths[1] * fair ? 2u : k10 >= job.prio_tenths[0] * fair ? 1u : 0u;
        }
        const uint32_t row = i / G, j = i - row * G;
        const size_t pos = static_cast<size_t>(row) * G + ((row & 1u) ? G - 1u - j : j);
        out.entries[pos] = kv.second | (prio << 28);
        out.shares[pos] = static_cast<uint16_t>(std::min(65535u, kv.first));
    }
    out.grid = G;
    out.view_serial = job.view_serial;
    out.has_dp = has_dp;
    out.trimmable = !moving && !measuring;
    out.trim_round = 0;
    out.final_for_view = out.trimmable && job.trim_rounds == 0;
}

// Re-balance a standing view's list from the times its workgroups took (feedback thread).  The counted costs predict a
// workgroup's time to within a few percent (profiles/r02_wave_trace.txt: end times spread over ~4 us of 33, and the spread
// repeats from frame to frame); the frame ends with the LAST workgroup.  So: every workgroup's measured duration gives its
// own rate (time per unit of cost, for the entries it holds); workgroups that ended after the mean hand entries worth
// `damp` x their excess to a pool, and the pool goes, largest first, to whichever workgroup is predicted to end first.
// Scheduling only: the pixels do not change.  Measured (scripts/trim_rounds.py, 1080p bonsai): 33.95 us without, 33.6-33.7 us
// with 1..8 rounds -- the spread of the end times halves (28.0..34.1 -> 29.7..32.3 us) but their MEAN rises as it does: the
// workgroups that used to finish early no longer leave the others a quieter machine.  In alternating 20 000-frame runs of two
// builds the gain is 0.1 us (33.17 -> 33.07), and one run in six came out at 33.85: a list trimmed from a capture that caught a
// hiccup is final, and wrong, for as long as the view stands.  A deterministic list is worth more than 0.3 %: OFF by default
// (FbJob::trim_rounds = 0; VOLYM_OPT_REBALANCE_ROUNDS turns it on).
static bool trim_list(const volym_ctx* c, const volym_ctx::FbJob& job, const WorkList& in, const uint32_t* times, WorkList& out)
{
    const uint32_t G = in.grid, waves = job.waves;
    if (G == 0 || G != job.grid || in.entries.size() % G != 0 || in.shares.size() != in.entries.size()) return false;
    const size_t rows = in.entries.size() / G;
    const uint32_t* starts = times + static_cast<size_t>(G) * waves;
    const uint32_t ref = starts[0];
    int32_t t0 = 0;
    for (uint32_t b = 0; b < G; ++b) t0 = std::min(t0, static_cast<int32_t>(starts[b] - ref));
    struct Ent { uint32_t w, code; };
    std::vector<std::vector<Ent>> wg(G);
    std::vector<double> dur(G), weight(G, 0.0), rate(G), pred(G);
    double mean = 0.0;
    for (uint32_t b = 0; b < G; ++b) {
        int32_t end = 0;
        for (uint32_t w = 0; w < waves; ++w) end = std::max(end, static_cast<int32_t>(times[static_cast<size_t>(b) * waves + w] - ref));
        dur[b] = std::max(1.0, static_cast<double>(end - t0));
        wg[b].reserve(rows + 8);
        for (size_t r = 0; r < rows; ++r) {
            const size_t pos = r * G + b;
            if (in.entries[pos] == PQ_NO_ITEM) continue;
            wg[b].push_back(Ent{in.shares[pos] + 1u, in.entries[pos]});
            weight[b] += in.shares[pos] + 1u;
        }
        mean += dur[b];
    }
    mean /= G;
    if (mean > 1.0e6) return false;                        // a second: not a frame's times (counter wrap, garbage)
    const double damp = 0.8;
    std::vector<Ent> pool;
    for (uint32_t b = 0; b < G; ++b) {
        rate[b] = dur[b] / std::max(1.0, weight[b]);
        pred[b] = dur[b];
        double excess = damp * (dur[b] - mean);
        if (excess <= 0.0) continue;
        // largest entries that fit first (the list of a workgroup is in order of decreasing share); constant tiles stay
        std::vector<Ent> keep;
        keep.reserve(wg[b].size());
        for (const Ent& e : wg[b]) {
            const double t = e.w * rate[b];
            if (e.w > 1u && t <= excess) { pool.push_back(e); excess -= t; pred[b] -= t; }
            else keep.push_back(e);
        }
        wg[b].swap(keep);
    }
    std::stable_sort(pool.begin(), pool.end(), [](const Ent& a, const Ent& b) { return a.w > b.w; });
    for (const Ent& e : pool) {
        uint32_t best = 0;
        double best_t = 1.0e300;
        for (uint32_t b = 0; b < G; ++b) { const double t = pred[b] + e.w * rate[b]; if (t < best_t) { best_t = t; best = b; } }
        pred[best] = best_t;
        wg[best].push_back(e);
    }
#if VOLYM_DEV_SWITCHES
    if (std::getenv("VOLYM_TRIM_LOG")) {
        double mx = 0, mn = 1e300, pmx = 0, pmn = 1e300;
        for (uint32_t b = 0; b < G; ++b) { mx = std::max(mx, dur[b]); mn = std::min(mn, dur[b]); pmx = std::max(pmx, pred[b]); pmn = std::min(pmn, pred[b]); }
        std::fprintf(stderr, "trim round %u: workgroup durations min %.2f mean %.2f max %.2f us; %zu entries moved; predicted min %.2f max %.2f\n", in.trim_round + 1, mn / 100, mean / 100,
                     mx / 100, pool.size(), pmn / 100, pmx / 100);
        double xs[8] = {}, xw[8] = {}; uint32_t xn[8] = {};
        for (uint32_t b = 0; b < G; ++b) { xs[b & 7u] += dur[b]; xw[b & 7u] += weight[b]; xn[b & 7u]++; }
        std::fprintf(stderr, "   by XCD (workgroup %% 8): mean duration");
        for (int k = 0; k < 8; ++k) std::fprintf(stderr, " %.2f", xs[k] / std::max(1u, xn[k]) / 100);
        std::fprintf(stderr, " ; mean weight");
        for (int k = 0; k < 8; ++k) std::fprintf(stderr, " %.0f", xw[k] / std::max(1u, xn[k]));
        std::fprintf(stderr, "\n");
    }
#endif
    size_t rows_out = 0;
    for (uint32_t b = 0; b < G; ++b) {
        std::stable_sort(wg[b].begin(), wg[b].end(), [](const Ent& a, const Ent& b2) { return a.w > b2.w; });
        rows_out = std::max(rows_out, wg[b].size());
    }
    if (static_cast<size_t>(G) * rows_out > c->list_capacity) return false;
    out.entries.assign(static_cast<size_t>(G) * rows_out, PQ_NO_ITEM);
    out.shares.assign(static_cast<size_t>(G) * rows_out, 0);
    for (uint32_t b = 0; b < G; ++b)
        for (size_t r = 0; r < wg[b].size(); ++r) {
            out.entries[r * G + b] = wg[b][r].code;
            out.shares[r * G + b] = static_cast<uint16_t>(wg[b][r].w - 1u);
        }
    out.grid = G;
    out.view_serial = in.view_serial;
    out.has_dp = in.has_dp;
    out.trimmable = true;
    out.trim_round = in.trim_round + 1;
    out.final_for_view = out.trim_round >= job.trim_rounds;
    return true;
}

// device form of a list: {entry, x | y << 16 of the entry's 16x16 tile} (the kernel does no integer division)
static void list_to_device_form(const volym_ctx* c, const std::vector<uint32_t>& entries, uint32_t* out)
{
    for (size_t i = 0; i < entries.size(); ++i) {
        const uint32_t raw_p = entries[i];
        uint32_t xy = 0;
        if (raw_p != PQ_NO_ITEM) {
            const uint32_t raw = raw_p & ~0x30000000u;
            const uint32_t lt = (raw >> 31) ? ((raw & 0x7fffffffu) >> 4) : ((raw >> 30) == 1u ? (raw & 0x3fffffffu) : (raw >> 2));
            const uint32_t tile = lt * c->world + c->rank;
            xy = (tile % c->tiles_x) | ((tile / c->tiles_x) << 16);
        }
        out[2 * i] = raw_p;
        out[2 * i + 1] = xy;
    }
}

// costs by list position -> costs by item
static void costs_to_items(const WorkList& list, const uint16_t* cost, uint32_t n_entries, std::vector<uint16_t>& item_cost)
{
    std::vector<uint32_t> q_max(item_cost.size(), 0);
    std::vector<uint8_t> q_seen(item_cost.size(), 0);
    for (uint32_t p = 0; p < n_entries && p < list.entries.size(); ++p) {
        const uint32_t raw_p = list.entries[p];
        if (raw_p == PQ_NO_ITEM) continue;
        const uint32_t raw = raw_p & ~0x30000000u;
        const uint32_t k = cost[p];
        if (raw >> 31) {                                   // depth-parallel quarter: the tile's cost is 5 + the slowest quarter
            const uint32_t item = (raw & 0x7fffffffu) >> 2;
            if (item < item_cost.size()) { q_max[item] = std::max(q_max[item], k); q_seen[item] = 1; }
        } else if ((raw >> 30) == 1u) {                    // super fill: zero, or the sum of four marched sub-tiles
            const uint32_t lt = raw & 0x3fffffffu;
            for (uint32_t sub = 0; sub < 4; ++sub)
                if (lt * 4u + sub < item_cost.size()) item_cost[lt * 4u + sub] = static_cast<uint16_t>(k == 0 ? 0u : std::max(1u, k / 4u));
        } else if (raw < item_cost.size()) {
            item_cost[raw] = static_cast<uint16_t>(k);
        }
    }
    for (size_t i = 0; i < item_cost.size(); ++i)
        if (q_seen[i]) item_cost[i] = static_cast<uint16_t>(std::min(65535u, 5u + q_max[i]));
}

static void feedback_thread(volym_ctx* c)
{
    (void)hipSetDevice(c->device);
    for (;;) {
        {
            std::unique_lock<std::mutex> lk(c->fb_mu);
            c->fb_cv.wait(lk, [&] { const int s = c->fb_state.load(std::memory_order_acquire); return s == volym_ctx::FB_CAPTURED || s == volym_ctx::FB_QUIT; });
        }
        if (c->fb_state.load(std::memory_order_acquire) == volym_ctx::FB_QUIT) return;
        volym_ctx::FbJob& job = c->fb_job;
        job.error.clear();
        auto now_us = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        job.t_us[1] = now_us();
        hipError_t e = hipEventSynchronize(c->ev_cost);            // the captured launch and its cost copy are done
        job.t_us[2] = now_us();
        const int next = job.list ^ 1;
        if (e == hipSuccess) {
            // A list dealt for this very view from whole-entry costs is not dealt again (its split tiles report estimates): it is
            // re-balanced from the workgroups' measured times, job.trim_rounds times.
            const WorkList& ran = c->lists[job.list];
            const bool same_view = c->view_serial.load(std::memory_order_relaxed) == job.view_serial && ran.view_serial == job.view_serial;
            bool trimmed = false;
            if (same_view && ran.trimmable && ran.trim_round < job.trim_rounds)
                trimmed = trim_list(c, job, ran, reinterpret_cast<const uint32_t*>(c->h_cost_pinned + ((job.n_entries + 1u) & ~1u)), c->lists[next]);
            job.t_us[3] = now_us();
            if (!trimmed) {
                costs_to_items(ran, c->h_cost_pinned, job.n_entries, c->item_cost);
                job.t_us[3] = now_us();
                deal_list(c, job, c->item_cost, c->item_is_dp, c->geometric, c->n_local, c->lists[next]);
            }
            job.t_us[4] = now_us();
            if (c->lists[next].entries.size() > c->list_capacity) {
                job.error = "work list larger than its buffers";    // cannot happen: capacity is the worst case
            } else {
                list_to_device_form(c, c->lists[next].entries, c->h_list_pinned);
                // every launch that read d_list[next] finished before the captured launch did (same stream, in order)
                e = hipMemcpyAsync(c->d_list[next], c->h_list_pinned, c->lists[next].entries.size() * 2u * sizeof(uint32_t), hipMemcpyHostToDevice, c->copy_stream);
                if (e == hipSuccess) e = hipEventRecord(c->ev_list, c->copy_stream);
                if (e == hipSuccess) e = hipEventSynchronize(c->ev_list);
            }
        }
        if (e != hipSuccess) job.error = std::string("cost feedback: ") + hipGetErrorString(e);
        job.t_us[5] = now_us();
        {
            // under the mutex: a caller in feedback_quiesce that has just found the state CAPTURED must be inside wait() before this
            // store and its notification happen, or it would sleep through them
            std::lock_guard<std::mutex> lk(c->fb_mu);
            c->fb_state.store(volym_ctx::FB_READY, std::memory_order_release);
        }
        c->fb_cv.notify_all();
    }
}

// caller side: adopt a finished list (never blocks)
static void feedback_poll(volym_ctx* c)
{
    if (c->fb_state.load(std::memory_order_acquire) != volym_ctx::FB_READY) return;
    if (c->fb_job.error.empty()) c->cur = c->fb_job.list ^ 1;
    c->fb_state.store(volym_ctx::FB_IDLE, std::memory_order_release);
}

// caller side, blocking (set-up calls, volym_settle): wait for a job in flight and adopt its list
static void feedback_quiesce(volym_ctx* c)
{
    if (c->fb_state.load(std::memory_order_acquire) == volym_ctx::FB_CAPTURED) {
        std::unique_lock<std::mutex> lk(c->fb_mu);
        c->fb_cv.wait(lk, [&] { return c->fb_state.load(std::memory_order_acquire) != volym_ctx::FB_CAPTURED; });
    }
    feedback_poll(c);
}

// (Re)build the lists of this shard: geometric order, no costs.  Blocking set-up path (create, set_shard, options).
static int rebuild_lists(volym_ctx* c)
{
    feedback_quiesce(c);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    build_geometric(c);
    // worst case: every 8x8 item split into four quarters, plus padding to a multiple of the grid
    const size_t need = static_cast<size_t>(c->n_local) * 16u + 2u * static_cast<size_t>(c->n_cus) * 8u + 64u;
    if (need > c->list_capacity) {
        for (int i = 0; i < 2; ++i) { if (c->d_list[i]) (void)hipFree(c->d_list[i]); c->d_list[i] = nullptr; }
        if (c->d_cost) (void)hipFree(c->d_cost);
        if (c->h_list_pinned) (void)hipHostFree(c->h_list_pinned);
        if (c->h_cost_pinned) (void)hipHostFree(c->h_cost_pinned);
        c->d_cost = nullptr; c->h_list_pinned = nullptr; c->h_cost_pinned = nullptr; c->list_capacity = 0;
        hipError_t e = hipMalloc(&c->d_list[0], need * 2u * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMalloc(&c->d_list[1], need * 2u * sizeof(uint32_t));
        // + the times of the captured launch: a u32 per wave and per workgroup (raymarch_pq.h wg_time)
        const size_t cost_bytes = (need + 2u) * sizeof(uint16_t) + (static_cast<size_t>(c->n_cus) * 8u * (PQ_WAVES + 1u) + 64u) * sizeof(uint32_t);
        if (e == hipSuccess) e = hipMalloc(&c->d_cost, cost_bytes);
        if (e == hipSuccess) e = hipHostMalloc(&c->h_list_pinned, need * 2u * sizeof(uint32_t), hipHostMallocDefault);
        if (e == hipSuccess) e = hipHostMalloc(&c->h_cost_pinned, cost_bytes, hipHostMallocDefault);
        if (e != hipSuccess) return fail(c, VOLYM_E_NOMEM, std::string("work lists: ") + hipGetErrorString(e));
        c->list_capacity = need;
    }
    c->item_cost.assign(static_cast<size_t>(c->n_local) * 4, 0);
    c->item_is_dp.assign(static_cast<size_t>(c->n_local) * 4, 0);
    c->cur = 0;
    c->lists[0] = WorkList();                 // (no split entries, not final, dealt for no view)
    c->lists[0].entries = c->geometric;
    c->lists[1] = WorkList();
    if (!c->geometric.empty()) {
        list_to_device_form(c, c->geometric, c->h_list_pinned);
        HIPCHK(c, hipMemcpy(c->d_list[0], c->h_list_pinned, c->geometric.size() * 2u * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    c->lists_ready = true;
    return VOLYM_OK;
}

// costs no longer describe the scene (new volume, transfer function, threshold grid...): back to the geometric order
static int forget_costs(volym_ctx* c)
{
    if (!c->lists_ready) return VOLYM_OK;
    return rebuild_lists(c);
}

// Macro-cell maxima, their host copy and the occupied-cell AABB for every threshold byte (set-up path: blocks).
static int build_macro_cells(volym_ctx* c)
{
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->d_mc) { HIPCHK(c, hipFree(c->d_mc)); c->d_mc = nullptr; }
    if (c->d_df) { HIPCHK(c, hipFree(c->d_df)); c->d_df = nullptr; }
    const uint32_t n = c->mc_n, cells = n * n * n;
    hipError_t e = hipMalloc(&c->d_mc, cells);
    if (e == hipSuccess) e = hipMalloc(&c->d_df, (cells / 2u + 15u) / 16u * 16u);
    if (e != hipSuccess) return fail(c, VOLYM_E_NOMEM, std::string("hipMalloc(macro cells): ") + hipGetErrorString(e));
    hipLaunchKernelGGL(volym_macrocell_kernel, dim3(cells), dim3(256), 0, c->stream, c->d_vol, c->d_mc, c->nx, c->ny, c->nz, c->mc_n, c->bricked ? 1u : 0u);
    HIPCHK(c, hipGetLastError());
    c->h_mc.resize(cells);
    HIPCHK(c, hipMemcpyAsync(c->h_mc.data(), c->d_mc, cells, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    // AABB of the cells whose maximum reaches b, for every b: boxes of the cells with maximum exactly v, then a suffix union
    int box[257][6];
    for (int v = 0; v <= 256; ++v) { box[v][0] = box[v][1] = box[v][2] = 1 << 30; box[v][3] = box[v][4] = box[v][5] = -1; }
    for (uint32_t z = 0; z < n; ++z)
        for (uint32_t y = 0; y < n; ++y)
            for (uint32_t x = 0; x < n; ++x) {
                int* b = box[c->h_mc[(z * n + y) * n + x]];
                const int p[3] = {static_cast<int>(x), static_cast<int>(y), static_cast<int>(z)};
                for (int i = 0; i < 3; ++i) { b[i] = std::min(b[i], p[i]); b[3 + i] = std::max(b[3 + i], p[i]); }
            }
    int run[6] = {1 << 30, 1 << 30, 1 << 30, -1, -1, -1};
    for (int i = 0; i < 6; ++i) c->aabb_tab[256][i] = i < 3 ? 0 : -1;         // threshold byte 256: nothing is dense
    for (int v = 255; v >= 0; --v) {
        for (int i = 0; i < 3; ++i) { run[i] = std::min(run[i], box[v][i]); run[3 + i] = std::max(run[3 + i], box[v][3 + i]); }
        for (int i = 0; i < 6; ++i) c->aabb_tab[v][i] = run[3] < 0 ? (i < 3 ? 0 : -1) : run[i];
    }
    c->df_thr_byte = 0xffffffffu;
    c->hull_dirty = true;
    return VOLYM_OK;
}

extern "C" {

int volym_abi_version(void) { return VOLYM_ABI_VERSION; }

const char* volym_last_error(const volym_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

// ---- frames in flight ------------------------------------------------------------------------------------------------
// A frame of the persistent kernel ends on its longest chains: for the last fifth of its time the CUs empty one by one
// (DESIGN.md 5).  With VOLYM_OPT_FRAMES_IN_FLIGHT = 2 the context owns a twin -- a complete second context on the same device:
// stream, frame buffer, volume copy, work lists, feedback thread -- and volym_compute_pass alternates between the two, so the
// next frame's workgroups start on every CU the previous frame has left (measured: 31.9 -> 27.6 us per frame at 1920x1080).
// Everything that describes the scene goes to both; the reading calls take the frame of the latest pass.
#define TWIN_FORWARD(c, call)                                                                              \
    do {                                                                                                   \
        if ((c) && (c)->twin) {                                                                            \
            const int rt_ = (call);                                                                        \
            if (rt_ != VOLYM_OK) return fail((c), rt_, std::string("second frame context: ") + (c)->twin->err); \
        }                                                                                                  \
    } while (0)
#define TWIN_REFUSE(c, what)                                                                               \
    do {                                                                                                   \
        if ((c) && (c)->twin) return fail((c), VOLYM_E_STATE, what ": not with VOLYM_OPT_FRAMES_IN_FLIGHT = 2"); \
    } while (0)

int volym_create(volym_ctx** out, uint32_t width, uint32_t height, int device_id)
{
    if (!out) return fail(nullptr, VOLYM_E_INVALID, "volym_create: out is NULL");
    *out = nullptr;
    if (width == 0 || height == 0 || width > 32768 || height > 32768)
        return fail(nullptr, VOLYM_E_INVALID, "volym_create: viewport must be 1..32768 in each dimension");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
        return fail(nullptr, VOLYM_E_NO_DEVICE, "volym_create: no HIP device visible");
    int dev = device_id;
    if (dev < 0) { if (hipGetDevice(&dev) != hipSuccess) dev = 0; }
    if (dev >= n_dev) return fail(nullptr, VOLYM_E_NO_DEVICE, "volym_create: device_id out of range");
    hipDeviceProp_t prop;
    HIPCHK(nullptr, hipGetDeviceProperties(&prop, dev));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, VOLYM_E_NO_DEVICE,
                    std::string("volym_create: this library carries gfx950 code only, device is ") + prop.gcnArchName);
    HIPCHK(nullptr, hipSetDevice(dev));

    volym_ctx* c = new (std::nothrow) volym_ctx();
    if (!c) return fail(nullptr, VOLYM_E_NOMEM, "volym_create: out of host memory");
    c->device = dev;
    c->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    c->W = width; c->H = height;
    c->tiles_x = (width + 15u) / 16u;     // src/demos/pipeline.rs:83-87
    c->tiles_y = (height + 15u) / 16u;
    c->n_tiles = c->tiles_x * c->tiles_y;
    recompute_shard(c);
    std::memset(&c->fp, 0, sizeof c->fp);
    std::memset(&c->tables_now, 0, sizeof c->tables_now);
    std::memset(c->aabb_tab, 0, sizeof c->aabb_tab);

    auto bail = [&](hipError_t e, const char* what) {
        std::string m = std::string(what) + ": " + hipGetErrorString(e);
        volym_destroy(c);
        return fail(nullptr, e == hipErrorOutOfMemory ? VOLYM_E_NOMEM : VOLYM_E_HIP, m);
    };
    hipError_t e;
    if ((e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking)) != hipSuccess) return bail(e, "hipStreamCreate");
    if ((e = hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking)) != hipSuccess) return bail(e, "hipStreamCreate");
    c->stream = c->own_stream;
    const size_t frame_bytes = static_cast<size_t>(width) * height * 4;
    if ((e = hipMalloc(&c->d_frame_own, frame_bytes)) != hipSuccess) return bail(e, "hipMalloc(frame)");
    if ((e = hipMalloc(&c->d_shard_own, static_cast<size_t>(c->n_tiles) * 1024)) != hipSuccess) return bail(e, "hipMalloc(shard)");
    if ((e = hipMalloc(&c->d_tables, sizeof(FrameTables))) != hipSuccess) return bail(e, "hipMalloc(tables)");
    if ((e = hipMalloc(&c->d_counters, sizeof(Counters))) != hipSuccess) return bail(e, "hipMalloc(counters)");
    if ((e = hipMemset(c->d_frame_own, 0, frame_bytes)) != hipSuccess) return bail(e, "hipMemset(frame)");
    if ((e = hipMalloc(&c->d_aabb, 6 * sizeof(int))) != hipSuccess) return bail(e, "hipMalloc(aabb)");
    {
        const uint64_t tiles8 = static_cast<uint64_t>(c->tiles_x) * 2u * c->tiles_y * 2u;
        const uint64_t words = (tiles8 + 31u) / 32u;
        c->tile_mask_words = words <= VOLYM_TILE_MASK_MAX_WORDS ? static_cast<uint32_t>(words) : 0u;
        if (c->tile_mask_words && (e = hipMalloc(&c->d_tile_mask, 2u * c->tile_mask_words * sizeof(uint32_t))) != hipSuccess) return bail(e, "hipMalloc(tile mask)");
        if (c->tile_mask_words && (e = hipMemset(c->d_tile_mask, 0, 2u * c->tile_mask_words * sizeof(uint32_t))) != hipSuccess) return bail(e, "hipMemset(tile mask)");
    }
    if ((e = hipMalloc(&c->d_pack_counters, 4 * sizeof(uint32_t))) != hipSuccess) return bail(e, "hipMalloc(pack counters)");
    if ((e = hipMemset(c->d_pack_counters, 0, 4 * sizeof(uint32_t))) != hipSuccess) return bail(e, "hipMemset(pack counters)");
    if ((e = hipMalloc(&c->d_pool_sync, 4 * sizeof(uint32_t))) != hipSuccess) return bail(e, "hipMalloc(pool sync)");
    if ((e = hipMemset(c->d_pool_sync, 0, 4 * sizeof(uint32_t))) != hipSuccess) return bail(e, "hipMemset(pool sync)");
    if ((e = hipMalloc(&c->d_pool_dbg, static_cast<size_t>(c->n_cus) * 8u * PL_WAVES * 24u * sizeof(uint32_t))) != hipSuccess) return bail(e, "hipMalloc(pool timeline)");
    for (int i = 0; i < volym_ctx::TABLE_RING; ++i) {
        if ((e = hipHostMalloc(reinterpret_cast<void**>(&c->h_tables[i]), sizeof(FrameTables), hipHostMallocDefault)) != hipSuccess) return bail(e, "hipHostMalloc(tables)");
        if ((e = hipEventCreateWithFlags(&c->tables_ev[i], hipEventDisableTiming)) != hipSuccess) return bail(e, "hipEventCreate");
    }
    if ((e = hipEventCreateWithFlags(&c->ev_march, hipEventDisableTiming)) != hipSuccess) return bail(e, "hipEventCreate");
    if ((e = hipEventCreateWithFlags(&c->ev_cost, hipEventDisableTiming)) != hipSuccess) return bail(e, "hipEventCreate");
    if ((e = hipEventCreateWithFlags(&c->ev_list, hipEventDisableTiming)) != hipSuccess) return bail(e, "hipEventCreate");

    c->d_frame = c->d_frame_own;
    c->d_shard = c->d_shard_own;
    int rc = rebuild_lists(c);
    if (rc != VOLYM_OK) { const std::string m = c->err; volym_destroy(c); return fail(nullptr, rc, m); }
    try {
        c->fb_thread = std::thread(feedback_thread, c);
    } catch (...) {
        volym_destroy(c);
        return fail(nullptr, VOLYM_E_NOMEM, "volym_create: cannot start the feedback thread");
    }
    *out = c;
    return VOLYM_OK;
}

void volym_destroy(volym_ctx* c)
{
    if (!c) return;
    if (c->twin) { volym_destroy(c->twin); c->twin = nullptr; }
    (void)hipSetDevice(c->device);
    if (c->fb_thread.joinable()) {
        feedback_quiesce(c);
        {
            std::lock_guard<std::mutex> lk(c->fb_mu);
            c->fb_state.store(volym_ctx::FB_QUIT, std::memory_order_release);
        }
        c->fb_cv.notify_all();
        c->fb_thread.join();
    }
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
    (void)hipFree(c->d_vol); (void)hipFree(c->d_imp); (void)hipFree(c->d_tables); (void)hipFree(c->d_mc); (void)hipFree(c->d_df);
    (void)hipFree(c->d_shard_own); (void)hipFree(c->d_frame_own); (void)hipFree(c->d_f32); (void)hipFree(c->d_blit);
    (void)hipFree(c->d_gather_tmp); (void)hipFree(c->d_pack_counters); (void)hipFree(c->d_counters); (void)hipFree(c->d_aabb); (void)hipFree(c->d_tile_mask);
    (void)hipFree(c->d_list[0]); (void)hipFree(c->d_list[1]); (void)hipFree(c->d_cost);
    (void)hipFree(c->d_pool_sync); (void)hipFree(c->d_pool_dbg);
    if (c->h_list_pinned) (void)hipHostFree(c->h_list_pinned);
    if (c->h_cost_pinned) (void)hipHostFree(c->h_cost_pinned);
    for (int i = 0; i < volym_ctx::TABLE_RING; ++i) {
        if (c->h_tables[i]) (void)hipHostFree(c->h_tables[i]);
        if (c->tables_ev[i]) (void)hipEventDestroy(c->tables_ev[i]);
    }
    for (uint32_t i = 0; i < volym_ctx::THROTTLE_RING; ++i) if (c->throttle_ev[i]) (void)hipEventDestroy(c->throttle_ev[i]);
    if (c->ev_march) (void)hipEventDestroy(c->ev_march);
    if (c->ev_cost) (void)hipEventDestroy(c->ev_cost);
    if (c->ev_list) (void)hipEventDestroy(c->ev_list);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    delete c;
}

int volym_set_stream(volym_ctx* c, void* hip_stream)
{
    if (!c) return VOLYM_E_INVALID;
    TWIN_REFUSE(c, "volym_set_stream");
    HIPCHK(c, hipSetDevice(c->device));
    feedback_quiesce(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->own_stream;
    return VOLYM_OK;
}

int volym_settle(volym_ctx* c)
{
    if (!c) return VOLYM_E_INVALID;
    TWIN_FORWARD(c, volym_settle(c->twin));
    feedback_quiesce(c);
    if (!c->fb_job.error.empty()) { const std::string m = c->fb_job.error; c->fb_job.error.clear(); return fail(c, VOLYM_E_HIP, m); }
    // ... and run the feedback to its fixed point for the current view: frames of this view (what volym_compute_pass
    // enqueues) until the list in use is final -- measuring list, deal, re-balancing rounds (raymarch.hip, "cost feedback")
    if (c->have_frame && c->kernel_variant >= 2 && !frame_uses_pool(c, c->fp.flags) && c->feedback && !c->feedback_frozen && c->lists_ready) {
        for (int round = 0; round < 12; ++round) {
            const WorkList& wl = c->lists[c->cur];
            if (wl.entries.empty() || (wl.view_serial == c->view_serial.load(std::memory_order_relaxed) && wl.final_for_view)) break;
            int rc = volym::ctx_launch_march(c);
            if (rc != VOLYM_OK) return rc;
            feedback_quiesce(c);
            if (!c->fb_job.error.empty()) { const std::string m = c->fb_job.error; c->fb_job.error.clear(); return fail(c, VOLYM_E_HIP, m); }
        }
    }
    return VOLYM_OK;
}

static bool want_bricked(const volym_ctx* c, uint32_t nx, uint32_t ny, uint32_t nz);

static int set_option_one(volym_ctx* c, int key, int value);

int volym_set_option(volym_ctx* c, int key, int value)
{
    if (!c) return VOLYM_E_INVALID;
    if (key == VOLYM_OPT_FRAMES_IN_FLIGHT) {
        if (value != 1 && value != 2) return fail(c, VOLYM_E_INVALID, "VOLYM_OPT_FRAMES_IN_FLIGHT: 1 or 2");
        if (value == 1) {
            if (c->twin) {
                int rc = volym_sync(c);
                if (rc != VOLYM_OK) return rc;
                volym_destroy(c->twin);
                c->twin = nullptr; c->last = c->last_blit = nullptr; c->flight_parity = 0;
            }
            return VOLYM_OK;
        }
        if (c->twin) return VOLYM_OK;
        if (c->have_vol || c->have_imp || c->have_tf)
            return fail(c, VOLYM_E_STATE, "VOLYM_OPT_FRAMES_IN_FLIGHT = 2: set it before the volume, the importances and the transfer function");
        if (c->stream != c->own_stream) return fail(c, VOLYM_E_STATE, "VOLYM_OPT_FRAMES_IN_FLIGHT = 2: not with a caller's stream (volym_set_stream)");
        volym_ctx* t = nullptr;
        int rc = volym_create(&t, c->W, c->H, c->device);
        if (rc != VOLYM_OK) return fail(c, rc, std::string("second frame context: ") + volym_last_error(nullptr));
        for (const auto& kv : c->option_log) {
            rc = set_option_one(t, kv.first, kv.second);
            if (rc != VOLYM_OK) { const std::string m = t->err; volym_destroy(t); return fail(c, rc, "second frame context: " + m); }
        }
        if (c->world != 1u) {
            rc = volym_set_shard(t, c->rank, c->world);
            if (rc != VOLYM_OK) { const std::string m = t->err; volym_destroy(t); return fail(c, rc, "second frame context: " + m); }
        }
        c->twin = t;
        return VOLYM_OK;
    }
    const int rc = set_option_one(c, key, value);
    if (rc != VOLYM_OK) return rc;
    // (the log is what a twin created later starts from: it can only be created before the scene is set)
    if (!c->twin && !(c->have_vol || c->have_imp || c->have_tf)) c->option_log.emplace_back(key, value);
    TWIN_FORWARD(c, set_option_one(c->twin, key, value));
    return VOLYM_OK;
}

static int set_option_one(volym_ctx* c, int key, int value)
{
    if (!c) return VOLYM_E_INVALID;
    switch (key) {
    case VOLYM_OPT_KERNEL:
        if (value < 0 || value > 3) return fail(c, VOLYM_E_INVALID, "VOLYM_OPT_KERNEL: 0 (direct), 1 (macro-cell), 2 (persistent tiles + shading queue) or 3 (ray pool)");
        c->kernel_variant = value;
        return VOLYM_OK;
    case VOLYM_OPT_WRITE_F32:
        c->write_f32 = value != 0;
        if (c->write_f32 && !c->d_f32) {
            HIPCHK(c, hipSetDevice(c->device));
            hipError_t e = hipMalloc(&c->d_f32, static_cast<size_t>(c->W) * c->H * sizeof(float4));
            if (e != hipSuccess) { c->write_f32 = false; return fail(c, VOLYM_E_NOMEM, std::string("hipMalloc(f32 frame): ") + hipGetErrorString(e)); }
        }
        return VOLYM_OK;
    case VOLYM_OPT_MACRO_CELLS:
        // (64: the development build only -- a field that does not fit the LDS, measured slower; VOLYM_DF_IN_LDS)
        if (value < 4 || value > (VOLYM_DEV_SWITCHES ? 64 : 32) || (value & (value - 1)) != 0)
            return fail(c, VOLYM_E_INVALID, "VOLYM_OPT_MACRO_CELLS: power of two in 4..32");
        c->mc_n = static_cast<uint32_t>(value);
        if (c->have_vol) { int rc = build_macro_cells(c); if (rc != VOLYM_OK) return rc; }
        return forget_costs(c);
    case VOLYM_OPT_VOLUME_LAYOUT:
        if (value < -1 || value > 1) return fail(c, VOLYM_E_INVALID, "VOLYM_OPT_VOLUME_LAYOUT: -1 (by size), 0 (linear) or 1 (4x4x4 bricks)");
        c->layout_choice = value;
        return VOLYM_OK;
    case VOLYM_OPT_CULLING:
        c->culling = value != 0;
        c->hull_dirty = true;
        return VOLYM_OK;
    case VOLYM_OPT_COST_FEEDBACK:
        c->feedback = value != 0;
        return forget_costs(c);
    case VOLYM_OPT_DEPTH_PARALLEL:
        if (value < -100 || value > 65535) return fail(c, VOLYM_E_INVALID, "VOLYM_OPT_DEPTH_PARALLEL: < 0 adaptive (-N = N/10 x fair share), 0 off, else explicit cost");
        c->dp_min_cost = value;
        return forget_costs(c);
    case VOLYM_OPT_REBALANCE_ROUNDS:   // 0: the dealt list is final
        if (value < 0 || value > 8) return fail(c, VOLYM_E_INVALID, "VOLYM_OPT_REBALANCE_ROUNDS: 0..8");
        c->trim_rounds = static_cast<uint32_t>(value);
        return VOLYM_OK;
    case VOLYM_OPT_SETUP_IEEE:
        if (value != 0 && value != 1) return fail(c, VOLYM_E_INVALID, "VOLYM_OPT_SETUP_IEEE: 0 or 1");
        c->setup_ieee = value == 1;       // read by the next volym_update
        return VOLYM_OK;
    case VOLYM_OPT_XCD_BANDS:
        if (value < 0 || value > 64) return fail(c, VOLYM_E_INVALID, "VOLYM_OPT_XCD_BANDS: 0..64");
        c->xcd_bands = static_cast<uint32_t>(value);
        return VOLYM_OK;
#if VOLYM_DEV_SWITCHES
    // tuning knobs of the development build (make DEV=1; scripts/ablate.py)
    case 101:   // persistent workgroups per CU (variant 2)
        if (value < 1 || value > 8) return fail(c, VOLYM_E_INVALID, "workgroups per CU: 1..8");
        c->wgs_per_cu = static_cast<uint32_t>(value);
        return forget_costs(c);
    case 107:   // 0 disables the 16x16 super fill items
        c->super_fill = value != 0;
        return forget_costs(c);
    case 108:   // issue-priority thresholds t1 + 100*t2 + 10000*t3 in tenths of the fair share (0 = no priorities)
        if (value < 0) return fail(c, VOLYM_E_INVALID, "priority thresholds: t1 + 100*t2 + 10000*t3, tenths of the fair share");
        c->prio_tenths[0] = static_cast<uint32_t>(value % 100);
        c->prio_tenths[1] = static_cast<uint32_t>((value / 100) % 100);
        c->prio_tenths[2] = static_cast<uint32_t>(value / 10000);
        return forget_costs(c);
    case 109:   // keep only the depth-parallel items in the work list (the frame is then incomplete)
        c->dev_only_quarters = value != 0;
        return forget_costs(c);
    case 110:   // FrameParams::dev
        c->fp.dev = static_cast<uint32_t>(value);
        return VOLYM_OK;
    case 113:   // 1 freezes the cost feedback: no more captures, the current list stays (how fast do lists go stale?)
        feedback_quiesce(c);
        c->feedback_frozen = value != 0;
        return VOLYM_OK;
    case 114:   // dilation radius (in 8x8 items) of the cost map when a list is dealt
        c->cost_dilate = std::max(-1, std::min(value, 4));
        return VOLYM_OK;
    case 115:   // waves per workgroup of the instantiations that need more than 128 VGPRs: 0 default, 12 or 16
        if (value != 0 && value != 12 && value != 16) return fail(c, VOLYM_E_INVALID, "wide waves: 0, 12 or 16");
        c->wide_waves = value;
        return forget_costs(c);
    case 119:   // floor of the adaptive split threshold (cost units)
        c->dp_floor = static_cast<uint32_t>(std::max(value, 1));
        return forget_costs(c);
    case 118:   // drop the tiles of >= value/10 x the fair share from the lists (the frame is then incomplete): how much do they cost?
        c->dev_drop_tenths = static_cast<uint32_t>(std::max(value, 0));
        return forget_costs(c);
    case 121:   // 1: the straight look-ahead as jobs shared by the workgroup (raymarch_pq.h CJ = 2)
        c->straight_jobs = value != 0;
        return VOLYM_OK;
    case 122:   // 1: the speculative voxel fetches of the common instantiation through LDS-staged bricks (raymarch_pq.h LB; bricked layout)
        c->lds_bricks = value != 0;
        return VOLYM_OK;
    case 117:   // 0: no per-view tile mask (the hulls and the AABB clip stay); 2: a mask for every view, on its first frame
        c->tile_mask = value != 0;
        c->mask_eager = value == 2;
        c->hull_dirty = true;
        return forget_costs(c);
    case 111:   // balancing estimates, dp_share_pct + 1000 * fill_cost
        c->dp_share_pct = static_cast<uint32_t>(value % 1000);
        c->fill_cost = static_cast<uint32_t>(value / 1000);
        return forget_costs(c);
#endif
    default:
        return fail(c, VOLYM_E_INVALID, "volym_set_option: unknown key");
    }
}

int volym_set_shard(volym_ctx* c, uint32_t rank, uint32_t world)
{
    if (!c) return VOLYM_E_INVALID;
    TWIN_FORWARD(c, volym_set_shard(c->twin, rank, world));
    if (world == 0 || rank >= world || world > 4096) return fail(c, VOLYM_E_INVALID, "volym_set_shard: need rank < world <= 4096");
    // the feedback thread reads rank / world / n_local while it deals a list: let a job in flight finish (and the frames that
    // read the current lists) before any of them changes
    feedback_quiesce(c);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->rank = rank; c->world = world;
    recompute_shard(c);
    return rebuild_lists(c);
}

static int upload_volume(volym_ctx* c, uint8_t** dst, const uint8_t* src, uint32_t nx, uint32_t ny, uint32_t nz)
{
    const bool bricked = want_bricked(c, nx, ny, nz);
    if (!src || nx == 0 || ny == 0 || nz == 0) return fail(c, VOLYM_E_INVALID, "volume: NULL data or zero dimension");
    const uint64_t n = static_cast<uint64_t>(nx) * ny * nz;
    const uint64_t nb = bricked ? static_cast<uint64_t>(brick_count(nx)) * brick_count(ny) * brick_count(nz) * 64u : n;
    if (nx > 4096 || ny > 4096 || nz > 4096 || nb > 0xffffffffull)
        return fail(c, VOLYM_E_INVALID, "volume: each dimension <= 4096 and the brick-padded size < 2^32");
    HIPCHK(c, hipSetDevice(c->device));
    feedback_quiesce(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (*dst) { HIPCHK(c, hipFree(*dst)); *dst = nullptr; }
    uint8_t* staging = nullptr;
    hipError_t e = hipMalloc(dst, nb + 16);      // the trilinear fetch reads voxel pairs: one byte past the last voxel is touched
    if (e == hipSuccess) e = hipMemset(*dst + nb, 0, 16);
    if (e == hipSuccess && bricked) e = hipMalloc(&staging, n);
    if (e != hipSuccess) { (void)hipFree(staging); return fail(c, VOLYM_E_NOMEM, std::string("hipMalloc(volume): ") + hipGetErrorString(e)); }
    e = hipMemcpy(bricked ? staging : *dst, src, n, hipMemcpyHostToDevice);
    if (e == hipSuccess && bricked) {
        hipLaunchKernelGGL(volym_rebrick_kernel, dim3(static_cast<uint32_t>((nb + 255u) / 256u)), dim3(256), 0, c->stream, staging, *dst, nx, ny, nz);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    }
    (void)hipFree(staging);
    if (e != hipSuccess) return fail(c, VOLYM_E_HIP, std::string("volume upload: ") + hipGetErrorString(e));
    return VOLYM_OK;
}

// Bricks pay once the volume outgrows the L2s (measured: from 512^3 on; see raymarch_device.h); volume and importances of
// the same dimensions get the same answer.
static bool want_bricked(const volym_ctx* c, uint32_t nx, uint32_t ny, uint32_t nz)
{
    if (c->layout_choice >= 0) return c->layout_choice == 1;
    return static_cast<uint64_t>(nx) * ny * nz > c->brick_from_bytes;
}

int volym_set_volume(volym_ctx* c, const uint8_t* voxels, uint32_t nx, uint32_t ny, uint32_t nz, int filter)
{
    if (!c) return VOLYM_E_INVALID;
    TWIN_FORWARD(c, volym_set_volume(c->twin, voxels, nx, ny, nz, filter));
    if (filter != VOLYM_FILTER_NEAREST && filter != VOLYM_FILTER_LINEAR)
        return fail(c, VOLYM_E_INVALID, "volym_set_volume: filter must be VOLYM_FILTER_NEAREST or VOLYM_FILTER_LINEAR");
    int rc = upload_volume(c, &c->d_vol, voxels, nx, ny, nz);
    if (rc != VOLYM_OK) { c->have_vol = false; return rc; }
    c->nx = nx; c->ny = ny; c->nz = nz; c->filter = filter;
    c->bricked = want_bricked(c, nx, ny, nz);
    c->have_vol = true;
    rc = build_macro_cells(c);
    if (rc != VOLYM_OK) { c->have_vol = false; return rc; }
    return forget_costs(c);
}

// AABB (texel indices) of the importances a look-ahead probe counts as important (byte >= 128, wgsl:133, :155).  Host scan,
// eight bytes at a time (bit 7 of a byte <=> the byte is >= 128); set-up path.
static void important_texel_box(const uint8_t* imp, uint32_t nx, uint32_t ny, uint32_t nz, int (&lo)[3], int (&hi)[3])
{
    int x0 = INT32_MAX, y0 = INT32_MAX, z0 = INT32_MAX, x1 = -1, y1 = -1, z1 = -1;
    for (uint32_t z = 0; z < nz; ++z)
        for (uint32_t y = 0; y < ny; ++y) {
            const uint8_t* row = imp + (static_cast<size_t>(z) * ny + y) * nx;
            int first = -1, last = -1;
            uint32_t x = 0;
            for (; x + 8u <= nx; x += 8u) {
                uint64_t w;
                std::memcpy(&w, row + x, 8);
                w &= 0x8080808080808080ull;
                if (!w) continue;
                if (first < 0) first = static_cast<int>(x) + (__builtin_ctzll(w) >> 3);
                last = static_cast<int>(x) + 7 - (__builtin_clzll(w) >> 3);
            }
            for (; x < nx; ++x)
                if (row[x] & 0x80u) { if (first < 0) first = static_cast<int>(x); last = static_cast<int>(x); }
            if (first < 0) continue;
            x0 = std::min(x0, first); x1 = std::max(x1, last);
            y0 = std::min(y0, static_cast<int>(y)); y1 = std::max(y1, static_cast<int>(y));
            z0 = std::min(z0, static_cast<int>(z)); z1 = std::max(z1, static_cast<int>(z));
        }
    if (x1 < 0) { lo[0] = lo[1] = lo[2] = 1; hi[0] = hi[1] = hi[2] = 0; return; }
    lo[0] = x0; lo[1] = y0; lo[2] = z0; hi[0] = x1; hi[1] = y1; hi[2] = z1;
}

int volym_set_importances(volym_ctx* c, const uint8_t* importances, uint32_t nx, uint32_t ny, uint32_t nz)
{
    if (!c) return VOLYM_E_INVALID;
    TWIN_FORWARD(c, volym_set_importances(c->twin, importances, nx, ny, nz));
    int rc = upload_volume(c, &c->d_imp, importances, nx, ny, nz);
    if (rc != VOLYM_OK) { c->have_imp = false; return rc; }
    important_texel_box(importances, nx, ny, nz, c->imp_box_lo, c->imp_box_hi);
    c->inx = nx; c->iny = ny; c->inz = nz;
    c->have_imp = true;
    return forget_costs(c);
}

int volym_set_transfer_function(volym_ctx* c, const uint8_t* rgba8, uint32_t n)
{
    if (!c) return VOLYM_E_INVALID;
    TWIN_FORWARD(c, volym_set_transfer_function(c->twin, rgba8, n));
    if (!rgba8 || n < 1 || n > 256) return fail(c, VOLYM_E_INVALID, "volym_set_transfer_function: 1..256 RGBA8 texels");
    std::memset(c->lut, 0, sizeof c->lut);
    std::memcpy(c->lut, rgba8, static_cast<size_t>(n) * 4);
    c->tf_n = n;
    c->have_tf = true;
    c->tables_dirty = true;
    return VOLYM_OK;
}

}  // extern "C"

// Per-frame resources that depend on the uniforms: distance field for the threshold byte (a launch, in stream order) and
// the culling hulls (host arithmetic).  Nothing here allocates or waits.
static int ensure_frame_resources(volym_ctx* c)
{
    if (c->df_thr_byte != c->thr_byte_cull) {
        // stream order: earlier frames finish reading d_df before this kernel rewrites it
        if (c->mc_n <= 32u) hipLaunchKernelGGL(volym_distance_field_kernel<1>, dim3(1), dim3(1024), 0, c->stream, c->d_mc, c->d_df, c->d_aabb, c->mc_n, c->thr_byte_cull);
        else hipLaunchKernelGGL(volym_distance_field_kernel<4>, dim3(1), dim3(1024), 0, c->stream, c->d_mc, c->d_df, c->d_aabb, c->mc_n, c->thr_byte_cull);
        HIPCHK(c, hipGetLastError());
        c->df_thr_byte = c->thr_byte_cull;
        c->hull_dirty = true;
    }
    if (c->hull_dirty) {
        compute_culling(c);
        c->fp.tile_mask = nullptr;
        c->fp.mask_words = c->tile_mask_words;
        c->fp.tile_mask_spare = c->d_tile_mask ? c->d_tile_mask + static_cast<size_t>(c->mask_cur ^ 1) * c->tile_mask_words : nullptr;
        c->mask_pending = c->mask_wanted;
        c->view_launches = 0;
    }
    // The tile mask costs a kernel per view (~13 us: device-scope atomics) and changes which tiles are constant from view to view.
    // It is built when a view is rendered a SECOND time: a standing view has it from its second frame on; a moving camera
    // never pays for it -- and does not want it: the lists it runs were dealt for earlier views, and a 16x16 entry that was
    // constant under that view's mask is four marched tiles on one wave under this one's (turntable at 3840x2160, 1 degree per
    // frame: 130 us per frame without a mask per view, 164 with; at 1080p, 0.25 degrees: 52 and 62).
    if (c->mask_pending && (c->view_launches >= 1u || c->mask_eager)) {
        ClipMatrix M;
        std::memcpy(M.m, c->mask_clip, sizeof M.m);
        // into the buffer the launches so far have kept zeroed; the launches from here on read it and zero the other one
        c->mask_cur ^= 1;
        uint32_t* cur = c->d_tile_mask + static_cast<size_t>(c->mask_cur) * c->tile_mask_words;
        const uint32_t cells = c->mc_n * c->mc_n * c->mc_n;
        if (static_cast<size_t>(c->tile_mask_words) * sizeof(uint32_t) <= 48u * 1024u) {
            // the mask fits LDS: aggregated per block of cells, only the words that are not zero travel
            const uint32_t nb = (c->mc_n + 7u) / 8u;
            hipLaunchKernelGGL(volym_tile_mask_lds_kernel, dim3(nb * nb * ((c->mc_n + 3u) / 4u)), dim3(256), c->tile_mask_words * sizeof(uint32_t), c->stream, c->d_mc, c->mc_n,
                               c->thr_byte_cull, M, c->mask_margin, c->W, c->H, c->tiles_x * 2u, c->tile_mask_words, cur);
        } else {
            hipLaunchKernelGGL(volym_tile_mask_kernel, dim3((cells + 255u) / 256u), dim3(256), 0, c->stream, c->d_mc, c->mc_n, c->thr_byte_cull, M, c->mask_margin,
                               c->W, c->H, c->tiles_x * 2u, c->tile_mask_words, cur);
        }
        HIPCHK(c, hipGetLastError());
        c->fp.tile_mask = cur;
        c->fp.cull |= CULL_TILE_MASK;
        c->fp.tile_mask_spare = c->d_tile_mask + static_cast<size_t>(c->mask_cur ^ 1) * c->tile_mask_words;
        c->mask_pending = false;
        // costs measured before the mask existed describe tiles that are constant from here on: for the cost feedback this is a
        // new view (a list dealt from the old costs would balance work that is no longer there: 33.2 instead of 32.3 us)
        if (c->view_launches >= 1u) c->view_serial.fetch_add(1, std::memory_order_relaxed);
    }
    c->view_launches++;
    return VOLYM_OK;
}

extern "C" {

int volym_update(volym_ctx* c, const volym_camera_uniforms* cam, const volym_parameter_uniforms* par)
{
    if (!c) return VOLYM_E_INVALID;
    TWIN_FORWARD(c, volym_update(c->twin, cam, par));
    if (!cam || !par) return fail(c, VOLYM_E_INVALID, "volym_update: NULL uniforms");
    if (!c->have_vol || !c->have_imp || !c->have_tf)
        return fail(c, VOLYM_E_STATE, "volym_update: set volume, importances and transfer function first");
    if (c->nx != c->inx || c->ny != c->iny || c->nz != c->inz)
        return fail(c, VOLYM_E_STATE, "volym_update: volume and importances differ in size");
    // The reference loops `while t < exit` with t += step on the GPU; a step that cannot advance t
    // would never terminate there.  Refuse such inputs instead of hanging the device.
    const float step = par->raymarching_step_size;
    if (!(step >= 1.0e-4f && step <= 1.0f)) return fail(c, VOLYM_E_INVALID, "volym_update: raymarching_step_size must be in [1e-4, 1]");
    if (!std::isfinite(par->density_threshold)) return fail(c, VOLYM_E_INVALID, "volym_update: density_threshold is not finite");
    if (par->importance_check_ahead_steps > 4096u) return fail(c, VOLYM_E_INVALID, "volym_update: importance_check_ahead_steps > 4096");
    for (int i = 0; i < 16; ++i)
        if (!std::isfinite((&cam->inverse_view_proj[0][0])[i])) return fail(c, VOLYM_E_INVALID, "volym_update: inverse_view_proj is not finite");
    for (int i = 0; i < 3; ++i)
        if (!(std::fabs(cam->camera_position[i]) <= 64.0f)) return fail(c, VOLYM_E_INVALID, "volym_update: |camera_position| must be <= 64 per axis");

    HIPCHK(c, hipSetDevice(c->device));
    FrameParams& fp = c->fp;
    std::memcpy(fp.ivp, cam->inverse_view_proj, sizeof fp.ivp);
    fp.eye[0] = cam->camera_position[0]; fp.eye[1] = cam->camera_position[1]; fp.eye[2] = cam->camera_position[2];
    fp.thr = par->density_threshold;
    fp.base_step = step;
    fp.min_step = step * 0.25f;          // wgsl:244
    fp.alpha_y = fp.min_step * 100.0f;   // wgsl:314 with current_step_size == min_step_size
    fp.flags = (par->use_cone_importance_check == 1u ? F_CONE : 0u) | (par->use_importance_coloring == 1u ? F_IMP_COLORING : 0u) |
               (par->use_opacity == 1u ? F_OPACITY : 0u) | (par->use_importance_rendering == 1u ? F_IMP_RENDERING : 0u) |
               (par->use_gaussian_smoothing == 1u ? F_GAUSSIAN : 0u) | (c->filter == VOLYM_FILTER_LINEAR ? F_LINEAR : 0u);
    fp.ahead_steps = par->importance_check_ahead_steps;
    fp.W = c->W; fp.H = c->H;
    {
        // make_ray's shared-reciprocal divisions (raymarch_device.h): the host's part of the range argument
        fp.rcp_w = static_cast<float>(1.0 / static_cast<double>(c->W));
        fp.rcp_h = static_cast<float>(1.0 / static_cast<double>(c->H));
        bool vouch = c->W <= 16384u && c->H <= 16384u && !c->setup_ieee;
        for (int i = 0; i < 16; ++i) vouch = vouch && std::fabs(fp.ivp[i]) < 0x1p+60f;
        for (int i = 0; i < 3; ++i) {
            const float n0 = std::fabs(0.0f - fp.eye[i]), n1 = std::fabs(1.0f - fp.eye[i]);
            vouch = vouch && n0 >= 0x1p-40f && n0 <= 0x1p+40f && n1 >= 0x1p-40f && n1 <= 0x1p+40f;
        }
        fp.setup_lo = vouch ? 0x1p-40f : INFINITY;
    }
    fp.nx = c->nx; fp.ny = c->ny; fp.nz = c->nz;
    fp.tiles_x = c->tiles_x; fp.n_tiles = c->n_tiles;
    fp.tf_n = c->tf_n;
    {
        // the look-ahead's reject box (raymarch_device.h ahead_cannot_hit): positions u with clamp(floor(u * n), 0, n - 1) inside the
        // texel AABB of the important voxels, open-ended where the AABB touches the border; 2e-6 covers the rounding of u * n
        const uint32_t dims[3] = {c->inx, c->iny, c->inz};
        for (int a = 0; a < 3; ++a) {
            if (c->imp_box_lo[a] > c->imp_box_hi[a]) { fp.imp_lo[a] = INFINITY; fp.imp_hi[a] = -INFINITY; continue; }
            fp.imp_lo[a] = c->imp_box_lo[a] <= 0 ? -INFINITY : static_cast<float>(c->imp_box_lo[a]) / static_cast<float>(dims[a]) - 2.0e-6f;
            fp.imp_hi[a] = c->imp_box_hi[a] >= static_cast<int>(dims[a]) - 1 ? INFINITY : static_cast<float>(c->imp_box_hi[a] + 1) / static_cast<float>(dims[a]) + 2.0e-6f;
        }
    }
    const float sigma = 1.5f;            // wgsl:255
    for (int i = -2; i <= 2; ++i) {
        const float x = static_cast<float>(i) * 0.005f;
        fp.gauss_w[i + 2] = wgsl_exp(-(x * x) / (2.0f * sigma * sigma));
    }
    std::memcpy(fp.cone_cos, k_cone_cos, sizeof k_cone_cos);
    std::memcpy(fp.cone_sin, k_cone_sin, sizeof k_cone_sin);

    if (c->tables_dirty || c->tables_alpha_y != fp.alpha_y) {
        // New tables travel in stream order behind the frames that read the old ones.  The pinned staging slot must not be
        // rewritten before its copy has run: a ring of TABLE_RING slots, each with the event of its last copy.  Only a
        // caller that changes the step size or the transfer function TABLE_RING times while the device is that many frames
        // behind ever waits here.
        const int slot = c->tables_slot;
        c->tables_slot = (slot + 1) % volym_ctx::TABLE_RING;
        if (hipEventQuery(c->tables_ev[slot]) != hipSuccess) HIPCHK(c, hipEventSynchronize(c->tables_ev[slot]));
        build_tables(c, *c->h_tables[slot], fp.alpha_y);
        c->tables_now = *c->h_tables[slot];
        HIPCHK(c, hipMemcpyAsync(c->d_tables, c->h_tables[slot], sizeof(FrameTables), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipEventRecord(c->tables_ev[slot], c->stream));
        c->tables_dirty = false;
        c->tables_alpha_y = fp.alpha_y;
    }
    uint32_t tb = 256;
    for (int b = 255; b >= 0; --b)
        if (c->tables_now.rho[b] >= fp.thr) tb = static_cast<uint32_t>(b); else break;
    fp.thr_byte = tb;
    // continuous-rho modes (trilinear / smoothed) compare an interpolated value: give its rounding some room
    if (fp.flags & (F_LINEAR | F_GAUSSIAN)) {
        const float cons = fp.thr - std::fabs(fp.thr) * 1.0e-5f - 1.0e-7f;
        uint32_t tc = 256;
        for (int b = 255; b >= 0; --b)
            if (c->tables_now.rho[b] >= cons) tc = static_cast<uint32_t>(b); else break;
        c->thr_byte_cull = tc;
    } else {
        c->thr_byte_cull = tb;
    }
    if (!c->have_frame || std::memcmp(&c->cam_copy, cam, sizeof *cam) != 0 || std::memcmp(&c->par_copy, par, sizeof *par) != 0) {
        c->view_serial.fetch_add(1, std::memory_order_relaxed);   // the lists stay valid (they are scheduling only); the feedback follows the view
        c->hull_dirty = true;                                     // hulls, AABB clip and tile mask belong to the view
    }
    c->cam_copy = *cam;
    c->par_copy = *par;
    c->have_frame = true;
    return VOLYM_OK;
}

}  // extern "C"

template <bool COUNT, bool TRACE = false>
static int launch_march(volym_ctx* c)
{
    int rc = ensure_frame_resources(c);
    if (rc != VOLYM_OK) return rc;
    FrameParams fp = c->fp;
    fp.rank = c->rank; fp.world = c->world; fp.n_local = c->n_local;
    fp.mc_n = c->mc_n;
    fp.xcd_bands = c->xcd_bands;
    if (c->world == 1) fp.flags |= F_RASTER;
    if (c->write_f32 && c->world == 1) fp.flags |= F_WRITE_F32;
    if (c->n_local == 0) return VOLYM_OK;
    uint32_t grid = c->n_local;
    if (fp.xcd_bands) {
        const uint32_t chunks = 8u * fp.xcd_bands;
        const uint32_t per_chunk = (c->n_local + chunks - 1u) / chunks;
        grid = per_chunk * chunks;
    }
    Counters* cnt = (COUNT || (VOLYM_DEV_SWITCHES && (c->fp.dev & 512u))) ? c->d_counters : nullptr;
    uint4* trace = TRACE ? c->d_trace : nullptr;
    if (!COUNT && !TRACE && frame_uses_pool(c, fp.flags)) {
        // the ray pool (raymarch_pool.h): the common instantiation; every other flag set runs variant 2 below
        const uint32_t pgrid = std::min(256u, max_grid(c));         // (the lattice has 256 cells per superblock)
        uint32_t* dbg = c->pool_dbg ? c->d_pool_dbg : nullptr;
        if (c->bricked)
            hipLaunchKernelGGL((volym_raymarch_pool_kernel<true>), dim3(pgrid), dim3(PL_WAVES * 64), 0, c->stream, c->d_vol, c->d_tables, c->d_df, c->d_pool_sync, c->d_shard,
                               c->d_frame, c->d_f32, dbg, fp);
        else
            hipLaunchKernelGGL((volym_raymarch_pool_kernel<false>), dim3(pgrid), dim3(PL_WAVES * 64), 0, c->stream, c->d_vol, c->d_tables, c->d_df, c->d_pool_sync, c->d_shard,
                               c->d_frame, c->d_f32, dbg, fp);
        HIPCHK(c, hipGetLastError());
        c->pool_launched = true;
        return VOLYM_OK;
    }
    if (c->kernel_variant >= 2) {
        const bool plain = !COUNT && !TRACE;
        if (plain) feedback_poll(c);                      // adopt a list the feedback thread has finished
        const WorkList& wl = c->lists[c->cur];
        const uint32_t n_items = static_cast<uint32_t>(wl.entries.size());
        if (n_items == 0) return VOLYM_OK;
        // capture this launch's costs?  Only one capture is in flight; a list dealt from costs measured on this very view, on
        // whole 8x8 entries, is final
        const bool capture = plain && c->feedback && !c->feedback_frozen && c->fb_state.load(std::memory_order_acquire) == volym_ctx::FB_IDLE && (wl.view_serial != c->view_serial.load(std::memory_order_relaxed) || !wl.final_for_view);
        uint16_t* cost_out = capture ? c->d_cost : nullptr;
        const bool table = !(fp.flags & (F_LINEAR | F_GAUSSIAN));
        // IMP = false: opacity on and no importance colouring (the common cases), without (IR = false) or with (IR = true)
        // importance rendering; the instrumented launch always takes the general form
        const bool special = !COUNT && !(fp.flags & F_IMP_COLORING) && (fp.flags & F_OPACITY);
        const bool no_imp = special && !(fp.flags & F_IMP_RENDERING);
        const bool ir = special && (fp.flags & F_IMP_RENDERING);
        // Waves per workgroup of the instantiation (raymarch_pq.h WAVES).  Only the common instantiation fits the 128 VGPRs of a
        // 16-wave workgroup; the others either run 16 waves and keep 64-100 bytes per lane in scratch, or 12 waves without
        // scratch.  Measured (profiles/r02_kernel_resources.txt): while the volume is cache resident the fourth wave per SIMD
        // is worth more than the spills cost (importance 68 vs 71 us, smoothing 134 vs 148, trilinear 113 vs 127); on the
        // bricked 1024^3 volume with its label map (BASELINE configs[4] on one GPU) the 12-wave form wins (235 vs 259 us).
        const bool wide12 = c->wide_waves == 12 || (c->wide_waves == 0 && c->bricked && ir);
        const uint32_t waves = ((table && no_imp) || !wide12) ? PQ_WAVES : PQ_WAVES_WIDE;
        const uint32_t want = (n_items + waves - 1) / waves;
        const uint32_t pgrid = wl.grid ? wl.grid : std::max(1u, std::min(want, max_grid(c)));
#define VOLYM_PQ_LAUNCH_J(T, KS, I, B, R, WV, J)                                                                                 \
    hipLaunchKernelGGL((volym_raymarch_pq_kernel<T, COUNT && I, TRACE, KS, I, B, R, WV, J>), dim3(pgrid), dim3(WV * 64), 0, c->stream, c->d_vol,  \
                       c->d_imp, c->d_tables, c->d_df, reinterpret_cast<const uint2*>(c->d_list[c->cur]), n_items, cost_out, c->d_shard, c->d_frame, c->d_f32, cnt, trace, fp)
#define VOLYM_PQ_LAUNCH(T, KS, I, B, R, WV) VOLYM_PQ_LAUNCH_J(T, KS, I, B, R, WV, 0)
#if VOLYM_DEV_SWITCHES
#define VOLYM_PQ_LAUNCH_W(T, KS, I, B, R) do { if (wide12) VOLYM_PQ_LAUNCH(T, KS, I, B, R, PQ_WAVES_WIDE); else VOLYM_PQ_LAUNCH(T, KS, I, B, R, PQ_WAVES); } while (0)
#else
#define VOLYM_PQ_LAUNCH_W(T, KS, I, B, R) VOLYM_PQ_LAUNCH(T, KS, I, B, R, PQ_WAVES)
#endif
        // the cone look-ahead of the importance-rendering instantiation: its walks as jobs shared by the workgroup (raymarch_pq.h CJ)
        const bool cone_jobs = ir && (fp.flags & F_CONE) != 0u && !TRACE;
        // (dev, option 121: the straight look-ahead through the same ring, one lane per record -- measured: bonsai 79.6 us against 63.7,
        // teapot 92.4 against 79.7: a chain of 15 probes is too little work per record to pay for the ring; not in the product library)
        const bool straight_jobs = VOLYM_DEV_SWITCHES && ir && !(fp.flags & F_CONE) && !TRACE && c->straight_jobs;
        (void)straight_jobs;
        if (c->bricked) {
#if VOLYM_DEV_SWITCHES
            // (dev, option 122: north_star's LDS-staged bricks, measured in profiles/r03_lds_bricks_ab.txt; not in the product library)
            if (table && no_imp && c->lds_bricks && !COUNT && !TRACE)
                hipLaunchKernelGGL((volym_raymarch_pq_kernel<true, false, false, 4, false, true, false, PQ_WAVES, 0, true>), dim3(pgrid), dim3(PQ_WAVES * 64), 0, c->stream, c->d_vol,
                                   c->d_imp, c->d_tables, c->d_df, reinterpret_cast<const uint2*>(c->d_list[c->cur]), n_items, cost_out, c->d_shard, c->d_frame, c->d_f32, cnt, trace, fp);
            else
#endif
            if (table && no_imp) VOLYM_PQ_LAUNCH(true, 4, false, true, false, PQ_WAVES);
            else if (table && ir && cone_jobs) VOLYM_PQ_LAUNCH_J(true, 4, false, true, true, PQ_WAVES, 1);
#if VOLYM_DEV_SWITCHES
            else if (table && ir && straight_jobs) VOLYM_PQ_LAUNCH_J(true, 4, false, true, true, PQ_WAVES, 2);
#endif
            else if (table && ir) { if (wide12) VOLYM_PQ_LAUNCH(true, 4, false, true, true, PQ_WAVES_WIDE); else VOLYM_PQ_LAUNCH(true, 4, false, true, true, PQ_WAVES); }
            else if (table) VOLYM_PQ_LAUNCH_W(true, 4, true, true, false);
            else VOLYM_PQ_LAUNCH_W(false, 1, true, true, false);
        } else {
            if (table && no_imp) VOLYM_PQ_LAUNCH(true, 4, false, false, false, PQ_WAVES);
            else if (table && ir && cone_jobs) VOLYM_PQ_LAUNCH_J(true, 4, false, false, true, PQ_WAVES, 1);
#if VOLYM_DEV_SWITCHES
            else if (table && ir && straight_jobs) VOLYM_PQ_LAUNCH_J(true, 4, false, false, true, PQ_WAVES, 2);
#endif
            else if (table && ir) VOLYM_PQ_LAUNCH_W(true, 4, false, false, true);
            else if (table) VOLYM_PQ_LAUNCH_W(true, 4, true, false, false);
            else VOLYM_PQ_LAUNCH_W(false, 1, true, false, false);
        }
#undef VOLYM_PQ_LAUNCH_W
#undef VOLYM_PQ_LAUNCH
#undef VOLYM_PQ_LAUNCH_J
        HIPCHK(c, hipGetLastError());
        if (capture) {
            // costs -> pinned host memory on the copy stream, behind this launch; the feedback thread takes it from there
            HIPCHK(c, hipEventRecord(c->ev_march, c->stream));
            HIPCHK(c, hipStreamWaitEvent(c->copy_stream, c->ev_march, 0));
            const size_t cost_bytes = static_cast<size_t>((n_items + 1u) & ~1u) * sizeof(uint16_t) + static_cast<size_t>(pgrid) * (waves + 1u) * sizeof(uint32_t);
            HIPCHK(c, hipMemcpyAsync(c->h_cost_pinned, c->d_cost, cost_bytes, hipMemcpyDeviceToHost, c->copy_stream));
            HIPCHK(c, hipEventRecord(c->ev_cost, c->copy_stream));
            volym_ctx::FbJob& job = c->fb_job;
            job.list = c->cur;
            job.n_entries = n_items;
            job.view_serial = c->view_serial.load(std::memory_order_relaxed);
            job.captured_has_dp = wl.has_dp;
            job.continuous = (fp.flags & (F_LINEAR | F_GAUSSIAN)) != 0u;
            job.plain = table && no_imp;                  // the common instantiation: the split threshold's floor was measured for it
            job.max_grid = max_grid(c);
            job.waves = waves;
            job.dp_min_cost = c->dp_min_cost;
            job.dp_share_pct = c->dp_share_pct;
            job.fill_cost = c->fill_cost;
            job.super_fill = c->super_fill;
            job.only_quarters = c->dev_only_quarters;
            job.dilate = c->cost_dilate;
            job.grid = pgrid;
            job.dev_drop_tenths = c->dev_drop_tenths;
            job.dp_floor = c->dp_floor;
            job.trim_rounds = c->trim_rounds;
            job.t_us[0] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
            for (int i = 0; i < 3; ++i) job.prio_tenths[i] = c->prio_tenths[i];
            {
                std::lock_guard<std::mutex> lk(c->fb_mu);
                c->fb_state.store(volym_ctx::FB_CAPTURED, std::memory_order_release);
            }
            c->fb_cv.notify_all();
        }
        return VOLYM_OK;
    }
#define VOLYM_DIRECT_LAUNCH(V, B)                                                                                                \
    hipLaunchKernelGGL((volym_raymarch_kernel<V, COUNT, TRACE, B>), dim3(grid), dim3(256), 0, c->stream, c->d_vol, c->d_imp, c->d_tables, \
                       c->d_df, c->d_shard, c->d_frame, c->d_f32, cnt, trace, fp)
    if (c->kernel_variant >= 1) { if (c->bricked) VOLYM_DIRECT_LAUNCH(1, true); else VOLYM_DIRECT_LAUNCH(1, false); }
    else { if (c->bricked) VOLYM_DIRECT_LAUNCH(0, true); else VOLYM_DIRECT_LAUNCH(0, false); }
#undef VOLYM_DIRECT_LAUNCH
    HIPCHK(c, hipGetLastError());
    return VOLYM_OK;
}

int volym::ctx_launch_march(volym_ctx* c) { return launch_march<false>(c); }

extern "C" {

int volym_compute_pass(volym_ctx* c)
{
    if (!c) return VOLYM_E_INVALID;
    if (!c->have_frame) return fail(c, VOLYM_E_STATE, "volym_compute_pass: call volym_update first");
    HIPCHK(c, hipSetDevice(c->device));
    if (c->twin) {                                  // frames in flight: every other frame goes to the twin's stream and frame buffer
        volym_ctx* const t = (c->flight_parity++ & 1u) ? c->twin : c;
        c->last = t;
        if (t != c) { TWIN_FORWARD(c, launch_march<false>(t)); return VOLYM_OK; }
    }
    return launch_march<false>(c);
}

// Back-pressure of a frame loop: the reference's loop cannot run ahead of the device by more than its swap chain holds
// (surface.get_current_texture() blocks, src/event_loop.rs:114).  Call once per frame after volym_compute_pass: marks the
// work enqueued so far and waits until the mark made `max_in_flight` calls ago has been reached.
int volym_throttle(volym_ctx* c, uint32_t max_in_flight)
{
    if (!c) return VOLYM_E_INVALID;
    if (max_in_flight == 0 || max_in_flight >= volym_ctx::THROTTLE_RING) return fail(c, VOLYM_E_INVALID, "volym_throttle: max_in_flight must be 1..8");
    HIPCHK(c, hipSetDevice(c->device));
    // The ring holds one slot more than the deepest wait (9 slots, max_in_flight <= 8): the mark made max_in_flight calls ago
    // is never the slot recorded by this call, and the slot recorded here was last recorded 9 calls ago -- an earlier call has
    // already waited for a later mark than that one (marks complete in stream order).
    const uint32_t slot = c->throttle_head % volym_ctx::THROTTLE_RING;
    // a pacing mark: nothing is read on the strength of it, so no system-scope fence (cache write-back and invalidation) between
    // two frames -- with the default event the marches of a paced loop ran 6 us apart (scripts/turntable_trace.py)
    if (!c->throttle_ev[slot] && hipEventCreateWithFlags(&c->throttle_ev[slot], hipEventDisableTiming | hipEventDisableSystemFence) != hipSuccess) {
        (void)hipGetLastError();
        c->throttle_ev[slot] = nullptr;
        HIPCHK(c, hipEventCreateWithFlags(&c->throttle_ev[slot], hipEventDisableTiming));
    }
    HIPCHK(c, hipEventRecord(c->throttle_ev[slot], (c->twin && c->last) ? c->last->stream : c->stream));   // the frame just enqueued
    c->throttle_head++;
    if (c->throttle_head > max_in_flight) {
        const uint32_t old = (c->throttle_head - 1u - max_in_flight) % volym_ctx::THROTTLE_RING;
        if (c->throttle_ev[old]) HIPCHK(c, hipEventSynchronize(c->throttle_ev[old]));
    }
    return VOLYM_OK;
}

// The ray-pool kernel (variant 3) bounds every wait it contains and reports a wait that ran out (a bug, never an input) in
// d_pool_sync[2]; the blocking calls look at it, so that a broken frame is an error and not a picture.  Stream is idle here.
static int check_pool_error(volym_ctx* c)
{
    if (!c->pool_launched) return VOLYM_OK;
    uint32_t bits = 0;
    HIPCHK(c, hipMemcpy(&bits, c->d_pool_sync + 2, sizeof bits, hipMemcpyDeviceToHost));
    if (bits != 0u) {
        (void)hipMemset(c->d_pool_sync, 0, 4 * sizeof(uint32_t));
        return fail(c, VOLYM_E_HIP, "ray-pool kernel gave up waiting (error bits " + std::to_string(bits) + "): the frame is incomplete");
    }
    return VOLYM_OK;
}

int volym_sync(volym_ctx* c)
{
    if (!c) return VOLYM_E_INVALID;
    TWIN_FORWARD(c, volym_sync(c->twin));
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return check_pool_error(c);
}

int volym_read_rgba8(volym_ctx* c, uint8_t* out)
{
    if (!c || !out) return VOLYM_E_INVALID;
    if (c->twin && c->last == c->twin) { TWIN_FORWARD(c, volym_read_rgba8(c->twin, out)); return VOLYM_OK; }
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(out, c->d_frame, static_cast<size_t>(c->W) * c->H * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return check_pool_error(c);
}

int volym_read_rgba32f(volym_ctx* c, float* out)
{
    if (!c || !out) return VOLYM_E_INVALID;
    if (c->twin && c->last == c->twin) { TWIN_FORWARD(c, volym_read_rgba32f(c->twin, out)); return VOLYM_OK; }
    if (!c->write_f32 || !c->d_f32 || c->world != 1)
        return fail(c, VOLYM_E_STATE, "volym_read_rgba32f: needs VOLYM_OPT_WRITE_F32 = 1, world == 1 and a rendered frame");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(out, c->d_f32, static_cast<size_t>(c->W) * c->H * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return VOLYM_OK;
}

// ---- blit: the step after the path (src/render_pipeline.rs:88-130, shaders/render.wgsl:39-43) -------------------------
int volym_blit(volym_ctx* c, void* target_rgba8, uint32_t out_w, uint32_t out_h)
{
    if (!c) return VOLYM_E_INVALID;
    if (c->twin) {
        c->last_blit = (c->last == c->twin) ? c->twin : c;
        if (c->last_blit == c->twin) { TWIN_FORWARD(c, volym_blit(c->twin, target_rgba8, out_w, out_h)); return VOLYM_OK; }
    }
    if (out_w == 0 || out_h == 0 || out_w > 32768 || out_h > 32768) return fail(c, VOLYM_E_INVALID, "volym_blit: target must be 1..32768 in each dimension");
    HIPCHK(c, hipSetDevice(c->device));
    uint32_t* dst = static_cast<uint32_t*>(target_rgba8);
    if (!dst) {
        // our own target: sized on first use / on a size change (a set-up step: this is the one blocking path of the call)
        const size_t need = static_cast<size_t>(out_w) * out_h * 4;
        if (need > c->blit_bytes) {
            HIPCHK(c, hipStreamSynchronize(c->stream));
            if (c->d_blit) { HIPCHK(c, hipFree(c->d_blit)); c->d_blit = nullptr; c->blit_bytes = 0; }
            hipError_t e = hipMalloc(&c->d_blit, need);
            if (e != hipSuccess) return fail(c, VOLYM_E_NOMEM, std::string("hipMalloc(blit target): ") + hipGetErrorString(e));
            c->blit_bytes = need;
        }
        dst = c->d_blit;
        c->blit_w = out_w; c->blit_h = out_h;
    }
    hipLaunchKernelGGL(volym_blit_kernel, dim3((out_w + 63u) / 64u, (out_h + 3u) / 4u), dim3(64, 4), 0, c->stream, c->d_frame, c->W, c->H, dst, out_w, out_h);
    HIPCHK(c, hipGetLastError());
    return VOLYM_OK;
}

int volym_read_blit(volym_ctx* c, uint8_t* out)
{
    if (!c || !out) return VOLYM_E_INVALID;
    if (c->twin && c->last_blit == c->twin) { TWIN_FORWARD(c, volym_read_blit(c->twin, out)); return VOLYM_OK; }
    if (!c->d_blit || c->blit_w == 0) return fail(c, VOLYM_E_STATE, "volym_read_blit: no volym_blit into the context's own target yet");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(out, c->d_blit, static_cast<size_t>(c->blit_w) * c->blit_h * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return VOLYM_OK;
}

uint32_t volym_local_tiles(const volym_ctx* c) { return c ? c->n_local : 0u; }
size_t volym_shard_bytes(const volym_ctx* c) { return c ? static_cast<size_t>(c->shard_tiles) * 1024u : 0u; }
void* volym_shard_device_ptr(volym_ctx* c) { return c ? c->d_shard : nullptr; }
void* volym_frame_device_ptr(volym_ctx* c) { return c ? c->d_frame : nullptr; }

int volym_bind_output(volym_ctx* c, void* shard_rgba8, void* frame_rgba8)
{
    if (!c) return VOLYM_E_INVALID;
    // takes effect for launches enqueued after this call; earlier launches keep their pointers
    c->d_shard = shard_rgba8 ? static_cast<uint32_t*>(shard_rgba8) : c->d_shard_own;
    c->d_frame = frame_rgba8 ? static_cast<uint32_t*>(frame_rgba8) : c->d_frame_own;
    return VOLYM_OK;
}

int volym_read_shard(volym_ctx* c, uint8_t* out)
{
    TWIN_REFUSE(c, "volym_read_shard");
    if (!c || !out) return VOLYM_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    // the padding tile of a short shard is never written by the kernel: define it
    const size_t used = static_cast<size_t>(c->n_local) * 1024u, total = volym_shard_bytes(c);
    HIPCHK(c, hipMemcpyAsync(out, c->d_shard, used, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (total > used) std::memset(out + used, 0, total - used);
    return VOLYM_OK;
}

size_t volym_packed_shard_bytes(const volym_ctx* c, uint32_t tiles)
{
    if (!c) return 0u;
    return pack_header_bytes(c->shard_tiles) + static_cast<size_t>(std::min(tiles, c->shard_tiles)) * 1024u;
}

int volym_pack_shard(volym_ctx* c, void* packed, size_t capacity_bytes)
{
    TWIN_REFUSE(c, "volym_pack_shard");
    if (!c || !packed) return VOLYM_E_INVALID;
    const size_t header = pack_header_bytes(c->shard_tiles);
    if (capacity_bytes < header) return fail(c, VOLYM_E_INVALID, "volym_pack_shard: the buffer does not even hold the header (volym_packed_shard_bytes)");
    HIPCHK(c, hipSetDevice(c->device));
    if (c->n_local == 0) return VOLYM_OK;
    const uint32_t max_slots = static_cast<uint32_t>(std::min<size_t>((capacity_bytes - header) / 1024u, c->shard_tiles));
    hipLaunchKernelGGL(volym_pack_shard_kernel, dim3(c->n_local), dim3(64), 0, c->stream, c->d_shard, static_cast<uint8_t*>(packed), c->n_local,
                       c->shard_tiles, max_slots, c->d_pack_counters, c->pack_parity);
    HIPCHK(c, hipGetLastError());
    c->pack_parity ^= 1u;
    return VOLYM_OK;
}

int volym_packed_tiles(volym_ctx* c, uint32_t* tiles_used, uint32_t* overflowed)
{
    TWIN_REFUSE(c, "volym_packed_tiles");
    if (!c || !tiles_used) return VOLYM_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    uint32_t h[4] = {0, 0, 0, 0};
    HIPCHK(c, hipMemcpyAsync(h, c->d_pack_counters, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *tiles_used = h[c->pack_parity ^ 1u];        // the counter the last launch used
    if (overflowed) *overflowed = h[2];
    return VOLYM_OK;
}

int volym_assemble_packed(volym_ctx* c, const void* gathered, size_t stride_bytes)
{
    TWIN_REFUSE(c, "volym_assemble_packed");
    if (!c || !gathered) return VOLYM_E_INVALID;
    if (stride_bytes < pack_header_bytes(c->shard_tiles)) return fail(c, VOLYM_E_INVALID, "volym_assemble_packed: stride smaller than the header");
    HIPCHK(c, hipSetDevice(c->device));
    hipLaunchKernelGGL(volym_assemble_packed_kernel, dim3(c->n_tiles), dim3(256), 0, c->stream, static_cast<const uint8_t*>(gathered), stride_bytes,
                       c->d_frame, c->W, c->H, c->tiles_x, c->n_tiles, c->world, c->shard_tiles);
    HIPCHK(c, hipGetLastError());
    return VOLYM_OK;
}

int volym_assemble(volym_ctx* c, const void* gathered)
{
    TWIN_REFUSE(c, "volym_assemble");
    if (!c || !gathered) return VOLYM_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    hipLaunchKernelGGL(volym_assemble_kernel, dim3(c->n_tiles), dim3(256), 0, c->stream, static_cast<const uint32_t*>(gathered),
                       c->d_frame, c->W, c->H, c->tiles_x, c->n_tiles, c->world, c->shard_tiles);
    HIPCHK(c, hipGetLastError());
    return VOLYM_OK;
}

int volym_assemble_host(volym_ctx* c, const uint8_t* gathered_host)
{
    TWIN_REFUSE(c, "volym_assemble_host");
    if (!c || !gathered_host) return VOLYM_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    const size_t bytes = volym_shard_bytes(c) * c->world;
    if (c->gather_tmp_bytes < bytes) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->d_gather_tmp) { HIPCHK(c, hipFree(c->d_gather_tmp)); c->d_gather_tmp = nullptr; c->gather_tmp_bytes = 0; }
        hipError_t e = hipMalloc(&c->d_gather_tmp, bytes);
        if (e != hipSuccess) return fail(c, VOLYM_E_NOMEM, std::string("hipMalloc(gather): ") + hipGetErrorString(e));
        c->gather_tmp_bytes = bytes;
    }
    HIPCHK(c, hipMemcpyAsync(c->d_gather_tmp, gathered_host, bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return volym_assemble(c, c->d_gather_tmp);
}

int volym_stats_pass(volym_ctx* c, volym_stats* out)
{
    if (!c || !out) return VOLYM_E_INVALID;
    if (!c->have_frame) return fail(c, VOLYM_E_STATE, "volym_stats_pass: call volym_update first");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemsetAsync(c->d_counters, 0, sizeof(Counters), c->stream));
    int rc = launch_march<true>(c);
    if (rc != VOLYM_OK) return rc;
    Counters h;
    HIPCHK(c, hipMemcpyAsync(&h, c->d_counters, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    out->n_vol = h.n_vol; out->n_imp = h.n_imp; out->n_steps = h.n_steps; out->n_dense = h.n_dense; out->n_hit = h.n_hit;
    // every pixel of an owned tile that lies inside the frame launches a ray (wgsl:217-219)
    uint64_t rays = 0;
    for (uint32_t k = c->rank; k < c->n_tiles; k += c->world) {
        const uint32_t tx = k % c->tiles_x, ty = k / c->tiles_x;
        const uint32_t w = std::min(16u, c->W - tx * 16u), h2 = std::min(16u, c->H - ty * 16u);
        rays += static_cast<uint64_t>(w) * h2;
    }
    out->n_rays = rays;
    return VOLYM_OK;
}

// Both forms of the ray set-up (raymarch_device.h make_ray) for every pixel of the current frame, compared bit for bit on the
// device.  out[0]: rays whose shared-reciprocal set-up (as the march kernels run it, fallback included) differs from the plain
// divisions in any bit of direction / entry / exit / hit; out[1]: rays of waves that took the fallback; out[2]: rays.
int volym_selftest_ray_setup(volym_ctx* c, unsigned long long out[3])
{
    if (!c || !out) return VOLYM_E_INVALID;
    if (!c->have_frame) return fail(c, VOLYM_E_STATE, "volym_selftest_ray_setup: call volym_update first");
    HIPCHK(c, hipSetDevice(c->device));
    static_assert(sizeof(Counters) >= 3 * sizeof(unsigned long long), "counters");
    HIPCHK(c, hipMemsetAsync(c->d_counters, 0, sizeof(Counters), c->stream));
    const dim3 grid((c->W + 63u) / 64u, (c->H + 3u) / 4u);
    volym_ray_setup_selftest_kernel<<<grid, 256, 0, c->stream>>>(c->fp, reinterpret_cast<unsigned long long*>(c->d_counters));
    HIPCHK(c, hipGetLastError());
    Counters h;
    HIPCHK(c, hipMemcpyAsync(&h, c->d_counters, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    out[0] = h.n_vol; out[1] = h.n_imp; out[2] = h.n_steps;
    return VOLYM_OK;
}

int volym_time_batch(volym_ctx* c, uint32_t n, float* ms_total)
{
    if (!c || !ms_total || n == 0 || n > 1000000) return VOLYM_E_INVALID;
    if (!c->have_frame) return fail(c, VOLYM_E_STATE, "volym_time_batch: call volym_update first");
    HIPCHK(c, hipSetDevice(c->device));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { if (e0) (void)hipEventDestroy(e0); return fail(c, VOLYM_E_HIP, "hipEventCreate failed"); }
    int rc = ensure_frame_resources(c);
    if (rc == VOLYM_OK) {
        (void)hipEventRecord(e0, c->stream);
        for (uint32_t i = 0; i < n && rc == VOLYM_OK; ++i) rc = launch_march<false>(c);
        (void)hipEventRecord(e1, c->stream);
        if (hipStreamSynchronize(c->stream) != hipSuccess && rc == VOLYM_OK) rc = fail(c, VOLYM_E_HIP, "hipStreamSynchronize failed");
        if (rc == VOLYM_OK && hipEventElapsedTime(ms_total, e0, e1) != hipSuccess) rc = fail(c, VOLYM_E_HIP, "hipEventElapsedTime failed");
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return rc;
}

int volym_time_passes(volym_ctx* c, uint32_t n, float* ms_each)
{
    if (!c || !ms_each || n == 0 || n > 100000) return VOLYM_E_INVALID;
    if (!c->have_frame) return fail(c, VOLYM_E_STATE, "volym_time_passes: call volym_update first");
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<hipEvent_t> ev(n + 1, nullptr);
    int rc = VOLYM_OK;
    for (uint32_t i = 0; i <= n && rc == VOLYM_OK; ++i)
        if (hipEventCreate(&ev[i]) != hipSuccess) rc = fail(c, VOLYM_E_HIP, "hipEventCreate failed");
    if (rc == VOLYM_OK) rc = ensure_frame_resources(c);
    if (rc == VOLYM_OK) {
        (void)hipEventRecord(ev[0], c->stream);
        for (uint32_t i = 0; i < n && rc == VOLYM_OK; ++i) {
            rc = launch_march<false>(c);
            if (hipEventRecord(ev[i + 1], c->stream) != hipSuccess && rc == VOLYM_OK) rc = fail(c, VOLYM_E_HIP, "hipEventRecord failed");
        }
        if (hipStreamSynchronize(c->stream) != hipSuccess && rc == VOLYM_OK) rc = fail(c, VOLYM_E_HIP, "hipStreamSynchronize failed");
        if (rc == VOLYM_OK)
            for (uint32_t i = 0; i < n; ++i)
                if (hipEventElapsedTime(&ms_each[i], ev[i], ev[i + 1]) != hipSuccess) { rc = fail(c, VOLYM_E_HIP, "hipEventElapsedTime failed"); break; }
    }
    for (auto e : ev) if (e) (void)hipEventDestroy(e);
    return rc;
}

#if VOLYM_DEV_SWITCHES
// ---- development build only (make DEV=1): not declared in the public headers, not in the product library ------------
// per-item costs (uint16) as the feedback thread last saw them; out needs 4 * n_local entries
// development: per-wave timeline of the NEXT ray-pool launches (on != 0) / read the last one back (blocking)
int volym_dev_pool_timeline(volym_ctx* c, int on, uint32_t* out, uint32_t max_words)
{
    if (!c) return VOLYM_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const uint32_t words = std::min(max_words, static_cast<uint32_t>(max_grid(c)) * PL_WAVES * 24u);
    if (out && words) HIPCHK(c, hipMemcpy(out, c->d_pool_dbg, static_cast<size_t>(words) * sizeof(uint32_t), hipMemcpyDeviceToHost));
    c->pool_dbg = on != 0;
    return static_cast<int>(words);
}

// development: the raw counters (fp.dev & 512: the cone-job debug counts of the plain launches since the last reset); reset != 0 zeroes them
int volym_dev_counters(volym_ctx* c, unsigned long long out[5], int reset)
{
    if (!c) return VOLYM_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (out) HIPCHK(c, hipMemcpy(out, c->d_counters, 5 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    if (reset) HIPCHK(c, hipMemset(c->d_counters, 0, sizeof(Counters)));
    return VOLYM_OK;
}

int volym_dev_read_costs(volym_ctx* c, uint16_t* out, uint32_t max_items)
{
    if (!c || !out) return VOLYM_E_INVALID;
    feedback_quiesce(c);
    const uint32_t n = c->n_local * 4u;
    if (max_items < n || c->item_cost.size() < n) return VOLYM_E_INVALID;
    std::memcpy(out, c->item_cost.data(), n * sizeof(uint16_t));
    return static_cast<int>(n);
}

// wall-clock stamps (us) of the last finished feedback job: capture enqueued, worker woke, costs arrived, costs mapped to
// items, list dealt, list uploaded
int volym_dev_feedback_timing(volym_ctx* c, double out[6])
{
    if (!c || !out) return VOLYM_E_INVALID;
    feedback_quiesce(c);
    for (int i = 0; i < 6; ++i) out[i] = c->fb_job.t_us[i];
    return VOLYM_OK;
}

// the work list the next launch will read; returns the number of entries
int volym_dev_read_order(volym_ctx* c, uint32_t* out, uint32_t max_items)
{
    if (!c || !out) return VOLYM_E_INVALID;
    feedback_quiesce(c);
    const WorkList& wl = c->lists[c->cur];
    if (max_items < wl.entries.size()) return VOLYM_E_INVALID;
    std::memcpy(out, wl.entries.data(), wl.entries.size() * sizeof(uint32_t));
    return static_cast<int>(wl.entries.size());
}

// One instrumented launch that records, per wave, {start tick, duration ticks (100 MHz), loop iterations, ...} (two
// uint4 per wave, scripts/wave_trace.py); returns the number of records or a negative error.
int volym_dev_wave_trace(volym_ctx* c, uint32_t* out, uint32_t max_records)
{
    if (!c || !out) return VOLYM_E_INVALID;
    if (!c->have_frame) return fail(c, VOLYM_E_STATE, "volym_dev_wave_trace: call volym_update first");
    HIPCHK(c, hipSetDevice(c->device));
    // a lone launch on an idle GPU runs at idle clocks: trace the 31st of 31 back-to-back passes
    int rc = VOLYM_OK;
    for (int i = 0; i < 30 && rc == VOLYM_OK; ++i) rc = launch_march<false, false>(c);
    if (rc != VOLYM_OK) return rc;
    feedback_quiesce(c);
    // records: two per wave of the grid that is really launched (variant 2: the list's grid; variants 0/1: 4 waves per tile)
    const WorkList& wl = c->lists[c->cur];
    const uint32_t n_items = static_cast<uint32_t>(wl.entries.size());
    const uint32_t pgrid = wl.grid ? wl.grid : std::max(1u, std::min((n_items + PQ_WAVES - 1) / PQ_WAVES, max_grid(c)));
    const uint32_t waves = c->kernel_variant >= 2 ? pgrid * PQ_WAVES : (c->n_local + 64u * 8u) * 4u;   // PQ_WAVES >= every instantiation's WAVES
    const uint32_t records = waves * 2u;
    if (max_records < records) return fail(c, VOLYM_E_INVALID, "volym_dev_wave_trace: buffer too small");
    HIPCHK(c, hipMalloc(&c->d_trace, static_cast<size_t>(records) * sizeof(uint4)));
    HIPCHK(c, hipMemsetAsync(c->d_trace, 0, static_cast<size_t>(records) * sizeof(uint4), c->stream));
    rc = launch_march<false, true>(c);
    if (rc == VOLYM_OK) {
        hipError_t e = hipMemcpyAsync(out, c->d_trace, static_cast<size_t>(records) * sizeof(uint4), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) rc = fail(c, VOLYM_E_HIP, hipGetErrorString(e));
    }
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(c->d_trace);
    c->d_trace = nullptr;
    return rc == VOLYM_OK ? static_cast<int>(records) : rc;
}
#endif

}  // extern "C"

#include "mgpu.inc"
