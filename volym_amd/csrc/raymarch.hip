// libvolym_hip.so: context + C ABI (include/volym_hip.h) over the gfx950 kernels.
// Replaces the reference's gpu_context.rs / gpu_resources/* / demos/pipeline.rs for the
// ray-march path; citations are file:line under /root/reference/.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <queue>
#include <vector>

#include "../../include/volym_hip.h"
#include "raymarch_kernels.h"
#include "raymarch_pq.h"

#include <algorithm>

using namespace volym;

static_assert(sizeof(volym_camera_uniforms) == 208, "CameraUniforms is 208 bytes (src/gpu_resources/camera.rs:56-64)");
static_assert(sizeof(volym_parameter_uniforms) == 32, "ParameterUniforms is 32 bytes (src/gpu_resources/parameters.rs:55-66)");

// cos/sin of (s/8) * 2 * 3.14159 for s = 0..7 (wgsl:99-103), f32
static const float k_cone_cos[8] = {0x1p+0f, 0x1.6a09f6p-1f, 0x1.54442ep-20f, -0x1.6a09bap-1f,
                                    -0x1p+0f, -0x1.6a0a32p-1f, -0x1.fe6644p-19f, 0x1.6a097ep-1f};
static const float k_cone_sin[8] = {0x0p+0f, 0x1.6a09d8p-1f, 0x1p+0f, 0x1.6a0a14p-1f,
                                    0x1.54442ep-19f, -0x1.6a099cp-1f, -0x1p+0f, -0x1.6a0a5p-1f};

struct volym_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t own_stream = nullptr;
    uint32_t W = 0, H = 0, tiles_x = 0, tiles_y = 0, n_tiles = 0;
    uint32_t rank = 0, world = 1, n_local = 0, shard_tiles = 0;

    uint8_t* d_vol = nullptr;
    uint8_t* d_imp = nullptr;
    uint32_t nx = 0, ny = 0, nz = 0;
    uint32_t inx = 0, iny = 0, inz = 0;
    int filter = VOLYM_FILTER_NEAREST;
    uint8_t lut[256 * 4] = {};
    uint32_t tf_n = 0;
    bool have_vol = false, have_imp = false, have_tf = false, have_frame = false;

    FrameTables* d_tables = nullptr;
    FrameTables h_tables;
    bool tables_dirty = true;
    float tables_alpha_y = -1.0f;

    uint8_t* d_mc = nullptr;   // per-macro-cell density maxima
    uint8_t* d_df = nullptr;   // packed 4-bit distance field for (d_mc, thr_byte)
    uint32_t mc_n = 32, mc_built_n = 0;
    uint32_t df_thr_byte = 0xffffffffu;
    uint32_t thr_byte_cull = 256;   // threshold byte the macro-cell occupancy is built for (conservative in continuous modes)
    bool mc_dirty = true;

    uint32_t* d_shard_own = nullptr;
    uint32_t* d_frame_own = nullptr;
    uint32_t* d_shard = nullptr;
    uint32_t* d_frame = nullptr;
    float4* d_f32 = nullptr;
    uint8_t* d_gather_tmp = nullptr;
    uint32_t* d_pack_counters = nullptr;   // volym_pack_shard: slot counters of even / odd launches, overflow flag
    uint32_t pack_parity = 0;
    size_t gather_tmp_bytes = 0;
    Counters* d_counters = nullptr;
    uint4* d_trace = nullptr;   // development aid, see volym_dev_wave_trace
    uint32_t* d_order = nullptr;   // variant 2: this rank's 8x8 wave tiles, centre-first (or by measured cost)
    std::vector<uint32_t> h_order;   // the geometric (centre-first) list, kept for re-sorting
    uint16_t* d_cost = nullptr;    // per item: cost the last plain launch measured (loop iterations, flushes)
    uint32_t frames_since_change = 0;
    bool order_by_cost = false;    // d_order currently reflects measured cost
    bool feedback = true;
    bool super_fill = true;
    uint32_t prio_tenths[3] = {3, 6, 10};   // cost / fair share (tenths) from which an item runs at issue priority 1, 2, 3 (first 0: off)
    bool dev_only_quarters = false;
    bool bricked = false;          // layout of d_vol / d_imp (raymarch_device.h "Volume layout")
    uint64_t brick_from_bytes = 64ull << 20;    // volumes above this many bytes are bricked (512^3 at 1080p: 54.3 -> 39.7 us; 256^3: 36.6 -> 38.1)
    int layout_choice = -1;        // -1: by size, 0: linear, 1: bricked (dev option 112, before the uploads)
    size_t order_capacity = 0;     // entries d_order can hold
    uint32_t order_grid = 0;       // workgroups the cost-ordered list was dealt to (0: the geometric list, any grid)
    uint32_t dp_share_pct = 60;    // a quarter item's wave time as a percentage of the time its tile took as one item
    uint32_t fill_cost = 2;        // cost units charged for a constant 8x8 tile when balancing
    int dp_min_cost = -1;          // measured tile cost from which a tile is marched depth-parallel (0 = never, < 0 = adaptive)
    uint32_t n_items = 0;
    bool order_dirty = true;
    int n_cus = 256;
    uint32_t wgs_per_cu = 1;
    int kspec = 4;
    bool culling = true;
    int* d_aabb = nullptr;          // occupied macro cells: {xmin, ymin, zmin, xmax, ymax, zmax}
    int h_aabb[6] = {0, 0, 0, -1, -1, -1};
    bool hull_dirty = true;
    volym_camera_uniforms cam_copy;
    volym_parameter_uniforms par_copy;

    FrameParams fp;
    int kernel_variant = 2;
    bool write_f32 = false;
    uint32_t xcd_bands = 0;
    std::string err;
};

static thread_local std::string g_create_error;

static int fail(volym_ctx* c, int code, const std::string& msg)
{
    if (c) c->err = msg; else g_create_error = msg;
    return code;
}

#define HIPCHK(ctx, expr)                                                                      \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(ctx, VOLYM_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

// anything that changes what a tile costs: measured costs no longer apply
static void invalidate_costs(volym_ctx* c)
{
    c->frames_since_change = 0;
    if (c->order_by_cost) c->order_dirty = true;
}

static void recompute_shard(volym_ctx* c)
{
    c->n_local = c->n_tiles > c->rank ? (c->n_tiles - c->rank + c->world - 1) / c->world : 0;
    c->shard_tiles = (c->n_tiles + c->world - 1) / c->world;   // equal-sized shards, padded
}

extern "C" {

int volym_abi_version(void) { return VOLYM_ABI_VERSION; }

const char* volym_last_error(const volym_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

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
    std::memset(&c->h_tables, 0, sizeof c->h_tables);

    auto bail = [&](hipError_t e, const char* what) {
        std::string m = std::string(what) + ": " + hipGetErrorString(e);
        volym_destroy(c);
        return fail(nullptr, e == hipErrorOutOfMemory ? VOLYM_E_NOMEM : VOLYM_E_HIP, m);
    };
    hipError_t e;
    if ((e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking)) != hipSuccess) return bail(e, "hipStreamCreate");
    c->stream = c->own_stream;
    const size_t frame_bytes = static_cast<size_t>(width) * height * 4;
    if ((e = hipMalloc(&c->d_frame_own, frame_bytes)) != hipSuccess) return bail(e, "hipMalloc(frame)");
    if ((e = hipMalloc(&c->d_shard_own, static_cast<size_t>(c->n_tiles) * 1024)) != hipSuccess) return bail(e, "hipMalloc(shard)");
    if ((e = hipMalloc(&c->d_tables, sizeof(FrameTables))) != hipSuccess) return bail(e, "hipMalloc(tables)");
    if ((e = hipMalloc(&c->d_counters, sizeof(Counters))) != hipSuccess) return bail(e, "hipMalloc(counters)");
    if ((e = hipMemset(c->d_frame_own, 0, frame_bytes)) != hipSuccess) return bail(e, "hipMemset(frame)");
    if ((e = hipMalloc(&c->d_aabb, 6 * sizeof(int))) != hipSuccess) return bail(e, "hipMalloc(aabb)");

    c->d_frame = c->d_frame_own;
    c->d_shard = c->d_shard_own;
    *out = c;
    return VOLYM_OK;
}

void volym_destroy(volym_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    (void)hipFree(c->d_vol); (void)hipFree(c->d_imp); (void)hipFree(c->d_tables); (void)hipFree(c->d_mc); (void)hipFree(c->d_df);
    (void)hipFree(c->d_shard_own); (void)hipFree(c->d_frame_own); (void)hipFree(c->d_f32);
    (void)hipFree(c->d_gather_tmp); (void)hipFree(c->d_pack_counters); (void)hipFree(c->d_counters); (void)hipFree(c->d_order); (void)hipFree(c->d_cost); (void)hipFree(c->d_aabb);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

int volym_set_stream(volym_ctx* c, void* hip_stream)
{
    if (!c) return VOLYM_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->own_stream;
    return VOLYM_OK;
}

int volym_set_option(volym_ctx* c, int key, int value)
{
    if (!c) return VOLYM_E_INVALID;
    invalidate_costs(c);
    switch (key) {
    case VOLYM_OPT_KERNEL:
        if (value < 0 || value > 2) return fail(c, VOLYM_E_INVALID, "VOLYM_OPT_KERNEL: 0 (direct), 1 (macro-cell) or 2 (persistent + shading queue)");
        c->kernel_variant = value;
        return VOLYM_OK;
    case VOLYM_OPT_WRITE_F32:
        c->write_f32 = value != 0;
        return VOLYM_OK;
    case VOLYM_OPT_MACRO_CELLS:
        if (value < 4 || value > 32 || (value & (value - 1)) != 0)
            return fail(c, VOLYM_E_INVALID, "VOLYM_OPT_MACRO_CELLS: power of two in 4..32");
        c->mc_n = static_cast<uint32_t>(value);
        c->mc_dirty = true;
        return VOLYM_OK;
    case 101:   // undocumented tuning knob: persistent workgroups per CU (variant 2)
        if (value < 1 || value > 8) return fail(c, VOLYM_E_INVALID, "workgroups per CU: 1..8");
        c->wgs_per_cu = static_cast<uint32_t>(value);
        return VOLYM_OK;
    case 102:   // undocumented tuning knob: speculation depth of variant 2 (1, 2 or 4)
        if (value != 4) return fail(c, VOLYM_E_INVALID, "speculation depth: 4 (the shallower variants were dropped)");
        c->kspec = value;
        return VOLYM_OK;
    case 105:   // undocumented: measured cost from which tiles are marched depth-parallel (0 = never)
        if (value < -100 || value > 65535) return fail(c, VOLYM_E_INVALID, "dp cost threshold: < 0 adaptive (-N = N/10 x fair share), 0 off, else explicit");
        c->dp_min_cost = value;
        c->order_dirty = true;
        return VOLYM_OK;
    case 108:   // undocumented: issue-priority thresholds t1 + 100*t2 + 10000*t3 in tenths of the fair share (0 = no priorities)
        if (value < 0) return fail(c, VOLYM_E_INVALID, "priority thresholds: t1 + 100*t2 + 10000*t3, tenths of the fair share");
        c->prio_tenths[0] = static_cast<uint32_t>(value % 100);
        c->prio_tenths[1] = static_cast<uint32_t>((value / 100) % 100);
        c->prio_tenths[2] = static_cast<uint32_t>(value / 10000);
        c->order_dirty = true;
        return VOLYM_OK;
    case 110:   // undocumented experiment: FrameParams::dev
        c->fp.dev = static_cast<uint32_t>(value);
        return VOLYM_OK;
    case 112:   // undocumented: volume layout, -1 by size / 0 linear / 1 bricked; set before volym_set_volume / volym_set_importances
        if (value < -1 || value > 1) return fail(c, VOLYM_E_INVALID, "layout: -1, 0 or 1");
        c->layout_choice = value;
        return VOLYM_OK;
    case 111:   // undocumented: balancing estimates, dp_share_pct + 1000 * fill_cost
        c->dp_share_pct = static_cast<uint32_t>(value % 1000);
        c->fill_cost = static_cast<uint32_t>(value / 1000);
        c->order_dirty = true;
        return VOLYM_OK;
    case 109:   // undocumented experiment: keep only the depth-parallel items in the work list (the frame is then incomplete)
        c->dev_only_quarters = value != 0;
        c->order_dirty = true;
        return VOLYM_OK;
    case 107:   // undocumented: 0 disables the 16x16 super fill items (A/B tests)
        c->super_fill = value != 0;
        c->order_dirty = true;
        return VOLYM_OK;
    case 104:   // undocumented: 0 disables the cost-feedback reordering of variant 2 (A/B tests)
        c->feedback = value != 0;
        c->order_dirty = true;
        return VOLYM_OK;
    case 103:   // undocumented: 0 disables the exact culling of variant 2 (A/B tests)
        c->culling = value != 0;
        c->hull_dirty = true;
        return VOLYM_OK;
    case 100:   // undocumented tuning knob: block->tile remap bands per XCD (0 = identity)
        if (value < 0 || value > 64) return fail(c, VOLYM_E_INVALID, "xcd bands: 0..64");
        c->xcd_bands = static_cast<uint32_t>(value);
        return VOLYM_OK;
    default:
        return fail(c, VOLYM_E_INVALID, "volym_set_option: unknown key");
    }
}

int volym_set_shard(volym_ctx* c, uint32_t rank, uint32_t world)
{
    if (!c) return VOLYM_E_INVALID;
    if (world == 0 || rank >= world || world > 4096) return fail(c, VOLYM_E_INVALID, "volym_set_shard: need rank < world <= 4096");
    c->rank = rank; c->world = world;
    recompute_shard(c);
    c->order_dirty = true;
    return VOLYM_OK;
}

static bool want_bricked(const volym_ctx* c, uint32_t nx, uint32_t ny, uint32_t nz);

static int upload_volume(volym_ctx* c, uint8_t** dst, const uint8_t* src, uint32_t nx, uint32_t ny, uint32_t nz)
{
    const bool bricked = want_bricked(c, nx, ny, nz);
    if (!src || nx == 0 || ny == 0 || nz == 0) return fail(c, VOLYM_E_INVALID, "volume: NULL data or zero dimension");
    const uint64_t n = static_cast<uint64_t>(nx) * ny * nz;
    const uint64_t nb = bricked ? static_cast<uint64_t>(brick_count(nx)) * brick_count(ny) * brick_count(nz) * 64u : n;
    if (nx > 4096 || ny > 4096 || nz > 4096 || nb > 0xffffffffull)
        return fail(c, VOLYM_E_INVALID, "volume: each dimension <= 4096 and the brick-padded size < 2^32");
    HIPCHK(c, hipSetDevice(c->device));
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
    if (filter != VOLYM_FILTER_NEAREST && filter != VOLYM_FILTER_LINEAR)
        return fail(c, VOLYM_E_INVALID, "volym_set_volume: filter must be VOLYM_FILTER_NEAREST or VOLYM_FILTER_LINEAR");
    int rc = upload_volume(c, &c->d_vol, voxels, nx, ny, nz);
    if (rc != VOLYM_OK) { c->have_vol = false; return rc; }
    c->nx = nx; c->ny = ny; c->nz = nz; c->filter = filter;
    c->bricked = want_bricked(c, nx, ny, nz);
    c->have_vol = true;
    c->mc_dirty = true;
    invalidate_costs(c);
    return VOLYM_OK;
}

int volym_set_importances(volym_ctx* c, const uint8_t* importances, uint32_t nx, uint32_t ny, uint32_t nz)
{
    if (!c) return VOLYM_E_INVALID;
    int rc = upload_volume(c, &c->d_imp, importances, nx, ny, nz);
    if (rc != VOLYM_OK) { c->have_imp = false; return rc; }
    c->inx = nx; c->iny = ny; c->inz = nz;
    c->have_imp = true;
    invalidate_costs(c);
    return VOLYM_OK;
}

int volym_set_transfer_function(volym_ctx* c, const uint8_t* rgba8, uint32_t n)
{
    if (!c) return VOLYM_E_INVALID;
    if (!rgba8 || n < 1 || n > 256) return fail(c, VOLYM_E_INVALID, "volym_set_transfer_function: 1..256 RGBA8 texels");
    std::memset(c->lut, 0, sizeof c->lut);
    std::memcpy(c->lut, rgba8, static_cast<size_t>(n) * 4);
    c->tf_n = n;
    c->have_tf = true;
    c->tables_dirty = true;
    invalidate_costs(c);
    return VOLYM_OK;
}

}  // extern "C"

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

static void build_tables(volym_ctx* c, float alpha_y)
{
    FrameTables& t = c->h_tables;
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
}

// Variant 2 work list: the 8x8-pixel wave tiles of this rank's 16x16 tiles (item = local_tile*4 + sub),
// sorted by Chebyshev distance of the tile centre from the screen centre.  The orbit camera always
// targets the volume centre (src/camera.rs:23), so the long rays are the central ones: they start first.
static int build_order(volym_ctx* c)
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
    std::vector<uint32_t> order(keyed.size());
    for (size_t i = 0; i < keyed.size(); ++i) order[i] = keyed[i].second;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->d_order) { HIPCHK(c, hipFree(c->d_order)); c->d_order = nullptr; }
    if (c->d_cost) { HIPCHK(c, hipFree(c->d_cost)); c->d_cost = nullptr; }
    c->n_items = static_cast<uint32_t>(order.size());
    if (c->n_items) {
        c->order_capacity = order.size() * 4;
        c->order_grid = 0;
        hipError_t e = hipMalloc(&c->d_order, c->order_capacity * sizeof(uint32_t));   // room for quarter-tile items
        if (e == hipSuccess) e = hipMalloc(&c->d_cost, static_cast<size_t>(c->n_local) * 4 * sizeof(uint16_t));
        if (e != hipSuccess) return fail(c, VOLYM_E_NOMEM, std::string("hipMalloc(order): ") + hipGetErrorString(e));
        HIPCHK(c, hipMemcpy(c->d_order, order.data(), order.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        HIPCHK(c, hipMemset(c->d_cost, 0, static_cast<size_t>(c->n_local) * 4 * sizeof(uint16_t)));
    }
    c->h_order = std::move(order);
    c->order_dirty = false;
    c->order_by_cost = false;
    c->frames_since_change = 0;
    return VOLYM_OK;
}

// Scheduling feedback (variant 2).  The frame time is set by the few hundred tiles whose rays take ~10x the
// average number of dependent samples; they should be the first tickets of every workgroup, and only the
// previous frame knows which they are.  When a second frame is requested with nothing changed (the
// reference's benchmark and an idle window render a static view; an orbiting camera never gets here) the
// work list is re-sorted once by the cost the last launch measured, longest first; ties keep the
// centre-first order.  Pixels do not depend on the order.
static int reorder_by_cost(volym_ctx* c)
{
    if (!c->n_items || !c->d_cost) return VOLYM_OK;
    std::vector<uint16_t> cost(static_cast<size_t>(c->n_local) * 4);
    HIPCHK(c, hipMemcpyAsync(cost.data(), c->d_cost, cost.size() * sizeof(uint16_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    // Tiles above dp_min_cost are split into four 4x4 quarter tiles marched depth-parallel (raymarch_pq.h):
    // their cost is a long chain of dependent samples, which four lanes per ray walk ~4x faster, on four waves.
    // Which tiles?  Those that would keep one wave busy for more than about twice a wave's fair share of the
    // frame (sum of costs / resident waves): below that they hide in the bulk and splitting only adds work.
    uint64_t total_cost = 0;
    for (uint32_t item : c->h_order) total_cost += cost[item];
    const uint32_t resident_waves = static_cast<uint32_t>(c->n_cus) * c->wgs_per_cu * PQ_WAVES;
    // dp_min_cost < 0 encodes the factor in tenths (-20 = 2.0 x fair share, the default -1 means 2.0)
    // measured optimum: 1.5x for the table mode, 1.2x for the continuous-rho modes (their classic loop speculates only two
    // samples deep, a depth-parallel item four)
    const bool continuous = (c->fp.flags & (F_LINEAR | F_GAUSSIAN)) != 0u;
    const uint64_t tenths = c->dp_min_cost < -1 ? static_cast<uint64_t>(-c->dp_min_cost) : (continuous ? 12u : 15u);
    const uint32_t adaptive = static_cast<uint32_t>(std::max<uint64_t>(64, tenths * total_cost / (10u * std::max(1u, resident_waves)) + 16));
    const uint32_t dp_thr = c->dp_min_cost < 0 ? adaptive : static_cast<uint32_t>(c->dp_min_cost);
    const bool dp_ok = c->dp_min_cost != 0;
    std::vector<std::pair<uint32_t, uint32_t>> keyed;      // (cost share, item)
    keyed.reserve(c->h_order.size() * 2);
    // 16x16 tiles whose four sub-tiles were all constant become one "super" fill item (bit 30)
    std::vector<uint8_t> all_fill(c->n_local, 1), seen(c->n_local, 0);
    for (uint32_t item : c->h_order) if (cost[item] != 0) all_fill[item >> 2] = 0;
    {
        std::vector<uint8_t> cnt(c->n_local, 0);
        for (uint32_t item : c->h_order) cnt[item >> 2]++;
        for (uint32_t lt = 0; lt < c->n_local; ++lt) if (cnt[lt] != 4) all_fill[lt] = 0;   // sub-tiles outside the frame are not listed
    }
    for (uint32_t item : c->h_order) {
        const uint32_t k = cost[item];
        if (c->super_fill && all_fill[item >> 2]) {
            if (!seen[item >> 2]) { seen[item >> 2] = 1; keyed.emplace_back(0u, 0x40000000u | (item >> 2)); }
            continue;
        }
        if (dp_ok && k >= dp_thr)
            for (uint32_t qd = 0; qd < 4; ++qd) keyed.emplace_back((k * c->dp_share_pct + 99u) / 100u, 0x80000000u | (item << 2) | qd);
        else
            keyed.emplace_back(k, item);
    }
    std::stable_sort(keyed.begin(), keyed.end(), [](const std::pair<uint32_t, uint32_t>& a, const std::pair<uint32_t, uint32_t>& b) { return a.first > b.first; });
    if (c->dev_only_quarters) {     // experiment: how long do the depth-parallel items take with the machine to themselves?
        std::vector<std::pair<uint32_t, uint32_t>> q;
        for (const auto& kv : keyed) if (kv.second >> 31) q.push_back(kv);
        keyed.swap(q);
    }
    // issue priority (bits 28-29) from the item's cost relative to a wave's fair share of the frame
    const uint64_t fair = std::max<uint64_t>(1, total_cost / std::max(1u, resident_waves));
    const bool prio_ok = c->prio_tenths[0] > 0 && static_cast<uint64_t>(c->n_local) * 16u < (1u << 28);
    // Workgroup b reads items b, b + G, ... of the list.  The lists are filled longest-processing-time first: every item,
    // in order of decreasing cost, goes to the workgroup with the least work so far, so the sums differ by less than one
    // item; shorter lists are padded with PQ_NO_ITEM.
    const uint32_t n_keyed = static_cast<uint32_t>(keyed.size());
    const uint32_t G = std::max(1u, std::min((n_keyed + PQ_WAVES - 1) / PQ_WAVES, static_cast<uint32_t>(c->n_cus) * c->wgs_per_cu));
    std::vector<std::vector<uint32_t>> lists(G);
    {
        typedef std::pair<uint64_t, uint32_t> Load;      // (work so far, workgroup)
        std::priority_queue<Load, std::vector<Load>, std::greater<Load>> heap;
        for (uint32_t b = 0; b < G; ++b) heap.emplace(0u, b);
        for (const auto& kv : keyed) {
            uint32_t prio = 0;
            if (prio_ok && kv.first) {
                const uint64_t k10 = static_cast<uint64_t>(kv.first) * 10u;
                prio = k10 >= c->prio_tenths[2] * fair ? 3u : k10 >= c->prio_tenths[1] * fair ? 2u : k10 >= c->prio_tenths[0] * fair ? 1u : 0u;
            }
            Load l = heap.top();
            heap.pop();
            lists[l.second].push_back(kv.second | (prio << 28));
            // constant tiles were measured as 0: a store of 64 or 256 pixels is not free
            const uint32_t floor_share = ((kv.second >> 30) == 1u) ? c->fill_cost * 3u : c->fill_cost;
            l.first += std::max(kv.first, floor_share);
            heap.push(l);
        }
    }
    size_t maxlen = 0;
    for (const auto& l : lists) maxlen = std::max(maxlen, l.size());
    std::vector<uint32_t> order(static_cast<size_t>(G) * maxlen, PQ_NO_ITEM);
    for (uint32_t b = 0; b < G; ++b)
        for (size_t i = 0; i < lists[b].size(); ++i) order[b + static_cast<size_t>(G) * i] = lists[b][i];
    if (order.size() > c->order_capacity) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipFree(c->d_order));
        c->d_order = nullptr;
        HIPCHK(c, hipMalloc(&c->d_order, order.size() * sizeof(uint32_t)));
        c->order_capacity = order.size();
    }
    c->n_items = static_cast<uint32_t>(order.size());
    c->order_grid = G;
    HIPCHK(c, hipMemcpy(c->d_order, order.data(), order.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    c->order_by_cost = true;
    return VOLYM_OK;
}

// ---- exact culling inputs (raymarch_pq.h): hulls of the projected unit cube and of the projected AABB of the
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
    c->hull_dirty = false;
    if (!c->culling) return;
    // AABB of the occupied cells, grown by what a sample may reach beyond its own position
    const bool none = c->h_aabb[3] < c->h_aabb[0];
    if (none) fp.cull |= CULL_NOTHING_DENSE;
    double lo[3] = {0, 0, 0}, hi[3] = {1, 1, 1};
    const double margin = 1.0e-4 + ((fp.flags & F_GAUSSIAN) ? 0.0101 : 0.0);   // smoothing taps sit up to 2*0.005 along the ray (wgsl:53-60)
    if (!none) {
        for (int i = 0; i < 3; ++i) {
            lo[i] = static_cast<double>(c->h_aabb[i]) / c->mc_n - margin;
            hi[i] = static_cast<double>(c->h_aabb[3 + i] + 1) / c->mc_n + margin;
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
    auto project_box = [&](const double blo[3], const double bhi[3], float out[8][4]) -> bool {
        P2 pts[8];
        for (int k = 0; k < 8; ++k) {
            double q[4];
            clip((k & 1) ? bhi[0] : blo[0], (k & 2) ? bhi[1] : blo[1], (k & 4) ? bhi[2] : blo[2], q);
            if (!(q[3] > 1e-3 * scale)) return false;       // a corner at or behind the eye plane: no hull
            pts[k].x = (q[0] / q[3] + 1.0) * 0.5 * c->W;     // wgsl:221-229 inverted: pixel = (ndc + 1)/2 * W
            pts[k].y = (1.0 - q[1] / q[3]) * 0.5 * c->H;
            if (!std::isfinite(pts[k].x) || !std::isfinite(pts[k].y)) return false;
        }
        return hull_edges(pts, 8, out);
    };
    const double c0[3] = {0, 0, 0}, c1[3] = {1, 1, 1};
    if (project_box(c0, c1, fp.hull[0])) fp.cull |= CULL_CUBE_HULL;
    if (!none && project_box(lo, hi, fp.hull[1])) fp.cull |= CULL_OBJ_HULL;
}

static int ensure_frame_resources(volym_ctx* c)
{
    if (c->order_dirty) {
        int rc = build_order(c);
        if (rc != VOLYM_OK) return rc;
    }
    if (c->write_f32 && !c->d_f32) {
        hipError_t e = hipMalloc(&c->d_f32, static_cast<size_t>(c->W) * c->H * sizeof(float4));
        if (e != hipSuccess) return fail(c, VOLYM_E_NOMEM, std::string("hipMalloc(f32 frame): ") + hipGetErrorString(e));
    }
    if (c->mc_dirty || c->mc_built_n != c->mc_n) {
        if (c->d_mc) { HIPCHK(c, hipFree(c->d_mc)); c->d_mc = nullptr; }
        if (c->d_df) { HIPCHK(c, hipFree(c->d_df)); c->d_df = nullptr; }
        const uint32_t cells = c->mc_n * c->mc_n * c->mc_n;
        hipError_t e = hipMalloc(&c->d_mc, cells);
        if (e == hipSuccess) e = hipMalloc(&c->d_df, (cells / 2u + 15u) / 16u * 16u);
        if (e != hipSuccess) return fail(c, VOLYM_E_NOMEM, std::string("hipMalloc(macro cells): ") + hipGetErrorString(e));
        hipLaunchKernelGGL(volym_macrocell_kernel, dim3(cells), dim3(256), 0, c->stream, c->d_vol, c->d_mc, c->nx, c->ny, c->nz, c->mc_n, c->bricked ? 1u : 0u);
        HIPCHK(c, hipGetLastError());
        c->mc_dirty = false;
        c->mc_built_n = c->mc_n;
        c->df_thr_byte = 0xffffffffu;
    }
    if (c->df_thr_byte != c->thr_byte_cull) {
        // stream order: earlier frames finish reading d_df before this kernel rewrites it
        hipLaunchKernelGGL(volym_distance_field_kernel, dim3(1), dim3(1024), 0, c->stream, c->d_mc, c->d_df, c->d_aabb, c->mc_n, c->thr_byte_cull);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(c->h_aabb, c->d_aabb, sizeof c->h_aabb, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));   // only when volume / threshold / grid changed
        c->df_thr_byte = c->thr_byte_cull;
        c->hull_dirty = true;
    }
    if (c->hull_dirty) compute_culling(c);
    return VOLYM_OK;
}

extern "C" {

int volym_update(volym_ctx* c, const volym_camera_uniforms* cam, const volym_parameter_uniforms* par)
{
    if (!c) return VOLYM_E_INVALID;
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
    fp.nx = c->nx; fp.ny = c->ny; fp.nz = c->nz;
    fp.tiles_x = c->tiles_x; fp.n_tiles = c->n_tiles;
    fp.tf_n = c->tf_n;
    const float sigma = 1.5f;            // wgsl:255
    for (int i = -2; i <= 2; ++i) {
        const float x = static_cast<float>(i) * 0.005f;
        fp.gauss_w[i + 2] = wgsl_exp(-(x * x) / (2.0f * sigma * sigma));
    }
    std::memcpy(fp.cone_cos, k_cone_cos, sizeof k_cone_cos);
    std::memcpy(fp.cone_sin, k_cone_sin, sizeof k_cone_sin);

    if (c->tables_dirty || c->tables_alpha_y != fp.alpha_y) {
        build_tables(c, fp.alpha_y);
        HIPCHK(c, hipStreamSynchronize(c->stream));   // the previous frame may still read d_tables
        HIPCHK(c, hipMemcpy(c->d_tables, &c->h_tables, sizeof(FrameTables), hipMemcpyHostToDevice));
        c->tables_dirty = false;
        c->tables_alpha_y = fp.alpha_y;
    }
    uint32_t tb = 256;
    for (int b = 255; b >= 0; --b)
        if (c->h_tables.rho[b] >= fp.thr) tb = static_cast<uint32_t>(b); else break;
    fp.thr_byte = tb;
    // continuous-rho modes (trilinear / smoothed) compare an interpolated value: give its rounding some room
    if (fp.flags & (F_LINEAR | F_GAUSSIAN)) {
        const float cons = fp.thr - std::fabs(fp.thr) * 1.0e-5f - 1.0e-7f;
        uint32_t tc = 256;
        for (int b = 255; b >= 0; --b)
            if (c->h_tables.rho[b] >= cons) tc = static_cast<uint32_t>(b); else break;
        c->thr_byte_cull = tc;
    } else {
        c->thr_byte_cull = tb;
    }
    if (!c->have_frame || std::memcmp(&c->cam_copy, cam, sizeof *cam) != 0 || std::memcmp(&c->par_copy, par, sizeof *par) != 0) {
        c->frames_since_change = 0;          // the measured costs describe another view
        if (c->order_by_cost) c->order_dirty = true;   // back to the geometric order until re-measured
    }
    c->cam_copy = *cam;
    c->par_copy = *par;
    c->hull_dirty = true;
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
    Counters* cnt = COUNT ? c->d_counters : nullptr;
    uint4* trace = TRACE ? c->d_trace : nullptr;
    if (c->kernel_variant == 2) {
        if (c->n_items == 0) return VOLYM_OK;
        const bool plain = !COUNT && !TRACE;
        if (plain && c->feedback && !c->order_by_cost && c->frames_since_change == 1) {
            int rcc = reorder_by_cost(c);
            if (rcc != VOLYM_OK) return rcc;
        }
        uint16_t* cost_out = (plain && c->feedback && !c->order_by_cost) ? c->d_cost : nullptr;
        if (plain) c->frames_since_change++;
        const uint32_t want = (c->n_items + PQ_WAVES - 1) / PQ_WAVES;
        const uint32_t pgrid = c->order_grid ? c->order_grid : std::max(1u, std::min(want, static_cast<uint32_t>(c->n_cus) * c->wgs_per_cu));
        const bool table = !(fp.flags & (F_LINEAR | F_GAUSSIAN));
#define VOLYM_PQ_LAUNCH(T, KS, I, B, R)                                                                                          \
    hipLaunchKernelGGL((volym_raymarch_pq_kernel<T, COUNT && I, TRACE, KS, I, B, R>), dim3(pgrid), dim3(PQ_THREADS), 0, c->stream, c->d_vol,  \
                       c->d_imp, c->d_tables, c->d_df, c->d_order, c->n_items, cost_out, c->d_shard, c->d_frame, c->d_f32, cnt, trace, fp)
        // IMP = false: opacity on and no importance colouring (the common cases), without (IR = false) or with (IR = true)
        // importance rendering; the instrumented launch always takes the general form
        const bool special = !COUNT && !(fp.flags & F_IMP_COLORING) && (fp.flags & F_OPACITY);
        const bool no_imp = special && !(fp.flags & F_IMP_RENDERING);
        const bool ir = special && (fp.flags & F_IMP_RENDERING);
        if (c->bricked) {
            if (table && no_imp) VOLYM_PQ_LAUNCH(true, 4, false, true, false);
            else if (table && ir) VOLYM_PQ_LAUNCH(true, 4, false, true, true);
            else if (table) VOLYM_PQ_LAUNCH(true, 4, true, true, false);
            else VOLYM_PQ_LAUNCH(false, 1, true, true, false);
        } else {
            if (table && no_imp) VOLYM_PQ_LAUNCH(true, 4, false, false, false);
            else if (table && ir) VOLYM_PQ_LAUNCH(true, 4, false, false, true);
            else if (table) VOLYM_PQ_LAUNCH(true, 4, true, false, false);
            else VOLYM_PQ_LAUNCH(false, 1, true, false, false);
        }
#undef VOLYM_PQ_LAUNCH
        HIPCHK(c, hipGetLastError());
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

extern "C" {

int volym_compute_pass(volym_ctx* c)
{
    if (!c) return VOLYM_E_INVALID;
    if (!c->have_frame) return fail(c, VOLYM_E_STATE, "volym_compute_pass: call volym_update first");
    HIPCHK(c, hipSetDevice(c->device));
    return launch_march<false>(c);
}

int volym_sync(volym_ctx* c)
{
    if (!c) return VOLYM_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return VOLYM_OK;
}

int volym_read_rgba8(volym_ctx* c, uint8_t* out)
{
    if (!c || !out) return VOLYM_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(out, c->d_frame, static_cast<size_t>(c->W) * c->H * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return VOLYM_OK;
}

int volym_read_rgba32f(volym_ctx* c, float* out)
{
    if (!c || !out) return VOLYM_E_INVALID;
    if (!c->write_f32 || !c->d_f32 || c->world != 1)
        return fail(c, VOLYM_E_STATE, "volym_read_rgba32f: needs VOLYM_OPT_WRITE_F32 = 1, world == 1 and a rendered frame");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(out, c->d_f32, static_cast<size_t>(c->W) * c->H * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
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
    if (!c || !packed) return VOLYM_E_INVALID;
    const size_t header = pack_header_bytes(c->shard_tiles);
    if (capacity_bytes < header) return fail(c, VOLYM_E_INVALID, "volym_pack_shard: the buffer does not even hold the header (volym_packed_shard_bytes)");
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->d_pack_counters) {
        HIPCHK(c, hipMalloc(&c->d_pack_counters, 4 * sizeof(uint32_t)));
        HIPCHK(c, hipMemsetAsync(c->d_pack_counters, 0, 4 * sizeof(uint32_t), c->stream));
    }
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
    if (!c || !tiles_used) return VOLYM_E_INVALID;
    if (!c->d_pack_counters) return fail(c, VOLYM_E_STATE, "volym_packed_tiles: no volym_pack_shard yet");
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
    if (!c || !gathered) return VOLYM_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    hipLaunchKernelGGL(volym_assemble_kernel, dim3(c->n_tiles), dim3(256), 0, c->stream, static_cast<const uint32_t*>(gathered),
                       c->d_frame, c->W, c->H, c->tiles_x, c->n_tiles, c->world, c->shard_tiles);
    HIPCHK(c, hipGetLastError());
    return VOLYM_OK;
}

int volym_assemble_host(volym_ctx* c, const uint8_t* gathered_host)
{
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

// Development aid: per-item costs (uint16) of the last measuring launch; out needs 4 * n_local entries.
int volym_dev_read_costs(volym_ctx* c, uint16_t* out, uint32_t max_items)
{
    if (!c || !out || !c->d_cost) return VOLYM_E_INVALID;
    const uint32_t n = c->n_local * 4u;
    if (max_items < n) return VOLYM_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(out, c->d_cost, n * sizeof(uint16_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return static_cast<int>(n);
}

// Development aid: the work list as the kernel reads it (after cost feedback); returns the number of items.
int volym_dev_read_order(volym_ctx* c, uint32_t* out, uint32_t max_items)
{
    if (!c || !out || !c->d_order) return VOLYM_E_INVALID;
    if (max_items < c->n_items) return VOLYM_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(out, c->d_order, c->n_items * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return static_cast<int>(c->n_items);
}

// Development aid (not declared in the public header): one instrumented launch that records, per
// wave, {start tick, duration ticks (100 MHz), max loop iterations of a lane, max dense samples}.
// out: 4 * grid_blocks uint4 records; returns the number of records or a negative error.
int volym_dev_wave_trace(volym_ctx* c, uint32_t* out, uint32_t max_records)
{
    if (!c || !out) return VOLYM_E_INVALID;
    if (!c->have_frame) return fail(c, VOLYM_E_STATE, "volym_dev_wave_trace: call volym_update first");
    HIPCHK(c, hipSetDevice(c->device));
    const uint32_t records = (c->n_local + 64u) * 4u * 2u;
    if (max_records < records) return fail(c, VOLYM_E_INVALID, "volym_dev_wave_trace: buffer too small");
    HIPCHK(c, hipMalloc(&c->d_trace, static_cast<size_t>(records) * sizeof(uint4)));
    HIPCHK(c, hipMemsetAsync(c->d_trace, 0, static_cast<size_t>(records) * sizeof(uint4), c->stream));
    // a lone launch on an idle GPU runs at idle clocks: trace the 31st of 31 back-to-back passes
    int rc = VOLYM_OK;
    for (int i = 0; i < 30 && rc == VOLYM_OK; ++i) rc = launch_march<false, false>(c);
    if (rc == VOLYM_OK) rc = launch_march<false, true>(c);
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

}  // extern "C"
