// Internal: the context behind include/volym_hip.h (one device, one stream, one W x H output) and the pieces of host
// logic that more than one translation unit needs (raymarch.hip: C ABI; mgpu.hip: the native multi-GPU loop).
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/volym_hip.h"
#include "raymarch_device.h"

namespace volym {

// One work list as the kernel reads it: workgroup b takes entries b, b + grid, ... (raymarch_pq.h).
struct WorkList {
    std::vector<uint32_t> entries;   // host copy (the feedback thread maps list positions back to tiles)
    uint32_t grid = 0;               // workgroups the list was dealt to (0: the geometric list, any grid)
    uint64_t view_serial = 0;        // the view whose measured costs produced it (0: none, geometric order)
    bool has_dp = false;             // holds depth-parallel entries (their costs come back as estimates)
    bool trimmable = false;          // dealt for a standing view from costs measured on whole entries
    uint32_t trim_round = 0;         // times the list was re-balanced from measured workgroup times since it was dealt
    bool final_for_view = false;     // trimmable and trimmed as often as asked: no more captures
    std::vector<uint16_t> shares;    // by list position: the cost share the entry was dealt with (trimmable lists)
};

}  // namespace volym

struct volym_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t own_stream = nullptr;
    hipStream_t copy_stream = nullptr;      // cost read-backs and work-list uploads of the feedback thread
    uint32_t W = 0, H = 0, tiles_x = 0, tiles_y = 0, n_tiles = 0;
    uint32_t rank = 0, world = 1, n_local = 0, shard_tiles = 0;

    uint8_t* d_vol = nullptr;
    uint8_t* d_imp = nullptr;
    uint32_t nx = 0, ny = 0, nz = 0;
    uint32_t inx = 0, iny = 0, inz = 0;
    int imp_box_lo[3] = {1, 1, 1}, imp_box_hi[3] = {0, 0, 0};   // texel AABB of the importances >= 128 (lo > hi: none)
    int filter = VOLYM_FILTER_NEAREST;
    uint8_t lut[256 * 4] = {};
    uint32_t tf_n = 0;
    bool have_vol = false, have_imp = false, have_tf = false, have_frame = false;

    // per-(transfer function, step) tables: one device copy, refreshed in stream order from a ring of pinned stagings
    static constexpr int TABLE_RING = 8;
    volym::FrameTables* d_tables = nullptr;
    volym::FrameTables* h_tables[TABLE_RING] = {};
    hipEvent_t tables_ev[TABLE_RING] = {};
    int tables_slot = 0;
    volym::FrameTables tables_now;           // what d_tables holds (or will, in stream order)
    bool tables_dirty = true;
    float tables_alpha_y = -1.0f;

    uint8_t* d_mc = nullptr;                 // per-macro-cell density maxima
    uint8_t* d_df = nullptr;                 // packed 4-bit distance field for (d_mc, thr_byte)
    std::vector<uint8_t> h_mc;               // host copy of d_mc
    int aabb_tab[257][6];                    // occupied-cell AABB per threshold byte {x0,y0,z0,x1,y1,z1}; x1 < x0: none
    uint32_t mc_n = 32;
    uint32_t df_thr_byte = 0xffffffffu;
    uint32_t thr_byte_cull = 256;
    uint32_t* d_tile_mask = nullptr;         // one bit per 8x8 pixel tile: some occupied macro cell projects onto it (per view); two
                                             // buffers of tile_mask_words: the one in use and the one being kept zeroed for the next view
    int mask_cur = 0;
    uint32_t tile_mask_words = 0;            // 0: the frame has more tiles than the mask kernel holds in LDS -- no mask
    bool tile_mask = true;                   // dev switch
    bool mask_wanted = false;                // compute_culling: this view gets a mask
    bool mask_pending = false;               // ... and has not got it yet
    bool mask_eager = false;                 // dev
    uint32_t view_launches = 0;              // launches since the view last changed
    float mask_clip[16] = {};                // world -> clip of the view (f32 copy for the mask kernel)
    float mask_margin = 0.0f;
    int* d_aabb = nullptr;                   // written by the distance-field kernel (kept for the dev tools)

    uint32_t* d_shard_own = nullptr;
    uint32_t* d_frame_own = nullptr;
    uint32_t* d_shard = nullptr;
    uint32_t* d_frame = nullptr;
    float4* d_f32 = nullptr;
    uint32_t* d_blit = nullptr;              // volym_blit target when the caller passes none
    size_t blit_bytes = 0;
    uint32_t blit_w = 0, blit_h = 0;
    uint8_t* d_gather_tmp = nullptr;
    uint32_t* d_pack_counters = nullptr;
    uint32_t pack_parity = 0;
    size_t gather_tmp_bytes = 0;
    volym::Counters* d_counters = nullptr;
    uint4* d_trace = nullptr;

    // ---- work lists + cost feedback (variant 2) ----
    uint32_t* d_list[2] = {nullptr, nullptr};   // device lists: `cur` is launched from, the other is the feedback thread's
    uint32_t* h_list_pinned = nullptr;          // staging of the list the feedback thread uploads
    uint16_t* d_cost = nullptr;                 // position-indexed costs of ONE captured launch, then (u32) the end time of every
    uint16_t* h_cost_pinned = nullptr;          // wave and the start time of every workgroup of that launch (raymarch_pq.h)
    size_t list_capacity = 0;                   // entries each of the above can hold
    int cur = 0;
    volym::WorkList lists[2];
    std::vector<uint32_t> geometric;            // centre-first list of this shard (rebuilt by the setup calls)
    std::vector<uint16_t> item_cost;            // last measured / estimated cost per 8x8 item (4 * n_local), carried across views
    std::vector<uint8_t> item_is_dp;            // hysteresis of the depth-parallel split
    std::atomic<uint64_t> view_serial{1};       // bumped by every volym_update that changes the uniforms (read by the feedback thread)
    bool lists_ready = false;
    // ---- variant 3 (ray pool): {-, -, error bits of the frames so far}
    uint32_t* d_pool_sync = nullptr;
    bool pool_launched = false;
    uint32_t* d_pool_dbg = nullptr;             // development timeline of variant 3 (volym_dev_pool_timeline)
    bool pool_dbg = false;

    enum : int { FB_IDLE = 0, FB_CAPTURED = 1, FB_READY = 2, FB_QUIT = 3 };
    std::thread fb_thread;
    std::mutex fb_mu;
    std::condition_variable fb_cv;
    std::atomic<int> fb_state{FB_IDLE};
    hipEvent_t ev_march = nullptr, ev_cost = nullptr, ev_list = nullptr;
    struct FbJob {                               // written by the caller before FB_CAPTURED, by the worker before FB_READY
        int list = 0;                            // which of lists[] the captured launch ran
        uint32_t n_entries = 0;
        uint64_t view_serial = 0;
        bool captured_has_dp = false;
        bool continuous = false;
        bool plain = false;                      // table mode, no importance mode (the common instantiation)
        uint32_t max_grid = 0, waves = 16;
        int dp_min_cost = -1;
        uint32_t dp_share_pct = 60, fill_cost = 2;
        bool super_fill = true, only_quarters = false;
        int dilate = -1;
        uint32_t grid = 0;                       // workgroups of the captured launch
        uint32_t dev_drop_tenths = 0;            // dev
        uint32_t dp_floor = 64;
        uint32_t trim_rounds = 0;
        double t_us[6] = {};                     // dev: wall-clock stamps of the job's stages
        uint32_t prio_tenths[3] = {3, 6, 10};
        std::string error;                       // worker -> caller
    } fb_job;

    static constexpr uint32_t THROTTLE_RING = 9;          // one more than the deepest wait volym_throttle accepts (8)
    hipEvent_t throttle_ev[THROTTLE_RING] = {};
    uint32_t throttle_head = 0;

    bool feedback = true;
    bool feedback_frozen = false;               // dev
    int wide_waves = 0;                         // dev: 0 default choice, 12 or 16 (raymarch.hip launch_march)
    uint32_t trim_rounds = 0;                   // re-balancing rounds from measured workgroup times after a standing view's list is dealt (VOLYM_OPT_REBALANCE_ROUNDS; off: see trim_list)
    int cost_dilate = -1;                       // radius (8x8 items) of the max-filter over the cost map before dealing; -1: 1 while the view moves, else 0
    bool super_fill = true;
    uint32_t prio_tenths[3] = {3, 6, 10};
    bool dev_only_quarters = false;
    uint32_t dev_drop_tenths = 0;
    uint32_t dp_floor = 64;                      // floor of the adaptive split threshold, cost units (deal_list)
    bool bricked = false;
    uint64_t brick_from_bytes = 64ull << 20;
    int layout_choice = -1;
    uint32_t dp_share_pct = 60;
    uint32_t fill_cost = 2;
    int dp_min_cost = -1;
    int n_cus = 256;
    uint32_t wgs_per_cu = 1;
    int kspec = 4;
    bool culling = true;
    // VOLYM_OPT_FRAMES_IN_FLIGHT = 2: a second context on the same device (own stream, frame buffer, lists and feedback) that renders
    // every other frame, so that a frame's workgroups take the CUs the previous frame's tail leaves idle (raymarch.hip "frames in flight")
    volym_ctx* twin = nullptr;
    volym_ctx* last = nullptr;                  // the context the latest volym_compute_pass went to (this one or the twin)
    volym_ctx* last_blit = nullptr;
    uint32_t flight_parity = 0;
    std::vector<std::pair<int, int>> option_log;    // options set so far: replayed into a twin created later
    bool setup_ieee = false;                    // VOLYM_OPT_SETUP_IEEE: make_ray with plain divisions (FrameParams::setup_lo = +inf)
    bool straight_jobs = false;                 // dev switch (option 121): CJ = 2 instantiation for the straight look-ahead
    bool lds_bricks = false;        // dev option 122: LDS-staged bricks in the common instantiation (bricked layout)
    bool hull_dirty = true;
    volym_camera_uniforms cam_copy;
    volym_parameter_uniforms par_copy;

    volym::FrameParams fp;
    int kernel_variant = 2;
    bool write_f32 = false;
    uint32_t xcd_bands = 0;
    std::string err;
};

namespace volym {

int ctx_fail(volym_ctx* c, int code, const std::string& msg);
// one plain ray-march launch on the context's stream (what volym_compute_pass enqueues); used by the multi-GPU loop
int ctx_launch_march(volym_ctx* c);

}  // namespace volym

#define VOLYM_HIPCHK(ctx, expr)                                                                          \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess)                                                                            \
            return volym::ctx_fail(ctx, VOLYM_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)
