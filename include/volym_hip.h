/*
 * volym_hip.h -- C ABI of the MI355X-native ray-march path (libvolym_hip.so).
 *
 * This is the drop-in boundary for the reference's compute plugin:
 *
 *     trait ComputeDemo { fn init(ctx, state, output_texture) -> Result<Self>;
 *                         fn update_gpu_state(&self, ctx, state) -> Result<()>;
 *                         fn compute_pass(&self, ctx) -> Result<()>; }
 *                                              -- /root/reference/src/demos/mod.rs:9-17
 *
 * and for the wgpu resource wrappers it owns (src/gpu_resources/, src/gpu_context.rs,
 * src/demos/pipeline.rs).  Plain C, no C++/torch types, no exceptions: a Rust
 * `extern "C"` block binds it unchanged (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - every function returns 0 on success or a negative VOLYM_E_* code; the text of
 *     the last failure is volym_last_error(ctx) (volym_last_error(NULL) for create).
 *   - a context is bound to one device and is NOT thread-safe (the reference calls
 *     everything from the winit event-loop thread, src/event_loop.rs:62).
 *   - volym_set_* copy from caller memory; the caller may free immediately (as
 *     queue.write_texture does, src/gpu_resources/volume.rs:81-90).
 *   - volym_update and volym_compute_pass only enqueue (src/demos/pipeline.rs:97 queue.submit): no allocation,
 *     no stream synchronisation on their path.  The one exception is spelled out at volym_update.
 *     Blocking calls: volym_create / volym_destroy, volym_set_* (uploads; they also wait for the frames in flight),
 *     volym_set_option, volym_set_shard, volym_set_stream, volym_sync, volym_settle, volym_read_*, volym_packed_tiles,
 *     volym_assemble_host, volym_stats_pass, volym_time_*; volym_blit blocks only when it has to (re)size its own target.
 *   - the default kernel schedules its work from lists that a feedback thread inside the library re-deals from the
 *     counted cost of earlier frames (asynchronously: a frame never waits for it, and every list renders the same
 *     pixels); volym_settle runs that loop to its fixed point for the current view.
 */
#ifndef VOLYM_HIP_H
#define VOLYM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VOLYM_ABI_VERSION 2

enum {
    VOLYM_OK = 0,
    VOLYM_E_INVALID = -1,   /* bad argument / call order */
    VOLYM_E_HIP = -2,       /* a HIP runtime call failed */
    VOLYM_E_NO_DEVICE = -3, /* no usable gfx950 device */
    VOLYM_E_NOMEM = -4,
    VOLYM_E_STATE = -5      /* volume / importances / transfer function not all set */
};

/* Volume sampler.  NEAREST is what the reference runs (SamplerDescriptor::default(),
 * src/gpu_resources/volume.rs:92-95); LINEAR is the trilinear mode BASELINE.json names. */
enum { VOLYM_FILTER_NEAREST = 0, VOLYM_FILTER_LINEAR = 1 };

/* volym_set_option keys */
enum {
    VOLYM_OPT_KERNEL = 1,     /* 0 = direct march (every reference fetch issued),
                                 1 = macro-cell march: empty-space fetches elided, step
                                     arithmetic replayed exactly,
                                 2 = (default) 1 + persistent workgroups, centre-first tile
                                     order, per-wave shading queue, speculative sample batches,
                                 3 = ray pool (round 3): ray state in LDS, phase lists, pixel-block
                                     lattice dealing, no work list and no cost feedback; the common
                                     flag set only (nearest filter, opacity on, no importance mode --
                                     every other frame runs kernel 2); same pixels as 2, ~3x slower at
                                     1920x1080 */
    VOLYM_OPT_WRITE_F32 = 2,  /* 1 = also store pre-quantisation float RGBA (parity tests) */
    VOLYM_OPT_MACRO_CELLS = 3, /* macro cells per axis (power of two, 4..32; default 32)    */
    VOLYM_OPT_VOLUME_LAYOUT = 4, /* device layout of the NEXT volume / importance upload: -1 = by size (default: 4x4x4
                                    bricks above 64 MiB), 0 = linear, 1 = bricks.  Invisible at this boundary. */
    VOLYM_OPT_CULLING = 5,     /* 0 = no exact culling (projected hulls, AABB clip, per-view tile mask of the occupied cells) in
                                  kernel 2; default 1 */
    VOLYM_OPT_COST_FEEDBACK = 6, /* 0 = kernel 2 keeps its centre-first work list; default 1 (lists re-dealt from counted costs) */
    VOLYM_OPT_DEPTH_PARALLEL = 7, /* tile cost from which kernel 2 marches a tile as four depth-parallel quarter items:
                                    < 0 adaptive (-N = N/10 x a wave's fair share of the frame; -1 = default), 0 never, > 0 explicit */
    VOLYM_OPT_XCD_BANDS = 8,   /* kernels 0/1: block -> tile remap bands per XCD (0 = identity, default) */
    VOLYM_OPT_REBALANCE_ROUNDS = 9, /* kernel 2: after a standing view's list is dealt, re-balance it this many times (0..8) from
                                    the times its workgroups took (measured: the list then depends on the weather).  Default 0. */
    VOLYM_OPT_FRAMES_IN_FLIGHT = 11, /* 1 (default) = one frame after the other on the context's stream.  2 = the context keeps a
                                  second complete context on the same device (stream, frame buffer, copy of the volume, lists,
                                  feedback) and volym_compute_pass alternates between the two: a frame of the persistent kernel
                                  ends on its longest chains, and the next frame's workgroups take the CUs it leaves idle
                                  (1920x1080: 31.9 -> 27.6 us per frame; 3840x2160: 74.7 -> 70.7).  Set it before the volume,
                                  the importances and the transfer function (they go to both), and not with a caller's stream.
                                  volym_update / _settle / _sync / _set_* act on both; volym_read_rgba8 / _rgba32f / volym_blit
                                  take the frame of the latest pass; volym_throttle marks it; volym_stats_pass, volym_time_*
                                  and volym_selftest_ray_setup use the first context alone (one frame at a time); the shard /
                                  pack / assemble calls and volym_set_stream return VOLYM_E_STATE.  Memory: everything twice. */
    VOLYM_OPT_SETUP_IEEE = 10  /* 1 = the ray set-up (wgsl:221-241) runs its 14 divisions as 14 plain IEEE divisions; default 0: the
                                  divisions that share a denominator share its refined reciprocal -- the same instructions on the
                                  same values, so the same bits (raymarch_device.h make_ray; volym_selftest_ray_setup).  Takes
                                  effect with the next volym_update. */
};

/* CameraUniforms, byte-for-byte (src/gpu_resources/camera.rs:56-64; WGSL mirror
 * shaders/importance_driven_volume_rendering.wgsl:2-7).  Column-major m[col][row]. */
typedef struct volym_camera_uniforms {
    float view_matrix[4][4];
    float projection_matrix[4][4];
    float inverse_view_proj[4][4];
    float camera_position[3];
    float _padding;
} volym_camera_uniforms;

/* ParameterUniforms, byte-for-byte (src/gpu_resources/parameters.rs:55-66; WGSL mirror
 * shaders/importance_driven_volume_rendering.wgsl:9-18). */
typedef struct volym_parameter_uniforms {
    float density_threshold;
    uint32_t use_cone_importance_check;
    uint32_t use_importance_coloring;
    uint32_t use_opacity;
    uint32_t use_importance_rendering;
    uint32_t use_gaussian_smoothing;
    uint32_t importance_check_ahead_steps;
    float raymarching_step_size;
} volym_parameter_uniforms;

/* Reference texture fetches of the current frame, counted by an instrumented
 * (untimed) launch: B_alg = n_vol*b_vol + n_imp + 4*W*H  (SURVEY.md section 8d). */
typedef struct volym_stats {
    uint64_t n_vol;    /* density fetches the reference shader executes   */
    uint64_t n_imp;    /* importance fetches the reference shader executes */
    uint64_t n_steps;  /* march-loop iterations                             */
    uint64_t n_dense;  /* iterations with rho >= density_threshold          */
    uint64_t n_hit;    /* rays that hit the unit cube                       */
    uint64_t n_rays;   /* pixels owned by this context (all of them count)  */
} volym_stats;

typedef struct volym_ctx volym_ctx;

/* --- lifetime: GpuContext::new + GpuWriteTexture2D::new (src/gpu_context.rs:20-62,
 * src/gpu_resources/texture.rs:40-59).  device_id < 0 = current device. */
int volym_create(volym_ctx** out, uint32_t width, uint32_t height, int device_id);
void volym_destroy(volym_ctx* ctx);
const char* volym_last_error(const volym_ctx* ctx);
int volym_abi_version(void);

/* Use the caller's HIP stream (hipStream_t as void*; NULL = the context's own). */
int volym_set_stream(volym_ctx* ctx, void* hip_stream);
int volym_set_option(volym_ctx* ctx, int key, int value);

/* Screen-tile sharding: this context renders the 16x16 tiles
 * k with k % world == rank into a compact buffer of volym_local_tiles() tiles. */
int volym_set_shard(volym_ctx* ctx, uint32_t rank, uint32_t world);

/* --- resources: Simple::init (src/demos/simple/mod.rs:36-110) --------------------- */
/* GpuVolume::init upload (src/gpu_resources/volume.rs:63-95): nx*ny*nz bytes, x fastest,
 * already padded/flipped by the host shim (volym_host.h volym_prepare_volume).  The caller's
 * layout is always x fastest; on the device, volumes above 64 MiB are re-laid into 4x4x4 bricks
 * (DESIGN.md section 3), which is invisible at this boundary. */
int volym_set_volume(volym_ctx* ctx, const uint8_t* voxels, uint32_t nx, uint32_t ny,
                     uint32_t nz, int filter);
/* GpuImportances::init upload (src/demos/simple/importance.rs:93-131); same dims. */
int volym_set_importances(volym_ctx* ctx, const uint8_t* importances, uint32_t nx,
                          uint32_t ny, uint32_t nz);
/* GPUTransferFunction::new_texture_1d_rgbt upload (src/gpu_resources/transfer_function.rs:36-90):
 * n RGBA8 texels (the reference uses n = 256), Linear/ClampToEdge sampler. */
int volym_set_transfer_function(volym_ctx* ctx, const uint8_t* rgba8, uint32_t n);

/* --- per frame ------------------------------------------------------------------ */
/* ComputeDemo::update_gpu_state (src/demos/pipeline.rs:208-212): GpuCamera::update +
 * GpuParameters::update.  Fails with VOLYM_E_INVALID on out-of-range parameters.  Enqueue only; a change of the step size
 * or of the transfer function uploads 10 KiB of tables in stream order from a ring of 8 staging buffers, and only a caller
 * that makes 8 such changes while the device is still 8 frames behind waits for a slot. */
int volym_update(volym_ctx* ctx, const volym_camera_uniforms* camera,
                 const volym_parameter_uniforms* parameters);
/* ComputeDemo::compute_pass (src/demos/pipeline.rs:62-102, :214-225): enqueue one
 * ray-march of every owned tile on the context's stream; returns immediately. */
int volym_compute_pass(volym_ctx* ctx);
int volym_sync(volym_ctx* ctx);
/* Back-pressure for a frame loop, the role surface.get_current_texture() plays in the reference (src/event_loop.rs:114: it
 * blocks while the swap chain's images are all in flight).  Call once per frame after volym_compute_pass: it marks the work
 * enqueued so far and waits until at most `max_in_flight` (1..8) such marks are outstanding.  A loop that runs hundreds of
 * frames ahead of the device also runs hundreds of frames ahead of the cost feedback of its work lists. */
int volym_throttle(volym_ctx* ctx, uint32_t max_in_flight);
/* Bring the cost feedback (see the conventions above) to rest for the current view: waits for a re-deal in flight, then
 * enqueues frames of the current view itself (exactly what volym_compute_pass enqueues: the output buffers are rewritten
 * with the same pixels) until the list in use is the final one for this view.  A handful of frames at most; blocks.
 * The frames after it run at the steady rate of a standing view.  Never needed for correctness. */
int volym_settle(volym_ctx* ctx);

/* --- output --------------------------------------------------------------------- */
/* Full-frame readback (world == 1, or after volym_assemble on the root):
 * W*H*4 bytes as the rgba8unorm store leaves them / W*H*4 floats before quantisation
 * (the latter needs VOLYM_OPT_WRITE_F32 = 1). */
int volym_read_rgba8(volym_ctx* ctx, uint8_t* out);
int volym_read_rgba32f(volym_ctx* ctx, float* out);

/* The step after the path: RenderPipeline::render_pass (src/render_pipeline.rs:88-130) with shaders/render.wgsl:39-43 --
 * every pixel (x, y) of an out_w x out_h rgba8 target samples the frame at uv = (x + 0.5, y + 0.5) / (W, H) through a
 * Linear / ClampToEdge sampler (src/gpu_resources/texture.rs:84-101), BlendState::REPLACE.  Note the divisor: the INPUT
 * size, as in the shader, so the pass maps pixels 1:1 (a larger target repeats the edge texels, a smaller one crops).
 * target_rgba8 = device memory of out_w*out_h*4 bytes, or NULL for a target the context owns (volym_read_blit reads it
 * back).  Enqueued on the context's stream behind the frame. */
int volym_blit(volym_ctx* ctx, void* target_rgba8, uint32_t out_w, uint32_t out_h);
int volym_read_blit(volym_ctx* ctx, uint8_t* out);

/* Sharded output.  Local buffer = volym_local_tiles() tiles of 16x16 RGBA8 pixels
 * (1024 bytes each, padded to volym_shard_bytes()); device pointer for the collective. */
uint32_t volym_local_tiles(const volym_ctx* ctx);
size_t volym_shard_bytes(const volym_ctx* ctx);
void* volym_shard_device_ptr(volym_ctx* ctx);
void* volym_frame_device_ptr(volym_ctx* ctx);
/* Let the caller own the device buffers (e.g. torch tensors handed to RCCL):
 * shard_rgba8 = volym_shard_bytes() bytes, frame_rgba8 = W*H*4 bytes; NULL keeps ours. */
int volym_bind_output(volym_ctx* ctx, void* shard_rgba8, void* frame_rgba8);
/* Root side of the image gather: `gathered` = world shards back to back in rank order
 * (device memory, world * volym_shard_bytes() bytes) -> raster W*H*4 in the frame buffer. */
int volym_assemble(volym_ctx* ctx, const void* gathered);
/* Packed shards (new; the reference has no multi-GPU path): most 16x16 tiles of a frame are constant -- outside the
 * volume's silhouette -- and the gather only has to move the others.  volym_pack_shard enqueues, after a
 * volym_compute_pass, the compaction of the bound shard into `packed`: a header (one {slot | constant flag, value} pair per
 * local tile) followed by the 1 KiB tiles that are not constant; a tile that finds no room sets the overflow flag.
 * volym_packed_shard_bytes(ctx, tiles): bytes of a packed shard with room for `tiles` tiles (tiles >= local tiles: always
 * enough).  volym_packed_tiles: synchronises and reports how many tiles the last pack stored (size the steady-state buffers
 * with the maximum over the ranks).  volym_assemble_packed: root side, `gathered` = world packed shards `stride_bytes`
 * apart in rank order -> the raster in the frame buffer. */
size_t volym_packed_shard_bytes(const volym_ctx* ctx, uint32_t tiles);
int volym_pack_shard(volym_ctx* ctx, void* packed_device, size_t capacity_bytes);
int volym_packed_tiles(volym_ctx* ctx, uint32_t* tiles_used, uint32_t* overflowed);
int volym_assemble_packed(volym_ctx* ctx, const void* gathered_device, size_t stride_bytes);
/* Host-memory conveniences for callers without a device-side collective (tests, the CLI):
 * copy this context's shard out (volym_shard_bytes() bytes, padding zeroed) / assemble from
 * world shards held in host memory. */
int volym_read_shard(volym_ctx* ctx, uint8_t* out);
int volym_assemble_host(volym_ctx* ctx, const uint8_t* gathered_host);

/* --- measurement ------------------------------------------------------------------ */
int volym_stats_pass(volym_ctx* ctx, volym_stats* out);
/* n back-to-back compute passes timed with HIP events on the context's stream;
 * ms_each[n] receives each pass's duration (kernel only, inputs resident). */
int volym_time_passes(volym_ctx* ctx, uint32_t n, float* ms_each);
/* the same with ONE event pair around all n passes (no event packets between the kernels): total milliseconds */
int volym_time_batch(volym_ctx* ctx, uint32_t n, float* ms_total);
/* Self-test of the ray set-up for the frame of the last volym_update (wgsl:221-241; the reference has no counterpart): both
 * forms of its divisions -- shared reciprocals as the march kernels run them, plain IEEE divisions as the shader writes them --
 * for every pixel, compared bit for bit on the device.  out[0] = rays that differ in any bit of direction, entry, exit or hit
 * (must be 0), out[1] = rays of waves that fell back to the plain divisions, out[2] = rays.  Blocks. */
int volym_selftest_ray_setup(volym_ctx* ctx, unsigned long long out[3]);

#ifdef __cplusplus
}
#endif
#endif
