/*
 * volym_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A from-scratch scalar-float32 restatement of the importance-driven ray-march
 * of druskus20/volym (reference @ 2025-03-10) and of the host math that feeds
 * it.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this library; the product (volym_amd/) never links or calls it.
 *
 * PARITY UNPINNED: the reference holds no tests, golden images or known-answer
 * vectors for this path and cannot be built or run in
 * this image (Rust nightly + wgpu; no cargo, no Vulkan loader).  What pins
 * this file instead: hand-derived known-answer values (tests/test_oracle_kat.py),
 * closed-form analytic renders, and an independent NumPy restatement
 * (oracle/oracle_np.py) that must agree with it.
 *
 * Citations are file:line under /root/reference/.
 */
#ifndef VOLYM_ORACLE_H
#define VOLYM_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* src/gpu_resources/camera.rs:56-64  (#[repr(C, align(16))], 208 bytes,
 * column-major [[f32;4];4]: m[col][row]). */
typedef struct {
    float view_matrix[4][4];
    float projection_matrix[4][4];
    float inverse_view_proj[4][4];
    float camera_position[3];
    float _padding;
} vo_camera_uniforms;

/* src/gpu_resources/parameters.rs:55-66 / shaders/...rendering.wgsl:9-18, 32 bytes. */
typedef struct {
    float density_threshold;
    uint32_t use_cone_importance_check;
    uint32_t use_importance_coloring;
    uint32_t use_opacity;
    uint32_t use_importance_rendering;
    uint32_t use_gaussian_smoothing;
    uint32_t importance_check_ahead_steps;
    float raymarching_step_size;
} vo_parameters;

/* src/camera.rs:5-19 */
typedef struct {
    float position[3];
    float target[3];
    float up[3];
    float aspect, fovy, znear, zfar;
    float horizontal_angle, vertical_angle, distance, max_distance, min_distance;
} vo_camera;

/* Fetch counters of one render: the reference's texture fetches, counted where
 * the shader executes them (B_alg = n_vol*b_vol + n_imp + 4*W*H). */
typedef struct {
    uint64_t n_vol;   /* density fetches (march + 5-tap smoothing + 6-tap gradient) */
    uint64_t n_imp;   /* importance fetches (per step + look-ahead probes)          */
    uint64_t n_steps; /* march-loop iterations                                       */
    uint64_t n_dense; /* iterations that reached classification (rho >= threshold)  */
    uint64_t n_hit;   /* rays whose slab test hit the cube                           */
} vo_counters;

enum { VO_FILTER_NEAREST = 0, VO_FILTER_LINEAR = 1 };

/* --- elementary functions the shader leaves to the implementation ---------
 * WGSL pow/exp precision is implementation-defined; the oracle fixes ONE
 * definition built from plain IEEE f32 +,-,*,/ (no fma, no libm) so that any
 * other implementation of the same recipe is bit-identical.  DESIGN.md
 * "Elementary functions" states the recipe. */
float vo_wgsl_log2(float x);
float vo_wgsl_exp2(float z);
float vo_wgsl_pow(float x, float y);
float vo_wgsl_exp(float x);

/* --- host math -------------------------------------------------------------*/
/* src/transfer_function.rs:19-56,83-125 then src/gpu_resources/transfer_function.rs:58-69:
 * the default transfer function baked to 256 RGBA8 texels (1024 bytes). */
void vo_tf_default_lut(uint8_t lut[1024]);
/* general form: control points (iso, r,g,b) x n_rgb and (iso, a) x n_alpha, already sorted by iso. */
void vo_tf_bake(const float* rgb_points, int n_rgb, const float* alpha_points, int n_alpha,
                uint8_t lut[1024]);

/* src/camera.rs:22-45 */
void vo_camera_default(vo_camera* c, float aspect, const float position[3]);
/* src/camera.rs:47-61 */
void vo_camera_orbit(vo_camera* c, float horizontal_delta, float vertical_delta, float zoom_delta);
/* src/gpu_resources/camera.rs:66-85 (+ cgmath 0.18.0 look_at_rh / perspective / invert);
 * returns 0, or -1 when a matrix is singular. */
int vo_camera_uniforms_from(const vo_camera* c, vo_camera_uniforms* out);

/* src/gpu_resources/volume.rs:38-61 + src/gpu_resources/mod.rs:70-82: copy `len`
 * bytes of `raw` into out[nx*ny*nz], zero-padded at the end or truncated, then
 * (flip_y != 0) swap row j with row ny-1-j inside every z slice. */
void vo_prepare_volume(const uint8_t* raw, size_t len, int nx, int ny, int nz, int flip_y,
                       uint8_t* out);
/* src/demos/simple/importance.rs:148-158: label byte -> importance byte through
 * (label_value, importance) pairs, first match wins, none => 0.  In place. */
void vo_map_segments(uint8_t* data, size_t len, const uint8_t* label_values,
                     const uint8_t* importances, int n_segments);

/* --- the shader: shaders/importance_driven_volume_rendering.wgsl:213-330 --------
 * Renders pixel rows [y0, y1) of a W x H frame.  out_f32 (may be NULL): W*H*4
 * floats, pre-quantisation (C.rgb, alpha).  out_u8 (may be NULL): W*H*4 bytes as
 * an rgba8unorm store would leave them.  Rows outside [y0,y1) are untouched.
 * `threads` > 1 hands 16x16-pixel units (the reference's workgroup, wgsl:213) to that many pthreads.
 * counters may be NULL.  Returns 0, -1 on bad arguments, -2 when out of memory. */
int vo_render(const uint8_t* volume, const uint8_t* importances, int nx, int ny, int nz,
              int filter, const uint8_t* tf_lut, int tf_n,
              const vo_camera_uniforms* cam, const vo_parameters* par,
              int W, int H, int y0, int y1, int threads,
              float* out_f32, uint8_t* out_u8, vo_counters* counters);

/* The same for a list of rows (each row in [0, H); handed out as 16-pixel spans): sampled checks of large frames. */
int vo_render_rowlist(const uint8_t* volume, const uint8_t* importances, int nx, int ny, int nz,
                      int filter, const uint8_t* tf_lut, int tf_n,
                      const vo_camera_uniforms* cam, const vo_parameters* par,
                      int W, int H, const int* rows, int n_rows, int threads,
                      float* out_f32, uint8_t* out_u8, vo_counters* counters);

/* CPU-baseline timing (bench.py): the frame (rows == NULL) or the listed rows rendered `passes` times by ONE pool of
 * `threads` workers that persists over the passes; pass_seconds[passes] receives the wall time of every pass
 * (barrier to barrier, CLOCK_MONOTONIC).  Counters describe one pass. */
int vo_render_timed(const uint8_t* volume, const uint8_t* importances, int nx, int ny, int nz,
                    int filter, const uint8_t* tf_lut, int tf_n,
                    const vo_camera_uniforms* cam, const vo_parameters* par,
                    int W, int H, const int* rows, int n_rows, int threads,
                    int passes, double* pass_seconds, uint8_t* out_u8, vo_counters* counters);

/* The blit that follows the path (shaders/render.wgsl:39-43, sampler src/gpu_resources/texture.rs:84-101, REPLACE blend
 * into rgba8unorm src/render_pipeline.rs:60-64): out(x, y) = store(sample_linear_clamp(in, (x + 0.5, y + 0.5) / (in_w, in_h))).
 * The divisor is the INPUT size, as in the shader: pixels map 1:1, a larger target repeats the edge texels. */
int vo_blit(const uint8_t* in_rgba8, int in_w, int in_h, uint8_t* out_rgba8, int out_w, int out_h);

/* one pixel, for spot checks */
void vo_render_pixel(const uint8_t* volume, const uint8_t* importances, int nx, int ny, int nz,
                     int filter, const uint8_t* tf_lut, int tf_n,
                     const vo_camera_uniforms* cam, const vo_parameters* par,
                     int W, int H, int gx, int gy, float rgba[4], vo_counters* counters);

/* Checks of the product's shared-reciprocal ray set-up (volym_amd/csrc/raymarch_device.h) against this file's divisions;
 * tests/test_setup_division.py.  Mismatch counts: 0 is the claim. */
long vo_check_pixel_quotients(int w_max);
long vo_check_shared_division(uint64_t seed, long n, int rcp_skew, long* all_ones_bad);

#ifdef __cplusplus
}
#endif
#endif
