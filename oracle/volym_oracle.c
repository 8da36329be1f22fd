/*
 * volym_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 * See volym_oracle.h for scope and the "parity unpinned" statement.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (oracle/Makefile).  Every
 * float expression below is evaluated in IEEE binary32 in the order written;
 * no fused multiply-add is used anywhere, so the arithmetic can be replayed
 * exactly by any other IEEE implementation.
 *
 * Citations are file:line under /root/reference/; "wgsl" =
 * shaders/importance_driven_volume_rendering.wgsl.
 */
#include "volym_oracle.h"

#include <math.h>
#include <pthread.h>
#include <sched.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------------- */
/* Elementary functions (recipe in DESIGN.md "Elementary functions")          */
/* ------------------------------------------------------------------------- */

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

float vo_wgsl_log2(float x)
{
    /* x: positive, finite, normal.  x = m * 2^e, m in (sqrt(1/2), sqrt(2)]. */
    uint32_t bits = f2u(x);
    int e = (int)(bits >> 23) - 127;
    float m = u2f((bits & 0x007fffffu) | 0x3f800000u);
    if (m > 1.41421356f) { m = m * 0.5f; e += 1; }
    float s = (m - 1.0f) / (m + 1.0f);
    float s2 = s * s;
    /* log2(m) = (2/ln2) * (s + s^3/3 + s^5/5 + s^7/7 + s^9/9) */
    float p = 0.3205989f;          /* 2/(9 ln2) */
    p = p * s2 + 0.412198573f;       /* 2/(7 ln2) */
    p = p * s2 + 0.577078044f;       /* 2/(5 ln2) */
    p = p * s2 + 0.961796701f;       /* 2/(3 ln2) */
    p = p * s2 + 2.88539004f;       /* 2/ln2     */
    return (float)e + s * p;
}

float vo_wgsl_exp2(float z)
{
    if (!(z >= -126.0f)) return 0.0f;
    if (z > 127.0f) z = 127.0f;
    float n = rintf(z);             /* round half to even */
    float f = z - n;                /* [-0.5, 0.5] */
    float p = 1.52527336e-5f;       /* ln2^7/5040 */
    p = p * f + 1.54035297e-4f;     /* ln2^6/720  */
    p = p * f + 1.33335579e-3f;     /* ln2^5/120  */
    p = p * f + 9.61812865e-3f;     /* ln2^4/24   */
    p = p * f + 5.55041097e-2f;     /* ln2^3/6    */
    p = p * f + 2.40226507e-1f;     /* ln2^2/2    */
    p = p * f + 6.93147182e-1f;     /* ln2        */
    p = p * f + 1.0f;
    float scale = u2f((uint32_t)((int)n + 127) << 23);
    return p * scale;
}

float vo_wgsl_pow(float x, float y)
{
    if (y == 0.0f) return 1.0f;
    if (x == 0.0f) return 0.0f;     /* y > 0 on every call site (wgsl:205, :314) */
    return vo_wgsl_exp2(y * vo_wgsl_log2(x));
}

float vo_wgsl_exp(float x) { return vo_wgsl_exp2(x * 1.44269502f); }

/* ------------------------------------------------------------------------- */
/* small vector helpers (op order is part of the definition)                  */
/* ------------------------------------------------------------------------- */

typedef struct { float x, y, z; } v3;

static inline v3 v3_(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 add3(v3 a, v3 b) { return v3_(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub3(v3 a, v3 b) { return v3_(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mul3s(v3 a, float s) { return v3_(a.x * s, a.y * s, a.z * s); }
static inline v3 div3s(v3 a, float s) { return v3_(a.x / s, a.y / s, a.z / s); }
static inline float dot3(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline v3 cross3(v3 a, v3 b)
{
    return v3_(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float length3(v3 a) { return sqrtf(dot3(a, a)); }
/* WGSL normalize(v) = v / length(v) */
static inline v3 normalize3(v3 a) { return div3s(a, length3(a)); }

/* ------------------------------------------------------------------------- */
/* Transfer function                                                           */
/* ------------------------------------------------------------------------- */

/* Rust `f as u8`: truncate toward zero, saturate, NaN -> 0. */
static inline uint8_t rust_as_u8(float f)
{
    if (!(f > 0.0f)) return 0;
    if (f >= 255.0f) return 255;
    return (uint8_t)f;
}
static inline uint32_t rust_as_u32(float f)
{
    if (!(f > 0.0f)) return 0;
    if (f >= 4294967296.0f) return 0xffffffffu;
    return (uint32_t)f;
}

void vo_tf_bake(const float* rgb_points, int n_rgb, const float* alpha_points, int n_alpha,
                uint8_t lut[1024])
{
    const uint32_t max_density = 255;             /* src/transfer_function.rs:21 */
    float fv[256][4];
    memset(fv, 0, sizeof fv);                     /* src/transfer_function.rs:64 */

    /* src/transfer_function.rs:83-107: RGB windows */
    for (int w = 0; w + 1 < n_rgb; ++w) {
        const float* s = rgb_points + 4 * w;
        const float* e = rgb_points + 4 * (w + 1);
        uint32_t si = rust_as_u32(s[0] * (float)max_density);
        uint32_t ei = rust_as_u32(e[0] * (float)max_density);
        for (uint32_t x = si; x <= ei && x <= max_density; ++x) {
            float k = (ei == si) ? 0.0f : (float)(x - si) / (float)(ei - si);
            fv[x][0] = s[1] + (e[1] - s[1]) * k;
            fv[x][1] = s[2] + (e[2] - s[2]) * k;
            fv[x][2] = s[3] + (e[3] - s[3]) * k;
        }
    }
    /* src/transfer_function.rs:109-125: alpha windows */
    for (int w = 0; w + 1 < n_alpha; ++w) {
        const float* s = alpha_points + 2 * w;
        const float* e = alpha_points + 2 * (w + 1);
        uint32_t si = rust_as_u32(s[0] * (float)max_density);
        uint32_t ei = rust_as_u32(e[0] * (float)max_density);
        for (uint32_t x = si; x <= ei && x <= max_density; ++x) {
            float k = (ei == si) ? 0.0f : (float)(x - si) / (float)(ei - si);
            fv[x][3] = s[1] + (e[1] - s[1]) * k;
        }
    }
    /* src/gpu_resources/transfer_function.rs:36,58-69: 256 texels, get(i/256) */
    const uint32_t tf_size = max_density + 1;
    for (uint32_t i = 0; i < tf_size; ++i) {
        float value = (float)i / (float)tf_size;
        /* src/transfer_function.rs:127-144 */
        float idx = value * (float)max_density;
        if (idx < 0.0f) idx = 0.0f;
        if (idx > (float)max_density) idx = (float)max_density;
        float fl = floorf(idx);
        uint32_t i0 = (uint32_t)fl;
        uint32_t i1 = i0 + 1 < max_density ? i0 + 1 : max_density;
        float t = idx - fl;                       /* f32::fract for idx >= 0 */
        for (int c = 0; c < 4; ++c) {
            float v = fv[i0][c] + (fv[i1][c] - fv[i0][c]) * t;
            lut[4 * i + c] = rust_as_u8(v * 255.0f);
        }
    }
}

void vo_tf_default_lut(uint8_t lut[1024])
{
    /* src/transfer_function.rs:19-56 */
    static const float rgb[5 * 4] = {
        0.0f, 0.0f, 1.0f, 0.0f,
        0.2f, 0.0f, 1.0f, 1.0f,
        0.4f, 1.0f, 1.0f, 0.0f,
        0.6f, 1.0f, 0.0f, 1.0f,
        1.0f, 1.0f, 0.0f, 0.0f,
    };
    static const float alpha[2 * 2] = { 0.0f, 0.0f, 1.0f, 1.0f };
    vo_tf_bake(rgb, 5, alpha, 2, lut);
}

/* ------------------------------------------------------------------------- */
/* Camera (src/camera.rs + cgmath 0.18.0, restated from its published source)  */
/* ------------------------------------------------------------------------- */

void vo_camera_default(vo_camera* c, float aspect, const float position[3])
{
    /* src/camera.rs:22-45 */
    c->position[0] = position[0]; c->position[1] = position[1]; c->position[2] = position[2];
    c->target[0] = 0.5f; c->target[1] = 0.5f; c->target[2] = 0.5f;
    c->up[0] = 0.0f; c->up[1] = 1.0f; c->up[2] = 0.0f;
    c->aspect = aspect;
    c->fovy = 90.0f; c->znear = 0.01f; c->zfar = 1000.0f;
    c->horizontal_angle = 0.0f; c->vertical_angle = 0.0f; c->distance = 1.0f;
    c->max_distance = 10.0f; c->min_distance = 1.0f;
}

static inline float clampf(float v, float lo, float hi)
{
    if (v < lo) v = lo;
    if (v > hi) v = hi;
    return v;
}

void vo_camera_orbit(vo_camera* c, float hd, float vd, float zd)
{
    /* src/camera.rs:47-61 */
    c->horizontal_angle += hd;
    c->vertical_angle = clampf(c->vertical_angle + vd, -89.0f, 89.0f);
    c->distance = clampf(c->distance + zd, c->min_distance, c->max_distance);
    const float rads_per_deg = 3.14159265358979323846f / 180.0f;   /* f32::to_radians */
    float h = c->horizontal_angle * rads_per_deg;
    float v = c->vertical_angle * rads_per_deg;
    c->position[0] = c->target[0] + c->distance * sinf(h) * cosf(v);
    c->position[1] = c->target[1] + c->distance * sinf(v);
    c->position[2] = c->target[2] + c->distance * cosf(h) * cosf(v);
}

/* cgmath: dot = (x*x' + y*y') + z*z'; normalize = v * (1 / magnitude) */
static inline float cg_dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 cg_normalize(v3 a) { return mul3s(a, 1.0f / sqrtf(cg_dot(a, a))); }

static void cg_look_at_rh(const float eye[3], const float center[3], const float up[3],
                          float m[4][4])
{
    v3 e = v3_(eye[0], eye[1], eye[2]);
    v3 dir = v3_(center[0] - eye[0], center[1] - eye[1], center[2] - eye[2]);
    v3 f = cg_normalize(dir);
    v3 s = cg_normalize(cross3(f, v3_(up[0], up[1], up[2])));
    v3 u = cross3(s, f);
    m[0][0] = s.x; m[0][1] = u.x; m[0][2] = -f.x; m[0][3] = 0.0f;
    m[1][0] = s.y; m[1][1] = u.y; m[1][2] = -f.y; m[1][3] = 0.0f;
    m[2][0] = s.z; m[2][1] = u.z; m[2][2] = -f.z; m[2][3] = 0.0f;
    m[3][0] = -cg_dot(e, s); m[3][1] = -cg_dot(e, u); m[3][2] = cg_dot(e, f); m[3][3] = 1.0f;
}

static void cg_perspective_deg(float fovy_deg, float aspect, float near, float far,
                               float m[4][4])
{
    float fovy = fovy_deg * (float)(3.14159265358979323846 / 180.0);   /* Deg -> Rad */
    float f = 1.0f / tanf(fovy / 2.0f);                                /* Rad::cot     */
    memset(m, 0, 16 * sizeof(float));
    m[0][0] = f / aspect;
    m[1][1] = f;
    m[2][2] = (far + near) / (near - far);
    m[2][3] = -1.0f;
    m[3][2] = (2.0f * far * near) / (near - far);
}

/* 4x4 inverse by cofactors (adjugate / determinant), column-major m[c][r]. */
static int mat4_invert(const float m[4][4], float out[4][4])
{
    float a[16], inv[16];
    memcpy(a, m, sizeof a);
    inv[0] = a[5] * a[10] * a[15] - a[5] * a[11] * a[14] - a[9] * a[6] * a[15] +
             a[9] * a[7] * a[14] + a[13] * a[6] * a[11] - a[13] * a[7] * a[10];
    inv[4] = -a[4] * a[10] * a[15] + a[4] * a[11] * a[14] + a[8] * a[6] * a[15] -
             a[8] * a[7] * a[14] - a[12] * a[6] * a[11] + a[12] * a[7] * a[10];
    inv[8] = a[4] * a[9] * a[15] - a[4] * a[11] * a[13] - a[8] * a[5] * a[15] +
             a[8] * a[7] * a[13] + a[12] * a[5] * a[11] - a[12] * a[7] * a[9];
    inv[12] = -a[4] * a[9] * a[14] + a[4] * a[10] * a[13] + a[8] * a[5] * a[14] -
              a[8] * a[6] * a[13] - a[12] * a[5] * a[10] + a[12] * a[6] * a[9];
    inv[1] = -a[1] * a[10] * a[15] + a[1] * a[11] * a[14] + a[9] * a[2] * a[15] -
             a[9] * a[3] * a[14] - a[13] * a[2] * a[11] + a[13] * a[3] * a[10];
    inv[5] = a[0] * a[10] * a[15] - a[0] * a[11] * a[14] - a[8] * a[2] * a[15] +
             a[8] * a[3] * a[14] + a[12] * a[2] * a[11] - a[12] * a[3] * a[10];
    inv[9] = -a[0] * a[9] * a[15] + a[0] * a[11] * a[13] + a[8] * a[1] * a[15] -
             a[8] * a[3] * a[13] - a[12] * a[1] * a[11] + a[12] * a[3] * a[9];
    inv[13] = a[0] * a[9] * a[14] - a[0] * a[10] * a[13] - a[8] * a[1] * a[14] +
              a[8] * a[2] * a[13] + a[12] * a[1] * a[10] - a[12] * a[2] * a[9];
    inv[2] = a[1] * a[6] * a[15] - a[1] * a[7] * a[14] - a[5] * a[2] * a[15] +
             a[5] * a[3] * a[14] + a[13] * a[2] * a[7] - a[13] * a[3] * a[6];
    inv[6] = -a[0] * a[6] * a[15] + a[0] * a[7] * a[14] + a[4] * a[2] * a[15] -
             a[4] * a[3] * a[14] - a[12] * a[2] * a[7] + a[12] * a[3] * a[6];
    inv[10] = a[0] * a[5] * a[15] - a[0] * a[7] * a[13] - a[4] * a[1] * a[15] +
              a[4] * a[3] * a[13] + a[12] * a[1] * a[7] - a[12] * a[3] * a[5];
    inv[14] = -a[0] * a[5] * a[14] + a[0] * a[6] * a[13] + a[4] * a[1] * a[14] -
              a[4] * a[2] * a[13] - a[12] * a[1] * a[6] + a[12] * a[2] * a[5];
    inv[3] = -a[1] * a[6] * a[11] + a[1] * a[7] * a[10] + a[5] * a[2] * a[11] -
             a[5] * a[3] * a[10] - a[9] * a[2] * a[7] + a[9] * a[3] * a[6];
    inv[7] = a[0] * a[6] * a[11] - a[0] * a[7] * a[10] - a[4] * a[2] * a[11] +
             a[4] * a[3] * a[10] + a[8] * a[2] * a[7] - a[8] * a[3] * a[6];
    inv[11] = -a[0] * a[5] * a[11] + a[0] * a[7] * a[9] + a[4] * a[1] * a[11] -
              a[4] * a[3] * a[9] - a[8] * a[1] * a[7] + a[8] * a[3] * a[5];
    inv[15] = a[0] * a[5] * a[10] - a[0] * a[6] * a[9] - a[4] * a[1] * a[10] +
              a[4] * a[2] * a[9] + a[8] * a[1] * a[6] - a[8] * a[2] * a[5];
    float det = a[0] * inv[0] + a[1] * inv[4] + a[2] * inv[8] + a[3] * inv[12];
    if (det == 0.0f) return -1;
    float inv_det = 1.0f / det;
    float* o = &out[0][0];
    for (int i = 0; i < 16; ++i) o[i] = inv[i] * inv_det;
    return 0;
}

/* column-major product: (A*B)[c][r] = sum_k A[k][r] * B[c][k] */
static void mat4_mul(const float A[4][4], const float B[4][4], float out[4][4])
{
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r)
            out[c][r] = A[0][r] * B[c][0] + A[1][r] * B[c][1] + A[2][r] * B[c][2] +
                        A[3][r] * B[c][3];
}

int vo_camera_uniforms_from(const vo_camera* c, vo_camera_uniforms* out)
{
    /* src/gpu_resources/camera.rs:66-85; src/camera.rs:63-73 */
    float view[4][4], proj[4][4], vinv[4][4], pinv[4][4];
    cg_perspective_deg(c->fovy, c->aspect, c->znear, c->zfar, proj);
    cg_look_at_rh(c->position, c->target, c->up, view);
    if (mat4_invert(view, vinv) != 0) return -1;
    if (mat4_invert(proj, pinv) != 0) return -1;
    memcpy(out->view_matrix, view, sizeof view);
    memcpy(out->projection_matrix, proj, sizeof proj);
    mat4_mul(vinv, pinv, out->inverse_view_proj);
    out->camera_position[0] = c->position[0];
    out->camera_position[1] = c->position[1];
    out->camera_position[2] = c->position[2];
    out->_padding = 0.0f;
    return 0;
}

/* ------------------------------------------------------------------------- */
/* Volume preparation                                                          */
/* ------------------------------------------------------------------------- */

void vo_prepare_volume(const uint8_t* raw, size_t len, int nx, int ny, int nz, int flip_y,
                       uint8_t* out)
{
    /* src/gpu_resources/volume.rs:38-55 */
    size_t want = (size_t)nx * ny * nz;
    size_t ncopy = len < want ? len : want;
    memcpy(out, raw, ncopy);
    if (ncopy < want) memset(out + ncopy, 0, want - ncopy);
    /* src/gpu_resources/mod.rs:70-82 */
    if (flip_y) {
        for (int k = 0; k < nz; ++k)
            for (int j = 0; j < ny / 2; ++j) {
                uint8_t* top = out + (size_t)k * nx * ny + (size_t)j * nx;
                uint8_t* bot = out + (size_t)k * nx * ny + (size_t)(ny - j - 1) * nx;
                for (int i = 0; i < nx; ++i) { uint8_t t = top[i]; top[i] = bot[i]; bot[i] = t; }
            }
    }
}

void vo_map_segments(uint8_t* data, size_t len, const uint8_t* label_values,
                     const uint8_t* importances, int n_segments)
{
    /* src/demos/simple/importance.rs:148-158 */
    for (size_t i = 0; i < len; ++i) {
        uint8_t v = 0;
        for (int s = 0; s < n_segments; ++s)
            if (label_values[s] == data[i]) { v = importances[s]; break; }
        data[i] = v;
    }
}

/* ------------------------------------------------------------------------- */
/* Texture sampling (Vulkan texel-selection rules)  */
/* ------------------------------------------------------------------------- */

typedef struct {
    const uint8_t* vol;
    const uint8_t* imp;
    int nx, ny, nz;
    int filter;
    const uint8_t* lut;
    int tf_n;
    const vo_camera_uniforms* cam;
    const vo_parameters* par;
    int W, H;
    float gauss_w[5];
    float cone_cos[8], cone_sin[8];
} vo_scene;

static inline float unorm8(uint8_t b) { return (float)b / 255.0f; }

/* nearest filter, ClampToEdge: i = clamp(floor(u * n), 0, n-1) */
static inline int texel_nearest(float u, int n)
{
    float f = floorf(u * (float)n);
    if (!(f >= 0.0f)) f = 0.0f;
    float hi = (float)(n - 1);
    if (f > hi) f = hi;
    return (int)f;
}

static inline float fetch_nearest(const uint8_t* t, int nx, int ny, int nz, v3 p)
{
    int ix = texel_nearest(p.x, nx), iy = texel_nearest(p.y, ny), iz = texel_nearest(p.z, nz);
    return unorm8(t[(size_t)ix + (size_t)nx * ((size_t)iy + (size_t)ny * (size_t)iz)]);
}

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* linear filter, ClampToEdge: x = u*n - 0.5, i0 = floor(x), w = x - i0 */
static inline void texel_linear(float u, int n, int* i0, int* i1, float* w)
{
    float x = u * (float)n - 0.5f;
    float fl = floorf(x);
    *w = x - fl;
    /* keep the float->int conversion in range for far-away probes */
    if (!(fl >= -2.0f)) fl = -2.0f;
    if (fl > (float)n) fl = (float)n;
    int i = (int)fl;
    *i0 = clampi(i, 0, n - 1);
    *i1 = clampi(i + 1, 0, n - 1);
}

static inline float fetch_linear(const uint8_t* t, int nx, int ny, int nz, v3 p)
{
    int x0, x1, y0, y1, z0, z1;
    float fx, fy, fz;
    texel_linear(p.x, nx, &x0, &x1, &fx);
    texel_linear(p.y, ny, &y0, &y1, &fy);
    texel_linear(p.z, nz, &z0, &z1, &fz);
#define T(X, Y, Z) unorm8(t[(size_t)(X) + (size_t)nx * ((size_t)(Y) + (size_t)ny * (size_t)(Z))])
    float c00 = T(x0, y0, z0) * (1.0f - fx) + T(x1, y0, z0) * fx;
    float c10 = T(x0, y1, z0) * (1.0f - fx) + T(x1, y1, z0) * fx;
    float c01 = T(x0, y0, z1) * (1.0f - fx) + T(x1, y0, z1) * fx;
    float c11 = T(x0, y1, z1) * (1.0f - fx) + T(x1, y1, z1) * fx;
#undef T
    float c0 = c00 * (1.0f - fy) + c10 * fy;
    float c1 = c01 * (1.0f - fy) + c11 * fy;
    return c0 * (1.0f - fz) + c1 * fz;
}

/* volume_texture + volume_sampler (src/gpu_resources/volume.rs:75,92-95) */
static inline float sample_volume(const vo_scene* s, v3 p, vo_counters* k)
{
    k->n_vol++;
    return s->filter == VO_FILTER_LINEAR ? fetch_linear(s->vol, s->nx, s->ny, s->nz, p)
                                         : fetch_nearest(s->vol, s->nx, s->ny, s->nz, p);
}

/* importances_texture + importances_sampler (src/demos/simple/importance.rs:105,122-131) */
static inline float sample_importance(const vo_scene* s, v3 p, vo_counters* k)
{
    k->n_imp++;
    return fetch_nearest(s->imp, s->nx, s->ny, s->nz, p);
}

/* transfer_function_texture + sampler, Linear/Clamp (src/gpu_resources/transfer_function.rs:92-101) */
static inline void sample_tf(const vo_scene* s, float u, float out[4])
{
    int i0, i1;
    float w;
    texel_linear(u, s->tf_n, &i0, &i1, &w);
    for (int c = 0; c < 4; ++c)
        out[c] = unorm8(s->lut[4 * i0 + c]) * (1.0f - w) + unorm8(s->lut[4 * i1 + c]) * w;
}

/* ------------------------------------------------------------------------- */
/* The shader                                                                  */
/* ------------------------------------------------------------------------- */

static inline int any_outside01(v3 p)
{
    return (p.x < 0.0f) || (p.y < 0.0f) || (p.z < 0.0f) || (p.x > 1.0f) || (p.y > 1.0f) ||
           (p.z > 1.0f);
}

/* wgsl:44-75 */
static float sample_volume_smoothed(const vo_scene* s, v3 pos, v3 dir, vo_counters* k)
{
    float sum = 0.0f, weight_sum = 0.0f;
    for (int i = -2; i <= 2; ++i) {
        float offset = (float)i * 0.005f;
        v3 sp = add3(pos, mul3s(dir, offset));
        if (any_outside01(sp)) continue;
        float weight = s->gauss_w[i + 2];
        float sample = sample_volume(s, sp, k);
        sum += sample * weight;
        weight_sum += weight;
    }
    return sum / weight_sum;
}

/* wgsl:141-160 */
static int ahead_straight(const vo_scene* s, v3 current_pos, v3 dir, float max_distance,
                          vo_counters* k)
{
    v3 pos = current_pos;
    int check_steps = (int)s->par->importance_check_ahead_steps;
    float step = (max_distance - length3(current_pos)) / (float)check_steps;
    for (int i = 0; i < check_steps; ++i) {
        pos = add3(pos, mul3s(dir, step));
        float importance = sample_importance(s, pos, k);
        if (importance >= 0.5f) return 1;
    }
    return 0;
}

/* wgsl:94-139 */
static int ahead_cone(const vo_scene* s, v3 current_pos, v3 main_dir, float max_distance,
                      vo_counters* k)
{
    int check_steps = (int)s->par->importance_check_ahead_steps;
    float step = (max_distance - length3(current_pos)) / (float)check_steps;
    const float cone_angle = 0.2f;
    for (int c = 0; c < 8; ++c) {
        /* wgsl:94-106 */
        v3 up = v3_(0.0f, 1.0f, 0.0f);
        v3 right = normalize3(cross3(main_dir, up));
        v3 new_up = cross3(main_dir, right);
        float x_offset = s->cone_cos[c] * cone_angle;
        float y_offset = s->cone_sin[c] * cone_angle;
        v3 sd = normalize3(add3(add3(main_dir, mul3s(right, x_offset)), mul3s(new_up, y_offset)));
        v3 pos = current_pos;
        for (int i = 0; i < check_steps; ++i) {
            pos = add3(pos, mul3s(sd, step));
            if (any_outside01(pos)) break;
            float importance = sample_importance(s, pos, k);
            if (importance >= 0.5f) return 1;
        }
    }
    return 0;
}

/* wgsl:181-211 */
static v3 blinn_phong_shade(const vo_scene* s, v3 pos, v3 color, vo_counters* k)
{
    const float o = 0.01f;
    float gx = (sample_volume(s, v3_(pos.x + o, pos.y, pos.z), k) -
                sample_volume(s, v3_(pos.x - o, pos.y, pos.z), k)) / (2.0f * o);
    float gy = (sample_volume(s, v3_(pos.x, pos.y + o, pos.z), k) -
                sample_volume(s, v3_(pos.x, pos.y - o, pos.z), k)) / (2.0f * o);
    float gz = (sample_volume(s, v3_(pos.x, pos.y, pos.z + o), k) -
                sample_volume(s, v3_(pos.x, pos.y, pos.z - o), k)) / (2.0f * o);
    v3 n = normalize3(v3_(gx, gy, gz));
    if (length3(n) > 0.0f) {                      /* false for the NaN of a zero gradient */
        v3 eye = v3_(s->cam->camera_position[0], s->cam->camera_position[1],
                     s->cam->camera_position[2]);
        v3 L = normalize3(v3_(1.0f, 1.0f, 1.0f));
        v3 E = normalize3(sub3(eye, pos));
        v3 Hh = normalize3(add3(E, L));
        float ambient = 0.2f;
        float diffuse = fmaxf(0.0f, dot3(n, L));
        float specular = vo_wgsl_pow(fmaxf(0.0f, dot3(Hh, n)), 24.0f);
        float kd = ambient + 0.7f * diffuse;
        float ks = 0.4f * specular;               /* vec3(1)*0.4*specular */
        return v3_(color.x * kd + ks, color.y * kd + ks, color.z * kd + ks);
    }
    return color;
}

static void render_pixel(const vo_scene* s, int gx, int gy, float out[4], vo_counters* k)
{
    const vo_camera_uniforms* cam = s->cam;
    const vo_parameters* par = s->par;
    /* wgsl:221-234 */
    float scx = (float)gx / (float)s->W;
    float scy = (float)gy / (float)s->H;
    float ndx = scx * 2.0f - 1.0f;
    float ndy = 1.0f - scy * 2.0f;
    v3 origin = v3_(cam->camera_position[0], cam->camera_position[1], cam->camera_position[2]);
    float wp[4];
    for (int r = 0; r < 4; ++r) {
        const float(*m)[4] = cam->inverse_view_proj;
        wp[r] = ((m[0][r] * ndx + m[1][r] * ndy) + m[2][r] * 0.0f) + m[3][r] * 1.0f;
    }
    v3 world = v3_(wp[0] / wp[3], wp[1] / wp[3], wp[2] / wp[3]);
    v3 dir = normalize3(sub3(world, origin));

    /* wgsl:162-179 */
    float t1x = (0.0f - origin.x) / dir.x, t2x = (1.0f - origin.x) / dir.x;
    float t1y = (0.0f - origin.y) / dir.y, t2y = (1.0f - origin.y) / dir.y;
    float t1z = (0.0f - origin.z) / dir.z, t2z = (1.0f - origin.z) / dir.z;
    float entry = fmaxf(fmaxf(fminf(t1x, t2x), fminf(t1y, t2y)), fminf(t1z, t2z));
    float exit_ = fminf(fminf(fmaxf(t1x, t2x), fmaxf(t1y, t2y)), fmaxf(t1z, t2z));
    float t_entry = fmaxf(entry, 0.0f);
    float t_exit = fmaxf(exit_, 0.0f);

    /* wgsl:238-241 */
    if (t_exit <= t_entry) { out[0] = out[1] = out[2] = 0.0f; out[3] = 1.0f; return; }
    k->n_hit++;

    /* wgsl:243-249 */
    float base_step = par->raymarching_step_size;
    float min_step = base_step * 0.25f;
    float cur_step = base_step;
    v3 acc = v3_(0.0f, 0.0f, 0.0f);
    float acc_a = 0.0f;
    float t = t_entry;

    while (t < t_exit && acc_a < 0.95f) {          /* wgsl:250 */
        k->n_steps++;
        v3 pos = add3(origin, mul3s(dir, t));      /* wgsl:251 */
        float density;
        if (par->use_gaussian_smoothing == 1) density = sample_volume_smoothed(s, pos, dir, k);
        else density = sample_volume(s, pos, k);   /* wgsl:253-259 */
        float importance = sample_importance(s, pos, k); /* wgsl:260 */

        if (density >= par->density_threshold) cur_step = min_step;   /* wgsl:263-269 */
        else cur_step = fminf(base_step, cur_step * 1.5f);
        /* wgsl:271-274 tests `density < threshold`, the complement of wgsl:263 for every number.  A NaN density (smoothing
         * on and all five taps outside the cube: 0/0, only on rays that graze a cube edge) would pass neither test and fall
         * through to the shading with the *grown* step size.  WGSL leaves that case open ("implementations may assume that
         * NaNs and infinities are not present at runtime ... an undefined value is produced instead"), so there is nothing
         * to be faithful to; this restatement, oracle_np.py and the HIP kernels all take "a NaN density is not dense". */
        if (!(density >= par->density_threshold)) { t += cur_step; continue; }
        k->n_dense++;

        float ca[4];
        int use_alpha = par->use_opacity == 1;     /* wgsl:277 */
        if (par->use_importance_coloring == 1) {   /* wgsl:279-281, 83-92 */
            ca[0] = fminf(importance * 1.5f, 1.0f);
            ca[1] = (1.0f - importance) * 1.2f;
            ca[2] = 0.2f;
            ca[3] = importance;
            use_alpha = 1;
        } else {
            if (par->use_importance_rendering == 1) {   /* wgsl:283-295 */
                int ahead = par->use_cone_importance_check == 1
                                ? ahead_cone(s, pos, dir, t_exit, k)
                                : ahead_straight(s, pos, dir, t_exit, k);
                if (importance < 1.0f && ahead) { t += cur_step; continue; }
            }
            sample_tf(s, density, ca);             /* wgsl:297-303 */
        }

        v3 shaded = blinn_phong_shade(s, pos, v3_(ca[0], ca[1], ca[2]), k); /* wgsl:306-311 */

        if (use_alpha) {                           /* wgsl:313-318 */
            float alpha = 1.0f - vo_wgsl_pow(1.0f - ca[3], cur_step * 100.0f);
            float w = (1.0f - acc_a) * alpha;
            acc = add3(acc, mul3s(shaded, w));
            acc_a += w;
        } else {                                   /* wgsl:319-323 */
            acc = shaded;
            acc_a = 1.0f;
            break;
        }
        t += cur_step;                             /* wgsl:325 */
    }
    out[0] = acc.x; out[1] = acc.y; out[2] = acc.z; out[3] = acc_a;   /* wgsl:328-329 */
}

/* rgba8unorm store (src/gpu_resources/texture.rs:51): clamp, scale, round to nearest */
static inline uint8_t to_unorm8(float v)
{
    if (!(v > 0.0f)) return 0;
    if (v >= 1.0f) return 255;
    return (uint8_t)floorf(v * 255.0f + 0.5f);
}

/* cos/sin of (s/8) * 2 * 3.14159 (wgsl:99-103), f32, as glibc 2.35 cosf/sinf return them;
 * tests/test_oracle_kat.py re-derives them in float64. */
static const float k_cone_cos[8] = {
    0x1p+0f, 0x1.6a09f6p-1f, 0x1.54442ep-20f, -0x1.6a09bap-1f,
    -0x1p+0f, -0x1.6a0a32p-1f, -0x1.fe6644p-19f, 0x1.6a097ep-1f };
static const float k_cone_sin[8] = {
    0x0p+0f, 0x1.6a09d8p-1f, 0x1p+0f, 0x1.6a0a14p-1f,
    0x1.54442ep-19f, -0x1.6a099cp-1f, -0x1p+0f, -0x1.6a0a5p-1f };

static void scene_init(vo_scene* s, const uint8_t* volume, const uint8_t* importances, int nx,
                       int ny, int nz, int filter, const uint8_t* lut, int tf_n,
                       const vo_camera_uniforms* cam, const vo_parameters* par, int W, int H)
{
    s->vol = volume; s->imp = importances; s->nx = nx; s->ny = ny; s->nz = nz;
    s->filter = filter; s->lut = lut; s->tf_n = tf_n; s->cam = cam; s->par = par;
    s->W = W; s->H = H;
    const float sigma = 1.5f;                      /* wgsl:255 */
    for (int i = -2; i <= 2; ++i) {                /* wgsl:44-46, 59, 67 */
        float x = (float)i * 0.005f;
        s->gauss_w[i + 2] = vo_wgsl_exp(-(x * x) / (2.0f * sigma * sigma));
    }
    for (int c = 0; c < 8; ++c) { s->cone_cos[c] = k_cone_cos[c]; s->cone_sin[c] = k_cone_sin[c]; }
}

void vo_render_pixel(const uint8_t* volume, const uint8_t* importances, int nx, int ny, int nz,
                     int filter, const uint8_t* tf_lut, int tf_n, const vo_camera_uniforms* cam,
                     const vo_parameters* par, int W, int H, int gx, int gy, float rgba[4],
                     vo_counters* counters)
{
    vo_scene s;
    vo_counters k = { 0, 0, 0, 0, 0 };
    scene_init(&s, volume, importances, nx, ny, nz, filter, tf_lut, tf_n, cam, par, W, H);
    render_pixel(&s, gx, gy, rgba, &k);
    if (counters) *counters = k;
}

/* Work distribution of vo_render / vo_render_rowlist: units of 16x16 pixels (the reference's workgroup,
 * wgsl:213) handed out through one atomic ticket; every worker keeps its counters in
 * a cache line of its own and they are summed after the join. */
typedef struct {
    const vo_scene* s;
    int y0, y1;                 /* rows [y0, y1) (rowlist == NULL) */
    const int* rowlist;         /* or: these rows, each as 16-pixel-wide units */
    int n_rows;
    int units_x, n_units;
    int* ticket;                /* one ticket counter per pass */
    int passes;                 /* the frame is rendered this many times (CPU-baseline timing); 1 otherwise */
    pthread_barrier_t* bar;     /* passes > 1: between the passes */
    double* pass_seconds;       /* passes > 1: wall time of every pass, written by worker 0 */
    int worker;
    int* go;                    /* passes > 1: 0 wait, 1 start, -1 give up (the pool could not be set up) */
    float* out_f32;
    uint8_t* out_u8;
    vo_counters k;
    char pad[64];
} __attribute__((aligned(64))) vo_job;

static void render_span(const vo_job* j, vo_counters* k, int y, int x0, int x1)
{
    const vo_scene* s = j->s;
    for (int x = x0; x < x1; ++x) {
        float px[4];
        render_pixel(s, x, y, px, k);
        size_t o = 4 * ((size_t)y * s->W + x);
        if (j->out_f32) memcpy(j->out_f32 + o, px, sizeof px);
        if (j->out_u8)
            for (int c = 0; c < 4; ++c) j->out_u8[o + c] = to_unorm8(px[c]);
    }
}

static void* render_units(void* arg)
{
    vo_job* j = (vo_job*)arg;
    const vo_scene* s = j->s;
    vo_counters k = { 0, 0, 0, 0, 0 };     /* thread-local: no shared cache line inside the loop */
    if (j->go && j->worker != 0) {
        int g;
        while ((g = __atomic_load_n(j->go, __ATOMIC_ACQUIRE)) == 0) sched_yield();
        if (g < 0) return NULL;
    }
    for (int pass = 0; pass < j->passes; ++pass) {
        struct timespec t0, t1;
        if (j->bar) {
            pthread_barrier_wait(j->bar);
            if (j->worker == 0) clock_gettime(CLOCK_MONOTONIC, &t0);
        }
        if (pass > 0) memset(&k, 0, sizeof k);                 /* counters describe one frame */
        for (;;) {
            const int u = __atomic_fetch_add(&j->ticket[pass], 1, __ATOMIC_RELAXED);
            if (u >= j->n_units) break;
            const int ux = u % j->units_x, uy = u / j->units_x;
            const int x0 = ux * 16, x1 = x0 + 16 < s->W ? x0 + 16 : s->W;
            if (j->rowlist) {
                render_span(j, &k, j->rowlist[uy], x0, x1);
            } else {
                const int ya = j->y0 + uy * 16, yb = ya + 16 < j->y1 ? ya + 16 : j->y1;
                for (int y = ya; y < yb; ++y) render_span(j, &k, y, x0, x1);
            }
        }
        if (j->bar) {
            pthread_barrier_wait(j->bar);
            if (j->worker == 0) {
                clock_gettime(CLOCK_MONOTONIC, &t1);
                j->pass_seconds[pass] = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
            }
        }
    }
    j->k = k;
    return NULL;
}

static int render_parallel(const vo_scene* s, int y0, int y1, const int* rowlist, int n_rows, int threads,
                           float* out_f32, uint8_t* out_u8, vo_counters* counters, int passes, double* pass_seconds)
{
    if (passes < 1 || passes > 4096) return -1;
    if (threads < 1) threads = 1;
    if (threads > 512) threads = 512;
    const int units_x = (s->W + 15) / 16;
    const int units_y = rowlist ? n_rows : (y1 - y0 + 15) / 16;
    const int n_units = units_x * (units_y > 0 ? units_y : 0);
    if (threads > n_units) threads = n_units > 0 ? n_units : 1;
    int* ticket = (int*)calloc((size_t)passes, sizeof(int));
    if (!ticket) return -2;
    vo_job* jobs = (vo_job*)aligned_alloc(64, sizeof(vo_job) * (size_t)threads);
    pthread_t* tids = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)threads);
    if (!jobs || !tids) { free(jobs); free(tids); free(ticket); return -2; }
    for (int i = 0; i < threads; ++i) {
        memset(&jobs[i], 0, sizeof jobs[i]);
        jobs[i].s = s; jobs[i].y0 = y0; jobs[i].y1 = y1; jobs[i].rowlist = rowlist; jobs[i].n_rows = n_rows;
        jobs[i].units_x = units_x; jobs[i].n_units = n_units; jobs[i].ticket = ticket;
        jobs[i].passes = passes; jobs[i].worker = i;
        jobs[i].out_f32 = out_f32; jobs[i].out_u8 = out_u8;
    }
    int started = 0;
    pthread_barrier_t bar;
    int go = 0;
    if (pass_seconds) {
        /* the pool persists over the passes.  The barrier is sized by the workers that really exist: they are created
         * first and wait for `go`; a pthread_create failure only makes the pool smaller. */
        for (int i = 0; i < threads; ++i) { jobs[i].bar = &bar; jobs[i].pass_seconds = pass_seconds; jobs[i].go = &go; }
        for (int i = 1; i < threads; ++i) {
            if (pthread_create(&tids[started], NULL, render_units, &jobs[i]) != 0) break;
            started++;
        }
        if (pthread_barrier_init(&bar, NULL, (unsigned)started + 1u) != 0) {
            __atomic_store_n(&go, -1, __ATOMIC_RELEASE);      /* workers leave without marching */
            for (int i = 0; i < started; ++i) pthread_join(tids[i], NULL);
            free(jobs); free(tids); free(ticket);
            return -2;
        }
        __atomic_store_n(&go, 1, __ATOMIC_RELEASE);
    } else if (threads > 1) {
        for (int i = 1; i < threads; ++i) {
            /* a thread that cannot be created is not joined; the others (and this one) drain the ticket */
            if (pthread_create(&tids[started], NULL, render_units, &jobs[i]) != 0) break;
            started++;
        }
    }
    render_units(&jobs[0]);
    for (int i = 0; i < started; ++i) pthread_join(tids[i], NULL);
    if (pass_seconds) pthread_barrier_destroy(&bar);
    if (counters) {
        memset(counters, 0, sizeof *counters);
        for (int i = 0; i < threads; ++i) {
            counters->n_vol += jobs[i].k.n_vol; counters->n_imp += jobs[i].k.n_imp;
            counters->n_steps += jobs[i].k.n_steps; counters->n_dense += jobs[i].k.n_dense;
            counters->n_hit += jobs[i].k.n_hit;
        }
    }
    free(jobs); free(tids); free(ticket);
    return 0;
}

int vo_render(const uint8_t* volume, const uint8_t* importances, int nx, int ny, int nz,
              int filter, const uint8_t* tf_lut, int tf_n, const vo_camera_uniforms* cam,
              const vo_parameters* par, int W, int H, int y0, int y1, int threads, float* out_f32,
              uint8_t* out_u8, vo_counters* counters)
{
    if (!volume || !importances || !tf_lut || !cam || !par) return -1;
    if (nx <= 0 || ny <= 0 || nz <= 0 || W <= 0 || H <= 0 || tf_n <= 0) return -1;
    if (y0 < 0) y0 = 0;
    if (y1 > H) y1 = H;
    vo_scene s;
    scene_init(&s, volume, importances, nx, ny, nz, filter, tf_lut, tf_n, cam, par, W, H);
    return render_parallel(&s, y0, y1, NULL, 0, threads, out_f32, out_u8, counters, 1, NULL);
}

int vo_render_rowlist(const uint8_t* volume, const uint8_t* importances, int nx, int ny, int nz,
                      int filter, const uint8_t* tf_lut, int tf_n, const vo_camera_uniforms* cam,
                      const vo_parameters* par, int W, int H, const int* rows, int n_rows, int threads,
                      float* out_f32, uint8_t* out_u8, vo_counters* counters)
{
    if (!volume || !importances || !tf_lut || !cam || !par || !rows) return -1;
    if (nx <= 0 || ny <= 0 || nz <= 0 || W <= 0 || H <= 0 || tf_n <= 0 || n_rows < 0) return -1;
    for (int i = 0; i < n_rows; ++i) if (rows[i] < 0 || rows[i] >= H) return -1;
    vo_scene s;
    scene_init(&s, volume, importances, nx, ny, nz, filter, tf_lut, tf_n, cam, par, W, H);
    return render_parallel(&s, 0, 0, rows, n_rows, threads, out_f32, out_u8, counters, 1, NULL);
}

int vo_render_timed(const uint8_t* volume, const uint8_t* importances, int nx, int ny, int nz,
                    int filter, const uint8_t* tf_lut, int tf_n, const vo_camera_uniforms* cam,
                    const vo_parameters* par, int W, int H, const int* rows, int n_rows, int threads,
                    int passes, double* pass_seconds, uint8_t* out_u8, vo_counters* counters)
{
    if (!volume || !importances || !tf_lut || !cam || !par || !pass_seconds) return -1;
    if (nx <= 0 || ny <= 0 || nz <= 0 || W <= 0 || H <= 0 || tf_n <= 0 || n_rows < 0) return -1;
    for (int i = 0; rows && i < n_rows; ++i) if (rows[i] < 0 || rows[i] >= H) return -1;
    vo_scene s;
    scene_init(&s, volume, importances, nx, ny, nz, filter, tf_lut, tf_n, cam, par, W, H);
    return render_parallel(&s, 0, rows ? 0 : H, rows, rows ? n_rows : 0, threads, NULL, out_u8, counters, passes, pass_seconds);
}


/* ------------------------------------------------------------------------- */
/* Blit: shaders/render.wgsl:39-43 through the sampler of                      */
/* src/gpu_resources/texture.rs:84-101 into an rgba8unorm target               */
/* (src/render_pipeline.rs:60-64 BlendState::REPLACE).                         */
/* ------------------------------------------------------------------------- */
static void blit_axis(float frag, float fn, int n, int* i0, int* i1, float* w)
{
    float u = frag / fn;                 /* wgsl:41  uv = frag_coord.xy / textureDimensions(input_texture) */
    float x = u * fn - 0.5f;             /* Vulkan linear filtering: unnormalised coordinate minus 0.5 */
    float fl = floorf(x);
    *w = x - fl;
    int i = (int)fl;
    *i0 = i < 0 ? 0 : (i > n - 1 ? n - 1 : i);              /* AddressMode::ClampToEdge */
    *i1 = i + 1 < 0 ? 0 : (i + 1 > n - 1 ? n - 1 : i + 1);
}

int vo_blit(const uint8_t* in_rgba8, int in_w, int in_h, uint8_t* out_rgba8, int out_w, int out_h)
{
    if (!in_rgba8 || !out_rgba8 || in_w <= 0 || in_h <= 0 || out_w <= 0 || out_h <= 0) return -1;
    for (int y = 0; y < out_h; ++y) {
        int y0, y1; float wy;
        blit_axis((float)y + 0.5f, (float)in_h, in_h, &y0, &y1, &wy);      /* frag_coord = pixel centre */
        for (int x = 0; x < out_w; ++x) {
            int x0, x1; float wx;
            blit_axis((float)x + 0.5f, (float)in_w, in_w, &x0, &x1, &wx);
            for (int ch = 0; ch < 4; ++ch) {
                float a = (float)in_rgba8[4 * ((size_t)y0 * in_w + x0) + ch] / 255.0f;
                float b = (float)in_rgba8[4 * ((size_t)y0 * in_w + x1) + ch] / 255.0f;
                float c = (float)in_rgba8[4 * ((size_t)y1 * in_w + x0) + ch] / 255.0f;
                float d = (float)in_rgba8[4 * ((size_t)y1 * in_w + x1) + ch] / 255.0f;
                float top = a * (1.0f - wx) + b * wx, bot = c * (1.0f - wx) + d * wx;
                out_rgba8[4 * ((size_t)y * out_w + x) + ch] = to_unorm8(top * (1.0f - wy) + bot * wy);
            }
        }
    }
    return 0;
}

/* ---- checks of the product's ray set-up arithmetic (volym_amd/csrc/raymarch_device.h: div_pixel, rcp_refined / div_by) ----
 * The product shares refined reciprocals between the divisions of wgsl:221-241 that have a common denominator.  These two
 * functions restate its sequences with fmaf and compare them with this file's `/` (what vo_render uses). */

/* g / W with r = RN(1 / W) and one correction, for every 0 <= g < W <= w_max: number of pairs whose bits differ from g / W */
long vo_check_pixel_quotients(int w_max)
{
    long bad = 0;
    for (int W = 1; W <= w_max; ++W) {
        const float fw = (float)W, r = (float)(1.0 / (double)W);
        for (int g = 0; g < W; ++g) {
            const float a = (float)g, ref = a / fw;
            const float q = a * r;
            const float q1 = fmaf(fmaf(-fw, q, a), r, q);
            if (memcmp(&q1, &ref, 4) != 0) bad++;
        }
    }
    return bad;
}

/* n pseudo-random pairs with magnitudes in [2^-40, 2^40) (every 8th denominator with a significand of all ones or all ones but
 * the last bit, every 16th a power of two): the hardware's sequence -- reciprocal estimate, two refinements, quotient, two
 * corrections -- against num / den.  The estimate is RN(1 / den) moved by rcp_skew units in the last place (v_rcp_f32 is
 * specified to 1 ulp).  Returns the mismatches among denominators whose significand is not 0x7fffff; *all_ones_bad counts the
 * mismatches among those that are (with a skewed estimate the textbook exception: 1 / den just above a rounding boundary). */
long vo_check_shared_division(uint64_t seed, long n, int rcp_skew, long* all_ones_bad)
{
    uint64_t s = seed ? seed : 88172645463325252ull;
    long bad = 0, bad_ones = 0;
    for (long i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const uint64_t x = s;
        const uint32_t ed = 87u + (uint32_t)(x % 80u), en = 87u + (uint32_t)((x >> 8) % 80u);
        uint32_t md = (uint32_t)(x >> 16) & 0x7fffffu;
        const uint32_t mn = (uint32_t)(x >> 40) & 0x7fffffu;
        if ((i & 7) == 0) md = (i & 8) ? 0x7fffffu : 0x7ffffeu;
        if ((i & 15) == 1) md = 0u;
        const uint32_t db = (ed << 23) | md | ((uint32_t)(x >> 63) << 31), nb = (en << 23) | mn | ((uint32_t)((x >> 62) & 1u) << 31);
        float d, a;
        memcpy(&d, &db, 4); memcpy(&a, &nb, 4);
        const float ref = a / d;
        float r = (float)(1.0 / (double)d);
        uint32_t rb;
        memcpy(&rb, &r, 4); rb += (uint32_t)rcp_skew; memcpy(&r, &rb, 4);
        r = fmaf(fmaf(-d, r, 1.0f), r, r);
        float q = a * r;
        q = fmaf(fmaf(-d, q, a), r, q);
        q = fmaf(fmaf(-d, q, a), r, q);
        if (memcmp(&q, &ref, 4) != 0) { if (md == 0x7fffffu) bad_ones++; else bad++; }
    }
    if (all_ones_bad) *all_ones_bad = bad_ones;
    return bad;
}
