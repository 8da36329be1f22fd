// Host scene model (see scene.hpp).  Compiled with -ffp-contract=off: the f32 operation
// order below is part of the contract (the uniforms it emits decide voxel indices).
#include "scene.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace volym {

// ---------------------------------------------------------------------------------------
// cgmath slice
// ---------------------------------------------------------------------------------------
float Vector3::magnitude() const { return std::sqrt(dot(*this)); }
Vector3 Vector3::normalize() const { return *this * (1.0f / magnitude()); }

Matrix4 Matrix4::look_at_rh(Vector3 eye, Vector3 center, Vector3 up)
{
    // cgmath Matrix4::look_at_rh -> look_to_rh(eye, center - eye, up)
    Vector3 f = (center - eye).normalize();
    Vector3 s = f.cross(up).normalize();
    Vector3 u = s.cross(f);
    Matrix4 r;
    r.m[0][0] = s.x; r.m[0][1] = u.x; r.m[0][2] = -f.x; r.m[0][3] = 0.0f;
    r.m[1][0] = s.y; r.m[1][1] = u.y; r.m[1][2] = -f.y; r.m[1][3] = 0.0f;
    r.m[2][0] = s.z; r.m[2][1] = u.z; r.m[2][2] = -f.z; r.m[2][3] = 0.0f;
    r.m[3][0] = -eye.dot(s); r.m[3][1] = -eye.dot(u); r.m[3][2] = eye.dot(f); r.m[3][3] = 1.0f;
    return r;
}

Matrix4 Matrix4::perspective_deg(float fovy_deg, float aspect, float near, float far)
{
    // cgmath perspective(Deg(fovy), aspect, near, far) -> PerspectiveFov -> Matrix4
    const float rad = fovy_deg * static_cast<float>(3.14159265358979323846 / 180.0);
    const float f = 1.0f / std::tan(rad / 2.0f);   // Rad::cot
    Matrix4 r;
    std::memset(r.m, 0, sizeof r.m);
    r.m[0][0] = f / aspect;
    r.m[1][1] = f;
    r.m[2][2] = (far + near) / (near - far);
    r.m[2][3] = -1.0f;
    r.m[3][2] = (2.0f * far * near) / (near - far);
    return r;
}

bool Matrix4::invert(Matrix4& out) const
{
    // adjugate / determinant; term order fixed (DESIGN.md "Host math")
    const float* a = &m[0][0];
    float c[16];
    c[0] = a[5] * a[10] * a[15] - a[5] * a[11] * a[14] - a[9] * a[6] * a[15] +
           a[9] * a[7] * a[14] + a[13] * a[6] * a[11] - a[13] * a[7] * a[10];
    c[4] = -a[4] * a[10] * a[15] + a[4] * a[11] * a[14] + a[8] * a[6] * a[15] -
           a[8] * a[7] * a[14] - a[12] * a[6] * a[11] + a[12] * a[7] * a[10];
    c[8] = a[4] * a[9] * a[15] - a[4] * a[11] * a[13] - a[8] * a[5] * a[15] +
           a[8] * a[7] * a[13] + a[12] * a[5] * a[11] - a[12] * a[7] * a[9];
    c[12] = -a[4] * a[9] * a[14] + a[4] * a[10] * a[13] + a[8] * a[5] * a[14] -
            a[8] * a[6] * a[13] - a[12] * a[5] * a[10] + a[12] * a[6] * a[9];
    c[1] = -a[1] * a[10] * a[15] + a[1] * a[11] * a[14] + a[9] * a[2] * a[15] -
           a[9] * a[3] * a[14] - a[13] * a[2] * a[11] + a[13] * a[3] * a[10];
    c[5] = a[0] * a[10] * a[15] - a[0] * a[11] * a[14] - a[8] * a[2] * a[15] +
           a[8] * a[3] * a[14] + a[12] * a[2] * a[11] - a[12] * a[3] * a[10];
    c[9] = -a[0] * a[9] * a[15] + a[0] * a[11] * a[13] + a[8] * a[1] * a[15] -
           a[8] * a[3] * a[13] - a[12] * a[1] * a[11] + a[12] * a[3] * a[9];
    c[13] = a[0] * a[9] * a[14] - a[0] * a[10] * a[13] - a[8] * a[1] * a[14] +
            a[8] * a[2] * a[13] + a[12] * a[1] * a[10] - a[12] * a[2] * a[9];
    c[2] = a[1] * a[6] * a[15] - a[1] * a[7] * a[14] - a[5] * a[2] * a[15] +
           a[5] * a[3] * a[14] + a[13] * a[2] * a[7] - a[13] * a[3] * a[6];
    c[6] = -a[0] * a[6] * a[15] + a[0] * a[7] * a[14] + a[4] * a[2] * a[15] -
           a[4] * a[3] * a[14] - a[12] * a[2] * a[7] + a[12] * a[3] * a[6];
    c[10] = a[0] * a[5] * a[15] - a[0] * a[7] * a[13] - a[4] * a[1] * a[15] +
            a[4] * a[3] * a[13] + a[12] * a[1] * a[7] - a[12] * a[3] * a[5];
    c[14] = -a[0] * a[5] * a[14] + a[0] * a[6] * a[13] + a[4] * a[1] * a[14] -
            a[4] * a[2] * a[13] - a[12] * a[1] * a[6] + a[12] * a[2] * a[5];
    c[3] = -a[1] * a[6] * a[11] + a[1] * a[7] * a[10] + a[5] * a[2] * a[11] -
           a[5] * a[3] * a[10] - a[9] * a[2] * a[7] + a[9] * a[3] * a[6];
    c[7] = a[0] * a[6] * a[11] - a[0] * a[7] * a[10] - a[4] * a[2] * a[11] +
           a[4] * a[3] * a[10] + a[8] * a[2] * a[7] - a[8] * a[3] * a[6];
    c[11] = -a[0] * a[5] * a[11] + a[0] * a[7] * a[9] + a[4] * a[1] * a[11] -
            a[4] * a[3] * a[9] - a[8] * a[1] * a[7] + a[8] * a[3] * a[5];
    c[15] = a[0] * a[5] * a[10] - a[0] * a[6] * a[9] - a[4] * a[1] * a[10] +
            a[4] * a[2] * a[9] + a[8] * a[1] * a[6] - a[8] * a[2] * a[5];
    const float det = a[0] * c[0] + a[1] * c[4] + a[2] * c[8] + a[3] * c[12];
    if (det == 0.0f) return false;
    const float inv_det = 1.0f / det;
    float* o = &out.m[0][0];
    for (int i = 0; i < 16; ++i) o[i] = c[i] * inv_det;
    return true;
}

Matrix4 Matrix4::operator*(const Matrix4& rhs) const
{
    Matrix4 r;
    for (int col = 0; col < 4; ++col)
        for (int row = 0; row < 4; ++row)
            r.m[col][row] = m[0][row] * rhs.m[col][0] + m[1][row] * rhs.m[col][1] +
                            m[2][row] * rhs.m[col][2] + m[3][row] * rhs.m[col][3];
    return r;
}

// ---------------------------------------------------------------------------------------
// Camera (src/camera.rs)
// ---------------------------------------------------------------------------------------
Camera Camera::default_with_aspect_and_pos(float aspect, const float position[3])
{
    Camera c;
    c.position[0] = position[0]; c.position[1] = position[1]; c.position[2] = position[2];
    c.target[0] = c.target[1] = c.target[2] = 0.5f;     // src/camera.rs:23
    c.up[0] = 0.0f; c.up[1] = 1.0f; c.up[2] = 0.0f;
    c.aspect = aspect;
    c.fovy = 90.0f; c.znear = 0.01f; c.zfar = 1000.0f;  // src/camera.rs:25-27
    c.horizontal_angle = 0.0f; c.vertical_angle = 0.0f;
    c.distance = 1.0f;                                  // src/camera.rs:39
    c.max_distance = 10.0f; c.min_distance = 1.0f;      // src/camera.rs:28-29
    return c;
}

static float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

void Camera::orbit(float hd, float vd, float zd)
{
    horizontal_angle += hd;
    vertical_angle = clampf(vertical_angle + vd, -89.0f, 89.0f);
    distance = clampf(distance + zd, min_distance, max_distance);
    const float rads_per_deg = 3.14159265358979323846f / 180.0f;   // f32::to_radians
    const float h = horizontal_angle * rads_per_deg;
    const float v = vertical_angle * rads_per_deg;
    position[0] = target[0] + distance * std::sin(h) * std::cos(v);
    position[1] = target[1] + distance * std::sin(v);
    position[2] = target[2] + distance * std::cos(h) * std::cos(v);
}

Matrix4 Camera::view_matrix() const
{
    return Matrix4::look_at_rh({position[0], position[1], position[2]},
                               {target[0], target[1], target[2]}, {up[0], up[1], up[2]});
}

Matrix4 Camera::projection_matrix() const
{
    return Matrix4::perspective_deg(fovy, aspect, znear, zfar);
}

CameraController::CameraController(float s, float zs)
{
    rotate_horizontal = 0.0f; rotate_vertical = 0.0f; scroll = 0.0f;
    sensitivity = s; zoom_sensitivity = zs;
}

void CameraController::process_mouse(double dx, double dy)
{
    rotate_horizontal = -static_cast<float>(dx) * sensitivity;   // src/camera.rs:96-99
    rotate_vertical = -static_cast<float>(dy) * sensitivity;
}

void CameraController::process_scroll(float line_delta)
{
    scroll = -line_delta * zoom_sensitivity;                      // src/camera.rs:101-108
}

void CameraController::update_camera(volym_camera& camera)
{
    static_cast<Camera&>(camera).orbit(rotate_horizontal, rotate_vertical, scroll);
    rotate_horizontal = 0.0f; rotate_vertical = 0.0f; scroll = 0.0f;
}

bool camera_uniforms_from(const volym_camera& camera, volym_camera_uniforms& out)
{
    const Camera& c = static_cast<const Camera&>(camera);
    const Matrix4 proj = c.projection_matrix();
    const Matrix4 view = c.view_matrix();
    Matrix4 vinv, pinv;
    if (!view.invert(vinv) || !proj.invert(pinv)) return false;
    const Matrix4 ivp = vinv * pinv;                              // src/gpu_resources/camera.rs:72-76
    std::memcpy(out.view_matrix, view.m, sizeof view.m);
    std::memcpy(out.projection_matrix, proj.m, sizeof proj.m);
    std::memcpy(out.inverse_view_proj, ivp.m, sizeof ivp.m);
    out.camera_position[0] = c.position[0];
    out.camera_position[1] = c.position[1];
    out.camera_position[2] = c.position[2];
    out._padding = 0.0f;
    return true;
}

// ---------------------------------------------------------------------------------------
// State (src/state.rs)
// ---------------------------------------------------------------------------------------
StateParameters::StateParameters()
{
    camera_position[0] = camera_position[1] = camera_position[2] = 0.5f;
    use_cone_importance_check = 0; use_importance_coloring = 0; use_opacity = 1;
    use_importance_rendering = 0;
    density_trheshold = 0.12f;
    use_gaussian_smoothing = 1;
    importance_check_ahead_steps = 12;
    raymarching_step_size = 0.010f;
}

StateParameters StateParameters::benchmark()
{
    StateParameters p;
    p.camera_position[0] = 0.5f; p.camera_position[1] = 0.5f; p.camera_position[2] = 3.5f;
    p.use_opacity = 1;
    p.density_trheshold = 0.15f;
    p.use_cone_importance_check = 0; p.use_importance_coloring = 0;
    p.use_importance_rendering = 0; p.use_gaussian_smoothing = 0;
    p.importance_check_ahead_steps = 15;
    p.raymarching_step_size = 0.020f;
    return p;
}

State State::with_parameters(float aspect, const volym_state_parameters& p)
{
    State s;
    static_cast<volym_camera&>(s.camera) = Camera::default_with_aspect_and_pos(aspect, p.camera_position);
    static_cast<volym_camera_controller&>(s.camera_controller) = CameraController(0.2f, 0.2f);
    s.density_threshold = p.density_trheshold;
    s.use_cone_importance_check = p.use_cone_importance_check;
    s.use_importance_coloring = p.use_importance_coloring;
    s.use_opacity = p.use_opacity;
    s.use_importance_rendering = p.use_importance_rendering;
    s.use_gaussian_smoothing = p.use_gaussian_smoothing;
    s.importance_check_ahead_steps = p.importance_check_ahead_steps;
    s.raymarching_step_size = p.raymarching_step_size;
    return s;
}

void State::update()
{
    static_cast<CameraController&>(camera_controller).update_camera(camera);
}

void parameter_uniforms_from(const volym_state& s, volym_parameter_uniforms& out)
{
    out.use_cone_importance_check = s.use_cone_importance_check ? 1u : 0u;
    out.use_importance_coloring = s.use_importance_coloring ? 1u : 0u;
    out.use_opacity = s.use_opacity ? 1u : 0u;
    out.use_importance_rendering = s.use_importance_rendering ? 1u : 0u;
    out.density_threshold = s.density_threshold;
    out.use_gaussian_smoothing = s.use_gaussian_smoothing ? 1u : 0u;
    out.importance_check_ahead_steps = s.importance_check_ahead_steps;
    out.raymarching_step_size = s.raymarching_step_size;
}

// ---------------------------------------------------------------------------------------
// TransferFunction (src/transfer_function.rs)
// ---------------------------------------------------------------------------------------
static uint32_t as_u32(float f)
{
    if (!(f > 0.0f)) return 0u;
    if (f >= 4294967296.0f) return 0xffffffffu;
    return static_cast<uint32_t>(f);
}
static uint8_t as_u8(float f)
{
    if (!(f > 0.0f)) return 0;
    if (f >= 255.0f) return 255;
    return static_cast<uint8_t>(f);
}

TransferFunction::TransferFunction(uint32_t md)
    : max_density(md), function_vec_(static_cast<size_t>(md + 1) * 4, 0.0f) {}

TransferFunction TransferFunction::default_()
{
    TransferFunction tf(255);
    tf.add_rgb_control_point({{0.0f, 1.0f, 0.0f, 1.0f}, 0.0f});
    tf.add_rgb_control_point({{0.0f, 1.0f, 1.0f, 1.0f}, 0.2f});
    tf.add_rgb_control_point({{1.0f, 1.0f, 0.0f, 1.0f}, 0.4f});
    tf.add_rgb_control_point({{1.0f, 0.0f, 1.0f, 1.0f}, 0.6f});
    tf.add_rgb_control_point({{1.0f, 0.0f, 0.0f, 1.0f}, 1.0f});
    tf.add_alpha_control_point({{0.0f, 0.0f, 0.0f, 0.0f}, 0.0f});
    tf.add_alpha_control_point({{0.0f, 0.0f, 0.0f, 1.0f}, 1.0f});
    tf.build_linear();
    return tf;
}

static void insert_sorted(std::vector<TransferControlPoint>& v, const TransferControlPoint& p)
{
    v.push_back(p);
    std::stable_sort(v.begin(), v.end(), [](const TransferControlPoint& a, const TransferControlPoint& b) {
        return a.iso_value < b.iso_value;
    });
}

void TransferFunction::add_rgb_control_point(const TransferControlPoint& p) { insert_sorted(rgb_points_, p); }
void TransferFunction::add_alpha_control_point(const TransferControlPoint& p) { insert_sorted(alpha_points_, p); }

void TransferFunction::build_linear()
{
    const float md = static_cast<float>(max_density);
    for (size_t w = 0; w + 1 < rgb_points_.size(); ++w) {
        const TransferControlPoint& s = rgb_points_[w];
        const TransferControlPoint& e = rgb_points_[w + 1];
        const uint32_t si = as_u32(s.iso_value * md), ei = as_u32(e.iso_value * md);
        for (uint32_t x = si; x <= ei && x <= max_density; ++x) {
            const float k = ei == si ? 0.0f : static_cast<float>(x - si) / static_cast<float>(ei - si);
            float* f = &function_vec_[4 * static_cast<size_t>(x)];
            f[0] = s.color[0] + (e.color[0] - s.color[0]) * k;
            f[1] = s.color[1] + (e.color[1] - s.color[1]) * k;
            f[2] = s.color[2] + (e.color[2] - s.color[2]) * k;
        }
    }
    for (size_t w = 0; w + 1 < alpha_points_.size(); ++w) {
        const TransferControlPoint& s = alpha_points_[w];
        const TransferControlPoint& e = alpha_points_[w + 1];
        const uint32_t si = as_u32(s.iso_value * md), ei = as_u32(e.iso_value * md);
        for (uint32_t x = si; x <= ei && x <= max_density; ++x) {
            const float k = ei == si ? 0.0f : static_cast<float>(x - si) / static_cast<float>(ei - si);
            function_vec_[4 * static_cast<size_t>(x) + 3] = s.color[3] + (e.color[3] - s.color[3]) * k;
        }
    }
}

void TransferFunction::get(float value, float out[4]) const
{
    const float md = static_cast<float>(max_density);
    const float idx = clampf(value * md, 0.0f, md);
    const float fl = std::floor(idx);
    const size_t i0 = static_cast<size_t>(fl);
    const size_t i1 = std::min<size_t>(i0 + 1, max_density);
    const float t = idx - fl;
    for (int c = 0; c < 4; ++c) {
        const float v1 = function_vec_[4 * i0 + c], v2 = function_vec_[4 * i1 + c];
        out[c] = v1 + (v2 - v1) * t;
    }
}

std::vector<uint8_t> TransferFunction::bake_rgba8() const
{
    const uint32_t tf_size = max_density + 1;
    std::vector<uint8_t> data;
    data.reserve(static_cast<size_t>(tf_size) * 4);
    for (uint32_t i = 0; i < tf_size; ++i) {
        float v[4];
        get(static_cast<float>(i) / static_cast<float>(tf_size), v);
        for (int c = 0; c < 4; ++c) data.push_back(as_u8(v[c] * 255.0f));
    }
    return data;
}

// ---------------------------------------------------------------------------------------
// assets
// ---------------------------------------------------------------------------------------
void flip_3d_texture_y(uint8_t* data, size_t x, size_t y, size_t z)
{
    for (size_t k = 0; k < z; ++k)
        for (size_t j = 0; j < y / 2; ++j)
            std::swap_ranges(data + k * x * y + j * x, data + k * x * y + j * x + x,
                             data + k * x * y + (y - j - 1) * x);
}

void prepare_volume(const uint8_t* raw, size_t len, size_t nx, size_t ny, size_t nz, bool flip_y,
                    uint8_t* out)
{
    const size_t desired = nx * ny * nz;
    const size_t n = std::min(len, desired);
    std::memcpy(out, raw, n);
    std::memset(out + n, 0, desired - n);
    if (flip_y) flip_3d_texture_y(out, nx, ny, nz);
}

void map_segments_to_importance(uint8_t* data, size_t len, const uint8_t* label_values,
                                const uint8_t* importances, size_t n)
{
    uint8_t table[256];
    std::memset(table, 0, sizeof table);
    for (size_t s = n; s-- > 0;) table[label_values[s]] = importances[s];   // first match wins
    for (size_t i = 0; i < len; ++i) data[i] = table[data[i]];
}

// --- tiny JSON reader for [{"id": "...", "importance": N, ...}, ...] -----------------------
namespace {
struct Cursor {
    const std::string& s;
    size_t i;
    void ws() { while (i < s.size() && (s[i] == ' ' || s[i] == '\n' || s[i] == '\r' || s[i] == '\t')) ++i; }
    bool eat(char c) { ws(); if (i < s.size() && s[i] == c) { ++i; return true; } return false; }
    bool str(std::string& out)
    {
        ws();
        if (i >= s.size() || s[i] != '"') return false;
        ++i; out.clear();
        while (i < s.size() && s[i] != '"') { if (s[i] == '\\' && i + 1 < s.size()) ++i; out.push_back(s[i++]); }
        if (i >= s.size()) return false;
        ++i; return true;
    }
    bool num(long& out)
    {
        ws();
        size_t b = i; bool neg = false;
        if (i < s.size() && s[i] == '-') { neg = true; ++i; }
        long v = 0; size_t d = i;
        while (i < s.size() && s[i] >= '0' && s[i] <= '9') v = v * 10 + (s[i++] - '0');
        if (i == d) { i = b; return false; }
        out = neg ? -v : v; return true;
    }
};
}  // namespace

bool parse_segments_json(const std::string& text, std::vector<SegmentInfo>& out)
{
    Cursor c{text, 0};
    out.clear();
    if (!c.eat('[')) return false;
    if (c.eat(']')) return true;
    do {
        if (!c.eat('{')) return false;
        SegmentInfo seg{};
        bool have_label = false, have_imp = false;
        if (!c.eat('}')) {
            do {
                std::string key;
                if (!c.str(key) || !c.eat(':')) return false;
                std::string sv; long nv = 0;
                if (c.str(sv)) {
                    if (key == "id") seg.id = sv; else if (key == "name") seg.name = sv;
                } else if (c.num(nv)) {
                    if (nv < 0 || nv > 255) return false;   // u8 fields
                    if (key == "index") seg.index = static_cast<uint8_t>(nv);
                    else if (key == "label_value") { seg.label_value = static_cast<uint8_t>(nv); have_label = true; }
                    else if (key == "importance") { seg.importance = static_cast<uint8_t>(nv); have_imp = true; }
                } else return false;
            } while (c.eat(','));
            if (!c.eat('}')) return false;
        }
        if (!have_label || !have_imp) return false;
        out.push_back(seg);
    } while (c.eat(','));
    return c.eat(']');
}

}  // namespace volym

// ---------------------------------------------------------------------------------------
// C ABI (include/volym_host.h)
// ---------------------------------------------------------------------------------------
using namespace volym;

extern "C" {

void volym_camera_default_with_aspect_and_pos(volym_camera* c, float aspect, const float position[3])
{
    *c = Camera::default_with_aspect_and_pos(aspect, position);
}
void volym_camera_orbit(volym_camera* c, float h, float v, float z) { static_cast<Camera*>(c)->orbit(h, v, z); }
void volym_camera_view_matrix(const volym_camera* c, float out[4][4])
{
    Matrix4 m = static_cast<const Camera*>(c)->view_matrix();
    std::memcpy(out, m.m, sizeof m.m);
}
void volym_camera_projection_matrix(const volym_camera* c, float out[4][4])
{
    Matrix4 m = static_cast<const Camera*>(c)->projection_matrix();
    std::memcpy(out, m.m, sizeof m.m);
}
int volym_camera_uniforms_from(const volym_camera* c, volym_camera_uniforms* out)
{
    if (!c || !out) return VOLYM_E_INVALID;
    return camera_uniforms_from(*c, *out) ? VOLYM_OK : VOLYM_E_INVALID;
}

void volym_camera_controller_new(volym_camera_controller* cc, float s, float zs) { *cc = CameraController(s, zs); }
void volym_camera_controller_process_mouse(volym_camera_controller* cc, double dx, double dy)
{
    static_cast<CameraController*>(cc)->process_mouse(dx, dy);
}
void volym_camera_controller_process_scroll(volym_camera_controller* cc, float d)
{
    static_cast<CameraController*>(cc)->process_scroll(d);
}
void volym_camera_controller_update_camera(volym_camera_controller* cc, volym_camera* cam)
{
    static_cast<CameraController*>(cc)->update_camera(*cam);
}

void volym_state_parameters_default(volym_state_parameters* p) { *p = StateParameters(); }
void volym_state_parameters_benchmark(volym_state_parameters* p) { *p = StateParameters::benchmark(); }
void volym_state_with_parameters(volym_state* s, float aspect, const volym_state_parameters* p)
{
    *s = State::with_parameters(aspect, *p);
}
void volym_state_update(volym_state* s) { static_cast<State*>(s)->update(); }
int volym_parameter_uniforms_from(const volym_state* s, volym_parameter_uniforms* out)
{
    if (!s || !out) return VOLYM_E_INVALID;
    parameter_uniforms_from(*s, *out);
    return VOLYM_OK;
}

void volym_transfer_function_default_lut(uint8_t lut[1024])
{
    std::vector<uint8_t> d = TransferFunction::default_().bake_rgba8();
    std::memcpy(lut, d.data(), 1024);
}

int volym_transfer_function_bake(const float* rgb, uint32_t n_rgb, const float* alpha, uint32_t n_alpha,
                                 uint8_t lut[1024])
{
    if ((!rgb && n_rgb) || (!alpha && n_alpha) || !lut) return VOLYM_E_INVALID;
    TransferFunction tf(255);
    for (uint32_t i = 0; i < n_rgb; ++i) {
        if (!(rgb[4 * i] >= 0.0f && rgb[4 * i] <= 1.0f)) return VOLYM_E_INVALID;
        tf.add_rgb_control_point({{rgb[4 * i + 1], rgb[4 * i + 2], rgb[4 * i + 3], 1.0f}, rgb[4 * i]});
    }
    for (uint32_t i = 0; i < n_alpha; ++i) {
        if (!(alpha[2 * i] >= 0.0f && alpha[2 * i] <= 1.0f)) return VOLYM_E_INVALID;
        tf.add_alpha_control_point({{0.0f, 0.0f, 0.0f, alpha[2 * i + 1]}, alpha[2 * i]});
    }
    tf.build_linear();
    std::vector<uint8_t> d = tf.bake_rgba8();
    std::memcpy(lut, d.data(), 1024);
    return VOLYM_OK;
}

int volym_prepare_volume(const uint8_t* raw, size_t len, uint32_t nx, uint32_t ny, uint32_t nz, int flip_y,
                         uint8_t* out)
{
    if ((!raw && len) || !out || !nx || !ny || !nz) return VOLYM_E_INVALID;
    prepare_volume(raw, len, nx, ny, nz, flip_y != 0, out);
    return VOLYM_OK;
}

int volym_map_segments_to_importance(uint8_t* data, size_t len, const uint8_t* lv, const uint8_t* im,
                                     uint32_t n)
{
    if ((!data && len) || ((!lv || !im) && n)) return VOLYM_E_INVALID;
    map_segments_to_importance(data, len, lv, im, n);
    return VOLYM_OK;
}

}  // extern "C"
