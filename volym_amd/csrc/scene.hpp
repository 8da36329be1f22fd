// Host scene model of volym, restated in C++ for the MI355X build.
// The reference keeps these in Rust (src/camera.rs, src/state.rs, src/transfer_function.rs,
// src/gpu_resources/{camera,parameters,transfer_function,volume}.rs); citations below are
// file:line under /root/reference/.  Nothing here touches the GPU.
#pragma once

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/volym_host.h"

namespace volym {

// ---- the slice of cgmath 0.18.0 the reference uses (Cargo.lock:500-501) -------------
struct Vector3 {
    float x, y, z;
    Vector3 operator+(Vector3 o) const { return {x + o.x, y + o.y, z + o.z}; }
    Vector3 operator-(Vector3 o) const { return {x - o.x, y - o.y, z - o.z}; }
    Vector3 operator*(float s) const { return {x * s, y * s, z * s}; }
    float dot(Vector3 o) const { return x * o.x + y * o.y + z * o.z; }
    Vector3 cross(Vector3 o) const {
        return {y * o.z - z * o.y, z * o.x - x * o.z, x * o.y - y * o.x};
    }
    float magnitude() const;
    Vector3 normalize() const;   // self * (1 / magnitude)
};

struct Matrix4 {
    float m[4][4];               // column-major: m[col][row]
    static Matrix4 look_at_rh(Vector3 eye, Vector3 center, Vector3 up);
    static Matrix4 perspective_deg(float fovy_deg, float aspect, float near, float far);
    bool invert(Matrix4& out) const;
    Matrix4 operator*(const Matrix4& rhs) const;
};

// ---- src/camera.rs ------------------------------------------------------------------
struct Camera : volym_camera {
    static Camera default_with_aspect_and_pos(float aspect, const float position[3]);
    void orbit(float horizontal_delta, float vertical_delta, float zoom_delta);
    Matrix4 view_matrix() const;
    Matrix4 projection_matrix() const;
};

struct CameraController : volym_camera_controller {
    CameraController(float sensitivity, float zoom_sensitivity);
    void process_mouse(double mouse_dx, double mouse_dy);
    void process_scroll(float line_delta);
    void update_camera(volym_camera& camera);
};

// ---- src/gpu_resources/camera.rs:66-85 ------------------------------------------------
bool camera_uniforms_from(const volym_camera& camera, volym_camera_uniforms& out);

// ---- src/state.rs ---------------------------------------------------------------------
struct StateParameters : volym_state_parameters {
    StateParameters();                       // Default (src/state.rs:41-55)
    static StateParameters benchmark();      // src/main.rs:180-190
};

struct State : volym_state {
    static State with_parameters(float aspect, const volym_state_parameters& p);
    void update();
};

void parameter_uniforms_from(const volym_state& s, volym_parameter_uniforms& out);

// ---- src/transfer_function.rs -----------------------------------------------------------
struct TransferControlPoint {
    float color[4];
    float iso_value;
};

class TransferFunction {
public:
    explicit TransferFunction(uint32_t max_density);
    static TransferFunction default_();      // impl Default (src/transfer_function.rs:19-56)
    void add_rgb_control_point(const TransferControlPoint& p);
    void add_alpha_control_point(const TransferControlPoint& p);
    void build_linear();
    void get(float value, float out[4]) const;
    // GPUTransferFunction::new_texture_1d_rgbt's bake (src/gpu_resources/transfer_function.rs:58-69)
    std::vector<uint8_t> bake_rgba8() const;
    uint32_t max_density;

private:
    std::vector<TransferControlPoint> rgb_points_, alpha_points_;
    std::vector<float> function_vec_;        // (max_density + 1) x 4
};

// ---- asset preparation ------------------------------------------------------------------
void flip_3d_texture_y(uint8_t* data, size_t x, size_t y, size_t z);   // src/gpu_resources/mod.rs:70-82
void prepare_volume(const uint8_t* raw, size_t len, size_t nx, size_t ny, size_t nz, bool flip_y,
                    uint8_t* out);                                     // src/gpu_resources/volume.rs:38-61
struct SegmentInfo {                                                   // src/demos/simple/importance.rs:13-20
    std::string id, name;
    uint8_t index, label_value, importance;
};
void map_segments_to_importance(uint8_t* data, size_t len, const uint8_t* label_values,
                                const uint8_t* importances, size_t n);
// minimal reader for the segments JSON the reference ships
// (assets/boston_teapot_256x256x178_uint8_segments.json); false on malformed input.
bool parse_segments_json(const std::string& text, std::vector<SegmentInfo>& out);

}  // namespace volym
