// C++ face of the compute-plugin boundary, over the C ABI of libvolym_hip.so.
//
//     trait ComputeDemo { init(ctx, state, output); update_gpu_state(ctx, state); compute_pass(ctx) }
//                                                         -- /root/reference/src/demos/mod.rs:9-17
//     struct Simple                                       -- /root/reference/src/demos/simple/mod.rs:26-121
//
// GpuContext stands where src/gpu_context.rs + GpuWriteTexture2D stood: a device and a W x H rgba8 output.
#pragma once

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/volym_host.h"
#include "scene.hpp"

namespace volym {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error("volym error " + std::to_string(c) + ": " + m), code(c) {}
};

class GpuContext {
public:
    GpuContext(uint32_t width, uint32_t height, int device_id = -1) : width(width), height(height)
    {
        const int rc = volym_create(&ctx_, width, height, device_id);
        if (rc != VOLYM_OK) throw Error(rc, volym_last_error(nullptr));
    }
    ~GpuContext() { volym_destroy(ctx_); }
    GpuContext(const GpuContext&) = delete;
    GpuContext& operator=(const GpuContext&) = delete;
    volym_ctx* handle() const { return ctx_; }
    void check(int rc) const { if (rc != VOLYM_OK) throw Error(rc, volym_last_error(ctx_)); }
    const uint32_t width, height;

private:
    volym_ctx* ctx_ = nullptr;
};

class ComputeDemo {
public:
    virtual ~ComputeDemo() = default;
    virtual void update_gpu_state(const GpuContext& ctx, const volym_state& state) = 0;
    virtual void compute_pass(const GpuContext& ctx) = 0;
};

// The assets Simple::init reads from hard-coded paths (src/demos/simple/mod.rs:40-55), passed explicitly.
struct SimpleAssets {
    std::vector<uint8_t> volume_raw, labels_raw;
    std::vector<SegmentInfo> segments;
    uint32_t nx = 256, ny = 256, nz = 256;       // src/gpu_resources/volume.rs:41
    int filter = VOLYM_FILTER_NEAREST;           // src/gpu_resources/volume.rs:92-95
};

class Simple : public ComputeDemo {
public:
    static Simple init(const GpuContext& ctx, const volym_state& state, const SimpleAssets& a)
    {
        const size_t n = static_cast<size_t>(a.nx) * a.ny * a.nz;
        std::vector<uint8_t> vol(n), imp(n), labels(a.labels_raw);
        prepare_volume(a.volume_raw.data(), a.volume_raw.size(), a.nx, a.ny, a.nz, true, vol.data());   // GpuVolume::init
        std::vector<uint8_t> lv, im;
        for (const SegmentInfo& s : a.segments) { lv.push_back(s.label_value); im.push_back(s.importance); }
        map_segments_to_importance(labels.data(), labels.size(), lv.data(), im.data(), lv.size());       // GpuImportances::init
        prepare_volume(labels.data(), labels.size(), a.nx, a.ny, a.nz, true, imp.data());
        ctx.check(volym_set_volume(ctx.handle(), vol.data(), a.nx, a.ny, a.nz, a.filter));
        ctx.check(volym_set_importances(ctx.handle(), imp.data(), a.nx, a.ny, a.nz));
        const std::vector<uint8_t> lut = TransferFunction::default_().bake_rgba8();                       // src/demos/simple/mod.rs:64-66
        ctx.check(volym_set_transfer_function(ctx.handle(), lut.data(), 256));
        Simple s;
        s.update_gpu_state(ctx, state);
        return s;
    }
    void update_gpu_state(const GpuContext& ctx, const volym_state& state) override   // src/demos/pipeline.rs:208-212
    {
        volym_camera_uniforms cam;
        volym_parameter_uniforms par;
        if (!camera_uniforms_from(state.camera, cam)) throw Error(VOLYM_E_INVALID, "inverse_view_proj inversion failed");
        parameter_uniforms_from(state, par);
        ctx.check(volym_update(ctx.handle(), &cam, &par));
    }
    void compute_pass(const GpuContext& ctx) override { ctx.check(volym_compute_pass(ctx.handle())); }   // src/demos/pipeline.rs:62-102
};

}  // namespace volym
