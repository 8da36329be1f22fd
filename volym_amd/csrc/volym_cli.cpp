// volym [run simple | benchmark] [-d] -- headless counterpart of the reference binary
// (src/cli.rs:4-56, src/main.rs:42-49).  See volym_amd/__main__.py for the Python twin.
//
//   benchmark : the reference's sweep (src/main.rs:178-345): 4 step sizes x {Base, Importance x {10,15,20},
//               ImportanceCone x {10,15,20}} = 28 rows, 3 trials, 1024x768; benchmark_results.csv with the
//               reference's columns (src/main.rs:71-85) + Mrays/s, algorithmic bytes, GB/s, roofline fraction.
//               A trial is --secs of back-to-back compute passes timed with HIP events (the reference counts
//               presented frames over 2 s of wall clock, blit/GUI/vsync included).
//   run simple: one frame of the interactive default view (src/state.rs:41-55) to frame.ppm.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <iterator>
#include <sstream>
#include <string>
#include <vector>

#include "demo.hpp"

using namespace volym;

namespace {

std::vector<uint8_t> read_file(const std::string& path)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) throw Error(VOLYM_E_INVALID, "cannot read " + path);
    return std::vector<uint8_t>(std::istreambuf_iterator<char>(f), {});
}

struct Options {
    std::string command = "run";
    std::string volume, labels, segments, output = "benchmark_results.csv";
    uint32_t width = 0, height = 0;
    double secs = 0.25;
    int device = 0;
    int frames_in_flight = 1;      // 2: VOLYM_OPT_FRAMES_IN_FLIGHT (benchmark: frames per wall clock)
    bool debug = false;
};

SimpleAssets load_assets(const Options& o, std::string& what)
{
    SimpleAssets a;
    if (!o.volume.empty()) {
        a.volume_raw = read_file(o.volume);
        if (!o.labels.empty()) a.labels_raw = read_file(o.labels);
        if (!o.segments.empty()) {
            const std::vector<uint8_t> j = read_file(o.segments);
            if (!parse_segments_json(std::string(j.begin(), j.end()), a.segments)) throw Error(VOLYM_E_INVALID, "bad segments JSON");
        }
        what = "file:" + o.volume;
        return a;
    }
    // the reference's default dataset is not distributed (.MISSING_LARGE_BLOBS): synthetic stand-in
    const uint32_t nx = 256, ny = 256, nz = 178;
    a.volume_raw.resize(static_cast<size_t>(nx) * ny * nz);
    a.labels_raw.resize(a.volume_raw.size());
    volym_synth_teapot(nx, ny, nz, 20250310u, a.volume_raw.data(), a.labels_raw.data());
    a.segments = {{"Segment_4", "Cup", 1, 3, 0}, {"Segment_5", "Ground", 2, 4, 0}, {"Segment_2", "Lobster", 0, 2, 255}};
    what = "synthetic teapot 256x256x178";
    return a;
}

void mean_std(const std::vector<double>& v, double& m, double& s)
{
    m = 0; for (double x : v) m += x; m /= v.size();
    s = 0; for (double x : v) s += (x - m) * (x - m); s = std::sqrt(s / v.size());   // population, src/main.rs:124-158
}

int benchmark_all(const Options& o)
{
    const uint32_t W = o.width ? o.width : 1024, H = o.height ? o.height : 768;       // src/main.rs:356-359
    std::string what;
    const SimpleAssets assets = load_assets(o, what);
    const float step_sizes[] = {0.0030f, 0.0050f, 0.0100f, 0.0200f};                  // src/main.rs:192
    const uint32_t importance_steps[] = {10, 15, 20};                                  // src/main.rs:193
    const int NUM_TRIALS = 3;                                                          // src/main.rs:179
    struct Row { const char* algo; float step; uint32_t isteps; bool cone; };
    std::vector<Row> rows;
    for (float s : step_sizes) rows.push_back({"Base", s, 0, false});
    for (float s : step_sizes) for (uint32_t n : importance_steps) rows.push_back({"Importance", s, n, false});
    for (float s : step_sizes) for (uint32_t n : importance_steps) rows.push_back({"ImportanceCone", s, n, true});

    std::printf("volym benchmark: %s, %ux%u, %zu rows x %d trials of %.2f s\n", what.c_str(), W, H, rows.size(), NUM_TRIALS, o.secs);
    GpuContext ctx(W, H, o.device);
    if (o.frames_in_flight == 2) ctx.check(volym_set_option(ctx.handle(), VOLYM_OPT_FRAMES_IN_FLIGHT, 2));   // before the scene
    // total milliseconds of n frames: per-launch HIP events on one stream, or -- two frames in flight -- the wall clock of n compute
    // passes enqueued back to back (the reference counts presented frames over wall time, src/main.rs:113-135)
    auto trial = [&](uint32_t n, std::vector<float>& ms) -> double {
        if (o.frames_in_flight != 2) {
            ms.assign(n, 0.0f);
            ctx.check(volym_time_passes(ctx.handle(), n, ms.data()));
            double total = 0; for (float x : ms) total += x;
            return total;
        }
        ctx.check(volym_sync(ctx.handle()));
        const auto t0 = std::chrono::steady_clock::now();
        for (uint32_t i = 0; i < n; ++i) ctx.check(volym_compute_pass(ctx.handle()));
        ctx.check(volym_sync(ctx.handle()));
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    };
    const StateParameters base = StateParameters::benchmark();                        // src/main.rs:180-190
    State state = State::with_parameters(static_cast<float>(W) / static_cast<float>(H), base);
    Simple demo = Simple::init(ctx, state, assets);
    std::ofstream csv(o.output);
    csv << "algorithm,step_size,importance_steps,use_cone,avg_total_frames,avg_total_time_ms,avg_frame_time_ms,avg_fps,"
           "std_dev_total_frames,std_dev_total_time_ms,std_dev_frame_time_ms,std_dev_fps,"
           "mrays_per_s,b_alg_bytes_per_frame,algorithmic_gb_per_s,hbm_roofline_fraction,n_gpus\n";
    for (const Row& r : rows) {
        StateParameters p = base;
        p.raymarching_step_size = r.step;
        if (std::strcmp(r.algo, "Base") != 0) {
            p.use_importance_rendering = 1; p.importance_check_ahead_steps = r.isteps; p.use_cone_importance_check = r.cone ? 1 : 0;
        }
        state = State::with_parameters(static_cast<float>(W) / static_cast<float>(H), p);
        state.update();                                                               // src/event_loop.rs:100
        demo.update_gpu_state(ctx, state);
        std::vector<float> ms(8);
        ctx.check(volym_time_passes(ctx.handle(), 8, ms.data()));
        std::sort(ms.begin(), ms.end());
        const double per = std::max<double>(ms[4], 1e-3);
        const uint32_t n = static_cast<uint32_t>(std::min(std::max(o.secs * 1e3 / per, 4.0), 20000.0));
        std::vector<double> frames, times, ftimes, fps;
        if (o.frames_in_flight == 2) { trial(8, ms); ctx.check(volym_settle(ctx.handle())); }   // both frame contexts warm, their lists in place
        for (int t = 0; t < NUM_TRIALS; ++t) {
            const double total = trial(n, ms);
            frames.push_back(n); times.push_back(total); ftimes.push_back(total / n); fps.push_back(n / (total * 1e-3));
        }
        volym_stats st;
        ctx.check(volym_stats_pass(ctx.handle(), &st));
        const double b_alg = static_cast<double>(st.n_vol) + st.n_imp + 4.0 * W * H;
        double m[4], s[4];
        mean_std(frames, m[0], s[0]); mean_std(times, m[1], s[1]); mean_std(ftimes, m[2], s[2]); mean_std(fps, m[3], s[3]);
        const double mrays = W * static_cast<double>(H) / (m[2] * 1e-3) / 1e6, gbs = b_alg / (m[2] * 1e-3) / 1e9;
        csv << r.algo << ',' << r.step << ',' << r.isteps << ',' << (r.cone ? "true" : "false");
        for (double v : m) csv << ',' << v;
        for (double v : s) csv << ',' << v;
        csv << ',' << mrays << ',' << static_cast<unsigned long long>(b_alg) << ',' << gbs << ',' << gbs / 8000.0 << ",1\n";
        std::printf("%-14s step %.4f steps %2u: %8.3f ms/frame %9.1f fps %9.0f Mrays/s  B_alg %6.1f MB  %5.1f%% of HBM roofline\n", r.algo,
                    r.step, r.isteps, m[2], m[3], mrays, b_alg / 1e6, 100.0 * gbs / 8000.0);
        std::fflush(stdout);
    }
    std::printf("wrote %s\n", o.output.c_str());
    return 0;
}

int run_simple(const Options& o)
{
    const uint32_t W = o.width ? o.width : 1280, H = o.height ? o.height : 720;
    std::string what;
    const SimpleAssets assets = load_assets(o, what);
    GpuContext ctx(W, H, o.device);
    State state = State::with_parameters(static_cast<float>(W) / static_cast<float>(H), StateParameters());   // src/state.rs:41-55
    state.update();
    Simple demo = Simple::init(ctx, state, assets);
    demo.update_gpu_state(ctx, state);
    demo.compute_pass(ctx);
    ctx.check(volym_sync(ctx.handle()));
    std::vector<uint8_t> rgba(static_cast<size_t>(W) * H * 4);
    ctx.check(volym_read_rgba8(ctx.handle(), rgba.data()));
    const std::string path = o.output == "benchmark_results.csv" ? "frame.ppm" : o.output;
    std::ofstream f(path, std::ios::binary);
    f << "P6\n" << W << ' ' << H << "\n255\n";
    for (size_t i = 0; i < static_cast<size_t>(W) * H; ++i) f.write(reinterpret_cast<const char*>(&rgba[4 * i]), 3);
    std::printf("run simple: %s, %ux%u -> %s\n", what.c_str(), W, H, path.c_str());
    return 0;
}

}  // namespace

int main(int argc, char** argv)
{
    Options o;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto next = [&]() -> std::string { if (i + 1 >= argc) throw Error(VOLYM_E_INVALID, "missing value for " + a); return argv[++i]; };
        try {
            if (a == "run") { o.command = "run"; if (i + 1 < argc && std::string(argv[i + 1]) == "simple") ++i; }
            else if (a == "benchmark") o.command = "benchmark";
            else if (a == "-d" || a == "--debug") o.debug = true;
            else if (a == "--volume") o.volume = next();
            else if (a == "--labels") o.labels = next();
            else if (a == "--segments") o.segments = next();
            else if (a == "--output") o.output = next();
            else if (a == "--width") o.width = static_cast<uint32_t>(std::stoul(next()));
            else if (a == "--height") o.height = static_cast<uint32_t>(std::stoul(next()));
            else if (a == "--secs") o.secs = std::stod(next());
            else if (a == "--device") o.device = std::stoi(next());
            else if (a == "--frames-in-flight") { o.frames_in_flight = std::stoi(next()); if (o.frames_in_flight != 1 && o.frames_in_flight != 2) throw Error(VOLYM_E_INVALID, "--frames-in-flight: 1 or 2"); }
            else { std::fprintf(stderr, "usage: volym [run simple | benchmark] [-d] [--volume f --labels f --segments f] [--width n --height n] [--secs s] [--output f] [--frames-in-flight 1|2]\n"); return 2; }
        } catch (const std::exception& e) { std::fprintf(stderr, "%s\n", e.what()); return 2; }
    }
    try {
        return o.command == "benchmark" ? benchmark_all(o) : run_simple(o);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 1;
    }
}
