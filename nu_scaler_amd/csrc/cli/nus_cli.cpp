// nus_cli.cpp -- `nu_scaler_cli`: the image-file commands of the north star's CLI as a native program on the
// C ABI (include/nuscaler_hip.h).  Shape: `upscale_image_file` of the legacy crate
// (Nu_scale/src/upscale/mod.rs:307-338) and the option names of its `fullscreen` subcommand
// (Nu_scale/src/main.rs:36-73: --tech, --quality, --algorithm).
//
//   nu_scaler_cli upscale <in.png> <out.png> [--algorithm A] [--scale S] [--tech T] [--quality Q] [--device N]
//   nu_scaler_cli interpolate <a.png> <b.png> <out.png> [--t X] [--flow] [--device N]
//   nu_scaler_cli png-copy <in.png> <out.png>        (decode + encode only; no GPU: codec self-check)
//
// Pixels go through the HIP kernels only: without a device the commands fail.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../../include/nuscaler_hip.h"
#include "nus_png.hpp"

namespace {

int fail(const std::string &msg)
{
    std::fprintf(stderr, "nu_scaler_cli: error: %s\n", msg.c_str());
    return 1;
}

int usage(int rc)
{
    std::fprintf(rc ? stderr : stdout,
                 "usage: nu_scaler_cli upscale <in.png> <out.png> [--algorithm nearest|bilinear|bicubic|lanczos3|triangle|fsr1|easu]\n"
                 "                             [--scale S] [--tech fsr|fallback|none] [--quality ultra|quality|balanced|performance]\n"
                 "                             [--device N]\n"
                 "       nu_scaler_cli interpolate <a.png> <b.png> <out.png> [--t X] [--flow] [--device N]\n"
                 "       nu_scaler_cli png-copy <in.png> <out.png>\n");
    return rc;
}

struct Args {
    std::vector<std::string> positional;
    std::map<std::string, std::string> options;
    bool flow = false;
};

bool parse(int argc, char **argv, Args &a, std::string &err)
{
    for (int i = 2; i < argc; ++i) {
        const std::string s = argv[i];
        if (s == "--flow") {
            a.flow = true;
        } else if (s.rfind("--", 0) == 0) {
            if (i + 1 >= argc) {
                err = "option " + s + " needs a value";
                return false;
            }
            a.options[s.substr(2)] = argv[++i];
        } else {
            a.positional.push_back(s);
        }
    }
    return true;
}

std::string lower(std::string s)
{
    for (char &c : s) c = (char)std::tolower((unsigned char)c);
    return s;
}

int quality_of(const std::string &q)
{
    const std::string s = lower(q);
    if (s == "ultra") return NUS_QUALITY_ULTRA;
    if (s == "balanced") return NUS_QUALITY_BALANCED;
    if (s == "performance") return NUS_QUALITY_PERFORMANCE;
    return NUS_QUALITY_QUALITY; // unknown strings default, as lib.rs:51-57
}

int cmd_upscale(const Args &a)
{
    if (a.positional.size() != 2) return usage(2);
    const auto opt = [&](const char *k, const char *dflt) {
        auto it = a.options.find(k);
        return it == a.options.end() ? std::string(dflt) : it->second;
    };
    const std::string tech = lower(opt("tech", "fallback")), quality = lower(opt("quality", "quality"));
    const float scale = (float)std::atof(opt("scale", "2.0").c_str());
    nus_cli::Image in;
    std::string err = nus_cli::read_png(a.positional[0], in);
    if (!err.empty()) return fail(err);
    if (tech == "none") { // PassThroughUpscaler
        err = nus_cli::write_png(a.positional[1], in);
        if (!err.empty()) return fail(err);
        std::printf("%s: %ux%u\n", a.positional[1].c_str(), in.width, in.height);
        return 0;
    }
    if (tech == "dlss") return fail("technology 'dlss' is not available on this device");
    if (tech != "fallback" && tech != "fsr" && tech != "wgpu") return fail("unknown technology '" + tech + "'");
    // (w as f32 * scale) as u32 -- Nu_scale/src/upscale/mod.rs:320-321
    const uint32_t ow = (uint32_t)((float)in.width * scale), oh = (uint32_t)((float)in.height * scale);
    if (ow == 0 || oh == 0) return fail("scale factor gives an empty output image");
    std::string alg = lower(opt("algorithm", ""));
    if (tech == "fsr") {
        alg = "fsr1";
    } else if (alg.empty()) { // quality -> algorithm, Nu_scale/src/upscale/mod.rs:295-303
        alg = quality == "ultra" ? "lanczos3" : (quality == "performance" ? "bilinear" : "bicubic");
    }
    static const std::map<std::string, int> kAlg = {
        {"nearest", NUS_ALG_NEAREST}, {"bilinear", NUS_ALG_BILINEAR}, {"lanczos3", NUS_ALG_LANCZOS3}, {"lanczos", NUS_ALG_LANCZOS3},
        {"bicubic", NUS_ALG_BICUBIC}, {"catmullrom", NUS_ALG_BICUBIC}, {"triangle", NUS_ALG_TRIANGLE},
        {"fsr1", NUS_ALG_FSR1}, {"fsr", NUS_ALG_FSR1}, {"easu", NUS_ALG_FSR_EASU}};
    const auto it = kAlg.find(alg);
    nus_upscaler *u = nus_upscaler_create(it == kAlg.end() ? NUS_ALG_NEAREST : it->second, quality_of(quality));
    if (!u) return fail(nus_last_error());
    int rc = nus_upscaler_set_device(u, std::atoi(opt("device", "0").c_str()));
    if (rc == NUS_OK) rc = nus_upscaler_initialize(u, in.width, in.height, ow, oh);
    nus_cli::Image out;
    out.width = ow;
    out.height = oh;
    out.rgba.resize((size_t)ow * oh * 4);
    if (rc == NUS_OK) rc = nus_upscaler_upscale(u, in.rgba.data(), in.rgba.size(), out.rgba.data(), out.rgba.size());
    if (rc != NUS_OK) {
        const std::string msg = nus_upscaler_last_error(u);
        nus_upscaler_destroy(u);
        return fail(msg);
    }
    nus_upscaler_destroy(u);
    err = nus_cli::write_png(a.positional[1], out);
    if (!err.empty()) return fail(err);
    std::printf("%s: %ux%u\n", a.positional[1].c_str(), ow, oh);
    return 0;
}

int cmd_interpolate(const Args &a)
{
    if (a.positional.size() != 3) return usage(2);
    nus_cli::Image fa, fb;
    std::string err = nus_cli::read_png(a.positional[0], fa);
    if (err.empty()) err = nus_cli::read_png(a.positional[1], fb);
    if (!err.empty()) return fail(err);
    if (fa.width != fb.width || fa.height != fb.height) return fail("frame sizes differ");
    const auto t_it = a.options.find("t");
    const float t = t_it == a.options.end() ? 0.5f : (float)std::atof(t_it->second.c_str());
    const auto d_it = a.options.find("device");
    const int device = d_it == a.options.end() ? 0 : std::atoi(d_it->second.c_str());
    std::vector<float> flow;
    if (a.flow) { // pyramid + Horn-Schunck front end instead of the reference's zero flow
        nus_flow *f = nus_flow_create();
        if (!f) return fail(nus_last_error());
        flow.resize((size_t)fa.width * fa.height * 2);
        int rc = nus_flow_set_device(f, device);
        // 3 levels, 50 coarse + 10 refining Jacobi steps, lambda as the Python mirror's FlowEstimator defaults
        if (rc == NUS_OK)
            rc = nus_flow_estimate(f, fa.rgba.data(), fb.rgba.data(), fa.width, fa.height, 3, 50, 10, 0.0004f, flow.data());
        const std::string msg = rc == NUS_OK ? "" : nus_flow_last_error(f);
        nus_flow_destroy(f);
        if (rc != NUS_OK) return fail(msg);
    }
    nus_interp *it = nus_interp_create(NUS_WG_WIDE_32X8);
    if (!it) return fail(nus_last_error());
    nus_cli::Image out;
    out.width = fa.width;
    out.height = fa.height;
    out.rgba.resize(fa.rgba.size());
    int rc = nus_interp_set_device(it, device);
    if (rc == NUS_OK)
        rc = nus_interp_interpolate(it, fa.rgba.data(), fa.rgba.size(), fb.rgba.data(), fb.rgba.size(),
                                    a.flow ? flow.data() : nullptr, fa.width, fa.height, t, out.rgba.data(), out.rgba.size());
    const std::string msg = rc == NUS_OK ? "" : nus_interp_last_error(it);
    nus_interp_destroy(it);
    if (rc != NUS_OK) return fail(msg);
    err = nus_cli::write_png(a.positional[2], out);
    if (!err.empty()) return fail(err);
    std::printf("%s: %ux%u\n", a.positional[2].c_str(), out.width, out.height);
    return 0;
}

int cmd_png_copy(const Args &a)
{
    if (a.positional.size() != 2) return usage(2);
    nus_cli::Image img;
    std::string err = nus_cli::read_png(a.positional[0], img);
    if (err.empty()) err = nus_cli::write_png(a.positional[1], img);
    if (!err.empty()) return fail(err);
    std::printf("%s: %ux%u\n", a.positional[1].c_str(), img.width, img.height);
    return 0;
}

} // namespace

int main(int argc, char **argv)
{
    if (argc < 2) return usage(2);
    const std::string cmd = argv[1];
    if (cmd == "--help" || cmd == "-h" || cmd == "help") return usage(0);
    Args a;
    std::string err;
    if (!parse(argc, argv, a, err)) return fail(err);
    if (cmd == "upscale") return cmd_upscale(a);
    if (cmd == "interpolate") return cmd_interpolate(a);
    if (cmd == "png-copy") return cmd_png_copy(a);
    return usage(2);
}
