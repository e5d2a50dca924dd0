// args.hpp -- drt::Args / drt::parse_args with the flags and defaults of the reference's CLI
// (src/args.hpp:19-67): -x/--width 640, -y/--height 480, -n/--samples 100, -b/--min-bounces 1,
// -p/--absorb-prob 0.5, -o/--output (required), -h/--help, --version (0.1).  TCLAP (branch 1.4 in
// the reference's .gitmodules) is not available here, so this is a small self-contained parser.
// Additive flags for the device path: --backend cpu|hip, --backward, --seed, --max-depth,
// --devices a,b,..., --f64, --unbiased (backward with integrate(..., unbiased = true)),
// --front diffuse|specular|mirror (the BxDF of sphere_front).
#pragma once

#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace drt {

struct Args {
    std::size_t width;
    std::size_t height;
    std::size_t samples;
    std::size_t min_bounces;
    double absorb_prob;
    std::string output;
    // additive
    std::string backend = "hip";
    bool backward = false;
    bool unbiased = false;
    unsigned seed = 1;
    int max_depth = 0;
    std::vector<int> devices = {0};
    bool f64 = false;
    int repeat = 1;                   // --repeat N: render N times (an optimisation loop's cost per iteration)
    std::string front = "diffuse";   // BxDF of sphere_front: diffuse (render.cpp:39) | specular (render.cpp:35's) | mirror
};

inline bool parse_args(int argc, const char* const* argv, Args* args)
{
    args->width = 640;
    args->height = 480;
    args->samples = 100;
    args->min_bounces = 1;
    args->absorb_prob = 0.5;
    args->output.clear();
    bool have_output = false;
    auto usage = [&](FILE* f) {
        std::fprintf(f,
            "USAGE: %s -o <string> [-x <integer>] [-y <integer>] [-n <integer>] [-b <integer>] [-p <number>]\n"
            "       [--backend cpu|hip] [--backward] [--unbiased] [--seed <integer>] [--max-depth <integer>]\n"
            "       [--devices a,b,...] [--f64] [--front diffuse|specular|mirror] [--repeat <integer>] [--version] [-h]\n\n"
            "A simple differentiable path tracer\n"
            "  -x, --width        Output image width (640)\n"
            "  -y, --height       Output image height (480)\n"
            "  -n, --samples      Number of samples per pixel (100)\n"
            "  -b, --min-bounces  Min. number of light bounces (1)\n"
            "  -p, --absorb-prob  Ray absorbption prob. per bounce (after min. bounces) (0.5)\n"
            "  -o, --output       Output path (required)\n", argc > 0 ? argv[0] : "render");
    };
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto value = [&](const char*& out) { if (i + 1 >= argc) return false; out = argv[++i]; return true; };
        const char* v = nullptr;
        char* end = nullptr;
        if (a == "-h" || a == "--help") { usage(stdout); std::exit(0); }
        else if (a == "--version") { std::printf("%s  version: 0.1\n", argc > 0 ? argv[0] : "render"); std::exit(0); }
        else if (a == "-x" || a == "--width") { if (!value(v)) return false; args->width = std::strtoull(v, &end, 10); if (*end) return false; }
        else if (a == "-y" || a == "--height") { if (!value(v)) return false; args->height = std::strtoull(v, &end, 10); if (*end) return false; }
        else if (a == "-n" || a == "--samples") { if (!value(v)) return false; args->samples = std::strtoull(v, &end, 10); if (*end) return false; }
        else if (a == "-b" || a == "--min-bounces") { if (!value(v)) return false; args->min_bounces = std::strtoull(v, &end, 10); if (*end) return false; }
        else if (a == "-p" || a == "--absorb-prob") { if (!value(v)) return false; args->absorb_prob = std::strtod(v, &end); if (*end) return false; }
        else if (a == "-o" || a == "--output") { if (!value(v)) return false; args->output = v; have_output = true; }
        else if (a == "--backend") { if (!value(v)) return false; args->backend = v; if (args->backend != "cpu" && args->backend != "hip") return false; }
        else if (a == "--backward") { args->backward = true; }
        else if (a == "--unbiased") { args->backward = true; args->unbiased = true; }
        else if (a == "--f64") { args->f64 = true; }
        else if (a == "--seed") { if (!value(v)) return false; args->seed = (unsigned)std::strtoul(v, &end, 10); if (*end) return false; }
        else if (a == "--max-depth") { if (!value(v)) return false; args->max_depth = (int)std::strtol(v, &end, 10); if (*end) return false; }
        else if (a == "--repeat") { if (!value(v)) return false; args->repeat = (int)std::strtol(v, &end, 10); if (*end || args->repeat < 1) return false; }
        else if (a == "--front") { if (!value(v)) return false; args->front = v; if (args->front != "diffuse" && args->front != "specular" && args->front != "mirror") return false; }
        else if (a == "--devices") {
            if (!value(v)) return false;
            args->devices.clear();
            for (const char* p = v; *p;) {
                args->devices.push_back((int)std::strtol(p, &end, 10));
                if (end == p) return false;
                p = *end == ',' ? end + 1 : end;
                if (*end && *end != ',') return false;
            }
            if (args->devices.empty()) return false;
        }
        else { std::fprintf(stderr, "PARSE ERROR: Argument: %s\n             Couldn't find match for argument\n", a.c_str()); usage(stderr); return false; }
    }
    if (!have_output) {
        std::fprintf(stderr, "PARSE ERROR:\n             Required argument missing: output\n");
        usage(stderr);
        return false;
    }
    return true;
}

} // namespace drt
