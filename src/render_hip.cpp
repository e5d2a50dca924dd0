// render_hip.cpp -- the sample application of the reference (src/render.cpp) with the pixel x
// sample loop (render.cpp:72-86) replaced by ONE call of drt::hip::render, which runs the whole
// frame through the MI355X wavefront pipeline.  Scene, camera and flags are the reference's
// (render.cpp:26-69).  --backend cpu keeps the per-ray loop on the host API for comparison.
#include <chrono>
#include <deque>
#include <cstdio>
#include <cstdlib>

#include "drt/bxdf.hpp"
#include "drt/camera.hpp"
#include "drt/dual.hpp"
#include "drt/emitter.hpp"
#include "drt/hip.hpp"
#include "drt/integrate.hpp"
#include "drt/pathtracer.hpp"
#include "drt/shape.hpp"
#include "drt/vector.hpp"
#include "args.hpp"
#include "write.hpp"

using namespace drt;

int main(int argc, const char* argv[])
{
    Args args;
    if (!parse_args(argc, argv, &args))
        return EXIT_FAILURE;

    using T = double;

    // scene parameters
    Vector<T, 3, true> red(Vector<T, 3>{0.5, 0, 0}, true);
    Vector<T, 3, true> green(Vector<T, 3>{0, 0.5, 0}, true);
    Vector<T, 3, true> white(Vector<T, 3>{0.5, 0.5, 0.5}, true);
    Vector<T, 3, true> emission(Vector<T, 3>(1), true);

    // materials
    auto diffuse_red = std::make_shared<DiffuseBxDF<T>>(red);
    auto diffuse_green = std::make_shared<DiffuseBxDF<T>>(green);
    auto diffuse_white = std::make_shared<DiffuseBxDF<T>>(white);
    auto emitter = std::make_shared<AreaEmitter<T>>(emission);
    std::shared_ptr<BxDF<T>> front = diffuse_white;                                  // render.cpp:39
    if (args.front == "specular")
        front = std::make_shared<SpecularBxDF<T>>(white, 30);                        // render.cpp:35
    else if (args.front == "mirror")
        front = std::make_shared<MirrorBxDF<T>>();

    // shapes
    Sphere<T> sphere_front(Vector<T, 3>{0., 0., 3.}, 1., front);
    Sphere<T> sphere_back(Vector<T, 3>{-1., 1., 4.5}, 1., diffuse_white);
    Plane<T> left_plane(Vector<T, 3>{-1., 0., 0.}, -3., diffuse_red);
    Plane<T> right_plane(Vector<T, 3>{1., 0., 0.1}, -3., diffuse_green);
    Plane<T> back_plane(Vector<T, 3>{0., 0., -1.}, -6., diffuse_white);
    Plane<T> front_plane(Vector<T, 3>{0, 0, 1}, 0, diffuse_white);
    Plane<T> ground_plane(Vector<T, 3>{0., 1., 0.}, -3., diffuse_white);
    Plane<T> ceiling_plane(Vector<T, 3>{0., -1., 0.}, -3., diffuse_white);
    Sphere<T> light(Vector<T, 3>{0., 3., 3.}, 1., nullptr, emitter);

    Scene<T> scene{&sphere_front, &sphere_back, &left_plane, &right_plane, &back_plane,
                   &front_plane, &ground_plane, &ceiling_plane, &light};

    const std::size_t width = args.width, height = args.height;
    Camera<T> cam(width, height);
    cam.look_at(Vector<T, 3>{0, 0, 0}, Vector<T, 3>{0, 0, 1});
    std::vector<Vector<double, 3>> img(width * height, Vector<double, 3>(0.));
    Pathtracer<T> tracer(args.absorb_prob, args.min_bounces);

    const auto t0 = std::chrono::steady_clock::now();
    unsigned long long segments = 0;
    if (args.backend == "hip") {
        hip::Options opt;
        opt.backward = args.backward;
        opt.unbiased = args.unbiased;
        opt.seed = args.seed;
        opt.max_depth = args.max_depth;
        opt.devices = args.devices;
        opt.f64 = args.f64;
        // --repeat N: what an optimisation loop pays per iteration (the device context and its queues are
        // kept between calls; gradients accumulate like the reference's autograd, so they are zeroed in between)
        for (int it = 0; it < args.repeat; ++it) {
            if (it > 0) {
                for (Vector<T, 3, true>* p : {&red, &green, &white, &emission})
                    p->grad() = Vector<T, 3>(0.);
            }
            const auto ti = std::chrono::steady_clock::now();
            const hip::Stats st = hip::render(scene, cam, tracer, args.samples, img.data(), opt);
            segments = st.segments;
            if (args.repeat > 1)
                std::printf("call %d: %.3f ms\n", it,
                            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ti).count());
        }
        if (args.repeat > 2 && args.devices.size() == 1) {
            // the same frames with up to four in flight (drt::hip::submit / Pending::get): a frame's copy to the host overlaps
            // the next frames' kernels, and the path kernels of consecutive frames overlap.  (The first frames set up the
            // sets of buffers: timed from the fifth on.)
            const int slots = DRT_HIP_FRAMES_IN_FLIGHT;
            std::vector<std::vector<Vector<double, 3>>> imgs((std::size_t)slots, std::vector<Vector<double, 3>>(width * height, Vector<double, 3>(0.)));
            auto zero_grads = [&]() {
                for (Vector<T, 3, true>* p : {&red, &green, &white, &emission})
                    p->grad() = Vector<T, 3>(0.);
            };
            const int n_pipe = args.repeat + slots;
            auto tp = std::chrono::steady_clock::now();
            std::deque<hip::Pending<T>> flying;
            for (int it = 0; it < n_pipe; ++it) {
                if (it == slots)
                    tp = std::chrono::steady_clock::now();
                flying.push_back(hip::submit(scene, cam, tracer, args.samples, imgs[(std::size_t)(it % slots)].data(), opt));
                if ((int)flying.size() >= slots) {
                    zero_grads();
                    flying.front().get();
                    flying.pop_front();
                }
            }
            while (!flying.empty()) {
                zero_grads();
                flying.front().get();
                flying.pop_front();
            }
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tp).count() / args.repeat;
            img = imgs[(std::size_t)((n_pipe - 1) % slots)];
            std::printf("pipelined (up to %d frames in flight): %.3f ms per frame\n", slots - 1, ms);
        }
    } else {
        // the reference's loop on the host API, drawing the same per-path RNG streams
        for (std::size_t y = 0; y < cam.height(); ++y) {
            for (std::size_t x = 0; x < cam.width(); ++x) {
                Vector<T, 3> pixel_radiance(0);
                for (std::size_t i = 0; i < args.samples; ++i) {
                    random::begin_path(args.seed, (uint64_t)(y * width + x) * args.samples + i);
                    auto [dir, pdf] = cam.sample(x, y);
                    Vector<T, 3, true> radiance = tracer.trace(scene, cam.eye(), dir);
                    pixel_radiance += radiance.detach() / pdf;
                    if (args.backward)
                        radiance.backward(Vector<T, 3>(1));
                }
                img[y * width + x] = pixel_radiance / args.samples;
            }
            std::printf("% 5.2f%%\r", 100. * (y + 1) / cam.height());
            std::fflush(stdout);
        }
        std::printf("\n");
    }
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::printf("rendered %zux%zu, %zu spp on %s in %.3f s", width, height, args.samples, args.backend.c_str(), secs);
    if (segments)
        std::printf(" (%llu rays, %.1f Mray/s)", segments, segments / secs * 1e-6);
    std::printf("\n");
    if (args.backward) {
        std::printf("grad red      = (%.9g, %.9g, %.9g)\n", red.grad()[0], red.grad()[1], red.grad()[2]);
        std::printf("grad green    = (%.9g, %.9g, %.9g)\n", green.grad()[0], green.grad()[1], green.grad()[2]);
        std::printf("grad white    = (%.9g, %.9g, %.9g)\n", white.grad()[0], white.grad()[1], white.grad()[2]);
        std::printf("grad emission = (%.9g, %.9g, %.9g)\n", emission.grad()[0], emission.grad()[1], emission.grad()[2]);
    }
    write_exr(args.output.c_str(), img.data(), width, height);
    return 0;
}
