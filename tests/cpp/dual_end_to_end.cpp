// T = Dual<double> end to end on the host API (in the reference this does not
// compile): forward-mode derivative of the render w.r.t. one parameter channel
// must equal the reverse-mode gradient of the same render with T = double, same RNG streams.
#include <cmath>
#include <cstdio>
#include <memory>

#include "drt/bxdf.hpp"
#include "drt/camera.hpp"
#include "drt/dual.hpp"
#include "drt/emitter.hpp"
#include "drt/integrate.hpp"
#include "drt/pathtracer.hpp"
#include "drt/shape.hpp"
#include "drt/vector.hpp"

using namespace drt;

// value and, for Dual, derivative part of a scalar
static double val(double x) { return x; }
static double val(const Dual<double>& x) { return x.real(); }
static double der(double) { return 0; }
static double der(const Dual<double>& x) { return x.dual(); }

template <typename T>
struct Result { double sum[3], dsum[3], grad_red[3], grad_emis[3]; };

template <typename T>
Result<T> render(T red_r, T emis_g, bool backward)
{
    Vector<T, 3, true> red(Vector<T, 3>{red_r, T(0), T(0)}, true);
    Vector<T, 3, true> green(Vector<T, 3>{T(0), T(0.5), T(0)}, true);
    Vector<T, 3, true> white(Vector<T, 3>{T(0.5), T(0.5), T(0.5)}, true);
    Vector<T, 3, true> emission(Vector<T, 3>{T(1), emis_g, T(1)}, true);
    auto dred = std::make_shared<DiffuseBxDF<T>>(red);
    auto dgreen = std::make_shared<DiffuseBxDF<T>>(green);
    auto dwhite = std::make_shared<DiffuseBxDF<T>>(white);
    auto spec = std::make_shared<SpecularBxDF<T>>(white, 30);
    auto emitter = std::make_shared<AreaEmitter<T>>(emission);
    Sphere<T> s1(Vector<T, 3>{0., 0., 3.}, 1., spec), s2(Vector<T, 3>{-1., 1., 4.5}, 1., dwhite);
    Plane<T> p1(Vector<T, 3>{-1., 0., 0.}, -3., dred), p2(Vector<T, 3>{1., 0., 0.1}, -3., dgreen),
        p3(Vector<T, 3>{0., 0., -1.}, -6., dwhite), p4(Vector<T, 3>{0, 0, 1}, 0, dwhite),
        p5(Vector<T, 3>{0., 1., 0.}, -3., dwhite), p6(Vector<T, 3>{0., -1., 0.}, -3., dwhite);
    Sphere<T> light(Vector<T, 3>{0., 3., 3.}, 1., nullptr, emitter);
    Scene<T> scene{&s1, &s2, &p1, &p2, &p3, &p4, &p5, &p6, &light};
    const std::size_t W = 24, H = 18, spp = 4;
    Camera<T> cam(W, H);
    cam.look_at(Vector<T, 3>{0, 0, 0}, Vector<T, 3>{0, 0, 1});
    Pathtracer<T> tracer(0.3, 2);
    Result<T> r{};
    for (std::size_t y = 0; y < H; ++y)
        for (std::size_t x = 0; x < W; ++x)
            for (std::size_t i = 0; i < spp; ++i) {
                random::begin_path(11, (uint64_t)(y * W + x) * spp + i);
                auto [dir, pdf] = cam.sample(x, y);
                auto radiance = tracer.trace(scene, cam.eye(), dir);
                for (int c = 0; c < 3; ++c) {
                    r.sum[c] += val(radiance[c]);
                    r.dsum[c] += der(radiance[c]);
                }
                if (backward)
                    radiance.backward(Vector<T, 3>(T(1)));
            }
    if (backward)
        for (int c = 0; c < 3; ++c) {
            r.grad_red[c] = val(red.grad()[c]);
            r.grad_emis[c] = val(emission.grad()[c]);
        }
    return r;
}

static bool close(double a, double b) { return std::fabs(a - b) <= 1e-10 * (1 + std::fabs(b)); }

int main()
{
    const Result<double> rev = render<double>(0.5, 1.0, true);
    // d/d red.r : seed the dual part of red.r
    const Result<Dual<double>> f1 = render<Dual<double>>(Dual<double>(0.5, 1.0), Dual<double>(1.0, 0.0), false);
    // d/d emission.g
    const Result<Dual<double>> f2 = render<Dual<double>>(Dual<double>(0.5, 0.0), Dual<double>(1.0, 1.0), false);
    for (int c = 0; c < 3; ++c)
        if (!close(f1.sum[c], rev.sum[c]) || !close(f2.sum[c], rev.sum[c])) { std::printf("value mismatch\n"); return 1; }
    // reverse mode back-propagates the seed (1,1,1): grad[param channel] = d(sum of all radiance channels)/d that channel...
    // red.r only feeds channel r of every product, emission.g only channel g
    if (!close(f1.dsum[0], rev.grad_red[0]) || !close(f1.dsum[1], 0) || !close(f1.dsum[2], 0)) {
        std::printf("d/d red.r: forward %.12g vs reverse %.12g\n", f1.dsum[0], rev.grad_red[0]);
        return 1;
    }
    if (!close(f2.dsum[1], rev.grad_emis[1]) || !close(f2.dsum[0], 0)) {
        std::printf("d/d emission.g: forward %.12g vs reverse %.12g\n", f2.dsum[1], rev.grad_emis[1]);
        return 1;
    }
    std::printf("ok %.10g %.10g\n", f1.dsum[0], f2.dsum[1]);
    return 0;
}
