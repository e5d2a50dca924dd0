// A shape and a material the library does not know, written the way a user of the reference writes them -- subclasses of
// drt::Shape<T> (shape.hpp:11-35) and drt::BxDF<T> (bxdf.hpp:12-25) -- rendered by the device through drt::hip::render and by the host API's own per-ray loop (render.cpp:72-86)
// on the same per-path random streams.  The one additive piece is describe(): the record and the two bodies as HIP source.
// Prints "ok <max relative gradient difference>" and exits 0, or says what differs.
#include <cmath>
#include <cstdio>
#include <memory>
#include <tuple>
#include <vector>

#include "drt/bxdf.hpp"
#include "drt/camera.hpp"
#include "drt/emitter.hpp"
#include "drt/hip.hpp"
#include "drt/integrate.hpp"
#include "drt/pathtracer.hpp"
#include "drt/random.hpp"
#include "drt/shape.hpp"
#include "drt/vector.hpp"

using namespace drt;
using T = double;
using Vec3 = Vector<T, 3>;
using Var3 = Vector<T, 3, true>;

class Disc : public Shape<T> {
public:
    Disc(Vec3 c, Vec3 n, double r, std::shared_ptr<BxDF<T>> bxdf = nullptr, std::shared_ptr<Emitter<T>> emitter = nullptr)
      : Shape<T>(bxdf, emitter), m_c(c), m_n(n), m_r(r) { }
    bool intersect(Vec3 orig, Vec3 dir, double& t) const override
    {
        const double den = dot(dir, m_n);
        if (den == 0)
            return false;
        t = dot(m_c - orig, m_n) / den;
        if (!(t > 0))
            return false;
        const Vec3 q = (orig + dir * t) - m_c;
        return dot(q, q) <= m_r * m_r;
    }
    Vec3 normal(Vec3) const override { return m_n; }
    ShapeRecord describe() const override
    {
        ShapeRecord r;
        r.kind = ShapeKind::User;
        r.kind_name = "disc";
        r.p[0] = m_c[0]; r.p[1] = m_c[1]; r.p[2] = m_c[2]; r.p[3] = m_n[0];
        r.q[0] = m_n[1]; r.q[1] = m_n[2]; r.q[2] = m_r;
        r.intersect_src =
            "const V3<R> c = mk<R>(p[0], p[1], p[2]), n = mk<R>(p[3], p[4], p[5]);\n"
            "const R den = dot(d, n);\n"
            "if (den == R(0)) return false;\n"
            "t = dot(c - o, n) / den;\n"
            "if (!(t > R(0))) return false;\n"
            "const V3<R> q = (o + d * t) - c;\n"
            "return dot(q, q) <= p[6] * p[6];\n";
        r.normal_src = "(void)P; return mk<R>(p[3], p[4], p[5]);\n";
        return r;
    }
private:
    Vec3 m_c, m_n;
    double m_r;
};

// ... and a material the library does not know: a power-cosine lobe around the normal, colour x (k + 2) / (2 pi) cos^k
class CosLobe : public BxDF<T> {
public:
    CosLobe(const Var3& color, double k) : m_color(color), m_k(k) { }
    Var3 operator()(const Vec3& normal, const Vec3&, const Vec3& dir_out) const override
    {
        const double c = dot(normal, dir_out);
        const double factor = c > 0 ? (m_k + 2) / (2 * pi) * std::pow(c, m_k) : 0.0;
        return factor * m_color;
    }
    std::tuple<Vec3, double> sample(const Vec3& normal, const Vec3&) const override
    {
        const double cos_t = std::pow(random::uniform(), 1 / (m_k + 1));
        const double theta = std::acos(cos_t);
        const double phi = 2 * pi * random::uniform();
        const auto frame = internal::make_frame(normal);
        return std::make_tuple(internal::angle_to_dir(theta, phi, frame), (m_k + 1) / (2 * pi) * std::pow(cos_t, m_k));
    }
    BxDFKind kind() const override { return BxDFKind::User; }
    const Var3* parameter() const override { return &m_color; }
    double exponent() const override { return m_k; }
    const char* device_kind_name() const override { return "coslobe"; }
    const char* device_sample_src() const override
    {
        return "(void)d; const R k = p[0]; const R ct = pow_r(u1, R(1) / (k + R(1))); const R st = sqrt_r(max_r(R(0), R(1) - ct * ct));\n"
               "R sphi, cphi; sincospi_r(R(2) * u2, &sphi, &cphi); V3<R> t, b; make_frame(n, t, b);\n"
               "wo = t * (cphi * st) + b * (sphi * st) + n * ct; pdf = (k + R(1)) * R(0.15915494309189535) * pow_r(ct, k);\n"
               "const R c = dot(n, wo); bs = c > R(0) ? (k + R(2)) * R(0.15915494309189535) * pow_r(c, k) : R(0);\n";
    }
private:
    Var3 m_color;
    double m_k;
};

int main()
{
    // the reference's scene, render.cpp:26-59, and a tilted disc with an albedo of its own in it
    Var3 red(Vec3{0.5, 0., 0.}, true), green(Vec3{0., 0.5, 0.}, true), white(Vec3(0.5), true), emission(Vec3(1.), true);
    Var3 disc_albedo(Vec3{0.7, 0.6, 0.2}, true), lobe_albedo(Vec3{0.8, 0.7, 0.5}, true);
    auto diffuse_red = std::make_shared<DiffuseBxDF<T>>(red);
    auto diffuse_green = std::make_shared<DiffuseBxDF<T>>(green);
    auto diffuse_white = std::make_shared<DiffuseBxDF<T>>(white);
    auto diffuse_disc = std::make_shared<DiffuseBxDF<T>>(disc_albedo);
    auto emitter = std::make_shared<AreaEmitter<T>>(emission);
    auto lobe = std::make_shared<CosLobe>(lobe_albedo, 6.0);
    Sphere<T> sphere_front(Vec3{0., 0., 3.}, 1., lobe);
    Sphere<T> sphere_back(Vec3{-1., 1., 4.5}, 1., diffuse_white);
    Plane<T> left(Vec3{-1., 0., 0.}, -3., diffuse_red), right(Vec3{1., 0., 0.1}, -3., diffuse_green);
    Plane<T> back(Vec3{0., 0., -1.}, -6., diffuse_white), front(Vec3{0, 0, 1}, 0, diffuse_white);
    Plane<T> ground(Vec3{0., 1., 0.}, -3., diffuse_white), ceiling(Vec3{0., -1., 0.}, -3., diffuse_white);
    const Vec3 n = normalize(Vec3{0.2, 1.0, -0.3});
    Disc disc(Vec3{0.9, -1.4, 3.3}, n, 0.9, diffuse_disc);
    Sphere<T> light(Vec3{0., 3., 3.}, 1., nullptr, emitter);
    Scene<T> scene{&sphere_front, &sphere_back, &left, &right, &back, &front, &ground, &ceiling, &disc, &light};

    const std::size_t W = 40, H = 32, spp = 6;
    Camera<T> cam(W, H);
    cam.look_at(Vec3{0., 0., 0.}, Vec3{0., 0., 1.});
    Pathtracer<T> tracer(1.0, 5);
    std::vector<Var3*> params = {&red, &green, &white, &emission, &disc_albedo, &lobe_albedo};

    // the host API's own loop (render.cpp:72-86) on the device's per-path streams
    std::vector<Vec3> cpu(W * H, Vec3(0.));
    for (std::size_t y = 0; y < H; ++y)
        for (std::size_t x = 0; x < W; ++x)
            for (std::size_t i = 0; i < spp; ++i) {
                random::begin_path(7u, (uint64_t)(y * W + x) * spp + i);
                Vec3 dir;
                double pdf;
                std::tie(dir, pdf) = cam.sample(x, y);
                Var3 radiance = tracer.trace(scene, cam.eye(), dir);
                cpu[y * W + x] += radiance.detach() / pdf / double(spp);
                if (radiance.requires_grad())
                    radiance.backward(Vec3(1.));
            }
    random::use_libc();
    std::vector<Vec3> g_cpu;
    for (Var3* p : params) {
        g_cpu.push_back(p->grad());
        p->grad() = Vec3(0.);
    }

    std::vector<Vec3> dev(W * H, Vec3(0.));
    hip::Options opt;
    opt.backward = true;
    opt.f64 = true;
    opt.seed = 7;
    hip::Stats st = hip::render(scene, cam, tracer, spp, dev.data(), opt);
    double worst = 0, scale = 0;
    for (std::size_t k = 0; k < params.size(); ++k)
        for (int c = 0; c < 3; ++c) {
            scale = std::fmax(scale, std::fabs(g_cpu[k][c]));
            worst = std::fmax(worst, std::fabs(params[k]->grad()[c] - g_cpu[k][c]));
        }
    double img_worst = 0;
    for (std::size_t i = 0; i < W * H; ++i)
        for (int c = 0; c < 3; ++c)
            img_worst = std::fmax(img_worst, std::fabs(dev[i][c] - cpu[i][c]));
    hip::release_contexts();
    if (!(worst <= 1e-9 * scale) || !(img_worst <= 1e-6) || !(std::fabs(g_cpu[4][0]) > 0) || !(std::fabs(g_cpu[5][0]) > 0)) {
        std::printf("FAILED: gradient difference %.3g of %.3g, image difference %.3g, d/d(disc albedo) %.3g\n", worst, scale, img_worst, g_cpu[4][0]);
        return 1;
    }
    std::printf("ok %.3g (%llu segments)\n", worst / scale, st.segments);
    return 0;
}
