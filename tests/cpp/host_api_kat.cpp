// Known-answer tests of the drt:: host API (README.md:44-101).
// Prints "ok" and exits 0, or reports the first failure.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <sstream>
#include <string>

#include "drt/bxdf.hpp"
#include "drt/camera.hpp"
#include "drt/dual.hpp"
#include "drt/emitter.hpp"
#include "drt/hip.hpp"
#include "drt/integrate.hpp"
#include "drt/mesh.hpp"
#include "drt/pathtracer.hpp"
#include "drt/shape.hpp"
#include "drt/vector.hpp"

using namespace drt;
using V = Vector<double, 3>;
using P = Vector<double, 3, true>;

#define CHECK(cond) do { if (!(cond)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); return 1; } } while (0)
static bool close(double a, double b, double tol = 1e-12) { return std::fabs(a - b) <= tol * (1 + std::fabs(b)); }

int main()
{
    // plain vectors
    V a{1, 2, 3}, b{4, 5, 6};
    CHECK((a + b)[2] == 9 && (b - a)[0] == 3 && (a * b)[1] == 10 && (b / a)[2] == 2);
    CHECK((2 * a)[2] == 6 && (a * 2)[0] == 2 && (a / 2)[1] == 1 && (-a)[0] == -1);
    CHECK(dot(a, b) == 32 && close(norm(V{3, 4, 0}), 5) && close(norm(normalize(b)), 1));
    CHECK(cross(V{1, 0, 0}, V{0, 1, 0})[2] == 1);
    V r = reflect(V{1, -1, 0}, V{0, 1, 0});
    CHECK(r[0] == -1 && r[1] == -1 && r[2] == 0);
    try { V bad{1, 2}; (void)bad; CHECK(false); } catch (const std::runtime_error& e) {
        CHECK(std::string(e.what()) == "incorrect number of initializers for `Vector`"); }
    std::ostringstream os; os << V{1, 2, 3};
    CHECK(os.str() == "Vector<d, 3>{1, 2, 3}");
    std::ostringstream os2; os2 << P(V{1, 2, 3}, true);
    CHECK(os2.str() == "Vector<d, 3, true>{1, 2, 3}");

    // tape: z = (x*c)/x + 2x - x  =>  dz/dx = 1
    P x(V{1, 2, 3}, true);
    P c(V{4, 5, 6});
    P z = (x * c) / x + 2 * x - x;
    CHECK(z.requires_grad() && !c.requires_grad() && close(z[0], 5) && close(z[2], 9));
    z.backward(V(1.));
    CHECK(close(x.grad()[0], 1) && close(x.grad()[1], 1) && close(x.grad()[2], 1));
    z.backward(V(1.));                                  // gradients accumulate
    CHECK(close(x.grad()[1], 2));
    try { (void)c.grad(); CHECK(false); } catch (const std::runtime_error& e) {
        CHECK(std::string(e.what()) == "Vector has no gradient (not a variable)"); }
    P k = c + P(V{7, 8, 9});                            // constants only: not recorded
    CHECK(!k.requires_grad() && k[0] == 11);
    // product / quotient / scalar rules
    P u(V{2, 3, 4}, true), w(V{5, 6, 7}, true);
    (u * w / 2.0).backward(V{1, 1, 1});
    CHECK(close(u.grad()[0], 2.5) && close(w.grad()[2], 2.0));
    P q(V{2, 4, 8}, true), d(V{1, 2, 4}, true);
    (q / d).backward(V(1.));
    CHECK(close(q.grad()[1], 0.5) && close(d.grad()[2], -8.0 / 16.0));
    P m(V{1, 1, 1}, true);
    (-m).backward(V{1, 2, 3});
    CHECK(close(m.grad()[2], -3));
    // shared nodes: a copied handle is the same variable
    P shared = x;
    CHECK(shared.id() == x.id());
    // compound assignment rebinds the handle to a new node
    P acc(0.);
    acc += x * 3.0;
    acc.backward(V(1.));
    CHECK(close(x.grad()[0], 2 + 3));
    // custom backward (README.md:72-77)
    int called = 0;
    P custom(V{1, 1, 1}, [&](const V& g) { called += (int)g[0]; });
    custom.backward(V(7.));
    CHECK(called == 7 && custom.requires_grad());

    // integrate: biased records the samples, unbiased re-samples in backward (integrate.hpp)
    random::begin_path(1, 0);
    Vector<double, 1, true> slope(Vector<double, 1>{2.0}, true);
    auto fwd = [&](const double& s) { return slope * s; };
    auto smp = [&]() { return std::make_tuple(random::uniform(), 1.0); };
    auto biased = integrate<double, 1>(fwd, smp, 4, false);
    biased.backward(Vector<double, 1>(1.));
    CHECK(close(slope.grad()[0] * 2.0, biased[0]));     // same samples both ways
    slope.grad() = Vector<double, 1>(0.);
    auto unbiased = integrate<double, 1>(fwd, smp, 4, true);
    unbiased.backward(Vector<double, 1>(1.));
    CHECK(!close(slope.grad()[0] * 2.0, unbiased[0], 1e-6));   // fresh samples in backward
    random::use_libc();

    // dual numbers
    Dual<double> dx(3.0, 1.0);
    Dual<double> f = dx * dx + 2 * dx - 1 / dx;
    CHECK(close(f.real(), 9 + 6 - 1.0 / 3) && close(f.dual(), 6 + 2 + 1.0 / 9));
    CHECK(close(sqrt(Dual<double>(4.0, 1.0)).dual(), 0.25) && close(real(f), f.real()));

    // plugins + flattening for the device
    P red(V{0.5, 0, 0}, true), white(V{0.5, 0.5, 0.5}, true), emis(V(1.), true);
    auto mred = std::make_shared<DiffuseBxDF<double>>(red);
    auto mwhite = std::make_shared<DiffuseBxDF<double>>(white);
    auto mspec = std::make_shared<SpecularBxDF<double>>(white, 30);
    auto light = std::make_shared<AreaEmitter<double>>(emis);
    Sphere<double> s1(V{0, 0, 3}, 1, mwhite), s2(V{0, 3, 3}, 1, nullptr, light), s3(V{1, 1, 4}, 0.5, mspec);
    Plane<double> p1(V{1, 0, 0.1}, -3, mred);
    Scene<double> scene{&s1, &p1, &s2, &s3};
    auto flat = hip::flatten(scene);
    CHECK(flat.shapes.size() == 4 && flat.materials.size() == 3 && flat.emitters.size() == 1);
    CHECK(flat.requires_grad.size() == 3);               // white shared by two materials
    CHECK(flat.shapes[1].type == DRT_SHAPE_PLANE && flat.shapes[1].p[2] == 0.1 && flat.shapes[1].p[3] == -3);
    CHECK(flat.shapes[2].material == -1 && flat.shapes[2].emitter == 0);
    CHECK(flat.materials[2].type == DRT_BXDF_SPECULAR && flat.materials[2].exponent == 30 &&
          flat.materials[2].param == flat.materials[0].param);
    double t;
    CHECK(s1.intersect(V{0, 0, 0}, V{0, 0, 1}, t) && close(t, 2));
    CHECK(p1.intersect(V{0, 0, 0}, V{-1, 0, 0}, t) && close(t, 3));   // hit when moving against the normal
    CHECK(!p1.intersect(V{0, 0, 0}, V{1, 0, 0}, t));
    Camera<double> cam(640, 480);
    cam.look_at(V{0, 0, 0}, V{0, 0, 1});
    CHECK(cam.forward()[2] == 1 && cam.right()[0] == -1 && cam.up()[1] == 1 && close(cam.aspect(), 4.0 / 3));
    MirrorBxDF<double> mirror;
    auto mdir = std::get<0>(mirror.sample(V{0, 1, 0}, V{1, 1, 0}));
    CHECK(mdir[0] == -1 && mdir[1] == 1);
    // f = 1 / cos on every channel; every BxDF sample advances the stream by two draws, the mirror too
    CHECK(close(mirror(V{0, 1, 0}, V{1, 1, 0}, V{-0.6, 0.8, 0}).detach()[2], 1.25));
    random::begin_path(7, 42);
    (void)mirror.sample(V{0, 1, 0}, V{0, 1, 0});
    CHECK(close(random::uniform(), drt_rng_u31(7, 42, 2) / 2147483647.0));
    auto mmirror = std::make_shared<MirrorBxDF<double>>();
    Sphere<double> s4(V{0, -1, 3}, 0.5, mmirror);
    Scene<double> scene3{&s1, &s4, &s2};
    auto flat3 = hip::flatten(scene3);
    CHECK(flat3.materials.size() == 2 && flat3.materials[1].type == DRT_BXDF_MIRROR && flat3.materials[1].param == -1);
    CHECK(flat3.requires_grad.size() == 2);              // white and the emission: the mirror adds no parameter
    // mesh extension: brute-force triangles on the host path, flattened for the device
    std::vector<V> mv{V{-1, -1, 2}, V{1, -1, 2}, V{0, 1, 2}, V{0, 0, 4}};
    std::vector<std::array<uint32_t, 3>> mt{{{0, 1, 2}}, {{0, 1, 3}}};
    Mesh<double> mesh(mv, mt, mwhite);
    CHECK(mesh.intersect(V{0, 0, 0}, V{0, 0, 1}, t) && close(t, 2));
    CHECK(close(std::fabs(mesh.normal(V(0.))[2]), 1));
    CHECK(!mesh.intersect(V{5, 5, 0}, V{0, 0, 1}, t));
    CHECK(mesh.intersect(V{0, 0, 3}, V{0, 0, -1}, t) && close(t, 1));       // two-sided
    Scene<double> scene2{&mesh, &s2};
    auto flat2 = hip::flatten(scene2);
    CHECK(flat2.meshes.size() == 1 && flat2.meshes[0].n_triangles == 2 && flat2.meshes[0].n_vertices == 4);
    CHECK(flat2.shapes[0].type == DRT_SHAPE_MESH && flat2.shapes[0].mesh == 0 && flat2.desc().n_meshes == 1);
    CHECK(flat2.meshes[0].vertices[3 * 3 + 2] == 4 && flat2.meshes[0].indices[5] == 3);
    std::printf("ok\n");
    return 0;
}
