// drt/pathtracer.hpp -- Scene<T> and the recursive Russian-roulette path tracer of the reference
// (include/drt/pathtracer.hpp: Scene :12-13, null-safe helpers :17-49, Pathtracer :53-136).
// Same public interface: Pathtracer<T>(absorb, min_bounces).trace(scene, orig, dir, depth).
// Differences: includes integrate.hpp itself (the reference forgets to) and
// exposes absorb()/min_bounces() so drt::hip::render can read them.
//
// This per-ray trace() is the CPU reference path of the API.  The MI355X path is the batched
// drt::hip::render in drt/hip.hpp; it does not call into this file.
#pragma once

#include <cmath>
#include <tuple>
#include <vector>

#include "bxdf.hpp"
#include "emitter.hpp"
#include "integrate.hpp"
#include "shape.hpp"
#include "vector.hpp"

namespace drt {

template <typename T>
using Scene = std::vector<Shape<T>*>;

namespace internal {

template <typename T>
std::tuple<Vector<T, 3>, double> sample_bxdf(const BxDF<T>* bxdf, Vector<T, 3> normal, Vector<T, 3> dir_in)
{
    if (!bxdf)
        return std::make_tuple(Vector<T, 3>(0), 1.0);   // no BxDF: zero direction, pdf 1
    return bxdf->sample(normal, dir_in);
}

template <typename T>
Vector<T, 3, true> eval_bxdf(const BxDF<T>* bxdf, Vector<T, 3> normal, Vector<T, 3> dir_in, Vector<T, 3> dir_out)
{
    if (!bxdf)
        return Vector<T, 3, true>(Vector<T, 3>(0));
    return (*bxdf)(normal, dir_in, dir_out);
}

template <typename T>
Vector<T, 3, true> emission(const Emitter<T>* emitter)
{
    if (!emitter)
        return Vector<T, 3, true>(Vector<T, 3>(0));
    return emitter->emission();
}

} // namespace internal

template <typename T>
class Pathtracer {
public:
    Pathtracer(double absorb, std::size_t min_bounces) : m_absorb(absorb), m_min_bounces(min_bounces) { }

    double absorb() const { return m_absorb; }
    std::size_t min_bounces() const { return m_min_bounces; }

    Vector<T, 3, true> trace(const Scene<T>& scene, Vector<T, 3> orig, Vector<T, 3> dir, std::size_t depth = 0) const
    {
        const bool roulette = depth >= m_min_bounces;
        if (roulette && random::uniform() < m_absorb)
            return Vector<T, 3, true>(Vector<T, 3>(0));
        const double survive = roulette ? (1 - m_absorb) : 1;
        Hit hit;
        if (!closest(scene, orig, dir, hit))
            return Vector<T, 3, true>(Vector<T, 3>(0));
        return shade(scene, hit, dir, depth) / survive;
    }

private:
    struct Hit {
        Vector<T, 3> point, normal;
        BxDF<T>* bxdf = nullptr;
        Emitter<T>* emitter = nullptr;
    };

    // linear scan, first shape wins ties (t >= best is skipped)
    bool closest(const Scene<T>& scene, Vector<T, 3> orig, Vector<T, 3> dir, Hit& hit) const
    {
        double best = inf;
        for (Shape<T>* shape : scene) {
            double t;
            if (!shape->intersect(orig, dir, t) || t >= best)
                continue;
            best = t;
            hit.point = orig + t * dir;
            hit.normal = shape->normal(hit.point);
            hit.bxdf = shape->bxdf();
            hit.emitter = shape->emitter();
        }
        return !std::isinf(best);
    }

    // emission + one-sample estimate of the scattering integral (biased mode of integrate)
    Vector<T, 3, true> shade(const Scene<T>& scene, const Hit& hit, Vector<T, 3> dir_in, std::size_t depth) const
    {
        const Vector<T, 3> toward_viewer = -dir_in;
        Vector<T, 3, true> scattered = integrate<T, 3>(
            [=](const Vector<T, 3>& dir_out) {
                const Vector<T, 3> next_orig = hit.point + 1e-3 * dir_out;
                Vector<T, 3, true> f = internal::eval_bxdf(hit.bxdf, hit.normal, toward_viewer, dir_out);
                Vector<T, 3, true> incoming = trace(scene, next_orig, dir_out, depth + 1);
                const double cosine = real(dot(hit.normal, dir_out));
                return f * incoming * cosine;
            },
            [=]() { return internal::sample_bxdf(hit.bxdf, hit.normal, toward_viewer); },
            1, false);
        return internal::emission(hit.emitter) + scattered;
    }

    double m_absorb;
    std::size_t m_min_bounces;
};

} // namespace drt
