// drt/mesh.hpp -- EXTENSION (the reference has no triangles): a triangle mesh as a plugin of the
// Shape<T> interface (reference: include/drt/shape.hpp:11-35).  On the host path it is a linear
// scan over its triangles with the semantics pinned in oracle/ref_harness.cpp (two-sided
// Moller-Trumbore, hit iff t > 0, earlier triangle wins exact ties, geometric normal never
// flipped); drt::hip::render flattens it into a drt_mesh_desc and the device walks a BVH instead.
#pragma once

#include <array>
#include <cstdint>
#include <memory>
#include <vector>

#include "shape.hpp"

namespace drt {

template <typename T>
class Mesh : public Shape<T> {
public:
    Mesh(std::vector<Vector<T, 3>> vertices, std::vector<std::array<uint32_t, 3>> triangles,
         std::shared_ptr<BxDF<T>> bxdf = nullptr, std::shared_ptr<Emitter<T>> emitter = nullptr)
      : Shape<T>(bxdf, emitter), m_vertices(std::move(vertices)), m_triangles(std::move(triangles))
    {
        for (const auto& t : m_triangles) {
            Corner c;
            c.v0 = m_vertices.at(t[0]);
            c.e1 = m_vertices.at(t[1]) - c.v0;
            c.e2 = m_vertices.at(t[2]) - c.v0;
            c.n = normalize(cross(c.e1, c.e2));
            m_corners.push_back(c);
        }
    }

    bool intersect(Vector<T, 3> orig, Vector<T, 3> dir, double& t) const override
    {
        double best = inf;
        for (std::size_t k = 0; k < m_corners.size(); ++k) {
            const Corner& c = m_corners[k];
            const Vector<T, 3> pvec = cross(dir, c.e2);
            const double det = real(dot(c.e1, pvec));
            if (det == 0)
                continue;
            const double inv = 1 / det;
            const Vector<T, 3> tvec = orig - c.v0;
            const double u = real(dot(tvec, pvec)) * inv;
            if (u < 0 || u > 1)
                continue;
            const Vector<T, 3> qvec = cross(tvec, c.e1);
            const double v = real(dot(dir, qvec)) * inv;
            if (v < 0 || u + v > 1)
                continue;
            const double tk = real(dot(c.e2, qvec)) * inv;
            if (!(tk > 0) || tk >= best)
                continue;
            best = tk;
            m_last = k;
        }
        t = best;
        return !std::isinf(best);
    }

    // normal of the triangle found by the last successful intersect() (the reference's raycast
    // asks for it right after, pathtracer.hpp:83-84); like the reference, not re-entrant
    Vector<T, 3> normal(Vector<T, 3>) const override { return m_corners.at(m_last).n; }

    ShapeRecord describe() const override
    {
        ShapeRecord r;
        r.kind = ShapeKind::Mesh;
        return r;
    }

    const std::vector<Vector<T, 3>>& vertices() const { return m_vertices; }
    const std::vector<std::array<uint32_t, 3>>& triangles() const { return m_triangles; }

private:
    struct Corner { Vector<T, 3> v0, e1, e2, n; };
    std::vector<Vector<T, 3>> m_vertices;
    std::vector<std::array<uint32_t, 3>> m_triangles;
    std::vector<Corner> m_corners;
    mutable std::size_t m_last = 0;
};

} // namespace drt
