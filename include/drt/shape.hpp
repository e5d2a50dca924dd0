// drt/shape.hpp -- Shape<T> plugin interface with the analytic Plane and Sphere of the reference
// (include/drt/shape.hpp: interface :11-35, Plane :37-64, Sphere :66-111).  Same signatures and
// the same predicates (hit iff t > 0; plane normal used un-normalised; sphere assumes a unit
// direction and never flips its normal).
// Additive: ShapeKind / describe() expose the private fields so a scene can be flattened.
#pragma once

#include <cmath>
#include <memory>

#include "bxdf.hpp"
#include "constants.hpp"
#include "emitter.hpp"
#include "vector.hpp"

namespace drt {

enum class ShapeKind { Plane, Sphere, Mesh, Other, User };

// What a shape tells the device path about itself.  Plane / Sphere / Mesh: the library's own kinds.  User: ANY other analytic
// shape -- a subclass describes itself with a record of up to 8 values and the bodies of its intersect() / normal() as HIP source
// over that record (include/drt_hip.h: drt_shape_kind_desc says what the source may use); drt::hip::render has it compiled into the
// scene's path kernel.  Shapes of one class share one `kind_name` and one pair of sources.
struct ShapeRecord {
    ShapeKind kind = ShapeKind::Other;
    double p[4] = {0, 0, 0, 0};   // Plane: normal.xyz, offset   Sphere: center.xyz, radius   User: values 0..3 of the record
    double q[4] = {0, 0, 0, 0};   // User: values 4..7
    const char* kind_name = nullptr;        // User
    const char* intersect_src = nullptr;    // User: body of  template <typename R> bool intersect(const R* p, V3<R> o, V3<R> d, R& t)
    const char* normal_src = nullptr;       // User: body of  template <typename R> V3<R> normal(const R* p, V3<R> P)
};

template <typename T>
class Shape {
public:
    Shape(std::shared_ptr<BxDF<T>> bxdf = nullptr, std::shared_ptr<Emitter<T>> emitter = nullptr)
      : m_material(std::move(bxdf)), m_light(std::move(emitter)) { }
    virtual ~Shape() = default;

    virtual bool intersect(Vector<T, 3> orig, Vector<T, 3> dir, double& t) const = 0;
    virtual Vector<T, 3> normal(Vector<T, 3> point) const = 0;
    virtual ShapeRecord describe() const { return ShapeRecord(); }   // additive

    BxDF<T>* bxdf() { return m_material.get(); }
    Emitter<T>* emitter() { return m_light.get(); }

private:
    std::shared_ptr<BxDF<T>> m_material;
    std::shared_ptr<Emitter<T>> m_light;
};

template <typename T>
class Plane : public Shape<T> {
public:
    Plane(Vector<T, 3> normal, double offset, std::shared_ptr<BxDF<T>> bxdf = nullptr,
          std::shared_ptr<Emitter<T>> emitter = nullptr)
      : Shape<T>(bxdf, emitter), m_n(normal), m_d(offset) { }

    bool intersect(Vector<T, 3> orig, Vector<T, 3> dir, double& t) const override
    {
        const double height = real(dot(orig, m_n)) - m_d;
        t = height / real(dot(dir, -m_n));
        return t > 0;
    }

    Vector<T, 3> normal(Vector<T, 3>) const override { return m_n; }

    ShapeRecord describe() const override
    {
        ShapeRecord r;
        r.kind = ShapeKind::Plane;
        r.p[0] = real(m_n[0]); r.p[1] = real(m_n[1]); r.p[2] = real(m_n[2]); r.p[3] = m_d;
        return r;
    }

private:
    Vector<T, 3> m_n;
    double m_d;
};

template <typename T>
class Sphere : public Shape<T> {
public:
    Sphere(Vector<T, 3> center, double radius, std::shared_ptr<BxDF<T>> bxdf = nullptr,
           std::shared_ptr<Emitter<T>> emitter = nullptr)
      : Shape<T>(bxdf, emitter), m_c(center), m_r(radius) { }

    bool intersect(Vector<T, 3> orig, Vector<T, 3> dir, double& t) const override
    {
        orig -= m_c;
        const double a = 1;                       // unit direction assumed
        const double b = 2 * real(dot(orig, dir));
        const double c = real(dot(orig, orig)) - m_r * m_r;
        const double disc = b * b - 4 * a * c;
        if (disc < 0)
            return false;
        const double near = (-b - std::sqrt(disc)) / (2 * a);
        const double far = (-b + std::sqrt(disc)) / (2 * a);
        if (near > 0 && far > 0)
            t = std::min(near, far);
        else if (near > 0)
            t = near;
        else if (far > 0)
            t = far;
        else
            return false;
        return true;
    }

    Vector<T, 3> normal(Vector<T, 3> point) const override { return normalize(point - m_c); }

    ShapeRecord describe() const override
    {
        ShapeRecord r;
        r.kind = ShapeKind::Sphere;
        r.p[0] = real(m_c[0]); r.p[1] = real(m_c[1]); r.p[2] = real(m_c[2]); r.p[3] = m_r;
        return r;
    }

private:
    Vector<T, 3> m_c;
    double m_r;
};

} // namespace drt
