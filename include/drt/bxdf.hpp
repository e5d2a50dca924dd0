// drt/bxdf.hpp -- BxDF<T> plugin interface with the Lambertian, Phong-like specular and mirror
// materials of the reference (include/drt/bxdf.hpp: interface :12-25, frame helpers :29-52,
// DiffuseBxDF :56-83, SpecularBxDF :85-124, MirrorBxDF :126-144).  Same class names, constructor
// and method signatures; the arithmetic (cosine sampling through asin(sqrt(u)), half-vector lobe,
// Gram-Schmidt frame that keeps the normal AS GIVEN) is the reference's.
// Additive: kind()/parameter()/exponent() so a scene can be flattened for the device.
// MirrorBxDF: the reference's does not compile once instantiated (:135 returns a double where a
// Vector is expected); here it is a well-formed delta reflector with the same interface.  Its
// sample() draws and discards two numbers: in this build every BxDF sample advances the stream by
// two draws, so the position of a draw is a closed form of the depth (device path, drt_hip.h).
#pragma once

#include <array>
#include <cmath>
#include <memory>
#include <tuple>

#include "constants.hpp"
#include "random.hpp"
#include "vector.hpp"

namespace drt {

enum class BxDFKind { Diffuse, Specular, Mirror, Other, User };

template <typename T>
class BxDF {
public:
    virtual ~BxDF() = default;
    virtual Vector<T, 3, true> operator()(const Vector<T, 3>& normal, const Vector<T, 3>& dir_in,
                                          const Vector<T, 3>& dir_out) const = 0;
    virtual std::tuple<Vector<T, 3>, double> sample(const Vector<T, 3>& normal,
                                                    const Vector<T, 3>& dir_in) const = 0;
    // additive (device flattening)
    virtual BxDFKind kind() const { return BxDFKind::Other; }
    virtual const Vector<T, 3, true>* parameter() const { return nullptr; }
    virtual double exponent() const { return 0; }
    // BxDFKind::User: ANY other BxDF of the form colour x scalar lobe reaches the device by naming its kind and handing over sample()
    // and operator() as ONE body of HIP source over its two draws and a record of two values -- exponent() and value1() --
    // (include/drt_hip.h: drt_bxdf_kind_desc says what the source sees); parameter() is its colour.
    virtual const char* device_kind_name() const { return nullptr; }
    virtual const char* device_sample_src() const { return nullptr; }
    virtual double value1() const { return 0; }
};

namespace internal {

// tangent frame {t, b, n}: t from e1 or e2 (whichever is less aligned with n), n kept as given
template <typename T>
inline std::array<Vector<T, 3>, 3> make_frame(const Vector<T, 3>& normal)
{
    const Vector<T, 3> ex{1., 0., 0.}, ey{0., 1., 0.};
    const bool use_x = std::abs(real(dot(ex, normal))) < std::abs(real(dot(ey, normal)));
    const Vector<T, 3>& e = use_x ? ex : ey;
    const Vector<T, 3> tangent = normalize(e - normal * dot(e, normal));
    const Vector<T, 3> bitangent = normalize(cross(normal, tangent));
    return {tangent, bitangent, normal};
}

template <typename T>
inline Vector<T, 3> angle_to_dir(double theta, double phi, const std::array<Vector<T, 3>, 3>& frame)
{
    const double x = std::cos(phi) * std::sin(theta);
    const double y = std::sin(phi) * std::sin(theta);
    const double z = std::cos(theta);
    return x * frame[0] + y * frame[1] + z * frame[2];
}

} // namespace internal

template <typename T>
class DiffuseBxDF : public BxDF<T> {
public:
    DiffuseBxDF(const Vector<T, 3, true>& color) : m_albedo(color) { }

    Vector<T, 3, true> operator()(const Vector<T, 3>&, const Vector<T, 3>&, const Vector<T, 3>&) const override
    {
        return m_albedo / pi;
    }

    std::tuple<Vector<T, 3>, double> sample(const Vector<T, 3>& normal, const Vector<T, 3>&) const override
    {
        const double theta = std::asin(std::sqrt(random::uniform()));
        const double phi = 2 * pi * random::uniform();
        const auto dir = internal::angle_to_dir(theta, phi, internal::make_frame(normal));
        return std::make_tuple(dir, std::cos(theta) / pi);
    }

    BxDFKind kind() const override { return BxDFKind::Diffuse; }
    const Vector<T, 3, true>* parameter() const override { return &m_albedo; }

private:
    Vector<T, 3, true> m_albedo;
};

template <typename T>
class SpecularBxDF : public BxDF<T> {
public:
    SpecularBxDF(const Vector<T, 3, true>& color, double exponent) : m_albedo(color), m_shininess(exponent) { }

    Vector<T, 3, true> operator()(const Vector<T, 3>& normal, const Vector<T, 3>& dir_in,
                                  const Vector<T, 3>& dir_out) const override
    {
        const Vector<T, 3> half = normalize(dir_in + dir_out);
        const double c = real(dot(normal, half));
        const double s = std::sqrt(1 - c * c);
        const double lobe = (m_shininess + 2) / (2 * pi) * std::pow(c, m_shininess) * s;
        return lobe * m_albedo;
    }

    std::tuple<Vector<T, 3>, double> sample(const Vector<T, 3>& normal, const Vector<T, 3>& dir_in) const override
    {
        const double theta = std::acos(std::sqrt(std::pow(random::uniform(), 2 / (m_shininess + 2))));
        const double phi = 2 * pi * random::uniform();
        auto half = internal::angle_to_dir(theta, phi, internal::make_frame(normal));
        if (real(dot(half, dir_in)) < 0)
            half = reflect(half, normal);
        const auto dir = reflect(dir_in, half);
        const double pdf = (m_shininess + 2) / (2 * pi) * std::pow(std::cos(theta), m_shininess + 1) * std::sin(theta);
        return std::make_tuple(dir, pdf);
    }

    BxDFKind kind() const override { return BxDFKind::Specular; }
    const Vector<T, 3, true>* parameter() const override { return &m_albedo; }
    double exponent() const override { return m_shininess; }

private:
    Vector<T, 3, true> m_albedo;
    double m_shininess;
};

template <typename T>
class MirrorBxDF : public BxDF<T> {
public:
    Vector<T, 3, true> operator()(const Vector<T, 3>& normal, const Vector<T, 3>&,
                                  const Vector<T, 3>& dir_out) const override
    {
        return Vector<T, 3, true>(T(1 / real(dot(normal, dir_out))));   // cancels the cosine
    }

    std::tuple<Vector<T, 3>, double> sample(const Vector<T, 3>& normal, const Vector<T, 3>& dir_in) const override
    {
        (void)random::uniform();
        (void)random::uniform();
        return std::make_tuple(reflect(dir_in, normal), 1.0);
    }

    BxDFKind kind() const override { return BxDFKind::Mirror; }
};

} // namespace drt
