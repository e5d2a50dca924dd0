// drt/camera.hpp -- pinhole camera with jittered pixel samples (reference:
// include/drt/camera.hpp:10-70; same constructor defaults, look_at and sample arithmetic).
// Additive: read accessors for the fields the reference keeps private (:62-69).
#pragma once

#include <cmath>
#include <cstddef>
#include <tuple>

#include "random.hpp"
#include "vector.hpp"

namespace drt {

template <typename T>
class Camera {
    using V = Vector<T, 3>;

public:
    Camera(std::size_t width, std::size_t height, double vfov = 1.3963, V eye = V(0),
           V forward = V{0, 0, -1}, V right = V{1, 0, 0}, V up = V{0, 1, 0})
      : m_w(width), m_h(height), m_vfov(vfov), m_eye(eye), m_fwd(forward), m_right(right), m_up(up) { }

    void look_at(V eye, V at, V up = V{0, 1, 0})
    {
        m_eye = eye;
        m_fwd = normalize(at - eye);
        m_right = normalize(cross(m_fwd, up));
        m_up = cross(m_right, m_fwd);
    }

    std::size_t width() const { return m_w; }
    std::size_t height() const { return m_h; }
    V eye() const { return m_eye; }
    double aspect() const { return double(m_w) / m_h; }
    // additive accessors
    double vfov() const { return m_vfov; }
    V forward() const { return m_fwd; }
    V right() const { return m_right; }
    V up() const { return m_up; }

    // direction through a uniformly jittered point of pixel (x, y); pdf 1
    std::tuple<V, double> sample(std::size_t x, std::size_t y) const
    {
        const double s = (x + random::uniform()) / m_w;
        const double t = (y + random::uniform()) / m_h;
        const double half = std::tan(m_vfov / 2.);
        V dir = m_fwd;
        dir += (2. * s - 1.) * aspect() * half * m_right;
        dir += (2. * t - 1.) * half * -m_up;
        return std::make_tuple(normalize(dir), 1.0);
    }

private:
    std::size_t m_w, m_h;
    double m_vfov;
    V m_eye, m_fwd, m_right, m_up;
};

} // namespace drt
