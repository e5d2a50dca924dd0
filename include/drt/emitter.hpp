// drt/emitter.hpp -- Emitter<T> plugin interface and AreaEmitter (reference:
// include/drt/emitter.hpp:7-25). Additive: parameter() exposes the emission handle so a scene can
// be flattened for the device.
#pragma once

#include "vector.hpp"

namespace drt {

template <typename T>
class Emitter {
public:
    virtual ~Emitter() = default;
    virtual Vector<T, 3, true> emission() const = 0;
};

template <typename T>
class AreaEmitter : public Emitter<T> {
public:
    AreaEmitter(Vector<T, 3, true> emission) : m_radiance(emission) { }
    Vector<T, 3, true> emission() const override { return m_radiance; }
    const Vector<T, 3, true>& parameter() const { return m_radiance; }

private:
    Vector<T, 3, true> m_radiance;
};

} // namespace drt
