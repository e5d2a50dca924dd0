// drt/emitter.hpp -- light sources as plugins.  Same public surface as the reference's
// include/drt/emitter.hpp:7-25 (an abstract Emitter<T> whose emission() hands back a differentiable
// RGB handle, and AreaEmitter<T> holding one such handle).  Additive: parameter() exposes the stored
// handle by reference, so drt::hip::flatten can identify the scene parameter behind a light.
#pragma once

#include <utility>

#include "vector.hpp"

namespace drt {

// What a hit on an emissive shape adds to the path's radiance (pathtracer.hpp:113-114).
template <typename T>
class Emitter {
public:
    using Radiance = Vector<T, 3, true>;

    Emitter() = default;
    Emitter(const Emitter&) = default;
    Emitter& operator=(const Emitter&) = default;
    virtual ~Emitter() = default;

    virtual Radiance emission() const = 0;
};

// A surface that emits the same radiance in every direction; the value is a tape handle, so it can be
// a scene parameter (render.cpp:29,36) and collect a gradient.
template <typename T>
class AreaEmitter : public Emitter<T> {
    using typename Emitter<T>::Radiance;
    Radiance m_radiance;

public:
    explicit AreaEmitter(Radiance radiance) : m_radiance(std::move(radiance)) { }

    Radiance emission() const override { return m_radiance; }
    const Radiance& parameter() const { return m_radiance; }
};

} // namespace drt
