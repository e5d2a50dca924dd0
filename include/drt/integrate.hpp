// drt/integrate.hpp -- the Monte-Carlo integration operator drt::integrate<T,N>(forward, sampler,
// n_samples, unbiased) of the reference (include/drt/integrate.hpp:56-66).
//   biased   (:26-37): sum forward(s) / pdf on the tape; backward reuses the forward samples.
//   unbiased (:39-52, README.md:104-136): forward values are detached; backward draws FRESH
//            samples and back-propagates grad / pdf through a new forward evaluation (:11-24).
#pragma once

#include <cstddef>
#include <tuple>
#include <type_traits>

#include "vector.hpp"

namespace drt {

template <typename T, std::size_t N, typename Forward, typename Sampler>
inline Vector<T, N, true> integrate(const Forward& forward, const Sampler& sampler,
                                    std::size_t n_samples, bool unbiased = false)
{
    if (!unbiased) {
        Vector<T, N, true> sum(T(0));
        for (std::size_t i = 0; i < n_samples; ++i) {
            auto drawn = sampler();
            sum += forward(std::get<0>(drawn)) / std::get<1>(drawn);
        }
        return sum;
    }
    Vector<T, N> value(T(0));
    for (std::size_t i = 0; i < n_samples; ++i) {
        auto drawn = sampler();
        value += forward(std::get<0>(drawn)).detach() / std::get<1>(drawn);
    }
    typename std::decay<Forward>::type fwd = forward;
    typename std::decay<Sampler>::type smp = sampler;
    return Vector<T, N, true>(value, [fwd, smp, n_samples](const Vector<T, N>& grad) {
        for (std::size_t i = 0; i < n_samples; ++i) {
            auto drawn = smp();
            fwd(std::get<0>(drawn)).backward(grad / std::get<1>(drawn));
        }
    });
}

} // namespace drt
