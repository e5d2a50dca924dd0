// drt/constants.hpp -- same names and values as the reference's include/drt/constants.hpp:9-11.
#pragma once

#include <limits>

namespace drt {

constexpr double pi = 3.14159265358979323846;
constexpr double inv_pi = 0.31830988618379067153;
constexpr double inf = std::numeric_limits<double>::infinity();

} // namespace drt
