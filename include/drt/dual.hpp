// drt/dual.hpp -- forward-mode dual numbers (validation aid, host only; the reference's
// include/drt/dual.hpp: class :9-72, operators :74-152, real/sqrt :160-172). Same public names.
#pragma once

#include <cmath>
#include <iostream>
#include <type_traits>

namespace drt {

template <typename T>
class Dual {
public:
    Dual(const T& real = T(), const T& dual = T()) : m_re(real), m_eps(dual) { }

    T& real() { return m_re; }
    const T& real() const { return m_re; }
    T& dual() { return m_eps; }
    const T& dual() const { return m_eps; }

    Dual& operator+=(const Dual& o) { m_re += o.m_re; m_eps += o.m_eps; return *this; }
    Dual& operator-=(const Dual& o) { m_re -= o.m_re; m_eps -= o.m_eps; return *this; }
    Dual& operator*=(const Dual& o)
    {
        const T eps = m_re * o.m_eps + m_eps * o.m_re;   // product rule
        m_re = m_re * o.m_re;
        m_eps = eps;
        return *this;
    }
    Dual& operator/=(const Dual& o)
    {
        const T eps = (m_eps * o.m_re - m_re * o.m_eps) / (o.m_re * o.m_re);   // quotient rule
        m_re = m_re / o.m_re;
        m_eps = eps;
        return *this;
    }

private:
    T m_re, m_eps;
};

#define DRT_DUAL_BINARY(SYM)                                                                         \
    template <typename T> inline Dual<T> operator SYM(Dual<T> a, const Dual<T>& b) { return a SYM##= b; } \
    template <typename T, typename S, typename = std::enable_if_t<std::is_convertible_v<S, T>>>      \
    inline Dual<T> operator SYM(Dual<T> a, S s) { return a SYM##= Dual<T>(T(s)); }                    \
    template <typename T, typename S, typename = std::enable_if_t<std::is_convertible_v<S, T>>>      \
    inline Dual<T> operator SYM(S s, const Dual<T>& b) { Dual<T> a{T(s)}; return a SYM##= b; }
DRT_DUAL_BINARY(+)
DRT_DUAL_BINARY(-)
DRT_DUAL_BINARY(*)
DRT_DUAL_BINARY(/)
#undef DRT_DUAL_BINARY

template <typename T>
inline std::ostream& operator<<(std::ostream& os, const Dual<T>& n)
{
    return os << n.real() << " + " << n.dual() << "e";
}

template <typename T> inline T real(const Dual<T>& n) { return n.real(); }

template <typename T>
inline Dual<T> sqrt(const Dual<T>& n)
{
    using std::sqrt;
    const T r = sqrt(n.real());
    return Dual<T>(r, n.dual() / (2 * r));
}

} // namespace drt
