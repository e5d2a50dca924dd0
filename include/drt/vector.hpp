// drt/vector.hpp -- fixed-size vectors and the reverse-mode tape handle.
//
// Source-compatible with the public surface of the reference's include/drt/vector.hpp
// (plain Vector<T,N> :18-118, tape handle Vector<T,N,true> :217-318, free operators :320-370,
// detach/requires_grad/backward :372-416, recording operators :488-557, geometry :573-606),
// written from scratch with a different core: ONE node type with an op tag instead of a class
// per backward functor, and an ITERATIVE depth-first backward pass (explicit stack, left operand
// first -- the visiting order of the reference's recursion, so gradients accumulate in the same
// order) that cannot overflow the call stack on long paths.
//
// Additions over the reference (all additive): Vector<T,N,true>::id() (node identity, used to
// deduplicate scene parameters when a scene is flattened for the device), zero-initialised
// gradients (the reference leaves VariableNode::m_grad indeterminate, :191).
#pragma once

#include <cstddef>
#include <algorithm>
#include <array>
#include <cmath>
#include <functional>
#include <initializer_list>
#include <iostream>
#include <memory>
#include <stdexcept>
#include <type_traits>
#include <typeinfo>
#include <utility>
#include <vector>

namespace drt {

template <typename T, std::size_t N, bool Autograd = false>
class Vector;

// value part of a scalar: identity for built-in types, the real part of a drt::Dual (dual.hpp).
// Geometry (hit distances, cosines, pdfs) is never differentiated (README.md:147-151), so the
// plugins take real() of such quantities; this is what lets T = Dual<double> compile end to end.
inline double real(double x) { return x; }
inline float real(float x) { return x; }

// ---- plain value vector ------------------------------------------------------------------------
template <typename T, std::size_t N>
class Vector<T, N> {
public:
    using iterator = typename std::array<T, N>::iterator;
    using const_iterator = typename std::array<T, N>::const_iterator;

    Vector() = default;
    explicit Vector(T value) { m_e.fill(value); }
    Vector(std::initializer_list<T> init)
    {
        if (init.size() != N)
            throw std::runtime_error("incorrect number of initializers for `Vector`");
        std::size_t i = 0;
        for (const T& x : init)
            m_e[i++] = x;
    }

    T& operator[](std::size_t pos) { return m_e[pos]; }
    const T& operator[](std::size_t pos) const { return m_e[pos]; }
    iterator begin() { return m_e.begin(); }
    const_iterator begin() const { return m_e.begin(); }
    iterator end() { return m_e.end(); }
    const_iterator end() const { return m_e.end(); }
    constexpr std::size_t size() const { return N; }

    Vector& operator+=(const Vector& r) { for (std::size_t i = 0; i < N; ++i) m_e[i] = m_e[i] + r.m_e[i]; return *this; }
    Vector& operator-=(const Vector& r) { for (std::size_t i = 0; i < N; ++i) m_e[i] = m_e[i] - r.m_e[i]; return *this; }
    Vector& operator*=(const Vector& r) { for (std::size_t i = 0; i < N; ++i) m_e[i] = m_e[i] * r.m_e[i]; return *this; }
    Vector& operator/=(const Vector& r) { for (std::size_t i = 0; i < N; ++i) m_e[i] = m_e[i] / r.m_e[i]; return *this; }
    Vector& operator*=(T s) { for (std::size_t i = 0; i < N; ++i) m_e[i] = m_e[i] * s; return *this; }
    Vector& operator/=(T s) { for (std::size_t i = 0; i < N; ++i) m_e[i] = m_e[i] / s; return *this; }

private:
    std::array<T, N> m_e;
};

// ---- tape ----------------------------------------------------------------------------------------
namespace tape {

enum class Op : unsigned char { Constant, Variable, Add, Sub, Mul, Div, Scale, InvScale, Custom };

template <typename T, std::size_t N>
struct Node {
    Vector<T, N> value;
    Op op = Op::Constant;
    std::shared_ptr<Node> lhs, rhs;     // operands (rhs unused by Scale / InvScale)
    T scalar{};                         // Scale: s * v   InvScale: v / s
    Vector<T, N> grad = Vector<T, N>(T());   // Variable only; zero-initialised
    std::function<void(const Vector<T, N>&)> custom;

    bool needs_grad() const { return op != Op::Constant; }
};

// Depth-first, left operand first: the order in which the reference's recursive functors reach
// the variables (vector.hpp:420-484), so sums round identically.
template <typename T, std::size_t N>
inline void backpropagate(const Node<T, N>* root, const Vector<T, N>& seed)
{
    std::vector<std::pair<const Node<T, N>*, Vector<T, N>>> todo;
    todo.emplace_back(root, seed);
    while (!todo.empty()) {
        const Node<T, N>* n = todo.back().first;
        const Vector<T, N> g = todo.back().second;
        todo.pop_back();
        switch (n->op) {
        case Op::Constant:
            break;
        case Op::Variable:
            const_cast<Node<T, N>*>(n)->grad += g;                      // the accumulator
            break;
        case Op::Add:
            todo.emplace_back(n->rhs.get(), g);
            todo.emplace_back(n->lhs.get(), g);
            break;
        case Op::Sub: {
            Vector<T, N> ng = g;
            ng *= T(-1);
            todo.emplace_back(n->rhs.get(), ng);
            todo.emplace_back(n->lhs.get(), g);
            break;
        }
        case Op::Mul: {
            Vector<T, N> gl = n->rhs->value; gl *= g;
            Vector<T, N> gr = n->lhs->value; gr *= g;
            todo.emplace_back(n->rhs.get(), gr);
            todo.emplace_back(n->lhs.get(), gl);
            break;
        }
        case Op::Div: {
            Vector<T, N> gl = g; gl /= n->rhs->value;
            Vector<T, N> gr = n->lhs->value; gr *= T(-1); gr *= g;
            Vector<T, N> den = n->rhs->value; den *= n->rhs->value;
            gr /= den;
            todo.emplace_back(n->rhs.get(), gr);
            todo.emplace_back(n->lhs.get(), gl);
            break;
        }
        case Op::Scale: {
            Vector<T, N> gv = g; gv *= n->scalar;
            todo.emplace_back(n->lhs.get(), gv);
            break;
        }
        case Op::InvScale: {
            Vector<T, N> gv = g; gv /= n->scalar;
            todo.emplace_back(n->lhs.get(), gv);
            break;
        }
        case Op::Custom:
            n->custom(g);
            break;
        }
    }
}

} // namespace tape

// ---- tape handle ---------------------------------------------------------------------------------
template <typename T, std::size_t N>
class Vector<T, N, true> {
    using NodeT = tape::Node<T, N>;

public:
    explicit Vector(T value, bool requires_grad = false) : Vector(Vector<T, N>(value), requires_grad) { }
    Vector(std::initializer_list<T> init, bool requires_grad = false) : Vector(Vector<T, N>(init), requires_grad) { }
    Vector(const Vector<T, N>& v, bool requires_grad = false) : m_node(std::make_shared<NodeT>())
    {
        m_node->value = v;
        m_node->op = requires_grad ? tape::Op::Variable : tape::Op::Constant;
    }
    // value + user-supplied backward callable (README.md:72-77, integrate.hpp:49-51)
    template <typename Backward>
    Vector(const Vector<T, N>& v, const Backward& backward) : m_node(std::make_shared<NodeT>())
    {
        m_node->value = v;
        m_node->op = tape::Op::Custom;
        m_node->custom = backward;
    }

    T& operator[](std::size_t pos) { return m_node->value[pos]; }
    const T& operator[](std::size_t pos) const { return m_node->value[pos]; }
    constexpr std::size_t size() const { return N; }

    Vector<T, N>& detach() { return m_node->value; }
    const Vector<T, N>& detach() const { return m_node->value; }

    Vector<T, N>& grad()
    {
        if (m_node->op != tape::Op::Variable)
            throw std::runtime_error("Vector has no gradient (not a variable)");
        return m_node->grad;
    }
    const Vector<T, N>& grad() const
    {
        if (m_node->op != tape::Op::Variable)
            throw std::runtime_error("Vector has no gradient (not a variable)");
        return m_node->grad;
    }

    bool requires_grad() const { return m_node->needs_grad(); }
    void backward(const Vector<T, N>& grad) const { tape::backpropagate(m_node.get(), grad); }

    // identity of the underlying node: handles copied from one another share it
    const void* id() const { return m_node.get(); }

    Vector& operator+=(const Vector& rhs) { return *this = *this + rhs; }
    Vector& operator-=(const Vector& rhs) { return *this = *this - rhs; }
    Vector& operator*=(const Vector& rhs) { return *this = *this * rhs; }
    Vector& operator/=(const Vector& rhs) { return *this = *this / rhs; }
    Vector& operator*=(T s) { return *this = *this * s; }
    Vector& operator/=(T s) { return *this = *this / s; }

    // internal: build a recorded node
    static Vector record(const Vector<T, N>& value, tape::Op op, const Vector* a, const Vector* b, T s = T())
    {
        Vector r(value, false);
        r.m_node->op = op;
        if (a) r.m_node->lhs = a->m_node;
        if (b) r.m_node->rhs = b->m_node;
        r.m_node->scalar = s;
        return r;
    }

private:
    std::shared_ptr<NodeT> m_node;
};

// ---- helpers that treat both kinds uniformly -----------------------------------------------------
template <typename T, std::size_t N> inline Vector<T, N>& detach(Vector<T, N>& v) { return v; }
template <typename T, std::size_t N> inline const Vector<T, N>& detach(const Vector<T, N>& v) { return v; }
template <typename T, std::size_t N> inline Vector<T, N>& detach(Vector<T, N, true>& v) { return v.detach(); }
template <typename T, std::size_t N> inline const Vector<T, N>& detach(const Vector<T, N, true>& v) { return v.detach(); }
template <typename T, std::size_t N> inline constexpr bool requires_grad(const Vector<T, N>&) { return false; }
template <typename T, std::size_t N> inline bool requires_grad(const Vector<T, N, true>& v) { return v.requires_grad(); }
template <typename T, std::size_t N> inline void backward(Vector<T, N>&, const Vector<T, N>&) { }
template <typename T, std::size_t N> inline void backward(Vector<T, N, true>& v, const Vector<T, N>& g) { v.backward(g); }

namespace tape {
template <typename T, std::size_t N> inline Vector<T, N, true> lift(const Vector<T, N>& v) { return Vector<T, N, true>(v, false); }
template <typename T, std::size_t N> inline const Vector<T, N, true>& lift(const Vector<T, N, true>& v) { return v; }
} // namespace tape

// ---- plain operators -------------------------------------------------------------------------------
template <typename T, std::size_t N> inline Vector<T, N> operator+(Vector<T, N> a, const Vector<T, N>& b) { return a += b; }
template <typename T, std::size_t N> inline Vector<T, N> operator-(Vector<T, N> a, const Vector<T, N>& b) { return a -= b; }
template <typename T, std::size_t N> inline Vector<T, N> operator*(Vector<T, N> a, const Vector<T, N>& b) { return a *= b; }
template <typename T, std::size_t N> inline Vector<T, N> operator/(Vector<T, N> a, const Vector<T, N>& b) { return a /= b; }
template <typename T, std::size_t N, typename S, typename = std::enable_if_t<std::is_convertible_v<S, T>>>
inline Vector<T, N> operator*(Vector<T, N> v, S s) { return v *= s; }
template <typename T, std::size_t N, typename S, typename = std::enable_if_t<std::is_convertible_v<S, T>>>
inline Vector<T, N> operator*(S s, Vector<T, N> v) { return v *= s; }
template <typename T, std::size_t N, typename S, typename = std::enable_if_t<std::is_convertible_v<S, T>>>
inline Vector<T, N> operator/(Vector<T, N> v, S s) { return v /= s; }

// unary minus is "-1 * v" for both kinds (so it is recorded as a Scale on the tape)
template <typename T, std::size_t N, bool Ag, typename = std::enable_if_t<std::is_convertible_v<int, T>>>
inline Vector<T, N, Ag> operator-(const Vector<T, N, Ag>& v) { return -1 * v; }

// ---- recording operators (at least one tape operand) ---------------------------------------------
#define DRT_RECORDING_BINARY(SYM, OPTAG)                                                          \
    template <typename T, std::size_t N, bool A1, bool A2, typename = std::enable_if_t<A1 || A2>>  \
    inline Vector<T, N, true> operator SYM(const Vector<T, N, A1>& a, const Vector<T, N, A2>& b)   \
    {                                                                                              \
        Vector<T, N> value = detach(a) SYM detach(b);                                              \
        if (!requires_grad(a) && !requires_grad(b))                                                \
            return Vector<T, N, true>(value, false);                                               \
        const auto& la = tape::lift(a);                                                            \
        const auto& lb = tape::lift(b);                                                            \
        return Vector<T, N, true>::record(value, tape::Op::OPTAG, &la, &lb);                       \
    }
DRT_RECORDING_BINARY(+, Add)
DRT_RECORDING_BINARY(-, Sub)
DRT_RECORDING_BINARY(*, Mul)
DRT_RECORDING_BINARY(/, Div)
#undef DRT_RECORDING_BINARY

template <typename T, std::size_t N, typename S, typename = std::enable_if_t<std::is_convertible_v<S, T>>>
inline Vector<T, N, true> operator*(S s, Vector<T, N, true> v)
{
    Vector<T, N> value = s * v.detach();
    if (!v.requires_grad())
        return Vector<T, N, true>(value, false);
    return Vector<T, N, true>::record(value, tape::Op::Scale, &v, nullptr, T(s));
}
template <typename T, std::size_t N, typename S, typename = std::enable_if_t<std::is_convertible_v<S, T>>>
inline Vector<T, N, true> operator*(Vector<T, N, true> v, S s) { return s * v; }
template <typename T, std::size_t N, typename S, typename = std::enable_if_t<std::is_convertible_v<S, T>>>
inline Vector<T, N, true> operator/(Vector<T, N, true> v, S s)
{
    Vector<T, N> value = v.detach() / s;
    if (!v.requires_grad())
        return Vector<T, N, true>(value, false);
    return Vector<T, N, true>::record(value, tape::Op::InvScale, &v, nullptr, T(s));
}

// ---- printing and geometry -----------------------------------------------------------------------
template <typename T, std::size_t N, bool Ag>
inline std::ostream& operator<<(std::ostream& os, const Vector<T, N, Ag>& v)
{
    os << "Vector<" << typeid(T).name() << ", " << N << (Ag ? ", true" : "") << ">{";
    for (std::size_t i = 0; i < N; ++i)
        os << (i ? ", " : "") << v[i];
    return os << "}";
}

template <typename T, std::size_t N>
inline T dot(const Vector<T, N>& a, const Vector<T, N>& b)
{
    T sum = T();
    for (std::size_t i = 0; i < N; ++i)
        sum = sum + a[i] * b[i];
    return sum;
}

template <typename T, std::size_t N> inline T norm(const Vector<T, N>& v) { using std::sqrt; return sqrt(dot(v, v)); }
template <typename T, std::size_t N> inline Vector<T, N> normalize(const Vector<T, N>& v) { return v / norm(v); }

template <typename T>
inline Vector<T, 3> cross(const Vector<T, 3>& a, const Vector<T, 3>& b)
{
    return Vector<T, 3>{a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
}

// mirror v about n: -v + 2 (n.v) n
template <typename T, std::size_t N>
inline Vector<T, N> reflect(const Vector<T, N>& v, const Vector<T, N>& n)
{
    return -v + 2 * dot(n, v) * n;
}

} // namespace drt
