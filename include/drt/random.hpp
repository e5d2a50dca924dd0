// drt/random.hpp -- drt::random::uniform(), the reference's single RNG entry point
// (include/drt/random.hpp:7-10: double(rand()) / RAND_MAX, global libc state).
//
// Default behaviour is unchanged (libc stream).  Additive: begin_path(seed, path) switches the
// calling thread to the per-path counter RNG of include/drt_hip.h, the stream the device kernels
// and the oracle draw from, so the CPU drop-in path can replay a device render sample by sample;
// use_libc() switches back.
#pragma once

#include <cstdint>
#include <cstdlib>

#include "../drt_hip.h"

namespace drt { namespace random {

struct Stream {
    bool keyed = false;
    drt_rng_key path_key = {0, 0};
    uint32_t draw = 0;
};

inline Stream& stream()
{
    static thread_local Stream s;
    return s;
}

inline void begin_path(uint32_t seed, uint64_t path)
{
    Stream& s = stream();
    s.keyed = true;
    s.path_key = drt_rng_path_key(seed, path);
    s.draw = 0;
}

inline void use_libc() { stream().keyed = false; }

inline double uniform()
{
    Stream& s = stream();
    if (s.keyed)
        return double(drt_rng_draw(s.path_key, s.draw++)) / 2147483647.0;
    return double(rand()) / RAND_MAX;
}

} } // namespace drt::random
