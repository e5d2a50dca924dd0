// drt_tuning.h -- every environment variable libdrt_hip.so reads, in ONE table: name, default, meaning.  Read once per
// process (the first context).  None of them changes a result beyond f32 summation order; they size grids and batches, pick
// between equivalent routes for measurement, or switch debugging output on.  What a caller is meant to choose lives in the ABI
// (drt_render_params, the DRT_RENDER_* flags, drt_hip_set_specialisation), not here.  INTEGRATION.md section 2 lists this
// table; tests/test_abi.py checks that the two agree.
#pragma once

#include <cstdlib>
#include <cstring>

namespace {

struct Tuning {
    // ---- k_path
    int jit = 1;                   // DRT_HIP_JIT              -1 | 0 | 1 | force: default of drt_hip_set_specialisation for new contexts
    bool jit_verbose = false;      // DRT_HIP_JIT_VERBOSE      print a failed run-time compile to stderr
    bool builtin_program = true;   // DRT_HIP_BUILTIN_PROGRAM  0: the reference's own scene is specialised at run time like any other (test of hiprtc against the library's own build)
    bool overlap_frames = true;    // DRT_HIP_OVERLAP_FRAMES   0: the path kernels of consecutive frames never overlap (what DRT_RENDER_SERIAL asks per frame)
    int path_spr = 0;              // DRT_HIP_PATH_SPR         samples per wave range of k_path; 0 = automatic (~112 waves per CU)
    int path_regen = -1;           // DRT_HIP_PATH_REGEN       1 / 0: force the regenerating / lockstep form of k_path; -1 = the cheaper by the library's estimate
    int path_regen_min = 8;        // DRT_HIP_PATH_REGEN_MIN   idle lanes it takes for the regenerating form to run the camera code
    bool path_general = true;      // DRT_HIP_PATH_GENERAL     0: gradients of more than 8 parameters take the queue wavefront (tape + K6) instead of the one-launch kernels' general form
    int gen_above = 4;             // DRT_HIP_GEN_ABOVE        scenes with more parameters than this take the general form (history + tables); up to 8 the register / column form exists too
    int gen_hist_lds = -1;         // DRT_HIP_GEN_HIST_LDS     history words (four vertices each) per thread the general form keeps in LDS; -1 = 4 in the lockstep kernel, 0 in the regenerating ones
    int gen_copies_log2 = -1;      // DRT_HIP_GEN_COPIES_LOG2  log2 of the copies a wave keeps of every row of its gradient table (general form); -1 = as many as fit, at most 16
    // ---- the queue wavefront
    long long batch_paths = 0;     // DRT_HIP_BATCH_PATHS      paths per batch; 0 = drt_render_params.batch_paths, else sized by the device's memory
    int region_size = 0;           // DRT_HIP_REGION_SIZE      slots per queue region (a wave's share of a queue); 0 = 256
    int shade_bounces = 0;         // DRT_HIP_SHADE_BOUNCES    overrides drt_render_params.bounces_per_launch
    // ---- triangle meshes
    int mesh_blocks_per_cu = 0;    // DRT_HIP_MESH_BLOCKS_PER_CU   blocks of the BVH walk's persistent grid per CU; 0 = what the occupancy query says (5)
    int tail_bounces = 0;          // DRT_HIP_TAIL_BOUNCES         mesh scenes: 2 = a shade launch keeps the rays that miss the mesh bounds in registers through one more vertex, 1 = every ray is queued, 0 = 2 for the unbiased operator's rounds, else 1
    int shade_list_group = 4;      // DRT_HIP_SHADE_LIST_GROUP     region lists the walk pulls at a time
    int bvh_refill = -1;           // DRT_HIP_BVH_REFILL           idle lanes before the walk's waves refill; -1 = DRT_BVH_REFILL (16)
    int bvh_descend_min = -1;      // DRT_HIP_BVH_DESCEND_MIN      lanes that keep the interior-node loop going; -1 = DRT_BVH_DESCEND_MIN (32)
    long long mesh_path_max = 1 << 20;   // DRT_HIP_MESH_PATH_MAX   mesh scenes: frames of at most this many camera samples take the one-launch k_path_mesh (0: never), larger ones the queue wavefront
    int mesh_shade_min = 32;       // DRT_HIP_MESH_SHADE_MIN       k_path_mesh: lanes with a final hit (or without a path) it takes to run the shade step
    // ---- host buffers, groups
    int copy_blocks = 64;          // DRT_HIP_COPY_BLOCKS      blocks of the launch that carries an asynchronous frame to the pinned block
    bool sync_zero_copy = true;    // DRT_HIP_SYNC_ZERO_COPY   0: a synchronous host-buffer render copies its image with hipMemcpyAsync after the finishing kernel instead of having it stored into the pinned block
    int sync_spin_us = 2000;       // DRT_HIP_SYNC_SPIN_US     how long a synchronous host-buffer render polls its completion word before it falls back to hipStreamSynchronize (0: never polls)
    bool async_copy_inline = false;// DRT_HIP_ASYNC_COPY       inline: asynchronous frames on ONE stream, the finishing kernels store into the pinned block
    bool group_threads = true;     // DRT_HIP_GROUP_THREADS    0: a group context's members enqueue in turn, not from a thread each
    // ---- debugging
    long long dump_path = -1;      // DRT_HIP_DUMP_PATH        print the tape of this path of the last batch (queue route) and, under the unbiased operator, its chain round by round; -2: every path's chain
};

inline const Tuning& tuning()
{
    static const Tuning t = [] {
        Tuning v;
        auto num = [](const char* name, long long dflt) { const char* e = getenv(name); return e ? atoll(e) : dflt; };
        auto off = [](const char* name) { const char* e = getenv(name); return e && atoi(e) == 0; };
        if (const char* e = getenv("DRT_HIP_JIT"))
            v.jit = !strcmp(e, "force") ? 2 : (atoi(e) > 0 ? 1 : (atoi(e) < 0 ? -1 : 0));
        v.jit_verbose = getenv("DRT_HIP_JIT_VERBOSE") != nullptr;
        v.builtin_program = !off("DRT_HIP_BUILTIN_PROGRAM");
        v.overlap_frames = !off("DRT_HIP_OVERLAP_FRAMES");
        v.path_spr = (int)num("DRT_HIP_PATH_SPR", 0);
        v.path_regen = (int)num("DRT_HIP_PATH_REGEN", -1);
        v.path_regen_min = (int)num("DRT_HIP_PATH_REGEN_MIN", 8);
        v.path_general = !off("DRT_HIP_PATH_GENERAL");
        v.gen_copies_log2 = (int)num("DRT_HIP_GEN_COPIES_LOG2", -1);
        v.gen_above = (int)num("DRT_HIP_GEN_ABOVE", 4);
        if (v.gen_above < 0) v.gen_above = 0;
        if (v.gen_above > DRT_FAST_PARAMS) v.gen_above = DRT_FAST_PARAMS;
        v.gen_hist_lds = (int)num("DRT_HIP_GEN_HIST_LDS", -1);
        if (v.gen_hist_lds > 16) v.gen_hist_lds = 16;
        if (v.gen_copies_log2 > 4) v.gen_copies_log2 = 4;
        v.batch_paths = num("DRT_HIP_BATCH_PATHS", 0);
        v.region_size = (int)num("DRT_HIP_REGION_SIZE", 0);
        v.shade_bounces = (int)num("DRT_HIP_SHADE_BOUNCES", 0);
        v.mesh_blocks_per_cu = (int)num("DRT_HIP_MESH_BLOCKS_PER_CU", 0);
        v.shade_list_group = (int)num("DRT_HIP_SHADE_LIST_GROUP", 4);
        v.tail_bounces = (int)num("DRT_HIP_TAIL_BOUNCES", 0);
        if (v.tail_bounces < 0 || v.tail_bounces > 2) v.tail_bounces = 0;
        v.bvh_refill = (int)num("DRT_HIP_BVH_REFILL", -1);
        v.bvh_descend_min = (int)num("DRT_HIP_BVH_DESCEND_MIN", -1);
        v.mesh_path_max = num("DRT_HIP_MESH_PATH_MAX", 1 << 20);
        v.mesh_shade_min = (int)num("DRT_HIP_MESH_SHADE_MIN", 32);
        v.copy_blocks = (int)num("DRT_HIP_COPY_BLOCKS", 64);
        v.sync_zero_copy = !off("DRT_HIP_SYNC_ZERO_COPY");
        v.sync_spin_us = (int)num("DRT_HIP_SYNC_SPIN_US", 2000);
        v.async_copy_inline = getenv("DRT_HIP_ASYNC_COPY") && !strcmp(getenv("DRT_HIP_ASYNC_COPY"), "inline");
        v.group_threads = !off("DRT_HIP_GROUP_THREADS");
        v.dump_path = num("DRT_HIP_DUMP_PATH", -1);
        if (v.path_regen_min < 1) v.path_regen_min = 1;
        if (v.mesh_shade_min < 1) v.mesh_shade_min = 1;
        if (v.mesh_shade_min > 64) v.mesh_shade_min = 64;
        if (v.shade_list_group < 1) v.shade_list_group = 1;
        if (v.copy_blocks < 1) v.copy_blocks = 1;
        return v;
    }();
    return t;
}

// DRT_HIP_TAIL_BOUNCES as it stands NOW: the one knob read at every render (the two settings differ by a few per cent, less than
// one process differs from the next on a box that warms up: tools/tail_check.py alternates them inside one process)
inline int tail_bounces_now()
{
    if (const char* e = getenv("DRT_HIP_TAIL_BOUNCES"))
        return ((e[0] == '1' || e[0] == '2') && !e[1]) ? e[0] - '0' : 0;
    return tuning().tail_bounces;
}

} // namespace
