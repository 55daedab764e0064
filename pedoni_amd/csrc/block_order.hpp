// block_order.hpp -- which tile (a run of consecutive agents of the sorted order) a hardware workgroup works
// on.  Plain integer arithmetic, free of HIP types, so that tests/cpp/test_block_order.cpp can check on the CPU
// what the kernels rely on: every mapping here is a BIJECTION of [0, n_blocks) for any grid size.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define PEDONI_HOSTDEV __attribute__((host)) __attribute__((device)) inline __attribute__((always_inline))
#else
#define PEDONI_HOSTDEV inline
#endif

namespace pedoni {

// Workgroups are dealt round-robin to the 8 XCDs, each with a private L2.  Agents are sorted
// by cell, so neighbouring workgroups share most of their candidate lines: map the hardware
// block id so that every XCD works through ONE contiguous eighth of the agents (its blocks
// b, b+8, b+16 ... become logical blocks k, k+1, k+2 ...) and neighbour rows are served by
// the same L2.  Bijective for any grid size (cdna_hip_programming.md T1); placement is a
// speed matter only.
PEDONI_HOSTDEV uint32_t xcd_contiguous_block(uint32_t b, uint32_t n_blocks)
{
    const uint32_t q = n_blocks / 8u, r = n_blocks % 8u, xcd = b % 8u;
    return (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + b / 8u;
}

// Edge-first order (force_kernel_queue_edge_first): the e_lo workgroups the hardware starts first take tiles
// [0, e_lo), the next e_hi take tiles [t_hi, t_hi + e_hi), the others the tiles left, in order (XCD-contiguous
// among themselves when `remap`).  Needs e_lo <= t_hi and t_hi + e_hi <= n_blocks (the host clamps).
PEDONI_HOSTDEV uint32_t edge_first_tile(uint32_t b, uint32_t n_blocks, uint32_t e_lo, uint32_t e_hi, uint32_t t_hi,
                                        int32_t remap)
{
    const uint32_t n_edge = e_lo + e_hi;
    if (b < e_lo) return b;
    if (b < n_edge) return t_hi + (b - e_lo);
    // the i-th tile that is neither in [0, e_lo) nor in [t_hi, t_hi + e_hi)
    const uint32_t i = remap ? xcd_contiguous_block(b - n_edge, n_blocks - n_edge) : b - n_edge;
    const uint32_t t = e_lo + i;
    return t < t_hi ? t : t + e_hi;
}

// The host's side of that order: how many workgroups to start first for either edge, and where the high
// edge's tiles should be -- from the capacity of the exchanged lists (three rows at either edge, each
// <= ~0.7 capacity) and the host's estimate of where the live agents end (`n` agents covered by the launch,
// of which the last `slack` may be stale slots).  A hint: the kernel finds the real edge tiles itself.
struct EdgeHint { uint32_t e_lo, e_hi, t_hi; };
inline EdgeHint edge_first_hint(uint32_t n, uint32_t threads, uint32_t halo_cap, uint64_t slack)
{
    const uint32_t nb = (n + threads - 1u) / threads;
    const uint32_t cap = halo_cap ? halo_cap : 1u;
    uint32_t each = (3u * cap + threads - 1u) / threads;
    if (each > nb / 2u) each = nb / 2u;
    const uint32_t live_est = (uint32_t)(n > slack ? n - slack : 0u) + cap / 2u;
    uint32_t end_tile = (live_est + threads - 1u) / threads;
    if (end_tile > nb) end_tile = nb;
    EdgeHint h{each, each < nb - each ? each : nb - each, 0u};
    h.t_hi = end_tile > h.e_hi ? end_tile - h.e_hi : 0u;
    if (h.t_hi < h.e_lo) h.t_hi = h.e_lo;
    if (h.t_hi + h.e_hi > nb) h.t_hi = nb - h.e_hi;
    return h;
}

} // namespace pedoni
