// host_diag.hpp -- the host side of the diagnostics build (included by pedoni_hip.hip under PEDONI_DIAGNOSTICS,
// i.e. part of libpedoni_hip_diag.so only): which diagnostic instantiation of the force kernel a launch takes
// (PEDONI_FORCE_TRACE, PEDONI_ABLATE, PEDONI_FORCE_PERSIST) and the place kernel's ablation / dispatch-cost probes.
#pragma once

namespace {

// the place kernel with parts switched off (m->place_ablate: pedoni_hip_debug_set_ablate bits 8 and up), and the
// dispatch-cost probes launched behind it (tools/place_probe.sh, profiles/r03_place_ablation.txt)
inline void diag_launch_place(PedoniModel* m, uint32_t blocks, uint32_t bs, uint32_t i0, uint32_t n_total, const BandView& band,
                              int cs_old, int cs_new, uint32_t parity, const SoA& soa, HaloIn* consumed, int32_t row0, int32_t row1,
                              uint32_t* tickets, uint32_t* done_count)
{
    const dim3 g(blocks), b(bs);
    uint32_t* const none = nullptr;
    HaloIn* const no_halo = nullptr;
    hipLaunchKernelGGL(place_kernel_diag, g, b, 0, m->stream, m->d_key, i0, n_total, m->grid, band, m->d_cs[cs_old], m->d_cs[cs_new],
                       m->d_flags, parity, m->d_scan_in, soa, m->d_slots, consumed, m->d_row_count, row0, row1, m->d_live + 1,
                       tickets, done_count, m->place_ablate);   // (no tile order in a diagnostic pass)
    if (m->place_ablate & 256u)      // the same body a second time under another name, with the switches of bits 16 and up (default: returning at once)
        hipLaunchKernelGGL(place_kernel_probe, g, b, 0, m->stream, m->d_key, i0, n_total, m->grid, band, m->d_cs[cs_old], m->d_cs[cs_new],
                           m->d_flags, parity, m->d_scan_in, soa, m->d_slots, no_halo, m->d_row_count, row0, row1, m->d_live + 1, none, none,
                           (m->place_ablate >> 16) ? (m->place_ablate >> 16) : 128u);
    if (m->place_ablate & 4096u) {   // the signature reading one / all of its arguments
        hipLaunchKernelGGL(place_reads_one, g, b, 0, m->stream, m->d_key, i0, n_total, m->grid, band, m->d_cs[cs_old], m->d_cs[cs_new],
                           m->d_flags, parity, m->d_scan_in, soa, m->d_slots, no_halo, m->d_row_count, row0, row1, m->d_live + 1, none, none, 128u);
        hipLaunchKernelGGL(place_reads_all, g, b, 0, m->stream, m->d_key, i0, n_total, m->grid, band, m->d_cs[cs_old], m->d_cs[cs_new],
                           m->d_flags, parity, m->d_scan_in, soa, m->d_slots, no_halo, m->d_row_count, row0, row1, m->d_live + 1, none, none, 128u);
    }
    if (m->place_ablate & 2048u)     // the place kernel's signature, empty body, same grid and arguments
        hipLaunchKernelGGL(place_signature_only, g, b, 0, m->stream, m->d_key, i0, n_total, m->grid, band, m->d_cs[cs_old], m->d_cs[cs_new],
                           m->d_flags, parity, m->d_scan_in, soa, m->d_slots, no_halo, m->d_row_count, row0, row1, m->d_live + 1, none, none, 128u);
    if (m->place_ablate & 1024u)     // a two-argument empty kernel on the same grid
        hipLaunchKernelGGL(probe_empty_kernel, g, b, 0, m->stream, m->d_slots, 1u);
    if (m->place_ablate & 512u)      // a grid of 64 workgroups, returning at once
        hipLaunchKernelGGL(place_kernel_diag, dim3(64), b, 0, m->stream, m->d_key, i0, n_total, m->grid, band, m->d_cs[cs_old], m->d_cs[cs_new],
                           m->d_flags, parity, m->d_scan_in, soa, m->d_slots, no_halo, m->d_row_count, row0, row1, m->d_live + 1, none, none, 128u);
}

// Does this launch take a diagnostic instantiation of the force kernel?  true: it has been launched (*rc says how
// that went) and launch_force is done; false: the product kernel follows.  `c` is the plan of the product launch.
inline bool diag_launch_force(PedoniModel* m, ForceArgs& a, const ForcePlan& c, uint32_t n, dim3 grid, dim3 block, hipStream_t stream,
                              bool fast, int part, bool on_side_stream, int* rc)
{
    auto launched = [&]() -> int {
        const hipError_t e = hipGetLastError();
        return e == hipSuccess ? PEDONI_OK : fail(PEDONI_E_HIP, std::string("diagnostic force launch: ") + hipGetErrorString(e));
    };
    const bool on = on_side_stream;
    // diagnostic instantiations (per-phase trace, ablation switches): a build of their own
    if (m->d_trace && c.group > 1) {       // per-wave records of the group kernel (tools/group_trace.py)
        const dim3 ggrid(blocks_for(n, FORCE_THREADS / (uint32_t)c.group));
        if ((size_t)ggrid.x * FORCE_WAVES <= TRACE_WAVES) {
            if (c.group == 2) hipLaunchKernelGGL((force_kernel_queue_group_trace<0, 8, 2>), ggrid, block, 0, stream, a);
            else hipLaunchKernelGGL((force_kernel_queue_group_trace<0, 6, 4>), ggrid, block, 0, stream, a);
            *rc = launched();
            return true;
        }
    }
    if (m->d_trace && (size_t)grid.x * FORCE_WAVES <= TRACE_WAVES) {
        if (fast) hipLaunchKernelGGL((force_kernel_queue_trace<1, 6>), grid, block, 0, stream, a);
        else hipLaunchKernelGGL((force_kernel_queue_trace<0, 6>), grid, block, 0, stream, a);
        *rc = launched();
        return true;
    }
    if (m->ablate) {
        if (fast) hipLaunchKernelGGL((force_kernel_queue_ablate<1, 6>), grid, block, 0, stream, a);
        else hipLaunchKernelGGL((force_kernel_queue_ablate<0, 6>), grid, block, 0, stream, a);
        *rc = launched();
        return true;
    }
    if (m->force_persist == 21 && part == 0) {     // experiment: one wave per workgroup
        const dim3 g1(((blocks_for(n, 64) + 3u) / 4u) * 4u), b1(64);
        if (fast) hipLaunchKernelGGL((force_kernel_queue_w1<1, 6>), g1, b1, 0, stream, a);
        else hipLaunchKernelGGL((force_kernel_queue_w1<0, 6>), g1, b1, 0, stream, a);
        *rc = launched();
        return true;
    }
    // persistent-wave forms (kernels.hpp: a measured dead end, PEDONI_FORCE_PERSIST): whole-array
    // launches right after a sort pass, whose place kernel zeroed the tile tickets
    const bool persist = part == 0 && !on && m->tickets_fresh && c.build == ForceBuild::S94 && c.slots == 6 &&
                         m->force_persist > 0;
    if (persist) {
        a.tickets = m->d_tickets;
        a.n_tiles = blocks_for(n, 64);
        m->tickets_fresh = false;
        // the grid is what the chip holds at once: waves per SIMD (= blocks per CU) x 256 CUs
        const uint32_t waves = m->force_persist == 5 || m->force_persist == 6 ? (uint32_t)m->force_persist : 7u;
        const dim3 pgrid(std::min(blocks_for(n, FORCE_THREADS), waves * 256u));
        if (m->force_persist >= 15 && m->force_persist <= 17) {
            const dim3 sgrid(std::min(blocks_for(n, FORCE_THREADS), (uint32_t)(m->force_persist - 10) * 256u));
            if (m->force_persist == 15) {
                if (fast) hipLaunchKernelGGL((force_kernel_queue_static5<1, 6>), sgrid, block, 0, stream, a);
                else hipLaunchKernelGGL((force_kernel_queue_static5<0, 6>), sgrid, block, 0, stream, a);
            } else if (m->force_persist == 16) {
                if (fast) hipLaunchKernelGGL((force_kernel_queue_static6<1, 6>), sgrid, block, 0, stream, a);
                else hipLaunchKernelGGL((force_kernel_queue_static6<0, 6>), sgrid, block, 0, stream, a);
            } else {
                if (fast) hipLaunchKernelGGL((force_kernel_queue_static7<1, 6>), sgrid, block, 0, stream, a);
                else hipLaunchKernelGGL((force_kernel_queue_static7<0, 6>), sgrid, block, 0, stream, a);
            }
        } else if (waves == 7) {
            if (fast) hipLaunchKernelGGL((force_kernel_queue_persist<1, 6>), pgrid, block, 0, stream, a);
            else hipLaunchKernelGGL((force_kernel_queue_persist<0, 6>), pgrid, block, 0, stream, a);
        } else if (waves == 6) {
            if (fast) hipLaunchKernelGGL((force_kernel_queue_persist6<1, 6>), pgrid, block, 0, stream, a);
            else hipLaunchKernelGGL((force_kernel_queue_persist6<0, 6>), pgrid, block, 0, stream, a);
        } else {
            if (fast) hipLaunchKernelGGL((force_kernel_queue_persist5<1, 6>), pgrid, block, 0, stream, a);
            else hipLaunchKernelGGL((force_kernel_queue_persist5<0, 6>), pgrid, block, 0, stream, a);
        }
        *rc = launched();
        return true;
    }
    return false;
}

} // namespace
