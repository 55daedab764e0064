// The block-order arithmetic of pedoni_amd/csrc/block_order.hpp on the CPU: whatever the grid size and wherever the
// host's hint puts the edge tiles, every hardware workgroup gets a tile of its own and every tile a workgroup.
#include "block_order.hpp"
#include <cstdio>
#include <vector>
using namespace pedoni;

static int failures = 0;
#define CHECK(cond, ...) do { if (!(cond)) { ++failures; std::printf("FAILED %s:%d: ", __FILE__, __LINE__); std::printf(__VA_ARGS__); std::printf("\n"); } } while (0)

template <typename F> static bool bijection(uint32_t n, F&& f)
{
    std::vector<char> seen(n, 0);
    for (uint32_t b = 0; b < n; ++b) {
        const uint32_t t = f(b);
        if (t >= n || seen[t]) return false;
        seen[t] = 1;
    }
    return true;
}

int main()
{
    // XCD-contiguous order: a bijection, and the blocks of one XCD (b % 8 equal) take consecutive tiles
    for (uint32_t n = 1; n <= 700; ++n) {
        CHECK(bijection(n, [&](uint32_t b) { return xcd_contiguous_block(b, n); }), "xcd order, %u blocks", n);
        for (uint32_t b = 0; b + 8 < n; ++b)
            CHECK(xcd_contiguous_block(b + 8, n) == xcd_contiguous_block(b, n) + 1, "xcd order, %u blocks, block %u", n, b);
    }
    // edge-first order: every admissible (e_lo, e_hi, t_hi), with and without the XCD order inside
    for (uint32_t n = 1; n <= 48; ++n)
        for (uint32_t e_lo = 0; e_lo <= n; ++e_lo)
            for (uint32_t e_hi = 0; e_lo + e_hi <= n; ++e_hi)
                for (uint32_t t_hi = e_lo; t_hi + e_hi <= n; ++t_hi)
                    for (int remap = 0; remap < 2; ++remap) {
                        CHECK(bijection(n, [&](uint32_t b) { return edge_first_tile(b, n, e_lo, e_hi, t_hi, remap); }),
                              "edge-first order, n %u e_lo %u e_hi %u t_hi %u remap %d", n, e_lo, e_hi, t_hi, remap);
                        for (uint32_t b = 0; b < e_lo; ++b)
                            CHECK(edge_first_tile(b, n, e_lo, e_hi, t_hi, remap) == b, "low edge first");
                        for (uint32_t b = 0; b < e_hi; ++b)
                            CHECK(edge_first_tile(e_lo + b, n, e_lo, e_hi, t_hi, remap) == t_hi + b, "high edge next");
                    }
    // the host's hint is always admissible, and a bijection follows from it
    const uint32_t T = 256;
    for (uint32_t n : {1u, 2u, 255u, 256u, 257u, 1000u, 5000u, 40000u, 250000u, 1000003u, 1300000u})
        for (uint32_t cap : {0u, 1u, 256u, 2304u, 4096u, 100000u})
            for (uint64_t slack : {0ull, 1ull, 2304ull, 18432ull, 36864ull, 1ull << 40}) {
                const EdgeHint h = edge_first_hint(n, T, cap, slack);
                const uint32_t nb = (n + T - 1) / T;
                CHECK(h.e_lo <= h.t_hi && h.t_hi + h.e_hi <= nb && h.e_lo + h.e_hi <= nb,
                      "hint n %u cap %u slack %llu: e_lo %u e_hi %u t_hi %u of %u", n, cap, (unsigned long long)slack, h.e_lo, h.e_hi, h.t_hi, nb);
                if (nb <= 6000)
                    CHECK(bijection(nb, [&](uint32_t b) { return edge_first_tile(b, nb, h.e_lo, h.e_hi, h.t_hi, 1); }),
                          "hinted order n %u cap %u", n, cap);
            }
    // a band of 1e6 agents, 2304-agent lists, one unpack since the count was read: both edges get their workgroups,
    // the high edge's tiles end about where the live agents do
    {
        const uint32_t n = 1000000u + 2u * 2304u;
        const EdgeHint h = edge_first_hint(n, T, 2304u, 2u * 2304u);
        CHECK(h.e_lo == 27 && h.e_hi == 27, "3 x 2304 agents = 27 workgroups per edge, got %u / %u", h.e_lo, h.e_hi);
        const uint32_t live_tile = 1000000u / T;
        CHECK(h.t_hi <= live_tile - 20 && h.t_hi + h.e_hi >= live_tile, "high edge tiles [%u, %u) around tile %u", h.t_hi, h.t_hi + h.e_hi, live_tile);
    }
    if (failures) { std::printf("%d checks failed\n", failures); return 1; }
    std::printf("all checks passed\n");
    return 0;
}
