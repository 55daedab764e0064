/* oracle_sim.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * Restatement of the spawn logic of pedoni-simulator/src/lib.rs (Simulator::new
 * :37-52 and Simulator::tick :67-85): which agents are handed to
 * PedestrianModel::spawn_pedestrians, drawn with the build-owned RNG (the reference's
 * fastrand global is unseeded).
 */
#include "pedoni_oracle.h"
#include "oracle_math.h"

/* lib.rs:38-50: for each `once{count}` spawner: pos = p1.lerp(p2, fastrand::f32()).
 * Emits up to `cap` agents; returns the number the scenario asks for. */
uint32_t oracle_sim_spawn_once(uint64_t* rng, const oracle_segment* origin_line, int32_t count,
                               uint32_t destination, float* pos_xy, uint32_t* dest_out,
                               uint32_t cap)
{
    uint32_t n = 0;
    for (int32_t k = 0; k < count; ++k) {
        ovec2 p = ov_lerp(ov(origin_line->x0, origin_line->y0),
                          ov(origin_line->x1, origin_line->y1), oracle_rng_f32(rng));
        if (n < cap) { pos_xy[2 * n] = p.x; pos_xy[2 * n + 1] = p.y; dest_out[n] = destination; }
        ++n;
    }
    return n;
}

/* lib.rs:70-83: for each `periodic{frequency}` spawner: count = poisson(frequency / 10.0) */
uint32_t oracle_sim_spawn_periodic(uint64_t* rng, const oracle_segment* origin_line,
                                   double frequency, uint32_t destination, float* pos_xy,
                                   uint32_t* dest_out, uint32_t cap)
{
    int32_t count = oracle_poisson(rng, frequency / 10.0);
    return oracle_sim_spawn_once(rng, origin_line, count, destination, pos_xy, dest_out, cap);
}
