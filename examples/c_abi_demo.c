/* c_abi_demo.c -- the drop-in boundary from plain C.
 *
 * Parses a scenario TOML and builds the Field with the host mirror (include/pedoni_host.h),
 * then drives the model through the five `trait PedestrianModel` entry points of
 * include/pedoni_hip.h exactly as Simulator::new / Simulator::tick do upstream
 * (pedoni-simulator/src/lib.rs:27-61, 64-100).
 *
 *   gcc -std=c11 -Iinclude examples/c_abi_demo.c -Lpedoni_amd/lib -lpedoni_host -lpedoni_hip \
 *       -Wl,-rpath,$PWD/pedoni_amd/lib -o c_abi_demo && ./c_abi_demo scenario.toml 200
 */
#include <stdio.h>
#include <stdlib.h>

#include "pedoni_hip.h"
#include "pedoni_host.h"

#define CHECK_HOST(call) do { if ((call) != 0) { fprintf(stderr, "%s: %s\n", #call, pedoni_host_last_error()); return 1; } } while (0)
#define CHECK_HIP(call) do { if ((call) != 0) { fprintf(stderr, "%s: %s\n", #call, pedoni_hip_last_error()); return 1; } } while (0)

int main(int argc, char** argv)
{
    if (argc < 2) { fprintf(stderr, "usage: %s scenario.toml [ticks]\n", argv[0]); return 2; }
    int ticks = argc > 2 ? atoi(argv[2]) : 100;

    FILE* fp = fopen(argv[1], "rb");
    if (!fp) { perror(argv[1]); return 2; }
    static char text[1 << 20];
    size_t n = fread(text, 1, sizeof text - 1, fp);
    fclose(fp);
    text[n] = 0;

    PedoniScenario* sc = NULL;
    PedoniField* field = NULL;
    CHECK_HOST(pedoni_scenario_parse(text, &sc));                    /* toml::from_str */
    CHECK_HOST(pedoni_field_from_scenario(sc, 0.25f, &field));       /* Field::from_scenario */

    float size[2];
    uint32_t rows, cols, n_maps, n_obs = 0, n_wp = 0;
    float unit;
    CHECK_HOST(pedoni_scenario_size(sc, size));
    CHECK_HOST(pedoni_field_shape(field, &rows, &cols, &n_maps, &unit));
    CHECK_HOST(pedoni_scenario_segments(sc, 1, NULL, 0, &n_obs));
    CHECK_HOST(pedoni_scenario_segments(sc, 0, NULL, 0, &n_wp));
    PedoniObstacle* obs = calloc(n_obs ? n_obs : 1, sizeof *obs);
    float* wps = calloc(n_wp ? n_wp : 1, 5 * sizeof(float));
    CHECK_HOST(pedoni_scenario_segments(sc, 1, (float*)obs, n_obs, &n_obs));
    CHECK_HOST(pedoni_scenario_segments(sc, 0, wps, n_wp, &n_wp));
    const float** maps = calloc(n_maps ? n_maps : 1, sizeof *maps);
    for (uint32_t k = 0; k < n_maps; ++k) maps[k] = pedoni_field_potential_map(field, k);

    PedoniOptions opt;
    pedoni_hip_default_options(&opt);                                /* SimulatorOptions::default */
    PedoniModel* model = NULL;
    CHECK_HIP(pedoni_hip_create(&opt, size[0], size[1], pedoni_field_distance_map(field), maps, n_maps,
                                rows, cols, unit, obs, n_obs, 0, &model)); /* PedestrianModel::new */

    /* Simulator::new: 20 agents on waypoint 0's line, walking to waypoint 1 */
    PedoniPedestrian peds[20];
    for (int k = 0; k < 20; ++k) {
        float u = (k + 0.5f) / 20.0f;
        peds[k].x = wps[0] * (1.0f - u) + wps[2] * u;
        peds[k].y = wps[1] * (1.0f - u) + wps[3] * u;
        peds[k].destination = n_wp > 1 ? 1 : 0;
    }
    CHECK_HIP(pedoni_hip_spawn_pedestrians(model, peds, 20));

    int32_t count = 0;
    for (int t = 0; t < ticks; ++t) {                                /* Simulator::tick */
        CHECK_HIP(pedoni_hip_spawn_pedestrians(model, NULL, 0));
        CHECK_HIP(pedoni_hip_update_states(model));
        CHECK_HIP(pedoni_hip_get_pedestrian_count(model, &count));
    }
    PedoniPedestrian out[20];
    uint32_t live = 0;
    CHECK_HIP(pedoni_hip_list_pedestrians(model, out, 20, &live));
    printf("ticks=%d active=%d", ticks, count);
    if (live) printf(" first=(%.4f, %.4f)->%llu", out[0].x, out[0].y, (unsigned long long)out[0].destination);
    printf("\n");

    pedoni_hip_destroy(model);
    pedoni_field_free(field);
    pedoni_scenario_free(sc);
    free(obs); free(wps); free((void*)maps);
    return 0;
}
