/* c_shard_demo.c -- a multi-GPU host in plain C: one process per GPU, nothing but the C-ABI.
 *
 * What a Rust / C / C++ host needs to run N GPUs of one node (include/pedoni_hip.h,
 * pedoni_shard_*): the library owns the RCCL communicator and sends / receives the ghost and
 * migrant lists itself; the host only hands the 128-byte id of rank 0 to every rank (here:
 * through a file) and appends each band's agents.
 *
 *   gcc -std=c11 -Iinclude examples/c_shard_demo.c -Lpedoni_amd/lib -lpedoni_host -lpedoni_hip -lm \
 *       -Wl,-rpath,$PWD/pedoni_amd/lib -o c_shard_demo
 *   for r in 0 1 2 3 4 5 6 7; do ./c_shard_demo scenario.toml $r 8 /tmp/pedoni.id 1000 & done; wait
 *
 * Rank r uses HIP device r.  With world = 1 no file is needed (pass "-").
 */
#define _POSIX_C_SOURCE 200809L
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "pedoni_hip.h"
#include "pedoni_host.h"

#define CHECK_HOST(call) do { if ((call) != 0) { fprintf(stderr, "%s: %s\n", #call, pedoni_host_last_error()); return 1; } } while (0)
#define CHECK_HIP(call) do { if ((call) != 0) { fprintf(stderr, "%s: %s\n", #call, pedoni_hip_last_error()); return 1; } } while (0)

static int share_id(const char* path, int rank, uint8_t id[PEDONI_SHARD_ID_BYTES])
{
    char tmp[1024];
    if (rank == 0) {
        if (pedoni_shard_unique_id(id) != 0) { fprintf(stderr, "unique_id: %s\n", pedoni_hip_last_error()); return 1; }
        snprintf(tmp, sizeof tmp, "%s.tmp", path);
        FILE* f = fopen(tmp, "wb");
        if (!f || fwrite(id, 1, PEDONI_SHARD_ID_BYTES, f) != PEDONI_SHARD_ID_BYTES) { perror(tmp); return 1; }
        fclose(f);
        if (rename(tmp, path) != 0) { perror(path); return 1; }    /* appears whole or not at all */
        return 0;
    }
    for (int tries = 0; tries < 600; ++tries) {                    /* up to 60 s */
        FILE* f = fopen(path, "rb");
        if (f) {
            size_t n = fread(id, 1, PEDONI_SHARD_ID_BYTES, f);
            fclose(f);
            if (n == PEDONI_SHARD_ID_BYTES) return 0;
        }
        struct timespec ts = {0, 100 * 1000 * 1000};
        nanosleep(&ts, NULL);
    }
    fprintf(stderr, "rank %d: no id in %s\n", rank, path);
    return 1;
}

int main(int argc, char** argv)
{
    if (argc < 5) { fprintf(stderr, "usage: %s scenario.toml rank world idfile|- [ticks]\n", argv[0]); return 2; }
    const int rank = atoi(argv[2]), world = atoi(argv[3]);
    const int ticks = argc > 5 ? atoi(argv[5]) : 100;
    if (world < 1 || rank < 0 || rank >= world) { fprintf(stderr, "bad rank / world\n"); return 2; }

    FILE* fp = fopen(argv[1], "rb");
    if (!fp) { perror(argv[1]); return 2; }
    static char text[1 << 20];
    size_t n = fread(text, 1, sizeof text - 1, fp);
    fclose(fp);
    text[n] = 0;

    PedoniScenario* sc = NULL;
    PedoniField* field = NULL;
    CHECK_HOST(pedoni_scenario_parse(text, &sc));
    CHECK_HOST(pedoni_field_from_scenario(sc, 0.25f, &field));      /* every rank builds the whole field ... */
    float size[2], unit;
    uint32_t frows, fcols, n_maps, n_obs = 0, n_wp = 0;
    CHECK_HOST(pedoni_scenario_size(sc, size));
    CHECK_HOST(pedoni_field_shape(field, &frows, &fcols, &n_maps, &unit));
    CHECK_HOST(pedoni_scenario_segments(sc, 1, NULL, 0, &n_obs));
    CHECK_HOST(pedoni_scenario_segments(sc, 0, NULL, 0, &n_wp));
    PedoniObstacle* obs = calloc(n_obs ? n_obs : 1, sizeof *obs);
    CHECK_HOST(pedoni_scenario_segments(sc, 1, (float*)obs, n_obs, &n_obs));
    const float** maps = calloc(n_maps ? n_maps : 1, sizeof *maps);
    for (uint32_t k = 0; k < n_maps; ++k) maps[k] = pedoni_field_potential_map(field, k);

    PedoniOptions opt;
    pedoni_hip_default_options(&opt);
    /* bands of equal grid rows (a host with the crowd at hand would call pedoni_shard_balanced_bounds) */
    const int32_t grid_rows = (int32_t)ceilf(size[1] / opt.neighbor_grid_unit);   /* neighbor_grid.rs:14-20 */
    int32_t* bounds = calloc((size_t)world + 1, sizeof *bounds);
    for (int r = 0; r <= world; ++r) bounds[r] = (int32_t)((int64_t)grid_rows * r / world);
    const int32_t slack = 4;
    uint32_t y0 = 0, y1 = frows;
    if (world > 1)                                                   /* ... but uploads only its band's rows */
        CHECK_HIP(pedoni_shard_map_rows(bounds[rank], bounds[rank + 1], slack, opt.neighbor_grid_unit, unit, frows, &y0, &y1));

    PedoniModel* model = NULL;
    CHECK_HIP(pedoni_hip_create_rows(&opt, size[0], size[1], pedoni_field_distance_map(field), maps, n_maps, frows,
                                     fcols, unit, obs, n_obs, /*device=*/world > 1 ? rank : 0, y0, y1, &model));
    uint8_t id[PEDONI_SHARD_ID_BYTES];
    const uint8_t* idp = NULL;
    if (strcmp(argv[4], "-") != 0) {
        if (share_id(argv[4], rank, id)) return 1;
        idp = id;
    } else if (world > 1) {
        fprintf(stderr, "world > 1 needs an id file\n");
        return 2;
    } else {
        CHECK_HIP(pedoni_shard_unique_id(id));                       /* a one-rank communicator all the same */
        idp = id;
    }
    PedoniShard* shard = NULL;
    CHECK_HIP(pedoni_shard_create(model, rank, world, idp, bounds, /*halo_cap=*/4096, &shard));
    CHECK_HIP(pedoni_shard_selftest(shard));
    if (world > 1) CHECK_HIP(pedoni_shard_set_rebalance(shard, 64, 2, slack));

    /* this band's agents: a 0.9 m lattice over its rows, where the distance map says free space */
    const float y_lo = bounds[rank] * opt.neighbor_grid_unit, y_hi = bounds[rank + 1] * opt.neighbor_grid_unit;
    size_t cap = 1024, n_agents = 0;
    float* pos = malloc(cap * 2 * sizeof *pos);
    uint32_t* dest = malloc(cap * sizeof *dest);
    for (float y = 1.0f; y < size[1] - 1.0f; y += 0.9f) {
        if (!(y >= y_lo && y < y_hi)) continue;
        for (float x = 1.0f; x < size[0] - 1.0f; x += 0.9f) {
            float d = 0.0f;
            CHECK_HOST(pedoni_field_get_obstacle_distance(field, x, y, &d));
            if (!(d > 0.6f)) continue;
            if (n_agents == cap) {
                cap *= 2;
                pos = realloc(pos, cap * 2 * sizeof *pos);
                dest = realloc(dest, cap * sizeof *dest);
            }
            pos[2 * n_agents] = x; pos[2 * n_agents + 1] = y;
            dest[n_agents++] = n_wp > 1 ? 1u : 0u;
        }
    }
    if (n_agents) CHECK_HIP(pedoni_hip_append(model, pos, dest, NULL, NULL, (uint32_t)n_agents));
    CHECK_HIP(pedoni_shard_begin(shard));
    int32_t before = 0, after = 0, lo = 0, hi = 0;
    CHECK_HIP(pedoni_shard_owned_count(shard, &before));
    CHECK_HIP(pedoni_shard_tick_n(shard, (uint32_t)ticks));
    CHECK_HIP(pedoni_shard_owned_count(shard, &after));
    CHECK_HIP(pedoni_shard_band(shard, &lo, &hi));
    printf("rank=%d/%d band=[%d,%d) map_rows=[%u,%u) of %u ticks=%d owned %d -> %d\n", rank, world, lo, hi, y0, y1,
           frows, ticks, before, after);

    pedoni_shard_destroy(shard);
    pedoni_hip_destroy(model);
    pedoni_field_free(field);
    pedoni_scenario_free(sc);
    free(obs); free((void*)maps); free(bounds); free(pos); free(dest);
    return 0;
}
