/* oracle_sfm.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * Restatement of pedoni-simulator/src/models/sfm.rs (SocialForceModel, the CPU path
 * that is the parity target) and src/neighbor_grid.rs.
 *
 * Structure mirrors the reference: serial append / bin / stable cell sort / despawn
 * (sfm.rs:48-89), parallel-for over agents for accelerations where the reference
 * uses rayon (sfm.rs:93-95; OpenMP here), serial integrator (sfm.rs:245-254).
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -fopenmp.
 */
#include "pedoni_oracle.h"
#include "oracle_math.h"

#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* sfm.rs:16 */
static const float COS_PHI = -0.17364817766693036f;

/* sfm.rs:26-33: soa_derive turns `Pedestrian` into four parallel Vecs */
typedef struct {
    ovec2* position;
    uint32_t* destination;
    ovec2* velocity;
    float* desired_speed;
    size_t len, cap;
} ped_vec;

struct oracle_model {
    ped_vec peds;
    ped_vec scratch;                 /* `sorted_pedestrians` of sfm.rs:61 */
    int has_grid;                    /* Option<NeighborGrid> */
    float grid_unit;
    int32_t grid_rows, grid_cols;    /* NeighborGrid.shape = (rows, cols) */
    uint32_t* cell_count;            /* per-cell list lengths  (neighbor_grid.rs data) */
    uint32_t* cell_offset;           /* per-cell list starts   */
    uint32_t* cell_items;            /* concatenated per-cell index lists */
    size_t cell_items_cap;
    uint32_t* neighbor_grid_indices; /* sfm.rs:22; rows*cols+1 once built */
    size_t ngi_len;
    oracle_options opt;
    uint64_t rng;
    int32_t n_threads;
};

static void pv_reserve(ped_vec* v, size_t cap)
{
    if (cap <= v->cap) return;
    size_t nc = v->cap ? v->cap : 64;
    while (nc < cap) nc *= 2;
    v->position = (ovec2*)realloc(v->position, nc * sizeof(ovec2));
    v->destination = (uint32_t*)realloc(v->destination, nc * sizeof(uint32_t));
    v->velocity = (ovec2*)realloc(v->velocity, nc * sizeof(ovec2));
    v->desired_speed = (float*)realloc(v->desired_speed, nc * sizeof(float));
    v->cap = nc;
}

static void pv_free(ped_vec* v)
{
    free(v->position); free(v->destination); free(v->velocity); free(v->desired_speed);
    memset(v, 0, sizeof(*v));
}

static inline void pv_push_from(ped_vec* dst, const ped_vec* src, size_t i)
{
    size_t k = dst->len++;
    dst->position[k] = src->position[i];
    dst->destination[k] = src->destination[i];
    dst->velocity[k] = src->velocity[i];
    dst->desired_speed[k] = src->desired_speed[i];
}

/* neighbor_grid.rs:14-20 */
void oracle_neighbor_grid_shape(float size_x, float size_y, float unit,
                                int32_t* rows, int32_t* cols)
{
    *rows = (int32_t)o_f32_as_usize(ceilf(size_y / unit));
    *cols = (int32_t)o_f32_as_usize(ceilf(size_x / unit));
}

/* sfm.rs:36-46 */
oracle_model* oracle_model_new(const oracle_options* opt, float size_x, float size_y)
{
    oracle_model* m = (oracle_model*)calloc(1, sizeof(oracle_model));
    m->opt = *opt;
    m->has_grid = opt->use_neighbor_grid != 0;
    if (m->has_grid) {
        m->grid_unit = opt->neighbor_grid_unit;
        oracle_neighbor_grid_shape(size_x, size_y, m->grid_unit, &m->grid_rows, &m->grid_cols);
        size_t cells = (size_t)m->grid_rows * m->grid_cols;
        m->cell_count = (uint32_t*)calloc(cells + 1, sizeof(uint32_t));
        m->cell_offset = (uint32_t*)calloc(cells + 1, sizeof(uint32_t));
        m->neighbor_grid_indices = (uint32_t*)calloc(cells + 1, sizeof(uint32_t));
    }
    m->rng = 12345;
    m->n_threads = 0;
    return m;
}

void oracle_model_free(oracle_model* m)
{
    if (!m) return;
    pv_free(&m->peds); pv_free(&m->scratch);
    free(m->cell_count); free(m->cell_offset); free(m->cell_items);
    free(m->neighbor_grid_indices);
    free(m);
}

void oracle_model_seed(oracle_model* m, uint64_t seed) { m->rng = seed; }
void oracle_model_set_threads(oracle_model* m, int32_t n) { m->n_threads = n; }

/* neighbor_grid.rs:22-36 `update`: clear every cell, then push index i into the cell
 * of (pos / unit).as_ivec2() when Index::index_checked accepts it (negative or
 * out-of-shape -> skipped).  Per-cell push order = ascending i; a two-pass counting
 * fill reproduces exactly those lists. */
static inline int64_t grid_cell_of(const oracle_model* m, ovec2 pos)
{
    int32_t ix = o_f32_as_i32(pos.x / m->grid_unit);
    int32_t iy = o_f32_as_i32(pos.y / m->grid_unit);
    if (ix < 0 || iy < 0 || iy >= m->grid_rows || ix >= m->grid_cols) return -1;
    return (int64_t)iy * m->grid_cols + ix;
}

static void neighbor_grid_update(oracle_model* m)
{
    size_t cells = (size_t)m->grid_rows * m->grid_cols, n = m->peds.len;
    memset(m->cell_count, 0, cells * sizeof(uint32_t));
    for (size_t i = 0; i < n; ++i) {
        int64_t c = grid_cell_of(m, m->peds.position[i]);
        if (c >= 0) m->cell_count[c]++;
    }
    uint32_t run = 0;
    for (size_t c = 0; c < cells; ++c) { m->cell_offset[c] = run; run += m->cell_count[c]; }
    m->cell_offset[cells] = run;
    if (n > m->cell_items_cap) {
        m->cell_items_cap = n * 2;
        m->cell_items = (uint32_t*)realloc(m->cell_items, m->cell_items_cap * sizeof(uint32_t));
    }
    memset(m->cell_count, 0, cells * sizeof(uint32_t));
    for (size_t i = 0; i < n; ++i) {
        int64_t c = grid_cell_of(m, m->peds.position[i]);
        if (c >= 0) m->cell_items[m->cell_offset[c] + m->cell_count[c]++] = (uint32_t)i;
    }
}

/* sfm.rs:48-89 */
void oracle_spawn_pedestrians(oracle_model* m, const oracle_field* f,
                              const float* pos_xy, const uint32_t* destination, uint32_t n,
                              const float* desired_speed, const float* vel_xy)
{
    pv_reserve(&m->peds, m->peds.len + n);
    for (uint32_t k = 0; k < n; ++k) {            /* :49-56 */
        size_t i = m->peds.len++;
        m->peds.position[i] = ov(pos_xy[2 * k], pos_xy[2 * k + 1]);
        m->peds.destination[i] = destination[k];
        m->peds.velocity[i] = vel_xy ? ov(vel_xy[2 * k], vel_xy[2 * k + 1]) : ov(0.0f, 0.0f);
        m->peds.desired_speed[i] = desired_speed
            ? desired_speed[k]
            : oracle_rng_normal_approx(&m->rng, 1.34f, 0.26f);
    }

    pv_reserve(&m->scratch, m->peds.len);
    m->scratch.len = 0;

    if (m->has_grid) {                            /* :58-77 */
        neighbor_grid_update(m);                  /* :59 */
        size_t cells = (size_t)m->grid_rows * m->grid_cols;
        size_t index = 0;
        m->neighbor_grid_indices[0] = 0;          /* :63 */
        for (size_t c = 0; c < cells; ++c) {      /* :66 row-major Array2::iter() */
            uint32_t off = m->cell_offset[c], cnt = m->cell_count[c];
            for (uint32_t j = 0; j < cnt; ++j) {  /* :67 */
                size_t i = m->cell_items[off + j];
                ovec2 p = m->peds.position[i];
                if (oracle_get_potential(f, m->peds.destination[i], p.x, p.y) > 0.25f) { /* :69 */
                    pv_push_from(&m->scratch, &m->peds, i);
                    index += 1;
                }
            }
            m->neighbor_grid_indices[c + 1] = (uint32_t)index; /* :74 */
        }
        m->ngi_len = cells + 1;
    } else {                                      /* :78-88 */
        for (size_t i = 0; i < m->peds.len; ++i) {
            ovec2 p = m->peds.position[i];
            if (oracle_get_potential(f, m->peds.destination[i], p.x, p.y) > 0.25f)
                pv_push_from(&m->scratch, &m->peds, i);
        }
    }
    ped_vec t = m->peds; m->peds = m->scratch; m->scratch = t;  /* :77,87 */
}

/* sfm.rs:131-153 == :160-182: force exerted on (pos, e) by neighbour (pos_i, vel_i) */
static inline void pair_force(ovec2 pos, ovec2 e, ovec2 pos_i, ovec2 vel_i, ovec2* acc)
{
    ovec2 difference = ov_sub(pos, pos_i);                     /* :131 */
    float distance_squared = ov_length_squared(difference);    /* :132 */
    if (distance_squared > 4.0f) return;                       /* :133-135 */

    float distance = sqrtf(distance_squared);                  /* :137 */
    ovec2 direction = ov_normalize(difference);                /* :138 */

    ovec2 t1 = ov_sub(difference, ov_scale(vel_i, 0.1f));      /* :141 */
    float t1_length = ov_length(t1);                           /* :142 */
    float t2 = distance + t1_length;                           /* :143 */
    float vl = ov_length(vel_i) * 0.1f;
    float b = sqrtf(t2 * t2 - vl * vl) * 0.5f;                 /* :144 powi(2) == x*x */

    /* :146  t2 * (direction + t1 / t1_length) / (4.0 * b) */
    ovec2 nabla_b = ov_div(ov_scale(ov_add(direction, ov_div(t1, t1_length)), t2), 4.0f * b);
    /* :147  ((2.1 / 0.3) * exp(-b / 0.3)) * nabla_b ; 2.1f/0.3f folds to 6.9999995f */
    float k = (2.1f / 0.3f) * expf(-b / 0.3f);
    ovec2 force = ov_scale(nabla_b, k);

    if (ov_dot(e, ov_neg(force)) < ov_length(force) * COS_PHI)  /* :149 */
        force = ov_scale(force, 0.5f);                         /* :150 */

    *acc = ov_add(*acc, force);                                /* :153 */
}

/* sfm.rs:96-239: acceleration of agent `id` */
static ovec2 acceleration_of(const oracle_model* m, const oracle_field* f,
                             const oracle_segment* obstacles, uint32_t n_obstacles, size_t id)
{
    const ped_vec* P = &m->peds;
    ovec2 pos = P->position[id];
    uint32_t destination = P->destination[id];
    ovec2 vel = P->velocity[id];
    float desired_speed = P->desired_speed[id];

    ovec2 acc = ov(0.0f, 0.0f);                                /* :104 */

    float g[2];
    oracle_get_potential_grad(f, destination, pos.x, pos.y, g); /* :107 */
    ovec2 e = ov_normalize(ov(g[0], g[1]));                    /* :108 */
    acc = ov_add(acc, ov_div(ov_sub(ov_scale(e, desired_speed), vel), 0.5f)); /* :109 */

    if (m->has_grid) {                                         /* :112-156 */
        int32_t ix = o_f32_as_i32(pos.x / m->grid_unit);       /* :113 */
        int32_t iy = o_f32_as_i32(pos.y / m->grid_unit);
        int32_t shape_x = m->grid_cols, shape_y = m->grid_rows; /* :116 */
        /* i32 arithmetic as upstream; saturated inputs wrap identically in int64->int32 */
        int64_t y_start = (int64_t)iy - 1; if (y_start < 0) y_start = 0;          /* :117 */
        int64_t y_end = (int64_t)iy + 1; if (y_end > shape_y - 1) y_end = shape_y - 1; /* :118 */
        int64_t x_start = (int64_t)ix - 1; if (x_start < 0) x_start = 0;          /* :119 */
        int64_t x_end = (int64_t)ix + 1; if (x_end > shape_x - 1) x_end = shape_x - 1; /* :120 */

        for (int64_t y = y_start; y <= y_end; ++y) {           /* :122 */
            int64_t offset = y * shape_x;                      /* :123 */
            size_t i_start = m->neighbor_grid_indices[offset + x_start];   /* :124-125 */
            size_t i_end = m->neighbor_grid_indices[offset + x_end + 1];   /* :126-127 */
            for (size_t i = i_start; i < i_end; ++i)           /* :129 */
                if (i != id)                                   /* :130 */
                    pair_force(pos, e, P->position[i], P->velocity[i], &acc);
        }
    } else {                                                   /* :157-185 */
        for (size_t i = 0; i < P->len; ++i)
            if (i != id)
                pair_force(pos, e, P->position[i], P->velocity[i], &acc);
    }

    if (m->opt.use_distance_map) {                             /* :188-192 */
        float distance = oracle_get_obstacle_distance(f, pos.x, pos.y);
        oracle_get_obstacle_distance_grad(f, pos.x, pos.y, g);
        ovec2 direction = ov_neg(ov_normalize(ov(g[0], g[1])));
        float k = (10.0f * 0.2f) * expf(-distance / 0.2f);     /* :191 */
        acc = ov_add(acc, ov_scale(direction, k));
    } else {                                                   /* :193-236 */
        for (uint32_t o = 0; o < n_obstacles; ++o) {
            ovec2 v0 = ov(obstacles[o].x0, obstacles[o].y0);
            ovec2 v1 = ov(obstacles[o].x1, obstacles[o].y1);
            float w = obstacles[o].width;
            ovec2 d = ov_sub(v1, v0);                          /* :197 */
            float h = ov_length(d);                            /* :198 */
            ovec2 n = ov_scale(ov_scale(ov_normalize_or_zero(ov(d.y, -d.x)), w), 0.5f); /* :199 */
            float lines[4][4] = {                              /* :200-205 */
                { v0.x + n.x, v0.y + n.y, v0.x - n.x, v0.y - n.y },
                { v1.x + n.x, v1.y + n.y, v1.x - n.x, v1.y - n.y },
                { v0.x + n.x, v0.y + n.y, v1.x + n.x, v1.y + n.y },
                { v0.x - n.x, v0.y - n.y, v1.x - n.x, v1.y - n.y },
            };
            float diffs[4][2], distances[4];
            for (int k = 0; k < 4; ++k) {                      /* :206-210 */
                oracle_distance_from_line(pos.x, pos.y, lines[k], diffs[k]);
                distances[k] = ov_length(ov(diffs[k][0], diffs[k][1]));
            }
            if (distances[0] < w && distances[1] < w && distances[2] < h && distances[3] < h)
                continue;                                      /* :211-217 */
            int min_index = 0;                                 /* :218-222 min_by: first minimum */
            for (int k = 1; k < 4; ++k)
                if (distances[k] < distances[min_index]) min_index = k;
            float min_d = distances[min_index];
            ovec2 direction = ov_normalize(ov(diffs[min_index][0], diffs[min_index][1])); /* :223 */
            float k2 = (10.0f * 0.2f) * expf(-min_d / 0.2f);   /* :225 */
            acc = ov_add(acc, ov_scale(direction, k2));        /* :226 */
        }
    }
    return acc;
}

/* test hook: sfm.rs:130-153 for n independent (agent, neighbour) pairs; acc_xy is in/out */
void oracle_pair_forces(const float* pos_xy, const float* e_xy, const float* pos_i_xy,
                        const float* vel_i_xy, float* acc_xy, uint32_t n)
{
    for (uint32_t k = 0; k < n; ++k) {
        ovec2 acc = ov(acc_xy[2 * k], acc_xy[2 * k + 1]);
        pair_force(ov(pos_xy[2 * k], pos_xy[2 * k + 1]), ov(e_xy[2 * k], e_xy[2 * k + 1]),
                   ov(pos_i_xy[2 * k], pos_i_xy[2 * k + 1]), ov(vel_i_xy[2 * k], vel_i_xy[2 * k + 1]), &acc);
        acc_xy[2 * k] = acc.x;
        acc_xy[2 * k + 1] = acc.y;
    }
}

void oracle_calc_accelerations(const oracle_model* m, const oracle_field* f,
                               const oracle_segment* obstacles, uint32_t n_obstacles,
                               float* acc_xy)
{
    long n = (long)m->peds.len;
#ifdef _OPENMP
    int nt = m->n_threads > 0 ? m->n_threads : omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 256) num_threads(nt)
#endif
    for (long id = 0; id < n; ++id) {                          /* :93-95 into_par_iter */
        ovec2 a = acceleration_of(m, f, obstacles, n_obstacles, (size_t)id);
        acc_xy[2 * id] = a.x;
        acc_xy[2 * id + 1] = a.y;
    }
}

/* sfm.rs:91-255 */
void oracle_update_states(oracle_model* m, const oracle_field* f,
                          const oracle_segment* obstacles, uint32_t n_obstacles)
{
    size_t n = m->peds.len;
    float* acc = (float*)malloc((n ? n : 1) * 2 * sizeof(float)); /* Vec<Vec2> :93 */
    oracle_calc_accelerations(m, f, obstacles, n_obstacles, acc);

    for (size_t i = 0; i < n; ++i) {                           /* :245-254 serial */
        ovec2 vel = m->peds.velocity[i];
        float desired_speed = m->peds.desired_speed[i];
        ovec2 vel_prev = vel;                                  /* :250 */
        vel = ov_add(vel, ov_scale(ov(acc[2 * i], acc[2 * i + 1]), 0.1f)); /* :251 */
        vel = ov_clamp_length_max(vel, desired_speed * 1.3f);  /* :252 */
        m->peds.velocity[i] = vel;
        m->peds.position[i] =
            ov_add(m->peds.position[i], ov_scale(ov_add(vel, vel_prev), 0.05f)); /* :253 */
    }
    free(acc);
}

int32_t oracle_get_pedestrian_count(const oracle_model* m) { return (int32_t)m->peds.len; }

void oracle_download(const oracle_model* m, float* pos_xy, uint32_t* destination,
                     float* vel_xy, float* desired_speed)
{
    size_t n = m->peds.len;
    if (pos_xy) memcpy(pos_xy, m->peds.position, n * sizeof(ovec2));
    if (destination) memcpy(destination, m->peds.destination, n * sizeof(uint32_t));
    if (vel_xy) memcpy(vel_xy, m->peds.velocity, n * sizeof(ovec2));
    if (desired_speed) memcpy(desired_speed, m->peds.desired_speed, n * sizeof(float));
}

uint32_t oracle_neighbor_grid_indices(const oracle_model* m, uint32_t* out, uint32_t cap)
{
    uint32_t n = (uint32_t)m->ngi_len;
    if (out) memcpy(out, m->neighbor_grid_indices, (n < cap ? n : cap) * sizeof(uint32_t));
    return n;
}
