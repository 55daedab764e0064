/* pedoni_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement, in plain C, of the reference's per-timestep pedestrian update
 * (pedoni-simulator: models/sfm.rs, neighbor_grid.rs, field.rs, util.rs, lib.rs).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it,
 * and only as the checker / the timed CPU baseline -- never as the shipped path.
 *
 * PARITY PINNING STATUS
 *   pinned   : bilinear (util.rs:157-163, 4 values) and distance_from_line
 *              (util.rs:149-154, 2 values) -- the reference's only asserted tests.
 *   unpinned : everything else on the path ("parity unpinned").  The reference is
 *              Rust; no cargo/rustc exists in this image, so it cannot be run, and
 *              its tests hold no golden vectors for the social-force path.  The
 *              restatement is instead cross-checked against an independently
 *              written NumPy-float32 restatement and closed-form two-body cases.
 *
 * All functions cite the reference file:line they follow (paths relative to
 * /root/reference/pedoni-simulator/src/).
 */
#ifndef PEDONI_ORACLE_H
#define PEDONI_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* field.rs:194-205 `Field` (row-major (y, x) f32 maps) */
typedef struct {
    float unit;
    int32_t rows, cols;
    const float* distance_map;          /* rows*cols */
    const float* const* potential_maps; /* n_maps pointers to rows*cols */
    int32_t n_maps;
} oracle_field;

/* scenario.rs:23-52 ObstacleConfig / WaypointConfig: a segment with a width */
typedef struct { float x0, y0, x1, y1, width; } oracle_segment;

/* lib.rs:108-135 SimulatorOptions (fields the CPU model reads) */
typedef struct {
    float neighbor_grid_unit; /* default 1.4  */
    float field_grid_unit;    /* default 0.25 */
    int32_t use_neighbor_grid;
    int32_t use_distance_map;
} oracle_options;

typedef struct oracle_model oracle_model;

/* ---- util.rs ---------------------------------------------------------- */
float oracle_bilinear(const float* grid, int32_t rows, int32_t cols, float px, float py);
void oracle_sobel_filter(const float* grid, int32_t rows, int32_t cols, float px, float py,
                         float* out_xy);
/* test hook: both of the above at n points */
void oracle_sample_many(const float* grid, int32_t rows, int32_t cols, const float* px,
                        const float* py, float* grad_xy, float* centre, uint32_t n);
void oracle_distance_from_line(float px, float py, const float* line_xyxy, float* out_xy);
void oracle_line_with_width(const float* line_xyxy, float width, float* out_4xy);
int32_t oracle_poisson(uint64_t* rng_state, double lambda);

/* build-owned RNG (the reference never seeds fastrand) */
uint64_t oracle_rng_next(uint64_t* state);
float oracle_rng_f32(uint64_t* state);
double oracle_rng_f64(uint64_t* state);
float oracle_rng_normal_approx(uint64_t* state, float mu, float sigma);

/* restatement of glibc's expf algorithm in double (what the HIP kernels run) */
float oracle_expf_restated(float x);

/* ---- field.rs --------------------------------------------------------- */
float oracle_get_potential(const oracle_field* f, uint32_t waypoint, float px, float py);
float oracle_get_obstacle_distance(const oracle_field* f, float px, float py);
void oracle_get_potential_grad(const oracle_field* f, uint32_t waypoint, float px, float py,
                               float* out_xy);
void oracle_get_obstacle_distance_grad(const oracle_field* f, float px, float py, float* out_xy);
/* field.rs:16-232: FieldBuilder + apply_fmm.  Caller provides rows*cols buffers. */
void oracle_field_shape(float size_x, float size_y, float unit, int32_t* rows, int32_t* cols);
void oracle_field_build(float size_x, float size_y, float unit,
                        const oracle_segment* obstacles, uint32_t n_obstacles,
                        const oracle_segment* waypoints, uint32_t n_waypoints,
                        uint8_t* obstacle_exist, float* distance_map, float* potential_maps);
void oracle_apply_fmm(float* potential, const float* f, int32_t rows, int32_t cols);
void oracle_rasterize_outline(const float* verts_4xy_cells, int32_t rows, int32_t cols,
                              uint8_t* mask);

/* ---- lib.rs spawn logic (Simulator::new :37-52, ::tick :67-85) --------- */
uint32_t oracle_sim_spawn_once(uint64_t* rng, const oracle_segment* origin_line, int32_t count,
                               uint32_t destination, float* pos_xy, uint32_t* dest_out,
                               uint32_t cap);
uint32_t oracle_sim_spawn_periodic(uint64_t* rng, const oracle_segment* origin_line,
                                   double frequency, uint32_t destination, float* pos_xy,
                                   uint32_t* dest_out, uint32_t cap);

/* ---- neighbor_grid.rs ------------------------------------------------- */
void oracle_neighbor_grid_shape(float size_x, float size_y, float unit,
                                int32_t* rows, int32_t* cols);

/* ---- models/sfm.rs ---------------------------------------------------- */
oracle_model* oracle_model_new(const oracle_options* opt, float size_x, float size_y);
void oracle_model_free(oracle_model* m);
void oracle_model_seed(oracle_model* m, uint64_t seed);
void oracle_model_set_threads(oracle_model* m, int32_t n_threads);
/* sfm.rs:48-89.  desired_speed / vel_xy may be NULL: reference behaviour
 * (velocity 0, desired speed drawn inside the model).  Non-NULL = state injection. */
void oracle_spawn_pedestrians(oracle_model* m, const oracle_field* f,
                              const float* pos_xy, const uint32_t* destination, uint32_t n,
                              const float* desired_speed, const float* vel_xy);
/* sfm.rs:91-255 */
void oracle_update_states(oracle_model* m, const oracle_field* f,
                          const oracle_segment* obstacles, uint32_t n_obstacles);
/* sfm.rs:93-241 only (accelerations, no integration) */
/* test hook: the pair force of sfm.rs:130-153 for n independent pairs (acc_xy in/out) */
void oracle_pair_forces(const float* pos_xy, const float* e_xy, const float* pos_i_xy,
                        const float* vel_i_xy, float* acc_xy, uint32_t n);
void oracle_calc_accelerations(const oracle_model* m, const oracle_field* f,
                               const oracle_segment* obstacles, uint32_t n_obstacles,
                               float* acc_xy);
/* sfm.rs:267-269 */
int32_t oracle_get_pedestrian_count(const oracle_model* m);
/* full SoA state out (sfm.rs:257-265 returns pos+destination only; any pointer may be NULL) */
void oracle_download(const oracle_model* m, float* pos_xy, uint32_t* destination,
                     float* vel_xy, float* desired_speed);
/* neighbor_grid_indices (sfm.rs:22,62-74): rows*cols+1 entries; returns length */
uint32_t oracle_neighbor_grid_indices(const oracle_model* m, uint32_t* out, uint32_t cap);

#ifdef __cplusplus
}
#endif
#endif
