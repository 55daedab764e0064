/* pedoni_host.h -- C ABI of the C++ host mirror of the pedoni-simulator crate.
 *
 * The reference's host side is Rust (pedoni-simulator/src/lib.rs, scenario.rs, field.rs);
 * this image has no Rust toolchain, so the host above include/pedoni_hip.h is C++
 * (pedoni_amd/csrc/host/) with the crate's names and argument meaning:
 *   Simulator::new / tick / list_pedestrians, pub field `step`   lib.rs:17-23,27,64,102
 *   SimulatorOptions + Backend                                   lib.rs:108-142
 *   Scenario (serde + TOML)                                      scenario.rs:9-66
 *   Field::from_scenario (FieldBuilder, fast marching)           field.rs:16-232
 * This header flattens that C++ surface for ctypes-driven tests and bench.py.
 * Status codes and error string follow pedoni_hip.h (pedoni_host_last_error()).
 */
#ifndef PEDONI_HOST_H
#define PEDONI_HOST_H

#include <stdint.h>

#include "pedoni_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* lib.rs:138-142 `Backend` + the variant this build adds */
#define PEDONI_BACKEND_CPU 0 /* reference SocialForceModel: not part of this build -> error */
#define PEDONI_BACKEND_GPU 1 /* reference OpenCL model: not part of this build -> error   */
#define PEDONI_BACKEND_HIP 2 /* MI355X backend behind pedoni_hip.h */

/* lib.rs:108-122 `SimulatorOptions` */
typedef struct {
    int32_t backend;
    float neighbor_grid_unit;
    float field_grid_unit;
    int32_t use_neighbor_grid;
    int32_t use_distance_map;
    int32_t gpu_work_size;
    /* build-owned additions */
    int32_t math_mode; /* PEDONI_MATH_* */
    int32_t device;
    uint64_t seed;     /* spawn-position / Poisson / desired-speed streams (reference: unseeded) */
} PedoniSimulatorOptions;

typedef struct PedoniScenario PedoniScenario;
typedef struct PedoniField PedoniField;
typedef struct PedoniSimulator PedoniSimulator;

const char* pedoni_host_last_error(void);
void pedoni_simulator_default_options(PedoniSimulatorOptions* opt); /* lib.rs:124-135 */

/* ---- scenario.rs --------------------------------------------------------------------- */
/* toml::from_str::<Scenario>(text) (pedoni/src/main.rs:55) */
int pedoni_scenario_parse(const char* toml_text, PedoniScenario** out);
void pedoni_scenario_free(PedoniScenario* s);
int pedoni_scenario_size(const PedoniScenario* s, float* size_xy);
/* kind: 0 = waypoints, 1 = obstacles; rows of (x0, y0, x1, y1, width) */
int pedoni_scenario_segments(const PedoniScenario* s, int32_t kind, float* out, uint32_t cap,
                             uint32_t* n);
/* rows of (origin, destination, spawn kind [0 periodic, 1 once], frequency | count) */
int pedoni_scenario_pedestrians(const PedoniScenario* s, double* out, uint32_t cap, uint32_t* n);

/* ---- field.rs ------------------------------------------------------------------------ */
/* Field::from_scenario(&scenario, unit) (field.rs:220-232); potential maps built in
 * parallel threads as upstream does with rayon (field.rs:103-105) */
int pedoni_field_from_scenario(const PedoniScenario* s, float unit, PedoniField** out);
/* same from raw segments (synthetic geometries) */
int pedoni_field_build(float size_x, float size_y, float unit, const PedoniObstacle* obstacles,
                       uint32_t n_obstacles, const PedoniObstacle* waypoints,
                       uint32_t n_waypoints, PedoniField** out);
/* [ext] opt-in, NOT the reference's numbers: the same rasterisation, the maps by the GPU eikonal
 * solver (pedoni_hip_eikonal: block fast iterative method on the first-order upwind scheme)
 * instead of upstream's heap pass (field.rs:118-192), whose result depends on its pop order and
 * cannot be reproduced in parallel.  For start-up time on large fields; the parity tests of the
 * per-step path never use it.  `launches` (may be NULL) receives the relaxation launches run. */
int pedoni_field_build_gpu(float size_x, float size_y, float unit, const PedoniObstacle* obstacles,
                           uint32_t n_obstacles, const PedoniObstacle* waypoints,
                           uint32_t n_waypoints, int32_t device, uint32_t* launches,
                           PedoniField** out);
void pedoni_field_free(PedoniField* f);
int pedoni_field_shape(const PedoniField* f, uint32_t* rows, uint32_t* cols, uint32_t* n_maps,
                       float* unit);
/* borrowed pointers, valid until pedoni_field_free */
const float* pedoni_field_distance_map(const PedoniField* f);
const float* pedoni_field_potential_map(const PedoniField* f, uint32_t waypoint);
const uint8_t* pedoni_field_obstacle_exist(const PedoniField* f);
/* Field::get_potential / get_obstacle_distance (field.rs:235-245), host-side sampling */
int pedoni_field_get_potential(const PedoniField* f, uint32_t waypoint, float x, float y,
                               float* out);
int pedoni_field_get_obstacle_distance(const PedoniField* f, float x, float y, float* out);

/* ---- lib.rs -------------------------------------------------------------------------- */
/* Simulator::new(options, scenario) (lib.rs:27-61): builds the field, creates the model,
 * performs the `once` spawns. */
int pedoni_simulator_new(const PedoniSimulatorOptions* opt, const PedoniScenario* scenario,
                         PedoniSimulator** out);
void pedoni_simulator_free(PedoniSimulator* sim);
/* Simulator::tick (lib.rs:64-100) */
int pedoni_simulator_tick(PedoniSimulator* sim, PedoniStepMetrics* metrics);
/* build-owned: `n` ticks with the periodic spawners evaluated on the device (bit-identical
 * crowd to n calls of pedoni_simulator_tick, no per-tick host work); metrics of the batch */
int pedoni_simulator_tick_n(PedoniSimulator* sim, uint32_t n, PedoniStepMetrics* metrics);
/* pub field `step` (lib.rs:22) */
int pedoni_simulator_step(const PedoniSimulator* sim, int32_t* step);
/* Simulator::list_pedestrians (lib.rs:102-104) */
int pedoni_simulator_list_pedestrians(PedoniSimulator* sim, PedoniPedestrian* out, uint32_t cap,
                                      uint32_t* n);
/* build-owned checkpoint / resume (upstream has none): step counter, both
 * generator states and the model's full SoA state; a resumed run continues bit for bit like
 * the uninterrupted one.  `resume` replaces pedoni_simulator_new (no `once` spawns) and
 * fails when the file was saved with another scenario or other simulator options. */
int pedoni_simulator_save_checkpoint(PedoniSimulator* sim, const char* path);
int pedoni_simulator_resume(const PedoniSimulatorOptions* opt, const PedoniScenario* scenario,
                            const char* path, PedoniSimulator** out);
/* pub fields `model`, `field` (lib.rs:20-21), borrowed */
PedoniModel* pedoni_simulator_model(PedoniSimulator* sim);
const PedoniField* pedoni_simulator_field(const PedoniSimulator* sim);

#ifdef __cplusplus
}
#endif
#endif
