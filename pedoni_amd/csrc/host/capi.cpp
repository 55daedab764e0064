// capi.cpp -- include/pedoni_host.h: the C++ host mirror flattened to a C ABI.
#include "pedoni_host.h"
#include "pedoni_host.hpp"

#include <algorithm>
#include <cstring>
#include <string>

using namespace pedoni_host;

struct PedoniScenario { Scenario sc; };
struct PedoniField { Field f; };
struct PedoniSimulator { std::unique_ptr<Simulator> sim; PedoniField field_view; };

namespace {
thread_local std::string g_err;
int fail(int code, const std::string& msg) { g_err = msg; return code; }

template <typename F> int guarded(F&& body)
{
    try {
        return body();
    } catch (const std::exception& e) {
        return fail(PEDONI_E_INVALID, e.what());
    } catch (...) {
        return fail(PEDONI_E_INVALID, "unknown C++ exception");
    }
}
} // namespace

extern "C" {

const char* pedoni_host_last_error(void) { return g_err.c_str(); }

void pedoni_simulator_default_options(PedoniSimulatorOptions* o)
{
    if (!o) return;
    std::memset(o, 0, sizeof(*o));
    o->backend = PEDONI_BACKEND_HIP;
    o->neighbor_grid_unit = 1.4f;
    o->field_grid_unit = 0.25f;
    o->use_neighbor_grid = 1;
    o->use_distance_map = 1;
    o->gpu_work_size = 64;
    o->math_mode = PEDONI_MATH_EXACT;
    o->device = 0;
    o->seed = 12345;
}

int pedoni_scenario_parse(const char* text, PedoniScenario** out)
{
    if (!text || !out) return fail(PEDONI_E_INVALID, "null argument");
    return guarded([&] {
        auto* s = new PedoniScenario{Scenario::from_toml(text)};
        *out = s;
        return PEDONI_OK;
    });
}

void pedoni_scenario_free(PedoniScenario* s) { delete s; }

int pedoni_scenario_size(const PedoniScenario* s, float* size_xy)
{
    if (!s || !size_xy) return fail(PEDONI_E_INVALID, "null argument");
    size_xy[0] = s->sc.field.size.x;
    size_xy[1] = s->sc.field.size.y;
    return PEDONI_OK;
}

int pedoni_scenario_segments(const PedoniScenario* s, int32_t kind, float* out, uint32_t cap, uint32_t* n)
{
    if (!s || !n) return fail(PEDONI_E_INVALID, "null argument");
    auto emit = [&](uint32_t i, const Vec2 line[2], float w) {
        if (out && i < cap) {
            out[5 * i] = line[0].x; out[5 * i + 1] = line[0].y;
            out[5 * i + 2] = line[1].x; out[5 * i + 3] = line[1].y; out[5 * i + 4] = w;
        }
    };
    if (kind == 0) {
        *n = (uint32_t)s->sc.waypoints.size();
        for (uint32_t i = 0; i < *n; ++i) emit(i, s->sc.waypoints[i].line, s->sc.waypoints[i].width);
    } else if (kind == 1) {
        *n = (uint32_t)s->sc.obstacles.size();
        for (uint32_t i = 0; i < *n; ++i) emit(i, s->sc.obstacles[i].line, s->sc.obstacles[i].width);
    } else {
        return fail(PEDONI_E_INVALID, "kind must be 0 (waypoints) or 1 (obstacles)");
    }
    return PEDONI_OK;
}

int pedoni_scenario_pedestrians(const PedoniScenario* s, double* out, uint32_t cap, uint32_t* n)
{
    if (!s || !n) return fail(PEDONI_E_INVALID, "null argument");
    *n = (uint32_t)s->sc.pedestrians.size();
    for (uint32_t i = 0; i < *n && out && i < cap; ++i) {
        const PedestrianConfig& p = s->sc.pedestrians[i];
        out[4 * i] = (double)p.origin;
        out[4 * i + 1] = (double)p.destination;
        out[4 * i + 2] = p.spawn.kind == PedestrianSpawnConfig::Once ? 1.0 : 0.0;
        out[4 * i + 3] = p.spawn.kind == PedestrianSpawnConfig::Once ? (double)p.spawn.count : p.spawn.frequency;
    }
    return PEDONI_OK;
}

int pedoni_field_from_scenario(const PedoniScenario* s, float unit, PedoniField** out)
{
    if (!s || !out) return fail(PEDONI_E_INVALID, "null argument");
    return guarded([&] {
        *out = new PedoniField{Field::from_scenario(s->sc, unit)};
        return PEDONI_OK;
    });
}

int pedoni_field_build(float size_x, float size_y, float unit, const PedoniObstacle* obstacles,
                       uint32_t n_obstacles, const PedoniObstacle* waypoints, uint32_t n_waypoints,
                       PedoniField** out)
{
    if (!out || (n_obstacles && !obstacles) || (n_waypoints && !waypoints))
        return fail(PEDONI_E_INVALID, "null argument");
    return guarded([&] {
        Scenario sc;
        sc.field.size = Vec2{size_x, size_y};
        for (uint32_t i = 0; i < n_obstacles; ++i) {
            ObstacleConfig c;
            c.line[0] = Vec2{obstacles[i].x0, obstacles[i].y0};
            c.line[1] = Vec2{obstacles[i].x1, obstacles[i].y1};
            c.width = obstacles[i].width;
            sc.obstacles.push_back(c);
        }
        for (uint32_t i = 0; i < n_waypoints; ++i) {
            WaypointConfig c;
            c.line[0] = Vec2{waypoints[i].x0, waypoints[i].y0};
            c.line[1] = Vec2{waypoints[i].x1, waypoints[i].y1};
            c.width = waypoints[i].width;
            sc.waypoints.push_back(c);
        }
        *out = new PedoniField{Field::from_scenario(sc, unit)};
        return PEDONI_OK;
    });
}

int pedoni_field_build_gpu(float size_x, float size_y, float unit, const PedoniObstacle* obstacles,
                           uint32_t n_obstacles, const PedoniObstacle* waypoints, uint32_t n_waypoints,
                           int32_t device, uint32_t* launches, PedoniField** out)
{
    if (!out || (n_obstacles && !obstacles) || (n_waypoints && !waypoints))
        return fail(PEDONI_E_INVALID, "null argument");
    return guarded([&] {
        Scenario sc;
        sc.field.size = Vec2{size_x, size_y};
        for (uint32_t i = 0; i < n_obstacles; ++i) {
            ObstacleConfig c;
            c.line[0] = Vec2{obstacles[i].x0, obstacles[i].y0};
            c.line[1] = Vec2{obstacles[i].x1, obstacles[i].y1};
            c.width = obstacles[i].width;
            sc.obstacles.push_back(c);
        }
        for (uint32_t i = 0; i < n_waypoints; ++i) {
            WaypointConfig c;
            c.line[0] = Vec2{waypoints[i].x0, waypoints[i].y0};
            c.line[1] = Vec2{waypoints[i].x1, waypoints[i].y1};
            c.width = waypoints[i].width;
            sc.waypoints.push_back(c);
        }
        *out = new PedoniField{Field::from_scenario_gpu(sc, unit, device, launches)};
        return PEDONI_OK;
    });
}

void pedoni_field_free(PedoniField* f) { delete f; }

int pedoni_field_shape(const PedoniField* f, uint32_t* rows, uint32_t* cols, uint32_t* n_maps, float* unit)
{
    if (!f) return fail(PEDONI_E_INVALID, "null field");
    if (rows) *rows = (uint32_t)f->f.rows;
    if (cols) *cols = (uint32_t)f->f.cols;
    if (n_maps) *n_maps = (uint32_t)f->f.potential_maps.size();
    if (unit) *unit = f->f.unit;
    return PEDONI_OK;
}

const float* pedoni_field_distance_map(const PedoniField* f) { return f ? f->f.distance_map.data() : nullptr; }
const float* pedoni_field_potential_map(const PedoniField* f, uint32_t w)
{
    return (f && w < f->f.potential_maps.size()) ? f->f.potential_maps[w].data() : nullptr;
}
const uint8_t* pedoni_field_obstacle_exist(const PedoniField* f) { return f ? f->f.obstacle_exist.data() : nullptr; }

int pedoni_field_get_potential(const PedoniField* f, uint32_t w, float x, float y, float* out)
{
    if (!f || !out || w >= f->f.potential_maps.size()) return fail(PEDONI_E_INVALID, "bad argument");
    *out = f->f.get_potential(w, Vec2{x, y});
    return PEDONI_OK;
}

int pedoni_field_get_obstacle_distance(const PedoniField* f, float x, float y, float* out)
{
    if (!f || !out) return fail(PEDONI_E_INVALID, "bad argument");
    *out = f->f.get_obstacle_distance(Vec2{x, y});
    return PEDONI_OK;
}

namespace {
SimulatorOptions to_options(const PedoniSimulatorOptions* o)
{
    SimulatorOptions so;
    so.backend = o->backend == PEDONI_BACKEND_CPU ? Backend::Cpu
               : o->backend == PEDONI_BACKEND_GPU ? Backend::Gpu : Backend::Hip;
    so.neighbor_grid_unit = o->neighbor_grid_unit;
    so.field_grid_unit = o->field_grid_unit;
    so.use_neighbor_grid = o->use_neighbor_grid != 0;
    so.use_distance_map = o->use_distance_map != 0;
    so.gpu_work_size = (size_t)std::max(o->gpu_work_size, 0);
    so.math_mode = o->math_mode;
    so.device = o->device;
    so.seed = o->seed;
    return so;
}
} // namespace

int pedoni_simulator_new(const PedoniSimulatorOptions* o, const PedoniScenario* scenario, PedoniSimulator** out)
{
    if (!o || !scenario || !out) return fail(PEDONI_E_INVALID, "null argument");
    return guarded([&] {
        auto* s = new PedoniSimulator();
        try {
            s->sim = std::make_unique<Simulator>(to_options(o), scenario->sc);
        } catch (...) {
            delete s;
            throw;
        }
        *out = s;
        return PEDONI_OK;
    });
}

int pedoni_simulator_resume(const PedoniSimulatorOptions* o, const PedoniScenario* scenario,
                            const char* path, PedoniSimulator** out)
{
    if (!o || !scenario || !path || !out) return fail(PEDONI_E_INVALID, "null argument");
    return guarded([&] {
        auto* s = new PedoniSimulator();
        try {
            s->sim = Simulator::resume(to_options(o), scenario->sc, path);
        } catch (...) {
            delete s;
            throw;
        }
        *out = s;
        return PEDONI_OK;
    });
}

int pedoni_simulator_save_checkpoint(PedoniSimulator* sim, const char* path)
{
    if (!sim || !path) return fail(PEDONI_E_INVALID, "null argument");
    return guarded([&] {
        sim->sim->save_checkpoint(path);
        return PEDONI_OK;
    });
}

void pedoni_simulator_free(PedoniSimulator* sim) { delete sim; }

int pedoni_simulator_tick(PedoniSimulator* sim, PedoniStepMetrics* m)
{
    if (!sim) return fail(PEDONI_E_INVALID, "null simulator");
    return guarded([&] {
        StepMetrics sm = sim->sim->tick();
        if (m) {
            m->active_ped_count = sm.active_ped_count;
            m->time_spawn = sm.time_spawn;
            m->time_calc_state = sm.time_calc_state;
            m->time_calc_state_kernel = sm.time_calc_state_kernel.value_or(-1.0);
        }
        return PEDONI_OK;
    });
}

int pedoni_simulator_tick_n(PedoniSimulator* sim, uint32_t n, PedoniStepMetrics* m)
{
    if (!sim) return fail(PEDONI_E_INVALID, "null simulator");
    return guarded([&] {
        StepMetrics sm = sim->sim->tick_n(n);
        if (m) {
            m->active_ped_count = sm.active_ped_count;
            m->time_spawn = sm.time_spawn;
            m->time_calc_state = sm.time_calc_state;
            m->time_calc_state_kernel = sm.time_calc_state_kernel.value_or(-1.0);
        }
        return PEDONI_OK;
    });
}

int pedoni_simulator_step(const PedoniSimulator* sim, int32_t* step)
{
    if (!sim || !step) return fail(PEDONI_E_INVALID, "null argument");
    *step = sim->sim->step;
    return PEDONI_OK;
}

int pedoni_simulator_list_pedestrians(PedoniSimulator* sim, PedoniPedestrian* out, uint32_t cap, uint32_t* n)
{
    if (!sim || !n) return fail(PEDONI_E_INVALID, "null argument");
    return guarded([&] {
        auto peds = sim->sim->list_pedestrians();
        *n = (uint32_t)peds.size();
        for (uint32_t i = 0; i < *n && out && i < cap; ++i)
            out[i] = PedoniPedestrian{peds[i].pos.x, peds[i].pos.y, (uint64_t)peds[i].destination};
        return PEDONI_OK;
    });
}

PedoniModel* pedoni_simulator_model(PedoniSimulator* sim)
{
    if (!sim) return nullptr;
    auto* hip = dynamic_cast<SocialForceModelHip*>(sim->sim->model.get());
    return hip ? hip->handle() : nullptr;
}

const PedoniField* pedoni_simulator_field(const PedoniSimulator* sim)
{
    // Field is owned by the Simulator; expose it through the same opaque layout
    return sim ? reinterpret_cast<const PedoniField*>(&sim->sim->field) : nullptr;
}

} // extern "C"
