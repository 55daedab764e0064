// simulator.cpp -- Simulator / SocialForceModelHip (pedoni-simulator/src/lib.rs,
// models/mod.rs) over the C-ABI of include/pedoni_hip.h.
#include "pedoni_host.hpp"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <stdexcept>

namespace pedoni_host {

// ---- build-owned RNG: WyRand step, 24-bit f32, 53-bit f64 (same spec as the HIP library's
// desired-speed generator; the reference's fastrand global is never seeded) ----------------
uint64_t Rng::next()
{
    state += 0xa0761d6478bd642fULL;
    __uint128_t t = (__uint128_t)state * (__uint128_t)(state ^ 0xe7037ed1a0b428dbULL);
    return (uint64_t)(t >> 64) ^ (uint64_t)t;
}
float Rng::f32() { return (float)(next() >> 40) * 0x1.0p-24f; }
double Rng::f64() { return (double)(next() >> 11) * 0x1.0p-53; }
int32_t Rng::poisson(double lambda) // util.rs:78-89
{
    int32_t y = 0;
    double x = f64();
    const double exp_lambda = std::exp(-lambda);
    while (x >= exp_lambda) {
        x *= f64();
        y += 1;
    }
    return y;
}

namespace {

void check(int rc, const char* what)
{
    if (rc != PEDONI_OK)
        throw std::runtime_error(std::string(what) + ": " + pedoni_hip_last_error());
}

// glam 0.29 Vec2::lerp: self * (1 - s) + rhs * s   (lib.rs:43,76)
Vec2 lerp(Vec2 a, Vec2 b, float s)
{
    const float k = 1.0f - s;
    return Vec2{a.x * k + b.x * s, a.y * k + b.y * s};
}

} // namespace

// ---- SocialForceModelHip -------------------------------------------------------------------
SocialForceModelHip::SocialForceModelHip(const SimulatorOptions& options, const Scenario& scenario,
                                         const Field& field)
{
    PedoniOptions o;
    pedoni_hip_default_options(&o);
    o.neighbor_grid_unit = options.neighbor_grid_unit;
    o.field_grid_unit = options.field_grid_unit;
    o.use_neighbor_grid = options.use_neighbor_grid;
    o.use_distance_map = options.use_distance_map;
    // lib.rs:132 default 64 is an OpenCL local size; any multiple of 64 is honoured here
    o.gpu_work_size = (options.gpu_work_size % 64 == 0 && options.gpu_work_size <= 1024)
                          ? (int32_t)options.gpu_work_size : 0;
    o.math_mode = options.math_mode;
    o.seed = options.seed ^ 0x5eedULL; // desired-speed stream, distinct from the spawn stream

    std::vector<const float*> maps;
    for (const auto& pm : field.potential_maps) maps.push_back(pm.data());
    std::vector<PedoniObstacle> obs;
    for (const ObstacleConfig& c : scenario.obstacles)
        obs.push_back(PedoniObstacle{c.line[0].x, c.line[0].y, c.line[1].x, c.line[1].y, c.width});
    check(pedoni_hip_create(&o, scenario.field.size.x, scenario.field.size.y,
                            field.distance_map.data(), maps.data(), (uint32_t)maps.size(),
                            (uint32_t)field.rows, (uint32_t)field.cols, field.unit, obs.data(),
                            (uint32_t)obs.size(), options.device, &model_),
          "SocialForceModelHip::new");
}

SocialForceModelHip::~SocialForceModelHip() { pedoni_hip_destroy(model_); }

void SocialForceModelHip::spawn_pedestrians(const Field&, std::vector<Pedestrian> new_pedestrians)
{
    std::vector<PedoniPedestrian> peds(new_pedestrians.size());
    for (size_t i = 0; i < peds.size(); ++i)
        peds[i] = PedoniPedestrian{new_pedestrians[i].pos.x, new_pedestrians[i].pos.y,
                                   (uint64_t)new_pedestrians[i].destination};
    check(pedoni_hip_spawn_pedestrians(model_, peds.data(), (uint32_t)peds.size()),
          "spawn_pedestrians");
}

void SocialForceModelHip::update_states(const Scenario&, const Field&)
{
    // device time of the force + integrate kernel, from the library's hipEvent pair
    // (diagnostic.rs:45-50 time_calc_state_kernel; upstream's OpenCL model reads its event's
    // start / end and then drops them, sfm_gpu.rs:234-236, lib.rs:98)
    check(pedoni_hip_profile(model_, 1 << PEDONI_K_FORCE), "update_states");
    check(pedoni_hip_update_states(model_), "update_states");
    PedoniKernelTimes t{};
    check(pedoni_hip_kernel_times(model_, &t, /*reset=*/1), "update_states");
    check(pedoni_hip_profile(model_, 0), "update_states");
    last_kernel_s_ = t.launches[PEDONI_K_FORCE] ? std::optional<double>(t.total_ms[PEDONI_K_FORCE] * 1e-3)
                                                : std::nullopt;   // no agents: nothing was launched
}

std::vector<Pedestrian> SocialForceModelHip::list_pedestrians() const
{
    uint32_t n = 0;
    check(pedoni_hip_list_pedestrians(model_, nullptr, 0, &n), "list_pedestrians");
    std::vector<PedoniPedestrian> raw(n);
    check(pedoni_hip_list_pedestrians(model_, raw.data(), n, &n), "list_pedestrians");
    std::vector<Pedestrian> out(n);
    for (uint32_t i = 0; i < n; ++i) {
        out[i].pos = Vec2{raw[i].x, raw[i].y};
        out[i].destination = (size_t)raw[i].destination;
    }
    return out;
}

int32_t SocialForceModelHip::get_pedestrian_count() const
{
    int32_t c = 0;
    check(pedoni_hip_get_pedestrian_count(model_, &c), "get_pedestrian_count");
    return c;
}

// ---- Simulator -------------------------------------------------------------------------------
void Simulator::init_field_and_model()
{
    rng_.state = options.seed;
    for (const PedestrianConfig& p : scenario.pedestrians)
        if (p.origin >= scenario.waypoints.size() || p.destination >= scenario.waypoints.size())
            throw std::runtime_error("Simulator::new: pedestrian origin/destination out of range");

    field = Field::from_scenario(scenario, options.field_grid_unit);        // lib.rs:30

    switch (options.backend) {                                              // lib.rs:32-35
    case Backend::Hip:
        model = std::make_unique<SocialForceModelHip>(options, scenario, field);
        break;
    case Backend::Cpu:
        throw std::runtime_error("Backend::Cpu (the reference's SocialForceModel) is not part of "
                                 "this build; use Backend::Hip");
    case Backend::Gpu:
        throw std::runtime_error("Backend::Gpu (the reference's OpenCL model) is not part of "
                                 "this build; use Backend::Hip");
    }
}

Simulator::Simulator(SimulatorOptions options_, Scenario scenario_)
    : options(options_), scenario(std::move(scenario_))
{
    init_field_and_model();
    std::vector<Pedestrian> new_pedestrians;                                // lib.rs:37-51
    for (const PedestrianConfig& p : scenario.pedestrians) {
        if (p.spawn.kind != PedestrianSpawnConfig::Once) continue;
        const Vec2 p1 = scenario.waypoints[p.origin].line[0], p2 = scenario.waypoints[p.origin].line[1];
        for (int32_t k = 0; k < p.spawn.count; ++k)
            new_pedestrians.push_back(Pedestrian{lerp(p1, p2, rng_.f32()), p.destination});
    }
    model->spawn_pedestrians(field, std::move(new_pedestrians));           // lib.rs:52
}

Simulator::Simulator(SimulatorOptions options_, Scenario scenario_, ResumeTag)
    : options(options_), scenario(std::move(scenario_))
{
    init_field_and_model();
}

// ---- checkpoint / resume (build-owned) ---------------------------------------------------------
namespace {

constexpr char CKPT_MAGIC[8] = {'P', 'E', 'D', 'O', 'N', 'I', 'C', 'K'};
constexpr uint32_t CKPT_VERSION = 1;

struct CheckpointHeader {          // little-endian, 80 bytes, followed by the four arrays
    char magic[8];
    uint32_t version;
    int32_t step;
    uint64_t n_agents;
    uint64_t rng_position, rng_speed;
    float size_x, size_y, neighbor_grid_unit, field_grid_unit;
    uint32_t n_waypoints, n_obstacles;
    uint32_t use_neighbor_grid, use_distance_map;
    uint32_t math_mode, reserved;
};
static_assert(sizeof(CheckpointHeader) == 80, "checkpoint header layout");

} // namespace

void Simulator::save_checkpoint(const std::string& path)
{
    auto* hip = dynamic_cast<SocialForceModelHip*>(model.get());
    if (!hip) throw std::runtime_error("Simulator::save_checkpoint needs the Hip backend");
    take_spawning_back();   // both generator states back on the host side of the boundary
    CheckpointHeader h{};
    std::memcpy(h.magic, CKPT_MAGIC, sizeof h.magic);
    h.version = CKPT_VERSION;
    h.step = step;
    h.rng_position = rng_.state;
    check(pedoni_hip_get_spawn_rng(hip->handle(), nullptr, &h.rng_speed), "save_checkpoint");
    h.size_x = scenario.field.size.x;
    h.size_y = scenario.field.size.y;
    h.neighbor_grid_unit = options.neighbor_grid_unit;
    h.field_grid_unit = options.field_grid_unit;
    h.n_waypoints = (uint32_t)scenario.waypoints.size();
    h.n_obstacles = (uint32_t)scenario.obstacles.size();
    h.use_neighbor_grid = options.use_neighbor_grid;
    h.use_distance_map = options.use_distance_map;
    h.math_mode = (uint32_t)options.math_mode;

    uint32_t n = 0;
    check(pedoni_hip_download(hip->handle(), nullptr, nullptr, nullptr, nullptr, 0, &n), "save_checkpoint");
    std::vector<float> pos(2 * (size_t)n), vel(2 * (size_t)n), v0(n);
    std::vector<uint32_t> dest(n);
    check(pedoni_hip_download(hip->handle(), pos.data(), dest.data(), vel.data(), v0.data(), n, &n),
          "save_checkpoint");
    h.n_agents = n;

    const std::string tmp = path + ".part";   // never leave a half-written checkpoint behind
    FILE* f = std::fopen(tmp.c_str(), "wb");
    if (!f) throw std::runtime_error("save_checkpoint: cannot create '" + tmp + "'");
    bool ok = std::fwrite(&h, sizeof h, 1, f) == 1;
    ok = ok && (n == 0 || (std::fwrite(pos.data(), sizeof(float), pos.size(), f) == pos.size() &&
                           std::fwrite(vel.data(), sizeof(float), vel.size(), f) == vel.size() &&
                           std::fwrite(v0.data(), sizeof(float), v0.size(), f) == v0.size() &&
                           std::fwrite(dest.data(), sizeof(uint32_t), dest.size(), f) == dest.size()));
    ok = (std::fclose(f) == 0) && ok;
    if (!ok || std::rename(tmp.c_str(), path.c_str()) != 0) {
        std::remove(tmp.c_str());
        throw std::runtime_error("save_checkpoint: cannot write '" + path + "'");
    }
}

std::unique_ptr<Simulator> Simulator::resume(SimulatorOptions options_, Scenario scenario_,
                                             const std::string& path)
{
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("Simulator::resume: cannot read '" + path + "'");
    struct Closer { FILE* f; ~Closer() { std::fclose(f); } } closer{f};
    CheckpointHeader h{};
    if (std::fread(&h, sizeof h, 1, f) != 1 || std::memcmp(h.magic, CKPT_MAGIC, sizeof h.magic) != 0)
        throw std::runtime_error("Simulator::resume: '" + path + "' is not a pedoni checkpoint");
    if (h.version != CKPT_VERSION)
        throw std::runtime_error("Simulator::resume: unsupported checkpoint version");
    // the agents' state only means something in the world it was saved in
    if (h.size_x != scenario_.field.size.x || h.size_y != scenario_.field.size.y ||
        h.n_waypoints != scenario_.waypoints.size() || h.n_obstacles != scenario_.obstacles.size() ||
        h.neighbor_grid_unit != options_.neighbor_grid_unit || h.field_grid_unit != options_.field_grid_unit ||
        h.use_neighbor_grid != (uint32_t)options_.use_neighbor_grid ||
        h.use_distance_map != (uint32_t)options_.use_distance_map)
        throw std::runtime_error("Simulator::resume: the checkpoint was saved with another scenario "
                                 "or other simulator options");
    if (h.n_agents > 0xfffffff0ull) throw std::runtime_error("Simulator::resume: corrupt agent count");
    const size_t n = (size_t)h.n_agents;
    std::vector<float> pos(2 * n), vel(2 * n), v0(n);
    std::vector<uint32_t> dest(n);
    if (n && (std::fread(pos.data(), sizeof(float), pos.size(), f) != pos.size() ||
              std::fread(vel.data(), sizeof(float), vel.size(), f) != vel.size() ||
              std::fread(v0.data(), sizeof(float), v0.size(), f) != v0.size() ||
              std::fread(dest.data(), sizeof(uint32_t), dest.size(), f) != dest.size()))
        throw std::runtime_error("Simulator::resume: '" + path + "' is truncated");

    std::unique_ptr<Simulator> sim(new Simulator(options_, std::move(scenario_), ResumeTag{}));
    auto* hip = dynamic_cast<SocialForceModelHip*>(sim->model.get());
    if (!hip) throw std::runtime_error("Simulator::resume needs the Hip backend");
    // model order = the order of the last sort pass: appended in that order, the next pass's
    // stable sort reproduces the uninterrupted run's order
    check(pedoni_hip_append(hip->handle(), pos.data(), dest.data(), v0.data(), vel.data(), (uint32_t)n),
          "Simulator::resume");
    // bin them now (the pass the next tick runs anyway: a second pass over unmoved agents
    // changes nothing), so that either tick() or tick_n() may follow
    check(pedoni_hip_sort_despawn(hip->handle()), "Simulator::resume");
    check(pedoni_hip_set_speed_rng(hip->handle(), h.rng_speed), "Simulator::resume");
    sim->rng_.state = h.rng_position;
    sim->step = h.step;
    return sim;
}

void Simulator::hand_spawning_to_device()
{
    auto* hip = dynamic_cast<SocialForceModelHip*>(model.get());
    if (!hip) throw std::runtime_error("Simulator::tick_n needs the Hip backend");
    std::vector<PedoniSpawner> sp;
    double per_tick = 0.0;
    for (const PedestrianConfig& p : scenario.pedestrians) {                // lib.rs:70-72, same order
        if (p.spawn.kind != PedestrianSpawnConfig::Periodic) continue;
        const Vec2 p1 = scenario.waypoints[p.origin].line[0], p2 = scenario.waypoints[p.origin].line[1];
        sp.push_back(PedoniSpawner{p1.x, p1.y, p2.x, p2.y, (uint32_t)p.destination, 0u, p.spawn.frequency});
        per_tick += p.spawn.frequency / 10.0;
    }
    // Poisson mean + 12 sigma + slack: exceeding it is reported, never silently dropped
    const uint32_t cap = (uint32_t)(per_tick + 12.0 * std::sqrt(per_tick + 1.0) + 64.0);
    check(pedoni_hip_set_spawners(hip->handle(), sp.data(), (uint32_t)sp.size(), rng_.state, cap),
          "Simulator::tick_n");
    device_spawners_ = !sp.empty();
}

void Simulator::take_spawning_back()
{
    auto* hip = dynamic_cast<SocialForceModelHip*>(model.get());
    if (!hip || !device_spawners_) return;
    uint64_t pos_state = rng_.state;
    check(pedoni_hip_get_spawn_rng(hip->handle(), &pos_state, nullptr), "Simulator::tick");
    rng_.state = pos_state;
    check(pedoni_hip_set_spawners(hip->handle(), nullptr, 0, 0, 0), "Simulator::tick");
    device_spawners_ = false;
}

StepMetrics Simulator::tick_n(uint32_t n)
{
    using clk = std::chrono::steady_clock;
    auto* hip = dynamic_cast<SocialForceModelHip*>(model.get());
    if (!hip) throw std::runtime_error("Simulator::tick_n needs the Hip backend");
    if (!options.use_neighbor_grid) {
        // the brute-force option path (sfm.rs:78-88,157-185) keeps spawning on the host
        StepMetrics total;
        for (uint32_t k = 0; k < n; ++k) {
            StepMetrics m = tick();
            total.active_ped_count = m.active_ped_count;
            total.time_spawn += m.time_spawn;
            total.time_calc_state += m.time_calc_state;
        }
        return total;
    }
    if (!device_spawners_) hand_spawning_to_device();
    const auto t0 = clk::now();
    check(pedoni_hip_tick_n(hip->handle(), n), "Simulator::tick_n");
    check(pedoni_hip_synchronize(hip->handle()), "Simulator::tick_n");
    step += (int32_t)n;
    StepMetrics m;
    m.active_ped_count = model->get_pedestrian_count();
    m.time_spawn = 0.0;
    m.time_calc_state = std::chrono::duration<double>(clk::now() - t0).count();
    return m;
}

StepMetrics Simulator::tick()
{
    using clk = std::chrono::steady_clock;
    take_spawning_back();   // (a previous tick_n left the spawn streams on the device)
    step += 1;                                                              // lib.rs:65

    auto instant = clk::now();                                              // lib.rs:68
    std::vector<Pedestrian> new_pedestrians;
    for (const PedestrianConfig& p : scenario.pedestrians) {                // lib.rs:70-84
        if (p.spawn.kind != PedestrianSpawnConfig::Periodic) continue;
        const Vec2 p1 = scenario.waypoints[p.origin].line[0], p2 = scenario.waypoints[p.origin].line[1];
        const int32_t count = rng_.poisson(p.spawn.frequency / 10.0);
        for (int32_t k = 0; k < count; ++k)
            new_pedestrians.push_back(Pedestrian{lerp(p1, p2, rng_.f32()), p.destination});
    }
    model->spawn_pedestrians(field, std::move(new_pedestrians));           // lib.rs:85
    auto* hip = dynamic_cast<SocialForceModelHip*>(model.get());
    if (hip) pedoni_hip_synchronize(hip->handle());  // the launches are asynchronous
    const double time_spawn = std::chrono::duration<double>(clk::now() - instant).count();

    instant = clk::now();                                                   // lib.rs:89
    model->update_states(scenario, field);                                  // lib.rs:90
    if (hip) pedoni_hip_synchronize(hip->handle());
    const double time_calc_state = std::chrono::duration<double>(clk::now() - instant).count();

    StepMetrics m;                                                          // lib.rs:94-99
    m.active_ped_count = model->get_pedestrian_count();
    m.time_spawn = time_spawn;
    m.time_calc_state = time_calc_state;
    // upstream leaves this None (lib.rs:98); SURVEY 8(f) rank 4 asks the backend to fill it
    m.time_calc_state_kernel = hip ? hip->last_kernel_seconds() : std::nullopt;
    return m;
}

} // namespace pedoni_host
