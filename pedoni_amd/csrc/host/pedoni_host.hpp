// pedoni_host.hpp -- C++ mirror of the pedoni-simulator crate's host-side API.
//
// Same names and argument meaning as the reference (pedoni-simulator/src/):
//   scenario.rs  Scenario, FieldConfig, ObstacleConfig, WaypointConfig, PedestrianConfig,
//                PedestrianSpawnConfig
//   field.rs     Field (+ FieldBuilder / apply_fmm behind Field::from_scenario)
//   models/mod.rs  trait PedestrianModel, struct Pedestrian
//   lib.rs       Simulator, SimulatorOptions, Backend
//   diagnostic.rs  StepMetrics
// The reference is Rust; this image has no Rust toolchain, so the host above the C-ABI
// (include/pedoni_hip.h) is C++.  Failures throw std::runtime_error where the reference
// panics.
#pragma once

#include <cstdint>
#include <memory>
#include <optional>
#include <string>
#include <vector>

#include "pedoni_hip.h"

namespace pedoni_host {

struct Vec2 {
    float x = 0.0f, y = 0.0f;
};

// ---- scenario.rs ------------------------------------------------------------------------
struct FieldConfig {            // scenario.rs:17-20
    Vec2 size;
};
struct ObstacleConfig {         // scenario.rs:22-35
    Vec2 line[2];
    float width = 1.0f;
};
struct WaypointConfig {         // scenario.rs:37-52
    Vec2 line[2];
    float width = 1.0f;
};
struct PedestrianSpawnConfig {  // scenario.rs:60-66, internally tagged on `kind`
    enum Kind { Periodic, Once } kind = Periodic;
    double frequency = 0.0;     // Periodic
    int32_t count = 0;          // Once
};
struct PedestrianConfig {       // scenario.rs:54-58
    size_t origin = 0;
    size_t destination = 0;
    PedestrianSpawnConfig spawn;
};
struct Scenario {               // scenario.rs:9-15
    FieldConfig field;
    std::vector<WaypointConfig> waypoints;
    std::vector<ObstacleConfig> obstacles;
    std::vector<PedestrianConfig> pedestrians;

    // toml::from_str::<Scenario>() (pedoni/src/main.rs:55): unknown keys ignored, integers
    // coerce to floats, `width` defaults to 1.0, all four top-level keys required.
    static Scenario from_toml(const std::string& text);
};

// ---- field.rs ---------------------------------------------------------------------------
struct Field {                  // field.rs:194-205
    float unit = 0.5f;
    size_t rows = 0, cols = 0;  // shape (y, x)
    std::vector<uint8_t> obstacle_exist;
    std::vector<float> distance_map;
    std::vector<std::vector<float>> potential_maps;

    static Field from_scenario(const Scenario& scenario, float unit); // field.rs:220-232
    // build-owned, opt-in: the maps by the GPU eikonal solver (NOT upstream's heap-order numbers)
    static Field from_scenario_gpu(const Scenario& scenario, float unit, int device = 0, uint32_t* launches = nullptr);
    float get_potential(size_t waypoint_id, Vec2 position) const;     // field.rs:235-239
    float get_obstacle_distance(Vec2 position) const;                 // field.rs:242-245
    Vec2 get_potential_grad(size_t waypoint_id, Vec2 position) const; // field.rs:248-252
    Vec2 get_obstacle_distance_grad(Vec2 position) const;             // field.rs:255-258

private:
    static Field build(const Scenario& scenario, float unit, bool gpu, int device, uint32_t* launches);
};

namespace util {                // util.rs
float bilinear(const std::vector<float>& grid, size_t rows, size_t cols, Vec2 pos); // :44-58
Vec2 sobel_filter(const std::vector<float>& grid, size_t rows, size_t cols, Vec2 pos); // :61-75
std::vector<Vec2> line_with_width(const Vec2 line[2], float width);                 // :106-111
} // namespace util

// fast marching (field.rs:118-192) on a rows x cols grid, in place
void apply_fmm(std::vector<float>& potential, const std::vector<float>& f, size_t rows, size_t cols);

// ---- build-owned RNG (the reference's fastrand global is unseeded) ------------------------
struct Rng {
    uint64_t state = 12345;
    uint64_t next();
    float f32();
    double f64();
    int32_t poisson(double lambda); // util.rs:78-89
};

// ---- models/mod.rs ------------------------------------------------------------------------
struct Pedestrian {             // models/mod.rs:28-32
    Vec2 pos;
    size_t destination = 0;
};

enum class Backend { Cpu, Gpu, Hip }; // lib.rs:138-142 + the variant this build adds

struct SimulatorOptions {       // lib.rs:108-135
    Backend backend = Backend::Hip;
    float neighbor_grid_unit = 1.4f;
    float field_grid_unit = 0.25f;
    bool use_neighbor_grid = true;
    bool use_distance_map = true;
    size_t gpu_work_size = 64;
    // build-owned
    int math_mode = PEDONI_MATH_EXACT;
    int device = 0;
    uint64_t seed = 12345;
};

class PedestrianModel {         // models/mod.rs:13-25
public:
    virtual ~PedestrianModel() = default;
    virtual void spawn_pedestrians(const Field& field, std::vector<Pedestrian> new_pedestrians) = 0;
    virtual void update_states(const Scenario& scenario, const Field& field) = 0;
    virtual std::vector<Pedestrian> list_pedestrians() const = 0;
    virtual int32_t get_pedestrian_count() const = 0;
};

// The MI355X backend: owns a PedoniModel* and forwards the trait methods to the C-ABI.
class SocialForceModelHip final : public PedestrianModel {
public:
    SocialForceModelHip(const SimulatorOptions& options, const Scenario& scenario, const Field& field);
    ~SocialForceModelHip() override;
    void spawn_pedestrians(const Field& field, std::vector<Pedestrian> new_pedestrians) override;
    void update_states(const Scenario& scenario, const Field& field) override;
    std::vector<Pedestrian> list_pedestrians() const override;
    int32_t get_pedestrian_count() const override;
    PedoniModel* handle() const { return model_; }
    // device time of the last update_states (fills StepMetrics.time_calc_state_kernel)
    std::optional<double> last_kernel_seconds() const { return last_kernel_s_; }

private:
    PedoniModel* model_ = nullptr;
    std::optional<double> last_kernel_s_;
};

struct StepMetrics {            // diagnostic.rs:45-50
    int32_t active_ped_count = 0;
    double time_spawn = 0.0;
    double time_calc_state = 0.0;
    std::optional<double> time_calc_state_kernel;
};

class Simulator {               // lib.rs:17-105
public:
    SimulatorOptions options;
    Scenario scenario;
    Field field;
    std::unique_ptr<PedestrianModel> model;
    int32_t step = 0;

    Simulator(SimulatorOptions options, Scenario scenario); // lib.rs:27-61
    // build-owned checkpoint / resume (upstream has none, and its
    // list_pedestrians drops velocity and desired speed).  The file holds the step counter,
    // both generator states and the model's full SoA state in model order; a resumed run
    // continues bit for bit like the uninterrupted one, spawns included.
    void save_checkpoint(const std::string& path);
    static std::unique_ptr<Simulator> resume(SimulatorOptions options, Scenario scenario,
                                             const std::string& path);
    StepMetrics tick();                                     // lib.rs:64-100
    // build-owned: `n` ticks with the periodic spawners evaluated ON THE DEVICE (same two RNG
    // streams, so the crowd is bit-identical to n calls of tick()); no per-tick host work.
    // The returned metrics cover the whole batch.
    StepMetrics tick_n(uint32_t n);
    std::vector<Pedestrian> list_pedestrians() const { return model->list_pedestrians(); } // :102

private:
    struct ResumeTag {};
    Simulator(SimulatorOptions options, Scenario scenario, ResumeTag); // no `once` spawns
    void init_field_and_model();
    Rng rng_;
    bool device_spawners_ = false;
    void hand_spawning_to_device();
    void take_spawning_back();
};

} // namespace pedoni_host
