// pedoni-headless -- the reference binary's headless mode (pedoni/src/main.rs:106-136,
// args.rs:11-44) on top of the C++ host mirror.  Same flags, same per-100-step log line,
// same wall-clock throttle (one tick per DELTA_TIME / speed seconds), same stop rule
// (stops once total_steps > max_steps, i.e. after max_steps + 1 ticks) and the same
// `logs/%Y-%m-%d_%H%M%S_log.json` DiagnositcLog [sic] (diagnostic.rs:5-50).  The renderer is
// out of scope: running without -H is an error here.
#include "pedoni_host.hpp"

#include <atomic>
#include <chrono>
#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fstream>
#include <sstream>
#include <string>
#include <sys/stat.h>
#include <thread>
#include <vector>

using namespace pedoni_host;

namespace {

constexpr float DELTA_TIME = 0.1f; // pedoni/src/main.rs:28
std::atomic<bool> g_sigint{false};

struct Args {                      // pedoni/src/args.rs:11-44
    std::string scenario = "scenarios/default.toml";
    bool headless = false;
    std::string backend = "hip";   // upstream: cpu | gpu (default cpu); this build adds hip
    float speed = 100.0f;
    bool no_neighbor_grid = false, no_distance_map = false;
    std::optional<float> field_unit, neighbor_unit;
    std::optional<size_t> work_size, max_steps;
    // build-owned
    uint64_t seed = 12345;
    int device = 0;
    std::string log_dir = "logs";
    bool fast_math = false;
    std::string load_state, save_state;   // checkpoint files
};

void usage()
{
    std::puts("Usage: pedoni-headless [OPTIONS] [SCENARIO]\n\n"
              "Arguments:\n  [SCENARIO]  Path to scenario file [default: scenarios/default.toml]\n\n"
              "Options:\n"
              "  -H, --headless               Runs in headless mode (required: no renderer in this build)\n"
              "  -b, --backend <BACKEND>      Backend [default: hip] [possible values: cpu, gpu, hip]\n"
              "  -s, --speed <SPEED>          Max playback speed [default: 100]\n"
              "      --no-neighbor-grid       Do not use grid for acceleration\n"
              "      --no-distance-map        Do not use distance map\n"
              "      --field-unit <F>         Unit length of field navigation grid\n"
              "      --neighbor-unit <F>      Unit length of neighbor search grid\n"
              "      --work-size <N>          Local work size of GPU kernel\n"
              "      --max-steps <N>          Max steps to simulate\n"
              "      --seed <N>               Seed of the spawn / desired-speed streams [default: 12345]\n"
              "      --device <N>             HIP device index [default: 0]\n"
              "      --log-dir <DIR>          Where the JSON log goes [default: logs]\n"
              "      --fast-math              PEDONI_MATH_FAST instead of bit-exact arithmetic\n"
              "      --load-state <FILE>      Resume from a checkpoint instead of the scenario's initial spawns\n"
              "      --save-state <FILE>      Write a checkpoint when the run ends\n"
              "  -h, --help                   Print help");
}

[[noreturn]] void die(const std::string& msg)
{
    std::fprintf(stderr, "error: %s\n", msg.c_str());
    std::exit(2);
}

Args parse(int argc, char** argv)
{
    Args a;
    bool have_scenario = false;
    auto value = [&](int& i, const char* name) -> std::string {
        if (i + 1 >= argc) die(std::string("a value is required for '") + name + "'");
        return argv[++i];
    };
    for (int i = 1; i < argc; ++i) {
        std::string s = argv[i];
        if (s == "-h" || s == "--help") { usage(); std::exit(0); }
        else if (s == "-H" || s == "--headless") a.headless = true;
        else if (s == "-b" || s == "--backend") a.backend = value(i, "--backend");
        else if (s == "-s" || s == "--speed") a.speed = std::stof(value(i, "--speed"));
        else if (s == "--no-neighbor-grid") a.no_neighbor_grid = true;
        else if (s == "--no-distance-map") a.no_distance_map = true;
        else if (s == "--field-unit") a.field_unit = std::stof(value(i, "--field-unit"));
        else if (s == "--neighbor-unit") a.neighbor_unit = std::stof(value(i, "--neighbor-unit"));
        else if (s == "--work-size") a.work_size = (size_t)std::stoul(value(i, "--work-size"));
        else if (s == "--max-steps") a.max_steps = (size_t)std::stoul(value(i, "--max-steps"));
        else if (s == "--seed") a.seed = std::stoull(value(i, "--seed"));
        else if (s == "--device") a.device = std::stoi(value(i, "--device"));
        else if (s == "--log-dir") a.log_dir = value(i, "--log-dir");
        else if (s == "--fast-math") a.fast_math = true;
        else if (s == "--load-state") a.load_state = value(i, "--load-state");
        else if (s == "--save-state") a.save_state = value(i, "--save-state");
        else if (!s.empty() && s[0] == '-') die("unexpected argument '" + s + "'");
        else if (!have_scenario) { a.scenario = s; have_scenario = true; }
        else die("unexpected argument '" + s + "'");
    }
    return a;
}

SimulatorOptions to_simulator_options(const Args& a) // args.rs:47-66
{
    SimulatorOptions o;
    if (a.backend == "cpu") o.backend = Backend::Cpu;
    else if (a.backend == "gpu") o.backend = Backend::Gpu;
    else if (a.backend == "hip") o.backend = Backend::Hip;
    else die("invalid value '" + a.backend + "' for '--backend' [possible values: cpu, gpu, hip]");
    o.use_neighbor_grid = !a.no_neighbor_grid;
    o.use_distance_map = !a.no_distance_map;
    if (a.field_unit) o.field_grid_unit = *a.field_unit;
    if (a.neighbor_unit) o.neighbor_grid_unit = *a.neighbor_unit;
    // upstream parses --work-size but never copies it (args.rs:38-40 vs :47-66); here it is honoured
    if (a.work_size) o.gpu_work_size = *a.work_size;
    o.math_mode = a.fast_math ? PEDONI_MATH_FAST : PEDONI_MATH_EXACT;
    o.device = a.device;
    o.seed = a.seed;
    return o;
}

std::string fmt_f64(double v)
{
    char buf[40];
    std::snprintf(buf, sizeof buf, "%.17g", v);
    std::string s = buf;
    if (s.find_first_of(".eEn") == std::string::npos) s += ".0"; // serde_json prints 1.0, not 1
    return s;
}

} // namespace

int main(int argc, char** argv)
{
    Args args = parse(argc, argv);
    if (!args.headless) die("this build has no renderer: run with -H / --headless");

    std::ifstream in(args.scenario);
    if (!in) die("cannot read scenario file '" + args.scenario + "'");
    std::stringstream text;
    text << in.rdbuf();

    try {
        Scenario scenario = Scenario::from_toml(text.str());              // main.rs:55
        auto t_field = std::chrono::steady_clock::now();
        std::unique_ptr<Simulator> sim_owner =                             // main.rs:79
            args.load_state.empty() ? std::make_unique<Simulator>(to_simulator_options(args), scenario)
                                    : Simulator::resume(to_simulator_options(args), scenario, args.load_state);
        Simulator& simulator = *sim_owner;
        const double time_new = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_field).count();

        std::signal(SIGINT, [](int) { g_sigint = true; });                 // main.rs:108
        std::fprintf(stderr, "[INFO  pedoni] Run as headless mode\n");

        std::vector<StepMetrics> log;                                      // DiagnositcLog.step_metrics
        size_t total_steps = 0;
        const auto min_interval = std::chrono::duration<double>(DELTA_TIME / args.speed); // main.rs:99
        while (!g_sigint && !(args.max_steps && total_steps > *args.max_steps)) { // main.rs:112-116
            const auto start = std::chrono::steady_clock::now();
            StepMetrics m = simulator.tick();                              // main.rs:86
            if (simulator.step % 100 == 0)                                 // main.rs:87-92
                std::fprintf(stderr, "[INFO  pedoni] Step: %6d, Active pedestrians: %6d\n",
                             simulator.step, m.active_ped_count);
            log.push_back(m);                                              // main.rs:96
            total_steps += 1;
            const auto spent = std::chrono::steady_clock::now() - start;
            if (spent < min_interval) std::this_thread::sleep_for(min_interval - spent); // main.rs:100-103
        }

        if (!args.save_state.empty()) {
            simulator.save_checkpoint(args.save_state);
            std::fprintf(stderr, "[INFO  pedoni] Saved checkpoint: %s (step %d)\n", args.save_state.c_str(),
                         simulator.step);
        }

        mkdir(args.log_dir.c_str(), 0777);                                 // main.rs:119
        char stamp[64];
        std::time_t now = std::time(nullptr);
        std::strftime(stamp, sizeof stamp, "%Y-%m-%d_%H%M%S_log.json", std::localtime(&now));
        const std::string path = args.log_dir + "/" + stamp;
        std::ofstream out(path);
        if (!out) die("cannot create '" + path + "'");
        // diagnostic.rs:5-43; model / scenario stay empty as upstream never fills them;
        // preprocess_metrics.time_calc_field gets Simulator::new's time (upstream leaves 0.0)
        out << "{\"model\":\"\",\"scenario\":\"\",\"total_steps\":" << total_steps
            << ",\"preprocess_metrics\":{\"time_calc_field\":" << fmt_f64(time_new) << "},\"step_metrics\":{";
        auto column = [&](const char* name, auto get, bool last) {
            out << "\"" << name << "\":[";
            for (size_t i = 0; i < log.size(); ++i) out << (i ? "," : "") << get(log[i]);
            out << "]" << (last ? "" : ",");
        };
        column("active_ped_count", [](const StepMetrics& m) { return std::to_string(m.active_ped_count); }, false);
        column("time_spawn", [](const StepMetrics& m) { return fmt_f64(m.time_spawn); }, false);
        column("time_calc_state", [](const StepMetrics& m) { return fmt_f64(m.time_calc_state); }, false);
        column("time_calc_state_kernel", [](const StepMetrics& m) {
            return m.time_calc_state_kernel ? fmt_f64(*m.time_calc_state_kernel) : std::string("null"); }, true);
        out << "}}";
        out.close();
        std::fprintf(stderr, "[INFO  pedoni] Exported log file: %s\n", path.c_str()); // main.rs:130
    } catch (const std::exception& e) {
        die(e.what());
    }
    return 0;
}
