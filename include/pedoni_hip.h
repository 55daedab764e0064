/* pedoni_hip.h -- C ABI of the MI355X (gfx950) social-force backend.
 *
 * This is the drop-in boundary for ONE path of qt2/pedoni: the model plugin
 * `trait PedestrianModel` (pedoni-simulator/src/models/mod.rs:13-25) as implemented by
 * the CPU `SocialForceModel` (models/sfm.rs).  A third `Backend` variant in the
 * reference (lib.rs:32-35,138-142) would own a `PedoniModel*` and forward the five trait
 * methods to the `pedoni_hip_*` entry points marked [trait] below; INTEGRATION.md shows
 * that Rust binding.  Everything else here ([ext]) is what the reference's trait cannot
 * express but a device-resident backend needs: full-state injection/export (the
 * reference's `list_pedestrians` drops velocity and desired speed, sfm.rs:257-265),
 * a stream hook, multi-step launches, kernel timing and the ghost-row exchange used
 * when agents are sharded over several GPUs.
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on success or a
 * negative PEDONI_E_* code, and `pedoni_hip_last_error()` describes the last failure
 * on the calling thread.  The reference's trait has no `Result`: failures panic
 * (sfm_gpu.rs:127 `.unwrap()`), so a binding should `panic!` on non-zero.  The
 * library copies every input it keeps; callers own all buffers they pass.  An object
 * may be created on one thread and used from another (the reference moves the
 * simulator into a worker thread, pedoni/src/main.rs:79-81): every entry point binds the
 * model's device first.  No entry point is re-entrant for the same model.
 *
 * Arithmetic is fp32 throughout, evaluated in the reference's operation order without
 * FMA contraction; exp() replays glibc's expf in f64 so that results are bit-identical
 * to the reference CPU path on a glibc host (math_mode PEDONI_MATH_EXACT).
 */
#ifndef PEDONI_HIP_H
#define PEDONI_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PEDONI_OK 0
#define PEDONI_E_INVALID (-1)  /* bad argument */
#define PEDONI_E_HIP (-2)      /* HIP runtime error (message has the hipError string) */
#define PEDONI_E_NO_DEVICE (-3)
#define PEDONI_E_CAPACITY (-4) /* fixed-capacity buffer too small (halo exchange) */

#define PEDONI_MATH_EXACT 0 /* IEEE div/sqrt + glibc-replay exp: bit parity with the CPU path */
#define PEDONI_MATH_FAST 1  /* v_rcp/v_rsq/v_exp hardware approximations (<= 1e-5 rel.) */

typedef struct PedoniModel PedoniModel;

/* lib.rs:108-135 `SimulatorOptions` (same defaults) + backend-only knobs. */
typedef struct {
    float neighbor_grid_unit;  /* lib.rs:113, default 1.4  */
    float field_grid_unit;     /* lib.rs:115, default 0.25 (informational: maps carry their unit) */
    int32_t use_neighbor_grid; /* lib.rs:117, default 1 */
    int32_t use_distance_map;  /* lib.rs:119, default 1 */
    int32_t gpu_work_size;     /* lib.rs:121, workgroup size of the force kernel; 0 = library default */
    int32_t math_mode;         /* PEDONI_MATH_* */
    uint64_t seed;             /* desired-speed RNG seed (reference: unseeded fastrand, sfm.rs:54) */
    uint32_t initial_capacity; /* agents to pre-allocate for; arrays grow on demand */
    uint32_t reserved;
} PedoniOptions;

/* scenario.rs:23-35 `ObstacleConfig { line: [Vec2; 2], width }` */
typedef struct { float x0, y0, x1, y1, width; } PedoniObstacle;

/* models/mod.rs:28-32 `Pedestrian { pos: Vec2, destination: usize }` in a fixed C layout */
typedef struct { float x, y; uint64_t destination; } PedoniPedestrian;

/* diagnostic.rs:45-50 `StepMetrics`; time_calc_state_kernel < 0 means None */
typedef struct {
    int32_t active_ped_count;
    double time_spawn;
    double time_calc_state;
    double time_calc_state_kernel;
} PedoniStepMetrics;

/* per-kernel device time accumulated while profiling is enabled (hipEvent pairs) */
#define PEDONI_N_KERNELS 8
typedef struct {
    double total_ms[PEDONI_N_KERNELS];
    uint64_t launches[PEDONI_N_KERNELS];
} PedoniKernelTimes;
/* indices into PedoniKernelTimes */
#define PEDONI_K_BIN 0       /* cell key + despawn test + cell count of agents stored since the last update */
#define PEDONI_K_SCAN 1      /* exclusive scan -> neighbor_grid_indices     */
#define PEDONI_K_SLOT 2      /* place: rank inside the new cell + SoA move (or provisional slot) */
#define PEDONI_K_REORDER 3   /* general form: in-cell rank + SoA scatter     */
#define PEDONI_K_FORCE 4     /* goal + pair + obstacle force + integrator + next key / cell count */
#define PEDONI_K_HALO_PACK 5
#define PEDONI_K_HALO_UNPACK 6
#define PEDONI_K_OTHER 7

const char* pedoni_hip_last_error(void);
void pedoni_hip_default_options(PedoniOptions* opt);
int pedoni_hip_device_count(int32_t* n);

/* [trait] PedestrianModel::new (models/mod.rs:14; sfm.rs:36-46).  `size_*` is
 * scenario.field.size; the maps are `Field.distance_map` / `Field.potential_maps`
 * (field.rs:194-205), row-major (y, x), `field_rows x field_cols`, sampled at
 * `field_unit`; obstacles = scenario.obstacles (read only when use_distance_map == 0,
 * sfm.rs:194).  All inputs are copied to the device. */
int pedoni_hip_create(const PedoniOptions* opt, float size_x, float size_y,
                      const float* distance_map, const float* const* potential_maps,
                      uint32_t n_maps, uint32_t field_rows, uint32_t field_cols,
                      float field_unit, const PedoniObstacle* obstacles, uint32_t n_obstacles,
                      int device, PedoniModel** out);
/* [ext] the same for one band of a sharded run: the map pointers still address the FULL
 * field_rows x field_cols host arrays, but only the texel rows [map_row_begin, map_row_end) are
 * copied to the device (a band of a 1000 x 8000 m field needs 1/8 of each 512 MB map).  The rows
 * must cover every position the band's agents -- owned, ghost, or one tick's step beyond --
 * can sample: pedoni_shard_map_rows computes them.  Sampling outside them is never a fault: it
 * raises a sticky device status (PEDONI_E_CAPACITY from the next read of device state). */
int pedoni_hip_create_rows(const PedoniOptions* opt, float size_x, float size_y,
                           const float* distance_map, const float* const* potential_maps,
                           uint32_t n_maps, uint32_t field_rows, uint32_t field_cols,
                           float field_unit, const PedoniObstacle* obstacles, uint32_t n_obstacles,
                           int device, uint32_t map_row_begin, uint32_t map_row_end,
                           PedoniModel** out);
void pedoni_hip_destroy(PedoniModel* m);

/* [trait] PedestrianModel::spawn_pedestrians (models/mod.rs:18; sfm.rs:48-89): append
 * `n` agents (velocity 0, desired speed drawn from N^(1.34, 0.26) inside the model),
 * then bin, stable-sort by cell and despawn.  Called with n == 0 every tick by
 * Simulator::tick (lib.rs:85) -- it IS the sort/despawn pass. */
int pedoni_hip_spawn_pedestrians(PedoniModel* m, const PedoniPedestrian* peds, uint32_t n);
/* [trait] PedestrianModel::update_states (models/mod.rs:20; sfm.rs:91-255) */
int pedoni_hip_update_states(PedoniModel* m);
/* [trait] PedestrianModel::list_pedestrians (models/mod.rs:22; sfm.rs:257-265).
 * Writes min(count, cap) entries, stores the live count in *n. */
int pedoni_hip_list_pedestrians(PedoniModel* m, PedoniPedestrian* out, uint32_t cap,
                                uint32_t* n);
/* [trait] PedestrianModel::get_pedestrian_count (models/mod.rs:24; sfm.rs:267-269) */
int pedoni_hip_get_pedestrian_count(PedoniModel* m, int32_t* count);

/* [ext] append with full state (state injection).  desired_speed NULL ->
 * drawn from the model RNG as the trait method does; vel_xy NULL -> zero.  Append only:
 * the agents take part from the next sort/despawn pass on. */
int pedoni_hip_append(PedoniModel* m, const float* pos_xy, const uint32_t* destination,
                      const float* desired_speed, const float* vel_xy, uint32_t n);
/* [ext] the sort/despawn half of spawn_pedestrians alone (sfm.rs:58-88) */
int pedoni_hip_sort_despawn(PedoniModel* m);
/* [ext] `steps` x (sort_despawn; update_states) with no host round trip in between; in steady
 * state (nothing appended, no exchange, no device spawning) runs of 16 / 8 / 4 / 2 ticks are replayed
 * from captured hipGraphs -- same kernels, one graph launch per run (PEDONI_NO_GRAPH=1 disables it,
 * PEDONI_GRAPH_TICKS=n caps the run length) */
int pedoni_hip_tick_n(PedoniModel* m, uint32_t steps);
/* [ext] one Simulator::tick-shaped step with StepMetrics (lib.rs:64-100), no new agents */
int pedoni_hip_tick(PedoniModel* m, PedoniStepMetrics* metrics);
/* [ext] full SoA state of the live agents in model order; any pointer may be NULL */
int pedoni_hip_download(PedoniModel* m, float* pos_xy, uint32_t* destination, float* vel_xy,
                        float* desired_speed, uint32_t cap, uint32_t* n);
/* [ext] drop all agents */
int pedoni_hip_clear(PedoniModel* m);
/* [ext] `neighbor_grid_indices` (sfm.rs:22,62-74): rows*cols+1 prefix counts.
 * Stores the length in *len; copies min(len, cap) entries when out != NULL. */
int pedoni_hip_neighbor_grid_indices(PedoniModel* m, uint32_t* out, uint32_t cap,
                                     uint32_t* len);
int pedoni_hip_neighbor_grid_shape(PedoniModel* m, uint32_t* rows, uint32_t* cols);
/* [ext] the per-cell early-out table built at create (rows * cols words, neighbor-grid order; *len = 0
 * when PEDONI_NO_CELL_FLAGS=1 left it out).  Bit m < 31 of word c: `get_potential(m, pos) > 0.25`
 * (sfm.rs:69) holds for every position in the 3 x 3 cells around c, so the despawn test of an agent that
 * starts its step in c and ends it there needs no sample; bit 31: the wall term of sfm.rs:188-192 is
 * (+-0, +-0) for every position in c (exp(-distance / 0.2) underflows to 0, the gradient cannot vanish).
 * Both are exact statements about the maps -- a set bit never changes a result -- and the tests check
 * them against the oracle's samples.  Same calling convention as neighbor_grid_indices. */
int pedoni_hip_cell_flags(PedoniModel* m, uint32_t* out, uint32_t cap, uint32_t* len);
/* [ext] the workgroup order of the force launch ("heaviest tiles first": a tile = 256 consecutive sorted
 * agents; placement only, results do not depend on it) as the last sort pass built it, and the per-tile
 * weights (candidate counts of the tile's 256 lanes) currently stored -- those of the
 * last force launch, i.e. the input of the NEXT pass's order.  *n_blocks = 0 when the launch keeps the plain
 * order (small crowds, bands, PEDONI_NO_TILE_ORDER=1); order[b] = tile of hardware workgroup b. */
int pedoni_hip_tile_order(PedoniModel* m, uint32_t* order, uint32_t* tile_weight, uint32_t cap, uint32_t* n_blocks);
/* [ext] accelerations of sfm.rs:93-241 for the current sorted state (no integration) */
int pedoni_hip_calc_accelerations(PedoniModel* m, float* acc_xy, uint32_t cap);

/* [ext] on-device periodic spawning.  Installs the scenario's
 * `periodic` spawners (lib.rs:70-84): from then on every tick of pedoni_hip_tick_n first
 * draws, on the device, count = poisson(frequency / 10) arrivals per spawner at
 * p1.lerp(p2, u) from the POSITION stream (continued from `position_rng_state`, the host
 * Simulator's generator state) and their desired speeds from the model's own stream --
 * draw for draw what Simulator::tick + spawn_pedestrians do on the host, so both routes give
 * bit-identical crowds.  `max_per_tick` bounds one tick's arrivals (exceeding it is reported
 * as PEDONI_E_CAPACITY by owned_count / get_spawn_rng callers; no silent drop).  n == 0
 * uninstalls.  Not available for a band of a sharded run. */
typedef struct { float x0, y0, x1, y1; uint32_t destination; uint32_t reserved; double frequency; } PedoniSpawner;
int pedoni_hip_set_spawners(PedoniModel* m, const PedoniSpawner* spawners, uint32_t n,
                            uint64_t position_rng_state, uint32_t max_per_tick);
/* current states of the two streams (to hand spawning back to the host) */
int pedoni_hip_get_spawn_rng(PedoniModel* m, uint64_t* position_rng_state, uint64_t* speed_rng_state);
/* [ext] restore the model's desired-speed stream (sfm.rs:54 draws from it), e.g. when a
 * checkpoint is resumed: with pedoni_hip_download / pedoni_hip_append (full SoA state) and
 * pedoni_hip_get_spawn_rng this makes the model's state exportable and importable whole
 * (upstream's list_pedestrians drops velocity and desired speed) */
int pedoni_hip_set_speed_rng(PedoniModel* m, uint64_t speed_rng_state);

/* [ext] stream / timing */
/* All launches and copies of the model go to `hip_stream` (a hipStream_t; NULL is HIP's
 * default stream) or, with use_library_stream != 0, back to the model's own stream. */
int pedoni_hip_set_stream(PedoniModel* m, void* hip_stream, int32_t use_library_stream);
int pedoni_hip_get_stream(PedoniModel* m, void** hip_stream);
int pedoni_hip_synchronize(PedoniModel* m);
/* bit k of `kernel_mask` = time launches of PEDONI_K_<k> with a hipEvent pair (-1 = all,
 * 0 = off).  An event pair costs a few microseconds per launch on the stream. */
int pedoni_hip_profile(PedoniModel* m, int32_t kernel_mask);
/* inside pedoni_hip_tick_n only every `every_ticks`-th tick is timed (default 1 = all): the
 * timed ticks launch eagerly, the others may replay the captured graph */
int pedoni_hip_profile_every(PedoniModel* m, uint32_t every_ticks);
/* [ext] ... or `burst_ticks` ticks in a row out of every `every_ticks`, the first burst starting with the next tick:
 * the timed ticks launch eagerly, and the plain ticks between two bursts replay from captured graphs in runs of up to
 * 16 ticks per graph launch -- a short timed region (bench.py's 20 steps) is then 7 timed ticks and two long runs
 * instead of 7 timed ticks with a pair of plain ones between each two */
int pedoni_hip_profile_burst(PedoniModel* m, uint32_t every_ticks, uint32_t burst_ticks);
int pedoni_hip_kernel_times(PedoniModel* m, PedoniKernelTimes* out, int32_t reset);
const char* pedoni_hip_kernel_name(int32_t k);

/* [ext] row-band sharding over several GPUs (no reference counterpart).
 * A model that owns neighbor-grid rows [row_begin, row_end) keeps the agents of those rows
 * plus ghost copies of rows row_begin-1 and row_end.  Each tick, before sort/despawn:
 *   halo_pack   writes the full 24-byte state of the owned agents whose CURRENT cell row
 *               is row_begin-1 or row_begin (the "down" list, for the band below) and
 *               row_end-1 or row_end (the "up" list, for the band above) into `send`, a
 *               caller-owned DEVICE buffer of pedoni_hip_halo_bytes(cap_each) bytes:
 *               [down: header, cap_each records][up: header, cap_each records];
 *   (caller)    exchanges the buffers with RCCL (torch.distributed all_gather_into_tensor);
 *   halo_unpack takes the whole buffer of the band below (its UP list is used) and of the
 *               band above (its DOWN list), NULL at the outer bands, and stores the first
 *               in FRONT of the model's own agents and the second BEHIND them, so that the
 *               stable cell sort reproduces the single-GPU order bit for bit (lower bands
 *               hold lower global indices).  Own agents are never moved; last tick's
 *               ghosts were marked dead by update_states.
 * Forces are evaluated and integrated for owned rows only.  An agent may cross at most
 * one grid row per tick (checked; PEDONI_E_INVALID from owned_count otherwise). */
#define PEDONI_HALO_HEADER_WORDS 4 /* u32 count, overflow flag, 2 reserved */
#define PEDONI_HALO_RECORD_WORDS 6 /* pos.xy, vel.xy, desired_speed, destination */
/* set_band must be called on a model that holds no agents; `halo_cap` = capacity of each
 * received list (agents), reserved in front of and behind the model's own agents. */
int pedoni_hip_set_band(PedoniModel* m, int32_t row_begin, int32_t row_end, uint32_t halo_cap);
int pedoni_hip_halo_bytes(uint32_t cap_each, uint64_t* bytes); /* size of one rank's buffer */
int pedoni_hip_halo_pack(PedoniModel* m, void* send_dev, uint32_t cap_each);
int pedoni_hip_halo_unpack(PedoniModel* m, const void* from_below_dev,
                           const void* from_above_dev, uint32_t cap_each);
/* one sharded tick after the exchange, in one call: halo_unpack, sort/despawn,
 * update_states, then halo_pack of the NEXT tick's lists into `send_dev` */
int pedoni_hip_halo_tick(PedoniModel* m, const void* from_below_dev, const void* from_above_dev,
                         void* send_dev, uint32_t cap_each);
/* the same tick in two halves, so the all-gather of the next tick's lists overlaps the bulk
 * of the force kernel: _begin = unpack, sort/despawn, update of the rows beside the band's
 * edges, pack into `send_dev` (start the exchange now); _end = update of the interior rows */
int pedoni_hip_halo_tick_begin(PedoniModel* m, const void* from_below_dev,
                               const void* from_above_dev, void* send_dev, uint32_t cap_each);
int pedoni_hip_halo_tick_end(PedoniModel* m);
/* owned-agent count (excludes ghosts) */
int pedoni_hip_owned_count(PedoniModel* m, int32_t* count);

/* [ext] the multi-GPU driver below the C-ABI (no reference counterpart; SURVEY 5.8, 8(e);
 * BASELINE north_star: "per-step RCCL all-gather over xGMI of ghost agents in halo grid
 * cells").  One process per GPU; a PedoniShard wraps the rank's PedoniModel, owns an RCCL
 * communicator and drives   exchange -> halo_unpack -> sort/despawn -> update_states ->
 * halo_pack   every tick on the model's stream, with no host synchronisation per tick.  The
 * exchange is ONE grouped ncclSend / ncclRecv pair with rank-1 and rank+1 (only neighbours
 * ever read a band's lists: 2 of the 7 xGMI links, minimal bytes -- SURVEY 5.8 option 1),
 * issued by this library straight into librccl (resolved with dlopen at first use, so the
 * library also loads where RCCL is absent).  A Rust / C++ / C host needs nothing else to run
 * 8 GPUs: distribute the 128-byte id of rank 0 by any channel, create, append the band's
 * agents, begin, tick_n.
 *
 * Band boundaries are grid rows: `row_bounds` holds world + 1 ascending entries, rank r owns
 * rows [row_bounds[r], row_bounds[r+1]).  pedoni_shard_balanced_bounds cuts the rows so that
 * every band holds about the same number of AGENTS given the per-row counts (the cell_start
 * prefix of a sorted crowd), with at least `min_rows` rows per band. */
typedef struct PedoniShard PedoniShard;
#define PEDONI_SHARD_ID_BYTES 128
int pedoni_shard_unique_id(uint8_t id[PEDONI_SHARD_ID_BYTES]);
/* texel rows of the field maps a band of grid rows [row_begin, row_end) needs, `slack_rows` grid
 * rows of room on either side included (room for the periodic re-cut to move the band) */
int pedoni_shard_map_rows(int32_t row_begin, int32_t row_end, int32_t slack_rows, float neighbor_grid_unit,
                          float field_unit, uint32_t field_rows, uint32_t* map_row_begin,
                          uint32_t* map_row_end);
int pedoni_shard_balanced_bounds(const uint32_t* row_counts, uint32_t n_rows, int32_t world,
                                 int32_t min_rows, int32_t* bounds_out /* world + 1 */);
/* one step of the periodic re-cut (pure host code; what every rank computes from the all-reduced
 * per-row counts): each boundary moves towards the agent-balanced cut by at most `max_shift`
 * rows, bands keep >= 6 rows, the rows handed over fit `bulk_cap` agents, and -- with
 * `map_slack_rows` >= 0 -- no boundary leaves bounds0[b] +- that slack */
int pedoni_shard_recut_bounds(const int32_t* bounds, int32_t world, const uint32_t* row_counts,
                              uint32_t n_rows, uint32_t max_shift, uint32_t bulk_cap,
                              const int32_t* bounds0, int32_t map_slack_rows, int32_t* bounds_out);
/* `id` NULL: no communicator (world == 1, or a member of a local group, below).  The model
 * must hold no agents; it is given the band [row_bounds[rank], row_bounds[rank+1]). */
int pedoni_shard_create(PedoniModel* m, int32_t rank, int32_t world, const uint8_t* id,
                        const int32_t* row_bounds, uint32_t halo_cap, PedoniShard** out);
void pedoni_shard_destroy(PedoniShard* s); /* does not destroy the model */
/* after the rank's own agents were appended (pedoni_hip_append): first pass + first pack */
int pedoni_shard_begin(PedoniShard* s);
int pedoni_shard_tick_n(PedoniShard* s, uint32_t steps);
int pedoni_shard_owned_count(PedoniShard* s, int32_t* count);
int pedoni_shard_band(PedoniShard* s, int32_t* row_begin, int32_t* row_end);
/* on: the exchange of the NEXT tick's lists runs on a stream of its own while this tick's interior
 * rows are still being computed: the force launch takes the tiles of the rows beside the band's edges
 * first and releases pack + send from inside the launch (small bands: an edge launch, pack, send; then
 * the interior launch).  Same results bit for bit; pays once the exchange costs more than the ~6 us of
 * the cross-stream join -- bench.py times both and keeps the faster. */
int pedoni_shard_set_overlap(PedoniShard* s, int32_t on);
/* [ext] how the overlapped ticks so far were run: `edge_first` counts those whose ONE force launch took
 * the edge rows' tiles first and released the exchange from inside the launch, `split` those run as an
 * edge launch + an interior launch (small bands on the G-lanes-per-agent kernels, PEDONI_SHARD_FORM=split),
 * `plain` every other tick (overlap off, band too thin to split, re-cut ticks). */
int pedoni_shard_tick_forms(PedoniShard* s, uint32_t* edge_first, uint32_t* split, uint32_t* plain);
/* a token ring through the very ncclSend / ncclRecv pair the exchange uses (self-addressed at
 * the outer bands): PEDONI_OK iff both neighbours' tokens arrived */
int pedoni_shard_selftest(PedoniShard* s);
/* every `every_ticks` ticks (0 = never) the bands are re-cut from the global per-row agent
 * counts so that each holds about N / world agents: the rows that change owner travel, full
 * state, to the neighbour in one grouped send / receive; results stay bit-identical to one GPU.
 * `map_slack_rows` < 0: every rank holds the whole field maps; >= 0: every rank's model was
 * created with pedoni_hip_create_rows over pedoni_shard_map_rows(its initial band, that slack),
 * and no boundary moves further than the slack from where it started */
int pedoni_shard_set_rebalance(PedoniShard* s, uint32_t every_ticks, uint32_t max_rows_per_step,
                               int32_t map_slack_rows);
/* G shards of ONE process on one device, the transport replaced by device copies: the same
 * driver code, testable on a single GPU (tests/test_gpu_shard.py).  Ticks all shards in
 * lockstep. */
int pedoni_shard_local_group_tick_n(PedoniShard** shards, uint32_t n_shards, uint32_t steps);

/* [ext] which instantiation of the force kernel a whole-array launch over `n_agents` agents of this
 * model takes -- its symbol as rocprofv3 prints it, e.g. "force_kernel_queue_s94<0, 6>" from 4e5
 * agents up, "force_kernel_queue_group<0, 8, 2>" (2 lanes per agent) for small crowds -- and how
 * many agents one wave of it owns.  bench.py prices a run's instruction issue with the committed
 * counter profile of exactly that kernel. */
int pedoni_hip_force_kernel_info(PedoniModel* m, uint32_t n_agents, char* name, uint32_t name_cap,
                                 uint32_t* agents_per_wave);

/* [ext] opt-in GPU builder of the field maps (no bit parity with upstream's
 * heap fast marching, field.rs:118-192, whose numbers depend on its pop order): solves
 * |grad u| = f on a rows x cols grid by a block fast iterative method on the first-order upwind
 * update.  `potential` (host, in/out): 0 on the zero set, >= 1e23 elsewhere; `slowness`: one f
 * per cell (host) or NULL for the constant `uniform_slowness`.  pedoni_field_build_gpu
 * (pedoni_host.h) wraps it into Field::from_scenario's shape. */
int pedoni_hip_eikonal(int device, float* potential, const float* slowness, float uniform_slowness,
                       uint32_t rows, uint32_t cols, uint32_t* launches_out);

#ifdef PEDONI_DIAGNOSTICS
/* Diagnostics: NOT part of the product library.  `python -m pedoni_amd.build` compiles a second
 * library, pedoni_amd/lib/libpedoni_hip_diag.so, with -DPEDONI_DIAGNOSTICS: the same sources plus
 * the instrumented / ablation instantiations of the force kernel, the PEDONI_ABLATE and
 * PEDONI_FORCE_TRACE switches and the three entry points below.  The tests that need the fault
 * injection hook and tools/force_trace.py, ablate_launch.py load that library; nothing else does. */
/* [ext] test hook: overwrite the model's sticky device status word (the word the scan and
 * place kernels raise when cell and row counts disagree or the live count exceeds the host's
 * bound of the arrays).  While it is non-zero every read of device state -- get_pedestrian_count,
 * download, list_pedestrians, owned_count -- fails with PEDONI_E_HIP / PEDONI_E_CAPACITY. */
int pedoni_hip_debug_set_status(PedoniModel* m, uint32_t status_word);
/* Diagnostics: the force kernel's ablation mask (what PEDONI_ABLATE sets at create): bit 0 no
 * goal sampling, 1 no wall term, 2 no pairs, 3 phase 2 without its gather, 4 phase 2 without its
 * arithmetic, 5 no despawn sampling, 6 no row counts, 7 no counts.  Timing only -- results are
 * wrong while any bit is set (tools/ablate_launch.py). */
int pedoni_hip_debug_set_ablate(PedoniModel* m, uint32_t bits);
/* [ext] diagnostic: a model created under PEDONI_FORCE_TRACE=1 runs an instrumented build of the
 * force kernel (never the product kernel) whose waves add the shader cycles they spent in the
 * prologue, phases 1 / 2 / 3 and the epilogue (sums7[0..4]), their lifetimes ([5]) and their
 * number ([6]); tools/force_trace.py prints the shares */
int pedoni_hip_debug_force_trace(PedoniModel* m, uint64_t* sums7, int32_t reset);
/* the raw records behind it: 8 words per wave -- the five phase sums, lifetime, launches, start stamp */
int pedoni_hip_debug_force_trace_raw(PedoniModel* m, uint64_t* out, uint32_t n_waves);
#endif /* PEDONI_DIAGNOSTICS */

/* [ext] device self-test hooks used by tests/: evaluate one device math primitive over
 * host arrays (op: 0 = a/b, 1 = sqrt(a), 2 = exp(a), 3 = a/0.3f, 4 = a/0.2f,
 * 5 = Rust-style `a as i32`, result bits returned in the float, 6 = a/b by the pair
 * force's unscaled division core -- equal to a/b wherever pair_force_hot uses it) */
int pedoni_hip_selftest_math(int device, int32_t op, int32_t math_mode, const float* a,
                             const float* b, float* out, uint32_t n);
/* [ext] the same for the pair force (sfm.rs:130-153) of n independent (agent, neighbour)
 * pairs: acc_xy (in/out) += force on an agent at pos_xy with goal direction e_xy from a
 * neighbour at pos_i_xy moving with vel_i_xy -- the device function both force kernels call */
/* [ext] the field stencil: sobel_filter and the bilinear centre sample (util.rs:44-75) of a
 * rows x cols map at n grid-coordinate points, by the shared-patch device form */
int pedoni_hip_selftest_field(int device, const float* grid, uint32_t rows, uint32_t cols,
                              const float* px, const float* py, float* grad_xy, float* centre,
                              uint32_t n);
int pedoni_hip_selftest_pair(int device, int32_t math_mode, const float* pos_xy, const float* e_xy,
                             const float* pos_i_xy, const float* vel_i_xy, float* acc_xy, uint32_t n);

#ifdef __cplusplus
}
#endif
#endif
