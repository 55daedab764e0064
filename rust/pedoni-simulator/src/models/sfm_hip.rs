//! MI355X (gfx950) backend of `trait PedestrianModel` (models/mod.rs:13-25): a thin binding of
//! the C-ABI in `include/pedoni_hip.h` (`libpedoni_hip.so`).  One trait method = one
//! `pedoni_hip_*` entry point; nothing is computed on the Rust side.
//!
//! STATUS: UNVERIFIED.  The image this was written in has no `cargo` / `rustc`, so this file
//! has never been compiled.  It is kept as source (not as Markdown) so that it can be, the day
//! a toolchain with the crate's 163 dependencies exists:
//!     cp rust/pedoni-simulator/src/models/sfm_hip.rs <pedoni>/pedoni-simulator/src/models/
//!     cp rust/pedoni-simulator/build.rs              <pedoni>/pedoni-simulator/
//!     (cd <pedoni> && patch -p1 < <repo>/rust/upstream.patch)
//!     PEDONI_HIP_LIB_DIR=<repo>/pedoni_amd/lib cargo run -r -- -b hip -H scenarios/narrow-gap.toml
//! The same five calls are exercised, tested, from C++ (`pedoni_amd/csrc/host/simulator.cpp`),
//! plain C (`examples/c_abi_demo.c`) and Python ctypes (`pedoni_amd/abi.py`).
use std::ffi::{c_char, c_int, c_void, CStr};
use glam::Vec2;
use super::{Pedestrian, PedestrianModel};
use crate::{field::Field, scenario::Scenario, SimulatorOptions};

#[repr(C)]
struct PedoniOptions {
    neighbor_grid_unit: f32, field_grid_unit: f32,
    use_neighbor_grid: i32, use_distance_map: i32,
    gpu_work_size: i32, math_mode: i32,
    seed: u64, initial_capacity: u32, reserved: u32,
}
#[repr(C)] #[derive(Clone, Copy)]
struct PedoniObstacle { x0: f32, y0: f32, x1: f32, y1: f32, width: f32 }
#[repr(C)] #[derive(Clone, Copy, Default)]
struct PedoniPedestrian { x: f32, y: f32, destination: u64 }

#[link(name = "pedoni_hip")]
extern "C" {
    fn pedoni_hip_last_error() -> *const c_char;
    fn pedoni_hip_create(opt: *const PedoniOptions, size_x: f32, size_y: f32,
        distance_map: *const f32, potential_maps: *const *const f32, n_maps: u32,
        field_rows: u32, field_cols: u32, field_unit: f32,
        obstacles: *const PedoniObstacle, n_obstacles: u32, device: c_int,
        out: *mut *mut c_void) -> c_int;
    fn pedoni_hip_destroy(m: *mut c_void);
    fn pedoni_hip_spawn_pedestrians(m: *mut c_void, peds: *const PedoniPedestrian, n: u32) -> c_int;
    fn pedoni_hip_update_states(m: *mut c_void) -> c_int;
    fn pedoni_hip_list_pedestrians(m: *mut c_void, out: *mut PedoniPedestrian, cap: u32, n: *mut u32) -> c_int;
    fn pedoni_hip_get_pedestrian_count(m: *mut c_void, count: *mut i32) -> c_int;
}

fn check(rc: c_int) {
    if rc != 0 {
        let msg = unsafe { CStr::from_ptr(pedoni_hip_last_error()) }.to_string_lossy();
        panic!("pedoni_hip error {rc}: {msg}");   // the trait has no Result (sfm_gpu.rs:127)
    }
}

pub struct SocialForceModelHip { handle: *mut c_void }
unsafe impl Send for SocialForceModelHip {}   // every entry point binds its device
unsafe impl Sync for SocialForceModelHip {}

impl PedestrianModel for SocialForceModelHip {
    fn new(options: &SimulatorOptions, scenario: &Scenario, field: &Field) -> Self {
        let opt = PedoniOptions {
            neighbor_grid_unit: options.neighbor_grid_unit,
            field_grid_unit: options.field_grid_unit,
            use_neighbor_grid: options.use_neighbor_grid as i32,
            use_distance_map: options.use_distance_map as i32,
            gpu_work_size: 0, math_mode: 0 /* PEDONI_MATH_EXACT */,
            seed: 12345, initial_capacity: 0, reserved: 0,
        };
        // Array2<f32> is row-major (y, x): field.rs:194-205
        let dist = field.distance_map.as_standard_layout();
        let pots: Vec<_> = field.potential_maps.iter().map(|p| p.as_standard_layout()).collect();
        let pot_ptrs: Vec<*const f32> = pots.iter().map(|p| p.as_ptr()).collect();
        let obs: Vec<PedoniObstacle> = scenario.obstacles.iter().map(|o| PedoniObstacle {
            x0: o.line[0].x, y0: o.line[0].y, x1: o.line[1].x, y1: o.line[1].y, width: o.width,
        }).collect();
        let mut handle = std::ptr::null_mut();
        check(unsafe { pedoni_hip_create(&opt, scenario.field.size.x, scenario.field.size.y,
            dist.as_ptr(), pot_ptrs.as_ptr(), pot_ptrs.len() as u32,
            field.shape.0 as u32, field.shape.1 as u32, field.unit,
            obs.as_ptr(), obs.len() as u32, 0, &mut handle) });
        SocialForceModelHip { handle }
    }
    fn spawn_pedestrians(&mut self, _field: &Field, new_pedestrians: Vec<Pedestrian>) {
        let peds: Vec<PedoniPedestrian> = new_pedestrians.iter().map(|p| PedoniPedestrian {
            x: p.pos.x, y: p.pos.y, destination: p.destination as u64 }).collect();
        check(unsafe { pedoni_hip_spawn_pedestrians(self.handle, peds.as_ptr(), peds.len() as u32) });
    }
    fn update_states(&mut self, _scenario: &Scenario, _field: &Field) {
        check(unsafe { pedoni_hip_update_states(self.handle) });
    }
    fn list_pedestrians(&self) -> Vec<Pedestrian> {
        let mut n = 0u32;
        check(unsafe { pedoni_hip_list_pedestrians(self.handle, std::ptr::null_mut(), 0, &mut n) });
        let mut raw = vec![PedoniPedestrian::default(); n as usize];
        check(unsafe { pedoni_hip_list_pedestrians(self.handle, raw.as_mut_ptr(), n, &mut n) });
        raw.iter().map(|p| Pedestrian { pos: Vec2::new(p.x, p.y), destination: p.destination as usize }).collect()
    }
    fn get_pedestrian_count(&self) -> i32 {
        let mut c = 0i32;
        check(unsafe { pedoni_hip_get_pedestrian_count(self.handle, &mut c) });
        c
    }
}
impl Drop for SocialForceModelHip {
    fn drop(&mut self) { unsafe { pedoni_hip_destroy(self.handle) } }
}
