//! takzero-hip: the reference's seams for the hot path on top of `takzero-hip-sys`.
//!
//! * [`HipNet`] implements `takzero::network::Network` (takzero/src/network/mod.rs:10-45) and
//!   `takzero::search::agent::Agent<Env>` (takzero/src/search/agent.rs:5-14; what `impl Agent for Net`, net5.rs:220-285, does through
//!   LibTorch).  With only this the reference's own CPU `BatchedMCTS` keeps running and just its network moves to the MI355X.
//! * [`HipBatchedMCTS`] has the methods of `takzero::search::node::batched::BatchedMCTS` (batched.rs:32-409) over the search that
//!   lives on the device; `selfplay/src/main.rs:80` and `reanalyze/src/main.rs:69` construct it instead.
//!
//! Randomness never crosses the ABI: Dirichlet / Gumbel / opening draws are made here from the caller's `Rng`, in the reference's
//! order.  No Rust toolchain exists in the image this was written in: the file is the binding as a maintainer would add it, checked
//! there only for what a parser can check (tests/test_rust_binding.py: every `sys::` symbol used below exists in the sys crate with
//! the arity it is called with).
use std::ffi::{CStr, CString};
use std::path::Path;

use fast_tak::takparse::Move;
use ordered_float::NotNan;
use rand::Rng;
use rand_distr::{Dirichlet, Distribution, Gumbel};
use takzero::network::{repr::move_index, Network};
use takzero::search::{agent::Agent, env::Terminal};
use takzero_hip_sys as sys;
use tch::TchError;

pub const N: usize = 5;
pub const HALF_KOMI: i8 = 4;
pub type Env = fast_tak::Game<N, HALF_KOMI>;

fn check(rc: i32) -> Result<(), TchError> {
    if rc == sys::TZ_OK {
        return Ok(());
    }
    let msg = unsafe { CStr::from_ptr(sys::tz_last_error()) }.to_string_lossy().into_owned();
    Err(TchError::Torch(format!("libtakzero_hip: {msg} (code {rc})")))
}

fn cstr(path: &Path) -> CString {
    CString::new(path.to_string_lossy().as_bytes()).expect("path without NUL")
}

/// The fields of `fast_tak::Game` the engine reads (repr.rs:169-228, env.rs:39-63): `sq = row * N + column`.
pub fn pack_state(game: &Env) -> sys::TzState {
    let mut s = sys::TzState {
        colors: [0; 36], height: [0; 36], top: [0; 36], stones: [0; 2], caps: [0; 2],
        to_move: game.to_move as u8, n: N as u8, half_komi: HALF_KOMI, pad0: 0,
        ply: game.ply as u16, reversible_plies: game.reversible_plies as u16,
    };
    for (sq, stack) in game.board.iter().flatten().enumerate() {
        for (i, colour) in stack.colors().enumerate() {           // bottom to top
            s.colors[sq] |= (colour as u64) << i;
        }
        s.height[sq] = stack.size() as u8;
        s.top[sq] = stack.top().map_or(sys::TZ_EMPTY as u8, |(piece, _)| piece as u8 + 1);
    }
    s.stones = [game.white_reserves.stones, game.black_reserves.stones];
    s.caps = [game.white_reserves.caps, game.black_reserves.caps];
    s
}

pub struct HipNet {
    raw: *mut sys::TzNet,
}
unsafe impl Send for HipNet {}

impl HipNet {
    /// `precision`: `sys::TZ_PREC_F16` is the throughput default (logits ~1e-3 *relative* to the fp32 LibTorch path); where the
    /// north star's absolute 1e-3 on a trained net's logits is wanted pass `sys::TZ_PREC_F16C6` (1.4e-4, 2.1x the kernel time),
    /// `TZ_PREC_F16C8` (1.3e-4, 2.3x) or `TZ_PREC_F16X2` (2.4e-5, 3.1x) - same files, same calls.
    pub fn with_precision(seed: Option<i64>, precision: i32, device: i32) -> Self {
        let mut raw = std::ptr::null_mut();
        check(unsafe { sys::tz_net_create(N as i32, sys::TZ_ARCH_NET5, device, precision, 0, &mut raw) }).unwrap();
        check(unsafe { sys::tz_net_init_random(raw, seed.unwrap_or(0) as u64) }).unwrap();   // tch's default initialisers
        Self { raw }
    }
    pub fn raw(&self) -> *mut sys::TzNet {
        self.raw
    }
}

impl Drop for HipNet {
    fn drop(&mut self) {
        unsafe { sys::tz_net_destroy(self.raw) };
    }
}

impl Network for HipNet {
    // network/mod.rs:10-45, all five members without LibTorch in the process
    fn new(_device: tch::Device, seed: Option<i64>) -> Self {
        Self::with_precision(seed, sys::TZ_PREC_F16, 0)
    }
    fn load(path: impl AsRef<Path>, device: tch::Device) -> Result<Self, TchError> {
        let nn = Self::new(device, None);
        // `model_latest.ot` exactly as `learn` wrote it with VarStore::save: the library reads the LibTorch archive itself; a SimHash
        // net also picks up bitvec.bin beside it.  A failed load returns Err and the caller keeps the old net (selfplay/src/main.rs:107-120).
        check(unsafe { sys::tz_net_load_weights(nn.raw, cstr(path.as_ref()).as_ptr()) })?;
        Ok(nn)
    }
    fn load_partial(path: impl AsRef<Path>, device: tch::Device) -> Result<Self, TchError> {
        let nn = Self::new(device, None);
        check(unsafe { sys::tz_net_load_partial(nn.raw, cstr(path.as_ref()).as_ptr(), std::ptr::null_mut(), 0, std::ptr::null_mut()) })?;
        Ok(nn)
    }
    fn save(&self, path: impl AsRef<Path>) -> Result<(), TchError> {
        check(unsafe { sys::tz_net_save(self.raw, cstr(path.as_ref()).as_ptr()) })
    }
    fn clone(&self, _device: tch::Device) -> Self {
        let mut raw = std::ptr::null_mut();
        check(unsafe { sys::tz_net_clone(self.raw, 0, &mut raw) }).unwrap();
        Self { raw }
    }
}

impl Agent<Env> for HipNet {
    fn policy_value_uncertainty(
        &self,
        env_batch: &[Env],
        actions_batch: &[Vec<Move>],
    ) -> impl Iterator<Item = (Vec<(Move, NotNan<f32>)>, f32, f32)> {
        assert_eq!(env_batch.len(), actions_batch.len());
        assert!(!env_batch.is_empty());                                   // net5.rs:226-227
        let amax = actions_batch.iter().map(Vec::len).max().unwrap_or(1).max(1);
        let states: Vec<sys::TzState> = env_batch.iter().map(pack_state).collect();
        let mut idx = vec![0u16; env_batch.len() * amax];
        let counts: Vec<i32> = actions_batch
            .iter()
            .enumerate()
            .map(|(i, a)| {
                for (j, m) in a.iter().enumerate() {
                    idx[i * amax + j] = move_index::<N>(m) as u16;      // repr.rs:49-71
                }
                a.len() as i32
            })
            .collect();
        let (mut logits, mut value, mut var) = (vec![0f32; idx.len()], vec![0f32; counts.len()], vec![0f32; counts.len()]);
        check(unsafe {
            sys::tz_net_eval(self.raw, counts.len() as i32, states.as_ptr(), idx.as_ptr(), counts.as_ptr(), amax as i32,
                             logits.as_mut_ptr(), value.as_mut_ptr(), var.as_mut_ptr())
        })
        .expect("tz_net_eval");
        actions_batch
            .iter()
            .enumerate()
            .map(move |(i, a)| {
                (a.iter().enumerate().map(|(j, m)| (*m, NotNan::new(logits[i * amax + j]).expect("logit should not be NaN"))).collect(),
                 value[i], var[i])
            })
            .collect::<Vec<_>>()
            .into_iter()
    }
}

/// `BatchedMCTS<B, Env>` (batched.rs:24-30) with the trees, the environments and the net on the device.
pub struct HipBatchedMCTS<const B: usize> {
    raw: *mut sys::TzSearch,
    amax: usize,
}

impl<const B: usize> Drop for HipBatchedMCTS<B> {
    fn drop(&mut self) {
        unsafe { sys::tz_search_destroy(self.raw) };
    }
}

impl<const B: usize> HipBatchedMCTS<B> {
    /// batched.rs:33-37: fresh games from random openings (env.rs:65-79: one of 8 symmetries x {adjacent, opposite})
    pub fn new(net: &HipNet, rng: &mut impl Rng) -> Self {
        let mut raw = std::ptr::null_mut();
        check(unsafe { sys::tz_search_create(net.raw, sys::TZ_AGENT_NET, B as i32, N as i32, HALF_KOMI as i32, 0, &mut raw) }).unwrap();
        let choice: [i32; B] = std::array::from_fn(|_| rng.random_range(0..16));
        check(unsafe { sys::tz_search_new_openings(raw, choice.as_ptr()) }).unwrap();
        let (mut batch, mut board_n, mut half_komi, mut amax) = (0i32, 0i32, 0i32, 0i32);
        check(unsafe { sys::tz_search_shape(raw, &mut batch, &mut board_n, &mut half_komi, &mut amax) }).unwrap();
        Self { raw, amax: amax as usize }
    }
    /// batched.rs:39-47 + reanalyze/src/main.rs:159-165: overwrite the environments, fresh trees
    pub fn set_envs(&mut self, envs: &[Env; B]) {
        let idx: [i32; B] = std::array::from_fn(|i| i as i32);
        let states: Vec<sys::TzState> = envs.iter().map(pack_state).collect();
        check(unsafe { sys::tz_search_set_positions(self.raw, B as i32, idx.as_ptr(), states.as_ptr()) }).unwrap();
    }
    /// batched.rs:63-128
    pub fn simulate(&mut self, betas: &[f32]) {
        check(unsafe { sys::tz_search_simulate(self.raw, betas.as_ptr(), 1) }).unwrap();
    }
    pub fn root_info(&self) -> Vec<sys::TzRootInfo> {
        let zero = sys::TzRootInfo { visit_count: 0, n_children: 0, eval_tag: 0, is_terminal_env: 0, ply: 0, eval_bits: 0, std_dev: 0.0, logit: 0.0, probability: 0.0 };
        let mut out = vec![zero; B];
        check(unsafe { sys::tz_search_root_info(self.raw, out.as_mut_ptr()) }).unwrap();
        out
    }
    /// noise.rs:10-26 for every root: the same draws in the same order as `apply_noise` of the reference
    pub fn apply_noise(&mut self, rng: &mut impl Rng, alpha: f32, ratio: f32) {
        let info = self.root_info();
        let mut noise = vec![0f32; B * self.amax];
        for (g, r) in info.iter().enumerate() {
            if r.n_children < 2 {
                continue;
            }
            let d = Dirichlet::new(&vec![alpha; r.n_children as usize]).unwrap().sample(rng);
            noise[g * self.amax..g * self.amax + d.len()].copy_from_slice(&d);
        }
        check(unsafe { sys::tz_search_apply_noise(self.raw, noise.as_ptr(), self.amax as i32, ratio) }).unwrap();
    }
    /// batched.rs:207-409: the Gumbel draws (one per child, in child order) are made here
    pub fn gumbel_sequential_halving(&mut self, betas: &[f32], k: usize, budget: u32, rng: &mut impl Rng) -> [u16; B] {
        let info = self.root_info();
        let gumbel_dist = Gumbel::new(0.0f32, 1.0).unwrap();
        let mut gumbel = vec![0f32; B * self.amax];
        for (g, r) in info.iter().enumerate() {
            for c in 0..r.n_children as usize {
                gumbel[g * self.amax + c] = gumbel_dist.sample(rng);
            }
        }
        let mut selected = [0u16; B];
        check(unsafe {
            sys::tz_search_gumbel_sh(self.raw, betas.as_ptr(), k as i32, budget as i32, gumbel.as_ptr(), self.amax as i32, selected.as_mut_ptr())
        })
        .unwrap();
        selected
    }
    /// batched.rs:146-163 (move indices as repr.rs:49-71 numbers them)
    pub fn select_best_actions(&self) -> [u16; B] {
        let mut out = [0u16; B];
        check(unsafe { sys::tz_search_select_best_actions(self.raw, out.as_mut_ptr()) }).unwrap();
        out
    }
    /// node/mod.rs:132-161 for every root: policy target over the children, in child order
    pub fn improved_policy(&self, visitations: f32) -> Vec<f32> {
        let mut out = vec![0f32; B * self.amax];
        check(unsafe { sys::tz_search_improved_policy(self.raw, visitations, self.amax as i32, out.as_mut_ptr()) }).unwrap();
        out
    }
    pub fn ube_target(&self, beta: f32) -> [f32; B] {
        let mut out = [0f32; B];
        check(unsafe { sys::tz_search_ube_target(self.raw, beta, out.as_mut_ptr()) }).unwrap();
        out
    }
    /// batched.rs:131-144: subtree reuse + the move on every board (terminal roots are skipped inside)
    pub fn step(&mut self, actions: &[u16; B]) {
        check(unsafe { sys::tz_search_step(self.raw, actions.as_ptr()) }).unwrap();
    }
    /// batched.rs:185-203: finished games are restarted from fresh openings; what they ended with comes back
    pub fn restart_terminal_envs(&mut self, rng: &mut impl Rng) -> [Option<Terminal>; B] {
        let choice: [i32; B] = std::array::from_fn(|_| rng.random_range(0..16));
        let mut terminal = [sys::TZ_TERMINAL_NONE as i8; B];
        check(unsafe { sys::tz_search_restart_terminal(self.raw, choice.as_ptr(), terminal.as_mut_ptr()) }).unwrap();
        terminal.map(|t| match t as i32 {
            sys::TZ_TERMINAL_WIN => Some(Terminal::Win),
            sys::TZ_TERMINAL_LOSS => Some(Terminal::Loss),
            sys::TZ_TERMINAL_DRAW => Some(Terminal::Draw),
            _ => None,
        })
    }
}
