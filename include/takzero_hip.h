/*
 * takzero_hip.h — C ABI of libtakzero_hip.so, the MI355X (gfx950) engine for the
 * takzero self-play / reanalyze hot path (batched MCTS + policy/value net forward).
 *
 * The reference (ViliamVadocz/takzero) has no FFI of its own for this path: its seams are
 * the Rust traits `Agent` (takzero/src/search/agent.rs:5-14), `Network`
 * (takzero/src/network/mod.rs:10-45) and the struct `BatchedMCTS`
 * (takzero/src/search/node/batched.rs:24-409).  Every entry point below names the
 * reference item it replaces.  A Rust adapter maps these calls back
 * onto the reference's types.
 *
 * Conventions
 *   - every function returns 0 on success, a negative TZ_E* code otherwise; the message
 *     is available from tz_last_error() (thread local).
 *   - caller owns every host buffer; the library owns device memory behind the handles.
 *   - one handle is driven from one host thread (the reference is single threaded,
 *     batched.rs:21-22); calls block until results are on the host unless noted.
 *   - randomness never crosses the ABI: Dirichlet / Gumbel / opening choices are inputs
 *     (the reference threads one StdRng through its calls; keeping the draws on the
 *     caller side is the only way an adapter can reproduce its stream).
 *   - moves are identified by the reference's policy index `move_index`
 *     (takzero/src/network/repr.rs:49-71): channel * N*N + row * N + column.
 *   - squares are indexed  sq = row * N + column  (row = rank-1, column = file).
 */
#ifndef TAKZERO_HIP_H
#define TAKZERO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TZ_MAX_N 6
#define TZ_MAX_SQUARES 36
/* upper bound on legal moves in one position for board sizes 3..5 (rows of 512); a 6x6 handle uses 1024: ask
 * tz_search_shape for the row width of a given handle */
#define TZ_MAX_ACTIONS 512
#define TZ_MAX_ACTIONS_6 1024

/* error codes */
#define TZ_OK 0
#define TZ_EINVAL (-1)   /* bad argument (size, index, null)                           */
#define TZ_EPARSE (-2)   /* malformed TPS / PTN / weight file                          */
#define TZ_EDEVICE (-3)  /* HIP runtime failure (message carries hipGetErrorString)    */
#define TZ_ENOMEM (-4)   /* host or device allocation failed                           */
#define TZ_ECAPACITY (-5)/* a per-game node pool or trajectory buffer overflowed       */
#define TZ_ESTATE (-6)   /* call not valid in the current state (e.g. noise on an      */
                         /* un-expanded root — batched.rs / noise.rs:12-15 assert)     */
#define TZ_ENUMERIC (-7) /* NaN produced by the net (reference panics: net5.rs:263)    */

/* piece type of the top of a stack */
#define TZ_EMPTY 0
#define TZ_FLAT 1
#define TZ_WALL 2
#define TZ_CAP 3

/* Eval tags, takzero/src/search/eval.rs:8-13 */
#define TZ_EVAL_VALUE 0
#define TZ_EVAL_WIN 1
#define TZ_EVAL_LOSS 2
#define TZ_EVAL_DRAW 3

/* Terminal, takzero/src/search/env.rs:27-31 (from the side to move) */
#define TZ_TERMINAL_NONE (-1)
#define TZ_TERMINAL_WIN 0
#define TZ_TERMINAL_LOSS 1
#define TZ_TERMINAL_DRAW 2

/*
 * Packed game state = the fields of fast_tak::Game<N,HALF_KOMI> that the reference reads
 * (repr.rs:169-228, env.rs:39-63).  Plain data, 376 bytes.
 */
typedef struct tz_state {
    uint64_t colors[TZ_MAX_SQUARES]; /* bit i = colour of the i-th piece from the bottom (0 white, 1 black) */
    uint8_t height[TZ_MAX_SQUARES];  /* number of pieces on the square                                     */
    uint8_t top[TZ_MAX_SQUARES];     /* TZ_EMPTY / TZ_FLAT / TZ_WALL / TZ_CAP                               */
    uint8_t stones[2];               /* remaining stones  [white, black]                                    */
    uint8_t caps[2];                 /* remaining capstones [white, black]                                  */
    uint8_t to_move;                 /* 0 white, 1 black                                                    */
    uint8_t n;                       /* board size 3..6                                                     */
    int8_t half_komi;                /* komi for black in half points (net5.rs:18 uses 4)                   */
    uint8_t pad0;
    uint16_t ply;
    uint16_t reversible_plies;
} tz_state;

typedef struct tz_net tz_net;       /* opaque: a network on one GPU  (Network + Agent impl)   */
typedef struct tz_search tz_search; /* opaque: BatchedMCTS<B, Game<N,HALF_KOMI>> on one GPU   */

/* architectures (takzero/src/network/{net4_simhash,net5,net6_simhash}.rs) */
#define TZ_ARCH_NET4_SIMHASH 4
#define TZ_ARCH_NET5 5
#define TZ_ARCH_NET6_SIMHASH 6
/* test architecture: same graph as net5 on any board size with configurable depth, no RND/hash */
#define TZ_ARCH_TEST 100

/* arithmetic of the trunk */
#define TZ_PREC_BF16 0 /* NHWC bf16 activations/weights, fp32 accumulate on MFMA: 5 % faster, logits 1e-3..7e-3 off fp32 */
#define TZ_PREC_F32 1  /* fp32 everywhere on plain FMA kernels (validation path)                          */
#define TZ_PREC_F16 2  /* the throughput default: the same MFMA kernels with IEEE fp16 storage (saturating), fp32
                          accumulate: logits within ~2e-4 relative of fp32 (1.5e-4 absolute at random-init scale) */
#define TZ_PREC_F16X2 3 /* split precision: every operand a hi/lo pair of halves (22-bit significand), three fp16 MFMAs
                          per product with fp32 accumulate: logits within 1e-3 absolute of the fp32 LibTorch graph at
                          trained logit scale (|logit| ~ 10), the north star's tolerance; ~3x the MFMA work */
#define TZ_PREC_F16C8 4 /* fp16 products with FP8 corrections: the main product wh*xh as in TZ_PREC_F16, the two correction
                          products wl*xh + wh*xl of the split form on OCP FP8 (E4M3) copies of the operands through
                          v_mfma_f32_16x16x128_f8f6f4 (twice the fp16 rate): a correction only has to be good to a few
                          bits.  Logits within 1e-3 absolute at trained logit scale like TZ_PREC_F16X2 (about 30x
                          closer to fp32 than TZ_PREC_F16), ~2x the MFMA time of TZ_PREC_F16 instead of 3x */
#define TZ_PREC_F16C6 5 /* the same with the corrections on FP6 (E2M3) copies and one power-of-two scale per block of 32 input
                          channels, through v_mfma_scale_f32_16x16x128_f8f6f4 at four times the fp16 rate (round 3): 1.5x the
                          MFMA time of TZ_PREC_F16, 8 boards per workgroup like it, the same logit error as TZ_PREC_F16C8.
                          5x5 and 6x6 networks. */

/* built-in agents for tz_search_create (takzero/src/search/agent.rs:16-87) */
#define TZ_AGENT_NET 0
#define TZ_AGENT_DUMMY 1
#define TZ_AGENT_SIMPLE 2

const char* tz_last_error(void);
int tz_version(void);
int tz_device_count(void);

/* ---------- text / index helpers (fast-tak / takparse formats) ---------- */
int tz_state_from_tps(const char* tps, int n, int half_komi, tz_state* out);
int tz_state_to_tps(const tz_state* s, char* buf, int buflen);
int tz_move_to_ptn(int n, uint16_t move_index, char* buf, int buflen);
int tz_move_from_ptn(int n, const char* ptn, uint16_t* move_index_out);
/* number of policy logits for board size n: output_size::<N>(), repr.rs:119-121 */
int tz_policy_size(int n);
/* number of input planes: input_channels::<N>(), repr.rs:137-142 */
int tz_input_channels(int n);

/* ---------- Network lifecycle: Network::{new, save, load, load_partial, clone}  (network/mod.rs:10-45) ---------- */
/* blocks = number of residual blocks for TZ_ARCH_TEST (ignored otherwise). */
int tz_net_create(int board_n, int arch, int device_id, int precision, int blocks, tz_net** out);
/* Network::new(device, seed) (network/mod.rs:11; net5.rs:152-170): tch's default initialisers, draws from a generator
 * keyed by (seed, variable name); the net is usable afterwards. */
int tz_net_init_random(tz_net* net, uint64_t seed);
/* Network::load (network/mod.rs:24-28; net6_simhash.rs:173-190).  The file is recognised by its content: a LibTorch
 * archive as tch's VarStore::save writes it (`model_latest.ot`, `model_NNNNNNN.ot`: read natively, no LibTorch needed) or
 * the flat .tzw container of takzero_amd.weights (named fp32 tensors).  SimHash nets also pick up `bitvec.bin` from the
 * same directory when it is there.  A failed load leaves the previous weights active (selfplay/src/main.rs:112-115). */
int tz_net_load_weights(tz_net* net, const char* path);
/* The same load in two halves, for a host that does not want to stop playing while a new model_latest.ot is read (selfplay's
 * hot reload, selfplay/src/main.rs:107-120).  tz_net_load_prepare parses the file and builds the device weights in fresh buffers;
 * it touches nothing of the live network and may run on another thread while `net` is evaluating.  tz_net_load_commit puts them
 * in place (a device synchronisation and a pointer swap; for a SimHash net also the bitvec.bin beside the file) and must be
 * called where tz_net_load_weights could be: not concurrently with a forward of `net`.  A failed prepare leaves nothing behind;
 * prepared weights that are not wanted any more go to tz_net_load_discard. */
typedef struct tz_pending_weights tz_pending_weights;
int tz_net_load_prepare(tz_net* net, const char* path, tz_pending_weights** out);
int tz_net_load_commit(tz_net* net, tz_pending_weights* pending);
int tz_net_load_discard(tz_pending_weights* pending);
int tz_net_load_weights_mem(tz_net* net, const void* data, size_t bytes);
/* Network::load_partial (network/mod.rs:30-35): variables the file does not hold (or holds with another size) keep their
 * values; their names come back newline-separated in missing_out (may be NULL), their number in n_missing_out. */
int tz_net_load_partial(tz_net* net, const char* path, char* missing_out, int missing_cap, int* n_missing_out);
/* Network::save (network/mod.rs:16-18; net6_simhash.rs:152-170): `*.tzw` = the flat container, anything else = a LibTorch
 * archive with tch's variable names, written to `<path>.part` and renamed; SimHash nets write `bitvec.bin` beside it. */
int tz_net_save(tz_net* net, const char* path);
/* Network::clone(device) (network/mod.rs:37-44): the same variables (and SimHash set) on another GPU. */
int tz_net_clone(tz_net* net, int device_id, tz_net** out);
/* One variable of the host-side VarStore by name (`.a.` / `.b.` for the two SmallBlocks of a block); out may be NULL to
 * ask for the element count only. */
int tz_net_get_tensor(tz_net* net, const char* name, float* out, uint64_t count, uint64_t* count_out);
/* enumeration of that store, in name order (as tz_trainer_tensor_count / _info) */
int tz_net_tensor_count(tz_net* net);
int tz_net_tensor_info(tz_net* net, int i, char* name_out, int name_cap, uint64_t* count_out);
/* Converts a model file between the two containers (by content on the way in, by extension on the way out). */
int tz_weights_convert(const char* src, const char* dst);
int tz_net_destroy(tz_net* net);

/*
 * Agent::policy_value_uncertainty  (agent.rs:5-14; net5.rs:220-285)
 *   states[batch]; legal_idx[batch][amax] = move_index of each legal action in the caller's
 *   order, legal_count[batch] <= amax.
 *   logits_out[batch][amax]: raw logits of exactly those actions in that order (entries past
 *   legal_count are 0); value_out[batch] in [-1,1] from the side to move;
 *   variance_out[batch] in [0,4].  batch == 0 is TZ_EINVAL (net5.rs:226-227 asserts).
 */
int tz_net_eval(tz_net* net, int batch, const tz_state* states, const uint16_t* legal_idx,
                const int32_t* legal_count, int amax, float* logits_out, float* value_out,
                float* variance_out);
/* Debug path: the input planes game_repr writes (repr.rs:169-228), NCHW fp32,
 * planes_out[batch][channels*n*n], computed on the device by the same encoder. */
int tz_net_encode(tz_net* net, int batch, const tz_state* states, float* planes_out);
/* Full policy tensor [batch][policy_size] in the reference's NCHW flattening (net5.rs:238). */
int tz_net_forward_raw(tz_net* net, int batch, const tz_state* states, float* policy_out,
                       float* value_out, float* ube_out);

/* SimHash nets (net4_simhash / net6_simhash): HashNetwork::get_indices / update_counts
 * (net6_simhash.rs:202-243) and the bitvec.bin file that accompanies a model (net6_simhash.rs:152-190). */
int tz_net_hash_indices(tz_net* net, int batch, const tz_state* states, uint32_t* indices_out, int update);
int tz_net_load_bitset(tz_net* net, const char* path);
int tz_net_save_bitset(tz_net* net, const char* path);

/* ---------- BatchedMCTS (search/node/batched.rs:32-409) ---------- */
/* BatchedMCTS::from_envs with default (empty-board) envs; node_capacity = node slots per game
 * (0 = default).  net may be NULL when agent_kind != TZ_AGENT_NET. */
int tz_search_create(tz_net* net, int agent_kind, int batch, int board_n, int half_komi,
                     int node_capacity, tz_search** out);
int tz_search_destroy(tz_search* s);
/* BATCH_SIZE, N, HALF_KOMI of the handle and the widest child list it can hold */
int tz_search_shape(tz_search* s, int* batch_out, int* board_n_out, int* half_komi_out, int* max_actions_out);
/* nodes_and_envs_mut: overwrite envs and reset their trees (reanalyze/src/main.rs:159-165). */
int tz_search_set_positions(tz_search* s, int count, const int32_t* game_idx, const tz_state* states);
int tz_search_get_positions(tz_search* s, tz_state* states_out /*[batch]*/);
/* BatchedMCTS::new openings: env.rs:65-79.  opening_choice[g] in [0,16): symmetry*2 + opposite. */
int tz_search_new_openings(tz_search* s, const int32_t* opening_choice);
/* BatchedMCTS::simulate called n_sims times (batched.rs:63-128); betas[batch]. Asynchronous
 * on the handle's stream; any later call that returns data synchronises. */
int tz_search_simulate(tz_search* s, const float* betas, int n_sims);
/* BatchedMCTS::apply_noise with caller-supplied Dirichlet samples (noise.rs:10-26):
 * noise[batch][amax], row g holds one sample of dimension n_children(g). */
int tz_search_apply_noise(tz_search* s, const float* noise, int amax, float ratio);
/* per-root summary: evaluation, visit_count, std_dev, number of children, terminal flag */
typedef struct tz_root_info {
    uint32_t visit_count;
    uint32_t n_children;
    uint8_t eval_tag;
    uint8_t is_terminal_env; /* env.terminal().is_some() */
    uint16_t ply;
    union { float value; uint32_t ply; } eval;
    float std_dev;
    float logit;
    float probability;
} tz_root_info;
int tz_search_root_info(tz_search* s, tz_root_info* out /*[batch]*/);
/* children of every root in fast-tak possible_moves order (callers zip by position:
 * selfplay/src/main.rs:249-253).  Arrays are [batch][amax]; any pointer may be NULL. */
int tz_search_root_children(tz_search* s, int amax, uint16_t* move_idx, uint32_t* visits,
                            uint8_t* eval_tag, uint32_t* eval_bits, float* logit, float* prob,
                            float* std_dev);
/* Below the root: the reference's Node.children is a public field (node/mod.rs:14-23) that puzzle and visualize_search read.
 * The node reached from game `game`'s root by the moves path[0..path_len) (move indices; path_len 0 = the root): node_out = its
 * statistics (ply = the root's ply + path_len; is_terminal_env = Node::is_terminal), children rows of width amax as in
 * tz_search_root_children (any pointer may be NULL).  TZ_EINVAL if the path leaves the tree. */
int tz_search_node(tz_search* s, int game, const uint16_t* path, int path_len, tz_root_info* node_out, int amax, uint16_t* move_idx,
                   uint32_t* visits, uint8_t* eval_tag, uint32_t* eval_bits, float* logit, float* prob, float* std_dev);
/* Node::select_best_action per root (node/mod.rs:132-161). 0xFFFF for roots without children. */
int tz_search_select_best_actions(tz_search* s, uint16_t* actions_out /*[batch]*/);
/* Node::improved_policy(visitations) per root (policy.rs:23-48), [batch][amax]. */
int tz_search_improved_policy(tz_search* s, float visitations, int amax, float* policy_out);
/* the same with one visitation count per game: reanalyze passes each root's most_visited_count()
 * (reanalyze/src/main.rs:196-202) */
int tz_search_improved_policy_each(tz_search* s, const float* visitations, int amax, float* policy_out);
/* Node::ube_target(beta) per root (node/mod.rs:215-230). */
int tz_search_ube_target(tz_search* s, float beta, float* out /*[batch]*/);
/* BatchedMCTS::step (batched.rs:131-144): descend (subtree reuse) + env.step; skipped for
 * terminal roots.  actions[batch] are move indices. */
int tz_search_step(tz_search* s, const uint16_t* actions);
/* BatchedMCTS::restart_terminal_envs (batched.rs:185-203): finished games get a fresh opening
 * (opening_choice as above) and a fresh tree; terminal_out[g] = TZ_TERMINAL_* of the game
 * that ended, TZ_TERMINAL_NONE otherwise. */
int tz_search_restart_terminal(tz_search* s, const int32_t* opening_choice, int8_t* terminal_out);
/* Details of the games that ended in the last tz_search_restart_terminal call: reason_out[g] = 1 road,
 * 2 flat count, 3 reversible-plies draw (0 if the game did not end); winner_out[g] = 0 white, 1 black, 2 draw.
 * Needed to write the PTN result of a Replay line (target.rs:215-232). */
int tz_search_terminal_details(tz_search* s, int8_t* reason_out, uint8_t* winner_out);
/* Validated move application without search (Replay::from_str / Replay::states, target.rs:205-212,248-268):
 * actions[g] (0xFFFF = none) is applied to game g iff it is legal there; ok_out[g] = 1 applied, 0 illegal or
 * none, -1 position already terminal.  Trees are reset. */
int tz_search_play_moves(tz_search* s, const uint16_t* actions, int8_t* ok_out);
/* BatchedMCTS::gumbel_sequential_halving with caller-supplied Gumbel(0,1) samples
 * (batched.rs:207-409): gumbel[batch][amax]; selected_out[batch] move indices. */
int tz_search_gumbel_sh(tz_search* s, const float* betas, int sampled_actions, int search_budget,
                        const float* gumbel, int amax, uint16_t* selected_out);
/* counters since creation: simulations (incl. Known hits) and network-evaluated leaves */
int tz_search_counters(tz_search* s, uint64_t* simulations, uint64_t* nn_leaf_evals);
/* The reference's trees are heap allocated and unbounded; a game's node pool here has `node_capacity` slots (default
 * 262 144 on 5x5: 66 GB for 4096 games).  When a leaf cannot get its children because the pool is full it is evaluated
 * and backed up without being expanded (it stays a leaf; the next tz_search_step compacts the kept subtree and frees
 * the rest), and the event is counted here.  With TZ_STRICT_CAPACITY set in the environment at creation a full pool is
 * TZ_ECAPACITY instead. */
int tz_search_pool_overflows(tz_search* s, uint64_t* skipped_expansions);
/* node slots in use in the fullest game's pool, and the per-game capacity */
int tz_search_pool_usage(tz_search* s, uint32_t* max_used, uint32_t* capacity);
int tz_search_sync(tz_search* s);
/* time spent (ms, HIP events on the handle's stream) in the dominant conv kernel and its
 * launch count since the last reset; used by bench.py for the roofline line. */
int tz_search_profile(tz_search* s, int reset, double* conv_ms, uint64_t* conv_launches,
                      double* tree_ms, uint64_t* steps);

/* ---------- selfplay::main above the search (selfplay/src/main.rs:63-387), native host code (csrc/tz_host.cpp) ----------
 * The outer loop of the reference's selfplay binary: search, move choice, take_a_step, restart_envs_and_complete_targets,
 * target / replay lines, buffer_lengths.txt back-pressure, append-only files.  Draws (openings, Dirichlet, Gumbel, early
 * move sampling) come from one seeded generator per (seed, shard).
 * search_kind: 0 PUCT + Dirichlet (:127-136), 1 Gumbel sequential halving (:138-153), 2 uniformly random moves
 * (learn's pre-training games, learn/src/main.rs:437-445).  exploration: cargo feature of that name (:79-86). */
typedef struct tz_selfplay tz_selfplay;
int tz_selfplay_create(tz_search* search, int sims_per_move, uint64_t seed, int shard, int search_kind, int sampled_actions,
                       int exploration, tz_selfplay** out);
int tz_selfplay_destroy(tz_selfplay* sp);
int tz_selfplay_play_move(tz_selfplay* sp);
int tz_selfplay_counters(tz_selfplay* sp, uint64_t* moves_out, uint64_t* targets_out, uint64_t* replays_out);
/* which: 0 target lines, 1 replay lines, 2 exploration replay lines finished since the last call */
int tz_selfplay_take_text(tz_selfplay* sp, int which, char* out, uint64_t cap, uint64_t* size_out);
/* the directory loop; reload(user) is called before every move (Net::load of model_latest, :107-121), may be NULL */
int tz_selfplay_run(tz_selfplay* sp, const char* directory, int moves, int max_buffer_len, const char* suffix,
                    int (*reload)(void*), void* reload_user, double wait_limit_s);

/* ---------- N shards: the exchange between the self-play processes of one job, csrc/tz_comm.cpp ----------
 * The reference runs N selfplay processes that append to the same files of one directory (README.md:130) and has no
 * collective.  With one process per GPU the shards hand over through a communicator: an all-gather of counts followed by
 * an all-gather of the packed target records / replay lines after every move, and a broadcast of a new model's weights.
 * Transports: RCCL (ncclAllGather / ncclBroadcast on the shard's GPU, xGMI inside a node; librccl is opened with dlopen on
 * first use) and "fs" (files in a shared directory, the reference's own medium: no GPU needed). */
typedef struct tz_comm tz_comm;
#define TZ_COMM_ID_BYTES 128
/* ncclGetUniqueId on one rank; the host carries the 128 bytes to the others (MPI, a TCP store, or the helper below) */
int tz_comm_unique_id(unsigned char* id_out /*[TZ_COMM_ID_BYTES]*/);
/* the same through `<directory>/rccl_id.bin`: rank 0 creates and publishes the id, the others wait for it */
int tz_comm_rendezvous_id(const char* directory, int rank, unsigned char* id_inout, double timeout_s);
/* ncclCommInitRank for this shard (collective over the world); device_id = the shard's GPU */
int tz_comm_create_rccl(const unsigned char* id, int rank, int world, int device_id, tz_comm** out);
int tz_comm_create_fs(const char* directory, int rank, int world, double timeout_s, tz_comm** out);
int tz_comm_destroy(tz_comm* c);
int tz_comm_info(tz_comm* c, int* rank_out, int* world_out, int* is_rccl_out, uint64_t* collectives_out, uint64_t* bytes_gathered_out);
/* variable-size all-gather of host bytes (counts, then padded payloads): sizes_out[world]; the payloads, back to back in
 * rank order (total_out bytes), stay with the handle until tz_comm_take copies them out */
int tz_comm_all_gather(tz_comm* c, const void* data, uint64_t bytes, uint64_t* sizes_out, uint64_t* total_out);
int tz_comm_take(tz_comm* c, void* out, uint64_t out_cap);
int tz_comm_broadcast(tz_comm* c, void* data, uint64_t bytes, int root);
int tz_comm_barrier(tz_comm* c);
/* Net::load for N shards (selfplay/src/main.rs:107-121): the root passes the result of its own load as `status` (0 = a new
 * model is active on it); every rank then takes the same branch: 0 = receive and activate those variables, else keep. */
int tz_net_broadcast(tz_net* net, tz_comm* c, int root, int status);
/* Self-play over N shards: with a communicator set, the targets a move finishes stay packed until tz_selfplay_exchange
 * (collective, once per tz_selfplay_play_move; tz_selfplay_run calls it) has gathered every rank's; afterwards rank
 * `writer_rank` (every rank if < 0) holds everybody's target / replay / exploration lines in rank order — for
 * tz_selfplay_take_text, or appended to the directory's files by tz_selfplay_run — and the other ranks hold none. */
int tz_selfplay_set_comm(tz_selfplay* sp, tz_comm* c, int writer_rank);
int tz_selfplay_exchange(tz_selfplay* sp);

/* ---------- reanalyze::main above the search (reanalyze/src/main.rs:60-290), native host code ----------
 * Position buffer fed from replays.txt (complete lines appended since the last call; line i belongs to rank i % world;
 * every pre-move state of a replay, moves re-validated on the device), B positions sampled without replacement,
 * fresh trees, search (0 = `sims` PUCT simulations, 1 = Gumbel sequential halving with budget `sims`), one target per
 * position: value = root evaluation if solved else -evaluation(selected child), policy =
 * improved_policy(most_visited_count()), ube = ube_target(0.25). */
typedef struct tz_reanalyze tz_reanalyze;
int tz_reanalyze_create(tz_search* search, int sims, uint64_t seed, int rank, int world, int search_kind, int sampled_actions,
                        tz_reanalyze** out);
int tz_reanalyze_destroy(tz_reanalyze* ra);
int tz_reanalyze_feed(tz_reanalyze* ra, const char* replays_path, uint64_t* added_out, uint64_t* total_out);
int tz_reanalyze_iterate(tz_reanalyze* ra);
int tz_reanalyze_take_text(tz_reanalyze* ra, char* out, uint64_t cap, uint64_t* size_out);
int tz_reanalyze_run(tz_reanalyze* ra, const char* directory, int iterations, int min_positions, const char* suffix,
                     int (*reload)(void*), void* reload_user, double wait_limit_s);

/* ---------- the other consumers of the search, native host code ----------
 * evaluation::compete (evaluation/src/main.rs:224-319): result_out[3] = wins, losses, draws for White. */
int tz_compete(tz_search* white, tz_search* black, const tz_state* games, float white_beta, float black_beta, uint64_t seed,
               int sampled_actions, int search_budget, int max_moves, int32_t* result_out);
/* puzzle benchmark (puzzle/src/main.rs:168-269): result_out[3] = attempted, solved, proven. */
int tz_puzzle_benchmark(tz_search* search, const tz_state* puzzles, const uint16_t* solutions, int count, int win, uint64_t seed,
                        int sampled_actions, int search_budget, int32_t* result_out);

/* ---------- Target lines in bulk (impl Display / FromStr for Target, target.rs:56-73, 99-143) ----------
 * "{tps};{value};{ube};{move}:{p},...\n" with Rust's `Display for f32`.  moves / policy are [count][amax]. */
int tz_format_targets(int n, int count, const tz_state* states, const uint16_t* moves, const float* policy,
                      const int32_t* nmoves, int amax, const float* value, const float* ube, char* out, uint64_t cap,
                      uint64_t* written_out);
/* parses the complete lines of text[0..len); unparsable lines are skipped (learn/src/main.rs:308);
 * consumed_out = bytes to advance the file offset by */
int tz_parse_targets(const char* text, uint64_t len, int n, int half_komi, int max_targets, int amax, tz_state* states,
                     uint16_t* moves, float* policy, int32_t* nmoves, float* value, float* ube, int32_t* count_out,
                     uint64_t* consumed_out, int32_t* skipped_out);

/* ---------- Trainer: the `learn` step (learn/src/main.rs:376-423), SURVEY.md 8f row 4 ----------
 * fp32 forward in training mode (BatchNorm batch statistics, running statistics updated with momentum 0.1),
 * masked log-softmax cross entropy + value MSE + UBE MSE, backward, Adam(lr) as tch's nn::Adam::default().
 * Tensors carry the VarStore names (`.a.` / `.b.` for the two SmallBlocks) and the reference's layouts.
 * batch must be a multiple of 64 (the reference uses 128, learn/src/main.rs:43). */
typedef struct tz_trainer tz_trainer;
int tz_trainer_create(int board_n, int arch, int device_id, int blocks, int batch, float learning_rate, tz_trainer** out);
int tz_trainer_destroy(tz_trainer* t);
int tz_trainer_tensor_count(tz_trainer* t);
int tz_trainer_tensor_info(tz_trainer* t, int i, char* name_out, int name_cap, uint64_t* count_out);
/* what: 0 parameter / buffer, 1 gradient of the last step, 2 / 3 Adam first / second moment */
int tz_trainer_set_tensor(tz_trainer* t, const char* name, int what, const float* data, uint64_t count);
int tz_trainer_get_tensor(tz_trainer* t, const char* name, int what, float* out, uint64_t count);
/* compute_loss_and_take_step: states[batch]; target_policy / mask [batch][tz_policy_size] (mask 1 = illegal,
 * move_mask); target_value[batch]; target_ube[batch] variances (ln + clamp to [-10, ln 4] applied inside,
 * learn/src/main.rs:360-363); train_ube = 0 in pre-training (:397-400); apply_step = 0 leaves the weights alone.
 * losses_out[3] = policy, value, ube. */
int tz_trainer_step(tz_trainer* t, const tz_state* states, const float* target_policy, const uint8_t* mask,
                    const float* target_value, const float* target_ube, int train_ube, int apply_step,
                    float* losses_out);
/* outputs of the last step's forward_t(xs, true): policy [batch][policy_size], value [batch], ube [batch] (log) */
int tz_trainer_outputs(tz_trainer* t, float* policy_out, float* value_out, float* ube_out);
/* what the last step's forward left after trunk layer `layer` (0 = input conv + BN + ReLU, then two per residual block, ReLU applied):
 * [batch][n*n][256] floats, pixel-major (NHWC).  out_cap in floats.  For tests: where an activation is zero the step's backward took
 * ReLU's derivative as zero — a comparison against another implementation can use the same side of the kink. */
int tz_trainer_activation(tz_trainer* t, int layer, float* out, uint64_t out_cap);

/* board size, batch and architecture of a trainer */
int tz_trainer_shape(tz_trainer* t, int* board_n_out, int* batch_out, int* arch_out);
/* Network::load / Network::save on the trainer's VarStore (learn/src/main.rs:107-120, 247-266): a LibTorch archive as tch
 * writes it or the .tzw container; variables the step never touches (RND nets, SimHash matrix) are carried through. */
int tz_trainer_load(tz_trainer* t, const char* path);
int tz_trainer_save(tz_trainer* t, const char* path);
/* the variables of a network (tz_net_init_random = Net::new, or a loaded model) become the trainer's, and back */
int tz_trainer_from_net(tz_trainer* t, tz_net* net);
int tz_trainer_to_net(tz_trainer* t, tz_net* net);

/* ---------- learn::main above the step (learn/src/main.rs:99-319, 486-516), native host code (csrc/tz_host_learn.cpp) ----------
 * The two target buffers with forced-use counts (SELFPLAY / REANALYZE_TARGET_FORCED_USES, :59-60) fed by tailing the
 * target files, create_batch (uniform sampling without replacement, a random board symmetry per target, dense
 * policy / mask tensors) and the training loop with buffer_lengths.txt.  which: 0 selfplay, 1 reanalyze. */
typedef struct tz_learn tz_learn;
int tz_learn_create(tz_trainer* trainer, int half_komi, uint64_t seed, int selfplay_forced_uses, int reanalyze_forced_uses,
                    tz_learn** out);
int tz_learn_destroy(tz_learn* l);
int tz_learn_feed(tz_learn* l, int which, const char* path, int model_steps, uint64_t* added_out);
int tz_learn_add_lines(tz_learn* l, int which, const char* text, uint64_t len, int model_steps, uint64_t* added_out);
int tz_learn_buffer_len(tz_learn* l, int which, uint64_t* len_out);
int tz_learn_step(tz_learn* l, int using_reanalyze, int train_ube, int augment, float* losses_out);
/* Diagnostic: the tensors of the batch tz_learn_step built last (any pointer may be NULL) */
int tz_learn_last_batch(tz_learn* l, tz_state* states_out, float* policy_out, uint8_t* mask_out, float* value_out, float* ube_out);
/* the main loop (:172-269); on_step(user, model_steps, losses[3], batch states, batch) runs after every step (save
 * points and SimHash update_counts live there); a non-zero return stops the loop */
int tz_learn_run(tz_learn* l, const char* directory, int64_t starting_steps, int64_t steps, int min_selfplay, int min_reanalyze,
                 int64_t steps_before_reanalyze, double read_interval_s, double sleep_s, double wait_limit_s,
                 int (*on_step)(void*, int64_t, const float*, const tz_state*, int), void* user, int64_t* model_steps_out);
/* The save points of learn::main inside tz_learn_run (learn/src/main.rs:247-266): `model_latest.ot` every steps_per_save
 * steps (100 in the reference), `model_<steps>.ot` every steps_per_checkpoint (50 000), as LibTorch archives written behind
 * the training loop; hash_net (may be NULL): a SimHash net whose set is updated with every batch (:418) and saved as
 * `bitvec.bin` beside the model.  0 = no such save point. */
int tz_learn_set_save_points(tz_learn* l, int steps_per_save, int steps_per_checkpoint, tz_net* hash_net);

/* Diagnostic: evaluates on the device the f32 primitives the tree kernels must compute exactly as
 * the host does (op 0 exp, 1 ln, 2 sqrt, 3 a/b, 4 0.997^int(a), 5 (a+b)*a). */
int tz_device_math(int op, const float* a, const float* b, float* out, int n);
/* Diagnostic: average ms per launch of the 5x5 residual-tower conv kernel on `positions` boards;
 * variant 0 = the shipped kernel, 1/2/3 = ablations (no LDS reads / no weight loads / neither). */
int tz_debug_conv_bench(tz_net* net, int variant, int positions, int iters, float* ms_out);
/* Diagnostic: the same for the fused residual-tower kernel (variant = its OPT bitmask). */
int tz_debug_tower_bench(tz_net* net, int variant, int positions, int iters, float* ms_out);
/* Diagnostic (builds with --ablations, TZ_NET_ABL=8): in-kernel shader clock of the last stamped launch of the net kernel
 * (median over workgroups of delta s_memtime / delta s_memrealtime x 100 MHz) and the median duration of its tower part. */
int tz_debug_net_clock(tz_net* net, double* mhz_out, double* tower_us_out);
/* diagnostic builds: the raw in-kernel stamps of the last stamped launch, [workgroup][4] (TZ_NET_ABL=8: memtime, memrealtime before
   and after the tower; TZ_NET_ABL=16: memtime after the barrier, the k-loop, the barrier and the epilogue of the middle layer) */
int tz_debug_net_stamps(tz_net* net, unsigned long long* out, int max_groups, int* groups_out);

#ifdef __cplusplus
}
#endif
#endif /* TAKZERO_HIP_H */
