/*
 * overcooked_hip.h -- C ABI of the MI355X-native Overcooked rollout engine.
 *
 * This is the drop-in boundary for the reference's native simulator binding
 * (Simplecooked "SimplecookedSimulator", /root/reference/src/overcooked2_env/bindings.cpp:27-96,
 * Manager in mgr.hpp:14-67 / mgr.cpp:128-272).  The reference exports a constructor taking the
 * flat layout parameters, a blocking step(), and ten zero-copy tensors; this ABI exports the same
 * capability as plain C: a handle that owns the per-kitchen state in HBM, a stream-ordered step()
 * that reads an action buffer and writes observation / reward / done buffers (device pointers
 * supplied by the caller -- torch tensors on the Python side), plus state import/export for tests,
 * checkpoints and trajectory exchange.
 *
 * Conventions
 *   - every entry point returns OC_OK (0) or a negative oc_status; oc_last_error() gives the text
 *     (thread-local).  No exceptions cross the boundary, nothing aborts the process.
 *   - all pointers named d_* are DEVICE pointers on the engine's GPU; `stream` is a hipStream_t
 *     passed as void* (NULL = the default stream).  Calls only enqueue work: no implicit
 *     device synchronisation except in oc_create / oc_destroy / oc_get_state / oc_set_state.
 *   - one handle = one GPU = one caller thread at a time (same contract as the reference,
 *     SURVEY.md 8(b) "threading").  Several handles per process are fine.
 *   - P = 2 players is the tuned path (the reference's training scripts and its dish-usefulness rule are 2-player): C = 5P+10
 *     = 20 channels.  P = 1, 3 or 4 (the reference's Python MDP and encoder are N-player: layouts simple_single,
 *     multiplayer_schelling; OvercookedMadrona(..., num_players)) run on a plain, separate kernel pair with the same buffers and
 *     semantics, C = 5P+10 channels; oc_rollout_fused, oc_encode_trajectory, groups and the fused
 *     policy kernel are 2-player only and return OC_ERR_UNSUPPORTED for them.  S = width*height <= 255 tiles.
 *
 * Tensor layouts (identical in meaning to the reference's exported tensors)
 *   actions  int32 or int64 [P][N]        ref: action_tensor   i32[P,N,1]   (mgr.cpp:225-229)
 *   obs      int8  [P][N][W][H][C]        ref: observation_tensor after the (P,H,W,N,C)->transpose(1,3)
 *                                              view of envs/overcooked2_env.py:103-114
 *   reward   int32 [P][N]  (sum over players, replicated)  ref: reward_tensor (sim.cpp:623)
 *   done     int32 [N]     1 on the step whose timestep reached the horizon; the kitchen is
 *                          reset inside that step and obs is the reset observation
 *                          ref: done_tensor / WorldReset (sim.cpp:592-640)
 *
 * State record (oc_get_state / oc_set_state / oc_export_states / oc_encode_states), little endian,
 * oc_state_record_bytes() bytes per kitchen = 16 * ceil((S + 4) / 4):
 *   uint32 tile[S]     name | onions<<8 | tomatoes<<16 | (uint8)cooking_tick<<24   (0 = empty)
 *   uint32 posor       p0.pos | p0.orient<<8 | p1.pos<<16 | p1.orient<<24
 *   uint32 held[2]     same packing as a tile object
 *   uint32 timestep
 *   zero padding to a multiple of 16 bytes
 * With P != 2 players: tile[S], uint32 player[P] = position | orientation<<8, uint32 held[P], uint32 timestep, padding
 * (16 * ceil((S + 2P + 1) / 4) bytes).
 */
#ifndef OVERCOOKED_HIP_H
#define OVERCOOKED_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OC_MAX_TILES 255
#define OC_MAX_PLAYERS 4
#define OC_NUM_ACTIONS 6
#define OC_ABI_VERSION 4

typedef enum {
    OC_OK = 0,
    OC_ERR_INVALID_ARG = -1,   /* NULL pointer, bad size, misaligned buffer, terrain code out of range ... */
    OC_ERR_UNSUPPORTED = -2,   /* more than 4 players, too many tiles/pots, cook time > 127, a 2-player-only call on another count */
    OC_ERR_HIP = -3,           /* a HIP runtime call failed; text in oc_last_error() */
    OC_ERR_NO_DEVICE = -4      /* no usable GPU: the engine never falls back to the CPU */
} oc_status;

typedef enum { OC_ACT_I32 = 0, OC_ACT_I64 = 1 } oc_action_dtype;

/* Flat layout parameters: field for field the keyword arguments of the reference constructor
 * (bindings.cpp:28-84; produced in Python by get_base_layout_params, envs/overcooked2_env.py:171-291). */
typedef struct {
    int32_t width, height, num_players;
    int32_t placement_in_pot_rew, dish_pickup_rew, soup_pickup_rew;
    int32_t horizon;
    int32_t recipe_values[16];          /* index 4*onions + tomatoes */
    int32_t recipe_times[16];
    int32_t start_player_x[OC_MAX_PLAYERS];
    int32_t start_player_y[OC_MAX_PLAYERS];
    int32_t terrain[256];               /* row-major, terrain[y*width+x], codes 0..6 */
} oc_layout_params;

typedef struct oc_engine oc_engine;
typedef struct oc_group oc_group;      /* several engines stepped by one launch, see oc_group_create */

/* replaces Manager::Manager(Config) (mgr.cpp:179-206) / SimplecookedSimulator.__init__.
 * Allocates the state block for n_kitchens kitchens on device_id and puts every kitchen in the
 * standard start state (ref envs/overcooked2_reimplement.py:387-391). */
int oc_create(const oc_layout_params *params, int64_t n_kitchens, int device_id, oc_engine **out);

/* replaces Manager::~Manager (mgr.cpp:206) */
int oc_destroy(oc_engine *e);

/* replaces Manager::step (mgr.cpp:208-211), but stream-ordered instead of blocking.
 * One fused launch: joint interact/move/collision transition, pot timers, horizon auto-reset,
 * reward, done, and the 20-channel lossless_state_encoding of the post-step state for both viewers. */
int oc_step(oc_engine *e, const void *d_actions, int action_dtype, int8_t *d_obs, int32_t *d_reward,
            int32_t *d_done, void *stream);

/* Observation of the current state without stepping: what OvercookedMadrona.n_reset()/get_obs()
 * return (envs/overcooked2_env.py:103-114,127-128 -- n_reset re-reads, it does not reset). */
int oc_observe(oc_engine *e, int8_t *d_obs, void *stream);

/* Put every kitchen back in the start state (what constructing a fresh simulator does,
 * sim.cpp:690-704 resetWorld) and, if d_obs != NULL, write the start observation. */
int oc_reset_all(oc_engine *e, int8_t *d_obs, void *stream);

/* Per-kitchen episode control on the device (staggered / mixed-play style schedules; the
 * reference can only reset by reaching the horizon).  d_reset_mask int32[N] or NULL: non-zero entries put that
 * kitchen back in the start state.  d_timesteps int32[N] or NULL: timestep[k] := d_timesteps[k] (clamped to
 * 0..horizon-1), so kitchens reach their horizon -- and auto-reset -- at different steps.  Applied mask first. */
int oc_episode_control(oc_engine *e, const int32_t *d_reset_mask, const int32_t *d_timesteps, void *stream);

/* T consecutive steps enqueued from C (no host work between launches): actions [T_act][P][N] are
 * consumed cyclically (step t uses slice t % T_act); obs/reward/done are overwritten every step
 * unless the *_step_stride (in elements, may be 0) says to advance per step.  Rollout driver for
 * bench.py and for trajectory collection straight into time-major buffers. */
int oc_rollout(oc_engine *e, const void *d_actions, int action_dtype, int64_t T_act, int64_t T,
               int8_t *d_obs, int64_t obs_step_stride, int32_t *d_reward, int64_t reward_step_stride,
               int32_t *d_done, int64_t done_step_stride, void *stream);

/* Same as oc_rollout but brackets the T launches with hipEvents on `stream`, synchronises, and
 * returns the elapsed milliseconds (kernel-time basis for the roofline figure). */
int oc_rollout_timed(oc_engine *e, const void *d_actions, int action_dtype, int64_t T_act, int64_t T,
                     int8_t *d_obs, int32_t *d_reward, int32_t *d_done, void *stream, float *elapsed_ms);

/* Whole rollout in ONE launch, for action streams that are known in advance (synthetic benchmarks, scripted or
 * replayed tapes; a policy in the loop needs oc_step).  Kitchens never interact, so each wavefront keeps its
 * kitchens' state in LDS for all T steps: no per-step launch boundary, no per-step state round trip through
 * HBM, and step t's plane stores drain while step t+1 is computed.  Outputs for step t go to
 * base + t * *_step_stride (obs stride in bytes, reward/done in elements; 0 = overwrite in place).
 * d_actions: tape [T_act][P][N] used cyclically, or NULL for the built-in counter-hash stream
 * action = oc_synthetic_action(rng_seed, kitchen, rng_t0 + t, player) (uniform on 0..5).
 * Results are identical to T calls of oc_step with the same actions. */
int oc_rollout_fused(oc_engine *e, const void *d_actions, int action_dtype, int64_t T_act, int64_t T, uint32_t rng_seed,
                     uint32_t rng_t0, int8_t *d_obs, int64_t obs_step_stride, int32_t *d_reward, int64_t reward_step_stride,
                     int32_t *d_done, int64_t done_step_stride, void *stream);
uint32_t oc_synthetic_action(uint32_t seed, uint32_t kitchen, uint32_t step, uint32_t player);

/* Featurise n externally supplied state records (device pointer, record format above) without
 * touching the engine's own kitchens: the receive side of a compact trajectory exchange. */
int oc_encode_states(oc_engine *e, const void *d_states, int64_t n, int8_t *d_obs, void *stream);
/* Same, writing viewer v's [n][W][H][C] block at d_obs + v * obs_player_stride bytes: lets a receiver
 * drop a peer's shard at its kitchen offset inside a larger [P][N_total][W][H][C] tensor. */
int oc_encode_states_strided(oc_engine *e, const void *d_states, int64_t n, int8_t *d_obs, int64_t obs_player_stride,
                             void *stream);

/* A whole gathered trajectory of one peer re-featurised in ONE launch (the receive side of the compact exchange,
 * SURVEY.md 8(e)): T blocks of n records, block t at d_states + t * state_step_stride bytes; viewer v's [n][W][H][C]
 * plane of block t goes to d_obs + t * obs_step_stride + v * obs_player_stride bytes (strides in bytes). */
int oc_encode_trajectory(oc_engine *e, const void *d_states, int64_t n, int64_t T, int64_t state_step_stride, int8_t *d_obs,
                         int64_t obs_player_stride, int64_t obs_step_stride, void *stream);

/* ---- several engines stepped by ONE launch ---------------------------------------------------------------------------
 * The reference evaluates and trains on its five layouts side by side, one simulator object per layout
 * (train/env_utils.py:26 generate_env, train/test_all.sh:8; BASELINE configs 3 and 5), i.e. five Manager::step calls per
 * joint step.  A group is the MI355X counterpart of that set: up to 8 engines on one GPU -- different layouts, sizes,
 * horizons; each still usable on its own -- whose step (or observe) is a single kernel launch.  The group module holds one
 * specialised kernel body per distinct layout and hands every engine a share of the persistent wavefronts in proportion to
 * its tiles, so the constant folding of the per-engine build is kept and the four extra launch boundaries are not paid.
 * The pointer tables are HOST arrays of oc_group_size() device pointers, in the order of oc_group_create; buffers, dtypes
 * and results per engine are exactly those of oc_step / oc_observe.  Without a usable hiprtc the calls fall back to one
 * launch per engine, same results (oc_group_build_info says which). */
int oc_group_create(oc_engine *const *engines, int32_t n_engines, oc_group **out);
int oc_group_destroy(oc_group *g);                    /* the engines belong to the caller: they stay alive, and must outlive the group */
int32_t oc_group_size(const oc_group *g);
int oc_group_step(oc_group *g, const void *const *d_actions, int action_dtype, int8_t *const *d_obs, int32_t *const *d_reward,
                  int32_t *const *d_done, void *stream);
int oc_group_observe(oc_group *g, int8_t *const *d_obs, void *stream);
/* T joint steps enqueued from C and bracketed by hipEvents on `stream` (the group counterpart of oc_rollout_timed):
 * engine i consumes its action tape d_actions[i] = [T_act][P][N_i] cyclically; outputs overwritten in place. */
int oc_group_rollout_timed(oc_group *g, const void *const *d_actions, int action_dtype, int64_t T_act, int64_t T, int8_t *const *d_obs,
                           int32_t *const *d_reward, int32_t *const *d_done, void *stream, float *elapsed_ms);
/* Launch tuning of the group (performance only): kitchens per wavefront group for every engine (host array of
 * oc_group_size() ints; NULL or 0 entries = chosen from the joint batch), passes per flush, persistent wavefronts per CU. */
int oc_group_set_tuning(oc_group *g, const int32_t *kitchens_per_wave, int passes_per_flush, int waves_per_cu);
const char *oc_group_build_info(const oc_group *g);

/* State import / export.  get/set are synchronous host copies (tests, checkpoints);
 * export/import are stream-ordered device-to-device copies of `count` records starting at `first`. */
int64_t oc_state_record_bytes(const oc_engine *e);
int oc_get_state(oc_engine *e, int64_t first, int64_t count, void *host_dst);
int oc_set_state(oc_engine *e, int64_t first, int64_t count, const void *host_src);
int oc_export_states(oc_engine *e, int64_t first, int64_t count, void *d_dst, void *stream);
int oc_import_states(oc_engine *e, int64_t first, int64_t count, const void *d_src, void *stream);
/* The same with a device-side check of every record (players on interior tiles with orientations 0..3, object names
 * 0..4 on tiles and in hands -- everything the kernels index with): records that fail are NOT copied and are counted in
 * *d_rejected (device int32, zeroed by the call).  Checkpoint reload of a whole engine without a host pass. */
int oc_import_states_checked(oc_engine *e, int64_t first, int64_t count, const void *d_src, int32_t *d_rejected, void *stream);
/* The check alone: *d_rejected = number of the `count` records at d_src that oc_import_states_checked would refuse; the engine's
 * kitchens are not touched.  An all-or-nothing reload is this call, a read of the count, and oc_import_states when it is zero.
 * Both calls take any player count (records in the engine's own format). */
int oc_validate_states(oc_engine *e, int64_t count, const void *d_src, int32_t *d_rejected, void *stream);

/* ---- runner-side ops ---------------------------------------------------------------
 * Generalised advantage estimation over a whole trajectory batch in one launch; replaces the GAE branches
 * of SharedReplayBuffer.compute_returns (train/MAPPO/utils/shared_buffer.py:248-304) and the advantage
 * subtraction of train/MAPPO/r_mappo.py:174-177.  All tensors f32, time-major, M = envs * agents:
 *   rewards [T][M], value_preds [T+1][M] (row T = next_value), masks [T+1][M], bad_masks [T+1][M] (only read
 *   with proper_time_limits), returns [T+1][M] (rows 0..T-1 written), advantages [T][M] or NULL.
 * denormalize != 0: value_preds are ValueNorm/PopArt-normalised, v*vn_std + vn_mean is applied as in
 * valuenorm.py:74-83.  Bit-identical to the reference's torch expressions (same f32 op order, no FMA). */
int oc_gae_returns(const float *d_rewards, const float *d_value_preds, const float *d_masks, const float *d_bad_masks,
                   float *d_returns, float *d_advantages, int64_t T, int64_t M, double gamma, double gae_lambda,
                   int denormalize, float vn_mean, float vn_std, int proper_time_limits, void *stream);

/* Advantage normalisation of train/XD/xd.py:240-244 (= train/MAPPO/r_mappo.py:177-181) over n f32 elements:
 *   mean = nanmean over entries with active_mask != 0 ; std = unbiased std ; out = (adv - mean) / (std + 1e-5).
 * std_over_active = 0 keeps the reference's behaviour (its `copy != nan` selects every entry, so the std --
 * and every output -- is NaN as soon as one entry is inactive); 1 takes the std of the active entries.
 * d_active_masks may be NULL (all active); d_out may alias d_adv or be NULL (statistics only);
 * d_stats receives {mean, std}; d_workspace holds oc_advantage_workspace_bytes() bytes.  Sums are taken in f64
 * in a fixed order: reproducible run to run, within 1e-6 of torch's f32 reductions. */
int64_t oc_advantage_workspace_bytes(void);
int oc_advantage_normalize(const float *d_adv, const float *d_active_masks, int64_t n, int std_over_active, float *d_out,
                           float *d_stats, void *d_workspace, void *stream);

/* Row gather, dst row i <- src row d_index[i] (NULL index = identity): the data movement of
 * SharedReplayBuffer.feed_forward_generator / partial_ff_generator (shared_buffer.py:331-345, 393-407) and of
 * diaginsert / partinsert (:167-179, :200-212) with the index arithmetic done by the caller on the device.
 * Rows are row_bytes bytes of int8 observations (multiple of 4, <= 5120); dst_dtype 0 copies the bytes,
 * 1 writes f32 (the reference's buffer dtype; cast fused).  Indices outside [0, src_rows) are clamped. */
int oc_gather_rows(const void *d_src, int64_t src_rows, int32_t row_bytes, const int64_t *d_index, int64_t n_rows, void *d_dst,
                   int dst_dtype, void *stream);
/* The scalar side of a minibatch: up to 8 columns of words[c] 4-byte words per row (value_preds, returns, masks,
 * active_masks, actions, action_log_probs, advantages, rnn_states ...) gathered in one launch; source row =
 * d_index[i] * row_mul + row_add -- (1, 0) for feed_forward_generator, (num_agents, agent) for
 * partial_ff_generator.  d_src / d_dst / src_rows / words / d_norm_stats are HOST arrays of n_cols entries;
 * a non-NULL d_norm_stats[c] (device {mean, std} from oc_advantage_normalize) normalises column c on the fly. */
int oc_gather_columns(int32_t n_cols, const void *const *d_src, const int64_t *src_rows, const int32_t *words, void *const *d_dst,
                      const float *const *d_norm_stats, const int64_t *d_index, int64_t n_rows, int64_t row_mul, int64_t row_add,
                      void *stream);
/* Sample records: oc_pack_columns interleaves n_cols (<= 8) one-word columns of n_rows rows into 32-byte records
 * d_packed[n_rows][8] (once per update, streaming); oc_gather_records reads one record per sample -- one 32-byte
 * sector instead of n_cols -- and writes the columns out separately and contiguously (d_dst: HOST array of n_cols
 * device pointers, each [n_rows] words).  Row mapping as oc_gather_columns; norm_col >= 0 normalises that column
 * with d_norm_stats = {mean, std} on the fly, -1 none. */
int oc_pack_columns(int32_t n_cols, const void *const *d_src, int64_t n_rows, void *d_packed, void *stream);
int oc_gather_records(const void *d_packed, int64_t src_rows, int32_t n_cols, void *const *d_dst, int32_t norm_col,
                      const float *d_norm_stats, const int64_t *d_index, int64_t n_rows, int64_t row_mul, int64_t row_add,
                      void *stream);

/* ---- fused policy inference on the matrix cores --------------------------------------------
 * One launch evaluates every (row range, network) tile of a rollout step -- all conventions' actors and critics,
 * both sides, every pairing segment -- reading the int8 observations oc_step wrote and writing the int32 actions
 * oc_step reads.  Replaces the per-segment loops over policy.get_actions (train/XD/xd_player.py:190-207,
 * train/partner_agents.py:97-111; rMAPPOPolicy.get_actions = R_Actor.forward + R_Critic.forward,
 * train/MAPPO/r_actor_critic.py:57-76,175-199) for the reference architecture: CNNBase (utils/cnn.py:12-60: conv 3x3
 * valid 20 -> hidden/2, ReLU, flatten, Linear -> hidden, ReLU, Linear -> hidden, ReLU) + one Linear head
 * (6 action logits, utils/act.py + distributions.py:60-80, or 1 value, `v_out`).  hidden_size 64 (train_xp.sh:12);
 * other sizes return OC_ERR_UNSUPPORTED.  Numerics: conv in exact integer arithmetic on `digits` base-128
 * fixed-point weight digits (4: quantisation 2^-28 of each channel's largest weight; the channel's bias rides in the
 * same units), dense layers as f32 x f32 products on the bf16 matrix pipe (both operands split exactly into three bf16
 * pieces, six piece products each, f32 accumulate: what is dropped is below 2^-23 of a product); outputs agree with
 * the reference's f32 forward to ~1e-6 of their magnitude (tools/policy_accuracy.py measures it against float64). */
typedef struct {
    int32_t width, height;     /* observation grid; rows are width*height*20 int8, C fastest, W before H */
    int32_t hidden;            /* reference --hidden_size: 64 */
    int32_t n_out;             /* head outputs of the network being packed: 1..8 (6 logits / 1 value) */
    int32_t digits;            /* conv weight digits 1..4 (7, 14, 21, 28 bits) */
} oc_policy_shape;

#define OC_POLICY_TILE_ACT 1   /* tile flag: actor epilogue (log-softmax, action draw); otherwise raw head outputs */
#define OC_POLICY_TILE_SELECT(v) ((((v) + 1) & 0xff) << 8)   /* tile flag: write only rows whose d_row_select entry == v (0..254) */

/* Bytes of one packed network (a function of width, height, digits only; multiple of 256). < 0 on error. */
int64_t oc_policy_packed_bytes(const oc_policy_shape *shape);
/* Pack one network from the reference's state-dict tensors (f32, contiguous): conv_w [32][20][3][3]
 * (base.cnn.cnn.0.weight), conv_b [32], fc1_w [64][32*(W-2)*(H-2)] (base.cnn.cnn.3), fc1_b, fc2_w [64][64]
 * (base.cnn.cnn.5), fc2_b, head_w [n_out][64] (act.action_out.linear / v_out), head_b.  _host: host pointers,
 * synchronous (tests, C clients); the other: device pointers, stream-ordered (after an optimiser step). */
int oc_policy_pack_host(const oc_policy_shape *shape, const float *conv_w, const float *conv_b, const float *fc1_w,
                        const float *fc1_b, const float *fc2_w, const float *fc2_b, const float *head_w, const float *head_b,
                        void *host_dst);
int oc_policy_pack(const oc_policy_shape *shape, const float *d_conv_w, const float *d_conv_b, const float *d_fc1_w,
                   const float *d_fc1_b, const float *d_fc2_w, const float *d_fc2_b, const float *d_head_w,
                   const float *d_head_b, void *d_dst, void *stream);
/* Cut segments (row0, rows, network index, flags) into workgroup tiles: host_tiles int32 [capacity][4] = {row0, rows,
 * net, flags}.  rows_per_tile > 0 (a multiple of 32): fixed-size tiles.  rows_per_tile < 0: -rows_per_tile workgroups
 * (pass the CU count) are shared out over the segments in proportion to their rows and every segment is cut into
 * near-equal pieces -- one balanced tile per CU, no workgroup ever changes network.  Returns the tile count
 * (host_tiles may be NULL to ask for it), < 0 on error.  oc_policy_forward runs a table of >= CUs/2 tiles one
 * workgroup per tile, a shorter one by splitting its 32-row blocks evenly over one workgroup per CU.
 * Pairing segments: train/XD/serial.py:27, xd_player.py:76-77,94. */
int64_t oc_policy_plan_tiles(const int64_t *seg_row0, const int64_t *seg_rows, const int32_t *seg_net, const int32_t *seg_flags,
                             int32_t n_segs, int32_t rows_per_tile, int32_t *host_tiles, int64_t capacity);
/* d_packed: n_nets packed networks of this shape (n_out may differ per network; it is read from each image),
 * d_obs int8 [n_rows][W][H][20], d_tiles int32 [n_tiles][4] on the device.  Tiles without OC_POLICY_TILE_ACT write
 * d_raw[row * n_out + a] (raw head outputs: critic values, or logits); tiles with it write d_actions[row] (int32),
 * d_action_logp[row] and, if non-NULL, d_logp[row * n_out + a] (normalised log-probabilities).  sample_mode 0 =
 * argmax (FixedCategorical.mode), 1 = Gumbel-max draw with u = oc_policy_uniform(seed, counter + *d_counter, row, a);
 * d_counter (device uint32, may be NULL = 0) lets a captured HIP graph draw fresh actions on every replay.
 * d_row_select (int32 [n_rows], may be NULL): per-row policy mixing -- tiles flagged OC_POLICY_TILE_SELECT(v) write only
 * the rows whose entry equals v, so two networks laid over the same rows fill one output buffer between them (the
 * reference's `mix_mask == i` indexing, train/XD/xd_player.py:287-305, train/partner_agents.py:171-196).
 * Rows and network indices of a bad tile are clamped; any output pointer may be NULL. */
int oc_policy_forward(const oc_policy_shape *shape, const void *d_packed, int32_t n_nets, const int8_t *d_obs, int64_t n_rows,
                      const int32_t *d_tiles, int32_t n_tiles, float *d_raw, int32_t *d_actions, float *d_action_logp,
                      float *d_logp, int sample_mode, uint32_t seed, uint32_t counter, const uint32_t *d_counter,
                      const int32_t *d_row_select, void *stream);
float oc_policy_uniform(uint32_t seed, uint32_t counter, uint32_t row, uint32_t action);
/* oc_policy_forward sized for a SHARE of the GPU: cu_budget = number of CUs the stream may use (a CU-masked stream from
 * oc_cu_partition_create; 0 = the whole device).  One workgroup per CU of the budget; plan the tile table for it
 * (oc_policy_plan_tiles(..., -cu_budget, ...)). */
int oc_policy_forward_cus(const oc_policy_shape *shape, const void *d_packed, int32_t n_nets, const int8_t *d_obs, int64_t n_rows,
                          const int32_t *d_tiles, int32_t n_tiles, float *d_raw, int32_t *d_actions, float *d_action_logp,
                          float *d_logp, int sample_mode, uint32_t seed, uint32_t counter, const uint32_t *d_counter,
                          const int32_t *d_row_select, int32_t cu_budget, void *stream);
/* One launch over SEVERAL observation tensors of different grids (the layouts of an engine group): every slot is what
 * oc_policy_forward takes for one grid; tiles name their slot with OC_POLICY_TILE_SLOT(i) in the flags.
 * oc_policy_plan_tiles_multi lays the 32-row blocks of all segments out as one sequence and gives every workgroup an equal share
 * of its COST (slot_cost[i]: relative cost of a block of slot i -- (W-2)*(H-2) + 2.5 fits the measured launches), cutting
 * segments at the share boundaries: host_tiles then holds *tiles_per_workgroup entries per workgroup (empty ones as padding),
 * workgroups * *tiles_per_workgroup in all (the return value; host_tiles may be NULL to ask for it).  Five per-layout
 * launches cannot share a CU (each fills its registers and LDS), so at 8192 kitchens per layout they leave half the SIMDs
 * without a block; one balanced table does not.  All slots' networks must use the same number of conv digits; at most 8 slots.
 * *waves_per_workgroup comes back 4, or 1 for tables of short segments (fewer than 12 blocks each on average: "wave tiles" -- the
 * table is then planned for 4 x workgroups one-wavefront workgroups that keep their whole network in L2 / registers, so four
 * different networks run on a CU side by side); pass both numbers on to oc_policy_forward_multi. */
#define OC_POLICY_TILE_SLOT(i) (((i) & 7) << 16)
typedef struct {
    const oc_policy_shape *shape;
    const void *d_packed;
    int32_t n_nets;
    const int8_t *d_obs;
    int64_t n_rows;
    float *d_raw;                  /* any output may be NULL */
    int32_t *d_actions;
    float *d_action_logp;
    float *d_logp;
    const int32_t *d_row_select;
    uint32_t seed;
} oc_policy_multi_slot;
int64_t oc_policy_plan_tiles_multi(const int64_t *seg_row0, const int64_t *seg_rows, const int32_t *seg_net, const int32_t *seg_flags,
                                   const int32_t *seg_slot, const float *slot_cost, int32_t n_slots, int32_t n_segs, int32_t workgroups,
                                   int32_t *host_tiles, int64_t capacity, int32_t *tiles_per_workgroup, int32_t *waves_per_workgroup);
int oc_policy_forward_multi(const oc_policy_multi_slot *slots, int32_t n_slots, const int32_t *d_tiles, int32_t n_tiles, int32_t tiles_per_workgroup,
                            int32_t waves_per_workgroup, int sample_mode, uint32_t counter, const uint32_t *d_counter, void *stream);
/* n HIP streams restricted to disjoint sets of the device's CUs, stream i getting about shares[i] / sum(shares) of them (at
 * least one; every set spread over the whole CU index range).  k mask bits give k CUs and proportional throughput
 * (tools/probes/cu_mask.hip); two isolated policy launches of 8184 asymmetric_advantages kitchens take 49.6 us together on two
 * half-device streams against 91 us one after the other.  For the five-layout rollout a fixed split measured slower than plain
 * streams (see oc_group_policy_rollout), so nothing in the package uses it by default.  out_streams: n hipStream_t (as void*),
 * out_cus (optional): CUs given to each.  Destroy with oc_cu_partition_destroy. */
int oc_cu_partition_create(int device_id, const float *shares, int32_t n, void **out_streams, int32_t *out_cus);
int oc_cu_partition_destroy(void *const *streams, int32_t n);
/* T rollout steps of every engine of a group with the policies in the loop, enqueued from C -- the reference's five-layout
 * training / evaluation loop (BASELINE configs 3 and 5).  Two modes:
 *   policy_streams == NULL: everything on `stream`; per joint step the engines' policy launches one after the other, then
 *     ONE env launch for all of them (oc_group_step);
 *   policy_streams given (ordinary streams, or CU-masked ones with their cu_budgets): kitchens of different engines never
 *     interact, so engine i runs its whole rollout -- per step its policy launch and its own env launch -- on policy_streams[i],
 *     side by side with the others and without any synchronisation between engines; `stream` is forked once at the start and
 *     joined once at the end.  (Joining the streams at every step for a common env launch was measured first: 1.3 ms per joint
 *     step.)  Config 3's shape: 173 us per joint step on five ordinary streams, 186 us on one stream, 423 us on CU-masked ones.
 * slots[i] describes engine i exactly as oc_policy_rollout's arguments do (same buffers, same in_place meaning; the draw counter
 * of step t is counter0 + t, the seed is offset per engine).  Results are identical in both modes. */
typedef struct {
    const oc_policy_shape *shape;
    const void *d_packed;
    int32_t n_nets;
    const int32_t *d_tiles;
    int32_t n_tiles;
    int8_t *d_obs;
    int32_t *d_actions;
    float *d_action_logp;      /* may be NULL */
    float *d_values;           /* may be NULL */
    int32_t *d_reward;
    int32_t *d_done;
} oc_group_policy_slot;
int oc_group_policy_rollout(oc_group *g, const oc_group_policy_slot *slots, void *const *policy_streams, const int32_t *cu_budgets, int64_t T,
                            int in_place, int sample_mode, uint32_t seed, uint32_t counter0, void *stream);
/* The same rollout with TWO launches per joint step on one stream: one oc_policy_forward_multi over every engine's rows
 * (d_tiles from oc_policy_plan_tiles_multi, slot i = engine i; the slots' own d_tiles are ignored) and one oc_group_step.
 * Same seeds and counters as oc_group_policy_rollout: identical results. */
int oc_group_policy_rollout_multi(oc_group *g, const oc_group_policy_slot *slots, const int32_t *d_tiles, int32_t n_tiles, int32_t tiles_per_workgroup,
                                  int32_t waves_per_workgroup, int64_t T, int in_place, int sample_mode, uint32_t seed, uint32_t counter0, void *stream);
/* T rollout steps with the policies in the loop, enqueued from C (no host work between launches): per step one
 * oc_policy_forward over the 2N rows of the engine's observation (tiles address rows player * N + kitchen; critic
 * tiles write d_values) and one oc_step.  What XDPlayer.next_step does per step through 2(2k+1) get_actions calls
 * (train/XD/xd_player.py:177-230).  in_place = 0: time-major trajectory -- d_obs int8 [T+1][P][N][W][H][C] with slot 0
 * holding the current observation (oc_observe), d_actions int32 [T][P][N], d_action_logp / d_values f32 [T][P][N] or
 * NULL, d_reward int32 [T][P][N], d_done int32 [T][N]; step t reads slot t and writes slot t+1.  in_place != 0: one
 * slot of each, overwritten every step.  The draw counter of step t is counter0 + t.  d_row_select: NULL, or int32 [T][2N]
 * per-row policy choices of every step (always time-major), as oc_policy_forward takes them for one step: the whole
 * mixed-play collection of XDPlayer.collect_mp_episode (train/XD/xd_player.py:232-356) is then one call. */
int oc_policy_rollout(oc_engine *e, const oc_policy_shape *shape, const void *d_packed, int32_t n_nets, const int32_t *d_tiles,
                      int32_t n_tiles, int64_t T, int in_place, int8_t *d_obs, int32_t *d_actions, float *d_action_logp,
                      float *d_values, int32_t *d_reward, int32_t *d_done, int sample_mode, uint32_t seed, uint32_t counter0,
                      const int32_t *d_row_select, void *stream);

/* Introspection */
int64_t oc_num_kitchens(const oc_engine *e);
int oc_obs_shape(const oc_engine *e, int64_t shape5[5]);   /* P, N, W, H, C */
int oc_device_id(const oc_engine *e);

/* Launch tuning (performance only, results never change): kitchens per wavefront group (1..64;
 * kitchens*tiles must be a multiple of 4, so any multiple of 4 works), 64-tile staging passes per flush (1..8), persistent wavefronts per CU (1..32).
 * 0 keeps the current value. */
int oc_set_tuning(oc_engine *e, int kitchens_per_wave, int passes_per_flush, int waves_per_cu);

/* Size the engine's launches for a stream that may only use `cus` of the GPU's CUs (a CU-masked stream from
 * oc_cu_partition_create): the persistent grid becomes waves_per_cu x cus wavefronts.  0 = the whole device (default). */
int oc_set_cu_budget(oc_engine *e, int32_t cus);

/* Which kernel build serves this engine: "specialised for gfx950..." when the per-engine hiprtc build
 * succeeded at oc_create, otherwise the reason the generic ahead-of-time kernel is in use. */
const char *oc_kernel_build_info(const oc_engine *e);

const char *oc_last_error(void);
int oc_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* OVERCOOKED_HIP_H */
