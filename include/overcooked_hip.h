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
 * This header is the environment itself: engines, groups of engines, state.  What sits on either side of the env step in
 * the training loop (policy inference on the matrix cores, GAE, minibatch gathers, rollouts with the policies in the loop)
 * is declared in overcooked_ops.h; measured-and-rejected experiments in overcooked_experiments.h; diagnostics in
 * overcooked_debug.h.  All four are served by the one library.
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
 *     multiplayer_schelling; OvercookedMadrona(..., num_players)) run on a kernel of their own (csrc/oc_np_kernels.h: ONE launch per
 *     step as well, records included) with the same buffers and semantics, C = 5P+10 channels, behind every entry point of this
 *     header -- oc_encode_trajectory (one launch for a whole trajectory) and oc_rollout_fused with a tape (one launch for T steps,
 *     records resident on chip) included; only oc_group_step runs a group that contains such engines as one launch per engine,
 *     same results.
 *     Only the fused policy kernel (overcooked_ops.h) reads 2-player observations only.  S = width*height <= 255 tiles.
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
#define OC_ABI_VERSION 7

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
 * reward, done, and the 20-channel lossless_state_encoding of the post-step state for both viewers.
 * d_records (may be NULL): trajectory recording without an extra pass -- the same launch also writes the POST-step state records
 * of all kitchens there ([N][oc_state_record_bytes()], 16-byte aligned; what oc_export_states would copy out after the step): the
 * compact trajectory of the exchange (oc_encode_trajectory re-featurises it on the receiver). */
int oc_step(oc_engine *e, const void *d_actions, int action_dtype, int8_t *d_obs, int32_t *d_reward,
            int32_t *d_done, void *d_records, void *stream);

/* oc_step with the two players' actions in two device buffers, d_actions[0] = player 0's [N], d_actions[1] = player 1's [N] (a
 * host array of two device pointers, same dtype): what VectorMultiAgentEnv.step holds before it assembles Manager's [P, N, 1] action
 * tensor (pantheonrl_extension/vectorenv.py:171-193, envs/overcooked2_env.py:116-119) -- the kernel reads them where they are, no
 * assembly pass.  Same results as oc_step on the stacked tensor.  Serves in-place destinations (d_obs = the previous step's) of
 * 2-player engines on the specialised kernel; anything else returns OC_ERR_UNSUPPORTED (stack and call oc_step).  The entry points
 * are a module of their own, built on first use or by oc_prepare_kernels. */
int oc_step_players(oc_engine *e, const void *const *d_actions, int action_dtype, int8_t *d_obs, int32_t *d_reward, int32_t *d_done,
                    void *stream);

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
 * bench.py and for trajectory collection straight into time-major buffers.  d_records (may be NULL): step t's post-step
 * records go to d_records + t * records_step_stride BYTES (a multiple of 16), see oc_step. */
int oc_rollout(oc_engine *e, const void *d_actions, int action_dtype, int64_t T_act, int64_t T,
               int8_t *d_obs, int64_t obs_step_stride, int32_t *d_reward, int64_t reward_step_stride,
               int32_t *d_done, int64_t done_step_stride, void *d_records, int64_t records_step_stride, void *stream);

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
 * d_records / records_step_stride (bytes): as oc_rollout -- the records leave the LDS-resident state once per step.
 * Results are identical to T calls of oc_step with the same actions. */
int oc_rollout_fused(oc_engine *e, const void *d_actions, int action_dtype, int64_t T_act, int64_t T, uint32_t rng_seed,
                     uint32_t rng_t0, int8_t *d_obs, int64_t obs_step_stride, int32_t *d_reward, int64_t reward_step_stride,
                     int32_t *d_done, int64_t done_step_stride, void *d_records, int64_t records_step_stride, void *stream);
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
                  int32_t *const *d_done, void *const *d_records /* NULL, or a table of per-engine record destinations (entries may be NULL) */,
                  void *stream);
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

/* Introspection */
int64_t oc_num_kitchens(const oc_engine *e);
int oc_obs_shape(const oc_engine *e, int64_t shape5[5]);   /* P, N, W, H, C */
int oc_device_id(const oc_engine *e);

/* Launch tuning (performance only, results never change): kitchens per wavefront group (1..64;
 * kitchens*tiles must be a multiple of 4, so any multiple of 4 works), 64-tile staging passes per flush (1..8), persistent wavefronts per CU (1..32).
 * 0 keeps the current value.  Engines with 1, 3 or 4 players: kitchens per group (x players <= 64, x 16-byte chunks per record
 * <= 256), passes per flush (1..8), wavefronts per CU (1..32); until it is called their launches choose a geometry per batch size. */
int oc_set_tuning(oc_engine *e, int kitchens_per_wave, int passes_per_flush, int waves_per_cu);

/* Build NOW what is otherwise compiled on first use -- the step kernels for fresh (time-major) destinations and the
 * K-steps-per-launch rollout kernels (hiprtc + module load, about a second each the first time, from the code-object cache
 * afterwards).  Call it before capturing oc_step / oc_rollout* into a HIP graph: a first use inside a capture would compile and
 * load a module there.  A build that fails is remembered and the generic kernel serves (results are identical).
 * Memory: an engine whose viewer planes are no multiple of 64 bytes (n_kitchens x tiles x 20 % 64 != 0) gets a SECOND record block
 * (n_kitchens x oc_state_record_bytes()) with that build -- its steps into fresh destinations read one block and write the other and
 * the two swap roles (the neighbour's last kitchen is stepped as a halo; if the allocation is refused the launches run without halo
 * kitchens: same results, and oc_kernel_build_info says so).
 * Graph capture: the record block's address is a kernel argument, so from the FIRST capture of any entry point of an engine on,
 * that engine's blocks never swap again -- graph replays and eager calls may then be mixed freely (tests/test_hip_parity.py:
 * capture, eager steps into fresh slots, replay); the price is that its later steps into fresh destinations run without halo kitchens
 * (~14.1 -> 18.3 us per step at 65 535 cramped_room kitchens).  Capture on an engine before its first eager fresh-destination step,
 * or not at all, if that matters. */
int oc_prepare_kernels(oc_engine *e);

/* Which kernel build serves this engine: "specialised for gfx950..." when the per-engine hiprtc build
 * succeeded at oc_create, otherwise the reason the generic ahead-of-time kernel is in use. */
const char *oc_kernel_build_info(const oc_engine *e);

const char *oc_last_error(void);
int oc_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* OVERCOOKED_HIP_H */
