/*
 * overcooked_ops.h -- the ops on either side of the env step in the training loop, same library and
 * conventions as overcooked_hip.h (status codes, d_* = device pointers, stream-ordered, no exceptions across the boundary):
 *   - fused policy inference on the matrix cores: every actor and critic of a rollout step in one launch (f2);
 *   - rollouts with the policies in the loop, enqueued from C: one engine, a group of engines, mixed play (f1, f2, f4);
 *   - generalised advantage estimation, advantage normalisation, minibatch gathers (f3).
 * The reference has no native code for any of this: it is Python over PyTorch (train/XD/xd_player.py, train/MAPPO/utils/
 * shared_buffer.py, train/partner_agents.py); every entry point cites the lines it stands in for.
 */
#ifndef OVERCOOKED_OPS_H
#define OVERCOOKED_OPS_H

#include "overcooked_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

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
 * other sizes return OC_ERR_UNSUPPORTED.  Numerics: conv in exact integer arithmetic on `digits` base-256
 * fixed-point weight digits (3, the default: quantisation 2^-24 of each channel's largest weight; the channel's bias
 * rides in the same units), dense layers as f32 x f32 products on the f16 matrix pipe: both operands as two f16 pieces
 * hi = f16(x), lo = f16(x - hi) (round to nearest; weights up-scaled per layer by a power of two, activations fed at
 * 2^4 and clamped at 65504 by the ReLU), three piece products hi hi + hi lo + lo hi, f32 accumulate: what is dropped is
 * below 2^-22 of a product; outputs agree with the reference's f32 forward to ~1e-6 of their magnitude
 * (tools/policy_accuracy.py measures it against float64: no worse than a plain f32 evaluation). */
typedef struct {
    int32_t width, height;     /* observation grid; rows are width*height*20 int8, C fastest, W before H */
    int32_t hidden;            /* reference --hidden_size: 64 */
    int32_t n_out;             /* head outputs of the network being packed: 1..8 (6 logits / 1 value) */
    int32_t digits;            /* conv weight digits 1..4 (8, 16, 24, 32 bits; 3 is the default of the Python host) */
} oc_policy_shape;

#define OC_POLICY_TILE_ACT 1   /* tile flag: actor epilogue (log-softmax, action draw); otherwise raw head outputs */
#define OC_POLICY_TILE_GATHER 16   /* tile flag: the tile is {list, class 0/1, net, flags} and evaluates the rows listed for that class (oc_policy_forward_gather) */
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
 * argmax (FixedCategorical.mode), 1 = a draw from the distribution (FixedCategorical.sample) by inverse CDF: the first action whose
 * cumulative probability exceeds u = oc_policy_uniform(seed, counter + *d_counter, row, 0) (rounds 1-3: Gumbel-max over six uniforms);
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

/* ---- per-row policy mixing without evaluating anything twice ----------------------------------------
 * The reference's mixed play gives every environment, at every step and on each side, to ONE of two policies and evaluates each
 * policy on exactly the rows it was given (`mix_mask == i` indexing: train/partner_agents.py:171-196, train/XD/xd_player.py:287-305).
 * oc_partition_rows: d_select int32 [n_lists][len] of 0 / non-0 choices -> d_index int32 [n_lists][len]: per list the ids of the
 * entries with choice 0 in ascending order, then the ids of the others; d_count int32 [n_lists][2]: how many of each.  The id of
 * entry k of list l is (l % id_period) * len + k: with one list per (step, side) and id_period = 2, ids are rows player * N + kitchen
 * of that step's observation.
 * oc_policy_forward_gather: oc_policy_forward whose tiles may carry OC_POLICY_TILE_GATHER: such a tile is {list, class, net, flags}
 * and evaluates rows d_row_index[list * gather_len + (class ? d_row_count[2 list] : 0) + i], i < d_row_count[2 list + class] -- the
 * kernel stages those rows through indirect LDS-DMA addresses and writes its outputs at the rows' own indices, so two policies
 * share one output buffer and nobody evaluates a row it does not own.  The table is split evenly over the CUs by the kernel
 * (the counts live on the device).  Same numerics, same draws per (seed, counter, row) as oc_policy_forward.
 * When W*H*20 is not a multiple of 16, gathered rows are read in 16-byte pieces that may run up to 12 bytes into the FOLLOWING
 * row; the tensor's last row is handled, nothing is read past n_rows rows. */
int oc_partition_rows(const int32_t *d_select, int64_t n_lists, int64_t len, int64_t id_period, int32_t *d_index, int32_t *d_count,
                      void *stream);
int oc_policy_forward_gather(const oc_policy_shape *shape, const void *d_packed, int32_t n_nets, const int8_t *d_obs, int64_t n_rows,
                             const int32_t *d_tiles, int32_t n_tiles, const int32_t *d_row_index, const int32_t *d_row_count, int64_t gather_len,
                             float *d_raw, int32_t *d_actions, float *d_action_logp, float *d_logp, int sample_mode, uint32_t seed,
                             uint32_t counter, const uint32_t *d_counter, void *stream);
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
 * different networks run on a CU side by side); pass both numbers on to oc_policy_forward_multi.  On ENTRY *waves_per_workgroup is
 * 0 (let the planner choose), 1 or 4 (the caller's choice; results are the same up to the f32 rounding of cooperative blocks). */
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
/* T rollout steps of every engine of a group with the policies in the loop, enqueued from C on `stream` -- the reference's
 * five-layout training / evaluation loop (BASELINE configs 3 and 5): per joint step the engines' policy launches one after the
 * other, then ONE env launch for all of them (oc_group_step).  oc_group_policy_rollout_multi below is the faster form (one
 * policy launch over all grids).
 * slots[i] describes engine i exactly as oc_policy_rollout's arguments do (same buffers, same in_place meaning; the draw counter
 * of step t is counter0 + t, the seed is offset per engine). */
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
    void *d_records;           /* may be NULL; as oc_policy_rollout's */
} oc_group_policy_slot;
int oc_group_policy_rollout(oc_group *g, const oc_group_policy_slot *slots, int64_t T, int in_place, int sample_mode, uint32_t seed,
                            uint32_t counter0, void *stream);
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
 * mixed-play collection of XDPlayer.collect_mp_episode (train/XD/xd_player.py:232-356) is then one call.
 * d_records: NULL, or the compact trajectory [T+1][N][oc_state_record_bytes()] (16-byte aligned, always time-major): the env launch
 * of step t writes the post-step records to slot t+1 (oc_step's d_records); slot 0 is the caller's (oc_export_states before the call). */
int oc_policy_rollout(oc_engine *e, const oc_policy_shape *shape, const void *d_packed, int32_t n_nets, const int32_t *d_tiles,
                      int32_t n_tiles, int64_t T, int in_place, int8_t *d_obs, int32_t *d_actions, float *d_action_logp,
                      float *d_values, int32_t *d_reward, int32_t *d_done, void *d_records, int sample_mode, uint32_t seed, uint32_t counter0,
                      const int32_t *d_row_select, void *stream);
/* The mixed-play collection of XDPlayer.collect_mp_episode (train/XD/xd_player.py:232-356) as ONE call: T steps, always time-major
 * (buffers as oc_policy_rollout with in_place = 0), d_choice int32 [T][2N] = which of the two candidate policies plays row
 * player * N + kitchen at step t.  The call partitions every step's rows by choice (oc_partition_rows into the caller's workspaces
 * d_index_ws int32 [T][2N], d_count_ws int32 [T][2][2]) and runs, per step, one oc_policy_forward_gather + one oc_step.
 * d_tiles: gather tiles {side, choice, net, flags | OC_POLICY_TILE_GATHER}: typically 8 (2 sides x 2 choices x actor / critic).
 * d_records: as oc_policy_rollout's. */
int oc_policy_rollout_mixed(oc_engine *e, const oc_policy_shape *shape, const void *d_packed, int32_t n_nets, const int32_t *d_tiles,
                            int32_t n_tiles, int64_t T, int8_t *d_obs, int32_t *d_actions, float *d_action_logp, float *d_values,
                            int32_t *d_reward, int32_t *d_done, void *d_records, int sample_mode, uint32_t seed, uint32_t counter0,
                            const int32_t *d_choice, int32_t *d_index_ws, int32_t *d_count_ws, void *stream);



#ifdef __cplusplus
}
#endif
#endif /* OVERCOOKED_OPS_H */
