/*
 * overcooked_experiments.h -- entry points of experiments that were built, measured and NOT adopted.  They exist in the
 * DIAGNOSTICS build of the library only (-DOC_DIAGNOSTICS -> csrc/liboc_hip_diag.so, what tools/ load: tools/exp_cu_mask.py,
 * exp_cu_partition_parts.py, exp_group_streams.py); the shipped liboc_hip.so exports none of them (round 6), nothing in the package
 * calls them, and they are not part of the drop-in boundary: signatures may change without an ABI bump.
 *
 * CU-masked streams: k mask bits give k CUs and proportional throughput (tools/probes/cu_mask.hip), and two isolated policy
 * launches of 8184 asymmetric_advantages kitchens take 49.6 us together on two half-device streams against 91 us one after the
 * other -- but for the five-layout rollout a fixed split measured 2-3x slower than one launch over all grids (306 vs 95 us per
 * joint step, profiles/r02_bench_group_policies.log): these launches already scale with the CUs they get, a split only
 * fragments the chip.  One ordinary stream per engine: 128 / 176 us against 95 / 155 us (same log).
 */
#ifndef OVERCOOKED_EXPERIMENTS_H
#define OVERCOOKED_EXPERIMENTS_H

#include "overcooked_ops.h"

#ifdef __cplusplus
extern "C" {
#endif

#ifdef OC_DIAGNOSTICS
/* n HIP streams restricted to disjoint sets of the device's CUs, stream i getting about shares[i] / sum(shares) of them (at
 * least one; every set spread over the whole CU index range).  out_streams: n hipStream_t (as void*), out_cus (optional): CUs
 * given to each.  Destroy with oc_exp_cu_partition_destroy. */
int oc_exp_cu_partition_create(int device_id, const float *shares, int32_t n, void **out_streams, int32_t *out_cus);
int oc_exp_cu_partition_destroy(void *const *streams, int32_t n);
/* oc_policy_forward sized for a SHARE of the GPU: cu_budget = number of CUs the stream may use (0 = the whole device).  One
 * workgroup per CU of the budget; plan the tile table for it (oc_policy_plan_tiles(..., -cu_budget, ...)). */
int oc_exp_policy_forward_cus(const oc_policy_shape *shape, const void *d_packed, int32_t n_nets, const int8_t *d_obs, int64_t n_rows,
                              const int32_t *d_tiles, int32_t n_tiles, float *d_raw, int32_t *d_actions, float *d_action_logp,
                              float *d_logp, int sample_mode, uint32_t seed, uint32_t counter, const uint32_t *d_counter,
                              const int32_t *d_row_select, int32_t cu_budget, void *stream);
/* Size an engine's launches for a stream that may only use `cus` of the GPU's CUs: the persistent grid becomes waves_per_cu x cus
 * wavefronts.  0 = the whole device (default). */
int oc_exp_set_cu_budget(oc_engine *e, int32_t cus);
/* oc_group_policy_rollout with every engine on its OWN stream (ordinary streams, or CU-masked ones with their cu_budgets): engine
 * i runs its whole rollout -- per step its policy launch and its own env launch -- on policy_streams[i], side by side with the
 * others and without synchronisation between engines; `stream` is forked once at the start and joined once at the end.  Same
 * results as oc_group_policy_rollout. */
int oc_exp_group_policy_rollout_streams(oc_group *g, const oc_group_policy_slot *slots, void *const *policy_streams, const int32_t *cu_budgets,
                                        int64_t T, int in_place, int sample_mode, uint32_t seed, uint32_t counter0, void *stream);
#endif

#ifdef __cplusplus
}
#endif
#endif /* OVERCOOKED_EXPERIMENTS_H */
