/*
 * overcooked_debug.h -- diagnostics; not part of the drop-in boundary.  Two kinds:
 *   - the compile-only checks of the hiprtc kernel builds: in every build of the library (CPU test-suite, __graft_entry__.build());
 *   - timing diagnostics (phase stamps, the folded env step's counter): ONLY in the diagnostics build, -DOC_DIAGNOSTICS ->
 *     csrc/liboc_hip_diag.so, which tools/ load with OC_DIAGNOSTICS=1.  That build also reads the experiment knobs (OC_ABLATE,
 *     OC_STORE_MODE, OC_POLICY_FOLD ...) from the environment; the product library liboc_hip.so contains neither these entry
 *     points nor the code behind the knobs, and reads none of them.
 */
#ifndef OVERCOOKED_DEBUG_H
#define OVERCOOKED_DEBUG_H

#include "overcooked_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Compile-only check of the per-engine (hiprtc) kernel build, no GPU needed: OC_OK and the code object size, or the compiler
 * log (tests/test_abi.py, __graft_entry__.build()). */
int oc_debug_jit_compile(const oc_layout_params *p, int64_t n_kitchens, const char *arch, int64_t *code_bytes, char *log_out, int64_t log_cap);
/* The same for a group module (one launch over n_engines engines): plans the geometry for `num_cus` CUs and compiles.
 * plan_out (optional, 2 * n_engines ints): kitchens per wavefront group and blocks of every engine. */
int oc_debug_jit_compile_group(const oc_layout_params *params, const int64_t *n_kitchens, int32_t n_engines, int32_t num_cus, const char *arch,
                               int64_t *code_bytes, int32_t *plan_out, char *log_out, int64_t log_cap);
/* Ahead-of-time builds (no GPU needed; __graft_entry__.build() calls them for the BASELINE shapes): every hiprtc module an engine of
 * this shape / a group of these engines loads, compiled for `arch` as the device reports it ("gfx950:sramecc+:xnack-") and written to
 * out_dir/<hash>.hsaco.  The library looks in <its own directory>/aot/ before it compiles anything (oc_kernel_build_info then says
 * "ahead of time").  *n_modules (optional): code objects written for the engine. */
int oc_debug_aot_build(const oc_layout_params *params, int64_t n_kitchens, const char *arch, const char *out_dir, int32_t *n_modules);
int oc_debug_aot_build_group(const oc_layout_params *params, const int64_t *n_kitchens, int32_t n_engines, int32_t num_cus, const char *arch,
                             const char *out_dir);
/* hiprtc compilations this process has run so far (code objects found in the process / ahead-of-time / disk caches do not count):
 * a compilation takes about a second, so a caller that times something reads this before and after to prove that none happened
 * inside its timed region (bench.py: `jit_builds_inside_timed_regions`). */
long long oc_debug_jit_builds(void);
#ifdef OC_DIAGNOSTICS
/* Per-wavefront phase stamps (s_memrealtime) of the step kernel's next launches (tools/trace_phases.py).  enable != 0: allocate
 * and zero the trace buffer; 0: copy [max_waves][16] stamps to host_out (may be NULL) and free it. */
int oc_debug_trace(oc_engine *e, int enable, unsigned long long *host_out, int64_t max_waves);
/* The same for the fused policy kernel (tools/trace_policy.py): [n_waves][32] stamps. */
int oc_policy_debug_trace(int enable, unsigned long long *host_out, int64_t n_waves);
/* Policy launches so far (this process) that carried the env step inside them (oc_policy_rollout's one-launch-per-step mode): lets a
 * test tell the folded path from its two-launch fallback. */
long long oc_debug_fold_launches(void);
#endif /* OC_DIAGNOSTICS */

#ifdef __cplusplus
}
#endif
#endif /* OVERCOOKED_DEBUG_H */
