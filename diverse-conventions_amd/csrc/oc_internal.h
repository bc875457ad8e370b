// oc_internal.h -- helpers shared by the translation units of liboc_hip.so (not part of the C ABI).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdlib>

// What the PRODUCT library reads from the environment: OC_NO_JIT (serve the ahead-of-time generic kernels), OC_JIT_CACHE (where
// compiled code objects are kept), OC_DWORD_STORES (the dword-store kernels for 4-byte-aligned-only consumers' tests) and
// OC_POLICY_COOP (0: no cooperative blocks, for plan-equality checks; documented f32-rounding effect; 2: two leftover blocks of an
// eight-wavefront stretch shared by two pairs as well -- A/B runs) and OC_POLICY_DUO (0 / 1: never /
// wherever it fits the two-blocks-per-wavefront policy kernel; bit-identical outputs either way) -- none of them changes an
// environment result.  Every experiment / timing knob (store modes, code-generation switches, phase ablation, traces, the folded env
// step ...) is read through OC_DIAG_ENV, which is getenv only in the diagnostics build (-DOC_DIAGNOSTICS -> csrc/liboc_hip_diag.so,
// loaded by tools/ through OC_DIAGNOSTICS=1): in the product library the names do not even exist as strings.
#ifdef OC_DIAGNOSTICS
#define OC_DIAG_ENV(name) getenv(name)
#else
#define OC_DIAG_ENV(name) (static_cast<const char *>(nullptr))
#endif

int oc_internal_fail(int code, const char *fmt, ...);      // overcooked_engine.hip: sets oc_last_error()
struct oc_group;
struct oc_engine;
oc_engine *oc_internal_group_engine(oc_group *g, int i);    // overcooked_engine.hip: engine i of a group (NULL if out of range)
int oc_internal_set_cu_budget(oc_engine *e, int cus);        // size the engine's launches for a CU-masked stream (0 = whole device)

#ifdef OC_DIAGNOSTICS
// ---- the env step folded into the policy launch (policy_ops.hip: oc_policy_rollout, "one launch per training step"): built, bit-exact,
// measured 2x slower than two launches -- kept for tools/exp_fold_ab.py in the diagnostics build only ----------
// A fold group is kFoldKitchens consecutive kitchens, stepped inside the policy kernel by the wavefront whose block completed
// the group's actions, as kFoldKitchens / kFoldKpw sub-groups of the generic step body (oc_step_kernel.h).
namespace oc_dev { struct StepArgs; }
constexpr int kFoldKitchens = 32, kFoldKpw = 16, kFoldPasses = 2, kFoldQN = 4;
// Step arguments of one step of `e` with the fold geometry (its state, tables, the given action / output buffers); fresh != 0:
// the destination is a new trajectory slot.  OC_ERR_UNSUPPORTED when this engine cannot be folded (players != 2, records too
// large for the fold's prefetch window); *lds_bytes: LDS the body needs per wavefront.
int oc_internal_fold_args(oc_engine *e, const void *d_actions, int8_t *d_obs, int32_t *d_reward, int32_t *d_done, oc_dev::StepArgs *out,
                          size_t *lds_bytes);
// Arrival counters and expected counts of the fold groups (device, engine-owned, allocated on first use; counters are zero
// between launches), and a device-resident copy of the engine's fold step arguments with NULL action / output pointers: the
// kernel's env call takes that template by pointer and the per-step pointers by value, which keeps a 250-byte struct out of its
// call frames.
int oc_internal_fold_workspace(oc_engine *e, int32_t **d_count, int32_t **d_expected, int32_t *n_groups, const oc_dev::StepArgs **d_args_template);
#endif  // OC_DIAGNOSTICS
void oc_internal_note_destination(oc_engine *e, const int8_t *d_obs);   // the engine's "previous destination" (store mode choice of oc_step)
// Rollouts with the policies in the loop: the planes a step writes into a fresh (time-major) slot are read by the very next launch, the
// policy kernel.  While the flag is set such steps store them plainly -- they stay in L2 for that launch, and their write-back hides
// under it -- instead of streaming them past the caches (tools/exp_collect_store_mode.py: 67.3 -> 61.4 us per step at 65 535 kitchens).
void oc_internal_planes_read_next(oc_engine *e, int on);

// Every launching entry point runs with the GPU that owns its buffers current, whatever the caller's (or torch's)
// current device is, and puts the caller's device back on return: kernels are enqueued on streams of that GPU, and
// CU counts / LDS opt-ins are taken from it.  Costs one hipGetDevice when the device is already current.
struct OcDeviceScope {
    int prev = -1;
    bool switched = false;
    bool ok = true;
    explicit OcDeviceScope(int dev) {
        if (dev < 0) return;
        if (hipGetDevice(&prev) != hipSuccess) { ok = false; return; }
        if (prev != dev) {
            ok = hipSetDevice(dev) == hipSuccess;
            switched = ok;
        }
    }
    ~OcDeviceScope() {
        if (switched) (void)hipSetDevice(prev);
    }
    OcDeviceScope(const OcDeviceScope &) = delete;
    OcDeviceScope &operator=(const OcDeviceScope &) = delete;
};

// GPU that owns a device allocation (-1: not a device pointer / unknown): one runtime query, well under a
// microsecond -- used by the entry points that have no engine handle to take the device from.
inline int oc_device_of_pointer(const void *p) {
    if (!p) return -1;
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
        (void)hipGetLastError();
        return -1;
    }
    return (attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged) ? attr.device : -1;
}
