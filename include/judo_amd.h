/*
 * judo_amd.h -- C ABI of libjudo_amd.so: the MI355X (gfx950) sampling-MPC rollout engine.
 *
 * Drop-in boundary for the hot path of bdaiinstitute/judo v0.0.7
 * (sample -> clip -> spline -> rollout -> cost -> update).  The reference has no C ABI of its own for
 * this path: the seam is three Python ABCs (Optimizer, RolloutBackend, Task) plus the pybind11 module
 * mujoco_extensions/policy_rollout.  Each entry point below names the reference interface it replaces;
 * INTEGRATION.md shows the ctypes binding a judo maintainer would add.
 *
 * Conventions
 *   - every `const float*` / `float*` is a DEVICE pointer (fp32, contiguous) unless marked HOST;
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream); all work is enqueued
 *     asynchronously on it, nothing synchronises;
 *   - the caller owns every buffer; the library owns only jh_model handles (model constants in device
 *     memory) -- no hidden allocation on the step path;
 *   - return 0 on success, a negative jh_status otherwise; jh_last_error() gives the message
 *     (thread-local).  Functions are re-entrant per jh_model.
 *   - rollouts are indexed n = 0..N-1 locally, n_offset + n globally; global sample 0 is the
 *     unperturbed nominal (judo/optimizers/mppi.py:59 `concatenate([nominal[None], noised])`).
 *   - noise layout is (K, nu, ldn) with the rollout index fastest (64 consecutive lanes read 256
 *     contiguous bytes); element (k,u,n) at noise[(k*nu+u)*ldn + n].
 */
#ifndef JUDO_AMD_H
#define JUDO_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct jh_model jh_model;

enum jh_status {
  JH_OK = 0,
  JH_ERR_INVALID = -1,   /* bad argument (shape, null pointer, unsupported size) */
  JH_ERR_HIP = -2,       /* HIP runtime error */
  JH_ERR_UNSUPPORTED = -3,
  JH_ERR_BLOB = -4       /* malformed model blob */
};

enum jh_task_kind { JH_TASK_CARTPOLE = 0, JH_TASK_CYLINDER_PUSH = 1, JH_TASK_LEAP_CUBE = 2, JH_TASK_FR3_PICK = 3 };
enum jh_spline_kind { JH_SPLINE_ZERO = 0, JH_SPLINE_LINEAR = 1, JH_SPLINE_CUBIC = 3 };

#define JH_MAX_TASK_PARAMS 32
#define JH_MAX_KNOT_DIM 512 /* K * nu */
#define JH_MAX_ELITES 32

const char* jh_last_error(void);
int jh_version(void);

/* Model constants (the MjModel the reference deep-copies per thread, judo/utils/mj_rollout_backend.py:41):
 * `blob` is a HOST buffer produced by judo_amd.models.pack_model(); it is parsed and uploaded to `device`. */
int jh_model_create(const void* blob, size_t nbytes, int device, jh_model** out);
void jh_model_destroy(jh_model* m);
/* dims[0..5] = nq, nv, nu, nsensordata, task kind, ntaskparams */
int jh_model_dims(const jh_model* m, int* dims /* HOST */);

/* Diagnostics accumulated by the articulated-body kernels since the last reset (synchronises the device):
 * out[0] contacts dropped because a rollout exceeded the per-rollout contact capacity, out[1] constraint solves that hit the
 * Newton iteration cap, out[2] Newton iterations (summed over rollouts), out[3] physics steps (summed over rollouts); leap_cube kernel generation 3 also:
 * out[4] Newton iterations executed by wavefronts (four rollouts advance in lock step: per step the maximum over the four), out[5] physics steps
 * summed over wavefronts; out[6] launches that ran without their overflow rows because the device's memory pool refused the scratch block (the contact capacity was then what
 * the LDS pool holds), out[7] reserved (0).  The counters are 32-bit and wrap: reset them at least every ~10^9 rollout-steps.  HOST pointer. */
int jh_model_stats(jh_model* m, int* out /* HOST, 8 ints */, int reset);

/* (The cross-check kernel generations of the parity tests and their registration hook are NOT part of this interface: include/judo_amd_xcheck.h.) */

/* leap_cube on kernel generation 3: model the hand's own contacts (every finger-finger / finger-palm geom pair MuJoCo's filters leave: same welded body,
 * parent-child, the 18 <exclude> pairs of leap_components/params_and_default.xml:76-101) next to the cube's -- the default, MuJoCo collides them -- or,
 * with on = 0, the cube's contacts alone (what generations 1 and 2 model). */
int jh_model_set_self_collision(jh_model* m, int on);

/* leap_cube family, kernel generation 3: contacts a rollout can hold.  48 (default: all in LDS) or 64 (a second build of the kernel, jh_engine_v5_cap64.hip: the 16 above the LDS
 * pool in a per-rollout row of global memory; 2.8 % slower on every plan step).  The headline workload drops 2e-6 contacts per rollout-step at 48; the shipped leap_cube_down and
 * caltech_leap_cube workloads 2e-4 .. 4e-4, which is why judo_amd selects 64 for those two models.  jh_model_limits out[3] reports the setting. */
int jh_model_set_contact_capacity(jh_model* m, int contacts);

/* Traces without a second rollout (judo/controller/controller.py:323-363, `update_traces`: line segments of the best rollouts' `trace*` framepos sensors).  The
 * reference reads them out of the sensor array its rollout materialises for every sample; the fused path has no such array, so round 1-2 re-rolled the elites in
 * materialise mode when the traces were read (8 ms on the headline workload: one lone wave for 64 serial steps).  jh_rollout_cost_traced (jh_rollout_cost with one more argument) on
 * leap_cube / fr3_pick (kernel generation 3), cartpole and cylinder_push also writes the trace sensors of EVERY rollout and step into `trace`: row n (rollout index within the launch) holds
 * H x out[1] floats, the sensors out[0] .. out[0] + out[1] - 1 of jh_model_trace_layout (leap_cube: 16, 15 = five site positions; fr3_pick: 8, 6; cartpole,
 * cylinder_push: 0, 6; out[1] = 0: the model's kernel writes none).  out[2] = 1: the buffer is column-major -- element i of rollout n at [i * N + n] (the one-lane-
 * per-rollout kernels of cartpole / cylinder_push: coalesced) -- else row-major, [n * H * out[1] + i].  60 B per rollout-step, a few hundred MB per plan step that
 * nobody reads back except the elites' rows: jh_trace_gather copies, for each record [cost, global index (bits), ...] of a jh_topk_partial result, the row of that
 * rollout behind cost and index -- the payload the ranks exchange. */
int jh_model_trace_layout(const jh_model* m, int* out /* HOST, 3 ints */);
int jh_trace_gather(const float* rec_in /* DEVICE, k x stride_in */, int k, int stride_in, int n_offset, int n_local, const float* trace /* DEVICE */, int row_floats,
                    int colmajor, float* rec_out /* DEVICE, k x (2 + row_floats) */, void* stream);

/* Small launches (latency mode): when N rollouts would leave SIMDs idle, the cooperative kernels (leap_cube, fr3_pick: four rows of 16 lanes per wave; Spot tree: two
 * rows of 32) let 4 or 2 rows of a wave compute the same rollout -- identical arithmetic, the first row writes -- so that a wave does not wait for the slowest of
 * its different rollouts' Newton solves in every step.  Results are bit-identical to the full mapping.  Chosen per launch from N and the CU count; the environment
 * variable JUDO_AMD_LATENCY_SHIFT=0 switches it off (1 / 2: force two / four copies). */
/* Limits of this model's kernels: out[0] = largest knot count K the fused kernel (jh_rollout_cost) accepts -- the cooperative fr3_pick kernel
 * keeps a lane's knots on chip (8 of them); the leap_cube kernel of generation 3 reads them from memory every step and is bounded by
 * JH_MAX_KNOT_DIM / nu alone.  A larger K goes through jh_spline_controls + jh_rollout_materialize + jh_task_reward, which have no such limit;
 * out[1] = JH_MAX_KNOT_DIM; out[2] = JH_MAX_ELITES; out[3] = contact capacity per rollout of the general pool (leap_cube generation 3: 48 or 64, jh_model_set_contact_capacity;
 * fr3_pick generation 3: 96 -- 32 on chip, 64 in a row of global memory -- next to its 96 pad-against-pad slots; 0 = the model has at most one contact).  HOST pointer. */
int jh_model_limits(const jh_model* m, int* out /* HOST, 4 ints */);
/* out[0] of jh_model_limits is an upper bound over all horizons.  The one-lane kernels (cartpole, cylinder_push, kernel generation 1) stage W (H x K) and
 * their lanes' knots in LDS, so their largest fused K also depends on the horizon: this returns the K up to which jh_rollout_cost accepts a launch of H
 * steps (>= 0), or a negative jh_status.  A Controller asks before every plan step and takes the materialise path above it (a live num_nodes / horizon
 * edit, judo/optimizers/base.py:15-21, must not raise out of update_action). */
int jh_model_max_fused_knots(const jh_model* m, int H);

/* Plan-step I/O in one call each (the two transfers of a plan step: < 2 KB down, the new nominal knots up): an asynchronous copy of `nbytes` from
 * pinned HOST memory to the device on `stream`; and an asynchronous copy from the device to pinned HOST memory followed by a wait for `stream`
 * (the only synchronisation of an optimiser iteration, judo/controller/controller.py:283 needs the new nominal on the host). */
int jh_upload_async(void* dst_device, const void* src_host, size_t nbytes, void* stream);
int jh_download_wait(void* dst_host, const void* src_device, size_t nbytes, void* stream);
/* The same in two halves: `begin` enqueues the copy and marks its end on `stream`; work enqueued afterwards (the trace records of update_traces)
 * runs behind it; `end` waits for the mark only.  Marks are kept per host thread and stream (a thread may drive controllers on several GPUs) and `end`
 * waits for the oldest mark of the calling thread that has not been waited for yet; a stream carries one mark at a time. */
int jh_download_begin(void* dst_host, const void* src_device, size_t nbytes, void* stream);
int jh_download_end(void);

/* Fused plan-step kernel.  Replaces, for N rollouts in one launch:
 *   Optimizer.sample_control_knots         judo/optimizers/{mppi.py:38-59,ps.py:29-50,cem.py:55-74}
 *   np.clip to actuator_ctrlrange          judo/controller/controller.py:253-257
 *   make_spline(...)(t + dt*arange(H))     judo/controller/controller.py:261-262  (as the H x K matrix W)
 *   RolloutBackend.rollout                 judo/utils/mj_rollout_backend.py:45-88 (mj_step x H)
 *   Task.reward                            judo/tasks/{cartpole.py:42,cylinder_push.py:50,leap_cube.py:63,fr3_pick.py:225}
 * costs[n] = -reward of rollout n.  knots_out (optional, may be NULL) receives the clipped candidates in
 * (K, nu, ldn) layout.  `phase` is the fr3_pick phase chosen on the host by Task.pre_rollout
 * (judo/tasks/fr3_pick.py:191-223); ignored by the other tasks. */
int jh_rollout_cost(const jh_model* m, const float* x0, const float* nominal, const float* noise, int ldn, const float* sigma,
                    const float* W, const float* ctrl_lo_hi, const float* task_params, int phase, int N, int n_offset, int H, int K,
                    float* costs, float* knots_out, void* stream);
int jh_rollout_cost_traced(const jh_model* m, const float* x0, const float* nominal, const float* noise, int ldn, const float* sigma,
                    const float* W, const float* ctrl_lo_hi, const float* task_params, int phase, int N, int n_offset, int H, int K,
                    float* costs, float* knots_out, float* trace /* DEVICE, N x H x jh_model_trace_layout out[1] floats, or NULL */, void* stream);

/* Drop-in RolloutBackend.rollout(x0, controls) -> (states, sensors)   judo/utils/rollout_backend.py:20-39.
 * controls (N,H,nu), states (N,H,nq+nv), sensors (N,H,nsensordata), all row-major; x0 is (nq+nv) or, when
 * x0_batched, (N,nq+nv).  states[n,h] is the state AFTER applying controls[n,h]; sensors[n,h] is what the
 * forward pass at the start of that step computed (MuJoCo semantics). */
int jh_rollout_materialize(const jh_model* m, const float* x0, int x0_batched, const float* controls, int N, int H, float* states,
                           float* sensors, void* stream);

/* Task.reward on materialised device arrays (same task_params as jh_rollout_cost); rewards[n] (not negated). */
int jh_task_reward(const jh_model* m, const float* states, const float* sensors, const float* controls, const float* task_params,
                   int phase, int N, int H, float* rewards, void* stream);

/* The optimizers' noise, `np.random.randn(num_rollouts - 1, num_nodes, nu)` of judo/optimizers/{mppi.py:52,ps.py:43,cem.py:67}, drawn on the device in the
 * kernels' layout: out[row * ldn + n] for row < rows (= K * nu) and local rollout n < n_local is the standard normal that belongs to GLOBAL rollout
 * n_offset + n of draw number `draw` under `seed` (Philox4x32-10, counter-based: a pure function of seed, draw, row and the global rollout index, two
 * Box-Muller pairs per block of four rollouts).  A rank generates exactly its shard's columns; the plan does not depend on the number of GPUs. */
int jh_noise_normal(unsigned long long seed, unsigned int draw, int rows, int n_offset, int n_local, float* out, int ldn, void* stream);

/* Optimizer.sample_control_knots for callers that want the (N,K,nu) row-major array itself
 * (row n = nominal + sigma * noise[:, :, n], row 0 of global index 0 = nominal), optionally clipped. */
int jh_sample_knots(const float* nominal, const float* noise, int ldn, const float* sigma, const float* ctrl_lo_hi /* NULL = no clip */,
                    int N, int n_offset, int K, int nu, float* knots_nku, void* stream);

/* Candidate control trajectories of the materialise path (judo/controller/controller.py:239-249: the candidate splines evaluated at
 * the rollout times): controls[n,h,u] = sum_k W[h,k] * knot(n,k,u), (N,H,nu) row-major, ready for jh_rollout_materialize.
 * W is the (H,K) interpolation matrix of the spline kind (linear in the knots for zero/linear/cubic interp1d); knots come from
 * `knots_nku` or, when it is NULL, are recomputed as clip(nominal + sigma*noise) exactly like jh_rollout_cost does. */
int jh_spline_controls(const float* W, const float* knots_nku, const float* nominal, const float* noise, int ldn, const float* sigma,
                       const float* ctrl_lo_hi, int N, int n_offset, int H, int K, int nu, float* controls, void* stream);

/* Moments of this shard's candidate knots for the "running" action normaliser (RunningMeanStdNormalizer.update on candidate_knots,
 * judo/utils/normalization.py:176-200, judo/controller/controller.py:290-291): out[u] = sum over rollouts and knots of (x - center[u]),
 * out[nu+u] = sum of (x - center[u])^2, x = knot of actuator u (same knot sources as jh_spline_controls).  `out` (2*nu floats) is zeroed first. */
int jh_knot_moments(const float* knots_nku, const float* nominal, const float* noise, int ldn, const float* sigma, const float* ctrl_lo_hi,
                    const float* center, int N, int n_offset, int K, int nu, float* out, void* stream);

/* MPPI.update_nominal_knots (judo/optimizers/mppi.py:61-82), shard-local part.
 * Writes one record rec[0] = beta = min cost, rec[1] = S = sum exp(-(c-beta)/lambda), rec[2..2+K*nu) = sum w*knots.
 * Knots come either from `knots_nku` ((N,K,nu) row-major, the drop-in path) or, when it is NULL, are recomputed as
 * clip(nominal + sigma*noise) from the (K,nu,ldn) noise (the fused path: noise is read once more, never stored).
 * `scratch` must hold jh_update_scratch_floats(N,K,nu) floats. */
size_t jh_update_scratch_floats(int N, int K, int nu);
int jh_mppi_partial(const float* costs, const float* knots_nku, const float* nominal, const float* noise, int ldn, const float* sigma,
                    const float* ctrl_lo_hi, int N, int n_offset, int K, int nu, float lambda, float* scratch, float* rec, void* stream);
/* Merge G shard records (after the all-gather) with a log-sum-exp rescale: nominal_out = sum_g e_g V_g / sum_g e_g S_g,
 * e_g = exp(-(beta_g - min beta)/lambda).  Identical result on every rank. */
int jh_mppi_merge(const float* recs, int G, int K, int nu, float lambda, float* nominal_out, void* stream);

/* CrossEntropyMethod / PredictiveSampling updates (judo/optimizers/cem.py:76-92, ps.py:52-65), shard-local part:
 * the k best rollouts (largest reward = smallest cost; ties: higher global index first when tie_high != 0, which is
 * what flip(argsort(.)) yields for a stable sort, lower index first otherwise = np.argmax), each as a record
 * [cost, global index, knots(K*nu)].  rec holds k such records. */
int jh_topk_partial(const float* costs, const float* knots_nku, const float* nominal, const float* noise, int ldn, const float* sigma,
                    const float* ctrl_lo_hi, int N, int n_offset, int K, int nu, int k, int tie_high, float* scratch, float* rec,
                    void* stream);
/* The whole update of a ONE-GPU plan step in one launch (Controller.update_action's tail, judo/controller/controller.py:288-299): what jh_mppi_partial + jh_mppi_merge
 * (mode 0) or jh_topk_partial + jh_elite_merge (mode 1: k elites, raw population std into sigma_out, which may be NULL) compute, plus -- when E > 0 -- the records
 * [cost, global index (bits), trace row] of the E best rollouts (ties: higher index first) that jh_topk_partial + jh_trace_gather produce, into trace_out (E x (2 + row_floats)).
 * Every workgroup writes its partial records to `scratch` and the last one to finish merges them: same arithmetic in the same order as the separate calls, bit-identical
 * outputs.  `scratch`: jh_update_fused_scratch_floats(N, K, nu) floats, ZERO before its first use (it holds the ticket counter, which every launch leaves at zero). */
size_t jh_update_fused_scratch_floats(int N, int K, int nu);
int jh_update_fused(const float* costs, const float* knots_nku, const float* nominal, const float* noise, int ldn, const float* sigma, const float* ctrl_lo_hi, int N,
                    int n_offset, int K, int nu, int mode, float lambda, int k, int tie_high, int E, const float* trace, int row_floats, int colmajor, float* scratch,
                    float* nominal_out, float* sigma_out, float* trace_out, void* stream);
/* One iteration of Controller.update_action's loop (judo/controller/controller.py:250-299) on ONE GPU as one call: upload of the packed host block
 * [x0 | nominal | sigma | task params | ctrl bounds] (float offsets o_* into it) into blk_dev, jh_rollout_cost_traced, jh_update_fused with its outputs at
 * out = [nominal K*nu | sigma K*nu | E x (2 + row_floats) trace records] (device memory, or device-visible pinned host memory: then no download is needed), and the
 * completion mark of jh_download_begin (zero bytes).  Asynchronous on `stream`; jh_download_end waits.  `trace` NULL: no trace records (E ignored).
 * Round 6, for plan steps whose kernels are a few tens of microseconds (cartpole, cylinder_push):
 *  - those models run rollout + cost + update as ONE launch (the update's tail inside the rollout kernel's launch: same bits as the two launches);
 *  - blk_dev == blk_host (a device-visible pinned host block): no upload, the kernel reads the block in place;
 *  - out_host_mark != out: a 4-byte word in device-visible pinned host memory; the update's last workgroup stores (its value at the time of this call) + 1 there behind
 *    the results and jh_download_end polls that word instead of waiting for the stream's event (one outstanding plan step per word).  out_host_mark == out: the event. */
int jh_plan_step(const jh_model* m, void* blk_dev, const void* blk_host, size_t blk_bytes, int o_nominal, int o_sigma, int o_tp, int o_lohi, const float* noise, int ldn,
                 const float* W, int phase, int N, int n_offset, int H, int K, float* costs, float* knots_out, float* trace, int mode, float lambda, int k, int tie_high, int E,
                 int row_floats, int colmajor, float* scratch, float* out, void* out_host_mark, void* const* timing, void* stream);
/* The same iteration with the rollouts sharded over G ranks (the reference has no multi-process form: judo/controller/controller.py:246-299 runs in one
 * process): launch -> all-gather -> merge.  jh_update_shard is jh_update_fused with the last stage left to the ranks' merge: it writes this rank's record
 *   [ MPPI: beta, S, V(K*nu)  |  elites: k x (cost, global index (bits), knots(K*nu)) ]  followed by  E x (cost, global index (bits), trace row(row_floats))
 * (jh_shard_record_floats floats) into rec_out; jh_shard_merge takes the G all-gathered records (rank-major) and writes nominal_out / sigma_out (raw population
 * std, may be NULL) / trace_out (the E best of the G * E trace records, best first, ties: higher index first) exactly as jh_update_fused does on one GPU.  Every rank
 * runs the same merge on the same bytes: identical nominal everywhere, no broadcast.  jh_plan_step_shard = upload + jh_rollout_cost_traced + jh_update_shard in one
 * call (timing: 3 events as for jh_plan_step); jh_plan_merge = jh_shard_merge into out = [nominal | sigma | E trace records] + the completion mark jh_download_end
 * waits for (timing_done: one event recorded behind the merge, or NULL). */
size_t jh_shard_record_floats(int K, int nu, int mode, int k, int E, int row_floats);
int jh_update_shard(const float* costs, const float* knots_nku, const float* nominal, const float* noise, int ldn, const float* sigma, const float* ctrl_lo_hi, int N,
                    int n_offset, int K, int nu, int mode, float lambda, int k, int tie_high, int E, const float* trace, int row_floats, int colmajor, float* scratch,
                    float* rec_out, void* stream);
int jh_shard_merge(const float* recs, int G, int K, int nu, int mode, float lambda, int k, int tie_high, int E, int row_floats, float* nominal_out, float* sigma_out,
                   float* trace_out, void* stream);
int jh_plan_step_shard(const jh_model* m, void* blk_dev, const void* blk_host, size_t blk_bytes, int o_nominal, int o_sigma, int o_tp, int o_lohi, const float* noise, int ldn,
                       const float* W, int phase, int N, int n_offset, int H, int K, float* costs, float* knots_out, float* trace, int mode, float lambda, int k, int tie_high,
                       int E, int row_floats, int colmajor, float* scratch, float* rec_out, void* const* timing, void* stream);
int jh_plan_merge(const float* recs, int G, int K, int nu, int mode, float lambda, int k, int tie_high, int E, int row_floats, float* out, void* out_host_mark,
                  void* timing_done, void* stream);
/* `timing`: NULL, or three events of jh_event_create recorded on `stream` before the rollout kernel, between it and the update, and behind the update
 * (what bench.py's roofline leg reads: jh_event_elapsed_ms waits for its second event). */
int jh_event_create(void** out);
void jh_event_destroy(void* ev);
int jh_event_record(void* ev, void* stream);
int jh_stream_wait_event(void* stream, void* ev); /* work enqueued on `stream` after this call waits (on the device) for `ev`, recorded on another stream */
int jh_event_elapsed_ms(void* a, void* b, float* ms);
/* Merge G*k records -> global k elites -> mean and clipped population std (ddof 0).  sigma_out may be NULL (PS, k=1). */
int jh_elite_merge(const float* recs, int G, int k, int K, int nu, int tie_high, float sigma_min, float sigma_max, float* nominal_out,
                   float* sigma_out, void* stream);

/* ---- policy half of the Spot policy rollout (mujoco_extensions/system/system_class.cpp:125-238; pybind entry
 * mujoco_extensions/policy_rollout/pybind/policy_rollout.cpp).  One call = System::setObservation + System::policyInference for N rollouts:
 * the 84-d observation from each rollout's state (row stride ld, qpos at 0, qvel at nq; free base at base_qpos / base_qvel, the 19 joints at
 * leg_qpos / leg_qvel), its 25-d command and its previous policy output; the actor 84-512-256-128-12 (Gemm + Elu of spot_locomotion.onnx, weights
 * passed in ONNX layout [out, in]) on exact-f32 MFMA tiles; the mapping to the 19 joint targets.  policy_out (N x 12) is read as the previous output
 * and overwritten with the new one; control is (N x 19); scratch holds jh_policy_scratch_floats(N) floats.  The physics substeps between two policy
 * steps are not part of this call. */
typedef struct jh_policy jh_policy;
int jh_policy_create(const float* const* weights, const float* const* biases, jh_policy** out);
void jh_policy_destroy(jh_policy* p);
size_t jh_policy_scratch_floats(int N);
int jh_policy_step(const jh_policy* p, const float* states, int ld, int nq, int base_qpos, int base_qvel, int leg_qpos, int leg_qvel,
                   const float* command, float* policy_out, float* control, float* scratch, int N, void* stream);

/* ---- physics half of the Spot policy rollout (System::rollout's inner mj_step loop, mujoco_extensions/system/system_class.cpp:300-318): advance
 * N rollouts of a floating-base robot on a ground plane by `substeps` engine steps with the control held.  The model image is what
 * judo_amd/tree_model.py packs (free base + 19 hinges in 5 chains, plane contacts, pyramidal cones, implicitfast).  state_in / state_out are
 * (N x 51) rows [qpos(26), qvel(25)] and may alias; ctrl is (N x 19) joint position targets; warmstart (N x 25, may be NULL) is the solver's
 * starting acceleration, read and overwritten with this call's last constraint-consistent acceleration (mjData.qacc_warmstart).  sensors_out (N x nsensordata,
 * may be NULL): mjData.sensordata as mj_step leaves it after the last step, i.e. the site positions / frame axes of that step's forward pass.
 * jh_tree_stats: [contacts dropped over capacity, steps at the iteration cap, Newton iterations, steps]. */
typedef struct jh_tree jh_tree;
int jh_tree_create(const void* blob, size_t nbytes, jh_tree** out);
void jh_tree_destroy(jh_tree* t);
int jh_tree_stats(jh_tree* t, int* out4, int reset);
/* The robot against itself (judo/models/xml/spot_primitive/contact.xml:4-14: every robot geom pair MuJoCo's static filters and the 11 <exclude> body pairs leave, listed
 * in the model image): on by default when the image lists pairs, as MuJoCo collides them; on = 0: the ground contacts only (rounds 1-4). */
int jh_tree_set_self_collision(jh_tree* t, int on);
int jh_tree_dims(const jh_tree* t, int* out4 /* nq, nv, joints (= controls), nsensordata */);
int jh_tree_substeps(const jh_tree* t, const float* state_in, const float* ctrl, float* warmstart, int N, int substeps, float* state_out, float* sensors_out,
                     void* stream);

/* ---- the whole of threaded_rollout (mujoco_extensions/system/system_class.cpp:277-367; pybind entry mujoco_extensions/policy_rollout/pybind/policy_rollout.cpp:65)
 * in the reference's array layouts: x0 is one (51) state (x0_batched = 0) or (N x 51); commands (N x T x 25); states (N x T x 51), row [n][i] = the state after
 * command row i's policy step and `substeps` engine steps, sensors (N x T x nsensordata, may be NULL) the sensordata of the same row; policy_out (N x 12) in/out (last_policy_output -> policy_outputs).  warmstart (N x 25, may be NULL)
 * carries mjData.qacc_warmstart from call to call as the reference's per-thread mjData does; reset_warmstart != 0 zeroes it before every control step instead.
 * cutoff_seconds >= 0: the rollout stops issuing command rows once that much DEVICE time has passed since the call started (checked against the control
 * step two back) and the remaining rows repeat the last computed state, as System::rollout does with its wall clock; < 0: no deadline.  *steps_done = rows
 * computed.  scratch holds jh_policy_rollout_scratch_floats(N) floats.  With a deadline the call synchronises with the stream two control steps behind. */
size_t jh_policy_rollout_scratch_floats(int N);
int jh_policy_rollout(const jh_policy* p, jh_tree* t, const float* x0, int x0_batched, const float* commands, float* policy_out, float* warmstart, int reset_warmstart,
                      int N, int T, int substeps, double cutoff_seconds, float* states, float* sensors, float* scratch, int* steps_done, void* stream);

#ifdef __cplusplus
}
#endif
#endif
