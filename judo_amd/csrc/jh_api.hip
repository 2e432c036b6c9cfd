// jh_api.hip -- C-ABI entry points of libjudo_amd.so (argument checks, model handles, dispatch).
#include <cstdlib>
#include <cstring>

#include "jh_internal.h"
#include "jh_update_dev.h"

static thread_local char g_err[512] = "";

void jh_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* jh_last_error(void) { return g_err; }
extern "C" int jh_version(void) { return 100; }

extern "C" int jh_model_create(const void* blob, size_t nbytes, int device, jh_model** out) {
  JH_REQUIRE(blob && out, "model_create: null pointer");
  if (nbytes < sizeof(jh_blob_header)) { jh_set_error("model_create: blob too small (%zu bytes)", nbytes); return JH_ERR_BLOB; }
  jh_blob_header h;
  memcpy(&h, blob, sizeof(h));
  if (h.magic != JH_BLOB_MAGIC || h.version != JH_BLOB_VERSION) { jh_set_error("model_create: bad magic/version (%08x, %u)", h.magic, h.version); return JH_ERR_BLOB; }
  size_t need = sizeof(h) + 4 * ((size_t)h.nfloat + h.nint);
  if (nbytes != need) { jh_set_error("model_create: blob size %zu != expected %zu", nbytes, need); return JH_ERR_BLOB; }
  if (h.kind > JH_TASK_FR3_PICK) { jh_set_error("model_create: unknown task kind %u", h.kind); return JH_ERR_BLOB; }
  if (h.kind == JH_TASK_CARTPOLE && h.nfloat < CP_NPARAM) { jh_set_error("model_create: cartpole blob has %u floats, need %d", h.nfloat, CP_NPARAM); return JH_ERR_BLOB; }
  if (h.kind == JH_TASK_CYLINDER_PUSH && h.nfloat < CY_NPARAM) { jh_set_error("model_create: cylinder blob has %u floats, need %d", h.nfloat, CY_NPARAM); return JH_ERR_BLOB; }
  if (h.ntaskparam > JH_MAX_TASK_PARAMS) { jh_set_error("model_create: too many task params"); return JH_ERR_BLOB; }
  JH_HIP(hipSetDevice(device));
  jh_model* m = new jh_model();
  m->device = device; m->kind = (int)h.kind; m->nq = h.nq; m->nv = h.nv; m->nu = h.nu; m->ns = h.ns; m->ntaskparam = h.ntaskparam;
  m->nf = h.nfloat; m->ni = h.nint; m->d_f = nullptr; m->d_i = nullptr; m->d_stats = nullptr; m->kernel_gen = (h.kind == JH_TASK_LEAP_CUBE || h.kind == JH_TASK_FR3_PICK) ? 3 : 2; m->self_collision = 1; m->contact_capacity = 48;
  const char* p = (const char*)blob + sizeof(h);
  m->h_f.assign((const float*)p, (const float*)p + h.nfloat);
  m->h_i.assign((const int*)(p + 4 * (size_t)h.nfloat), (const int*)(p + 4 * (size_t)h.nfloat) + h.nint);
  {  // the cooperative kernels take a per-launch scratch block from the device's stream-ordered pool (contacts above the LDS pool): let the pool keep what it is
     // handed back instead of returning it to the driver at every synchronisation (the default release threshold is 0: 0.2 ms per launch)
    hipMemPool_t pool = nullptr; int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetDefaultMemPool(&pool, dev) == hipSuccess && pool) {
      unsigned long long keep = ~0ull;
      (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
    }
  }
  hipError_t e = hipMalloc(&m->d_f, 4 * (m->nf ? m->nf : 1));
  if (e == hipSuccess) e = hipMalloc(&m->d_i, 4 * (m->ni ? m->ni : 1));
  if (e == hipSuccess) e = hipMalloc(&m->d_stats, JH_NSTATS * sizeof(int));
  if (e == hipSuccess) e = hipMemset(m->d_stats, 0, JH_NSTATS * sizeof(int));
  if (e == hipSuccess && m->nf) e = hipMemcpy(m->d_f, m->h_f.data(), 4 * m->nf, hipMemcpyHostToDevice);
  if (e == hipSuccess && m->ni) e = hipMemcpy(m->d_i, m->h_i.data(), 4 * m->ni, hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    jh_set_error("model_create: device upload failed: %s", hipGetErrorString(e));
    if (m->d_f) (void)hipFree(m->d_f);
    if (m->d_i) (void)hipFree(m->d_i);
    if (m->d_stats) (void)hipFree(m->d_stats);
    delete m;
    return JH_ERR_HIP;
  }
  *out = m;
  return JH_OK;
}

extern "C" void jh_model_destroy(jh_model* m) {
  if (!m) return;
  if (m->d_f) (void)hipFree(m->d_f);
  if (m->d_i) (void)hipFree(m->d_i);
  if (m->d_stats) (void)hipFree(m->d_stats);
  delete m;
}

extern "C" int jh_model_dims(const jh_model* m, int* dims) {
  JH_REQUIRE(m && dims, "model_dims: null pointer");
  dims[0] = m->nq; dims[1] = m->nv; dims[2] = m->nu; dims[3] = m->ns; dims[4] = m->kind; dims[5] = m->ntaskparam;
  return JH_OK;
}

extern "C" int jh_model_stats(jh_model* m, int* out, int reset) {
  JH_REQUIRE(m && out, "model_stats: null pointer");
  JH_HIP(hipMemcpy(out, m->d_stats, 4 * sizeof(int), hipMemcpyDeviceToHost));
  JH_HIP(hipMemcpy(out + 4, m->d_stats + 20, 2 * sizeof(int), hipMemcpyDeviceToHost));
  out[6] = reset ? __atomic_exchange_n(&m->ovf_fallbacks, 0, __ATOMIC_RELAXED) : __atomic_load_n(&m->ovf_fallbacks, __ATOMIC_RELAXED); out[7] = 0;
  if (reset) JH_HIP(hipMemset(m->d_stats, 0, JH_NSTATS * sizeof(int)));
  return JH_OK;
}

extern "C" int jh_model_hist(jh_model* m, int* out /* 24 ints: Newton-iteration histogram (profile builds only) */) {
  JH_REQUIRE(m && out, "model_hist: null pointer");
  JH_HIP(hipMemcpy(out, m->d_stats + 24, 40 * sizeof(int), hipMemcpyDeviceToHost));
  return JH_OK;
}

extern "C" int jh_model_counters(jh_model* m, int* out, int first, int count) {  // diagnostic builds (JH_V5_CENSUS ...): a range of the raw counter block
  JH_REQUIRE(m && out && first >= 0 && count >= 0 && first + count <= JH_NSTATS, "model_counters: bad range");
  JH_HIP(hipMemcpy(out, m->d_stats + first, count * sizeof(int), hipMemcpyDeviceToHost));
  return JH_OK;
}

static jh_xcheck_launchers g_xcheck = {nullptr, nullptr, nullptr};

extern "C" int jh_register_xcheck(const jh_xcheck_launchers* launchers) {
  JH_REQUIRE(launchers && launchers->rollout_cost && launchers->rollout_materialize && launchers->max_knots, "register_xcheck: incomplete launcher table");
  g_xcheck = *launchers;
  return JH_OK;
}

static bool articulated(const jh_model* m) { return m->kind == JH_TASK_LEAP_CUBE || m->kind == JH_TASK_FR3_PICK; }

extern "C" int jh_model_set_kernel(jh_model* m, int generation) {
  JH_REQUIRE(m && generation >= 1 && generation <= 3, "model_set_kernel: generation must be 1, 2 or 3");
  if (generation != 3 && articulated(m) && !g_xcheck.rollout_cost) {
    jh_set_error("model_set_kernel: generations 1 and 2 are cross-check kernels of the test build (libjudo_amd_xcheck.so), not part of this library");
    return JH_ERR_UNSUPPORTED;
  }
  m->kernel_gen = generation;
  return JH_OK;
}

// trace sensors the fused kernel can write per rollout and step: first sensor address, number of floats (0: none -- the elites are re-rolled in materialise mode instead)
static void trace_layout(const jh_model* m, int* adr, int* nfl, int* colmajor) {
  *adr = 0; *nfl = 0; *colmajor = 0;
  if (m->kind == JH_TASK_CARTPOLE || m->kind == JH_TASK_CYLINDER_PUSH) { *nfl = 6; *colmajor = 1; return; }  // both models' six sensors are their two trace sites; one lane per rollout: column-major
  if (m->kernel_gen != 3) return;
  if (m->kind == JH_TASK_LEAP_CUBE && m->ns == 31) { *adr = 16; *nfl = 15; }   // trace_cube, trace_{if,mf,rf,th}_tip (leap_cube.xml: the five framepos sensors)
  else if (m->kind == JH_TASK_FR3_PICK) { *adr = 8; *nfl = 6; }               // trace_object, trace_grasp_site
}

extern "C" int jh_model_trace_layout(const jh_model* m, int* out) {
  JH_REQUIRE(m && out, "model_trace_layout: null pointer");
  trace_layout(m, out, out + 1, out + 2);
  return JH_OK;
}

extern "C" int jh_model_set_contact_capacity(jh_model* m, int contacts) {
  JH_REQUIRE(m != nullptr, "model_set_contact_capacity: null pointer");
  JH_REQUIRE(m->kind == JH_TASK_LEAP_CUBE && (contacts == 48 || contacts == 64), "model_set_contact_capacity: the leap_cube kernel is built for 48 and for 64 contacts per rollout (got %d)", contacts);
  m->contact_capacity = contacts;
  return JH_OK;
}

extern "C" int jh_model_set_self_collision(jh_model* m, int on) {
  JH_REQUIRE(m != nullptr, "model_set_self_collision: null pointer");
  m->self_collision = on ? 1 : 0;
  return JH_OK;
}

int jh_latency_shift(int N, int rpw) {
  static int cus = 0;
  if (cus == 0) { int dev = 0, v = 0; cus = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256; }
  int most = 0; while ((2 << most) <= rpw) most++;
  const char* e = getenv("JUDO_AMD_LATENCY_SHIFT");  // diagnostic override: 0 = every row its own rollout, always
  if (e && e[0] >= '0' && e[0] <= '2') return (e[0] - '0') < most ? (e[0] - '0') : most;
  const long alone = (long)cus * 4;  // waves that each have a SIMD to themselves: a second wave on a SIMD takes from the first what the copies would gain
  for (int sh = most; sh > 0; sh--) if (((long)N << sh) <= alone * rpw) return sh;
  return 0;
}

static int max_fused_knots(const jh_model* m, int H) {
  const bool coop = m->kernel_gen >= 2 && (m->kind == JH_TASK_LEAP_CUBE || m->kind == JH_TASK_FR3_PICK);
  int k = JH_MAX_KNOT_DIM / (m->nu > 0 ? m->nu : 1);
  if (coop && m->kind == JH_TASK_LEAP_CUBE && m->kernel_gen >= 3) return k;  // generation 3 reads its knots from memory every step: no on-chip staging, no limit of its own
  if (coop) return k < 8 ? k : 8;
  // the one-lane kernels stage W (H x K) and 64 lanes' knots in LDS: the launcher's 64 KiB budget bounds K as well
  const int lds_k = (m->kind == JH_TASK_CARTPOLE || m->kind == JH_TASK_CYLINDER_PUSH) ? jh_simple_max_knots(m, H) : (g_xcheck.max_knots ? g_xcheck.max_knots(m, H) : 0);
  return k < lds_k ? k : lds_k;
}

extern "C" int jh_model_limits(const jh_model* m, int* out) {
  JH_REQUIRE(m && out, "model_limits: null pointer");
  out[0] = max_fused_knots(m, 1);  // upper bound over all horizons; jh_model_max_fused_knots(m, H) is the figure for a given H
  out[1] = JH_MAX_KNOT_DIM;
  out[2] = JH_MAX_ELITES;
  out[3] = m->kind == JH_TASK_LEAP_CUBE ? (m->kernel_gen >= 3 ? m->contact_capacity : 32) : (m->kind == JH_TASK_FR3_PICK ? (m->kernel_gen >= 3 ? 96 : 32) : 0);  // (generation 3: leap 48, all in LDS, or 64 with 16 in a row of global memory: jh_model_set_contact_capacity; fr3 32 in LDS + 64 in such a row, next to its 96 pad-against-pad slots)
  return JH_OK;
}

extern "C" int jh_model_max_fused_knots(const jh_model* m, int H) {
  JH_REQUIRE(m != nullptr && H >= 1, "model_max_fused_knots: null model or H < 1");
  return max_fused_knots(m, H);
}

extern "C" int jh_upload_async(void* dst, const void* src, size_t nbytes, void* stream) {
  JH_REQUIRE(dst && src, "upload_async: null pointer");
  JH_HIP(hipMemcpyAsync(dst, src, nbytes, hipMemcpyHostToDevice, (hipStream_t)stream));
  return JH_OK;
}

extern "C" int jh_download_wait(void* dst, const void* src, size_t nbytes, void* stream) {
  JH_REQUIRE(dst && src, "download_wait: null pointer");
  JH_HIP(hipMemcpyAsync(dst, src, nbytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
  JH_HIP(hipStreamSynchronize((hipStream_t)stream));
  return JH_OK;
}

// One completion mark per (host thread, stream): an event belongs to the device it was created on, so a thread that drives controllers on several GPUs
// (one stream each) needs one per stream, created with that stream's device current.  `end` waits for the marks in the order they were set.
namespace {
struct DlMark { void* stream; hipEvent_t ev; };
thread_local std::vector<DlMark> g_dl_marks;     // events owned by this thread, one per stream it has downloaded on
struct DlPending { hipEvent_t ev; const unsigned* flag; unsigned expect; };  // flag non-null: a word in pinned host memory the last workgroup of the update sets (jh_plan_step)
thread_local std::vector<DlPending> g_dl_pending;  // marks set by `begin` and not yet waited for, oldest first
}  // namespace

static int download_begin(void* dst, const void* src, size_t nbytes, void* stream, const unsigned* flag, unsigned expect) {
  JH_REQUIRE(dst && src, "download_begin: null pointer");
  hipEvent_t ev = nullptr;
  for (const DlMark& mk : g_dl_marks) if (mk.stream == stream) ev = mk.ev;
  if (!ev) {
    int cur = 0, dev = 0;
    JH_HIP(hipGetDevice(&cur));
    dev = cur;
    if (stream) { hipDevice_t sd; if (hipStreamGetDevice((hipStream_t)stream, &sd) == hipSuccess) dev = (int)sd; else (void)hipGetLastError(); }
    if (dev != cur) JH_HIP(hipSetDevice(dev));
    hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (dev != cur) (void)hipSetDevice(cur);
    JH_HIP(e);
    g_dl_marks.push_back({stream, ev});
  }
  if (nbytes > 0) JH_HIP(hipMemcpyAsync(dst, src, nbytes, hipMemcpyDeviceToHost, (hipStream_t)stream));  // (0 bytes: the kernels wrote the pinned host block themselves; the mark alone)
  JH_HIP(hipEventRecord(ev, (hipStream_t)stream));
  g_dl_pending.push_back({ev, flag, expect});
  return JH_OK;
}

extern "C" int jh_download_begin(void* dst, const void* src, size_t nbytes, void* stream) { return download_begin(dst, src, nbytes, stream, nullptr, 0u); }

extern "C" int jh_download_end(void) {
  JH_REQUIRE(!g_dl_pending.empty(), "download_end without download_begin");
  const DlPending pd = g_dl_pending.front();
  g_dl_pending.erase(g_dl_pending.begin());
  if (pd.flag) {
    // The update's last workgroup stored `expect` behind its results (system-scope release): the host sees them some microseconds before the stream's event -- the kernel's
    // end-of-launch write-back, the marker packet and its signal -- would report.  The event is looked at every few thousand polls so that a launch that died cannot hang the host.
    for (unsigned spin = 1;; spin++) {
      if (__atomic_load_n(pd.flag, __ATOMIC_ACQUIRE) == pd.expect) return JH_OK;
      __builtin_ia32_pause();
      if ((spin & 0x3FFFu) == 0u) {
        const hipError_t q = hipEventQuery(pd.ev);
        if (q == hipSuccess) { if (__atomic_load_n(pd.flag, __ATOMIC_ACQUIRE) != pd.expect) { jh_set_error("download_end: the launch finished without setting its completion flag"); return JH_ERR_HIP; } return JH_OK; }
        if (q != hipErrorNotReady) JH_HIP(q);
      }
    }
  }
  JH_HIP(hipEventSynchronize(pd.ev));
  return JH_OK;
}

extern "C" int jh_model_profile(jh_model* m, long long* out /* 10 phase cycle totals; zero unless the kernel was built with its phase clock (JH_V6_PHASES) */) {
  JH_REQUIRE(m && out, "model_profile: null pointer");
  JH_HIP(hipMemcpy(out, m->d_stats + 4, 10 * sizeof(long long), hipMemcpyDeviceToHost));
  return JH_OK;
}

extern "C" int jh_rollout_cost(const jh_model* m, const float* x0, const float* nominal, const float* noise, int ldn, const float* sigma,
                               const float* W, const float* lohi, const float* tp, int phase, int N, int n_offset, int H, int K, float* costs,
                               float* knots_out, void* stream) {
  return jh_rollout_cost_traced(m, x0, nominal, noise, ldn, sigma, W, lohi, tp, phase, N, n_offset, H, K, costs, knots_out, nullptr, stream);
}

extern "C" int jh_rollout_cost_traced(const jh_model* m, const float* x0, const float* nominal, const float* noise, int ldn, const float* sigma,
                                      const float* W, const float* lohi, const float* tp, int phase, int N, int n_offset, int H, int K, float* costs,
                                      float* knots_out, float* trace, void* stream) {
  JH_REQUIRE(m && x0 && nominal && noise && sigma && W && lohi && tp && costs, "rollout_cost: null pointer");
  if (trace) { int adr, nfl, cm; trace_layout(m, &adr, &nfl, &cm); JH_REQUIRE(nfl > 0, "rollout_cost_traced: this model's fused kernel writes no trace sensors (jh_model_trace_layout)"); }
  JH_REQUIRE(N > 0 && H > 0 && K >= 1, "rollout_cost: N, H, K must be positive (N=%d H=%d K=%d)", N, H, K);
  JH_REQUIRE(ldn >= N, "rollout_cost: ldn (%d) < N (%d)", ldn, N);
  JH_REQUIRE(K * m->nu <= JH_MAX_KNOT_DIM, "rollout_cost: K*nu = %d exceeds %d", K * m->nu, JH_MAX_KNOT_DIM);
  JH_REQUIRE(n_offset >= 0, "rollout_cost: negative n_offset");
  hipStream_t st = (hipStream_t)stream;
  if (m->kind == JH_TASK_CARTPOLE || m->kind == JH_TASK_CYLINDER_PUSH)
    return jh_simple_rollout_cost(m, x0, nominal, noise, ldn, sigma, W, lohi, tp, N, n_offset, H, K, costs, knots_out, trace, st);
  if (m->kind == JH_TASK_LEAP_CUBE && m->kernel_gen == 3)
    return m->contact_capacity > 48 ? jh_engine5_rollout_cost_cap64(m, x0, nominal, noise, ldn, sigma, W, lohi, tp, N, n_offset, H, K, costs, knots_out, trace, st)
                                    : jh_engine5_rollout_cost(m, x0, nominal, noise, ldn, sigma, W, lohi, tp, N, n_offset, H, K, costs, knots_out, trace, st);
  if (m->kind == JH_TASK_FR3_PICK && m->kernel_gen == 3) return jh_engine6_rollout_cost(m, x0, nominal, noise, ldn, sigma, W, lohi, tp, phase, N, n_offset, H, K, costs, knots_out, trace, st);
  JH_REQUIRE(trace == nullptr, "rollout_cost_traced: only the product kernels (generation 3, cartpole, cylinder_push) write trace sensors");
  if (!g_xcheck.rollout_cost) { jh_set_error("rollout_cost: no kernel for this model / generation in this library"); return JH_ERR_UNSUPPORTED; }
  return g_xcheck.rollout_cost(m, m->kernel_gen, x0, nominal, noise, ldn, sigma, W, lohi, tp, phase, N, n_offset, H, K, costs, knots_out, stream);
}

// Closed-form models (cartpole, cylinder_push): the rollout kernels are ~50 us, so a second launch and the gap in front of it are a fifth of the plan step -- the two run as
// one launch (jh_simple.hip).  JUDO_AMD_PLAN_STEP_LAUNCHES=2 keeps the two launches (A/B, tests/test_gpu_simple.py compares the two forms bit for bit).
static bool one_launch_plan_step(const jh_model* m, int N, int H, int K, const float* costs, const float* knots_out, const float* W, const float* noise, int ldn) {
  static const bool two = [] { const char* e = getenv("JUDO_AMD_PLAN_STEP_LAUNCHES"); return e && e[0] == '2'; }();
  if (two || !(m->kind == JH_TASK_CARTPOLE || m->kind == JH_TASK_CYLINDER_PUSH) || knots_out || !costs || !W || !noise) return false;
  return N > 0 && H > 0 && K >= 1 && ldn >= N && K * m->nu <= JH_MAX_KNOT_DIM && jh_simple_plan_step_fits(m, H, K);
}

// One plan-step iteration on one GPU as ONE call (Controller.update_action's loop body, judo/controller/controller.py:250-299): the packed host block
// x0 | nominal | sigma | task params | bounds goes up, the fused rollout + cost kernel runs, jh_update_fused reduces the costs to nominal | sigma | trace
// records -- written where `out` points, normally the pinned host block itself -- and the completion mark of jh_download_begin is set: jh_download_end waits for
// it.  Everything is enqueued on `stream`; nothing is waited for here.  Five ctypes calls less per plan step than the separate entry points: a tenth of a small one.
extern "C" int jh_plan_step(const jh_model* m, void* blk_dev, const void* blk_host, size_t blk_bytes, int o_nominal, int o_sigma, int o_tp, int o_lohi, const float* noise, int ldn,
                            const float* W, int phase, int N, int n_offset, int H, int K, float* costs, float* knots_out, float* trace, int mode, float lambda, int k, int tie_high, int E,
                            int row_floats, int colmajor, float* scratch, float* out, void* out_host_mark, void* const* timing /* 3 events of jh_event_create, or NULL */, void* stream) {
  JH_REQUIRE(m && blk_dev && blk_host && out && scratch, "plan_step: null pointer");
  const float* b = (const float*)blk_dev;
  hipStream_t st = (hipStream_t)stream;
  // blk_dev == blk_host: a device-visible pinned host block the kernels read in place (the closed-form models: a few hundred bytes read once per workgroup cost less than the copy in front of the launch)
  int rc = blk_dev == blk_host ? JH_OK : jh_upload_async(blk_dev, blk_host, blk_bytes, stream);
  if (rc == JH_OK && timing) JH_HIP(hipEventRecord((hipEvent_t)timing[0], st));
  const int KU = K * m->nu;
  // out_host_mark != out: a 4-byte word in device-visible pinned host memory -- the update's last workgroup stores its old value + 1 there behind the results and jh_download_end
  // polls it instead of waiting for the stream's event
  unsigned* flag = (out_host_mark && out_host_mark != (void*)out) ? (unsigned*)out_host_mark : nullptr;
  const unsigned expect = flag ? __atomic_load_n(flag, __ATOMIC_RELAXED) + 1u : 0u;
  if (rc == JH_OK && one_launch_plan_step(m, N, H, K, costs, knots_out, W, noise, ldn)) {
    // closed-form models: rollout + cost + the update's tail in ONE launch (jh_simple.hip k_plan_step); the rollout / update split of the timing events collapses
    jh_upd::TailArgs a;
    rc = jh_update_tail_args("plan_step", costs, nullptr, b + o_nominal, noise, ldn, b + o_sigma, b + o_lohi, N, n_offset, K, m->nu, mode, lambda, k, tie_high, trace ? E : 0, trace, row_floats,
                             colmajor, scratch, out, out + KU, (trace && E > 0) ? out + 2 * KU : nullptr, nullptr, &a);
    a.done_flag = flag; a.done_value = expect;
    if (rc == JH_OK) rc = jh_simple_plan_step(m, b, W, b + o_tp, H, K, a, st);
    if (rc == JH_OK && timing) { JH_HIP(hipEventRecord((hipEvent_t)timing[1], st)); JH_HIP(hipEventRecord((hipEvent_t)timing[2], st)); }
    if (rc == JH_OK) rc = download_begin(out, out, 0, stream, flag, expect);
    return rc;
  }
  if (rc == JH_OK) rc = jh_rollout_cost_traced(m, b, b + o_nominal, noise, ldn, b + o_sigma, W, b + o_lohi, b + o_tp, phase, N, n_offset, H, K, costs, knots_out, trace, stream);
  if (rc == JH_OK && timing) JH_HIP(hipEventRecord((hipEvent_t)timing[1], st));
  if (rc == JH_OK) {
    jh_upd::TailArgs a;
    rc = jh_update_tail_args("plan_step", costs, nullptr, b + o_nominal, noise, ldn, b + o_sigma, b + o_lohi, N, n_offset, K, m->nu, mode, lambda, k, tie_high, trace ? E : 0, trace, row_floats,
                             colmajor, scratch, out, out + KU, (trace && E > 0) ? out + 2 * KU : nullptr, nullptr, &a);
    a.done_flag = flag; a.done_value = expect;
    if (rc == JH_OK) rc = jh_update_tail_launch(a, st);
  }
  if (rc == JH_OK && timing) JH_HIP(hipEventRecord((hipEvent_t)timing[2], st));
  if (rc == JH_OK) rc = download_begin(out, out, 0, stream, flag, expect);
  return rc;
}

// The same iteration when the rollouts are sharded over G ranks: launch -> all-gather -> merge.  jh_plan_step_shard is jh_plan_step with the update's
// last stage left out: the tail launch writes this rank's RECORD (jh_update_shard) instead of the nominal; the caller all-gathers the G records (RCCL: one collective
// of <= a few KB) and hands them to jh_plan_merge, which finishes the update on every rank (jh_shard_merge: identical nominal everywhere, no broadcast) into the same
// output block jh_plan_step fills and sets the same completion mark.
extern "C" int jh_plan_step_shard(const jh_model* m, void* blk_dev, const void* blk_host, size_t blk_bytes, int o_nominal, int o_sigma, int o_tp, int o_lohi, const float* noise,
                                  int ldn, const float* W, int phase, int N, int n_offset, int H, int K, float* costs, float* knots_out, float* trace, int mode, float lambda, int k,
                                  int tie_high, int E, int row_floats, int colmajor, float* scratch, float* rec_out, void* const* timing /* 3 events, or NULL */, void* stream) {
  JH_REQUIRE(m && blk_dev && blk_host && rec_out && scratch, "plan_step_shard: null pointer");
  const float* b = (const float*)blk_dev;
  hipStream_t st = (hipStream_t)stream;
  int rc = blk_dev == blk_host ? JH_OK : jh_upload_async(blk_dev, blk_host, blk_bytes, stream);
  if (rc == JH_OK && timing) JH_HIP(hipEventRecord((hipEvent_t)timing[0], st));
  if (rc == JH_OK && one_launch_plan_step(m, N, H, K, costs, knots_out, W, noise, ldn)) {
    jh_upd::TailArgs a;
    rc = jh_update_tail_args("plan_step_shard", costs, nullptr, b + o_nominal, noise, ldn, b + o_sigma, b + o_lohi, N, n_offset, K, m->nu, mode, lambda, k, tie_high, trace ? E : 0, trace,
                             row_floats, colmajor, scratch, nullptr, nullptr, nullptr, rec_out, &a);
    if (rc == JH_OK) rc = jh_simple_plan_step(m, b, W, b + o_tp, H, K, a, st);
    if (rc == JH_OK && timing) { JH_HIP(hipEventRecord((hipEvent_t)timing[1], st)); JH_HIP(hipEventRecord((hipEvent_t)timing[2], st)); }
    return rc;
  }
  if (rc == JH_OK) rc = jh_rollout_cost_traced(m, b, b + o_nominal, noise, ldn, b + o_sigma, W, b + o_lohi, b + o_tp, phase, N, n_offset, H, K, costs, knots_out, trace, stream);
  if (rc == JH_OK && timing) JH_HIP(hipEventRecord((hipEvent_t)timing[1], st));
  if (rc == JH_OK) rc = jh_update_shard(costs, nullptr, b + o_nominal, noise, ldn, b + o_sigma, b + o_lohi, N, n_offset, K, m->nu, mode, lambda, k, tie_high, trace ? E : 0, trace, row_floats,
                                        colmajor, scratch, rec_out, stream);
  if (rc == JH_OK && timing) JH_HIP(hipEventRecord((hipEvent_t)timing[2], st));
  return rc;
}

extern "C" int jh_plan_merge(const float* recs, int G, int K, int nu, int mode, float lambda, int k, int tie_high, int E, int row_floats, float* out, void* out_host_mark,
                             void* timing_done /* event of jh_event_create recorded behind the merge, or NULL */, void* stream) {
  JH_REQUIRE(recs && out, "plan_merge: null pointer");
  const int KU = K * nu;
  int rc = jh_shard_merge(recs, G, K, nu, mode, lambda, k, tie_high, E, row_floats, out, out + KU, E > 0 ? out + 2 * KU : nullptr, stream);
  if (rc == JH_OK && timing_done) JH_HIP(hipEventRecord((hipEvent_t)timing_done, (hipStream_t)stream));
  if (rc == JH_OK) rc = jh_download_begin(out_host_mark, out, 0, stream);
  return rc;
}

// Timing events for callers that bracket kernels on the launch stream without torch (bench.py's roofline leg: the rollout kernel's duration inside a jh_plan_step call)
extern "C" int jh_event_create(void** out) { JH_REQUIRE(out, "event_create: null pointer"); hipEvent_t e; JH_HIP(hipEventCreate(&e)); *out = e; return JH_OK; }
extern "C" void jh_event_destroy(void* ev) { if (ev) (void)hipEventDestroy((hipEvent_t)ev); }
extern "C" int jh_event_record(void* ev, void* stream) { JH_REQUIRE(ev, "event_record: null pointer"); JH_HIP(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream)); return JH_OK; }
extern "C" int jh_stream_wait_event(void* stream, void* ev) {  // device-side: work enqueued on `stream` after this call waits for `ev` (recorded on another stream)
  JH_REQUIRE(ev, "stream_wait_event: null pointer");
  JH_HIP(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)ev, 0));
  return JH_OK;
}
extern "C" int jh_event_elapsed_ms(void* a, void* b, float* ms) {  // waits for b
  JH_REQUIRE(a && b && ms, "event_elapsed_ms: null pointer");
  JH_HIP(hipEventSynchronize((hipEvent_t)b));
  JH_HIP(hipEventElapsedTime(ms, (hipEvent_t)a, (hipEvent_t)b));
  return JH_OK;
}

extern "C" int jh_rollout_materialize(const jh_model* m, const float* x0, int x0_batched, const float* controls, int N, int H, float* states,
                                      float* sensors, void* stream) {
  JH_REQUIRE(m && x0 && controls, "rollout_materialize: null pointer");
  JH_REQUIRE(states || sensors, "rollout_materialize: both outputs are null");
  JH_REQUIRE(N > 0 && H > 0, "rollout_materialize: N and H must be positive (N=%d H=%d)", N, H);
  hipStream_t st = (hipStream_t)stream;
  if (m->kind == JH_TASK_CARTPOLE || m->kind == JH_TASK_CYLINDER_PUSH) return jh_simple_materialize(m, x0, x0_batched, controls, N, H, states, sensors, st);
  if (m->kind == JH_TASK_LEAP_CUBE && m->kernel_gen == 3)
    return m->contact_capacity > 48 ? jh_engine5_materialize_cap64(m, x0, x0_batched, controls, N, H, states, sensors, st) : jh_engine5_materialize(m, x0, x0_batched, controls, N, H, states, sensors, st);
  if (m->kind == JH_TASK_FR3_PICK && m->kernel_gen == 3) return jh_engine6_materialize(m, x0, x0_batched, controls, N, H, states, sensors, st);
  if (!g_xcheck.rollout_materialize) { jh_set_error("rollout_materialize: no kernel for this model / generation in this library"); return JH_ERR_UNSUPPORTED; }
  return g_xcheck.rollout_materialize(m, m->kernel_gen, x0, x0_batched, controls, N, H, states, sensors, stream);
}

extern "C" int jh_task_reward(const jh_model* m, const float* states, const float* sensors, const float* controls, const float* tp, int phase, int N,
                              int H, float* rewards, void* stream) {
  JH_REQUIRE(m && states && tp && rewards, "task_reward: null pointer");
  JH_REQUIRE(N > 0 && H > 0, "task_reward: N and H must be positive");
  hipStream_t st = (hipStream_t)stream;
  if (m->kind == JH_TASK_CARTPOLE || m->kind == JH_TASK_CYLINDER_PUSH) {
    JH_REQUIRE(m->kind != JH_TASK_CARTPOLE || controls, "task_reward: cartpole needs controls");
    return jh_simple_reward(m, states, controls, tp, N, H, rewards, st);
  }
  return jh_engine_reward(m, states, sensors, controls, tp, phase, N, H, rewards, st);
}
