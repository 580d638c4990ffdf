// sr_api.hip -- host side of libspectrobot_hip.so: the C ABI of
// include/spectrobot_hip.h on top of the kernels in sr_kernels.hip.
//
// Host work is limited to what the reference also does once per call in plain
// Python: line filtering / window centres (spect_classes.py:1384-1388, 1937-1943),
// per-layer scalars and level populations (spect_main_module.py:2049-2073),
// TIPS-2003 lookup + 4-point Lagrange (spect_classes.py:1680-1710).  Every
// per-(line, layer) and per-grid-point quantity is computed on the GPU; there is
// no CPU fallback for any of it.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/spectrobot_hip.h"
#include "sr_kernels.hpp"
#include "tips2003_tables.inc"

using namespace sr;

namespace {

thread_local std::string g_err;

int hip_fail(hipError_t e, const char *what) {
  g_err = std::string(what) + ": " + hipGetErrorString(e);
  return SR_ERR_HIP;
}
#define HIPCHK(expr)                                  \
  do {                                                \
    hipError_t e__ = (expr);                          \
    if (e__ != hipSuccess) return hip_fail(e__, #expr); \
  } while (0)
#define LAUNCHCHK(expr)                                              \
  do {                                                               \
    int e__ = (expr);                                                \
    if (e__ != 0) return hip_fail((hipError_t)e__, "kernel launch"); \
  } while (0)

// grow-only device buffer
struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  unsigned gen = 0; // bumped whenever the block is (re)allocated or released: what a cache of the CONTENTS keys on (a new
                    // block may come back at the old address)
  int ensure(size_t bytes) {
    if (bytes <= cap) return SR_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    ++gen;
    size_t want = bytes + bytes / 8 + 256;
    HIPCHK(hipMalloc(&p, want));
    cap = want;
    return SR_OK;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    ++gen;
  }
  template <class T> T *as() const { return static_cast<T *>(p); }
};

// Pinned host staging buffer with a device mirror: fill host(), then push() on
// the caller's stream; the next prepare() waits for the previous copy only.
struct Stager {
  void *h = nullptr;
  size_t hcap = 0;
  DevBuf d;
  hipEvent_t done = nullptr;
  bool pending = false;
  int dev = -1; // device the mirror, the event and the pinned buffer belong to
  int prepare(size_t bytes) {
    // the thread-local rings outlive a hipSetDevice(): a slot made on another device is rebuilt on this one
    int cur = 0;
    HIPCHK(hipGetDevice(&cur));
    if (dev != cur) {
      if (dev >= 0) {
        HIPCHK(hipSetDevice(dev));
        release();
        HIPCHK(hipSetDevice(cur));
      }
      dev = cur;
    }
    if (pending) {
      HIPCHK(hipEventSynchronize(done));
      pending = false;
    }
    if (!done) HIPCHK(hipEventCreateWithFlags(&done, hipEventDisableTiming));
    if (bytes > hcap) {
      if (h) (void)hipHostFree(h);
      h = nullptr;
      hcap = 0;
      const size_t want = bytes + bytes / 8 + 256;
      HIPCHK(hipHostMalloc(&h, want, hipHostMallocDefault));
      hcap = want;
    }
    return d.ensure(std::max<size_t>(bytes, 16));
  }
  int push(size_t bytes, hipStream_t st) {
    if (bytes) HIPCHK(hipMemcpyAsync(d.p, h, bytes, hipMemcpyHostToDevice, st));
    HIPCHK(hipEventRecord(done, st));
    pending = true;
    return SR_OK;
  }
  // The copy on a dedicated copy stream, `st` only waits for it.  Host-to-device copies of all streams
  // share the DMA queues in issue order: a small staging copy enqueued on `st` BEHIND 8 ms of kernels
  // held back the next call's layer-scalar copy (on the prep stream) and with it the pipelined
  // preparation of the next call's tables (+0.9 ms per step).  prepare() has already waited for the
  // slot's last consumers, so the copy may run at once.
  int push_early(size_t bytes, hipStream_t st) {
    static thread_local std::map<int, hipStream_t> copy_streams; // one per device this thread has used
    hipStream_t &copy_st = copy_streams[dev];
    if (!copy_st) HIPCHK(hipStreamCreateWithFlags(&copy_st, hipStreamNonBlocking));
    if (bytes) HIPCHK(hipMemcpyAsync(d.p, h, bytes, hipMemcpyHostToDevice, copy_st));
    HIPCHK(hipEventRecord(done, copy_st));
    HIPCHK(hipStreamWaitEvent(st, done, 0));
    pending = true;
    return SR_OK;
  }
  // push_early in two halves, for work that depends on the staged data alone (the LOS column integration): it runs
  // on the copy stream right behind the copy, i.e. as soon as the host has issued it -- on `st` it queued behind
  // everything the caller had submitted before (on a 1/8 shard: copy, columns kernel and two event hand-overs, ~35 us
  // of a 0.9 ms step, after the coefficient kernels instead of beside them).
  int begin_early(size_t bytes, hipStream_t *copy_stream_out) {
    static thread_local std::map<int, hipStream_t> work_streams; // one per device this thread has used
    hipStream_t &cs = work_streams[dev];
    if (!cs) HIPCHK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
    if (bytes) HIPCHK(hipMemcpyAsync(d.p, h, bytes, hipMemcpyHostToDevice, cs));
    *copy_stream_out = cs;
    return SR_OK;
  }
  int end_early(hipStream_t copy_stream, hipStream_t st) {
    HIPCHK(hipEventRecord(done, copy_stream));
    HIPCHK(hipStreamWaitEvent(st, done, 0));
    pending = true;
    return SR_OK;
  }
  // re-record `done` behind the kernels that read (or write) the device mirror, so that the slot is
  // not refilled while they run
  int mark(hipStream_t st) {
    HIPCHK(hipEventRecord(done, st));
    pending = true;
    return SR_OK;
  }
  void release() {
    if (pending && done) (void)hipEventSynchronize(done);
    pending = false;
    if (h) (void)hipHostFree(h);
    h = nullptr;
    hcap = 0;
    d.release();
    if (done) (void)hipEventDestroy(done);
    done = nullptr;
  }
  template <class T> T *host() const { return static_cast<T *>(h); }
};

// ---- TIPS-2003 (fparts_mod.f:33-295) + CalcPartitionSum (spect_classes.py:1692-1710) ----
int tips_row(int mol, int iso) {
  for (int i = 0; i < kTipsNTab; ++i)
    if (kTipsKey[i][0] == mol && kTipsKey[i][1] == iso) return i;
  return -1;
}

// scipy.interpolate.lagrange through the two table points <= T and the two > T:
// polynomial assembled in coefficient form by successive products with
// (x - x_k)/(x_j - x_k), summed, then evaluated by Horner -- the order of
// operations of scipy's poly1d arithmetic, which the result depends on at 1e-13.
double lagrange4(const double *tg, const double *qg, int n, double temp) {
  double xs[4], qs[4];
  int m = 0, n_le = 0;
  while (n_le < n && tg[n_le] <= temp) ++n_le;
  for (int i = std::max(0, n_le - 2); i < n_le; ++i, ++m) { xs[m] = tg[i]; qs[m] = qg[i]; }
  for (int i = n_le; i < std::min(n, n_le + 2); ++i, ++m) { xs[m] = tg[i]; qs[m] = qg[i]; }
  std::vector<double> poly(1, 0.0);
  for (int j = 0; j < m; ++j) {
    std::vector<double> pt(1, qs[j]);
    for (int k = 0; k < m; ++k) {
      if (k == j) continue;
      const double fac = xs[j] - xs[k];
      const double f0 = 1.0 / fac, f1 = -xs[k] / fac;
      std::vector<double> nx(pt.size() + 1, 0.0);
      for (size_t i = 0; i < nx.size(); ++i) {
        double s = 0.0;
        if (i < pt.size()) s += pt[i] * f0;
        if (i >= 1) s += pt[i - 1] * f1;
        nx[i] = s;
      }
      pt.swap(nx);
    }
    if (pt.size() > poly.size()) poly.insert(poly.begin(), pt.size() - poly.size(), 0.0);
    const size_t off = poly.size() - pt.size();
    for (size_t i = 0; i < pt.size(); ++i) poly[off + i] += pt[i];
  }
  double y = 0.0;
  for (double c : poly) y = y * temp + c;
  return y;
}

// Host copy of a (filtered, centre-sorted) line list in the layout of the device SoA: 12 double
// arrays then 3 int arrays of length md.
struct HostLines {
  size_t md = 1;   // array stride (>= 1)
  int64_t m = 0;   // lines
  std::vector<double> d;
  std::vector<int> i;
};

// Process-wide mode switches (sr_set_*).  Atomic: a call reads each ONCE at entry and works with that
// snapshot, so flipping a switch from another thread never changes a call half way.
std::atomic<int> g_variant{8};   // points per lane in the exact wings kernel
std::atomic<int> g_far_field{3}; // 1: far wings by per-line local expansions, 2: by box pairs (multipole -> local), 3 (default): 2, but sparse line sets by 1, 0: every evaluation exact
std::atomic<int> g_overlap{1};   // 1 (default): the decoupled, phased pipeline; 0: kernels one after the other
// sr_set_jac_layer_mode; SR_JAC_LAYER_MODE (environment, read once at load): its initial value, for A/B runs of whole programs
std::atomic<int> g_band_fusion{1}; // sr_set_band_fusion
std::atomic<int> g_jac_layer_forward{[] { const char *e = getenv("SR_JAC_LAYER_MODE"); const int v = e ? atoi(e) : 0; return v >= 0 && v <= 3 ? v : 0; }()};
// sr_set_kernel_repeat: a measurement hook of the SERIAL schedule -- kernel g_repeat_kernel of every call is launched
// g_repeat_n times back to back (energy per launch: tools/energy_by_kernel.py loops one kernel for seconds beside a power
// sampler).  0 prep, 1 level-0 far-field pass, 2 S2M + M2M, 3 M2L, 4 zones, 5 wings.  Results of a repeated M2L are NOT
// valid (it adds into the level-0 coefficients); everything else stores.
std::atomic<int> g_repeat_kernel{-1}, g_repeat_n{1};
std::atomic<int> g_level_route{1}; // sr_set_level_route: 1 (default) the multi-channel pass for level tables, 0 one coefficient op per level
std::atomic<int> g_timing{1};    // 0: no timing events in the coefficient op (sr_set_timing: seven hipEventRecord fewer per call)
std::atomic<int> g_counting{0};  // 1: counting instantiations of the far-field-mode kernels (sr_set_counting)
std::atomic<size_t> g_table_budget{(size_t)48 << 30}; // bytes of FastRec + ColdRec tables per layer batch

} // namespace

// Per-call scratch and stream / event state of the coefficient op.  A lineset owns one; the per-level sub-linesets
// (level_set) SHARE their parent's -- calls on a handle and its children are serialised through ev_last_done anyway,
// and twelve levels with their own record tables, far-field scratch, streams and events held twelve times the memory.
struct CoefWork {
  // [2]: with sr_set_overlap(1) the tables of call c+1 are prepared (on prep_st) while the kernels
  // of call c still read theirs
  Stager s_layers[2];
  DevBuf d_fast[2], d_cold[2], d_coef[2], d_zone, d_zone2[2], d_mom[2], d_outer_recs;
  hipStream_t aux = nullptr;     // second stream: zones kernel beside the far-field kernel
  hipStream_t prep_st = nullptr; // third stream: staging copy + sr_prep_kernel of the NEXT call
  hipEvent_t ev_prep_done[2] = {nullptr, nullptr}, ev_tables_free[2] = {nullptr, nullptr}, ev_op0 = nullptr;
  // the far-field chain (level-0 pass, moments, upward pass, translations) of call c + 1 runs on prep_st behind its
  // table preparation, i.e. beside the zones / wings kernels of call c (its own coefficient / moment buffers)
  hipEvent_t ev_far_done[2] = {nullptr, nullptr}, ev_zones_done[2] = {nullptr, nullptr};
  hipStream_t chain_st = nullptr, chain2_st = nullptr; // the far-field chain (decoupled pipeline): level-0 pass | moments, translations
  hipEvent_t ev_l0_done[2] = {nullptr, nullptr}, ev_s2m_done[2] = {nullptr, nullptr};
  bool free_recorded[2] = {false, false};
  int parity = 0;
  bool overlapped = false;       // last call ran that way (timing hook)
  bool pipelined = false;        // last call prepared its tables on prep_st
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  int n_timed = 0; // kernels timed in the last call
  bool timed = false;
  // d_coef / d_zone / d_counts and (without overlap) the single table set are shared by consecutive
  // calls: every call first makes its streams wait for the end of the previous call on this handle,
  // whatever stream that one ran on (ev_last_done), so calls on unrelated caller streams are safe.
  hipEvent_t ev_last_done = nullptr;
  bool last_done_recorded = false;
  DevBuf d_counts; // kCntN executed-work counters of the last counting call
  bool counted = false;
  int init() {
    for (auto &e : ev) HIPCHK(hipEventCreate(&e));
    HIPCHK(hipEventCreateWithFlags(&ev_last_done, hipEventDisableTiming));
    return SR_OK;
  }
  void release() {
    for (int b = 0; b < 2; ++b) {
      s_layers[b].release();
      d_fast[b].release();
      d_cold[b].release();
      if (ev_prep_done[b]) (void)hipEventDestroy(ev_prep_done[b]);
      if (ev_tables_free[b]) (void)hipEventDestroy(ev_tables_free[b]);
      if (ev_far_done[b]) (void)hipEventDestroy(ev_far_done[b]);
      if (ev_zones_done[b]) (void)hipEventDestroy(ev_zones_done[b]);
      if (ev_l0_done[b]) (void)hipEventDestroy(ev_l0_done[b]);
      if (ev_s2m_done[b]) (void)hipEventDestroy(ev_s2m_done[b]);
      d_zone2[b].release();
      d_coef[b].release();
      d_mom[b].release();
    }
    if (ev_op0) (void)hipEventDestroy(ev_op0);
    if (ev_last_done) (void)hipEventDestroy(ev_last_done);
    d_counts.release();
    d_outer_recs.release();
    if (prep_st) (void)hipStreamDestroy(prep_st);
    if (chain_st) (void)hipStreamDestroy(chain_st);
    if (chain2_st) (void)hipStreamDestroy(chain2_st);
    d_zone.release();
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    if (ev_join) (void)hipEventDestroy(ev_join);
    if (aux) (void)hipStreamDestroy(aux);
    for (auto &e : ev)
      if (e) (void)hipEventDestroy(e);
  }
};

// Scratch of the multi-channel pass (mc_pass): record tables of the FULL list with the three channel weights, the
// far-only coefficients of the level passes, its zones stream, and the CoefWorks the far-only passes run through (not
// the handle's shared one: folded ops on the handle and the passes of one table build do not wait for each other's scratch).
constexpr int kMcFarLanes = 4; // far-only level passes in flight side by side (each on its own CoefWork: tables, streams, events)
struct McWork {
  CoefWork fw[kMcFarLanes];
  bool fw_init = false, batch_pending = false;
  Stager s_layers, s_far, s_batch;
  DevBuf d_fast, d_cold, d_coef, d_outer_recs, d_bfast;
  hipStream_t zst = nullptr;
  hipEvent_t ev_prep = nullptr, ev_zones = nullptr;
  hipEvent_t ev_t[5] = {nullptr, nullptr, nullptr, nullptr, nullptr}; // timing: start, after the tables, after the zones kernel, before / after the wings kernel
  bool timed = false;
  void release() {
    if (fw_init)
      for (auto &f : fw) f.release();
    fw_init = false;
    s_layers.release();
    s_far.release();
    s_batch.release();
    d_fast.release();
    d_cold.release();
    d_coef.release();
    d_outer_recs.release();
    d_bfast.release();
    if (zst) (void)hipStreamDestroy(zst);
    if (ev_prep) (void)hipEventDestroy(ev_prep);
    if (ev_zones) (void)hipEventDestroy(ev_zones);
    for (auto &e : ev_t)
      if (e) { (void)hipEventDestroy(e); e = nullptr; }
    zst = nullptr;
    ev_prep = ev_zones = nullptr;
  }
};

struct sr_lineset {
  int64_t n_lines = 0; // main lines (centre inside its own window)
  HostLines host, host_outer; // host copies: per-level subsets are cut from them (level_set)
  int n_outer = 0;            // lines whose centre lies outside their window (outer branches)
  DevBuf d_lines_outer;
  LinesDev Lo{};
  // per-level sub-linesets (lines whose upper or lower level is L), built on first use by the
  // G-coefficient / tracked-level entry points; nullptr until then
  std::vector<sr_lineset *> level_sets;
  std::vector<sr_lineset *> level_up_sets; // the lines whose UPPER level is L only (the ind_emission pass)
  DevBuf d_gscratch;          // second output channel of the ind_emission pass
  GridParams gp{};
  int mol = 0, iso = 0, n_levels = 0;
  double mm = 0.0;
  std::vector<double> e_lev;
  std::vector<int> ic; // host copy, sorted
  double freq_max = 0.0;
  double gamma_max = 0.0, ndep_min = 0.0, ndep_max = 0.0; // air broadening / its temperature exponent over the lines
  int64_t n_disp_lo = 0, n_disp_hi = 0; // leading / trailing lines centred beyond the grid ends (window clamped to the end point)
  DevBuf d_lines;      // one allocation, carved below
  LinesDev L{};
  DevBuf d_first;
  int first_x0 = 0, first_n = 0; // IcIndex table domain
  CoefWork own_work;
  CoefWork *work = nullptr;      // &own_work, or the parent's for a per-level sub-lineset
  McWork mc;                     // the multi-channel pass of this handle (parents only)
  std::vector<double> bounds_temps; // sr_lineset_set_bounds_temps: empty = boundaries at the call's own temperatures
  sr_lineset *parent = nullptr;     // per-level sub-lineset: the handle it was cut from (its bounds_temps apply)
  bool linear_weights = false;      // sr_lineset_set_linear_weights (takes effect with bounds_temps only)
};

extern "C" {

const char *sr_strerror(int s) {
  switch (s) {
    case SR_OK: return "ok";
    case SR_ERR_ARG: return "bad argument";
    case SR_ERR_LIMIT: return "size limit exceeded";
    case SR_ERR_HIP: return "HIP runtime error";
    case SR_ERR_NODEVICE: return "no gfx950 device";
    case SR_ERR_UNSUPPORTED: return "unsupported input";
    case SR_ERR_TABLE: return "(mol, iso) not in TIPS-2003 tables";
    default: return "unknown status";
  }
}
const char *sr_last_error(void) { return g_err.c_str(); }
int sr_abi_version(void) { return 1; }

int sr_set_device(int device) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    g_err = "no HIP device visible";
    return SR_ERR_NODEVICE;
  }
  if (device < 0 || device >= n) return SR_ERR_ARG;
  HIPCHK(hipSetDevice(device));
  return SR_OK;
}

int sr_device_info(char *name, int name_len, int *cu_count, double *hbm_gib) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return SR_ERR_NODEVICE;
  hipDeviceProp_t p;
  HIPCHK(hipGetDeviceProperties(&p, dev));
  if (name && name_len > 0) snprintf(name, name_len, "%s (%s)", p.name, p.gcnArchName);
  if (cu_count) *cu_count = p.multiProcessorCount;
  if (hbm_gib) *hbm_gib = (double)p.totalGlobalMem / (1024.0 * 1024.0 * 1024.0);
  return SR_OK;
}

int sr_recommended_hw_queues(int *recommended, int *configured) {
  if (recommended) *recommended = 8;
  if (configured) {
    const char *e = getenv("GPU_MAX_HW_QUEUES");
    const int v = e ? atoi(e) : 0;
    *configured = v > 0 ? v : 4; // the runtime's default
  }
  return SR_OK;
}

int sr_set_table_budget(int64_t bytes) {
  if (bytes < (int64_t)(sizeof(FastRec) + sizeof(ColdRec))) return SR_ERR_ARG;
  g_table_budget.store((size_t)bytes);
  return SR_OK;
}

int sr_set_far_field(int on) {
  g_far_field.store(on < 0 ? 0 : (on > 3 ? 3 : on));
  return SR_OK;
}

int sr_set_overlap(int on) {
  g_overlap.store(on != 0 ? 1 : 0); // see spectrobot_hip.h
  return SR_OK;
}

int sr_set_kernel_repeat(int kernel, int n) {
  if (kernel < -1 || kernel > 5 || n < 1) return SR_ERR_ARG;
  g_repeat_kernel.store(kernel);
  g_repeat_n.store(kernel < 0 ? 1 : n);
  return SR_OK;
}

int sr_set_level_route(int multi_channel) {
  g_level_route.store(multi_channel ? 1 : 0);
  return SR_OK;
}

int sr_set_timing(int on) {
  g_timing.store(on == 2 ? 2 : (on ? 1 : 0));
  return SR_OK;
}

int sr_set_counting(int on) {
  g_counting.store(on ? 1 : 0);
  return SR_OK;
}

int sr_set_jac_layer_mode(int forward) {
  g_jac_layer_forward.store(forward == 2 || forward == 3 ? forward : (forward ? 1 : 0)); // see spectrobot_hip.h
  return SR_OK;
}

double sr_far_field_truncation_bound(void) { return 18.0 * std::pow((double)kTheta, -(double)(kFD + 1)); }

int sr_set_band_fusion(int on) {
  g_band_fusion.store(on ? 1 : 0);
  return SR_OK;
}

int sr_set_points_per_lane(int p) {
  if (p != 4 && p != 8) return SR_ERR_ARG;
  g_variant.store(p);
  return SR_OK;
}

// ------------------------------------------------------------------------
int sr_bd_tips_2003(int mol, int iso, double *gi, double *t_grid119, double *qt_grid119) {
  const int r = tips_row(mol, iso);
  if (r < 0) return SR_ERR_TABLE;
  if (gi) *gi = kTipsGi[r];
  if (t_grid119) std::memcpy(t_grid119, kTipsT, sizeof(double) * kTipsNT);
  if (qt_grid119) std::memcpy(qt_grid119, kTipsQ[r], sizeof(double) * kTipsNT);
  return SR_OK;
}

int sr_calc_partition_sum(int mol, int iso, const double *temps, int n, double *q_out) {
  if (!temps || !q_out || n < 0) return SR_ERR_ARG;
  const int r = tips_row(mol, iso);
  if (r < 0) return SR_ERR_TABLE;
  for (int i = 0; i < n; ++i) q_out[i] = lagrange4(kTipsT, kTipsQ[r], kTipsNT, temps[i]);
  return SR_OK;
}

} // extern "C"

// ------------------------------------------------------------------------
namespace {

int upload_soa(const HostLines &H, DevBuf &buf, LinesDev &L) {
  const size_t bytes_d = H.d.size() * sizeof(double), bytes_i = H.i.size() * sizeof(int);
  int rc = buf.ensure(bytes_d + bytes_i);
  if (rc) return rc;
  char *base = buf.as<char>();
  HIPCHK(hipMemcpy(base, H.d.data(), bytes_d, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(base + bytes_d, H.i.data(), bytes_i, hipMemcpyHostToDevice));
  const double *dd = reinterpret_cast<const double *>(base);
  const int *di = reinterpret_cast<const int *>(base + bytes_d);
  const size_t md = H.md;
  L.freq = dd + 0 * md; L.hcf = dd + 1 * md; L.a_coeff = dd + 2 * md; L.b21 = dd + 3 * md;
  L.b12 = dd + 4 * md; L.e_lower = dd + 5 * md; L.g_up = dd + 6 * md; L.g_lo = dd + 7 * md;
  L.air_broad = dd + 8 * md; L.t_dep = dd + 9 * md; L.evib_up = dd + 10 * md; L.evib_lo = dd + 11 * md;
  L.ic = di + 0 * md; L.lev_up = di + 1 * md; L.lev_lo = di + 2 * md;
  L.n_lines = (int)H.m;
  return SR_OK;
}

// rows `sel` of H (ascending, so the subset stays sorted by window centre)
HostLines subset(const HostLines &H, const std::vector<int64_t> &sel) {
  HostLines S;
  S.m = (int64_t)sel.size();
  S.md = (size_t)std::max<int64_t>(S.m, 1);
  S.d.assign(12 * S.md, 0.0);
  S.i.assign(3 * S.md, 0);
  for (int64_t q = 0; q < S.m; ++q) {
    for (int a = 0; a < 12; ++a) S.d[a * S.md + q] = H.d[a * H.md + sel[q]];
    for (int a = 0; a < 3; ++a) S.i[a * S.md + q] = H.i[a * H.md + sel[q]];
  }
  return S;
}

} // namespace

// Device side of a lineset from its host line lists (main: centre inside its window; outer: not).
static int lineset_upload(sr_lineset *ls) {
  const HostLines &H = ls->host;
  ls->n_lines = H.m;
  ls->ic.assign(H.i.begin(), H.i.begin() + H.m);
  ls->freq_max = 0.0;
  for (int64_t q = 0; q < H.m; ++q) ls->freq_max = std::max(ls->freq_max, H.d[q]);
  for (int64_t q = 0; q < ls->host_outer.m; ++q) ls->freq_max = std::max(ls->freq_max, ls->host_outer.d[q]);
  ls->gamma_max = 0.0;
  ls->ndep_min = ls->ndep_max = H.m > 0 ? H.d[9 * H.md] : 0.0;
  for (int64_t q = 0; q < H.m; ++q) {
    ls->gamma_max = std::max(ls->gamma_max, H.d[8 * H.md + q]);
    ls->ndep_min = std::min(ls->ndep_min, H.d[9 * H.md + q]);
    ls->ndep_max = std::max(ls->ndep_max, H.d[9 * H.md + q]);
  }
  ls->n_disp_lo = ls->n_disp_hi = 0;
  if (H.m > 0) {
    const double g0 = grid_at(ls->gp, 0), g1 = grid_at(ls->gp, (int)ls->gp.n_grid - 1), tol = 0.75 * ls->gp.gstep;
    while (ls->n_disp_lo < H.m && H.i[ls->n_disp_lo] == 0 && H.d[ls->n_disp_lo] < g0 - tol) ++ls->n_disp_lo;
    while (ls->n_disp_hi < H.m - ls->n_disp_lo && H.i[H.m - 1 - ls->n_disp_hi] == (int)ls->gp.n_grid - 1 &&
           H.d[H.m - 1 - ls->n_disp_hi] > g1 + tol)
      ++ls->n_disp_hi;
  }
  int rc = upload_soa(H, ls->d_lines, ls->L);
  if (rc) return rc;
  ls->n_outer = (int)ls->host_outer.m;
  if (ls->n_outer > 0) {
    rc = upload_soa(ls->host_outer, ls->d_lines_outer, ls->Lo);
    if (rc) return rc;
  }
  { // direct index into the sorted centres (IcIndex)
    const int64_t m = H.m;
    const int x0 = m > 0 ? ls->ic.front() : 0, n_tab = m > 0 ? ls->ic.back() - x0 + 2 : 1;
    std::vector<int> first((size_t)n_tab);
    size_t q = 0;
    for (int t = 0; t < n_tab; ++t) {
      while (q < (size_t)m && ls->ic[q] < x0 + t) ++q;
      first[(size_t)t] = (int)q;
    }
    rc = ls->d_first.ensure(sizeof(int) * (size_t)n_tab);
    if (rc) return rc;
    HIPCHK(hipMemcpy(ls->d_first.p, first.data(), sizeof(int) * (size_t)n_tab, hipMemcpyHostToDevice));
    ls->first_x0 = x0;
    ls->first_n = n_tab;
  }
  if (!ls->work) { // a lineset of its own (level_set points its children at the parent's before the upload)
    ls->work = &ls->own_work;
    return ls->own_work.init();
  }
  return SR_OK;
}

extern "C" {

int sr_lineset_create(const sr_lines_desc *ld, const sr_isomolec_desc *iso, const sr_grid_desc *gd,
                      sr_lineset **out, int64_t *n_kept) {
  if (!ld || !iso || !gd || !out) return SR_ERR_ARG;
  *out = nullptr;
  if (ld->n_lines < 0 || iso->n_levels < 0 || !(gd->step > 0.0) || !(iso->mm > 0.0)) return SR_ERR_ARG;
  if (gd->n_grid < 2) return SR_ERR_ARG;
  if (gd->n_grid > 2000000) return SR_ERR_LIMIT;             // imxsig_long, spect_classes.py:29,362
  if (iso->n_levels > SR_MAX_LEVELS) return SR_ERR_LIMIT;
  if (ld->n_lines > 0 && (!ld->freq || !ld->a_coeff || !ld->e_lower || !ld->g_up || !ld->g_lo ||
                          !ld->air_broad || !ld->t_dep_broad))
    return SR_ERR_ARG;
  if (iso->n_levels > 0 && (!iso->level_energy || (ld->n_lines > 0 && (!ld->lev_up || !ld->lev_lo))))
    return SR_ERR_ARG;

  GridParams gp;
  gp.w0 = gd->w0;
  gp.gstep = gd->step;
  gp.n_grid = (int)gd->n_grid;
  // spect_classes.py:1446: np.arange(-imxsig*s/2, imxsig*s/2, s) = start + m*delta
  gp.lin_start = -(double)kImxsig * gp.gstep / 2;
  {
    const double lin_stop = (double)kImxsig * gp.gstep / 2;
    const double len = std::ceil((lin_stop - gp.lin_start) / gp.gstep);
    if (len != (double)kImxsig) {
      g_err = "np.arange window would not have 13010 points for this step (f2py would reject it)";
      return SR_ERR_UNSUPPORTED;
    }
    const double nxt = gp.lin_start + gp.gstep;
    gp.lin_delta = nxt - gp.lin_start;
  }

  const int nlev = iso->n_levels;
  const int64_t n = ld->n_lines;
  std::vector<int64_t> keep;
  keep.reserve(n);
  for (int64_t i = 0; i < n; ++i) {
    if (nlev > 0) {
      const int lu = ld->lev_up[i], ll = ld->lev_lo[i];
      // LinkToMolec must find both levels; its if/elif never finds the lower
      // level when it equals the upper one (spect_classes.py:135-150)
      if (lu < 0 || ll < 0 || lu >= nlev || ll >= nlev || lu == ll) continue;
    }
    keep.push_back(i);
  }
  const int64_t m = (int64_t)keep.size();
  std::vector<int> ic(m);
  std::vector<char> is_outer(m, 0);
  const int ng = gp.n_grid;
  for (int64_t q = 0; q < m; ++q) {
    const double f = ld->freq[keep[q]];
    if (!(f == f)) return SR_ERR_ARG;
    // closest_grid: first arg-min of |grid - f| (spect_classes.py:1941)
    double t = std::floor((f - gp.w0) / gp.gstep);
    long j0 = t < -4 ? -4 : (t > ng + 4 ? ng + 4 : (long)t);
    long a = std::max<long>(0, j0 - 2), b = std::min<long>(ng - 1, j0 + 3);
    if (a > b) { a = b = (j0 < 0 ? 0 : ng - 1); }
    long best = a;
    double bv = std::fabs(grid_at(gp, (int)a) - f);
    for (long j = a + 1; j <= b; ++j) {
      const double v = std::fabs(grid_at(gp, (int)j) - f);
      if (v < bv) { bv = v; best = j; }
    }
    ic[q] = (int)best;
    // A line farther than half a window (~3.25 cm-1) from the grid has its centre outside its own
    // window: humliv_bb takes an outer branch (lineshape.f:272, 358).  Kept apart, see launch_outer.
    WinX xf{gp.lin_start, gp.lin_delta, grid_at(gp, (int)best)};
    is_outer[q] = !(xf(1) < f && f < xf(kImxsig));
  }
  std::vector<int64_t> ord(m);
  std::iota(ord.begin(), ord.end(), 0);
  std::stable_sort(ord.begin(), ord.end(), [&](int64_t x, int64_t y) { return ic[x] < ic[y]; });

  sr_lineset *ls = new sr_lineset();
  ls->gp = gp;
  ls->mol = iso->mol;
  ls->iso = iso->iso;
  ls->mm = iso->mm;
  ls->n_levels = nlev;
  ls->e_lev.assign(iso->level_energy, iso->level_energy + nlev);

  // the line list in device layout, main and outer lines apart (both in centre order)
  HostLines all;
  all.m = m;
  all.md = (size_t)std::max<int64_t>(m, 1);
  all.d.assign(12 * all.md, 0.0);
  all.i.assign(3 * all.md, 0);
  const size_t md = all.md;
  const double h = kHcgs, c = kCcgs;
  std::vector<int64_t> sel_main, sel_outer;
  for (int64_t q = 0; q < m; ++q) {
    const int64_t s = keep[ord[q]];
    const double f = ld->freq[s];
    const double fact_2 = 2 * h * (c * c) * std::pow(f, 3.0); // spect_classes.py:1743
    const double b21 = ld->a_coeff[s] / fact_2;               // :1750
    const double b12 = ld->g_lo[s] != 0.0 ? b21 * ld->g_up[s] / ld->g_lo[s] : 0.0; // :1783
    const int lu = nlev > 0 ? ld->lev_up[s] : 0, ll = nlev > 0 ? ld->lev_lo[s] : 0;
    double *d = all.d.data();
    d[0 * md + q] = f;
    d[1 * md + q] = h * c * f;
    d[2 * md + q] = ld->a_coeff[s];
    d[3 * md + q] = b21;
    d[4 * md + q] = b12;
    d[5 * md + q] = ld->e_lower[s];
    d[6 * md + q] = ld->g_up[s];
    d[7 * md + q] = ld->g_lo[s];
    d[8 * md + q] = ld->air_broad[s];
    d[9 * md + q] = ld->t_dep_broad[s];
    d[10 * md + q] = nlev > 0 ? iso->level_energy[lu] : 0.0; // spect_classes.py:318-324
    d[11 * md + q] = nlev > 0 ? iso->level_energy[ll] : 0.0;
    all.i[0 * md + q] = ic[ord[q]];
    all.i[1 * md + q] = lu;
    all.i[2 * md + q] = ll;
    (is_outer[ord[q]] ? sel_outer : sel_main).push_back(q);
  }
  ls->host = subset(all, sel_main);
  ls->host_outer = subset(all, sel_outer);
  const int rc = lineset_upload(ls);
  if (rc) { sr_lineset_destroy(ls); return rc; }
  if (n_kept) *n_kept = m;
  *out = ls;
  return SR_OK;
}

int sr_lineset_set_bounds_temps(sr_lineset *ls, const double *temps_bounds, int n_layers) {
  if (!ls || n_layers < 0) return SR_ERR_ARG;
  if (!temps_bounds || n_layers == 0) {
    ls->bounds_temps.clear();
    return SR_OK;
  }
  for (int k = 0; k < n_layers; ++k)
    if (!(temps_bounds[k] > 0.0)) return SR_ERR_ARG;
  ls->bounds_temps.assign(temps_bounds, temps_bounds + n_layers);
  return SR_OK;
}

int sr_lineset_set_linear_weights(sr_lineset *ls, int on) {
  if (!ls) return SR_ERR_ARG;
  ls->linear_weights = on != 0;
  return SR_OK;
}

int sr_lineset_destroy(sr_lineset *ls) {
  if (!ls) return SR_OK;
  (void)hipDeviceSynchronize(); // work of the last calls may still be in flight on the internal streams
  for (sr_lineset *child : ls->level_sets) sr_lineset_destroy(child);
  ls->level_sets.clear();
  for (sr_lineset *child : ls->level_up_sets) sr_lineset_destroy(child);
  ls->level_up_sets.clear();
  ls->d_lines_outer.release();
  ls->d_gscratch.release();
  ls->d_lines.release();
  ls->d_first.release();
  if (ls->work == &ls->own_work) ls->own_work.release();
  ls->mc.release();
  delete ls;
  return SR_OK;
}

} // extern "C"

// Translation operator of the box-pair far field: built once per process and device (0.8 MB).
static int m2l_table_dev(const double **out) {
  static std::mutex mu;
  static std::map<int, double *> tabs;
  int dev = 0;
  HIPCHK(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(mu);
  auto it = tabs.find(dev);
  if (it == tabs.end()) {
    const size_t n = (size_t)2 * kM2LOffsets * kM2LQ * kM2LRow;
    std::vector<double> h(n);
    m2l_table_host(h.data());
    double *d = nullptr;
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&d), sizeof(double) * n));
    HIPCHK(hipMemcpy(d, h.data(), sizeof(double) * n, hipMemcpyHostToDevice));
    it = tabs.emplace(dev, d).first;
  }
  *out = it->second;
  return SR_OK;
}

// Downward-pass operator of the hierarchy (sr_l2l_kernel): built once per process and device (8 KB).
static int l2l_table_dev(const double **out) {
  static std::mutex mu;
  static std::map<int, double *> tabs;
  int dev = 0;
  HIPCHK(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(mu);
  auto it = tabs.find(dev);
  if (it == tabs.end()) {
    const size_t n = (size_t)2 * kFC * kFC;
    std::vector<double> h(n);
    l2l_table_host(h.data());
    double *d = nullptr;
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&d), sizeof(double) * n));
    HIPCHK(hipMemcpy(d, h.data(), sizeof(double) * n, hipMemcpyHostToDevice));
    it = tabs.emplace(dev, d).first;
  }
  *out = it->second;
  return SR_OK;
}

// Per-layer scalars of a call (host, fp64): T, P [atm], 296/T, sqrt(2 N_A k T ln2 / MM), the same at the boundary
// temperatures, the level populations, then three int rows: pole margin, source pole radius, widest zone.
//   mar: the lineset whose lines bound the margins -- the handle itself, or its PARENT for the far-only level passes of
//        the multi-channel route (every kernel of that route must place a (line, slot) pair on the same side of the
//        near / far split: the pole margin enters the admissibility threshold)
static size_t layer_stage_doubles(int nl, int npop) { return (size_t)nl * (8 + npop); }
static size_t layer_stage_bytes(int nl, int npop) { return sizeof(double) * layer_stage_doubles(nl, npop) + sizeof(int) * 3 * (size_t)nl; }
static int fill_layer_stage(const sr_lineset *ls, const sr_lineset *mar, const sr_lineset *bown, const sr_layers_desc *atm, double *T) {
  const int nl = atm->n_layers, nlev = ls->n_levels, npop = nlev > 0 ? nlev : 1;
  const size_t hl_doubles = layer_stage_doubles(nl, npop);
  const bool frozen = !bown->bounds_temps.empty();
  double *pa = T + nl, *tr = pa + nl, *sq = tr + nl, *ltr = sq + nl, *ltrb = ltr + nl, *sqb = ltrb + nl, *tb = sqb + nl,
         *pop = tb + nl;
  std::vector<double> q(nl);
  if (atm->q_part) {
    std::copy(atm->q_part, atm->q_part + nl, q.begin());
  } else {
    const int rc = sr_calc_partition_sum(ls->mol, ls->iso, atm->temps, nl, q.data());
    if (rc) return rc;
  }
  for (int k = 0; k < nl; ++k) {
    T[k] = atm->temps[k];
    pa[k] = atm->press[k] * kHpaToAtm;                                        // spect_classes.py:2034
    tr[k] = kTref / T[k];                                                     // :1972
    sq[k] = std::sqrt(2 * kAvogadro * kKcgs * T[k] * kLn2 / ls->mm);          // :1984
    ltr[k] = std::log(tr[k]);
    {
      const double Tb = frozen ? bown->bounds_temps[k] : T[k]; // where the region boundaries are placed
      tb[k] = Tb;
      ltrb[k] = std::log(kTref / Tb);
      sqb[k] = std::sqrt(2 * kAvogadro * kKcgs * Tb * kLn2 / ls->mm);
    }
    {
      // pole margin of the far-field expansions: the region-1 rational has its poles at
      // |x| = sqrt(1/2 + ry^2), i.e. within 0.71 dw' of the line centre on the real axis.
      // Frozen boundaries (sr_lineset_set_bounds_temps): the zone of a line is placed with the widths of the
      // boundary temperature Tb, the poles sit where the call's own widths put them -- the bounds below cover both
      // (with the call's T alone a frozen zone wider than zmax, Tb > T, lost its outer points: ADVICE round 3).
      const double sq_w = std::max(sq[k], sqb[k]);
      const double dwp_max = mar->freq_max / kCcgs * sq_w / std::sqrt(kLn2);
      int *pmh = reinterpret_cast<int *>(T + hl_doubles);
      pmh[k] = (int)std::ceil(0.71 * dwp_max / ls->gp.gstep) + 1;
      // box-pair mode: the multipole series of a source box converges outside the largest |pole| =
      // sqrt(1/2 + ry^2) dw' = sqrt(dw'^2 / 2 + lw^2) of its lines (bound over the lines of the layer)
      const double trb = frozen ? kTref / bown->bounds_temps[k] : tr[k];
      const double lw_max = mar->gamma_max * pa[k] *
                            std::max(std::max(std::pow(tr[k], mar->ndep_min), std::pow(tr[k], mar->ndep_max)),
                                     std::max(std::pow(trb, mar->ndep_min), std::pow(trb, mar->ndep_max)));
      const double pole = std::sqrt(0.5 * dwp_max * dwp_max + lw_max * lw_max) / ls->gp.gstep;
      pmh[nl + k] = (int)std::ceil(std::min(pole, 1e6));
      // widest region-2/3/4 zone of the layer, in grid points from the line centre: region 1 starts where
      // |x| - ry >= 15 (lineshape.f:447-454), i.e. (lw + 15 dw') / step points out, +-1 for the nint and the
      // centre's offset inside its grid cell.  A bound over the lines (every kernel reads this one value, so they
      // agree on who evaluates what); kernels clamp it to the window half-width.
      pmh[2 * nl + k] = (int)std::min(std::ceil((lw_max + 15.0 * dwp_max) / ls->gp.gstep) + 2.0, (double)kHalf);
    }
    if (nlev > 0) {
      for (int lv = 0; lv < nlev; ++lv) {
        const double vibt = atm->tvib ? atm->tvib[(size_t)lv * nl + k] : T[k]; // smm:2062-2065
        pop[(size_t)k * npop + lv] = std::exp(-kC2 * ls->e_lev[lv] / vibt) / q[k]; // smm:2073
      }
    } else {
      pop[k] = 1 / q[k]; // smm:2054
    }
  }
  return SR_OK;
}
// The device view of a pushed layer stage.  d_pm: [3][n_layers] pole margin | source pole radius | widest zone.
static LayersDev layers_dev_of(const double *dl, int nl, int npop, bool frozen, bool linear_w, const int **d_pm) {
  LayersDev A;
  A.temps = dl; A.p_atm = dl + nl; A.trat = dl + 2 * nl; A.sqk = dl + 3 * nl; A.ltrat = dl + 4 * nl;
  A.ltrat_b = dl + 5 * nl; A.sqk_b = dl + 6 * nl; A.temps_b = dl + 7 * nl; A.pop = dl + 8 * nl;
  A.frozen = frozen ? 1 : 0;
  A.linear_w = frozen && linear_w ? 1 : 0;
  A.n_layers = nl; A.n_pop = npop;
  A.sqrt_ln2 = std::sqrt(kLn2);            // spect_classes.py:1999
  A.sqrt_pi_ln2 = std::sqrt(kPi / kLn2);   // :1997
  *d_pm = reinterpret_cast<const int *>(dl + layer_stage_doubles(nl, npop));
  return A;
}
// The far-field box hierarchy of a shard of n_pts points (the part of FarParams every kernel agrees on)
static void far_hierarchy(size_t n_pts, int nl, FarParams *fp) {
  fp->n_levels = kMaxFarLevels;
  fp->n_layers = nl;
  fp->n_boxes_total = 0;
  fp->top_first = n_pts <= 16384 ? 1 : 0; // see sr_farfield_kernel
  for (int lv = 0; lv < kMaxFarLevels; ++lv) {
    const int W = 64 << lv;
    fp->box_count[lv] = (int)((n_pts + W - 1) / W);
    fp->box_off[lv] = fp->n_boxes_total;
    fp->n_boxes_total += fp->box_count[lv];
  }
}

// Options of coef_op beyond the public entry points'.
//   far_coef: a FAR-ONLY pass for the multi-channel route (mc_pass) -- tables and the far-field chain only, the
//   coefficients [n_layers][n_boxes_total][2][kFC] written there and folded down to level 0 (sr_l2l_kernel: wider levels
//   hold partial sums afterwards, level 0 everything); no near kernels, no outer lines, abs_out / emi_out
//   untouched (may be null); the margins are the PARENT's (fill_layer_stage); *far_has = whether any line met the shard.
struct CoefOpt {
  double *far_coef = nullptr;
  bool *far_has = nullptr;
};

// The coefficient op with the output weights of `W` (sr_kernels.hpp); the public entry points below
// choose W.  abs_out / emi_out: DEVICE [n_layers][g_hi - g_lo].
static int coef_op(sr_lineset *ls, const sr_layers_desc *atm, int64_t g_lo, int64_t g_hi, double *abs_out,
                   double *emi_out, void *stream, const WeightMode W, const CoefOpt opt = CoefOpt()) {
  const bool far_only = opt.far_coef != nullptr;
  if (!ls || !atm || (!far_only && (!abs_out || !emi_out))) return SR_ERR_ARG;
  if (atm->n_layers <= 0 || !atm->temps || !atm->press) return SR_ERR_ARG;
  if (g_lo < 0 || g_hi > ls->gp.n_grid || g_lo >= g_hi) return SR_ERR_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  CoefWork &w = *ls->work;
  // frozen region boundaries are set on the handle the caller holds: a per-level sub-lineset follows its parent's
  sr_lineset *const bown = ls->parent ? ls->parent : ls;
  const int nl = atm->n_layers, nlev = ls->n_levels, npop = nlev > 0 ? nlev : 1;
  for (int k = 0; k < nl; ++k)
    if (!(atm->temps[k] > 0.0) || !(atm->press[k] >= 0.0)) return SR_ERR_ARG;
  // one snapshot of the mode switches per call
  const int variant = g_variant.load();
  const bool timing = g_timing.load() != 0;
  // Sparse line sets (the per-level sub-linesets of the pair tables: 9-15 % of a hot-band list): the box-pair far field
  // has a fixed cost per box and layer -- S2M, M2M, M2L over every box whatever it holds -- that the per-line
  // expansions at every level (mode 1) do not have: below ~0.37 lines per grid point they are the faster route
  // (tools/ff_mode_crossover.py: 0.3: 2.54 vs 2.61 ms, 0.4: 3.07 vs 3.03, 1.0: 6.74 vs 5.62; the 12 pair tables of the
  // configs[1] list 19.1 vs 21.5 ms).  Mode 3 (the default) switches at 0.35.
  constexpr double sparse_thr = 0.35;
  const int far_mode = g_far_field.load();
  const bool sparse_set = far_mode == 3 && (double)ls->n_lines < sparse_thr * (double)ls->gp.n_grid;
  const int far_field = far_mode == 3 ? (sparse_set ? 1 : 2) : far_mode;
  if (far_only && !far_field) return SR_ERR_UNSUPPORTED; // (mc_pass takes the per-level route in the exact mode)
  const bool counting = g_counting.load() != 0 && far_field;
  // 1: the decoupled, phased pipeline (far-field modes); 0: the kernels one after the other on the caller's stream, on
  // table set 0 -- sr_set_overlap(0), the exact mode and the counting passes (whose counters are zeroed and read on the
  // caller's stream).  (Round 3's order, overlap 2, was the A/B partner of rounds 4-5 and is gone.)
  const int overlap = (g_overlap.load() != 0 && far_field && !counting) ? 1 : 0;
  const size_t table_budget = g_table_budget.load();

  // The per-(line, layer) record tables cost 128 B each, the far-field scratch of a layer (local coefficients
  // of all levels' target boxes, multipole moments of the source boxes) 1 KB per 64 grid points + 1.3 KB per 64
  // points of grid and window halo; a long LOS (the reference allows imxstp = 8000 steps) is processed in layer
  // batches that keep them under g_table_budget.
  {
    const size_t n_pts_b = (size_t)(g_hi - g_lo);
    const size_t far_per_layer = !far_field ? 0
        : (2 * (n_pts_b / 64 + 2)) * (size_t)(2 * kFC) * sizeof(double) +
          (far_field == 2 ? (2 * ((n_pts_b + 64 * kSrcPad + kHalf) / 64 + 16)) * (size_t)kMomPerBox * sizeof(double) : 0);
    const size_t per_layer = (size_t)std::max<int64_t>(ls->n_lines, 1) * (sizeof(FastRec) + sizeof(ColdRec)) *
                                 (overlap ? 2 : 1) + far_per_layer // two table sets with overlap
                             + (overlap ? 2 * sizeof(double) * n_pts_b : 0)          // the zones kernel's private sums (small shards)
                             + sizeof(OuterRec) * (size_t)std::max(ls->n_outer, 0);  // records of the outer lines
    const int nl_max = (int)std::max<size_t>(1, table_budget / per_layer);
    if (nl > nl_max && far_only) {
      g_err = "far-only pass over more layers than the table budget holds (mc_pass sizes its row batches for the parent)";
      return SR_ERR_LIMIT;
    }
    if (nl > nl_max) {
      const size_t n_pts_all = (size_t)(g_hi - g_lo);
      // sr_lineset_set_bounds_temps: every batch sees its own slice of the boundary temperatures
      const std::vector<double> bounds_all = bown->bounds_temps;
      if (!bounds_all.empty() && (int)bounds_all.size() != nl) {
        g_err = "sr_lineset_set_bounds_temps was given another number of layers than this call";
        return SR_ERR_ARG;
      }
      struct Restore {
        sr_lineset *ls; const std::vector<double> &all;
        ~Restore() { ls->bounds_temps = all; }
      } restore{bown, bounds_all};
      for (int k0 = 0; k0 < nl; k0 += nl_max) {
        sr_layers_desc sub = *atm;
        sub.n_layers = std::min(nl_max, nl - k0);
        sub.temps = atm->temps + k0;
        sub.press = atm->press + k0;
        sub.q_part = atm->q_part ? atm->q_part + k0 : nullptr;
        std::vector<double> tv;
        if (atm->tvib) { // [n_levels][n_layers] -> the batch's columns
          tv.resize((size_t)nlev * sub.n_layers);
          for (int lv = 0; lv < nlev; ++lv)
            std::copy(atm->tvib + (size_t)lv * nl + k0, atm->tvib + (size_t)lv * nl + k0 + sub.n_layers,
                      tv.begin() + (size_t)lv * sub.n_layers);
          sub.tvib = tv.data();
        }
        if (!bounds_all.empty()) bown->bounds_temps.assign(bounds_all.begin() + k0, bounds_all.begin() + k0 + sub.n_layers);
        const int rc = coef_op(ls, &sub, g_lo, g_hi, abs_out + (size_t)k0 * n_pts_all,
                               emi_out + (size_t)k0 * n_pts_all, stream, W);
        if (rc) return rc;
      }
      return SR_OK;
    }
  }

  // per-layer scalars (host, fp64: fill_layer_stage)
  const bool frozen = !bown->bounds_temps.empty(); // sr_lineset_set_bounds_temps
  if (frozen && (int)bown->bounds_temps.size() != nl) {
    g_err = "sr_lineset_set_bounds_temps was given another number of layers than this call";
    return SR_ERR_ARG;
  }
  const size_t hl_bytes = layer_stage_bytes(nl, npop);
  // Table set of this call and the stream its preparation runs on.  With overlap, call c + 1
  // prepares set (c + 1) % 2 on prep_st while the kernels of call c (which the caller's stream is
  // still running) read set c % 2: the HBM-write-bound prep kernel hides behind the VALU-bound ones.
  const int b = overlap ? (w.parity ^= 1) : 0;
  hipStream_t pst = st;
  // Order this call after the previous one on this handle (see ev_last_done): a no-op when both use
  // the same stream.  The next call's table preparation (pst) needs only its own table set to be free.
  if (w.last_done_recorded) HIPCHK(hipStreamWaitEvent(st, w.ev_last_done, 0));
  if (overlap) {
    if (!w.prep_st) {
      HIPCHK(hipStreamCreateWithFlags(&w.prep_st, hipStreamNonBlocking));
      HIPCHK(hipEventCreateWithFlags(&w.ev_op0, hipEventDefault));
      for (int i = 0; i < 2; ++i) {
        HIPCHK(hipEventCreateWithFlags(&w.ev_prep_done[i], hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&w.ev_tables_free[i], hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&w.ev_far_done[i], hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&w.ev_zones_done[i], hipEventDisableTiming));
      }
    }
    pst = w.prep_st;
  }
  const bool decoupled = overlap == 1; // the decoupled, phased pipeline (see the far-field branch below)
  Stager &SL = w.s_layers[b];
  DevBuf &d_fast = w.d_fast[b], &d_cold = w.d_cold[b];
  int rc = SL.prepare(hl_bytes);
  if (rc) return rc;
  rc = fill_layer_stage(ls, far_only && ls->parent ? ls->parent : ls, bown, atm, SL.host<double>());
  if (rc) return rc;
  // set b was last read by the kernels of the call before the previous one
  if (overlap && w.free_recorded[b]) HIPCHK(hipStreamWaitEvent(pst, w.ev_tables_free[b], 0));
  rc = SL.push(hl_bytes, pst);
  if (rc) return rc;
  const int *d_pm = nullptr;
  const LayersDev A = layers_dev_of(SL.d.as<double>(), nl, npop, frozen, bown->linear_weights, &d_pm);
  const int *zmax_dev = d_pm + 2 * nl; // [n_layers] widest zone (host bound, see fill_layer_stage)

  // lines whose window [ic-6505, ic+6504] meets the shard
  const auto lo_it = std::lower_bound(ls->ic.begin(), ls->ic.end(), (int)g_lo - (kHalf - 1));
  const auto hi_it = std::upper_bound(ls->ic.begin(), ls->ic.end(), (int)g_hi - 1 + kHalf);
  const int line_lo = (int)(lo_it - ls->ic.begin());
  const int n_sub = (int)(hi_it - lo_it);
  const size_t n_pts = (size_t)(g_hi - g_lo);

  const IcIndex ix{ls->d_first.as<int>(), ls->first_x0, ls->first_n, line_lo, n_sub};

  w.timed = false;
  w.overlapped = false;
  w.counted = false;
  unsigned long long *d_cnt = nullptr;
  if (counting) {
    rc = w.d_counts.ensure(sizeof(unsigned long long) * kCntN);
    if (rc) return rc;
    d_cnt = w.d_counts.as<unsigned long long>();
    HIPCHK(hipMemsetAsync(d_cnt, 0, sizeof(unsigned long long) * kCntN, st));
  }
  // lines with their centre outside their own window: added after the main kernels (launch_outer)
  auto add_outer = [&]() -> int {
    if (ls->n_outer <= 0) return SR_OK;
    int rc2 = w.d_outer_recs.ensure(sizeof(OuterRec) * (size_t)ls->n_outer * nl);
    if (rc2) return rc2;
    LAUNCHCHK(launch_outer(ls->Lo, ls->n_outer, A, ls->gp, W, w.d_outer_recs.as<OuterRec>(), (int)g_lo, (int)g_hi,
                           abs_out, emi_out, st));
    return SR_OK;
  };
  if (opt.far_has) *opt.far_has = n_sub > 0;
  if (n_sub <= 0 && far_only) return SR_OK; // nothing staged on the internal streams: no event to record
  if (n_sub <= 0) {
    HIPCHK(hipMemsetAsync(abs_out, 0, sizeof(double) * n_pts * nl, st));
    HIPCHK(hipMemsetAsync(emi_out, 0, sizeof(double) * n_pts * nl, st));
    if (ls->n_outer > 0) {
      // the layer scalars were pushed on pst: the caller's stream must see them
      if (overlap) {
        HIPCHK(hipEventRecord(w.ev_prep_done[b], pst));
        HIPCHK(hipStreamWaitEvent(st, w.ev_prep_done[b], 0));
      }
      rc = add_outer();
      if (rc) return rc;
      if (overlap) {
        HIPCHK(hipEventRecord(w.ev_tables_free[b], st));
        w.free_recorded[b] = true;
      }
      HIPCHK(hipEventRecord(w.ev_last_done, st));
      w.last_done_recorded = true;
    }
    return SR_OK;
  }
  rc = d_fast.ensure(sizeof(FastRec) * ((size_t)n_sub * nl + 1));
  if (rc) return rc;
  rc = d_cold.ensure(sizeof(ColdRec) * ((size_t)n_sub * nl + 1));
  if (rc) return rc;


  // (measurement hook, serial schedule only: see g_repeat_kernel)
  const int rep_k = overlap ? -1 : g_repeat_kernel.load(), rep_n = g_repeat_n.load();
  auto reps = [&](int k) { return k == rep_k ? rep_n : 1; };
  if (timing) HIPCHK(hipEventRecord(w.ev[0], pst));
  // cold records: far-field mode reads them for zones inside the shard only, exact mode for window ends too
  for (int r = 0; r < reps(0); ++r)
  LAUNCHCHK(launch_prep(ls->L, A, ls->gp, W, line_lo, n_sub, far_only ? INT_MAX / 2 : (far_field ? (int)g_lo : INT_MIN / 2),
                        far_only ? INT_MIN / 2 : (far_field ? (int)g_hi - 1 : INT_MAX / 2), d_fast.as<FastRec>(),
                        d_cold.as<ColdRec>(), pst)); // (far-only: an empty cold range, no region-2..4 records)
  if (timing) HIPCHK(hipEventRecord(w.ev[1], pst));
  if (overlap) { // the caller's stream takes over once the tables are ready
    HIPCHK(hipEventRecord(w.ev_prep_done[b], pst));
    HIPCHK(hipStreamWaitEvent(st, w.ev_prep_done[b], 0));
    if (timing) HIPCHK(hipEventRecord(w.ev_op0, st));
  }
  if (far_field) {
    FarParams fp;
    far_hierarchy(n_pts, nl, &fp);
    if (!far_only) {
      rc = w.d_coef[b].ensure(sizeof(double) * (size_t)nl * fp.n_boxes_total * 2 * kFC);
      if (rc) return rc;
    }
    fp.pm = d_pm;
    fp.coef = far_only ? opt.far_coef : w.d_coef[b].as<double>();
    fp.m2l = far_field == 2 ? 1 : 0;
    fp.rows = sparse_set ? 1 : 0; // the sparse sets' own kernel (sr_farfield_rows_kernel: a box for eight layers per wave)
    fp.pm_src = d_pm + nl;
    fp.disp_lo_end = (int)std::min<int64_t>(std::max<int64_t>(ls->n_disp_lo - line_lo, 0), n_sub);
    fp.disp_hi_begin = (int)std::min<int64_t>(std::max<int64_t>(ls->n_lines - ls->n_disp_hi - line_lo, 0), n_sub);
    fp.mom = nullptr;
    fp.tab = nullptr;
    for (int lv = 0; lv < kMaxFarLevels; ++lv) fp.n_src[lv] = fp.src_off[lv] = 0;
    if (fp.m2l) {
      // level-0 source boxes: kSrcPad left of the shard, the shard, the window half-width right of it; a whole
      // number of widest boxes, so that every level halves exactly
      const int top_boxes = (int)(((size_t)kSrcPad * 64 + n_pts + kHalf + 64) >> (6 + kMaxFarLevels - 1)) + 1;
      int total = 0;
      for (int lv = 0; lv < kMaxFarLevels; ++lv) {
        fp.n_src[lv] = top_boxes << (kMaxFarLevels - 1 - lv);
        fp.src_off[lv] = total;
        total += fp.n_src[lv];
      }
      rc = w.d_mom[b].ensure(sizeof(double) * (size_t)total * nl * kMomPerBox);
      if (rc) return rc;
      fp.mom = w.d_mom[b].as<double>();
      rc = m2l_table_dev(&fp.tab);
      if (rc) return rc;
    }
    // far-field pass(es): per-line expansions (all levels, or level 0 of the box-pair mode), then the box pairs
    auto far_pass = [&](hipStream_t fs) -> int {
      for (int r = 0; r < reps(1); ++r)
        LAUNCHCHK(launch_farfield(d_fast.as<FastRec>(), ix, zmax_dev, n_sub, nl, (int)g_lo, (int)g_hi, fp,
                                  d_cnt, fs));
      if (fp.m2l) {
        for (int r = 0; r < reps(2); ++r)
          LAUNCHCHK(launch_m2l(d_fast.as<FastRec>(), ix, zmax_dev, n_sub, nl, (int)g_lo, (int)g_hi, fp, d_cnt, fs, 1));
        for (int r = 0; r < reps(3); ++r)
          LAUNCHCHK(launch_m2l(d_fast.as<FastRec>(), ix, zmax_dev, n_sub, nl, (int)g_lo, (int)g_hi, fp, d_cnt, fs, 2));
      }
      return SR_OK;
    };
    // Round 4: a decoupled, phased pipeline.  Between consecutive calls only the caller-visible output orders things:
    // the zones kernel (tables -> private sums), the far-field chain (tables -> coefficients) and the preparation of
    // the tables touch scratch of the call's own parity and nothing of the caller's, so each runs on an internal stream
    // of its own as soon as ITS inputs are ready; only the wings kernel (zones' sums + near region 1 + polynomials ->
    // abs / emi) sits on the caller's stream.  Who runs beside whom is decided by what FITS beside whom
    // (tools/r03_timeline.sh, tools/kernel_resources.sh): 16 zones waves fill a CU -- 120 VGPRs each, 4 x 120 of a
    // SIMD's 512, and 16 x 10 KB = all of its LDS -- and a retiring zones wave frees exactly one such slot, which the
    // next zones wave takes unless the other kernel's wave fits it: the wings kernel (80 VGPRs), M2M / M2L (106 / 104)
    // and the one-wave blocks of the preparation (104 VGPRs, 5 KB) do, the level-0 pass (140) and S2M (154) do not and
    // starved beside the zones kernel until it drained -- then S2M -> M2M -> M2L ran alone, latency-bound, for 0.66 ms
    // of every 5.6 ms step (round 3: the zones kernel forked off the caller's stream, the chain on it; the four-wave
    // blocks of the preparation, 20 KB of LDS each, ran only when everything else had drained).  So the step has two
    // phases: B = [wings(c) | level-0 pass(c + 1) | S2M(c + 1)] -- short waves that share the chip fairly --, then
    // A = [zones(c + 1) | M2M, M2L(c + 1) | prep(c + 2)]; the zones kernel is GATED behind the level-0 pass and S2M of
    // its own call (it needs neither), which is what keeps it from flooding the chip before they are through.
    if (decoupled) {
      if (!w.aux) {
        HIPCHK(hipStreamCreateWithFlags(&w.aux, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&w.ev_fork, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&w.ev_join, hipEventDisableTiming));
      }
      if (!w.chain_st) {
        HIPCHK(hipStreamCreateWithFlags(&w.chain_st, hipStreamNonBlocking));
        HIPCHK(hipStreamCreateWithFlags(&w.chain2_st, hipStreamNonBlocking));
        for (int i = 0; i < 2; ++i) {
          HIPCHK(hipEventCreateWithFlags(&w.ev_l0_done[i], hipEventDisableTiming));
          HIPCHK(hipEventCreateWithFlags(&w.ev_s2m_done[i], hipEventDisableTiming));
        }
      }
      double *z_abs = nullptr, *z_emi = nullptr;
      if (!far_only) {
        rc = w.d_zone2[b].ensure(sizeof(double) * 2 * n_pts * nl);
        if (rc) return rc;
        z_abs = w.d_zone2[b].as<double>();
        z_emi = z_abs + n_pts * nl;
      }
      // (the buffers of parity b were last read by the wings kernel two calls ago: the preparation waited for that)
      // far-field chain: level-0 pass on one stream, moments + upward pass on another, translations behind both
      HIPCHK(hipStreamWaitEvent(w.chain_st, w.ev_prep_done[b], 0));
      LAUNCHCHK(launch_farfield(d_fast.as<FastRec>(), ix, zmax_dev, n_sub, nl, (int)g_lo, (int)g_hi, fp, d_cnt, w.chain_st));
      HIPCHK(hipEventRecord(w.ev_l0_done[b], w.chain_st));
      hipStream_t last = w.chain_st;
      if (fp.m2l) {
        HIPCHK(hipStreamWaitEvent(w.chain2_st, w.ev_prep_done[b], 0));
        LAUNCHCHK(launch_m2l(d_fast.as<FastRec>(), ix, zmax_dev, n_sub, nl, (int)g_lo, (int)g_hi, fp, d_cnt, w.chain2_st, 1));
        HIPCHK(hipEventRecord(w.ev_s2m_done[b], w.chain2_st));
        HIPCHK(hipStreamWaitEvent(w.chain2_st, w.ev_l0_done[b], 0));
        LAUNCHCHK(launch_m2l(d_fast.as<FastRec>(), ix, zmax_dev, n_sub, nl, (int)g_lo, (int)g_hi, fp, d_cnt, w.chain2_st, 2));
        last = w.chain2_st;
      }
      if (far_only) { // the downward pass: the wider levels into the level-0 coefficients, behind the chain on its stream
        const double *l2l_tab = nullptr;
        rc = l2l_table_dev(&l2l_tab);
        if (rc) return rc;
        LAUNCHCHK(launch_l2l(fp.coef, nl, fp, l2l_tab, last));
      }
      HIPCHK(hipEventRecord(w.ev_far_done[b], last));
      if (far_only) { // the coefficients are the result: the caller's stream sees them complete
        HIPCHK(hipStreamWaitEvent(st, w.ev_far_done[b], 0));
      } else {
      // zones: needs the tables only; gated behind the kernels that cannot run beside it
      HIPCHK(hipStreamWaitEvent(w.aux, w.ev_prep_done[b], 0));
      HIPCHK(hipStreamWaitEvent(w.aux, w.ev_l0_done[b], 0));
      if (fp.m2l) HIPCHK(hipStreamWaitEvent(w.aux, w.ev_s2m_done[b], 0));
      LAUNCHCHK(launch_near(2, 0, d_fast.as<FastRec>(), d_cold.as<ColdRec>(), ix, zmax_dev, n_sub, nl, (int)g_lo,
                            (int)g_hi, ls->gp, fp, z_abs, z_emi, d_cnt, w.aux));
      HIPCHK(hipEventRecord(w.ev_zones_done[b], w.aux));
      HIPCHK(hipStreamWaitEvent(st, w.ev_far_done[b], 0));
      if (timing) HIPCHK(hipEventRecord(w.ev[2], st));
      HIPCHK(hipStreamWaitEvent(st, w.ev_zones_done[b], 0));
      LAUNCHCHK(launch_near(1, 0, d_fast.as<FastRec>(), d_cold.as<ColdRec>(), ix, zmax_dev, n_sub, nl, (int)g_lo,
                            (int)g_hi, ls->gp, fp, abs_out, emi_out, d_cnt, st, z_abs, z_emi));
      }
      if (timing) HIPCHK(hipEventRecord(w.ev[3], st));
      if (timing) HIPCHK(hipEventRecord(w.ev[4], st));
      w.overlapped = true;
    } else {
      w.overlapped = false;
      rc = far_pass(st);
      if (rc) return rc;
      if (far_only) {
        const double *l2l_tab = nullptr;
        rc = l2l_table_dev(&l2l_tab);
        if (rc) return rc;
        LAUNCHCHK(launch_l2l(fp.coef, nl, fp, l2l_tab, st));
      }
      if (timing) HIPCHK(hipEventRecord(w.ev[2], st));
      // wings (writes) then zones (adds); the repeat hook's zones launches store instead (idempotent)
      for (int part = 1; part <= 2 && !far_only; ++part) {
        const int n_rep = reps(part == 1 ? 5 : 4);
        for (int r = 0; r < n_rep; ++r)
          LAUNCHCHK(launch_near(part, part == 2 && n_rep == 1, d_fast.as<FastRec>(), d_cold.as<ColdRec>(), ix,
                                zmax_dev, n_sub, nl, (int)g_lo, (int)g_hi, ls->gp, fp, abs_out, emi_out,
                                d_cnt, st));
        if (timing) HIPCHK(hipEventRecord(w.ev[2 + part], st));
      }
    }
    w.n_timed = 4;
  } else {
    w.n_timed = 3;
    for (int which = 0; which < 2; ++which) {
      LAUNCHCHK(launch_abscoeff(variant, which, d_fast.as<FastRec>(), d_cold.as<ColdRec>(),
                                ix, zmax_dev, n_sub, nl, (int)g_lo, (int)g_hi, ls->gp,
                                abs_out, emi_out, st));
      if (timing) HIPCHK(hipEventRecord(w.ev[2 + which], st));
    }
  }
  if (!far_only) {
    rc = add_outer(); // after the timing events: not part of the per-kernel times
    if (rc) return rc;
  }
  // a serial call reads table set 0 too: a later pipelined call, which prepares its set on prep_st without waiting for
  // the caller's stream, must find the event behind THIS call's kernels
  if (w.ev_tables_free[b]) {
    HIPCHK(hipEventRecord(w.ev_tables_free[b], st));
    w.free_recorded[b] = true;
  }
  HIPCHK(hipEventRecord(w.ev_last_done, st));
  w.last_done_recorded = true;
  w.pipelined = overlap != 0;
  w.timed = timing;
  w.counted = counting;
  return SR_OK;
}

// Sub-lineset of the lines whose upper or lower level is `level` (same grid, iso-molecule and level
// table), built on first use.  For the 'all' set (no levels) the lineset itself.
// up_only: only the lines whose upper level is `level` (sp_emission / ind_emission select those, spcl:1304-1313).
static int level_set(sr_lineset *ls, int level, sr_lineset **out, bool up_only = false) {
  if (level == -1) { // every line of the iso-molecule
    *out = ls;
    return SR_OK;
  }
  if (ls->n_levels == 0) {
    if (level != 0) return SR_ERR_ARG;
    *out = ls;
    return SR_OK;
  }
  if (level < 0 || level >= ls->n_levels) return SR_ERR_ARG;
  std::vector<sr_lineset *> &sets = up_only ? ls->level_up_sets : ls->level_sets;
  if (sets.empty()) sets.assign((size_t)ls->n_levels, nullptr);
  if (!sets[(size_t)level]) {
    sr_lineset *c = new sr_lineset();
    c->gp = ls->gp;
    c->mol = ls->mol;
    c->iso = ls->iso;
    c->mm = ls->mm;
    c->n_levels = ls->n_levels;
    c->e_lev = ls->e_lev;
    c->work = ls->work; // scratch, streams and events of the parent (see CoefWork)
    c->parent = ls;
    for (int which = 0; which < 2; ++which) {
      const HostLines &H = which ? ls->host_outer : ls->host;
      std::vector<int64_t> sel;
      for (int64_t q = 0; q < H.m; ++q)
        if (H.i[1 * H.md + q] == level || (!up_only && H.i[2 * H.md + q] == level)) sel.push_back(q);
      (which ? c->host_outer : c->host) = subset(H, sel);
    }
    const int rc = lineset_upload(c);
    if (rc) { sr_lineset_destroy(c); return rc; }
    sets[(size_t)level] = c;
  }
  *out = sets[(size_t)level];
  return SR_OK;
}

// ------------------------------------------------------------------------
// The multi-channel pass: all level spectra of a (P, T) row stack from ONE walk of the full line list (near field:
// sr_zones_mc_kernel, sr_wings_mc_kernel) + one far-only pass per level sub-lineset (the far field is linear per output
// spectrum: its translations and polynomials cost the same per level whoever sums them).
//   ctypes3 = 0: out [n_levels][2][n_rows][n_pts], the pair tables of sr_glevel_pairs_dev
//   ctypes3 = 1: out [n_levels][3][n_rows][n_pts], sp_emission | ind_emission | absorption (sr_gcoeff_levels_dev)
// Returns SR_ERR_UNSUPPORTED (and does nothing) where the route does not apply -- exact mode, counting passes, more
// channels than an LDS image holds, 80-byte records -- the callers then run one coefficient op per level.
// spect_main_module.py:1122-1168 (add_PT per level), spect_classes.py:1304-1321 (which lines a level owns).
// ------------------------------------------------------------------------
static int mc_pass(sr_lineset *ls, const sr_layers_desc *atm, int64_t g_lo, int64_t g_hi, double *out, void *stream, int ctypes3) {
#if !SR_FASTREC64
  return SR_ERR_UNSUPPORTED;
#else
  const int nlev = ls->n_levels, n_rows = atm->n_layers;
  const int far_mode = g_far_field.load();
  if (nlev <= 0 || far_mode == 0 || g_counting.load() != 0 || g_level_route.load() == 0) return SR_ERR_UNSUPPORTED;
  if (!atm->temps || !atm->press) return SR_ERR_ARG;
  if (g_lo < 0 || g_hi > ls->gp.n_grid || g_lo >= g_hi) return SR_ERR_ARG;
  for (int k = 0; k < n_rows; ++k)
    if (!(atm->temps[k] > 0.0) || !(atm->press[k] >= 0.0)) return SR_ERR_ARG;
  McChannels mc;
  mc.stride = ctypes3 ? 3 : 2;
  mc.o_lo = ctypes3 ? 2 : 0;
  mc.o_up_e = ctypes3 ? 0 : 1;
  mc.o_up_a = ctypes3 ? 1 : 0;
  mc.n_ch = mc.stride * nlev;
  if (mc.n_ch > 1023) return SR_ERR_UNSUPPORTED; // (10 bits per channel in the zones kernel's packed item word)
  if (zones_mc_image(mc.n_ch) == 0 || wings_mc_lds(mc.n_ch) > (size_t)160 * 1024)
    return SR_ERR_UNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  CoefWork &w = *ls->work;
  McWork &m = ls->mc;
  const size_t n_pts = (size_t)(g_hi - g_lo);
  const int n_far = ctypes3 ? 2 * nlev : nlev;
  FarParams fp0;
  far_hierarchy(n_pts, 1, &fp0);
  const size_t coef_row = (size_t)fp0.n_boxes_total * 2 * kFC; // doubles per (far pass, row)
  // Row batches: the full list's records + every far pass's coefficients + what the largest level pass needs of the
  // shared CoefWork (two table sets, far-field scratch) stay under the table budget
  const size_t per_row = (size_t)std::max<int64_t>(ls->n_lines, 1) * (sizeof(FastRec) + sizeof(ColdRec)) * 4 +
                         sizeof(double) * coef_row * (size_t)(n_far + 2) + 4 * ((n_pts + 64 * kSrcPad + kHalf) / 64 + 16) * (size_t)kMomPerBox * sizeof(double);
  const int rows_max = (int)std::max<size_t>(1, g_table_budget.load() / per_row);
  const std::vector<double> bounds_all = ls->bounds_temps;
  if (!bounds_all.empty() && (int)bounds_all.size() != n_rows) {
    g_err = "sr_lineset_set_bounds_temps was given another number of layers than this call";
    return SR_ERR_ARG;
  }
  struct Restore {
    sr_lineset *ls; const std::vector<double> &all;
    ~Restore() { ls->bounds_temps = all; }
  } restore{ls, bounds_all};
  if (!m.zst) {
    HIPCHK(hipStreamCreateWithFlags(&m.zst, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&m.ev_prep, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&m.ev_zones, hipEventDisableTiming));
    for (auto &e : m.ev_t) HIPCHK(hipEventCreate(&e));
  }
  const bool timing = g_timing.load() != 0; // (the events of the LAST row batch stay readable: sr_last_level_tables_ms)
  m.timed = false;
  const int npop = nlev;
  for (int k0 = 0; k0 < n_rows; k0 += rows_max) {
    const int nl = std::min(rows_max, n_rows - k0);
    sr_layers_desc sub = *atm;
    sub.n_layers = nl;
    sub.temps = atm->temps + k0;
    sub.press = atm->press + k0;
    sub.q_part = atm->q_part ? atm->q_part + k0 : nullptr;
    sub.tvib = nullptr; // (no populations enter the level spectra)
    if (!bounds_all.empty()) ls->bounds_temps.assign(bounds_all.begin() + k0, bounds_all.begin() + k0 + nl);
    const bool frozen = !ls->bounds_temps.empty();
    // this batch after everything earlier on the handle (the previous batch's kernels read the tables refilled below)
    if (w.last_done_recorded) HIPCHK(hipStreamWaitEvent(st, w.ev_last_done, 0));
    const size_t hl_bytes = layer_stage_bytes(nl, npop);
    // (prepare() waits on the HOST for the mark behind the previous batch's -- or build's -- last kernel: nothing of this
    // batch is enqueued before the previous one has finished with the pass's scratch.  That is what keeps the far passes
    // below, whose internal streams are not ordered after the caller's, from writing m.d_coef under the wings kernel of
    // the batch before.)
    int rc = m.s_layers.prepare(hl_bytes);
    if (rc) return rc;
    rc = fill_layer_stage(ls, ls, ls, &sub, m.s_layers.host<double>());
    if (rc) return rc;
    rc = m.s_layers.push(hl_bytes, st);
    if (rc) return rc;
    const int *d_pm = nullptr;
    const LayersDev A = layers_dev_of(m.s_layers.d.as<double>(), nl, npop, frozen, ls->linear_weights, &d_pm);
    const int *zmax_dev = d_pm + 2 * nl;
    const auto lo_it = std::lower_bound(ls->ic.begin(), ls->ic.end(), (int)g_lo - (kHalf - 1));
    const auto hi_it = std::upper_bound(ls->ic.begin(), ls->ic.end(), (int)g_hi - 1 + kHalf);
    const int line_lo = (int)(lo_it - ls->ic.begin());
    const int n_sub = (int)(hi_it - lo_it);
    const IcIndex ix{ls->d_first.as<int>(), ls->first_x0, ls->first_n, line_lo, n_sub};
    auto chan_rows = [&](int c) { return out + ((size_t)c * n_rows + (size_t)k0) * n_pts; };
    if (n_sub <= 0) {
      for (int c = 0; c < mc.n_ch; ++c) HIPCHK(hipMemsetAsync(chan_rows(c), 0, sizeof(double) * n_pts * nl, st));
    } else {
      rc = m.d_fast.ensure(sizeof(FastRec) * ((size_t)n_sub * nl + 1));
      if (rc) return rc;
      rc = m.d_cold.ensure(sizeof(ColdRec) * ((size_t)n_sub * nl + 1));
      if (rc) return rc;
      rc = m.d_coef.ensure(sizeof(double) * coef_row * (size_t)nl * n_far);
      if (rc) return rc;
      if (timing) HIPCHK(hipEventRecord(m.ev_t[0], st));
      LAUNCHCHK(launch_prep(ls->L, A, ls->gp, WeightMode{kWeightChannels, ctypes3 ? 1 : 0}, line_lo, n_sub, (int)g_lo, (int)g_hi - 1,
                            m.d_fast.as<FastRec>(), m.d_cold.as<ColdRec>(), st));
      if (timing) HIPCHK(hipEventRecord(m.ev_t[1], st));
      HIPCHK(hipEventRecord(m.ev_prep, st)); // the layer stage, the tables and everything earlier on the caller's stream
      // The zones kernel beside the far passes, the wings kernel after both.  The build is bound by its kernels' WORK, not
      // their order -- 13.25 / 13.32 ms with the zones kernel gated behind the far passes, 13.06 / 13.13 beside them; the
      // sparse passes' single batched launch no longer starves beside it as their twelve small chains did
      // (profiles/r06_mc_timeline_v3 / v5).
      LAUNCHCHK(launch_zones_mc(m.d_fast.as<FastRec>(), m.d_cold.as<ColdRec>(), ls->L.lev_up + line_lo, ls->L.lev_lo + line_lo, ix,
                                zmax_dev, n_sub, nl, (int)g_lo, (int)g_hi, ls->gp, mc, out, n_rows, k0, st));
      if (timing) HIPCHK(hipEventRecord(m.ev_t[2], st));
      // far-only passes of the level sub-linesets
      rc = m.s_far.prepare(sizeof(McFarPass) * (size_t)n_far);
      if (rc) return rc;
      McFarPass *far = m.s_far.host<McFarPass>();
      // The passes are independent of each other: kMcFarLanes of them in flight side by side, each through a CoefWork
      // of its own (tables, streams, events) instead of the handle's shared one -- one after the other on one chain
      // stream the eleven sparse passes (0.33 ms each, latency-bound) and the dense ground-state pass were 7.7 ms, longer
      // than the zones kernel they run beside.  Largest sub-lineset first (its chain is the longest).
      if (!m.fw_init) {
        for (auto &fwk : m.fw) {
          rc = fwk.init();
          if (rc) return rc;
        }
        m.fw_init = true;
      }
      std::vector<sr_lineset *> child((size_t)n_far, nullptr);
      std::vector<int> order((size_t)n_far);
      for (int f = 0; f < n_far; ++f) {
        rc = level_set(ls, ctypes3 ? f / 2 : f, &child[(size_t)f], ctypes3 && (f & 1)); // ind_emission: the lines whose UPPER level is lv only
        if (rc) return rc;
        order[(size_t)f] = f;
      }
      std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return child[(size_t)x]->n_lines > child[(size_t)y]->n_lines; });
      auto pass_of = [&](int f, int *lv, bool *ind) { *lv = ctypes3 ? f / 2 : f; *ind = ctypes3 && (f & 1); };
      auto weights_of = [&](int lv, bool ind) {
        return !ctypes3 ? WeightMode{kWeightLevelPair, lv} : (ind ? WeightMode{kWeightGind, lv} : WeightMode{kWeightGabsGsp, lv});
      };
      // SPARSE sub-linesets (coef_op's rule: far-field mode 3, fewer than 0.35 lines per grid point -- per-line expansions
      // at every level, sr_farfield_rows_kernel): ALL of them in one batch of three launches -- tables, expansions, downward
      // pass -- on the zones stream, beside the dense passes' chains.  One coefficient op each, they were eleven launches
      // of 0.3 ms, latency-bound, each behind ~0.3 ms of host calls: 6 ms of a 13 ms build (gpurun_out/r06/tl_v4.txt).
      // They share this pass's layer stage: their own would hold the same numbers (margins of the parent).
      std::vector<int> dense;
      {
        rc = m.s_batch.prepare(sizeof(FarBatchItem) * (size_t)n_far);
        if (rc) return rc;
        FarBatchItem *items = m.s_batch.host<FarBatchItem>();
        int n_items = 0, max_sub = 0;
        size_t rec_total = 0;
        std::vector<size_t> rec_off;
        for (int q = 0; q < n_far; ++q) {
          const int f = order[(size_t)q];
          sr_lineset *c = child[(size_t)f];
          const bool sparse = far_mode == 3 && (double)c->n_lines < 0.35 * (double)c->gp.n_grid;
          if (!sparse) { dense.push_back(f); continue; }
          int lv; bool ind;
          pass_of(f, &lv, &ind);
          far[f].ch_a = !ctypes3 ? 2 * lv : (ind ? 3 * lv + 1 : 3 * lv + 2);
          far[f].ch_e = !ctypes3 ? 2 * lv + 1 : (ind ? -1 : 3 * lv);
          const auto lo_c = std::lower_bound(c->ic.begin(), c->ic.end(), (int)g_lo - (kHalf - 1));
          const auto hi_c = std::upper_bound(c->ic.begin(), c->ic.end(), (int)g_hi - 1 + kHalf);
          const int c_sub = (int)(hi_c - lo_c);
          if (c_sub <= 0) { far[f].coef = nullptr; continue; }
          FarBatchItem &it = items[n_items++];
          it.L = c->L;
          it.first = c->d_first.as<int>();
          it.first_x0 = c->first_x0;
          it.first_n = c->first_n;
          it.line_lo = (int)(lo_c - c->ic.begin());
          it.n_sub = c_sub;
          it.W = weights_of(lv, ind);
          it.coef = m.d_coef.as<double>() + coef_row * (size_t)nl * f;
          far[f].coef = it.coef;
          rec_off.push_back(rec_total);
          rec_total += (size_t)c_sub * nl;
          max_sub = std::max(max_sub, c_sub);
        }
        if (n_items > 0) {
          rc = m.d_bfast.ensure(sizeof(FastRec) * (rec_total + 1));
          if (rc) return rc;
          for (int i = 0; i < n_items; ++i) items[i].fast = m.d_bfast.as<FastRec>() + rec_off[(size_t)i];
          // (serial schedule, sr_set_overlap(0): everything on the caller's stream, one kernel after the other)
          hipStream_t bst = g_overlap.load() != 0 ? m.zst : st;
          if (bst != st) HIPCHK(hipStreamWaitEvent(bst, m.ev_prep, 0));
          rc = m.s_batch.push(sizeof(FarBatchItem) * (size_t)n_items, bst);
          if (rc) return rc;
          FarParams fpb;
          far_hierarchy(n_pts, nl, &fpb);
          fpb.pm = d_pm;
          fpb.coef = nullptr;
          fpb.m2l = 0; fpb.rows = 1; fpb.pm_src = d_pm + nl; fpb.disp_lo_end = 0; fpb.disp_hi_begin = 0; fpb.mom = nullptr; fpb.tab = nullptr;
          for (int lv = 0; lv < kMaxFarLevels; ++lv) fpb.n_src[lv] = fpb.src_off[lv] = 0;
          const double *l2l_tab = nullptr;
          rc = l2l_table_dev(&l2l_tab);
          if (rc) return rc;
          LAUNCHCHK(launch_far_batch(m.s_batch.d.as<FarBatchItem>(), n_items, max_sub, A, ls->gp, zmax_dev, (int)g_lo, fpb, l2l_tab, bst));
          HIPCHK(hipEventRecord(m.ev_zones, bst));
          rc = m.s_batch.mark(bst);
          if (rc) return rc;
        }
        m.batch_pending = n_items > 0;
      }
      // the dense passes (the ground state's, typically), each through a CoefWork of its own
      for (size_t q = 0; q < dense.size(); ++q) {
        const int f = dense[q];
        int lv; bool ind;
        pass_of(f, &lv, &ind);
        sr_lineset *c = child[(size_t)f];
        double *coef = m.d_coef.as<double>() + coef_row * (size_t)nl * f;
        bool has = false;
        CoefOpt o;
        o.far_coef = coef;
        o.far_has = &has;
        CoefWork *const shared = c->work;
        c->work = &m.fw[q % kMcFarLanes];
        rc = coef_op(c, &sub, g_lo, g_hi, nullptr, nullptr, stream, weights_of(lv, ind), o);
        c->work = shared;
        if (rc) return rc;
        far[f].coef = has ? coef : nullptr;
        far[f].ch_a = !ctypes3 ? 2 * lv : (ind ? 3 * lv + 1 : 3 * lv + 2);
        far[f].ch_e = !ctypes3 ? 2 * lv + 1 : (ind ? -1 : 3 * lv);
      }
      if (m.batch_pending) HIPCHK(hipStreamWaitEvent(st, m.ev_zones, 0));
      rc = m.s_far.push(sizeof(McFarPass) * (size_t)n_far, st);
      if (rc) return rc;
      FarParams fp;
      far_hierarchy(n_pts, nl, &fp);
      fp.pm = d_pm;
      fp.coef = nullptr;
      fp.m2l = 0; fp.rows = 0; fp.pm_src = d_pm + nl; fp.disp_lo_end = 0; fp.disp_hi_begin = 0; fp.mom = nullptr; fp.tab = nullptr;
      for (int lv = 0; lv < kMaxFarLevels; ++lv) fp.n_src[lv] = fp.src_off[lv] = 0;
      if (timing) HIPCHK(hipEventRecord(m.ev_t[3], st));
      LAUNCHCHK(launch_wings_mc(m.d_fast.as<FastRec>(), ls->L.lev_up + line_lo, ls->L.lev_lo + line_lo, ix, zmax_dev, n_sub, nl,
                                (int)g_lo, (int)g_hi, fp, mc, m.s_far.d.as<McFarPass>(), n_far, out, n_rows, k0, st));
      if (timing) {
        HIPCHK(hipEventRecord(m.ev_t[4], st));
        m.timed = true;
      }
      rc = m.s_far.mark(st);
      if (rc) return rc;
    }
    // lines whose centre lies outside their window (humliv_bb's outer branches): per level, as the per-level route
    if (ls->n_outer > 0) {
      // (one record buffer for the largest level: sized per level, a growing buffer was freed -- a device synchronisation --
      // between the levels' launches)
      rc = m.d_outer_recs.ensure(sizeof(OuterRec) * (size_t)ls->n_outer * nl);
      if (rc) return rc;
      for (int f = 0; f < n_far; ++f) {
        const int lv = ctypes3 ? f / 2 : f;
        const bool ind = ctypes3 && (f & 1);
        sr_lineset *c = nullptr;
        rc = level_set(ls, lv, &c, ind);
        if (rc) return rc;
        if (c->n_outer <= 0) continue;
        double *o_a, *o_e;
        if (!ctypes3) { o_a = chan_rows(2 * lv); o_e = chan_rows(2 * lv + 1); }
        else if (!ind) { o_a = chan_rows(3 * lv + 2); o_e = chan_rows(3 * lv); }
        else {
          rc = ls->d_gscratch.ensure(sizeof(double) * n_pts * nl); // the unused second channel of the ind_emission weights
          if (rc) return rc;
          o_a = chan_rows(3 * lv + 1); o_e = ls->d_gscratch.as<double>();
        }
        const WeightMode W = !ctypes3 ? WeightMode{kWeightLevelPair, lv} : (ind ? WeightMode{kWeightGind, lv} : WeightMode{kWeightGabsGsp, lv});
        LAUNCHCHK(launch_outer(c->Lo, c->n_outer, A, ls->gp, W, m.d_outer_recs.as<OuterRec>(), (int)g_lo, (int)g_hi, o_a, o_e, st));
      }
    }
    rc = m.s_layers.mark(st);
    if (rc) return rc;
    HIPCHK(hipEventRecord(w.ev_last_done, st));
    w.last_done_recorded = true;
  }
  return SR_OK;
#endif
}

extern "C" {

int sr_abscoeff_layers_dev(sr_lineset *ls, const sr_layers_desc *atm, int64_t g_lo, int64_t g_hi,
                           double *abs_out, double *emi_out, void *stream) {
  return coef_op(ls, atm, g_lo, g_hi, abs_out, emi_out, stream, WeightMode{kWeightFolded, 0});
}

int sr_gcoeff_layers_dev(sr_lineset *ls, const sr_layers_desc *atm, int level, int64_t g_lo, int64_t g_hi,
                         double *g_out, void *stream) {
  if (!ls || !atm || !g_out || atm->n_layers <= 0 || g_lo < 0 || g_lo >= g_hi) return SR_ERR_ARG;
  sr_lineset *c = nullptr;
  int rc = level_set(ls, level, &c);
  if (rc) return rc;
  const size_t plane = (size_t)atm->n_layers * (size_t)(g_hi - g_lo);
  rc = ls->d_gscratch.ensure(sizeof(double) * plane); // the parent's: one for all levels
  if (rc) return rc;
  // pass 1: abs channel = absorption (lines whose LOWER level is `level`), emi channel = sp_emission
  // (UPPER level); pass 2: abs channel = ind_emission (UPPER level), emi channel unused
  const int wl = ls->n_levels == 0 ? -1 : level; // no level table: lev_up = lev_lo = 0 for every line anyway
  rc = coef_op(c, atm, g_lo, g_hi, g_out + 2 * plane, g_out + 0 * plane, stream, WeightMode{kWeightGabsGsp, wl});
  if (rc) return rc;
  // ind_emission selects the lines whose UPPER level is `level`: the second pass runs on those alone (round 4; for
  // the ground level, which 80 % of a hot-band list have as their lower level, that is an empty set instead of
  // the whole list a second time)
  sr_lineset *cu = nullptr;
  rc = level_set(ls, level, &cu, true);
  if (rc) return rc;
  return coef_op(cu, atm, g_lo, g_hi, g_out + 1 * plane, ls->d_gscratch.as<double>(), stream,
                 WeightMode{kWeightGind, wl});
}

int sr_abscoeff_level_dev(sr_lineset *ls, const sr_layers_desc *atm, int level, int64_t g_lo, int64_t g_hi,
                          double *abs_out, double *emi_out, void *stream) {
  if (!ls) return SR_ERR_ARG;
  sr_lineset *c = nullptr;
  const int rc = level_set(ls, level, &c);
  if (rc) return rc;
  return coef_op(c, atm, g_lo, g_hi, abs_out, emi_out, stream, WeightMode{kWeightTracked, level});
}

int sr_glevel_pairs_dev(sr_lineset *ls, const sr_layers_desc *atm, int64_t g_lo, int64_t g_hi, double *out, void *stream) {
  if (!ls || !atm || !out || atm->n_layers <= 0 || g_lo < 0 || g_lo >= g_hi) return SR_ERR_ARG;
  const size_t plane = (size_t)atm->n_layers * (size_t)(g_hi - g_lo);
  if (ls->n_levels == 0) // the 'all' set: every line, pop = 1 / Q in the combine (smm:2052-2057)
    return coef_op(ls, atm, g_lo, g_hi, out, out + plane, stream, WeightMode{kWeightLevelPair, -1});
  {
    // every line once, its three weights to the spectra of its two levels (mc_pass); where that route does not apply
    // (exact mode, counting, sr_set_level_route(0)): one coefficient op per level, below
    const int rc = mc_pass(ls, atm, g_lo, g_hi, out, stream, 0);
    if (rc != SR_ERR_UNSUPPORTED) return rc;
  }
  for (int lv = 0; lv < ls->n_levels; ++lv) {
    sr_lineset *c = nullptr;
    int rc = level_set(ls, lv, &c); // lines whose upper or lower level is lv
    if (rc) return rc;
    rc = coef_op(c, atm, g_lo, g_hi, out + (size_t)(2 * lv) * plane, out + (size_t)(2 * lv + 1) * plane, stream,
                 WeightMode{kWeightLevelPair, lv});
    if (rc) return rc;
  }
  return SR_OK;
}

int sr_gcoeff_levels_dev(sr_lineset *ls, const sr_layers_desc *atm, int64_t g_lo, int64_t g_hi, double *g_out, void *stream) {
  if (!ls || !atm || !g_out || atm->n_layers <= 0 || g_lo < 0 || g_lo >= g_hi) return SR_ERR_ARG;
  if (ls->n_levels == 0) return sr_gcoeff_layers_dev(ls, atm, 0, g_lo, g_hi, g_out, stream);
  {
    const int rc = mc_pass(ls, atm, g_lo, g_hi, g_out, stream, 1);
    if (rc != SR_ERR_UNSUPPORTED) return rc;
  }
  const size_t plane3 = 3 * (size_t)atm->n_layers * (size_t)(g_hi - g_lo);
  for (int lv = 0; lv < ls->n_levels; ++lv) {
    const int rc = sr_gcoeff_layers_dev(ls, atm, lv, g_lo, g_hi, g_out + plane3 * (size_t)lv, stream);
    if (rc) return rc;
  }
  return SR_OK;
}

int sr_glevel_combine_dev(const double *tab, const double *tab_dT, int n_levels, int n_rows, int64_t n_pts, int n_steps,
                          const int32_t *step_row, const double *pop, const double *dpop, double inv_dT,
                          double *abs_out, double *emi_out, double *dabs_out, double *demi_out, void *stream) {
  if (!tab || !step_row || !pop || !abs_out || !emi_out || n_levels <= 0 || n_rows <= 0 || n_pts <= 0 || n_steps <= 0)
    return SR_ERR_ARG;
  if (n_levels > SR_MAX_LEVELS || n_pts > 2000000) return SR_ERR_LIMIT;
  if (tab_dT && (!dpop || !dabs_out || !demi_out || !(inv_dT == inv_dT))) return SR_ERR_ARG;
  for (int s = 0; s < n_steps; ++s)
    if (step_row[s] < 0 || step_row[s] >= n_rows) return SR_ERR_ARG; // would read out of the tables
  hipStream_t st = static_cast<hipStream_t>(stream);
  // the steps grouped by table row (counting sort: the order of the steps inside a row is the caller's)
  std::vector<int> count((size_t)n_rows + 1, 0);
  for (int s = 0; s < n_steps; ++s) ++count[(size_t)step_row[s] + 1];
  std::vector<int> rows_used, row_off(1, 0), start((size_t)n_rows, 0);
  for (int r = 0; r < n_rows; ++r)
    if (count[(size_t)r + 1] > 0) {
      start[(size_t)r] = row_off.back();
      rows_used.push_back(r);
      row_off.push_back(row_off.back() + count[(size_t)r + 1]);
    }
  std::vector<int> step_of((size_t)n_steps), fill(start);
  for (int s = 0; s < n_steps; ++s) step_of[(size_t)fill[(size_t)step_row[s]]++] = s;
  const int n_used = (int)rows_used.size();
  static thread_local Stager s_ring[4];
  static thread_local unsigned s_next = 0;
  Stager &sg = s_ring[s_next++ & 3];
  auto al = [](size_t v) { return (v + 15) / 16 * 16; };
  const size_t b_pop = sizeof(double) * (size_t)n_steps * n_levels;
  const size_t o_dpop = al(b_pop), o_used = al(o_dpop + (tab_dT ? b_pop : 0));
  const size_t o_off = al(o_used + sizeof(int) * (size_t)n_used), o_step = al(o_off + sizeof(int) * (size_t)(n_used + 1));
  const size_t total = al(o_step + sizeof(int) * (size_t)n_steps);
  int rc = sg.prepare(total);
  if (rc) return rc;
  char *h = sg.host<char>();
  std::memcpy(h, pop, b_pop);
  if (tab_dT) std::memcpy(h + o_dpop, dpop, b_pop);
  std::memcpy(h + o_used, rows_used.data(), sizeof(int) * (size_t)n_used);
  std::memcpy(h + o_off, row_off.data(), sizeof(int) * (size_t)(n_used + 1));
  std::memcpy(h + o_step, step_of.data(), sizeof(int) * (size_t)n_steps);
  rc = sg.push_early(total, st);
  if (rc) return rc;
  const char *d = sg.d.as<char>();
  LAUNCHCHK(launch_glevel_combine(tab, tab_dT, n_levels, n_rows, (int)n_pts, n_used, reinterpret_cast<const int *>(d + o_used),
                                  reinterpret_cast<const int *>(d + o_off), reinterpret_cast<const int *>(d + o_step),
                                  reinterpret_cast<const double *>(d), reinterpret_cast<const double *>(d + o_dpop), inv_dT,
                                  abs_out, emi_out, dabs_out, demi_out, st));
  return sg.mark(st);
}

int sr_abscoeff_layers(sr_lineset *ls, const sr_layers_desc *atm, int64_t g_lo, int64_t g_hi,
                       double *abs_out, double *emi_out) {
  if (!ls || !atm || !abs_out || !emi_out || g_lo >= g_hi || atm->n_layers <= 0) return SR_ERR_ARG;
  const size_t bytes = sizeof(double) * (size_t)(g_hi - g_lo) * atm->n_layers;
  DevBuf a, e;
  int rc = a.ensure(bytes);
  if (!rc) rc = e.ensure(bytes);
  if (!rc) rc = sr_abscoeff_layers_dev(ls, atm, g_lo, g_hi, a.as<double>(), e.as<double>(), nullptr);
  if (!rc) {
    hipError_t er = hipMemcpy(abs_out, a.p, bytes, hipMemcpyDeviceToHost);
    if (er == hipSuccess) er = hipMemcpy(emi_out, e.p, bytes, hipMemcpyDeviceToHost);
    if (er != hipSuccess) rc = hip_fail(er, "copy back");
  }
  a.release();
  e.release();
  return rc;
}

int sr_last_eval_counts(sr_lineset *ls, uint64_t *counts10) {
  uint64_t *counts8 = counts10;
  static_assert(kCntN == 10, "sr_last_eval_counts: the header documents ten counters");
  if (!ls || !counts8 || !ls->work->counted) return SR_ERR_ARG;
  HIPCHK(hipEventSynchronize(ls->work->ev_last_done));
  static_assert(sizeof(unsigned long long) == sizeof(uint64_t), "counter width");
  HIPCHK(hipMemcpy(counts8, ls->work->d_counts.p, sizeof(uint64_t) * kCntN, hipMemcpyDeviceToHost));
  return SR_OK;
}

int sr_last_kernel_ms(sr_lineset *ls, float *ms5) {
  if (!ls || !ls->work->timed || !ms5) return SR_ERR_ARG;
  HIPCHK(hipEventSynchronize(ls->work->ev[ls->work->n_timed]));
  for (int i = 0; i < 5; ++i) ms5[i] = 0.f;
  // ev[1] (end of prep) is on the prep stream when the call was pipelined: the kernels that follow
  // are measured from ev_op0, recorded on the caller's stream once it has the tables
  hipEvent_t first = ls->work->pipelined ? ls->work->ev_op0 : ls->work->ev[1];
  HIPCHK(hipEventElapsedTime(&ms5[0], ls->work->ev[0], ls->work->ev[1]));
  if (ls->work->overlapped) { // kernels run side by side: only the whole coefficient op has a duration
    HIPCHK(hipEventElapsedTime(&ms5[1], first, ls->work->ev[4]));
    return SR_OK;
  }
  for (int i = 1; i < ls->work->n_timed; ++i)
    HIPCHK(hipEventElapsedTime(&ms5[i], i == 1 ? first : ls->work->ev[i], ls->work->ev[i + 1]));
  return SR_OK;
}

int sr_last_level_tables_ms(sr_lineset *ls, float *ms4) {
  if (!ls || !ms4 || !ls->mc.timed) return SR_ERR_ARG;
  const McWork &m = ls->mc;
  HIPCHK(hipEventSynchronize(m.ev_t[4]));
  HIPCHK(hipEventElapsedTime(&ms4[0], m.ev_t[0], m.ev_t[1])); // tables of the full list
  HIPCHK(hipEventElapsedTime(&ms4[1], m.ev_t[1], m.ev_t[2])); // sr_zones_mc_kernel (beside the far passes unless the schedule is serial)
  HIPCHK(hipEventElapsedTime(&ms4[2], m.ev_t[2], m.ev_t[3])); // what of the far passes was left when it ended
  HIPCHK(hipEventElapsedTime(&ms4[3], m.ev_t[3], m.ev_t[4])); // sr_wings_mc_kernel
  return SR_OK;
}

// ------------------------------------------------------------------------
int sr_radiance_rays_dev(const double *abs_c, const double *emi_c, int n_layers, int64_t n_pts, int n_rays,
                         const int32_t *seg_off, const int32_t *seg_layer, const double *seg_col,
                         int init_from_rad, double *rad, void *stream) {
  if (!abs_c || !emi_c || !rad || !seg_off || n_layers <= 0 || n_pts <= 0 || n_rays <= 0) return SR_ERR_ARG;
  if (n_pts > 2000000) return SR_ERR_LIMIT;
  const int n_seg = seg_off[n_rays];
  if (seg_off[0] != 0 || n_seg < 0) return SR_ERR_ARG;
  if (n_seg > 0 && (!seg_layer || !seg_col)) return SR_ERR_ARG;
  for (int r = 0; r < n_rays; ++r)
    if (seg_off[r + 1] < seg_off[r]) return SR_ERR_ARG;
  for (int s = 0; s < n_seg; ++s)
    if (seg_layer[s] < 0 || seg_layer[s] >= n_layers) return SR_ERR_ARG; // would read out of bounds
  hipStream_t st = static_cast<hipStream_t>(stream);
  // a small ring of staging buffers: reusing one would make every call wait for the previous
  // call's copy, i.e. for the GPU to reach it, and keep the host at most one step ahead
  static thread_local Stager s_ring[4];
  static thread_local unsigned s_next = 0;
  Stager &s_seg = s_ring[s_next++ & 3];
  const size_t b_off = sizeof(int) * (size_t)(n_rays + 1), b_lay = sizeof(int) * (size_t)std::max(n_seg, 1);
  const size_t o_lay = (b_off + 15) / 16 * 16, o_col = (o_lay + b_lay + 15) / 16 * 16;
  const size_t total = o_col + sizeof(double) * (size_t)std::max(n_seg, 1);
  int rc = s_seg.prepare(total);
  if (rc) return rc;
  std::memcpy(s_seg.host<char>(), seg_off, b_off);
  if (n_seg > 0) {
    std::memcpy(s_seg.host<char>() + o_lay, seg_layer, sizeof(int) * (size_t)n_seg);
    std::memcpy(s_seg.host<char>() + o_col, seg_col, sizeof(double) * (size_t)n_seg);
  }
  rc = s_seg.push_early(total, st);
  if (rc) return rc;
  char *base = s_seg.d.as<char>();
  LAUNCHCHK(launch_radiance(abs_c, emi_c, (int)n_pts, n_rays, reinterpret_cast<const int *>(base),
                            reinterpret_cast<const int *>(base + o_lay),
                            reinterpret_cast<const double *>(base + o_col), init_from_rad, rad, st));
  return s_seg.mark(st); // the slot is refilled only after this kernel (the copy no longer rides on `st`)
}

// ------------------------------------------------------------------------
int sr_radiance_jac_layer_dev(const double *abs_c, const double *emi_c, const double *dabs, const double *demi,
                              int n_layers, int64_t n_pts, int n_rays, const int32_t *seg_off,
                              const int32_t *seg_layer, const double *seg_col, double *jac, void *stream) {
  if (!abs_c || !emi_c || !dabs || !demi || !jac || !seg_off || n_layers <= 0 || n_pts <= 0 || n_rays <= 0)
    return SR_ERR_ARG;
  if (n_pts > 2000000) return SR_ERR_LIMIT;
  const int n_seg = seg_off[n_rays];
  if (seg_off[0] != 0 || n_seg <= 0 || !seg_layer || !seg_col) return SR_ERR_ARG;
  for (int r = 0; r < n_rays; ++r)
    if (seg_off[r + 1] < seg_off[r]) return SR_ERR_ARG;
  for (int s = 0; s < n_seg; ++s)
    if (seg_layer[s] < 0 || seg_layer[s] >= n_layers) return SR_ERR_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  static thread_local Stager s_ring[4]; // as in sr_radiance_rays_dev
  static thread_local unsigned s_next = 0;
  Stager &s_seg = s_ring[s_next++ & 3];
  const size_t b_off = sizeof(int) * (size_t)(n_rays + 1), b_lay = sizeof(int) * (size_t)n_seg;
  const size_t o_lay = (b_off + 15) / 16 * 16, o_col = (o_lay + b_lay + 15) / 16 * 16;
  const size_t total = o_col + sizeof(double) * (size_t)n_seg;
  int rc = s_seg.prepare(total);
  if (rc) return rc;
  char *h = s_seg.host<char>();
  std::memcpy(h, seg_off, b_off);
  std::memcpy(h + o_lay, seg_layer, b_lay);
  std::memcpy(h + o_col, seg_col, sizeof(double) * (size_t)n_seg);
  rc = s_seg.push_early(total, st);
  if (rc) return rc;
  char *base = s_seg.d.as<char>();
  LAUNCHCHK(launch_radiance_jac_layer(abs_c, emi_c, dabs, demi, (int)n_pts, n_layers, n_rays,
                                      reinterpret_cast<const int *>(base),
                                      reinterpret_cast<const int *>(base + o_lay),
                                      reinterpret_cast<const double *>(base + o_col), jac, st));
  return s_seg.mark(st);
}

// ------------------------------------------------------------------------
int sr_radiance_jac_dev(const double *abs_c, const double *emi_c, int n_layers, int64_t n_pts, int n_rays,
                        const int32_t *seg_off, const int32_t *seg_layer, const double *seg_col,
                        const double *dcol_dpar, int n_par, double *rad, double *jac, void *stream) {
  if (!abs_c || !emi_c || !rad || !jac || !seg_off || !dcol_dpar || n_layers <= 0 || n_pts <= 0 || n_rays <= 0 ||
      n_par <= 0)
    return SR_ERR_ARG;
  if (n_pts > 2000000) return SR_ERR_LIMIT;
  const int n_seg = seg_off[n_rays];
  if (seg_off[0] != 0 || n_seg <= 0 || !seg_layer || !seg_col) return SR_ERR_ARG;
  for (int r = 0; r < n_rays; ++r)
    if (seg_off[r + 1] < seg_off[r]) return SR_ERR_ARG;
  for (int s = 0; s < n_seg; ++s)
    if (seg_layer[s] < 0 || seg_layer[s] >= n_layers) return SR_ERR_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  static thread_local Stager s_ring[4]; // as in sr_radiance_rays_dev
  static thread_local unsigned s_next = 0;
  Stager &s_seg = s_ring[s_next++ & 3];
  const size_t b_off = sizeof(int) * (size_t)(n_rays + 1), b_lay = sizeof(int) * (size_t)n_seg;
  const size_t o_lay = (b_off + 15) / 16 * 16, o_col = (o_lay + b_lay + 15) / 16 * 16;
  const size_t o_d = o_col + sizeof(double) * (size_t)n_seg;
  const size_t total = o_d + sizeof(double) * (size_t)n_seg * n_par;
  int rc = s_seg.prepare(total);
  if (rc) return rc;
  char *h = s_seg.host<char>();
  std::memcpy(h, seg_off, b_off);
  std::memcpy(h + o_lay, seg_layer, b_lay);
  std::memcpy(h + o_col, seg_col, sizeof(double) * (size_t)n_seg);
  std::memcpy(h + o_d, dcol_dpar, sizeof(double) * (size_t)n_seg * n_par);
  rc = s_seg.push_early(total, st);
  if (rc) return rc;
  char *base = s_seg.d.as<char>();
  LAUNCHCHK(launch_radiance_jac(abs_c, emi_c, (int)n_pts, n_rays, reinterpret_cast<const int *>(base),
                                reinterpret_cast<const int *>(base + o_lay),
                                reinterpret_cast<const double *>(base + o_col),
                                reinterpret_cast<const double *>(base + o_d), n_par, rad, jac, st));
  return s_seg.mark(st);
}

// ------------------------------------------------------------------------
// Device LOS pipeline
// ------------------------------------------------------------------------
} // extern "C"

namespace {

// One staged LOS description on the device (a slot of a small ring, see sr_radiance_rays_dev).
struct LosDev {
  const int *seg_off, *seg_layer, *pt_off, *par_gas;
  const double *x, *nd, *prof, *scale;
  double *col;  // [n_gas + n_par][n_seg]
  int n_seg, n_pt, n_prof;
  Stager *slot;
};

int check_los(const sr_los_desc *los, int n_layers, int *n_seg_out, int *n_pt_out) {
  if (!los || los->n_rays <= 0 || los->n_gas <= 0 || !los->seg_off || !los->x || !los->nd || !los->vmr) return SR_ERR_ARG;
  if (los->n_gas > 4) return SR_ERR_LIMIT;
  if (los->init_mode < 0 || los->init_mode > 2) return SR_ERR_ARG;
  if (los->init_mode == 2 && !(los->t_init > 0.0 && los->step > 0.0)) return SR_ERR_ARG;
  const int n_seg = los->seg_off[los->n_rays];
  if (los->seg_off[0] != 0 || n_seg <= 0 || !los->seg_layer || !los->pt_off) return SR_ERR_ARG;
  for (int r = 0; r < los->n_rays; ++r)
    if (los->seg_off[r + 1] < los->seg_off[r]) return SR_ERR_ARG;
  if (los->pt_off[0] != 0) return SR_ERR_ARG;
  for (int s = 0; s < n_seg; ++s) {
    if (n_layers > 0 && (los->seg_layer[s] < 0 || los->seg_layer[s] >= n_layers)) return SR_ERR_ARG; // out-of-bounds read
    const int np = los->pt_off[s + 1] - los->pt_off[s];
    if (np < 2) return SR_ERR_ARG;
    if (np > 8000) return SR_ERR_LIMIT; // imxstp, parameters.inc:64
  }
  *n_seg_out = n_seg;
  *n_pt_out = los->pt_off[n_seg];
  return SR_OK;
}

// Stage the LOS on `st` and evaluate the columns of the gases and of n_par profile parameters.
// own: the staging slot of a device-resident LOS (sr_los_create) instead of the per-call ring.
int stage_los(const sr_los_desc *los, int n_layers, int n_par, const int32_t *par_gas, const double *par_w,
              hipStream_t st, LosDev *out, Stager *own = nullptr) {
  int n_seg = 0, n_pt = 0;
  int rc = check_los(los, n_layers, &n_seg, &n_pt);
  if (rc) return rc;
  if (n_par < 0 || (n_par > 0 && (!par_gas || !par_w))) return SR_ERR_ARG;
  for (int p = 0; p < n_par; ++p)
    if (par_gas[p] < 0 || par_gas[p] >= los->n_gas) return SR_ERR_ARG;
  static thread_local Stager s_ring[4];
  static thread_local unsigned s_next = 0;
  Stager &sg = own ? *own : s_ring[s_next++ & 3];
  const int n_prof = los->n_gas + n_par, nr = los->n_rays;
  auto al = [](size_t v) { return (v + 15) / 16 * 16; };
  const size_t o_x = 0, o_nd = o_x + sizeof(double) * n_pt, o_prof = o_nd + sizeof(double) * n_pt;
  const size_t o_scale = o_prof + sizeof(double) * (size_t)n_prof * n_pt;
  const size_t o_soff = al(o_scale + sizeof(double) * n_prof), o_slay = al(o_soff + sizeof(int) * (nr + 1));
  const size_t o_poff = al(o_slay + sizeof(int) * n_seg), o_pgas = al(o_poff + sizeof(int) * (n_seg + 1));
  const size_t in_bytes = al(o_pgas + sizeof(int) * (size_t)std::max(n_par, 1));
  const size_t total = in_bytes + sizeof(double) * (size_t)n_prof * n_seg; // + the column table (device only)
  rc = sg.prepare(total);
  if (rc) return rc;
  char *h = sg.host<char>();
  std::memcpy(h + o_x, los->x, sizeof(double) * n_pt);
  std::memcpy(h + o_nd, los->nd, sizeof(double) * n_pt);
  std::memcpy(h + o_prof, los->vmr, sizeof(double) * (size_t)los->n_gas * n_pt);
  if (n_par) std::memcpy(h + o_prof + sizeof(double) * (size_t)los->n_gas * n_pt, par_w, sizeof(double) * (size_t)n_par * n_pt);
  double *sc = reinterpret_cast<double *>(h + o_scale);
  for (int g = 0; g < los->n_gas; ++g) sc[g] = los->col_scale ? los->col_scale[g] : 1.0;
  for (int p = 0; p < n_par; ++p) sc[los->n_gas + p] = sc[par_gas[p]];
  int *so = reinterpret_cast<int *>(h + o_soff), *sl = reinterpret_cast<int *>(h + o_slay);
  std::memcpy(so, los->seg_off, sizeof(int) * (nr + 1));
  if (los->los_order == 0) {
    std::memcpy(sl, los->seg_layer, sizeof(int) * n_seg);
  }
  // observer order: the recursion walks every ray's segments backwards.  Segment s keeps its index (and
  // its sample points, whose x must stay increasing); only the walk order changes, through a permutation
  // applied to seg_layer and to the column table's segment axis alike.
  std::vector<int> perm(n_seg);
  for (int r = 0; r < nr; ++r)
    for (int s = los->seg_off[r]; s < los->seg_off[r + 1]; ++s)
      perm[s] = los->los_order == 0 ? s : los->seg_off[r] + (los->seg_off[r + 1] - 1 - s);
  int *po = reinterpret_cast<int *>(h + o_poff);
  if (los->los_order != 0) {
    // re-list the segments in photon order: new segment q = old segment perm[q]; sample points re-packed
    std::vector<double> x2(n_pt), nd2(n_pt), pr2((size_t)n_prof * n_pt);
    const double *prs = reinterpret_cast<const double *>(h + o_prof);
    int at = 0;
    for (int q = 0; q < n_seg; ++q) {
      const int s = perm[q], a = los->pt_off[s], b = los->pt_off[s + 1];
      sl[q] = los->seg_layer[s];
      po[q] = at;
      for (int i = a; i < b; ++i, ++at) {
        x2[at] = los->x[i];
        nd2[at] = los->nd[i];
        for (int f = 0; f < n_prof; ++f) pr2[(size_t)f * n_pt + at] = prs[(size_t)f * n_pt + i];
      }
    }
    po[n_seg] = at;
    std::memcpy(h + o_x, x2.data(), sizeof(double) * n_pt);
    std::memcpy(h + o_nd, nd2.data(), sizeof(double) * n_pt);
    std::memcpy(h + o_prof, pr2.data(), sizeof(double) * (size_t)n_prof * n_pt);
  } else {
    std::memcpy(po, los->pt_off, sizeof(int) * (n_seg + 1));
  }
  if (n_par) std::memcpy(h + o_pgas, par_gas, sizeof(int) * n_par);
  hipStream_t cs = nullptr;
  rc = sg.begin_early(in_bytes, &cs);
  if (rc) return rc;
  char *d = sg.d.as<char>();
  out->x = reinterpret_cast<const double *>(d + o_x);
  out->nd = reinterpret_cast<const double *>(d + o_nd);
  out->prof = reinterpret_cast<const double *>(d + o_prof);
  out->scale = reinterpret_cast<const double *>(d + o_scale);
  out->seg_off = reinterpret_cast<const int *>(d + o_soff);
  out->seg_layer = reinterpret_cast<const int *>(d + o_slay);
  out->pt_off = reinterpret_cast<const int *>(d + o_poff);
  out->par_gas = reinterpret_cast<const int *>(d + o_pgas);
  out->col = reinterpret_cast<double *>(d + in_bytes);
  out->n_seg = n_seg;
  out->n_pt = n_pt;
  out->n_prof = n_prof;
  out->slot = &sg;
  LAUNCHCHK(launch_los_columns(out->nd, out->x, out->prof, out->scale, out->pt_off, n_seg, n_pt, n_prof, out->col, cs));
  return sg.end_early(cs, st);
}

LimbOpts limb_opts(const sr_los_desc *los, int n_seg) {
  LimbOpts o;
  o.n_gas = los->n_gas;
  o.n_seg_total = n_seg;
  o.solo_absorption = los->solo_absorption ? 1 : 0;
  o.init_mode = los->init_mode;
  o.g_lo = (int)los->g_lo;
  o.t_init = los->t_init;
  o.w0 = los->w0;
  o.gstep = los->step;
  return o;
}

} // namespace

namespace {

// The two sides of rays that walk the shells strictly inwards down to a turning shell and strictly outwards again (limb
// rays; slant / nadir rays are the outward half, their first segment counted as the inward one): far / near
// [n_rays][n_layers] = the segment's index in WALK order (LOS_order 'observer' reverses every ray, as stage_los lists the
// columns) or -1, and the range of shells touched.  false: some ray is not of that shape.
// row_of_seg (caller's segment order, or NULL: seg_layer): the SHELL of a segment where that is not its coefficient row
// (3-D paths: seg_jac_row).
bool fold_sides(const sr_los_desc *los, int n_layers, const int32_t *row_of_seg, std::vector<int> *far_out,
                std::vector<int> *near_out, int *l_min_out, int *l_max_out) {
  const int nr = los->n_rays;
  std::vector<int> &far = *far_out, &near = *near_out;
  far.assign((size_t)nr * n_layers, -1);
  near.assign((size_t)nr * n_layers, -1);
  int l_min = n_layers, l_max = -1;
  bool ok = true;
  for (int r = 0; r < nr && ok; ++r) {
    const int a = los->seg_off[r], m = los->seg_off[r + 1] - a;
    auto lay = [&](int q) { return (row_of_seg ? row_of_seg : los->seg_layer)[los->los_order == 0 ? a + q : a + (m - 1 - q)]; };
    int q = 0, prev = INT_MAX;
    for (; q < m; ++q) { // far side: strictly inwards
      const int k = lay(q);
      if (k >= prev) break;
      far[(size_t)r * n_layers + k] = a + q;
      prev = k;
      l_min = std::min(l_min, k); l_max = std::max(l_max, k);
    }
    prev = q < m ? lay(q) - 1 : prev;
    if (q < m && q > 0 && lay(q) < lay(q - 1)) ok = false;
    for (; q < m && ok; ++q) { // near side: strictly outwards (its first shell may be the far side's last)
      const int k = lay(q);
      if (k <= prev) { ok = false; break; }
      near[(size_t)r * n_layers + k] = a + q;
      prev = k;
      l_min = std::min(l_min, k); l_max = std::max(l_max, k);
    }
  }
  *l_min_out = l_min;
  *l_max_out = l_max;
  return ok && l_max >= l_min;
}

// The folded kernels' plan on the device: per ray and visited shell (outermost first) the layer and the ray's two
// segments there, + room for the packed records.  n_rec = 0: some ray is not V-shaped (nothing staged).
struct FoldStage {
  const int *plan = nullptr;
  FoldDense *rec = nullptr;
  int n_vis = 0, n_rec = 0;
  Stager *slot = nullptr;
};
int stage_fold(const sr_los_desc *los, int n_layers, hipStream_t st, FoldStage *out, Stager *own = nullptr) {
  std::vector<int> far, near;
  int l_min = 0, l_max = -1;
  if (!fold_sides(los, n_layers, nullptr, &far, &near, &l_min, &l_max)) return SR_OK;
  const int nr = los->n_rays;
  std::vector<int> shells;
  for (int k = l_max; k >= l_min; --k) {
    bool any = false;
    for (int r = 0; r < nr && !any; ++r) any = far[(size_t)r * n_layers + k] >= 0 || near[(size_t)r * n_layers + k] >= 0;
    if (any) shells.push_back(k);
  }
  if ((size_t)nr * shells.size() > ((size_t)1 << 21)) return SR_OK; // (2M records = 436 MB of plan: such batches keep the path-order kernels)
  // The fold pays when the rays SHARE shells.  A 3-D batch lists a row per LOS step (seg_layer = arange): every ray then
  // looks like a slant ray through rows of its own, the union of the shells is the steps of ALL rays and a ray's plan is
  // mostly empty visits (n_rays x n_seg_total records).  Such batches keep the path-order kernel: the union may be at
  // most twice the shells of the longest ray.
  size_t longest = 0;
  for (int r = 0; r < nr; ++r) {
    size_t n_real = 0;
    for (int k : shells) n_real += far[(size_t)r * n_layers + k] >= 0 || near[(size_t)r * n_layers + k] >= 0;
    longest = std::max(longest, n_real);
  }
  if (shells.size() > 2 * longest) return SR_OK;
  const int n_vis = (int)shells.size(), n_rec = nr * n_vis;
  static thread_local Stager s_ring[4];
  static thread_local unsigned s_next = 0;
  Stager &sg = own ? *own : s_ring[s_next++ & 3];
  const size_t plan_bytes = (sizeof(int) * 4 * (size_t)n_rec + 15) / 16 * 16;
  int rc = sg.prepare(plan_bytes + fold_dense_bytes(n_rec));
  if (rc) return rc;
  int *pl = sg.host<int>();
  for (int r = 0; r < nr; ++r)
    for (int v = 0; v < n_vis; ++v) {
      int *q = pl + ((size_t)r * n_vis + v) * 4;
      q[0] = shells[v];
      q[1] = far[(size_t)r * n_layers + shells[v]];
      q[2] = near[(size_t)r * n_layers + shells[v]];
      q[3] = 0;
    }
  rc = sg.push_early(plan_bytes, st);
  if (rc) return rc;
  char *d = sg.d.as<char>();
  out->plan = reinterpret_cast<const int *>(d);
  out->rec = reinterpret_cast<FoldDense *>(d + plan_bytes);
  out->n_vis = n_vis;
  out->n_rec = n_rec;
  out->slot = &sg;
  return SR_OK;
}

} // namespace

extern "C" {

int sr_los_columns(const sr_los_desc *los, double *col_out) {
  if (!col_out) return SR_ERR_ARG;
  LosDev D;
  int rc = stage_los(los, 0, 0, nullptr, nullptr, nullptr, &D);
  if (rc) return rc;
  std::vector<double> tmp((size_t)los->n_gas * D.n_seg);
  HIPCHK(hipMemcpy(tmp.data(), D.col, sizeof(double) * tmp.size(), hipMemcpyDeviceToHost));
  // back to the caller's segment numbering
  for (int g = 0; g < los->n_gas; ++g)
    for (int r = 0; r < los->n_rays; ++r)
      for (int s = los->seg_off[r]; s < los->seg_off[r + 1]; ++s) {
        const int q = los->los_order == 0 ? s : los->seg_off[r] + (los->seg_off[r + 1] - 1 - s);
        col_out[(size_t)g * D.n_seg + s] = tmp[(size_t)g * D.n_seg + q];
      }
  return SR_OK;
}

static thread_local int g_last_limb_route = 0;
int sr_last_limb_route(void) { return g_last_limb_route; }

} // extern "C"

// A LOS batch resident on the device (sr_los_create): the staged description, its Curtis-Godson columns and -- where
// the rays share their shells -- the folded sweep's packed records, all made ONCE.  The reference computes a LOS's
// steps once too (los.calc_radtran_steps, spect_main_module.py:2746-2767) and runs radtran on them many times.
struct sr_los {
  Stager s_los, s_fold, s_vmr, s_x;
  LosDev D{};
  FoldStage F{};
  sr_los_desc opt{}; // the scalar options; its pointers are not kept
  int n_layers = 0, dev = -1;
  int n_par = 0;                 // column parameters staged with the batch (sr_los_create_par)
  std::vector<int32_t> par_gas;  // their gases (host copy: a kernel argument of the folded kernel)
  // The handle's device state (profiles, columns, packed records) is rewritten by set_vmr / refresh_columns /
  // retrieval_forward on whatever stream the caller passes: every entry point orders its stream after the handle's
  // last use (as coef_op does with ev_last_done), so calls on streams that are not ordered against each other are safe
  hipEvent_t ev_last = nullptr;
  bool last_recorded = false;
  hipEvent_t ev_k0 = nullptr, ev_k1 = nullptr; // around the recursion of the last forward model (sr_set_timing(2))
  bool k_timed = false;
};

// Entry of every call on a resident batch: the handle belongs to the device it was made on (its buffers, and the
// pinned stagers that would rebuild themselves on another current device and then copy into this one's memory).
static int los_enter(sr_los *h, hipStream_t st) {
  int cur = -1;
  HIPCHK(hipGetDevice(&cur));
  if (cur != h->dev) {
    g_err = "sr_los handle used on device " + std::to_string(cur) + ", made on device " + std::to_string(h->dev);
    return SR_ERR_ARG;
  }
  if (h->last_recorded) HIPCHK(hipStreamWaitEvent(st, h->ev_last, 0));
  return SR_OK;
}
static int los_leave(sr_los *h, hipStream_t st) {
  if (!h->ev_last) HIPCHK(hipEventCreateWithFlags(&h->ev_last, hipEventDisableTiming));
  HIPCHK(hipEventRecord(h->ev_last, st));
  h->last_recorded = true;
  return SR_OK;
}

// ------------------------------------------------------------------------
// The instrument step's per-thread device state (sr_hires_to_lowres_shard_dev, and sr_retrieval_forward_dev whose
// recursion kernel integrates the bands itself).
// The weight table depends on the grid window and the bands alone: kept while they (and the buffer) stay the same -- the
// instrument step of a retrieval iteration is then launches only, no upload (the weights kernel was 19 us of it).
// (the key held the buffer's ADDRESS; a call with more rays re-allocates the buffer -- the partial sums
// follow the table in it -- and the new block can come back at the old address with nothing in it.  The key is the
// buffer's generation now, and the device: the thread-local buffers follow a hipSetDevice())
struct LowresState {
  Stager s_bands, s_land; // the bands' centres / widths on their way up; the band spectra's pinned landing buffer
  DevBuf d_out;
  DevBuf d_weights;       // the bands' weight table [n_bands][n_pts] + point ranges, then the partial sums
  int64_t n_pts = -1, g_lo = 0;
  double w0 = 0, step = 0, n_sigma = 0;
  bool valid = false;
  unsigned gen = 0;
  int dev = -1;
  std::vector<double> bands;
};
static thread_local LowresState t_lowres;

// Arguments checked, buffers for n_rows spectra in place, the weight table valid on `st` (its kernel launched there when
// the key changed: *fresh).  fused: partial sums per 64-point slot (launch_fold_dense with the scratch).
static int lowres_prepare(int n_rows, int64_t n_pts, int64_t g_lo, double w0, double step, const double *centers_nm,
                          const double *widths_nm, int n_bands, double n_sigma, int out_units, bool fused, hipStream_t st,
                          bool *fresh) {
  if (!centers_nm || !widths_nm || n_rows <= 0 || n_pts < 2 || n_bands <= 0 || g_lo < 0) return SR_ERR_ARG;
  if (g_lo + n_pts > 2000000) return SR_ERR_LIMIT;
  if (!(step > 0.0) || !(w0 > 0.0) || !(n_sigma > 0.0) || out_units < 0 || out_units > 2) return SR_ERR_ARG;
  for (int b = 0; b < n_bands; ++b)
    if (!(widths_nm[b] > 0.0)) return SR_ERR_ARG;
  LowresState &L = t_lowres;
  const size_t nb = (size_t)n_bands;
  int cur_dev = 0;
  HIPCHK(hipGetDevice(&cur_dev));
  if (L.dev != cur_dev) { // buffers of another device: start over on this one
    L.d_out.release();
    L.d_weights.release();
    L.valid = false;
    L.dev = cur_dev;
  }
  int rc = L.d_out.ensure(sizeof(double) * nb * n_rows);
  if (rc) return rc;
  rc = L.d_weights.ensure(lowres_scratch_bytes((int)n_pts, n_bands, n_rows, fused));
  if (rc) return rc;
  const bool same = L.valid && L.gen == L.d_weights.gen && L.n_pts == n_pts && L.g_lo == g_lo && L.w0 == w0 && L.step == step &&
                    L.n_sigma == n_sigma && L.bands.size() == 2 * nb &&
                    std::memcmp(L.bands.data(), centers_nm, sizeof(double) * nb) == 0 &&
                    std::memcmp(L.bands.data() + nb, widths_nm, sizeof(double) * nb) == 0;
  *fresh = !same;
  if (same) return SR_OK;
  L.valid = false; // (until the launch below has been issued)
  rc = L.s_bands.prepare(sizeof(double) * 2 * nb);
  if (rc) return rc;
  std::memcpy(L.s_bands.host<double>(), centers_nm, sizeof(double) * nb);
  std::memcpy(L.s_bands.host<double>() + nb, widths_nm, sizeof(double) * nb);
  rc = L.s_bands.push(sizeof(double) * 2 * nb, st);
  if (rc) return rc;
  LAUNCHCHK(launch_lowres_weights((int)n_pts, (int)g_lo, w0, step, L.s_bands.d.as<double>(), L.s_bands.d.as<double>() + nb, n_bands,
                                  n_sigma, L.d_weights.p, st));
  L.n_pts = n_pts; L.g_lo = g_lo; L.w0 = w0; L.step = step; L.n_sigma = n_sigma;
  L.bands.assign(centers_nm, centers_nm + nb);
  L.bands.insert(L.bands.end(), widths_nm, widths_nm + nb);
  L.gen = L.d_weights.gen;
  L.valid = true; // (the callers synchronise `st` before they return: the table is complete before any later call)
  return SR_OK;
}

// d_out [n_rows][n_bands] to the host through a pinned buffer of the library's own (a copy into the caller's pageable
// memory is staged by the runtime); synchronises `st`
static int lowres_land(int n_rows, int n_bands, double *out_host, hipStream_t st) {
  LowresState &L = t_lowres;
  const size_t bytes = sizeof(double) * (size_t)n_bands * n_rows;
  const int rc = L.s_land.prepare(bytes);
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(L.s_land.h, L.d_out.p, bytes, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  std::memcpy(out_host, L.s_land.h, bytes);
  return SR_OK;
}

// the recursion of a resident LOS on `st`: launches only
static int limb_rays_los(const double *abs_c, const double *emi_c, int n_layers, int64_t n_pts, sr_los *h, int64_t g_lo,
                         double *rad, hipStream_t st) {
  const int rc0 = los_enter(h, st);
  if (rc0) return rc0;
  sr_los_desc o = h->opt;
  o.g_lo = g_lo;
  if (h->F.n_rec > 0 && !limb_launch_is_small((int)n_pts, o.n_rays) && g_jac_layer_forward.load() == 0) {
    LAUNCHCHK(launch_fold_fwd(h->F.plan, h->D.col, h->D.n_seg, h->F.n_rec, h->F.rec, abs_c, emi_c, (int)n_pts, n_layers, o.n_rays,
                              h->F.n_vis, limb_opts(&o, h->D.n_seg), rad, st, /*pack=*/false));
    g_last_limb_route = 2;
    return los_leave(h, st);
  }
  LAUNCHCHK(launch_limb(abs_c, emi_c, (int)n_pts, n_layers, o.n_rays, h->D.seg_off, h->D.seg_layer, h->D.col,
                        limb_opts(&o, h->D.n_seg), rad, st));
  g_last_limb_route = 1;
  return los_leave(h, st);
}

extern "C" {

int sr_los_create_par(const sr_los_desc *los, int n_layers, int n_par, const int32_t *par_gas, const double *par_w,
                      sr_los **out) {
  if (!out || n_layers <= 0 || n_par < 0) return SR_ERR_ARG;
  *out = nullptr;
  sr_los *h = new sr_los();
  int rc = stage_los(los, n_layers, n_par, par_gas, par_w, nullptr, &h->D, &h->s_los);
  if (!rc) rc = stage_fold(los, n_layers, nullptr, &h->F, &h->s_fold);
  if (!rc && n_par == 0 && h->F.n_rec > 0) { // (with parameters the records are packed per call: the columns change with the VMRs)
    LimbOpts o = limb_opts(los, h->D.n_seg);
    rc = launch_fold_fwd(h->F.plan, h->D.col, h->D.n_seg, h->F.n_rec, h->F.rec, nullptr, nullptr, 0, n_layers, los->n_rays,
                         h->F.n_vis, o, nullptr, nullptr) ? SR_ERR_HIP : SR_OK;
  }
  if (!rc && hipStreamSynchronize(nullptr) != hipSuccess) rc = SR_ERR_HIP; // (the null stream waited for the copy stream's work)
  if (rc) {
    h->s_los.release();
    h->s_fold.release();
    delete h;
    return rc;
  }
  h->opt = *los;
  h->opt.seg_off = h->opt.seg_layer = h->opt.pt_off = nullptr;
  h->opt.x = h->opt.nd = h->opt.vmr = h->opt.col_scale = nullptr;
  h->n_layers = n_layers;
  h->n_par = n_par;
  if (n_par > 0) h->par_gas.assign(par_gas, par_gas + n_par);
  (void)hipGetDevice(&h->dev);
  *out = h;
  return SR_OK;
}

int sr_los_create(const sr_los_desc *los, int n_layers, sr_los **out) {
  return sr_los_create_par(los, n_layers, 0, nullptr, nullptr, out);
}

int sr_los_destroy(sr_los *h) {
  if (!h) return SR_OK;
  h->s_los.release();
  h->s_fold.release();
  h->s_vmr.release();
  h->s_x.release();
  if (h->ev_last) (void)hipEventDestroy(h->ev_last);
  if (h->ev_k0) (void)hipEventDestroy(h->ev_k0);
  if (h->ev_k1) (void)hipEventDestroy(h->ev_k1);
  delete h;
  return SR_OK;
}

// New VMRs at the sample points of a resident batch (a retrieval iteration: paths, densities and parameter weights
// stay): one small copy and the column integration of the gases' profiles, on `stream`.
int sr_los_set_vmr(sr_los *h, const double *vmr, void *stream) {
  if (!h || !vmr) return SR_ERR_ARG;
  if (h->opt.los_order != 0) {
    g_err = "sr_los_set_vmr: batches in observer order re-list their sample points; build a new handle";
    return SR_ERR_UNSUPPORTED;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t bytes = sizeof(double) * (size_t)h->opt.n_gas * h->D.n_pt;
  int rc = los_enter(h, st);
  if (rc) return rc;
  rc = h->s_vmr.prepare(bytes); // (waits for the previous update's copy: the one pinned buffer is being refilled)
  if (rc) return rc;
  std::memcpy(h->s_vmr.host<char>(), vmr, bytes);
  // straight into the batch's profile table: [n_gas + n_par][n_pt], the gases first
  HIPCHK(hipMemcpyAsync(const_cast<double *>(h->D.prof), h->s_vmr.h, bytes, hipMemcpyHostToDevice, st));
  rc = h->s_vmr.mark(st);
  if (rc) return rc;
  LAUNCHCHK(launch_los_columns(h->D.nd, h->D.x, h->D.prof, h->D.scale, h->D.pt_off, h->D.n_seg, h->D.n_pt, h->opt.n_gas, h->D.col, st));
  if (h->n_par == 0 && h->F.n_rec > 0) { // the radiance route keeps packed records: repack them with the new columns
    LimbOpts o = limb_opts(&h->opt, h->D.n_seg);
    LAUNCHCHK(launch_fold_fwd(h->F.plan, h->D.col, h->D.n_seg, h->F.n_rec, h->F.rec, nullptr, nullptr, 0, h->n_layers, h->opt.n_rays,
                              h->F.n_vis, o, nullptr, st));
  }
  return los_leave(h, st);
}

// The Curtis-Godson columns of a resident batch integrated again from its staged sample points (and the folded
// records repacked): two launches on `stream`, nothing copied.  For callers whose step is "columns + recursion" by
// definition (bench.py: the timed step keeps the column integration on the device, it only no longer re-stages a
// batch that has not changed).
int sr_los_refresh_columns(sr_los *h, void *stream) {
  if (!h) return SR_ERR_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int rc0 = los_enter(h, st);
  if (rc0) return rc0;
  LAUNCHCHK(launch_los_columns(h->D.nd, h->D.x, h->D.prof, h->D.scale, h->D.pt_off, h->D.n_seg, h->D.n_pt, h->D.n_prof, h->D.col, st));
  if (h->n_par == 0 && h->F.n_rec > 0) {
    LimbOpts o = limb_opts(&h->opt, h->D.n_seg);
    LAUNCHCHK(launch_fold_fwd(h->F.plan, h->D.col, h->D.n_seg, h->F.n_rec, h->F.rec, nullptr, nullptr, 0, h->n_layers, h->opt.n_rays,
                              h->F.n_vis, o, nullptr, st));
  }
  return los_leave(h, st);
}

// sr_limb_rays_jac_dev on a resident batch made with its column parameters (sr_los_create_par): launches only.
int sr_limb_rays_jac_los_dev(const double *abs_c, const double *emi_c, int n_layers, int64_t n_pts, sr_los *h, int64_t g_lo,
                             double *rad, double *jac, void *stream) {
  if (!abs_c || !emi_c || !rad || !jac || !h || n_layers != h->n_layers || n_pts <= 0 || h->n_par <= 0) return SR_ERR_ARG;
  if (n_pts > 2000000) return SR_ERR_LIMIT;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int rc0 = los_enter(h, st);
  if (rc0) return rc0;
  sr_los_desc o = h->opt;
  o.g_lo = g_lo;
  if (h->n_par <= kFoldDensePar && h->F.n_rec > 0 && g_jac_layer_forward.load() == 0) {
    LAUNCHCHK(launch_fold_dense(h->F.plan, h->D.col, h->par_gas.data(), h->n_par, h->D.n_seg, h->F.n_rec, h->F.rec, abs_c, emi_c,
                                (int)n_pts, n_layers, o.n_rays, h->F.n_vis, limb_opts(&o, h->D.n_seg), rad, jac, st));
    return los_leave(h, st);
  }
  LAUNCHCHK(launch_limb_jac(abs_c, emi_c, (int)n_pts, n_layers, o.n_rays, h->D.seg_off, h->D.seg_layer, h->D.col,
                            h->D.col + (size_t)o.n_gas * h->D.n_seg, h->D.par_gas, h->n_par, limb_opts(&o, h->D.n_seg), rad,
                            jac, st));
  return los_leave(h, st);
}

// The forward model of ONE retrieval iteration on a resident batch made with its parameters, in one call: parameter
// vector -> VMRs of the retrieved gases at the sample points (sr_los_vmr_from_params_kernel) -> Curtis-Godson columns
// -> radiances and parameter Jacobians of every ray (one launch) -> instrument bands (sr_hires_to_lowres_shard_dev's
// kernels, its cached band weights) -> one copy to the host -> the closed-form field-of-view integral of every pixel
// (three rays each; spect_main_module.py:3342-3374 in the closed form of the mirror's fov_closed_form, geometry factors
// from the caller).  What spect_main_module.py:2736-2940 does per iteration between add_clim and chicalc.
//   buf: device scratch [n_rays (1 + n_par)][n_pts] (radiances' rows first)
//   fov: [n_pix][7] = delta, delta^3, 2 dmax^2, edge, m2, esse, has_edge (0 / 1) per pixel, n_pix = n_rays / 3; or null: no
//        field of view, `out` then holds the rays themselves
//   out: host [n_pix or n_rays][1 + n_par][n_bands]
int sr_retrieval_forward_dev(const double *abs_c, const double *emi_c, int n_layers, int64_t n_pts, sr_los *h, int64_t g_lo,
                             const double *x, double w0, double step, const double *centers_nm, const double *widths_nm,
                             int n_bands, double n_sigma, int out_units, const double *fov, double *buf, double *out,
                             void *stream) {
  if (!h || !x || !buf || !out || h->n_par <= 0 || n_bands <= 0) return SR_ERR_ARG;
  if (h->opt.init_mode == 1) return SR_ERR_ARG; // (initial radiances from the output buffer: `buf` is scratch here)
  const int n_rays = h->opt.n_rays, n_par = h->n_par, n_row = 1 + n_par;
  if (fov && n_rays % 3 != 0) return SR_ERR_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  int rc = los_enter(h, st);
  if (rc) return rc;
  LAUNCHCHK(launch_los_vmr_from_params(const_cast<double *>(h->D.prof), h->opt.n_gas, n_par, h->D.n_pt, h->D.par_gas, x, st));
  LAUNCHCHK(launch_los_columns(h->D.nd, h->D.x, h->D.prof, h->D.scale, h->D.pt_off, h->D.n_seg, h->D.n_pt, h->opt.n_gas, h->D.col, st));
  static thread_local std::vector<double> low; // [n_rays + n_rays n_par][n_bands]
  low.resize((size_t)n_rays * n_row * n_bands);
  if (g_band_fusion.load() && n_par <= kFoldDensePar && h->F.n_rec > 0 && g_jac_layer_forward.load() == 0) {
    // the one-sweep kernel integrates the bands in its epilogue: no spectra are written (`buf` stays untouched)
    if (!abs_c || !emi_c || n_layers != h->n_layers || n_pts <= 0) return SR_ERR_ARG;
    if (n_pts > 2000000) return SR_ERR_LIMIT;
    bool fresh = false;
    rc = lowres_prepare(n_rays * n_row, n_pts, g_lo, w0, step, centers_nm, widths_nm, n_bands, n_sigma, out_units, /*fused=*/true, st,
                        &fresh);
    if (rc) return rc;
    sr_los_desc o = h->opt;
    o.g_lo = g_lo;
    const bool timed = g_timing.load() == 2;
    if (timed) {
      if (!h->ev_k0) {
        HIPCHK(hipEventCreate(&h->ev_k0));
        HIPCHK(hipEventCreate(&h->ev_k1));
      }
      HIPCHK(hipEventRecord(h->ev_k0, st));
    }
    LAUNCHCHK(launch_fold_dense(h->F.plan, h->D.col, h->par_gas.data(), n_par, h->D.n_seg, h->F.n_rec, h->F.rec, abs_c, emi_c,
                                (int)n_pts, n_layers, n_rays, h->F.n_vis, limb_opts(&o, h->D.n_seg), buf, buf + (size_t)n_rays * n_pts,
                                st, t_lowres.d_weights.p, n_bands));
    if (timed) HIPCHK(hipEventRecord(h->ev_k1, st));
    h->k_timed = timed;
    rc = los_leave(h, st);
    if (rc) return rc;
    // the sums go straight into the library's pinned landing buffer (16 KB over the bus from the kernel itself: no copy
    // command behind it)
    const size_t low_bytes = sizeof(double) * (size_t)n_rays * n_row * n_bands;
    rc = t_lowres.s_land.prepare(low_bytes);
    if (rc) return rc;
    LAUNCHCHK(launch_lowres_sum_blocks((int)n_pts, n_rays * n_row, n_bands, out_units, static_cast<double *>(t_lowres.s_land.h),
                                       t_lowres.d_weights.p, st));
    HIPCHK(hipStreamSynchronize(st));
    std::memcpy(low.data(), t_lowres.s_land.h, low_bytes);
  } else {
    rc = sr_limb_rays_jac_los_dev(abs_c, emi_c, n_layers, n_pts, h, g_lo, buf, buf + (size_t)n_rays * n_pts, stream);
    if (rc) return rc;
    rc = sr_hires_to_lowres_shard_dev(buf, n_rays * n_row, n_pts, g_lo, w0, step, centers_nm, widths_nm, n_bands, n_sigma, out_units,
                                      low.data(), stream);
    if (rc) return rc;
  }
  // row of (ray r, quantity q): q = 0 the radiance, q = 1 + p the derivative to parameter p
  auto row = [&](int r, int q) { return low.data() + (size_t)(q == 0 ? r : n_rays + r * n_par + (q - 1)) * n_bands; };
  if (!fov) {
    for (int r = 0; r < n_rays; ++r)
      for (int q = 0; q < n_row; ++q) std::memcpy(out + ((size_t)r * n_row + q) * n_bands, row(r, q), sizeof(double) * n_bands);
    return SR_OK;
  }
  for (int px = 0; px < n_rays / 3; ++px) {
    const double *f = fov + 7 * px;
    const double delta = f[0], delta3 = f[1], two_dmax2 = f[2], edge = f[3], m2 = f[4], esse = f[5];
    const bool has_edge = f[6] != 0.0;
    for (int q = 0; q < n_row; ++q) {
      const double *s0 = row(3 * px, q), *s1 = row(3 * px + 1, q), *s2 = row(3 * px + 2, q);
      double *o = out + ((size_t)px * n_row + q) * n_bands;
      for (int b = 0; b < n_bands; ++b) { // (the operations of fov_closed_form, in its order)
        const double c = (s0[b] + s2[b] - 2.0 * s1[b]) / two_dmax2;
        double total = 2.0 * (s1[b] * delta + c * delta3 / 3.0);
        if (has_edge) total = total + s1[b] * edge + 2.0 * c * m2 / edge;
        o[b] = esse * total;
      }
    }
  }
  return SR_OK;
}

// n x n systems of the optimal-estimation step, host fp64: LU with partial pivoting (what numpy.linalg.solve / inv do
// through LAPACK's dgesv, unblocked).  a: [n][n] row-major, destroyed; b: [n][m] right-hand sides, overwritten with
// the solutions.  false: singular.
static bool lu_solve(std::vector<double> &a, int n, std::vector<double> &b, int m) {
  for (int k = 0; k < n; ++k) {
    int piv = k;
    for (int i = k + 1; i < n; ++i)
      if (std::fabs(a[(size_t)i * n + k]) > std::fabs(a[(size_t)piv * n + k])) piv = i;
    if (a[(size_t)piv * n + k] == 0.0) return false;
    if (piv != k) {
      for (int j = 0; j < n; ++j) std::swap(a[(size_t)k * n + j], a[(size_t)piv * n + j]);
      for (int j = 0; j < m; ++j) std::swap(b[(size_t)k * m + j], b[(size_t)piv * m + j]);
    }
    for (int i = k + 1; i < n; ++i) {
      const double f = a[(size_t)i * n + k] / a[(size_t)k * n + k];
      if (f == 0.0) continue;
      for (int j = k + 1; j < n; ++j) a[(size_t)i * n + j] -= f * a[(size_t)k * n + j];
      for (int j = 0; j < m; ++j) b[(size_t)i * m + j] -= f * b[(size_t)k * m + j];
    }
  }
  for (int k = n - 1; k >= 0; --k)
    for (int j = 0; j < m; ++j) {
      double v = b[(size_t)k * m + j];
      for (int i = k + 1; i < n; ++i) v -= a[(size_t)k * n + i] * b[(size_t)i * m + j];
      b[(size_t)k * m + j] = v / a[(size_t)k * n + k];
    }
  return true;
}

int sr_retrieval_step_dev(const double *abs_c, const double *emi_c, int n_layers, int64_t n_pts, sr_los *h, int64_t g_lo,
                          const double *x, double w0, double step, const double *centers_nm, const double *widths_nm,
                          int n_bands, double n_sigma, int out_units, const double *fov, double *buf, double *out,
                          const sr_oe_desc *oe, double *chi_sum, int32_t *n_used, double *dx, double *s_x, double *avk,
                          void *stream) {
  if (!h || !oe || !oe->obs || !oe->noise || !oe->sa_inv || !oe->x_apriori || !chi_sum || !n_used || !dx || !s_x || !avk)
    return SR_ERR_ARG;
  const int n_rays = h->opt.n_rays, n_par = h->n_par, n_row = 1 + n_par;
  if (n_rays % 3 != 0 || n_par <= 0 || n_par > 64) return SR_ERR_ARG;
  const int n_pix = n_rays / 3;
  if (oe->n_obs != n_pix * n_bands) return SR_ERR_ARG;
  // forward model: with a field of view straight into `out`; without one the centre ray of every three
  static thread_local std::vector<double> rays;
  double *fwd = out;
  if (!fov) {
    rays.resize((size_t)n_rays * n_row * n_bands);
    fwd = rays.data();
  }
  const int rc = sr_retrieval_forward_dev(abs_c, emi_c, n_layers, n_pts, h, g_lo, x, w0, step, centers_nm, widths_nm, n_bands,
                                          n_sigma, out_units, fov, buf, fwd, stream);
  if (rc) return rc;
  if (!fov)
    for (int px = 0; px < n_pix; ++px)
      std::memcpy(out + (size_t)px * n_row * n_bands, fwd + (size_t)(3 * px + 1) * n_row * n_bands, sizeof(double) * n_row * n_bands);
  // the used observations: K [n_used][n_par] (build_jacobian: pixel-major, band-minor), y - F, 1 / sigma^2
  const int n_obs = oe->n_obs;
  std::vector<double> K, resid, w;
  K.reserve((size_t)n_obs * n_par);
  resid.reserve(n_obs);
  w.reserve(n_obs);
  double chi = 0.0;
  for (int px = 0; px < n_pix; ++px)
    for (int b = 0; b < n_bands; ++b) {
      const int i = px * n_bands + b;
      if (oe->mask && !oe->mask[i]) continue;
      const double *rowp = out + (size_t)px * n_row * n_bands;
      const double d = oe->obs[i] - rowp[b];
      const double q = d / oe->noise[i];
      chi += q * q;                                                   // chicalc, :3400-3410
      resid.push_back(d);
      w.push_back(1.0 / (oe->noise[i] * oe->noise[i]));
      for (int p = 0; p < n_par; ++p) K.push_back(rowp[(size_t)(1 + p) * n_bands + b]);
    }
  const int nu = (int)resid.size();
  *chi_sum = chi;
  *n_used = nu;
  // inversion_algebra (:3433-3469)
  const size_t np2 = (size_t)n_par * n_par;
  std::vector<double> G(np2, 0.0), rhs((size_t)n_par, 0.0);
  for (int i = 0; i < nu; ++i) {
    const double *k = K.data() + (size_t)i * n_par;
    for (int p = 0; p < n_par; ++p) {
      const double kw = k[p] * w[i];                                   // KtSy[p][i]
      rhs[p] += kw * resid[i];
      for (int q = 0; q < n_par; ++q) G[(size_t)p * n_par + q] += kw * k[q];
    }
  }
  std::vector<double> S_inv(np2), A(np2), I(np2, 0.0);
  for (size_t e = 0; e < np2; ++e) S_inv[e] = G[e] + oe->sa_inv[e];
  for (int p = 0; p < n_par; ++p) {
    double acc = 0.0;
    for (int q = 0; q < n_par; ++q) acc += oe->sa_inv[(size_t)p * n_par + q] * (oe->x_apriori[q] - x[q]);
    rhs[p] += acc;
    I[(size_t)p * n_par + p] = 1.0;
  }
  A = S_inv;
  if (!lu_solve(A, n_par, I, n_par)) return SR_ERR_TABLE;              // S_x = inv(S_inv)
  std::memcpy(s_x, I.data(), sizeof(double) * np2);
  for (int p = 0; p < n_par; ++p)
    for (int q = 0; q < n_par; ++q) {
      double acc = 0.0;
      for (int r = 0; r < n_par; ++r) acc += I[(size_t)p * n_par + r] * G[(size_t)r * n_par + q];
      avk[(size_t)p * n_par + q] = acc;                                // AVK = S_x G
    }
  A = S_inv;
  for (int p = 0; p < n_par; ++p) A[(size_t)p * n_par + p] += oe->lambda_lm * S_inv[(size_t)p * n_par + p];
  if (!lu_solve(A, n_par, rhs, 1)) return SR_ERR_TABLE;
  std::memcpy(dx, rhs.data(), sizeof(double) * (size_t)n_par);
  return SR_OK;
}

int sr_los_last_kernel_ms(sr_los *h, float *ms) {
  if (!h || !ms || !h->k_timed) return SR_ERR_ARG;
  HIPCHK(hipEventSynchronize(h->ev_k1));
  HIPCHK(hipEventElapsedTime(ms, h->ev_k0, h->ev_k1));
  return SR_OK;
}

// The retrieval loop itself (spect_main_module.py:2725-2987 with fixed coefficient spectra): sr_retrieval_step_dev per
// iteration, reduced chi square (:2949), stopping rule (:2960-2973), the update with the positivity rule (:616-624).
int sr_retrieval_loop_dev(const double *abs_c, const double *emi_c, int n_layers, int64_t n_pts, sr_los *h, int64_t g_lo,
                          const double *x0, double w0, double step, const double *centers_nm, const double *widths_nm,
                          int n_bands, double n_sigma, int out_units, const double *fov, double *buf, double *out,
                          const sr_oe_desc *oe, const sr_loop_desc *lp, double *chi_hist, double *x_hist, int32_t *n_it,
                          int32_t *stop, double *s_x, double *avk, void *stream) {
  if (!h || !x0 || !oe || !lp || !lp->positive || !chi_hist || !x_hist || !n_it || !stop || !s_x || !avk || lp->max_it < 0)
    return SR_ERR_ARG;
  const int n_par = h->n_par;
  if (n_par <= 0 || n_par > 64) return SR_ERR_ARG;
  const size_t np2 = (size_t)n_par * n_par;
  std::vector<double> x(x0, x0 + n_par), dx((size_t)n_par), sx(np2), av(np2);
  std::memcpy(x_hist, x.data(), sizeof(double) * (size_t)n_par);
  *n_it = 0;
  *stop = 0;
  double chi_old = 0.0;
  for (int it = 0; it < lp->max_it; ++it) {
    double chi_sum = 0.0;
    int32_t nu = 0;
    const int rc = sr_retrieval_step_dev(abs_c, emi_c, n_layers, n_pts, h, g_lo, x.data(), w0, step, centers_nm, widths_nm,
                                         n_bands, n_sigma, out_units, fov, buf, out, oe, &chi_sum, &nu, dx.data(), sx.data(),
                                         av.data(), stream);
    if (rc) return rc;
    const double chi = chi_sum / (double)(nu - lp->n_dof_par);
    chi_hist[it] = chi;
    *n_it = it + 1;
    if (it > 0) {
      if (std::fabs(chi - chi_old) / chi_old < lp->chi_threshold) { *stop = 1; break; }
      if (chi > chi_old) { *stop = 2; break; }
    }
    chi_old = chi;
    for (int p = 0; p < n_par; ++p) {
      double d = dx[p];
      if (lp->positive[p]) {
        if (!(x[p] > 0.0)) { // (the reference would halve for ever)
          g_err = "sr_retrieval_loop_dev: a positive-constrained parameter is not positive";
          return SR_ERR_ARG;
        }
        while (x[p] + d <= 0.0) d /= 2;
      }
      x[p] = x[p] + d;
    }
    std::memcpy(s_x, sx.data(), sizeof(double) * np2);
    std::memcpy(avk, av.data(), sizeof(double) * np2);
    std::memcpy(x_hist + (size_t)(it + 1) * n_par, x.data(), sizeof(double) * (size_t)n_par);
  }
  return SR_OK;
}

int sr_limb_rays_los_dev(const double *abs_c, const double *emi_c, int n_layers, int64_t n_pts, sr_los *h, int64_t g_lo,
                         double *rad, void *stream) {
  if (!abs_c || !emi_c || !rad || !h || n_layers != h->n_layers || n_pts <= 0) return SR_ERR_ARG;
  if (h->n_par != 0) return SR_ERR_ARG; // a batch made with parameters keeps no packed radiance records: sr_limb_rays_jac_los_dev
  if (n_pts > 2000000) return SR_ERR_LIMIT;
  return limb_rays_los(abs_c, emi_c, n_layers, n_pts, h, g_lo, rad, static_cast<hipStream_t>(stream));
}

int sr_limb_step_dev(sr_lineset *ls, const sr_layers_desc *atm, int64_t g_lo, int64_t g_hi, double *abs_out, double *emi_out,
                     sr_los *h, double *rad, void *stream) {
  if (!h || !rad || !atm || h->opt.n_gas != 1 || h->n_par != 0 || atm->n_layers != h->n_layers) return SR_ERR_ARG;
  const int rc = coef_op(ls, atm, g_lo, g_hi, abs_out, emi_out, stream, WeightMode{kWeightFolded, 0});
  if (rc) return rc;
  return limb_rays_los(abs_out, emi_out, atm->n_layers, g_hi - g_lo, h, g_lo, rad, static_cast<hipStream_t>(stream));
}

int sr_limb_rays_dev(const double *abs_c, const double *emi_c, int n_layers, int64_t n_pts, const sr_los_desc *los,
                     double *rad, void *stream) {
  if (!abs_c || !emi_c || !rad || n_layers <= 0 || n_pts <= 0) return SR_ERR_ARG;
  if (n_pts > 2000000) return SR_ERR_LIMIT;
  hipStream_t st = static_cast<hipStream_t>(stream);
  LosDev D;
  int rc = stage_los(los, n_layers, 0, nullptr, nullptr, st, &D);
  if (rc) return rc;
  // ray batches of a 1-D atmosphere: the folded sweep (a shell's loads and, the path being symmetric, its attenuation
  // once for the ray's two segments); small launches keep the split kernel, whose time is a chain's latency
  if (!limb_launch_is_small((int)n_pts, los->n_rays) && g_jac_layer_forward.load() == 0) {
    FoldStage F;
    rc = stage_fold(los, n_layers, st, &F);
    if (rc) return rc;
    if (F.n_rec > 0) {
      LAUNCHCHK(launch_fold_fwd(F.plan, D.col, D.n_seg, F.n_rec, F.rec, abs_c, emi_c, (int)n_pts, n_layers, los->n_rays, F.n_vis,
                                limb_opts(los, D.n_seg), rad, st));
      g_last_limb_route = 2;
      rc = F.slot->mark(st);
      if (rc) return rc;
      return D.slot->mark(st);
    }
  }
  LAUNCHCHK(launch_limb(abs_c, emi_c, (int)n_pts, n_layers, los->n_rays, D.seg_off, D.seg_layer, D.col,
                        limb_opts(los, D.n_seg), rad, st));
  g_last_limb_route = 1;
  return D.slot->mark(st);
}

} // extern "C"

namespace {

// Plan of the one-pass Jacobian kernel (sr_limb_adjoint_kernel): per segment, in the order the recursion walks
// them, the layer with its first-touch flag and the column parameters the segment touches (d col / d x_p != 0 iff
// the parameter's weight is non-zero at one of the segment's sample points), runs of consecutive touches carried
// in one of four registers; per ray the rows it never touches.  Returns false when a segment touches more than
// kAdjEnt parameters (the forward-sensitivity kernel takes those calls).
struct AdjPlan {
  std::vector<int> seg;      // [n_seg][kAdjPlanInts]
  std::vector<int> zero_off; // [n_rays + 1]
  std::vector<int> zero_row; // layer rows (< n_layers) and parameter rows (n_layers + p)
};
bool build_adj_plan(const sr_los_desc *los, int n_layers, const int32_t *seg_jrow, int n_jrows, int n_par,
                    const int32_t *par_gas, const double *par_w, bool want_layer, int n_pt, AdjPlan *out) {
  const int nr = los->n_rays, n_seg = los->seg_off[nr];
  out->seg.assign((size_t)n_seg * kAdjPlanInts, 0);
  out->zero_off.assign(nr + 1, 0);
  out->zero_row.clear();
  std::vector<char> lay_seen(n_jrows), par_seen(std::max(n_par, 1));
  std::vector<std::vector<int>> touch(n_par); // walk positions (within the ray) of the segments touching p
  for (int r = 0; r < nr; ++r) {
    const int a = los->seg_off[r], b = los->seg_off[r + 1], m = b - a;
    std::fill(lay_seen.begin(), lay_seen.end(), 0);
    auto orig = [&](int q) { return los->los_order == 0 ? a + q : a + (m - 1 - q); }; // walk position -> caller's segment
    for (int q = 0; q < m; ++q) {
      int *pl = &out->seg[(size_t)(a + q) * kAdjPlanInts];
      const int k = los->seg_layer[orig(q)], jr = seg_jrow ? seg_jrow[orig(q)] : k;
      pl[0] = k;
      pl[1] = lay_seen[jr] ? 0 : 1;
      pl[3] = jr;
      lay_seen[jr] = 1;
    }
    for (int p = 0; p < n_par; ++p) {
      touch[p].clear();
      const double *w = par_w + (size_t)p * n_pt;
      for (int q = 0; q < m; ++q) {
        const int s = orig(q);
        bool nz = false;
        for (int i = los->pt_off[s]; i < los->pt_off[s + 1] && !nz; ++i) nz = w[i] != 0.0;
        if (nz) touch[p].push_back(q);
      }
    }
    // entries per segment; runs of consecutive touches take a register while one is free
    bool slot_busy[4] = {false, false, false, false};
    std::vector<int> slot_of(std::max(n_par, 1), -1), run_left(std::max(n_par, 1), 0), pos(std::max(n_par, 1), 0);
    std::fill(par_seen.begin(), par_seen.end(), 0);
    for (int q = 0; q < m; ++q) {
      int *pl = &out->seg[(size_t)(a + q) * kAdjPlanInts];
      int n_ent = 0;
      for (int p = 0; p < n_par; ++p) {
        if (pos[p] >= (int)touch[p].size() || touch[p][pos[p]] != q) continue;
        if (n_ent == 4) return false;
        const bool next_too = pos[p] + 1 < (int)touch[p].size() && touch[p][pos[p] + 1] == q + 1;
        int fl = 0, sl = slot_of[p];
        if (sl < 0) { // no carry in: a run starts here
          fl |= 1;
          if (next_too)
            for (int c = 0; c < 4; ++c)
              if (!slot_busy[c]) { sl = c; slot_busy[c] = true; break; }
        }
        const bool carry_out = next_too && sl >= 0;
        if (!carry_out) {
          fl |= 2;                        // written here
          if (!par_seen[p]) fl |= 4;      // first write of this parameter in the ray: store
          par_seen[p] = 1;
          if (sl >= 0) slot_busy[sl] = false;
          slot_of[p] = -1;
        } else {
          slot_of[p] = sl;
        }
        pl[4 + n_ent] = p;
        pl[4 + 4 + n_ent] = par_gas[p] | (std::max(sl, 0) << 8) | (fl << 16);
        ++n_ent;
        ++pos[p];
      }
      pl[2] = n_ent;
    }
    if (want_layer)
      for (int k = 0; k < n_jrows; ++k)
        if (!lay_seen[k]) out->zero_row.push_back(k);
    for (int p = 0; p < n_par; ++p)
      if (!par_seen[p]) out->zero_row.push_back(n_jrows + p);
    out->zero_off[r + 1] = (int)out->zero_row.size();
  }
  return true;
}

// Radiances (rad may be NULL) + per-layer Jacobian (dabs / demi / jac_layer may be NULL) + column-parameter Jacobian
// (n_par may be 0) in one pass.  *done = 0 when the plan does not fit the kernel (nothing launched).
int limb_adjoint(const double *abs_c, const double *emi_c, const double *dabs, const double *demi, int n_layers,
                 int64_t n_pts, const sr_los_desc *los, const int32_t *seg_jrow, int n_jrows, int n_par,
                 const int32_t *par_gas, const double *par_w, double *rad, double *jac_layer, double *jac_par,
                 hipStream_t st, int *done) {
  *done = 0;
  int n_seg = 0, n_pt = 0;
  int rc = check_los(los, n_layers, &n_seg, &n_pt);
  if (rc) return rc;
  if (!rad && los->init_mode == 1) return SR_ERR_ARG; // limb_initial would read rad
  if (!seg_jrow) n_jrows = n_layers;
  if (seg_jrow)
    for (int q = 0; q < n_seg; ++q)
      if (seg_jrow[q] < 0 || seg_jrow[q] >= n_jrows) return SR_ERR_ARG; // would write out of the Jacobian
  AdjPlan plan;
  if (!build_adj_plan(los, n_layers, seg_jrow, n_jrows, n_par, par_gas, par_w, jac_layer != nullptr, n_pt, &plan)) return SR_OK;
  LosDev D;
  rc = stage_los(los, n_layers, n_par, par_gas, par_w, st, &D);
  if (rc) return rc;
  // Layer-synchronous schedule (sr_limb_adjoint_sync_kernel): when the rays share their coefficient rows (no per-step
  // rows: seg_jrow == NULL) and every ray walks them monotonically down to a turning shell and monotonically up again
  // (limb rays; slant / nadir rays are the upward half alone), the shells are listed once -- far side from the
  // outermost shell inwards, near side outwards -- with every ray's segment in each: the kernel then loads a shell's
  // coefficients once for kAdjSyncRays rays.  Anything else (3-D paths, unordered LOS) keeps one ray per thread.
  std::vector<int> sched, fplan;
  int n_visits = 0, n_fvis = 0, n_frec = 0;
  const int nr = los->n_rays, n_batches = (nr + kAdjSyncRays - 1) / kAdjSyncRays;
  const int jmode = g_jac_layer_forward.load();
  const bool two_rows = seg_jrow != nullptr;           // 3-D paths: a coefficient row per LOS step, shells = Jacobian rows
  const int n_sh = two_rows ? n_jrows : n_layers;      // shells
  const int fold_rays = two_rows ? 1 : kAdjFoldRays;   // rays per thread of the folded kernel
  if (jmode == 0 || (jmode == 3 && nr >= 2 && !two_rows)) {
    // per ray: far[shell] / near[shell] = walk-order segment index or -1
    std::vector<int> far, near;
    int l_min = n_sh, l_max = -1;
    const bool ok = fold_sides(los, n_sh, seg_jrow, &far, &near, &l_min, &l_max);
    // The folded plan (sr_limb_adjoint_fold_kernel, the default): one visit per shell, outermost first, with every
    // ray's far-side and near-side segment there; a column parameter's touches -- read off the per-segment plan -- are
    // planned over the VISITS (a level acts on the same shells on both sides: one run, one register, one store).
    if (ok && l_max >= l_min && jmode == 0) {
      std::vector<int> shells;
      for (int k = l_max; k >= l_min; --k) {
        bool any = false;
        for (int r = 0; r < nr && !any; ++r) any = far[(size_t)r * n_sh + k] >= 0 || near[(size_t)r * n_sh + k] >= 0;
        if (any) shells.push_back(k);
      }
      n_fvis = (int)shells.size();
      const int nb = (nr + fold_rays - 1) / fold_rays;
      n_frec = nb * n_fvis * fold_rays;
      fplan.assign((size_t)n_frec * kFoldPlanInts, 0);
      bool fits = true;
      std::vector<std::vector<int>> touch(n_par);
      std::vector<int> slot_of(std::max(n_par, 1)), pos(std::max(n_par, 1));
      std::vector<char> par_seen(std::max(n_par, 1));
      for (int rr = 0; rr < nb * fold_rays && fits; ++rr) {
        const int bt = rr / fold_rays, i = rr % fold_rays;
        auto rec_at = [&](int v) { return &fplan[(((size_t)bt * n_fvis + v) * fold_rays + i) * kFoldPlanInts]; };
        for (int v = 0; v < n_fvis; ++v) {
          int *pl = rec_at(v);
          pl[1] = rr < nr ? far[(size_t)rr * n_sh + shells[v]] : -1;
          pl[2] = rr < nr ? near[(size_t)rr * n_sh + shells[v]] : -1;
          // coefficient rows: the segments' own (a side without a segment takes the other's: its loads are not used)
          const int row_f = pl[1] >= 0 ? plan.seg[(size_t)pl[1] * kAdjPlanInts] : -1;
          const int row_n = pl[2] >= 0 ? plan.seg[(size_t)pl[2] * kAdjPlanInts] : -1;
          pl[0] = two_rows ? (row_f >= 0 ? row_f : std::max(row_n, 0)) : shells[v];
          pl[12] = two_rows ? (row_n >= 0 ? row_n : std::max(row_f, 0)) : shells[v];
          pl[13] = shells[v];
        }
        if (rr >= nr || n_par == 0) continue;
        for (int p = 0; p < n_par; ++p) touch[p].clear();
        for (int v = 0; v < n_fvis; ++v) {
          const int *pl = rec_at(v);
          for (int side = 1; side <= 2; ++side) {
            if (pl[side] < 0) continue;
            const int *sp = &plan.seg[(size_t)pl[side] * kAdjPlanInts];
            for (int e = 0; e < sp[2]; ++e)
              if (touch[sp[4 + e]].empty() || touch[sp[4 + e]].back() != v) touch[sp[4 + e]].push_back(v);
          }
        }
        bool slot_busy[4] = {false, false, false, false};
        std::fill(slot_of.begin(), slot_of.end(), -1);
        std::fill(pos.begin(), pos.end(), 0);
        std::fill(par_seen.begin(), par_seen.end(), 0);
        for (int v = 0; v < n_fvis && fits; ++v) {
          int *pl = rec_at(v);
          int n_ent = 0;
          for (int p = 0; p < n_par; ++p) {
            if (pos[p] >= (int)touch[p].size() || touch[p][pos[p]] != v) continue;
            if (n_ent == 4) { fits = false; break; }
            const bool next_too = pos[p] + 1 < (int)touch[p].size() && touch[p][pos[p] + 1] == v + 1;
            int fl = 0, sl = slot_of[p];
            if (sl < 0) { // a run starts here
              fl |= 1;
              if (next_too)
                for (int c = 0; c < 4; ++c)
                  if (!slot_busy[c]) { sl = c; slot_busy[c] = true; break; }
            }
            const bool carry_out = next_too && sl >= 0;
            if (!carry_out) {
              fl |= 2;
              if (!par_seen[p]) fl |= 4;
              par_seen[p] = 1;
              if (sl >= 0) slot_busy[sl] = false;
              slot_of[p] = -1;
            } else {
              slot_of[p] = sl;
            }
            pl[4 + n_ent] = p;
            pl[4 + 4 + n_ent] = par_gas[p] | (std::max(sl, 0) << 8) | (fl << 16);
            ++n_ent;
            ++pos[p];
          }
          pl[3] = n_ent;
        }
      }
      if (!fits) { // a shell's two segments touch more than four parameters between them: one ray per thread
        fplan.clear();
        n_fvis = n_frec = 0;
      }
    } else
    if (ok && l_max >= l_min && !two_rows) {
      // the visits every batch walks (a batch skips nothing: a visit none of its rays takes costs one load)
      std::vector<std::pair<int, int>> visits; // (side, layer)
      for (int k = l_max; k >= l_min; --k) {
        bool any = false;
        for (int r = 0; r < nr && !any; ++r) any = far[(size_t)r * n_layers + k] >= 0;
        if (any) visits.push_back({0, k});
      }
      for (int k = l_min; k <= l_max; ++k) {
        bool any = false;
        for (int r = 0; r < nr && !any; ++r) any = near[(size_t)r * n_layers + k] >= 0;
        if (any) visits.push_back({1, k});
      }
      n_visits = (int)visits.size();
      sched.assign((size_t)n_batches * n_visits * (1 + kAdjSyncRays), -1);
      for (int bt = 0; bt < n_batches; ++bt)
        for (int v = 0; v < n_visits; ++v) {
          int *sv = &sched[((size_t)bt * n_visits + v) * (1 + kAdjSyncRays)];
          sv[0] = visits[v].second;
          for (int i = 0; i < kAdjSyncRays; ++i) {
            const int r = bt * kAdjSyncRays + i;
            if (r < nr) sv[1 + i] = (visits[v].first ? near : far)[(size_t)r * n_layers + visits[v].second];
          }
        }
    }
  }
  static thread_local Stager s_ring[4];
  static thread_local unsigned s_next = 0;
  Stager &sg = s_ring[s_next++ & 3];
  auto al = [](size_t v) { return (v + 15) / 16 * 16; };
  const size_t b_plan = sizeof(int) * plan.seg.size(), o_zo = al(b_plan);
  const size_t o_zr = al(o_zo + sizeof(int) * plan.zero_off.size());
  const size_t o_sc = al(o_zr + sizeof(int) * std::max<size_t>(plan.zero_row.size(), 1));
  const size_t o_fp = al(o_sc + sizeof(int) * std::max<size_t>(sched.size(), 1));
  const size_t in_bytes = al(o_fp + sizeof(int) * std::max<size_t>(fplan.size(), 1));
  rc = sg.prepare(in_bytes + std::max(adj_prog_bytes(n_seg), fold_rec_bytes(n_frec)));
  if (rc) return rc;
  char *h = sg.host<char>();
  std::memcpy(h, plan.seg.data(), b_plan);
  std::memcpy(h + o_zo, plan.zero_off.data(), sizeof(int) * plan.zero_off.size());
  if (!plan.zero_row.empty()) std::memcpy(h + o_zr, plan.zero_row.data(), sizeof(int) * plan.zero_row.size());
  if (!sched.empty()) std::memcpy(h + o_sc, sched.data(), sizeof(int) * sched.size());
  if (!fplan.empty()) std::memcpy(h + o_fp, fplan.data(), sizeof(int) * fplan.size());
  rc = sg.push_early(in_bytes, st);
  if (rc) return rc;
  char *d = sg.d.as<char>();
  if (n_fvis > 0) {
    FoldRec *frec = reinterpret_cast<FoldRec *>(d + in_bytes);
    LAUNCHCHK(launch_fold_pack(reinterpret_cast<const int *>(d + o_fp), D.col, los->n_gas, n_seg, n_frec, frec, st));
    LAUNCHCHK(launch_limb_adjoint_fold(abs_c, emi_c, dabs, demi, (int)n_pts, n_layers, n_jrows, two_rows ? 1 : 0, nr, frec,
                                       reinterpret_cast<const int *>(d + o_zo), reinterpret_cast<const int *>(d + o_zr), n_par,
                                       limb_opts(los, D.n_seg), n_fvis, rad, jac_layer, jac_par, st));
    rc = sg.mark(st);
    if (rc) return rc;
    *done = 1;
    return D.slot->mark(st);
  }
  SegProg *prog = reinterpret_cast<SegProg *>(d + in_bytes);
  LAUNCHCHK(launch_adj_pack(reinterpret_cast<const int *>(d), D.col, los->n_gas, n_seg, prog, st));
  if (n_visits > 0)
    LAUNCHCHK(launch_limb_adjoint_sync(abs_c, emi_c, dabs, demi, (int)n_pts, n_layers, n_jrows, nr, prog,
                                       reinterpret_cast<const int *>(d + o_zo), reinterpret_cast<const int *>(d + o_zr), n_par,
                                       limb_opts(los, D.n_seg), reinterpret_cast<const int *>(d + o_sc), n_visits, rad,
                                       jac_layer, jac_par, st));
  else
  LAUNCHCHK(launch_limb_adjoint(abs_c, emi_c, dabs, demi, (int)n_pts, n_layers, n_jrows, los->n_rays, D.seg_off, prog,
                                reinterpret_cast<const int *>(d + o_zo), reinterpret_cast<const int *>(d + o_zr), n_par,
                                limb_opts(los, D.n_seg), rad, jac_layer, jac_par, st));
  rc = sg.mark(st);
  if (rc) return rc;
  *done = 1;
  return D.slot->mark(st);
}

} // namespace

extern "C" {

int sr_limb_rays_jacobians_dev(const double *abs_c, const double *emi_c, const double *dabs, const double *demi,
                               int n_layers, int64_t n_pts, const sr_los_desc *los, const int32_t *seg_jac_row,
                               int n_jac_rows, int n_par, const int32_t *par_gas, const double *par_w, double *rad,
                               double *jac_layer, double *jac_par, void *stream) {
  if (!abs_c || !emi_c || n_layers <= 0 || n_pts <= 0 || n_par < 0) return SR_ERR_ARG;
  if (seg_jac_row && n_jac_rows <= 0) return SR_ERR_ARG;
  if (n_pts > 2000000) return SR_ERR_LIMIT;
  const bool want_layer = jac_layer != nullptr, want_par = n_par > 0;
  if (want_layer != (dabs != nullptr && demi != nullptr) || (want_par && (!jac_par || !par_gas || !par_w))) return SR_ERR_ARG;
  if (!want_layer && !want_par) return SR_ERR_ARG;
  if (!rad && los && los->init_mode == 1) { // the initial intensity would be read from rad (a NULL read on the device)
    g_err = "sr_limb_rays_jacobians_dev: init_mode 1 reads the initial intensity from rad, which is NULL";
    return SR_ERR_ARG;
  }
  if (want_layer && los && los->init_mode == 1) {
    g_err = "per-layer Jacobians: init_mode 1 (intensity read from a buffer) is not supported, use 0 or 2";
    return SR_ERR_UNSUPPORTED;
  }
  if (want_par && los)
    for (int p = 0; p < n_par; ++p)
      if (par_gas[p] < 0 || par_gas[p] >= los->n_gas) return SR_ERR_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  int done = 0;
  if (g_jac_layer_forward.load() != 1) {
    const int rc = limb_adjoint(abs_c, emi_c, dabs, demi, n_layers, n_pts, los, seg_jac_row, n_jac_rows, n_par, par_gas,
                                par_w, rad, jac_layer, jac_par, st, &done);
    if (rc || done) return rc;
  }
  // forward-sensitivity kernels, one call per kind (a segment touching more than four parameters, or asked for)
  if (seg_jac_row && want_layer) {
    g_err = "sr_limb_rays_jacobians_dev: Jacobian rows other than the coefficient rows need the one-pass kernel "
            "(sr_set_jac_layer_mode(0) and at most four column parameters per segment)";
    return SR_ERR_UNSUPPORTED;
  }
  if (want_par) {
    double *rad_out = rad;
    static thread_local DevBuf scratch_rad; // the forward kernel writes the radiances too: somewhere, if not wanted
    if (!rad_out) {
      if (los && los->init_mode == 1) return SR_ERR_ARG; // an initial intensity is read from rad
      const int rc0 = scratch_rad.ensure(sizeof(double) * (size_t)los->n_rays * (size_t)n_pts);
      if (rc0) return rc0;
      rad_out = scratch_rad.as<double>();
    }
    const int rc = sr_limb_rays_jac_dev(abs_c, emi_c, n_layers, n_pts, los, n_par, par_gas, par_w, rad_out, jac_par, stream);
    if (rc) return rc;
  }
  if (want_layer) {
    const int rc = sr_limb_rays_jac_layer_dev(abs_c, emi_c, dabs, demi, n_layers, n_pts, los, jac_layer, stream);
    if (rc) return rc;
    if (rad && !want_par) return sr_limb_rays_dev(abs_c, emi_c, n_layers, n_pts, los, rad, stream);
  }
  return SR_OK;
}

int sr_limb_rays_jac_dev(const double *abs_c, const double *emi_c, int n_layers, int64_t n_pts,
                         const sr_los_desc *los, int n_par, const int32_t *par_gas, const double *par_w, double *rad,
                         double *jac, void *stream) {
  if (!abs_c || !emi_c || !rad || !jac || n_layers <= 0 || n_pts <= 0 || n_par <= 0) return SR_ERR_ARG;
  if (n_pts > 2000000) return SR_ERR_LIMIT;
  hipStream_t st = static_cast<hipStream_t>(stream);
  // many parameters: one pass per ray (the forward kernel repeats the recursion per 16 parameters); rad0 given
  // (init_mode 1) is read by both kernels before they write
  if (n_par > 8 && g_jac_layer_forward.load() != 1 && par_gas && par_w) {
    int done = 0;
    const int rc = limb_adjoint(abs_c, emi_c, nullptr, nullptr, n_layers, n_pts, los, nullptr, 0, n_par, par_gas, par_w,
                                rad, nullptr, jac, st, &done);
    if (rc || done) return rc;
  }
  LosDev D;
  int rc = stage_los(los, n_layers, n_par, par_gas, par_w, st, &D);
  if (rc) return rc;
  // Few parameters, 1-D limb / slant / nadir rays: the folded recursion with an accumulator per parameter
  // (sr_limb_fold_sens_lds_kernel) instead of the path-order forward sensitivities (which repeat the recursion per four parameters)
  if (n_par <= kFoldDensePar && g_jac_layer_forward.load() == 0) {
    FoldStage F;
    rc = stage_fold(los, n_layers, st, &F);
    if (rc) return rc;
    if (F.n_rec > 0) {
      LAUNCHCHK(launch_fold_dense(F.plan, D.col, par_gas, n_par, D.n_seg, F.n_rec, F.rec, abs_c, emi_c, (int)n_pts, n_layers,
                                  los->n_rays, F.n_vis, limb_opts(los, D.n_seg), rad, jac, st));
      rc = F.slot->mark(st);
      if (rc) return rc;
      return D.slot->mark(st);
    }
  }
  LAUNCHCHK(launch_limb_jac(abs_c, emi_c, (int)n_pts, n_layers, los->n_rays, D.seg_off, D.seg_layer, D.col,
                            D.col + (size_t)los->n_gas * D.n_seg, D.par_gas, n_par, limb_opts(los, D.n_seg), rad, jac,
                            st));
  return D.slot->mark(st);
}

int sr_limb_rays_jac_layer_dev(const double *abs_c, const double *emi_c, const double *dabs, const double *demi,
                               int n_layers, int64_t n_pts, const sr_los_desc *los, double *jac, void *stream) {
  if (!abs_c || !emi_c || !dabs || !demi || !jac || n_layers <= 0 || n_pts <= 0) return SR_ERR_ARG;
  if (n_pts > 2000000) return SR_ERR_LIMIT;
  if (los && los->init_mode == 1) {
    g_err = "sr_limb_rays_jac_layer_dev: init_mode 1 (intensity read from a buffer) is not supported, use 0 or 2";
    return SR_ERR_UNSUPPORTED;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (g_jac_layer_forward.load() != 1 && n_layers > 8) {
    int done = 0;
    const int rc = limb_adjoint(abs_c, emi_c, dabs, demi, n_layers, n_pts, los, nullptr, 0, 0, nullptr, nullptr, nullptr,
                                jac, nullptr, st, &done);
    if (rc || done) return rc;
  }
  LosDev D;
  int rc = stage_los(los, n_layers, 0, nullptr, nullptr, st, &D);
  if (rc) return rc;
  LAUNCHCHK(launch_limb_jac_layer(1, abs_c, emi_c, dabs, demi, (int)n_pts, n_layers, los->n_rays, D.seg_off, D.seg_layer,
                                  D.col, limb_opts(los, D.n_seg), jac, st));
  return D.slot->mark(st);
}

// ------------------------------------------------------------------------
int sr_lut_interp_dev(const double *g_tab, int n_pt, int64_t n_pts, int n_steps, const int32_t *idx4,
                      const double *wgt4, const double *pop, int combine, double *out_a, double *out_e,
                      void *stream) {
  if (!g_tab || !idx4 || !wgt4 || !out_a || n_pt <= 0 || n_pts <= 0 || n_steps <= 0) return SR_ERR_ARG;
  if (combine && (!pop || !out_e)) return SR_ERR_ARG;
  if (n_pts > 2000000) return SR_ERR_LIMIT;
  for (int s = 0; s < n_steps; ++s) {
    const int32_t *q = idx4 + 4 * s;
    const bool bil = q[2] >= 0;
    for (int i = 0; i < 4; ++i) {
      if (!bil && i >= 2) { if (q[i] >= 0) return SR_ERR_ARG; continue; }
      if (q[i] < 0 || q[i] >= n_pt) return SR_ERR_ARG; // would read out of the table
    }
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  static thread_local Stager s_ring[4];
  static thread_local unsigned s_next = 0;
  Stager &sg = s_ring[s_next++ & 3];
  const size_t b_w = sizeof(double) * 4 * (size_t)n_steps, b_p = sizeof(double) * (size_t)n_steps;
  const size_t b_i = sizeof(int) * 4 * (size_t)n_steps;
  int rc = sg.prepare(b_w + b_p + b_i);
  if (rc) return rc;
  char *h = sg.host<char>();
  std::memcpy(h, wgt4, b_w);
  if (pop) std::memcpy(h + b_w, pop, b_p); else std::memset(h + b_w, 0, b_p);
  std::memcpy(h + b_w + b_p, idx4, b_i);
  rc = sg.push_early(b_w + b_p + b_i, st);
  if (rc) return rc;
  char *d = sg.d.as<char>();
  LAUNCHCHK(launch_lut(combine, g_tab, n_pt, (int)n_pts, n_steps, reinterpret_cast<const int *>(d + b_w + b_p),
                       reinterpret_cast<const double *>(d), reinterpret_cast<const double *>(d + b_w), out_a, out_e,
                       st));
  return sg.mark(st);
}

// ------------------------------------------------------------------------
int sr_hires_to_lowres_dev(const double *rad, int n_rays, int64_t n_pts, double w0, double step,
                           const double *centers_nm, const double *widths_nm, int n_bands, double n_sigma,
                           int out_units, double *out_host, void *stream) {
  return sr_hires_to_lowres_shard_dev(rad, n_rays, n_pts, 0, w0, step, centers_nm, widths_nm, n_bands, n_sigma, out_units,
                                      out_host, stream);
}

int sr_hires_to_lowres_shard_dev(const double *rad, int n_rays, int64_t n_pts, int64_t g_lo, double w0, double step,
                                 const double *centers_nm, const double *widths_nm, int n_bands, double n_sigma,
                                 int out_units, double *out_host, void *stream) {
  if (!rad || !out_host) return SR_ERR_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  bool fresh = false;
  int rc = lowres_prepare(n_rays, n_pts, g_lo, w0, step, centers_nm, widths_nm, n_bands, n_sigma, out_units, /*fused=*/false, st, &fresh);
  if (rc) return rc;
  LowresState &L = t_lowres;
  LAUNCHCHK(launch_lowres(rad, (int)n_pts, (int)g_lo, n_rays, w0, step, nullptr, nullptr, n_bands, n_sigma, out_units,
                          L.d_out.as<double>(), L.d_weights.p, st, /*weights=*/false));
  return lowres_land(n_rays, n_bands, out_host, st);
}

// ------------------------------------------------------------------------
// f2py-shaped shims (host pointers)
// ------------------------------------------------------------------------
int sr_humliv_bb(const double *x, int n, int i1, int i2, double x0, double lw, double dw, double *y) {
  if (!x || !y || n < 2) return SR_ERR_ARG;
  if (i1 > i2 || i1 < 1 || i2 > n) return SR_ERR_ARG; // lineshape.f:253-256
  if (!(dw > 0.0)) return SR_ERR_ARG;                  // lineshape.f:260-264
  if (i2 - i1 + 1 > 32767) return SR_ERR_LIMIT;        // window indices are 16-bit (imxsig = 13010)
  if (i2 - i1 < 1) return SR_ERR_ARG;
  const int outer = !(x[i1 - 1] < x0 && x0 < x[i2 - 1]); // lineshape.f:272, 358
  DevBuf dx, dy;
  int rc = dx.ensure(sizeof(double) * n);
  if (!rc) rc = dy.ensure(sizeof(double) * n);
  if (rc) { dx.release(); dy.release(); return rc; }
  hipError_t e = hipMemcpy(dx.p, x, sizeof(double) * n, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemset(dy.p, 0, sizeof(double) * n);
  if (e == hipSuccess)
    e = (hipError_t)launch_humliv(dx.as<double>(), i1, i2, x0, lw, dw, dy.as<double>(), outer, nullptr);
  if (e == hipSuccess) e = hipMemcpy(y, dy.p, sizeof(double) * n, hipMemcpyDeviceToHost);
  dx.release();
  dy.release();
  return e == hipSuccess ? SR_OK : hip_fail(e, "sr_humliv_bb");
}

int sr_sum_all_lines(double *spe, int64_t n_spe, const double *rows, const int32_t *init, const int32_t *fin,
                     int n_lines, int row_len) {
  if (!spe || n_spe <= 0 || n_lines < 0 || row_len <= 0) return SR_ERR_ARG;
  if (n_lines > 0 && (!rows || !init || !fin)) return SR_ERR_ARG;
  if (n_spe > 2000000) return SR_ERR_LIMIT; // imxsig_long
  for (int l = 0; l < n_lines; ++l)           // the Fortran would write out of bounds
    if (init[l] < 1 || fin[l] > n_spe || fin[l] - init[l] + 1 > row_len) return SR_ERR_ARG;
  if (n_lines == 0) return SR_OK;
  DevBuf ds, dr, di;
  const size_t b_rows = sizeof(double) * (size_t)n_lines * row_len;
  int rc = ds.ensure(sizeof(double) * n_spe);
  if (!rc) rc = dr.ensure(b_rows);
  if (!rc) rc = di.ensure(sizeof(int) * 2 * (size_t)n_lines);
  hipError_t e = hipSuccess;
  if (!rc) {
    e = hipMemcpy(ds.p, spe, sizeof(double) * n_spe, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dr.p, rows, b_rows, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(di.p, init, sizeof(int) * n_lines, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(di.as<int>() + n_lines, fin, sizeof(int) * n_lines, hipMemcpyHostToDevice);
    if (e == hipSuccess)
      e = (hipError_t)launch_sum_lines(ds.as<double>(), (long)n_spe, dr.as<double>(), di.as<int>(),
                                       di.as<int>() + n_lines, n_lines, row_len, nullptr);
    if (e == hipSuccess) e = hipMemcpy(spe, ds.p, sizeof(double) * n_spe, hipMemcpyDeviceToHost);
  }
  ds.release(); dr.release(); di.release();
  if (rc) return rc;
  return e == hipSuccess ? SR_OK : hip_fail(e, "sr_sum_all_lines");
}

int sr_curgod(int which, const double *nd, const double *vmr, const double *f, const double *x,
              const int32_t *off, int n_seg, double *res) {
  if (which < 1 || which > 4 || !nd || !x || !off || !res || n_seg < 0) return SR_ERR_ARG;
  if (which >= 2 && !vmr) return SR_ERR_ARG;
  if (which >= 3 && !f) return SR_ERR_ARG;
  if (n_seg == 0) return SR_OK;
  if (off[0] != 0) return SR_ERR_ARG;
  for (int s = 0; s < n_seg; ++s) {
    if (off[s + 1] < off[s]) return SR_ERR_ARG;
    if (off[s + 1] - off[s] > 8000) return SR_ERR_LIMIT; // imxstp, parameters.inc:64
  }
  const size_t n = (size_t)off[n_seg];
  const int narr = which == 1 ? 2 : (which == 2 ? 3 : 4);
  DevBuf d, dofs, dres;
  int rc = d.ensure(sizeof(double) * std::max<size_t>(n, 1) * narr);
  if (!rc) rc = dofs.ensure(sizeof(int) * (size_t)(n_seg + 1));
  if (!rc) rc = dres.ensure(sizeof(double) * (size_t)n_seg);
  hipError_t e = hipSuccess;
  if (!rc) {
    double *b = d.as<double>();
    const double *srcs[4] = {nd, x, vmr, f};
    for (int a = 0; a < narr && e == hipSuccess; ++a)
      if (n) e = hipMemcpy(b + a * n, srcs[a], sizeof(double) * n, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dofs.p, off, sizeof(int) * (size_t)(n_seg + 1), hipMemcpyHostToDevice);
    if (e == hipSuccess)
      e = (hipError_t)launch_curgod(which, b, narr > 2 ? b + 2 * n : nullptr, narr > 3 ? b + 3 * n : nullptr,
                                    b + n, dofs.as<int>(), n_seg, dres.as<double>(), nullptr);
    if (e == hipSuccess) e = hipMemcpy(res, dres.p, sizeof(double) * (size_t)n_seg, hipMemcpyDeviceToHost);
  }
  d.release(); dofs.release(); dres.release();
  if (rc) return rc;
  return e == hipSuccess ? SR_OK : hip_fail(e, "sr_curgod");
}

} // extern "C"
