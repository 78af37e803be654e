// tehmm_hip.hip -- host side of libtehmm_hip.so: the C ABI declared in include/tehmm_hip.h.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared (see tehmm_amd/build.py).
#include <climits>
#include "tehmm_kernels.hip.h"
#include "tehmm_coop.hip.h"
#include "tehmm_lane.hip.h"
#include "tehmm_lane3.hip.h"
#include "tehmm_place.hip.h"
#include "tehmm_aux.hip.h"
#include "tehmm_fused.hip.h"
#include "tehmm_estep.hip.h"
#include "tehmm_wide_estep.hip.h"
#include "tehmm_wide.hip.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "../../include/tehmm_hip.h"

using namespace tehmm;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg) {
  g_err = msg;
  return code;
}

#define HIPCHK(expr)                                                                        \
  do {                                                                                      \
    hipError_t _e = (expr);                                                                 \
    if (_e != hipSuccess)                                                                   \
      return fail(TEHMM_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));        \
  } while (0)

// Device blocks of destroyed batches are kept for the next batch: a fresh batch per call (teHmmEval walks a genome in
// chunks) allocates the same ~17 GB of workspaces every time, and hipMalloc of memory the process has just given back
// costs up to half a second on some hosts (20 Mb batch: 24 ms of evaluation behind 500 ms of allocation).  Blocks of
// at least 1 MB are rounded to 2 MB multiples and cached PER DEVICE (a block is only ever handed out, synchronised or
// freed with its own device current) up to TEHMM_DEVICE_POOL_GB per device (default: an eighth of the device's memory,
// at most 48; 0 = off); a request takes the smallest cached block of its device that is not more than half again its
// size; when hipMalloc fails the device's cache is emptied and the request tried again.  Releasing waits for the
// block's device like hipFree does, so a block never changes hands under a running kernel.  The cache is invisible to
// other allocators in the process (torch): call tehmm_trim_pools() before handing the device to them.
struct DevBlock {
  void *p;
  size_t bytes;
  int dev;
};
struct DevPool {
  std::mutex mu;
  std::vector<DevBlock> free_blocks;
  std::vector<size_t> cached;              // bytes cached per device
  std::vector<size_t> cap;                 // per-device cap (0 = not yet known)
};
static DevPool &dev_pool() {
  static DevPool *p = new DevPool();       // (never destroyed: no hipFree behind the runtime's back at exit)
  return *p;
}
static int dev_current() {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess) { (void)hipGetLastError(); d = 0; }
  return d;
}
// cap of the CURRENT device's cache (needs the device for the default: a fraction of its memory)
static size_t dev_pool_cap() {
  static const double env_gb = [] {
    const char *s = std::getenv("TEHMM_DEVICE_POOL_GB");
    return s ? std::atof(s) : -1.0;
  }();
  if (env_gb == 0.0) return 0;
  if (env_gb > 0.0) return (size_t)(env_gb * 1073741824.0);
  const int d = dev_current();
  DevPool &dp = dev_pool();
  {
    std::lock_guard<std::mutex> lk(dp.mu);
    if ((size_t)d < dp.cap.size() && dp.cap[(size_t)d]) return dp.cap[(size_t)d];
  }
  size_t free_b = 0, total_b = 0;
  size_t cap = (size_t)48 << 30;
  if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b > 0) cap = std::min(cap, total_b / 8);
  else (void)hipGetLastError();
  std::lock_guard<std::mutex> lk(dp.mu);
  if (dp.cap.size() <= (size_t)d) dp.cap.resize((size_t)d + 1, 0);
  dp.cap[(size_t)d] = cap;
  return cap;
}
// (round 4: every block is pooled, small ones in 4 KB multiples -- hipFree waits for the whole device, and a batch
//  holds dozens of small tables: destroying one batch while another thread evaluated the next cost the evaluation's time)
constexpr size_t kDevPoolMin = 1, kDevPoolGran = (size_t)2 << 20, kDevPoolGranSmall = (size_t)4 << 10;
// gives back the cached blocks of one device (dev >= 0) or of all devices (dev < 0); the caller's device stays current
static size_t dev_pool_trim(int dev = -1) {
  DevPool &dp = dev_pool();
  std::vector<DevBlock> blocks, keep;
  {
    std::lock_guard<std::mutex> lk(dp.mu);
    for (auto &b : dp.free_blocks) (dev < 0 || b.dev == dev ? blocks : keep).push_back(b);
    dp.free_blocks.swap(keep);
    for (auto &b : blocks) dp.cached[(size_t)b.dev] -= b.bytes;
  }
  if (blocks.empty()) return 0;
  const int here = dev_current();
  int cur = here;
  size_t freed = 0;
  for (auto &b : blocks) {
    if (b.dev != cur) { (void)hipSetDevice(b.dev); cur = b.dev; }
    (void)hipFree(b.p);
    freed += b.bytes;
  }
  if (cur != here) (void)hipSetDevice(here);
  return freed;
}
static size_t dev_pool_cached(int dev) {
  DevPool &dp = dev_pool();
  std::lock_guard<std::mutex> lk(dp.mu);
  return (size_t)dev < dp.cached.size() ? dp.cached[(size_t)dev] : 0;
}
// *block_bytes = size of the block handed out (what dev_free must be told)
static hipError_t dev_alloc(void **out, size_t bytes, size_t *block_bytes) {
  const bool pooled = bytes >= kDevPoolMin && dev_pool_cap() > 0;
  const size_t gran = bytes >= ((size_t)1 << 20) ? kDevPoolGran : kDevPoolGranSmall;
  const size_t need = pooled ? (bytes + gran - 1) / gran * gran : bytes;
  const int dev = dev_current();
  if (pooled) {
    DevPool &dp = dev_pool();
    std::lock_guard<std::mutex> lk(dp.mu);
    size_t best = SIZE_MAX;
    for (size_t i = 0; i < dp.free_blocks.size(); ++i)
      if (dp.free_blocks[i].dev == dev && dp.free_blocks[i].bytes >= need && dp.free_blocks[i].bytes <= need + need / 2 &&
          (best == SIZE_MAX || dp.free_blocks[i].bytes < dp.free_blocks[best].bytes))
        best = i;
    if (best != SIZE_MAX) {
      *out = dp.free_blocks[best].p;
      *block_bytes = dp.free_blocks[best].bytes;
      dp.cached[(size_t)dev] -= dp.free_blocks[best].bytes;
      dp.free_blocks.erase(dp.free_blocks.begin() + (long)best);
      return hipSuccess;
    }
  }
  hipError_t e = hipMalloc(out, need);
  if (e != hipSuccess && dev_pool_cached(dev) > 0) {
    (void)hipGetLastError();
    (void)dev_pool_trim(dev);
    e = hipMalloc(out, need);
  }
  *block_bytes = e == hipSuccess ? need : 0;
  return e;
}
static thread_local bool t_streams_synced = false;
// dev = the device the block was allocated on (DBuf records it)
static void dev_free(void *p, size_t block_bytes, int dev) {
  if (!p) return;
  const int here = dev_current();
  if (dev != here) (void)hipSetDevice(dev);
  bool kept = false;
  if (block_bytes >= kDevPoolMin && dev_pool_cap() > 0) {
    // what hipFree would do: nothing on the block's device still uses it.  (A batch that is being destroyed has
    // synchronised its own streams -- the only ones that ever touch its blocks -- and says so: a device-wide wait
    // would hold this thread until every OTHER batch's kernels have finished, e.g. the next group of eval_stream.)
    if (!t_streams_synced) (void)hipDeviceSynchronize();
    const size_t cap = dev_pool_cap();
    DevPool &dp = dev_pool();
    std::lock_guard<std::mutex> lk(dp.mu);
    if (dp.cached.size() <= (size_t)dev) dp.cached.resize((size_t)dev + 1, 0);
    if (dp.cached[(size_t)dev] + block_bytes <= cap) {
      dp.free_blocks.push_back({p, block_bytes, dev});
      dp.cached[(size_t)dev] += block_bytes;
      kept = true;
    }
  }
  if (!kept) (void)hipFree(p);
  if (dev != here) (void)hipSetDevice(here);
}
// free device memory as the workspace decisions should see it: what the driver reports plus what the current device's
// pool holds (the workspaces these decisions are about are the pool's own clientele: a batch of the same shape takes
// the cached blocks back one for one, and an allocation that does not fit empties the cache before it fails)
static hipError_t dev_mem_info(size_t *free_b, size_t *total_b) {
  hipError_t e = hipMemGetInfo(free_b, total_b);
  if (e == hipSuccess) *free_b += dev_pool_cached(dev_current());
  return e;
}

// RAII device buffer
template <typename T>
struct DBuf {
  T *p = nullptr;
  size_t n = 0;
  size_t block = 0;            // bytes of the device block behind p (dev_alloc)
  int dev = 0;                 // device the block lives on
  DBuf() = default;
  DBuf(const DBuf &) = delete;
  DBuf &operator=(const DBuf &) = delete;
  ~DBuf() { release(); }
  void release() {
    if (p) dev_free(p, block, dev);
    p = nullptr;
    n = 0;
    cap = 0;
    block = 0;
  }
  hipError_t alloc(size_t count) {
    release();
    n = count;
    if (count == 0) return hipSuccess;
    void *q = nullptr;
    size_t blk = 0;
    hipError_t e = dev_alloc(&q, count * sizeof(T), &blk);
    if (e != hipSuccess) return e;
    p = (T *)q;
    block = blk;
    dev = dev_current();
    return hipSuccess;
  }
  hipError_t upload(const T *h, size_t count) {
    hipError_t e = alloc(count);
    if (e != hipSuccess || count == 0) return e;
    return hipMemcpy(p, h, count * sizeof(T), hipMemcpyHostToDevice);
  }
  // grow-only capacity (no hipFree / hipMalloc -- i.e. no device-wide synchronisation -- once warm)
  size_t cap = 0;
  hipError_t ensure(size_t count) {
    if (count <= cap && p) { n = count; return hipSuccess; }
    hipError_t e = alloc(count + count / 4 + 16);
    cap = e == hipSuccess ? n : 0;
    n = count;
    return e;
  }
  // Asynchronous fill on a stream; `h` must stay alive until the stream has been synchronised
  // (the callers keep such staging vectors in the batch).
  hipError_t fill_async(const T *h, size_t count, hipStream_t st) {
    hipError_t e = ensure(count);
    if (e != hipSuccess || count == 0) return e;
    return hipMemcpyAsync(p, h, count * sizeof(T), hipMemcpyHostToDevice, st);
  }
};

int grid_for(int64_t work, int block, int cap = 8192) {
  int64_t g = (work + block - 1) / block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

constexpr int kMaxStates = 128;
// segment ratios on the chunk-parallel Viterbi lane passes up to this many padded states (round 2: 36)
#ifndef TEHMM_RATIO_LANE_MAX
#define TEHMM_RATIO_LANE_MAX 64
#endif

// padded state count: the cooperative kernels are instantiated for these sizes
int pad_states(int N) {
  static const int sizes[] = {4, 8, 12, 16, 20, 24, 28, 32, 36, 40, 48, 56, 64};
  for (int s : sizes)
    if (N < s) return s;      // strictly greater: the last lane of the tile is always a pad lane
  return (N + 4) & ~3;
}

// Dynamic LDS above the default limit has to be requested per kernel function.
template <typename F>
void allow_lds(F *fn, size_t bytes) {
  (void)hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

}  // namespace

struct tehmm_model {
  int N = 0, NP = 0, K = 0, S = 0, R = 0;
  double normalize = 1.0;
  DBuf<double> lt, ltT, A, AT, pi, tab, ltab;
  DBuf<double> ltG, AG, ATG;   // output-group-major copies for the lane = item kernels (tehmm_lane.hip.h)
  DBuf<float> ltP;             // float pairs [o / 2][f][o % 2] for the packed P0 pass
  // emission tables of the fused lane passes (tehmm_fused.hip.h): [(R + 1)][4][KSP] + the LDS-staged copy
  DBuf<double> ptab, ptab_lds;
  DBuf<int> d_rowinfo;         // rowbase[K] | rowcnt[K] | ldsbase[K] on the device
  int KSP = 0;
  uint64_t uid = 0;            // unique per model handle (workspaces derived from the table layout are keyed on it)
  uint64_t version = 0;        // bumped by every M-step (what depends on the parameter VALUES is keyed on it too)
  bool ptab_log = false;       // log-domain rows (normalizeFac != 1): the product form does not apply
  int rowbase[TEHMM_MAX_TRACKS];
  int rowcnt[TEHMM_MAX_TRACKS];
  int ldsbase[TEHMM_MAX_TRACKS];
  int lds_rows = 0, lds_zero = 0;   // LDS-staged rows (incl. one zero row at index lds_zero)
  std::vector<double> h_lt;   // [N][N] host copy (diag etc.)
  // quantised transition tables of ALL binades TEHMM_SPEC_MIN_E .. TEHMM_PLACE_EMAX ([0] plain, [1] with the
  // segment-ratio header blocks), built once per parameter version (ensure_qtabs) for the device-side placement
  DBuf<double> qall[2];
  DBuf<int> qok[2];
  uint64_t q_version[2] = {~(uint64_t)0, ~(uint64_t)0};
};

// workspace of the fused E-step, kept with the batch so that EM iterations reuse it
struct EstepWork {
  DBuf<double> alpha, beta, wrows, fwd_lp;
  DBuf<double> statbuf;            // raw statistics of the host-array entry point (layout: tehmm_aux.hip.h)
  double *C = nullptr, *D = nullptr, *start = nullptr, *stat = nullptr;   // views into a statistics buffer
  DBuf<int> escale, dead, order, chunk_iv;
  DBuf<int64_t> grow0, chunk_t0;
  int64_t rows_cap = 0;
  int n_cap = 0, N = 0;
  // fused (chunk-parallel) E-step: the track groups of k_estep_reduce
  EstepGroups h_groups;
  DBuf<EstepGroups> d_groups;
  DBuf<double> part_xi, part_rt, part_lds;   // per-writer partial sums of the reductions (folded in order)
  uint64_t groups_model = 0;
};

// chunk bookkeeping of the speculative (chunk-parallel, exact) Viterbi
struct SpecWork {
  int CS = 0, n_chunks = 0, N = 0;
  std::vector<int> h_iv;
  std::vector<int64_t> h_t0, h_first;
  DBuf<int> iv, e, ok, stats, ntie, ties;
  DBuf<int64_t> t0, first;
  DBuf<double> gain, wmin, rows, tierows, scale, wstart, segmin, offend, clk, racc, rmn;
  DBuf<int64_t> rtarget, rsel;
  DBuf<unsigned> tsoft, tpar;
  DBuf<int> clink;
};

// item (sub-chunk) bookkeeping and item-interleaved buffers of the lane = item passes
struct LaneWork {
  int L = 0, CS = 0, NP = 0, n_items = 0, n_groups = 0;
  bool no_vlane = false;      // the fp64 log rows did not fit when the workspaces were sized
  std::vector<int> h_iv;
  std::vector<int64_t> h_t0, h_first;
  int64_t h_first_item(int id) const { return h_first[(size_t)id]; }
  DBuf<int> item_iv, ok_f, ok_b, link_f, link_b, runend_f, runstart_b;
  DBuf<double> glog_f, cpre_f, dl_f, lr_f, dl_b;
  DBuf<int64_t> item_t0, ifirst;
  DBuf<double> B, BH, MS, AL, BE, pre_f, end_f, pre_b, end_b, slog32;
  DBuf<double> chk, chkf;     // fused passes: speculative beta / alpha' rows at the chains' check positions
  DBuf<double> sink;          // fused backward pass: [tiles][64] where the masked elements of its posterior stores go
  DBuf<float> AL32;           // fused passes: alpha' rows as floats (al32_index)
  DBuf<float> GAM32, WZ32;    // fused E-step: gamma and wz rows, same layout (tehmm_estep.hip.h)
  DBuf<unsigned long long> rix;   // fused passes: observation rows as table-row index records (FusedTab::rixx)
  uint64_t rix_model = 0;
  int rix_L = 0, rix_Wu = 0;
  // forward / backward warm-up measured by k_fb_probe for (model, parameter version, item length)
  uint64_t wu_model = 0, wu_version = 0;
  int wu_L = 0, wu_val = 0;
  DBuf<int> probe_iv, probe_steps;
  DBuf<int64_t> probe_t0;
  // Viterbi lane passes
  DBuf<double> vpre, vend, vgain, vtierows, vpiecemin, qtabs;
  DBuf<float> B32;
  DBuf<VitChunks> d_vc;       // device copies of the argument tables of k_vit_lane
  DBuf<VitItems> d_vi;
  DBuf<int> vbad, vntie, vties, wk_g, wk_e, wk_items;
  DBuf<PlaceCounts> place;     // device-side placement (tehmm_place.hip.h)
  DBuf<int> gclass;
  VitChunks up_vc{};           // what d_vc / d_vi hold (uploaded again only when a pointer changed)
  VitItems up_vi{};
  bool up_valid = false;
  int soft_ties = 0;           // the last quantised pass went through rounding ties keeping its frame (tehmm_lane3.hip.h)
  // host staging of one evaluation: sources of asynchronous copies, alive until the call has
  // synchronised its streams
  std::vector<double> hs_gain, hs_cgain, hs_qt;
  std::vector<int> hs_e, hs_wkg, hs_wke, hs_wki;
  VitChunks hs_vc;
  VitItems hs_vi;
};

// item bookkeeping and buffers of the chunk-parallel posterior for 64 <= N <= 128 (tehmm_wide.hip.h)
struct WideWork {
  int L = 0, NPW = 0, n_items = 0, n_groups = 0, wu_ok = 0;
  uint64_t wu_model = 0, wu_version = 0;
  DBuf<int> item_iv, flags;
  DBuf<int64_t> item_t0, ifirst;
  DBuf<double> E, ms, pre_f, end_f, pre_b, end_b, SL, lr;
  DBuf<float> AL;
  // E-step on these passes (tehmm_wide_estep.hip.h): gamma / wz rows in the alpha' layout, the row table of the
  // gamma product, per-writer partial sums
  DBuf<float> GAM, WZ;
  bool al_zeroed = false;
  WideRows h_rows;
  DBuf<WideRows> d_rows;
  WideLds h_lds;
  DBuf<WideLds> d_lds;
  int lds_nsplit = 1;
  uint64_t rows_model = 0;
  int rows_npw = 0;
  DBuf<double> part_xi, part_rows, part_lds;
  DBuf<unsigned long long> d_rmax;
  double ratio_max = 0.0;          // largest segment ratio of the batch (0: not yet known)
  // chunk-parallel exact Viterbi (k_vit_wide_spec / k_vit_wide_fix)
  DBuf<double> BL;                 // log emission rows [internal position][128]
  DBuf<double> rows2;              // recorded rows of the second tie hypothesis
  DBuf<double> pre;                // [hypothesis][chunk][NP] vectors the quantised pass enters its chunks with
  DBuf<uint8_t> tb2;               // its traceback bytes
  DBuf<int64_t> sel_from;          // per chunk: first position the chain adopted from the quantised pass
  DBuf<int> sel_hyp;               // per chunk: which hypothesis it adopted (-1: none)
  DBuf<int> wk_c, wk_e;            // work lists: 8 chunks of one binade per workgroup
  std::vector<int> h_wkc, h_wke, h_e;
  std::vector<double> h_gain;
  // a posterior attempt in flight (posterior_wide_cp enqueued it, posterior_wide_finish checks its links)
  bool pp_active = false;
  int pp_Wu = 0;
  int *h_flags = nullptr;          // pinned: impossible rows / failed links of the attempt
  ~WideWork() { if (h_flags) (void)hipHostFree(h_flags); }
  DBuf<uint8_t> tb1;               // traceback bytes of the quantised pass, hypothesis 0 (tb2: hypothesis 1)
  DBuf<int> ready;                 // per chunk: the quantised pass is through with it (the chain polls, see k_vit_wide_fix)
  DBuf<int64_t> hstop;             // per interval: where its first speculated chunk starts (the head ends)
  DBuf<double> hvec;               // [n][NP] the chain's vector at hstop - 1 (phase 1 -> phase 2 of k_vit_wide_fix)
  DBuf<int> hflag;
  std::vector<int64_t> h_hstop;
};

struct tehmm_batch {
  EstepWork ew;
  WideWork ww;
  SpecWork sw;
  LaneWork lw;
  int n = 0, K = 0, KP = 0;
  int64_t total = 0;       // user rows
  int64_t total_pad = 0;   // internal positions (64-aligned per interval)
  bool has_ratios = false;
  std::vector<int64_t> h_off, h_pos0, h_len;
  std::vector<int> h_order;
  std::vector<int64_t> h_chunk0;
  int n_chunks = 0;
  DBuf<int64_t> d_out0, d_pos0, d_len, d_chunk0;
  DBuf<int> d_order, d_chunk_iv;
  DBuf<uint8_t> obs;
  DBuf<double> ratios;
  // results / workspace (allocated lazily for the model's N)
  int N = 0, NP = 0, TBW = 0;
  DBuf<int64_t> paths;
  DBuf<double> post;
  DBuf<uint8_t> tb;
  DBuf<uint8_t> G, bstate, Gg, tstate;     // chunk maps, boundary states; tile maps and tile states of the two-level scan
  DBuf<double> beta;
  DBuf<int> dead;
  DBuf<int> last_state;
  DBuf<double> vit_lp, fwd_lp;
  DBuf<int64_t> first_good;
  hipStream_t sV = nullptr, sP = nullptr, sB = nullptr;
  void *stage[2] = {nullptr, nullptr};     // pinned staging buffers of the D2H path (allocated on first use)
  hipEvent_t evX[2] = {nullptr, nullptr};
  hipEvent_t ev[16];
  int n_ev = 0;
  std::vector<double> h_fwd_lp;   // per-interval forward log-likelihood of the last estep / posterior evaluation
  std::vector<std::string> tnames;
  std::vector<std::pair<int, int>> tpairs;
  std::vector<double> tms;
};

// All tehmm_* functions below are declared extern "C" in include/tehmm_hip.h.

int tehmm_abi_version(void) { return 3; }
const char *tehmm_last_error(void) { return g_err.c_str(); }
int tehmm_max_states(void) { return kMaxStates; }

int tehmm_device_count(int *count) {
  if (!count) return fail(TEHMM_ERR_ARG, "count is NULL");
  hipError_t e = hipGetDeviceCount(count);
  if (e != hipSuccess) {
    *count = 0;
    return fail(TEHMM_ERR_HIP, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
  }
  return TEHMM_OK;
}

int tehmm_set_device(int device) {
  HIPCHK(hipSetDevice(device));
  return TEHMM_OK;
}

// ------------------------------------------------------------------------------------------
// array-level entry points
// ------------------------------------------------------------------------------------------
template <typename ObsT>
static int emission_impl(int64_t T, int K, int N, int S, const ObsT *obs, const double *logProbs,
                         double normalize, const double *segRatios, double *outProbs) {
  if (T < 0 || K <= 0 || N <= 0 || S <= 0 || !obs || !logProbs || !outProbs)
    return fail(TEHMM_ERR_ARG, "tehmm_emission: bad argument");
  if (T == 0) return TEHMM_OK;
  DBuf<ObsT> d_obs;
  DBuf<double> d_lp, d_r, d_out;
  DBuf<unsigned long long> d_fg;
  HIPCHK(d_obs.upload(obs, (size_t)T * K));
  HIPCHK(d_lp.upload(logProbs, (size_t)K * N * S));
  if (segRatios) HIPCHK(d_r.upload(segRatios, (size_t)T));
  HIPCHK(d_out.alloc((size_t)T * N));
  unsigned long long big = ~0ull;
  HIPCHK(d_fg.upload(&big, 1));
  int g = grid_for(T * N, 256);
  hipLaunchKernelGGL((k_emission<ObsT>), dim3(g), dim3(256), 0, 0, T, K, N, S, d_obs.p, d_lp.p,
                     normalize, d_r.p, d_out.p, d_fg.p);
  hipLaunchKernelGGL(k_emission_fix, dim3(g), dim3(256), 0, 0, T, N, d_out.p, d_fg.p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(outProbs, d_out.p, (size_t)T * N * sizeof(double), hipMemcpyDeviceToHost));
  return TEHMM_OK;
}

int tehmm_emission_u8(int64_t T, int K, int N, int S, const uint8_t *obs, const double *logProbs,
                      double normalize, const double *segRatios, double *outProbs) {
  return emission_impl<uint8_t>(T, K, N, S, obs, logProbs, normalize, segRatios, outProbs);
}
int tehmm_emission_u16(int64_t T, int K, int N, int S, const uint16_t *obs, const double *logProbs,
                       double normalize, const double *segRatios, double *outProbs) {
  return emission_impl<uint16_t>(T, K, N, S, obs, logProbs, normalize, segRatios, outProbs);
}
int tehmm_emission_i32(int64_t T, int K, int N, int S, const int32_t *obs, const double *logProbs,
                       double normalize, const double *segRatios, double *outProbs) {
  return emission_impl<int32_t>(T, K, N, S, obs, logProbs, normalize, segRatios, outProbs);
}

int tehmm_forward(int64_t T, int N, const double *pi, const double *lt, const double *frame,
                  const double *segRatios, double *fwd) {
  if (T < 0 || N <= 0 || !pi || !lt || !frame || !fwd)
    return fail(TEHMM_ERR_ARG, "tehmm_forward: bad argument");
  if (T == 0) return TEHMM_OK;
  DBuf<double> d_pi, d_lt, d_fr, d_r, d_out;
  HIPCHK(d_pi.upload(pi, N));
  HIPCHK(d_lt.upload(lt, (size_t)N * N));
  HIPCHK(d_fr.upload(frame, (size_t)T * N));
  if (segRatios) HIPCHK(d_r.upload(segRatios, (size_t)T));
  HIPCHK(d_out.alloc((size_t)T * N));
  hipLaunchKernelGGL(k_forward_log, dim3(1), dim3(64), 2 * N * sizeof(double), 0, T, N, d_pi.p,
                     d_lt.p, d_fr.p, d_r.p, d_out.p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(fwd, d_out.p, (size_t)T * N * sizeof(double), hipMemcpyDeviceToHost));
  return TEHMM_OK;
}

int tehmm_backward(int64_t T, int N, const double *pi, const double *lt, const double *frame,
                   const double *segRatios, double *bwd) {
  (void)pi;
  if (T < 0 || N <= 0 || !lt || !frame || !bwd)
    return fail(TEHMM_ERR_ARG, "tehmm_backward: bad argument");
  if (T == 0) return TEHMM_OK;
  DBuf<double> d_lt, d_fr, d_r, d_out;
  HIPCHK(d_lt.upload(lt, (size_t)N * N));
  HIPCHK(d_fr.upload(frame, (size_t)T * N));
  if (segRatios) HIPCHK(d_r.upload(segRatios, (size_t)T));
  HIPCHK(d_out.alloc((size_t)T * N));
  hipLaunchKernelGGL(k_backward_log, dim3(1), dim3(64), 3 * N * sizeof(double), 0, T, N, d_lt.p,
                     d_fr.p, d_r.p, d_out.p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(bwd, d_out.p, (size_t)T * N * sizeof(double), hipMemcpyDeviceToHost));
  return TEHMM_OK;
}

int tehmm_xi_logsum(int64_t T, int N, const double *fwd, const double *lt, const double *bwd,
                    const double *frame, double logprob, const double *segRatios, double *out) {
  if (T < 0 || N <= 0 || !fwd || !lt || !bwd || !frame || !out)
    return fail(TEHMM_ERR_ARG, "tehmm_xi_logsum: bad argument");
  DBuf<double> d_f, d_lt, d_b, d_fr, d_r, d_out;
  HIPCHK(d_f.upload(fwd, (size_t)T * N));
  HIPCHK(d_b.upload(bwd, (size_t)T * N));
  HIPCHK(d_fr.upload(frame, (size_t)T * N));
  HIPCHK(d_lt.upload(lt, (size_t)N * N));
  if (segRatios) HIPCHK(d_r.upload(segRatios, (size_t)T));
  HIPCHK(d_out.upload(out, (size_t)N * N));
  hipLaunchKernelGGL(k_xi_logsum, dim3(grid_for((int64_t)N * N, 64)), dim3(64), 0, 0, T, N, d_f.p,
                     d_lt.p, d_b.p, d_fr.p, logprob, d_r.p, d_out.p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(out, d_out.p, (size_t)N * N * sizeof(double), hipMemcpyDeviceToHost));
  return TEHMM_OK;
}

template <typename ObsT>
static int accumulate_impl(int64_t T, int K, int N, int S, const ObsT *obs, double *obsStats,
                           const double *post, const double *segRatios) {
  if (T < 0 || K <= 0 || N <= 0 || S <= 0 || !obs || !obsStats || !post)
    return fail(TEHMM_ERR_ARG, "tehmm_accumulate_obs: bad argument");
  if (T == 0) return TEHMM_OK;
  DBuf<ObsT> d_obs;
  DBuf<double> d_st, d_p, d_r;
  HIPCHK(d_obs.upload(obs, (size_t)T * K));
  HIPCHK(d_st.upload(obsStats, (size_t)K * N * S));
  HIPCHK(d_p.upload(post, (size_t)T * N));
  if (segRatios) HIPCHK(d_r.upload(segRatios, (size_t)T));
  hipLaunchKernelGGL((k_accumulate_obs<ObsT>), dim3(grid_for((int64_t)K * N, 64)), dim3(64), 0, 0, T, K, N,
                     S, d_obs.p, d_st.p, d_p.p, d_r.p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(obsStats, d_st.p, (size_t)K * N * S * sizeof(double), hipMemcpyDeviceToHost));
  return TEHMM_OK;
}

int tehmm_accumulate_obs_u8(int64_t T, int K, int N, int S, const uint8_t *obs, double *obsStats,
                            const double *post, const double *segRatios) {
  return accumulate_impl<uint8_t>(T, K, N, S, obs, obsStats, post, segRatios);
}
int tehmm_accumulate_obs_u16(int64_t T, int K, int N, int S, const uint16_t *obs, double *obsStats,
                             const double *post, const double *segRatios) {
  return accumulate_impl<uint16_t>(T, K, N, S, obs, obsStats, post, segRatios);
}
int tehmm_accumulate_obs_i32(int64_t T, int K, int N, int S, const int32_t *obs, double *obsStats,
                             const double *post, const double *segRatios) {
  return accumulate_impl<int32_t>(T, K, N, S, obs, obsStats, post, segRatios);
}

template <typename ObsT>
static int update_counts_impl(int64_t T, int K, int N, int S, const ObsT *obs, int n_iv,
                              const int64_t *starts, const int64_t *ends, const int32_t *states,
                              const double *segRatios, double *obsStats) {
  if (T < 0 || K <= 0 || N <= 0 || S <= 0 || n_iv < 0 || !obs || !obsStats || (n_iv > 0 && (!starts || !ends || !states)))
    return fail(TEHMM_ERR_ARG, "tehmm_update_counts: bad argument");
  for (int i = 0; i < n_iv; ++i)
    if (starts[i] < 0 || ends[i] > T || states[i] < 0 || states[i] >= N)
      return fail(TEHMM_ERR_ARG, "tehmm_update_counts: interval outside the table or state >= N");
  if (T == 0 || n_iv == 0) return TEHMM_OK;
  DBuf<ObsT> d_obs;
  DBuf<double> d_st, d_r;
  DBuf<int64_t> d_s, d_e;
  DBuf<int32_t> d_q;
  HIPCHK(d_obs.upload(obs, (size_t)T * K));
  HIPCHK(d_st.upload(obsStats, (size_t)K * N * S));
  HIPCHK(d_s.upload(starts, (size_t)n_iv));
  HIPCHK(d_e.upload(ends, (size_t)n_iv));
  HIPCHK(d_q.upload(states, (size_t)n_iv));
  if (segRatios) HIPCHK(d_r.upload(segRatios, (size_t)T));
  hipLaunchKernelGGL((k_update_counts<ObsT>), dim3(K, N), dim3(std::min(256, (S + 63) & ~63)), 0, 0, n_iv, d_s.p,
                     d_e.p, d_q.p, K, N, S, d_obs.p, d_st.p, d_r.p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(obsStats, d_st.p, (size_t)K * N * S * sizeof(double), hipMemcpyDeviceToHost));
  return TEHMM_OK;
}

int tehmm_update_counts_u8(int64_t T, int K, int N, int S, const uint8_t *obs, int n_iv, const int64_t *starts,
                           const int64_t *ends, const int32_t *states, const double *segRatios, double *obsStats) {
  return update_counts_impl<uint8_t>(T, K, N, S, obs, n_iv, starts, ends, states, segRatios, obsStats);
}
int tehmm_update_counts_u16(int64_t T, int K, int N, int S, const uint16_t *obs, int n_iv, const int64_t *starts,
                            const int64_t *ends, const int32_t *states, const double *segRatios, double *obsStats) {
  return update_counts_impl<uint16_t>(T, K, N, S, obs, n_iv, starts, ends, states, segRatios, obsStats);
}
int tehmm_update_counts_i32(int64_t T, int K, int N, int S, const int32_t *obs, int n_iv, const int64_t *starts,
                            const int64_t *ends, const int32_t *states, const double *segRatios, double *obsStats) {
  return update_counts_impl<int32_t>(T, K, N, S, obs, n_iv, starts, ends, states, segRatios, obsStats);
}

// ------------------------------------------------------------------------------------------
// model / batch handles
// ------------------------------------------------------------------------------------------
// (Re)build the fused passes' emission tables from m->tab (model creation, M-step).
static int build_ptab(tehmm_model *m) {
  if (m->NP > 64) return TEHMM_OK;              // the fused passes are instantiated up to 64 padded states
  const int KS = m->NP / 4, KSP = ((KS + 1) + 1) & ~1, ROW_D = 4 * KSP, ROW_L = ROW_D + 2, K = m->K;
  m->KSP = KSP;
  m->ptab_log = m->normalize != 1.0;
  if (!m->ptab.p) {
    HIPCHK(m->ptab.alloc((size_t)(m->R + 1) * ROW_D));
    HIPCHK(m->ptab_lds.alloc((size_t)std::max(1, m->lds_rows) * ROW_L));
    std::vector<int> info((size_t)3 * K);
    for (int k = 0; k < K; ++k) { info[k] = m->rowbase[k]; info[K + k] = m->rowcnt[k]; info[2 * K + k] = m->ldsbase[k]; }
    HIPCHK(m->d_rowinfo.upload(info.data(), info.size()));
  }
  hipLaunchKernelGGL(k_build_ptab, dim3(grid_for(m->R + 1, 64)), dim3(64), 0, 0, m->R + 1, m->N, m->NP, KSP,
                     (const double *)m->tab.p, m->ptab_log ? 1 : 0, m->ptab.p);
  if (m->lds_rows > 0)
    hipLaunchKernelGGL(k_pack_ptab_lds, dim3(K + 1), dim3(256), 0, 0, K, (const int *)m->d_rowinfo.p,
                       (const int *)m->d_rowinfo.p + K, (const int *)m->d_rowinfo.p + 2 * K, m->lds_zero, m->R, ROW_D,
                       ROW_L, (const double *)m->ptab.p, m->ptab_lds.p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipDeviceSynchronize());
  return TEHMM_OK;
}

int tehmm_model_create(int N, int K, int S, const double *lt, const double *pi,
                       const double *logProbs, double normalize, const int32_t *symbolsPerTrack,
                       tehmm_model_t **out) {
  if (!out) return fail(TEHMM_ERR_ARG, "tehmm_model_create: out is NULL");
  *out = nullptr;
  if (N <= 0 || K <= 0 || S <= 0 || !lt || !pi || !logProbs)
    return fail(TEHMM_ERR_ARG, "tehmm_model_create: bad argument");
  if (N > kMaxStates || K > TEHMM_MAX_TRACKS || S > 256)
    return fail(TEHMM_ERR_UNSUPPORTED, "tehmm_model_create: N > 128, K > 128 or S > 256");
  tehmm_model *m = new tehmm_model();
  {
    static std::atomic<uint64_t> next_uid{1};
    m->uid = next_uid.fetch_add(1);
  }
  m->N = N;
  m->K = K;
  m->S = S;
  m->NP = pad_states(N);
  m->normalize = normalize;
  const int NP = m->NP;
  // tables are [NP][NP]; padding states have -inf log-transitions / zero probability
  std::vector<double> hlt((size_t)NP * NP, -INFINITY), hA((size_t)NP * NP, 0.0),
      hAT((size_t)NP * NP, 0.0), hpi(NP, -INFINITY);
  for (int i = 0; i < N; ++i) {
    hpi[i] = pi[i];
    for (int j = 0; j < N; ++j) {
      double l = lt[(size_t)i * N + j];
      hlt[(size_t)i * NP + j] = l;
      double a = std::exp(l);
      hA[(size_t)i * NP + j] = a;
      hAT[(size_t)j * NP + i] = a;
    }
  }
  m->h_lt.assign(lt, lt + (size_t)N * N);
  int R = 0;
  for (int k = 0; k < K; ++k) {
    int cnt = S;
    if (symbolsPerTrack) {
      cnt = symbolsPerTrack[k] + 1;
      if (cnt < 1) cnt = 1;
      if (cnt > S) cnt = S;
    }
    m->rowbase[k] = R;
    m->rowcnt[k] = cnt;
    R += cnt;
  }
  m->R = R;
  // row R: the zero padding (EmisTab::zero_row); 128 doubles of slack behind it (k_wide_emis_tile reads a row up to
  // state NPW - 1 <= 127 whatever NP is)
  std::vector<double> htab((size_t)(R + 1) * NP + 128, 0.0);
  for (int k = 0; k < K; ++k)
    for (int s = 0; s < m->rowcnt[k]; ++s)
      for (int j = 0; j < N; ++j)
        htab[(size_t)(m->rowbase[k] + s) * NP + j] = logProbs[((size_t)k * N + j) * S + s];
  // Small tracks' rows are staged in LDS by the cooperative kernels: smallest tracks first while
  // they fit the budget (the 250-bin gaussian tracks stay in global memory / L2).
  {
    const size_t budget_rows = (size_t)(32 * 1024) / ((size_t)NP * sizeof(double));
    std::vector<int> ord(K);
    std::iota(ord.begin(), ord.end(), 0);
    std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return m->rowcnt[a] < m->rowcnt[b]; });
    for (int k = 0; k < K; ++k) m->ldsbase[k] = -1;
    int used = 0;
    for (int k : ord) {
      if ((size_t)(used + m->rowcnt[k] + 1) > budget_rows) break;
      m->ldsbase[k] = used;
      used += m->rowcnt[k];
    }
    m->lds_zero = used;                  // one row of zeros behind the staged rows
    m->lds_rows = used + 1;              // (always at least the zero row: the fused passes pad with it)
  }
  std::vector<double> hltab((size_t)std::max(1, m->lds_rows) * NP, 0.0);
  for (int k = 0; k < K; ++k)
    if (m->ldsbase[k] >= 0)
      for (int s = 0; s < m->rowcnt[k]; ++s)
        for (int j = 0; j < NP; ++j)
          hltab[(size_t)(m->ldsbase[k] + s) * NP + j] = htab[(size_t)(m->rowbase[k] + s) * NP + j];
  std::vector<double> hltT((size_t)NP * NP, -INFINITY);
  for (int i = 0; i < NP; ++i)
    for (int j = 0; j < NP; ++j) hltT[(size_t)j * NP + i] = hlt[(size_t)i * NP + j];
  auto group_major = [&](const std::vector<double> &src) {      // [f][o] -> [o / 4][f][o % 4]
    std::vector<double> g((size_t)NP * NP);
    for (int f = 0; f < NP; ++f)
      for (int o = 0; o < NP; ++o) g[(size_t)TEHMM_TG(f, o, NP)] = src[(size_t)f * NP + o];
    return g;
  };
  const std::vector<double> hltG = group_major(hlt), hAG = group_major(hA), hATG = group_major(hAT);
  hipError_t e = m->lt.upload(hlt.data(), hlt.size());
  std::vector<float> hltP((size_t)NP * NP + NP + 2, 0.f);    // + lt[o][o] per output and lt[0][0] (segment-ratio P0)
  for (int f = 0; f < NP; ++f)
    for (int o = 0; o < NP; ++o) hltP[((size_t)(o >> 1) * NP + f) * 2 + (o & 1)] = (float)hlt[(size_t)f * NP + o];
  for (int o = 0; o < N; ++o) hltP[(size_t)NP * NP + o] = (float)hlt[(size_t)o * NP + o];
  hltP[(size_t)NP * NP + NP] = (float)hlt[0];
  if (e == hipSuccess) e = m->ltG.upload(hltG.data(), hltG.size());
  if (e == hipSuccess) e = m->ltP.upload(hltP.data(), hltP.size());
  if (e == hipSuccess) e = m->AG.upload(hAG.data(), hAG.size());
  if (e == hipSuccess) e = m->ATG.upload(hATG.data(), hATG.size());
  if (e == hipSuccess) e = m->ltT.upload(hltT.data(), hltT.size());
  if (e == hipSuccess) e = m->ltab.upload(hltab.data(), hltab.size());
  if (e == hipSuccess) e = m->A.upload(hA.data(), hA.size());
  if (e == hipSuccess) e = m->AT.upload(hAT.data(), hAT.size());
  if (e == hipSuccess) e = m->pi.upload(hpi.data(), hpi.size());
  htab.resize(htab.size() + 64, 0.0);     // (slack: k_emis_gain_lane3's last wave reads a full slice of states past NP)
  if (e == hipSuccess) e = m->tab.upload(htab.data(), htab.size());
  if (e != hipSuccess) {
    delete m;
    return fail(TEHMM_ERR_HIP, std::string("tehmm_model_create: ") + hipGetErrorString(e));
  }
  if (int rc = build_ptab(m)) {
    delete m;
    return rc;
  }
  *out = m;
  return TEHMM_OK;
}

int tehmm_model_destroy(tehmm_model_t *model) {
  delete model;
  return TEHMM_OK;
}

// ---- streams and events of destroyed batches, kept for the next batch of the same device --------------------------
namespace {
struct StreamSet {
  int dev;
  hipStream_t sV, sP, sB;
  hipEvent_t evX[2], ev[16];
};
std::mutex g_sset_mu;
std::vector<StreamSet> g_sset;
bool stream_set_take(tehmm_batch *b) {
  const int dev = dev_current();
  std::lock_guard<std::mutex> lk(g_sset_mu);
  for (size_t i = 0; i < g_sset.size(); ++i)
    if (g_sset[i].dev == dev) {
      const StreamSet ss = g_sset[i];
      g_sset.erase(g_sset.begin() + (long)i);
      b->sV = ss.sV; b->sP = ss.sP; b->sB = ss.sB;
      b->evX[0] = ss.evX[0]; b->evX[1] = ss.evX[1];
      for (int k = 0; k < 16; ++k) b->ev[k] = ss.ev[k];
      b->n_ev = 16;
      return true;
    }
  return false;
}
bool stream_set_give(tehmm_batch *b) {
  if (!(b->sV && b->sP && b->sB && b->evX[0] && b->evX[1] && b->n_ev == 16)) return false;
  StreamSet ss;
  ss.dev = dev_current();
  ss.sV = b->sV; ss.sP = b->sP; ss.sB = b->sB;
  ss.evX[0] = b->evX[0]; ss.evX[1] = b->evX[1];
  for (int k = 0; k < 16; ++k) ss.ev[k] = b->ev[k];
  std::lock_guard<std::mutex> lk(g_sset_mu);
  if (g_sset.size() >= 16) return false;
  g_sset.push_back(ss);
  b->sV = b->sP = b->sB = nullptr;
  b->evX[0] = b->evX[1] = nullptr;
  b->n_ev = 0;
  return true;
}
}  // namespace

int tehmm_batch_create(int n, const int64_t *offsets, int K, const uint8_t *obs,
                       const double *segRatios, int obs_on_device, tehmm_batch_t **out) {
  if (!out) return fail(TEHMM_ERR_ARG, "tehmm_batch_create: out is NULL");
  *out = nullptr;
  if (n < 0 || !offsets || K <= 0 || K > TEHMM_MAX_TRACKS || (!obs && n > 0 && offsets[n] > 0))
    return fail(TEHMM_ERR_ARG, "tehmm_batch_create: bad argument");
  for (int i = 0; i < n; ++i)
    if (offsets[i + 1] < offsets[i] || offsets[0] != 0)
      return fail(TEHMM_ERR_ARG, "tehmm_batch_create: offsets must start at 0 and be non-decreasing");
  tehmm_batch *b = new tehmm_batch();
  b->n = n;
  b->K = K;
  b->KP = (K + 3) & ~3;
  b->total = n > 0 ? offsets[n] : 0;
  b->has_ratios = segRatios != nullptr;
  b->h_off.assign(offsets, offsets + n + 1);
  b->h_pos0.resize(n + 1);
  b->h_len.resize(n);
  b->h_chunk0.resize(n + 1);
  int64_t pos = 0, ch = 0;
  for (int i = 0; i < n; ++i) {
    int64_t T = offsets[i + 1] - offsets[i];
    b->h_len[i] = T;
    b->h_pos0[i] = pos;
    pos += (T + 63) & ~(int64_t)63;
    b->h_chunk0[i] = ch;
    ch += T > 1 ? (T - 1 + TEHMM_TB_CHUNK - 1) / TEHMM_TB_CHUNK : 0;
  }
  b->h_pos0[n] = pos;
  b->h_chunk0[n] = ch;
  b->total_pad = pos;
  if (ch > 0x7fffffff) {
    delete b;
    return fail(TEHMM_ERR_UNSUPPORTED, "tehmm_batch_create: too many traceback chunks");
  }
  b->n_chunks = (int)ch;
  b->h_order.resize(n);
  std::iota(b->h_order.begin(), b->h_order.end(), 0);
  std::stable_sort(b->h_order.begin(), b->h_order.end(),
                   [&](int x, int y) { return b->h_len[x] > b->h_len[y]; });
  std::vector<int> chunk_iv((size_t)b->n_chunks);
  for (int i = 0; i < n; ++i)
    for (int64_t c = b->h_chunk0[i]; c < b->h_chunk0[i + 1]; ++c) chunk_iv[(size_t)c] = i;

  hipError_t e = hipSuccess;
  auto up = [&](hipError_t r) { if (e == hipSuccess) e = r; };
  // the Viterbi pipeline is the longer one: its stream gets the higher dispatch priority so that the
  // wide posterior kernels fill in around it instead of delaying it
  // (streams and events of destroyed batches are reused: creating three hardware queues and eighteen events per batch
  //  is slow, and on this runtime it waits for transfers other threads have in flight)
  if (!stream_set_take(b)) {
    int prio_lo = 0, prio_hi = 0;
    if (e == hipSuccess) e = hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    if (e == hipSuccess) e = hipStreamCreateWithPriority(&b->sV, hipStreamNonBlocking, prio_hi);
    if (e == hipSuccess) e = hipStreamCreateWithPriority(&b->sP, hipStreamNonBlocking, prio_lo);
    if (e == hipSuccess) e = hipStreamCreateWithPriority(&b->sB, hipStreamNonBlocking, prio_lo);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&b->evX[0], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&b->evX[1], hipEventDisableTiming);
    for (int i = 0; i < 16 && e == hipSuccess; ++i) {
      e = hipEventCreate(&b->ev[i]);
      if (e == hipSuccess) b->n_ev = i + 1;
    }
  }
  // Everything below runs on the batch's OWN stream (round 4): synchronous copies and the legacy default stream are
  // shared by the whole process, so a batch created while another thread fetched a finished batch's posteriors waited
  // for that transfer (engine.eval_stream: 20 ms per group), and hipDeviceSynchronize waited for every other batch's
  // kernels.
  hipStream_t st = b->sV;
  auto upa = [&](auto &buf, const auto *h, size_t count) {
    if (e != hipSuccess) return;
    e = buf.alloc(count);
    if (e == hipSuccess && count > 0)
      e = hipMemcpyAsync(buf.p, h, count * sizeof(*h), hipMemcpyHostToDevice, st);
  };
  upa(b->d_out0, b->h_off.data(), (size_t)n + 1);
  upa(b->d_pos0, b->h_pos0.data(), (size_t)n + 1);
  upa(b->d_len, b->h_len.data(), (size_t)n);
  upa(b->d_chunk0, b->h_chunk0.data(), (size_t)n + 1);
  upa(b->d_order, b->h_order.data(), (size_t)n);
  upa(b->d_chunk_iv, chunk_iv.data(), chunk_iv.size());
  up(b->obs.alloc((size_t)b->total_pad * b->KP + 16));
  if (segRatios) up(b->ratios.alloc((size_t)b->total_pad + 8));
  if (e == hipSuccess && b->total_pad > 0) e = hipMemsetAsync(b->obs.p, 0, (size_t)b->total_pad * b->KP + 16, st);
  if (e == hipSuccess && segRatios && b->total_pad > 0)
    e = hipMemsetAsync(b->ratios.p, 0, ((size_t)b->total_pad + 8) * sizeof(double), st);
  DBuf<uint8_t> stage_obs;
  DBuf<double> stage_r;
  const uint8_t *src = obs;
  const double *rsrc = segRatios;
  if (e == hipSuccess && !obs_on_device && b->total > 0) {
    upa(stage_obs, obs, (size_t)b->total * K);
    src = stage_obs.p;
    if (segRatios) {
      upa(stage_r, segRatios, (size_t)b->total);
      rsrc = stage_r.p;
    }
  }
  if (e == hipSuccess && obs_on_device && b->total > 0) {
    // The caller's device arrays are read by a kernel on the batch's own (non-blocking) stream, which does not wait for
    // the default stream as the legacy-stream version of rounds 1-3 did: what the caller queued there (torch tensors
    // filled a moment ago) has to be through first.  (bench.py's segment-ratio extra read half-written ratios: no chunk
    // was ever placed and the exact chain walked 30 Mb.)  Producers on other streams synchronise themselves.
    e = hipStreamSynchronize(nullptr);
  }
  if (e == hipSuccess && n > 0 && b->total > 0) {
    int64_t maxT = b->h_len[b->h_order[0]];
    // gridDim.y is limited to 65535: repack in slices of intervals
    for (int i0 = 0; i0 < n; i0 += 32768) {
      const int cnt = std::min(32768, n - i0);
      dim3 grid(grid_for(maxT * b->KP, 256, cnt > 1024 ? 16 : 1024), cnt);
      hipLaunchKernelGGL(k_repack_obs, grid, dim3(256), 0, st, cnt, b->d_out0.p + i0, b->d_pos0.p + i0,
                         b->d_len.p + i0, K, b->KP, src, b->obs.p, rsrc, b->ratios.p);
    }
    up(hipGetLastError());
  }
  if (b->sV) {                                   // (the staging buffers and host vectors live until here)
    const hipError_t es = hipStreamSynchronize(st);
    if (e == hipSuccess) e = es;
  }
  if (e != hipSuccess) {
    tehmm_batch_destroy(b);
    return fail(TEHMM_ERR_HIP, std::string("tehmm_batch_create: ") + hipGetErrorString(e));
  }
  *out = b;
  return TEHMM_OK;
}

// uint16 / int32 observation tables on the fused path: narrowed to the byte rows the kernels keep (a symbol beyond 255
// cannot be: TEHMM_ERR_UNSUPPORTED, the array-level entry points take such tables)
template <typename T>
static int batch_create_narrow(int n, const int64_t *offsets, int K, const T *obs, const double *segRatios, tehmm_batch_t **out,
                               const char *who) {
  if (!out) return fail(TEHMM_ERR_ARG, std::string(who) + ": out is NULL");
  *out = nullptr;
  if (n < 0 || K <= 0 || !offsets || (n > 0 && offsets[n] > 0 && !obs)) return fail(TEHMM_ERR_ARG, std::string(who) + ": bad argument");
  const int64_t total = n > 0 ? offsets[n] : 0;
  if (total < 0) return fail(TEHMM_ERR_ARG, std::string(who) + ": bad offsets");
  std::vector<uint8_t> bytes((size_t)total * K);
  for (size_t i = 0; i < bytes.size(); ++i) {
    const long long v = (long long)obs[i];
    if (v < 0 || v > 255)
      return fail(TEHMM_ERR_UNSUPPORTED, std::string(who) + ": symbol outside 0..255 (the fused kernels keep one byte per track and "
                                                            "position: use the array-level entry points)");
    bytes[i] = (uint8_t)v;
  }
  return tehmm_batch_create(n, offsets, K, bytes.data(), segRatios, 0, out);
}
int tehmm_batch_create_u16(int n, const int64_t *offsets, int K, const uint16_t *obs, const double *segRatios, tehmm_batch_t **out) {
  return batch_create_narrow(n, offsets, K, obs, segRatios, out, "tehmm_batch_create_u16");
}
int tehmm_batch_create_i32(int n, const int64_t *offsets, int K, const int32_t *obs, const double *segRatios, tehmm_batch_t **out) {
  return batch_create_narrow(n, offsets, K, obs, segRatios, out, "tehmm_batch_create_i32");
}

int tehmm_batch_destroy(tehmm_batch_t *b) {
  if (!b) return TEHMM_OK;
  // the batch's blocks go back to the pool: wait for the streams that use them (all work of a batch is on its own
  // three streams since round 4), not for the whole device
  bool synced = b->sV && b->sP && b->sB;
  if (b->sV) synced = hipStreamSynchronize(b->sV) == hipSuccess && synced;
  if (b->sP) synced = hipStreamSynchronize(b->sP) == hipSuccess && synced;
  if (b->sB) synced = hipStreamSynchronize(b->sB) == hipSuccess && synced;
  struct Guard { bool prev; Guard(bool v) : prev(t_streams_synced) { t_streams_synced = v; } ~Guard() { t_streams_synced = prev; } } guard(synced);
  if (!(synced && stream_set_give(b))) {
    for (int i = 0; i < b->n_ev; ++i) (void)hipEventDestroy(b->ev[i]);
    if (b->sV) (void)hipStreamDestroy(b->sV);
    if (b->sP) (void)hipStreamDestroy(b->sP);
    if (b->sB) (void)hipStreamDestroy(b->sB);
    for (int i = 0; i < 2; ++i)
      if (b->evX[i]) (void)hipEventDestroy(b->evX[i]);
  }
  for (int i = 0; i < 2; ++i)
    if (b->stage[i]) (void)hipHostFree(b->stage[i]);
  delete b;
  return TEHMM_OK;
}

int64_t tehmm_batch_total(const tehmm_batch_t *b) { return b ? b->total : 0; }

int tehmm_batch_reset_cache(tehmm_batch_t *b) {
  if (!b) return fail(TEHMM_ERR_ARG, "tehmm_batch_reset_cache: NULL handle");
  b->lw.rix_model = 0;
  b->lw.rix_L = 0;
  b->lw.rix_Wu = 0;
  b->lw.wu_val = 0;
  return TEHMM_OK;
}

static void fill_tabs(const tehmm_model *m, const tehmm_batch *b, IntervalTab &iv, EmisTab &em,
                      bool emis_ratios) {
  iv.order = b->d_order.p;
  iv.pos0 = b->d_pos0.p;
  iv.len = b->d_len.p;
  iv.out0 = b->d_out0.p;
  iv.n = b->n;
  em.obs32 = (const uint32_t *)b->obs.p;
  em.tab = m->tab.p;
  em.ratios = emis_ratios ? b->ratios.p : nullptr;
  em.normalize = m->normalize;
  em.K = m->K;
  em.KPW = b->KP / 4;
  em.NP = m->NP;
  std::memcpy(em.rowbase, m->rowbase, sizeof(em.rowbase));
  std::memcpy(em.rowcnt, m->rowcnt, sizeof(em.rowcnt));
  std::memcpy(em.ldsbase, m->ldsbase, sizeof(em.ldsbase));
  em.lds_rows = m->lds_rows;
  em.ltab_src = m->ltab.p;
  em.zero_row = m->R;
  em.lds_zero = m->lds_zero;
}

// With more intervals than CUs the forward/backward kernel is throughput-bound: dropping the LDS
// copy of the small-track tables (they then come from L2) brings its LDS below 80 KB so that two
// workgroups share a CU.
static EmisTab without_lds_tables(const EmisTab &em) {
  EmisTab e = em;
  e.lds_rows = 0;
  for (int k = 0; k < TEHMM_MAX_TRACKS; ++k) e.ldsbase[k] = -1;
  return e;
}

static int ensure_workspace(tehmm_batch *b, const tehmm_model *m, int flags) {
  if (b->N != m->N) {
    b->paths.release();
    b->post.release();
    b->tb.release();
    b->G.release();
    b->beta.release();
    b->N = m->N;
    b->NP = m->NP;
    b->TBW = m->NP;
  }
  if ((flags & TEHMM_EVAL_VITERBI) && !b->paths.p) {
    HIPCHK(b->paths.alloc((size_t)b->total + 1));
    HIPCHK(b->tb.alloc((size_t)(b->total_pad + 1) * b->TBW));
    HIPCHK(b->G.alloc((size_t)(b->n_chunks + 1) * b->NP));
    HIPCHK(b->bstate.alloc((size_t)b->n_chunks + 1));
    HIPCHK(b->last_state.alloc((size_t)b->n + 1));
    HIPCHK(b->vit_lp.alloc((size_t)b->n + 1));
  }
  if ((flags & TEHMM_EVAL_POSTERIOR) && !b->post.p) {
    HIPCHK(b->post.alloc((size_t)b->total * m->N + 1));
    HIPCHK(b->fwd_lp.alloc((size_t)b->n + 1));
    HIPCHK(b->first_good.alloc((size_t)b->n + 1));
    HIPCHK(b->dead.alloc((size_t)b->n + 1));
  }
  return TEHMM_OK;
}

static int ensure_beta(tehmm_batch *b, const tehmm_model *m) {
  if (!b->beta.p) HIPCHK(b->beta.alloc((size_t)b->total * m->N + 1));
  return TEHMM_OK;
}

// 64 <= N <= 128: the four-wave sequential kernels (k_*_wide); TEHMM_WIDE=0 selects the one-wave kernels
static bool use_wide() {
  const char *s = std::getenv("TEHMM_WIDE");
  return !(s && std::atoi(s) == 0);
}

template <int SPL>
static void launch_viterbi(tehmm_batch *b, const tehmm_model *m, const IntervalTab &iv,
                           const EmisTab &em, bool ratio, hipStream_t st) {
  if (use_wide() && m->NP / 4 <= TEHMM_WIDE_QM && wide_lds_bytes(em.lds_rows, m->NP) <= 160 * 1024) {
    const size_t lds = wide_lds_bytes(em.lds_rows, m->NP);
    allow_lds(k_viterbi_wide<true>, lds);
    allow_lds(k_viterbi_wide<false>, lds);
    if (ratio)
      hipLaunchKernelGGL((k_viterbi_wide<true>), dim3(b->n), dim3(256), lds, st, iv, em, m->N, m->NP, m->lt.p, m->pi.p,
                         (const double *)b->ratios.p, b->TBW, b->tb.p, b->last_state.p, b->vit_lp.p);
    else
      hipLaunchKernelGGL((k_viterbi_wide<false>), dim3(b->n), dim3(256), lds, st, iv, em, m->N, m->NP, m->lt.p, m->pi.p,
                         (const double *)nullptr, b->TBW, b->tb.p, b->last_state.p, b->vit_lp.p);
    return;
  }
  size_t lds = ((size_t)m->N * m->NP + (TEHMM_PB + 2) * 64 * SPL) * sizeof(double);
  allow_lds(k_viterbi<SPL, true, false>, lds);
  allow_lds(k_viterbi<SPL, false, false>, lds);
  if (ratio)
    hipLaunchKernelGGL((k_viterbi<SPL, true, false>), dim3(b->n), dim3(64), lds, st, iv, em, m->N, m->NP,
                       m->lt.p, m->pi.p, b->ratios.p, (const double *)nullptr, b->TBW, b->tb.p,
                       b->last_state.p, b->vit_lp.p);
  else
    hipLaunchKernelGGL((k_viterbi<SPL, false, false>), dim3(b->n), dim3(64), lds, st, iv, em, m->N,
                       m->NP, m->lt.p, m->pi.p, (const double *)nullptr, (const double *)nullptr,
                       b->TBW, b->tb.p, b->last_state.p, b->vit_lp.p);
}

template <int SPL>
static void launch_posterior(tehmm_batch *b, const tehmm_model *m, const IntervalTab &iv,
                             const EmisTab &em, hipStream_t st, hipEvent_t mid) {
  if (use_wide() && m->NP / 4 <= TEHMM_WIDE_QM && wide_lds_bytes(em.lds_rows, m->NP) <= 160 * 1024) {
    const size_t lds = wide_lds_bytes(em.lds_rows, m->NP);
    allow_lds(k_forward_wide<false>, lds);
    allow_lds(k_backward_wide<false, true>, lds);
    hipLaunchKernelGGL((k_forward_wide<false>), dim3(b->n), dim3(256), lds, st, iv, em, m->N, m->NP, m->A.p, m->lt.p,
                       m->pi.p, (const double *)nullptr, b->post.p, b->fwd_lp.p, b->first_good.p);
    (void)hipEventRecord(mid, st);
    hipLaunchKernelGGL((k_backward_wide<false, true>), dim3(b->n), dim3(256), lds, st, iv, em, m->N, m->NP, m->AT.p,
                       m->lt.p, (const double *)nullptr, b->post.p, b->first_good.p);
    return;
  }
  size_t ldsF = ((size_t)m->N * m->NP + (TEHMM_PB + 2) * 64 * SPL + TEHMM_PB) * sizeof(double);
  size_t ldsB = ((size_t)m->N * m->NP + (TEHMM_PB + 2) * 64 * SPL) * sizeof(double);
  allow_lds(k_forward_lin<SPL, false>, ldsF);
  allow_lds(k_backward_lin<SPL, false, true>, ldsB);
  hipLaunchKernelGGL((k_forward_lin<SPL, false>), dim3(b->n), dim3(64), ldsF, st, iv, em, m->N, m->NP,
                     m->A.p, m->lt.p, m->pi.p, (const double *)nullptr, b->post.p, b->fwd_lp.p,
                     b->first_good.p);
  (void)hipEventRecord(mid, st);
  hipLaunchKernelGGL((k_backward_lin<SPL, false, true>), dim3(b->n), dim3(64), ldsB, st, iv, em, m->N,
                     m->NP, m->AT.p, m->lt.p, (const double *)nullptr, b->post.p, b->first_good.p);
}

// ---- cooperative (one workgroup per interval) path, N <= 64 --------------------------------
template <int NT>
static void launch_vit_coop(tehmm_batch *b, const tehmm_model *m, const IntervalTab &iv,
                            const EmisTab &em, bool ratio, hipStream_t st) {
  constexpr int CPB = NT <= 44 ? 64 : 32;
  size_t lds = ((size_t)3 * CPB * (NT + 1) + 2 * (CPB + 1) * (NT + 2) + (size_t)m->lds_rows * NT) * sizeof(double);
  if (ratio) {
    allow_lds(k_vit_coop<NT, CPB, true>, lds);
    hipLaunchKernelGGL((k_vit_coop<NT, CPB, true>), dim3(b->n), dim3(256), lds, st, iv, em, m->N,
                       m->lt.p, m->ltT.p, m->pi.p, b->ratios.p, b->tb.p, b->last_state.p, b->vit_lp.p);
  } else {
    allow_lds(k_vit_coop<NT, CPB, false>, lds);
    hipLaunchKernelGGL((k_vit_coop<NT, CPB, false>), dim3(b->n), dim3(256), lds, st, iv, em, m->N,
                       m->lt.p, m->ltT.p, m->pi.p, (const double *)nullptr, b->tb.p, b->last_state.p,
                       b->vit_lp.p);
  }
}

template <int NT>
static void launch_fb_coop(tehmm_batch *b, const tehmm_model *m, const IntervalTab &iv,
                           const EmisTab &em_in, hipStream_t st) {
  constexpr int CPB = NT <= 44 ? 64 : 32;
  const EmisTab em = b->n > 256 ? without_lds_tables(em_in) : em_in;
  size_t lds = ((size_t)4 * CPB * (NT + 1) + 4 * CPB + 5 * NT + (size_t)em.lds_rows * NT) * sizeof(double);
  allow_lds(k_fb_coop<NT, CPB, false>, lds);
  hipLaunchKernelGGL((k_fb_coop<NT, CPB, false>), dim3(b->n), dim3(256), lds, st, iv, em, m->N, m->A.p,
                     m->lt.p, m->pi.p, (const double *)nullptr, b->post.p, b->beta.p, b->fwd_lp.p,
                     b->dead.p, (double *)nullptr, (int *)nullptr);
}

// ---- speculative chunk-parallel exact Viterbi (no segment ratios) ----------------------------
static int spec_chunk_size() {
  const char *s = std::getenv("TEHMM_SPEC_CHUNK");     // 0 disables; tests use small chunks
  // (round 4: 512 instead of 1024 -- a chunk that crosses into the next binade is walked exactly up to the crossing,
  //  half a chunk on average, and with the rounding ties passed those walks were what the chain still did: one 10 Mb
  //  interval 15.6 -> 13.7 ms, exact blocks 452 -> 359; 256: 14.9 ms, the per-chunk kernels then cost more than
  //  the chain saves)
  int cs = s ? std::atoi(s) : 512;
  if (cs <= 0) return 0;
  return std::max(64, (cs + 63) & ~63);
}

static int spec_prepare(tehmm_batch *b, const tehmm_model *m, int CS) {
  SpecWork &sw = b->sw;
  if (sw.CS == CS && sw.N == m->N && sw.rows.p) return TEHMM_OK;
  sw.CS = CS;
  sw.N = m->N;
  sw.h_iv.clear();
  sw.h_t0.clear();
  sw.h_first.assign((size_t)b->n + 1, 0);
  for (int i = 0; i < b->n; ++i) {
    sw.h_first[i] = (int64_t)sw.h_iv.size();
    for (int64_t t0 = 0; t0 < b->h_len[i]; t0 += CS) {
      sw.h_iv.push_back(i);
      sw.h_t0.push_back(t0);
    }
  }
  sw.h_first[b->n] = (int64_t)sw.h_iv.size();
  sw.n_chunks = (int)sw.h_iv.size();
  const size_t nc = (size_t)std::max(1, sw.n_chunks);
  HIPCHK(sw.iv.upload(sw.h_iv.data(), sw.h_iv.size()));
  HIPCHK(sw.t0.upload(sw.h_t0.data(), sw.h_t0.size()));
  HIPCHK(sw.first.upload(sw.h_first.data(), sw.h_first.size()));
  HIPCHK(sw.e.alloc(nc));
  HIPCHK(sw.ok.alloc(nc));
  HIPCHK(sw.stats.alloc(16));
  HIPCHK(hipMemset(sw.stats.p, 0, 16 * sizeof(int)));
  HIPCHK(sw.scale.alloc(nc * (size_t)(CS / 32)));
  HIPCHK(sw.wstart.alloc(nc * (size_t)m->NP));
  HIPCHK(sw.gain.alloc(nc));
  HIPCHK(sw.wmin.alloc(nc));
  HIPCHK(sw.rows.alloc(nc * (size_t)(CS / TEHMM_VROW) * m->NP));
  HIPCHK(sw.ntie.alloc(nc));
  HIPCHK(sw.ties.alloc(nc * TEHMM_SPEC_MAXT));
  HIPCHK(sw.tierows.alloc(nc * TEHMM_SPEC_MAXT * (size_t)m->NP));
  HIPCHK(sw.segmin.alloc(nc * (TEHMM_SPEC_MAXT + 1)));
  HIPCHK(sw.offend.alloc(nc));
  HIPCHK(sw.clk.alloc(nc));
  HIPCHK(sw.racc.alloc(2 * nc));           // (the run scan is kept per parity of the chain's delta: [2][chunks])
  HIPCHK(sw.rmn.alloc(2 * nc));
  HIPCHK(sw.rtarget.alloc(2 * nc));
  HIPCHK(sw.rsel.alloc(2 * nc));
  HIPCHK(sw.tsoft.alloc(nc));
  HIPCHK(sw.tpar.alloc(nc));
  HIPCHK(hipMemset(sw.tsoft.p, 0, nc * sizeof(unsigned)));
  HIPCHK(hipMemset(sw.tpar.p, 0, nc * sizeof(unsigned)));
  HIPCHK(sw.clink.alloc(nc));
  HIPCHK(hipMemset(sw.clink.p, 0, nc * sizeof(int)));
  return TEHMM_OK;
}

// Which binade each chunk lives in, from the plain-fp chunk gains of pass P0 (approximate prefix
// sums are enough: a wrong or risky guess only sends that chunk to the sequential chain -- the chain's
// check demands every live value inside the chunk's binade and, from the recorded minima, up to the
// landing position).  The margin (512 + 2e-5 |V|; float P0 sums are good to ~2e-6) only keeps chunks
// that end within it of a binade boundary from being speculated in the wrong binade.
static void spec_assign_binades(const tehmm_batch *b, const std::vector<double> &gain, std::vector<int> &e) {
  const SpecWork &sw = b->sw;
  e.assign((size_t)std::max(1, sw.n_chunks), TEHMM_SPEC_NONE);
  for (int i = 0; i < b->n; ++i) {
    double v = 0.0;
    for (int64_t c = sw.h_first[i]; c < sw.h_first[i + 1]; ++c) {
      const double g = gain[(size_t)c];
      const double vs = v, ve = v + g;
      v = ve;
      const bool full = sw.h_t0[(size_t)c] + sw.CS <= b->h_len[i];
      if (c == sw.h_first[i] || !full || !(g == g) || !(g < 0.0)) continue;
      static const double rel = std::getenv("TEHMM_SPEC_MARGIN") ? std::atof(std::getenv("TEHMM_SPEC_MARGIN")) : 2e-5;
      const double margin = 512.0 + rel * std::fabs(ve);
      const double lo = std::fabs(vs) - margin, hi = std::fabs(ve) + margin;
      if (!(lo > 0.0)) continue;
      int ex = 0, exh = 0;
      (void)std::frexp(lo, &ex);          // lo = f * 2^ex, f in [0.5, 1)  ->  binade exponent ex - 1
      (void)std::frexp(hi, &exh);
      // A chunk that crosses into the next binade is quantised for the binade it ends in: the exact chain
      // runs it up to the crossing and is then verified against its rows like anywhere else (the check
      // demands every live value inside the binade, so nothing before the crossing is ever adopted).
      static const bool cross = !(std::getenv("TEHMM_SPEC_CROSS") && std::atoi(std::getenv("TEHMM_SPEC_CROSS")) == 0);
      if (exh != ex && !(cross && exh == ex + 1)) continue;
      const int be = exh - 1;
      if (be < TEHMM_SPEC_MIN_E) continue;
      e[(size_t)c] = be;
    }
  }
}

template <int NT>
static void launch_vit_spec(tehmm_batch *b, const tehmm_model *m, const IntervalTab &iv, const EmisTab &em,
                            const VitChunks &vc, bool quant, hipStream_t st) {
  size_t lds = (size_t)4 * 64 * (NT + 1) * sizeof(double) + 4 * 64 * sizeof(int);
  const int grid = std::min((vc.n + 3) / 4, 4096);
  if (quant) {
    allow_lds(k_vit_spec<NT, true>, lds);
    hipLaunchKernelGGL((k_vit_spec<NT, true>), dim3(grid), dim3(256), lds, st, iv, em, vc, m->N, m->lt.p,
                       b->tb.p);
  } else {
    allow_lds(k_vit_spec<NT, false>, lds);
    hipLaunchKernelGGL((k_vit_spec<NT, false>), dim3(grid), dim3(256), lds, st, iv, em, vc, m->N, m->lt.p,
                       b->tb.p);
  }
}

template <int NT>
static void launch_vit_fix(tehmm_batch *b, const tehmm_model *m, const IntervalTab &iv, const EmisTab &em,
                           const VitChunks &vc, bool segmin, bool ratio, hipStream_t st) {
  size_t lds = ((size_t)3 * 32 * (NT + 1) + 2 * 33 * (NT + 2) + (size_t)m->lds_rows * NT + 16) * sizeof(double);   // CPB = 32
  allow_lds(k_vit_fix<NT, false>, lds);
  allow_lds(k_vit_fix<NT, true>, lds);
  if constexpr (NT <= TEHMM_RATIO_LANE_MAX) if (ratio) {
    allow_lds(k_vit_fix<NT, true, true>, lds);
    hipLaunchKernelGGL((k_vit_fix<NT, true, true>), dim3(b->n), dim3(256), lds, st, iv, em, vc, m->N, m->lt.p,
                       m->ltT.p, m->pi.p, b->tb.p, b->last_state.p, b->vit_lp.p, b->sw.stats.p,
                       (const double *)b->ratios.p);
    return;
  }
  if (segmin)
    hipLaunchKernelGGL((k_vit_fix<NT, true>), dim3(b->n), dim3(256), lds, st, iv, em, vc, m->N, m->lt.p,
                       m->ltT.p, m->pi.p, b->tb.p, b->last_state.p, b->vit_lp.p, b->sw.stats.p,
                       (const double *)nullptr);
  else
    hipLaunchKernelGGL((k_vit_fix<NT, false>), dim3(b->n), dim3(256), lds, st, iv, em, vc, m->N, m->lt.p,
                       m->ltT.p, m->pi.p, b->tb.p, b->last_state.p, b->vit_lp.p, b->sw.stats.p,
                       (const double *)nullptr);
}

// ---- lane = item passes -----------------------------------------------------------------------
static LaneGeom lane_geom(const LaneWork &lw);

// Item length L (L | CS, 64 | L).  Every wave owns 64 items; 512 positions per item keep the warm-up
// overhead below 20 %, 256 / 128 are used for small batches so that the 1024 SIMDs still get a wave each.
// TEHMM_LANE_SUB overrides (0 disables the lane passes).
static int lane_sub_size(int CS, int64_t total) {
  if (CS <= 0) return 0;
  // (measured: 10 Mb in one interval runs the lane passes in 16.1 ms with 128-position items, 18.2 with 256)
  int L = total >= (int64_t)512 * 64 * 1024 ? 512 : (total >= (int64_t)256 * 64 * 1024 ? 256 : 128);
  if (const char *s = std::getenv("TEHMM_LANE_SUB")) {
    L = std::atoi(s);
    if (L <= 0) return 0;
    L = std::max(64, (L + 63) & ~63);
  }
  if (L > CS || CS % L != 0) L = CS;
  return L;
}

static int lane_prepare(tehmm_batch *b, const tehmm_model *m, int CS, int L, bool want_fb, bool want_vit,
                        bool want_gain, bool fused_fb, bool want_b32) {
  LaneWork &lw = b->lw;
  if (lw.L != L || lw.CS != CS || lw.NP != m->NP || !lw.item_iv.p) {
    std::vector<int> h_iv;
    std::vector<int64_t> h_t0, h_first((size_t)b->n + 1, 0);
    for (int i = 0; i < b->n; ++i) {
      h_first[i] = (int64_t)h_iv.size();
      for (int64_t t0 = 0; t0 < b->h_len[i]; t0 += L) {
        h_iv.push_back(i);
        h_t0.push_back(t0);
      }
    }
    h_first[b->n] = (int64_t)h_iv.size();
    for (DBuf<double> *d : {&lw.B, &lw.BH, &lw.MS, &lw.AL, &lw.BE, &lw.pre_f, &lw.end_f, &lw.pre_b, &lw.end_b,
                            &lw.slog32, &lw.vpre, &lw.vend, &lw.vgain, &lw.vtierows, &lw.vpiecemin, &lw.chk, &lw.chkf, &lw.sink})
      d->release();
    for (DBuf<int> *d : {&lw.ok_f, &lw.ok_b, &lw.vbad, &lw.vntie, &lw.vties, &lw.link_f, &lw.link_b, &lw.runend_f,
                         &lw.runstart_b})
      d->release();
    lw.glog_f.release();
    lw.cpre_f.release();
    lw.dl_f.release();
    lw.lr_f.release();
    lw.dl_b.release();
    lw.B32.release();
    lw.AL32.release();
    lw.GAM32.release();
    lw.WZ32.release();
    lw.L = L; lw.CS = CS; lw.NP = m->NP;
    lw.n_items = (int)h_iv.size();
    lw.n_groups = (lw.n_items + 63) / 64;
    lw.h_iv = h_iv;
    lw.h_t0 = h_t0;
    lw.h_first = h_first;
    HIPCHK(lw.item_iv.upload(h_iv.data(), h_iv.size()));
    HIPCHK(lw.item_t0.upload(h_t0.data(), h_t0.size()));
    HIPCHK(lw.ifirst.upload(h_first.data(), h_first.size()));
  }
  const size_t rows = (size_t)std::max(1, lw.n_groups) * L * 64;        // item-interleaved positions
  const size_t vecs = (size_t)std::max(1, lw.n_groups) * 64 * m->NP;
  if (want_fb && !lw.pre_f.p) {
    if (fused_fb) {
      const size_t na = (size_t)std::max(1, lw.n_groups) * L * 512 * al32_pairs(m->NP);
      HIPCHK(lw.AL32.alloc(na));
      HIPCHK(hipMemset(lw.AL32.p, 0, na * sizeof(float)));      // (slots of padding states are read, never NaN)
    }
    else HIPCHK(lw.AL.alloc(rows * m->NP));
    HIPCHK(lw.pre_f.alloc(vecs));
    HIPCHK(lw.end_f.alloc(vecs));
    HIPCHK(lw.pre_b.alloc(vecs));
    HIPCHK(lw.end_b.alloc(vecs));
    HIPCHK(lw.slog32.alloc((size_t)std::max(1, lw.n_groups) * 64 * (L / 32)));
    HIPCHK(lw.ok_f.alloc((size_t)std::max(1, b->sw.n_chunks)));
    HIPCHK(lw.ok_b.alloc((size_t)std::max(1, b->sw.n_chunks)));
    for (DBuf<int> *d : {&lw.link_f, &lw.link_b, &lw.runend_f, &lw.runstart_b})
      HIPCHK(d->alloc((size_t)std::max(1, b->sw.n_chunks)));
    HIPCHK(lw.glog_f.alloc((size_t)std::max(1, b->sw.n_chunks)));
    HIPCHK(lw.cpre_f.alloc((size_t)std::max(1, b->sw.n_chunks)));
    HIPCHK(lw.dl_f.alloc((size_t)std::max(1, lw.n_groups) * 64));
    HIPCHK(lw.lr_f.alloc((size_t)std::max(1, lw.n_groups) * 64));
    HIPCHK(lw.dl_b.alloc((size_t)std::max(1, lw.n_groups) * 64));
  }
  if (want_fb && fused_fb && !lw.sink.p) HIPCHK(lw.sink.alloc((size_t)std::max(1, lw.n_groups) * 4 * 64));
  if (want_fb && fused_fb && !lw.chk.p) {
    HIPCHK(lw.chk.alloc((size_t)std::max(1, lw.n_groups) * 64 * (size_t)(L / 64) * m->NP));
    HIPCHK(lw.chkf.alloc((size_t)std::max(1, lw.n_groups) * 64 * (size_t)(L / 64) * m->NP));
  }
  if (want_fb && fused_fb && !lw.AL32.p) {
    const size_t na = (size_t)std::max(1, lw.n_groups) * L * 512 * al32_pairs(m->NP);
    HIPCHK(lw.AL32.alloc(na));
    HIPCHK(hipMemset(lw.AL32.p, 0, na * sizeof(float)));
  }
  if (want_fb && !fused_fb && !lw.AL.p) HIPCHK(lw.AL.alloc(rows * m->NP));
  if (want_fb && !fused_fb && !lw.BH.p) {
    HIPCHK(lw.BE.alloc(rows * m->NP));
    HIPCHK(lw.BH.alloc(rows * m->NP));
    if (!lw.MS.p) HIPCHK(lw.MS.alloc(rows));
  }
  if (want_gain && !lw.vgain.p) HIPCHK(lw.vgain.alloc((size_t)std::max(1, lw.n_groups) * 64));
  if (want_gain && want_b32 && !lw.B32.p) {
    HIPCHK(lw.B32.alloc(rows * m->NP));
    if (!lw.MS.p) HIPCHK(lw.MS.alloc(rows));
  }
  if (want_vit && !lw.B.p) {
    const size_t ni = (size_t)std::max(1, lw.n_groups) * 64;
    HIPCHK(lw.B.alloc(rows * m->NP));
    if (!lw.MS.p) HIPCHK(lw.MS.alloc(rows));
    HIPCHK(lw.vpre.alloc(vecs));
    HIPCHK(lw.vend.alloc(vecs));
    if (!lw.vgain.p) HIPCHK(lw.vgain.alloc(ni));
    HIPCHK(lw.vbad.alloc(ni));
    HIPCHK(lw.vntie.alloc(ni));
    HIPCHK(lw.vties.alloc(ni * TEHMM_LANE_MAXTI));
    HIPCHK(lw.vtierows.alloc(ni * TEHMM_LANE_MAXTI * m->NP));
    HIPCHK(lw.vpiecemin.alloc(ni * (TEHMM_LANE_MAXTI + 1)));
    HIPCHK(hipMemset(lw.vbad.p, 0, ni * sizeof(int)));
    HIPCHK(hipMemset(lw.vntie.p, 0, ni * sizeof(int)));
  }
  return TEHMM_OK;
}

static VitItems lane_vit_items(LaneWork &lw) {
  VitItems vi;
  vi.pre = lw.vpre.p; vi.end = lw.vend.p; vi.gain = lw.vgain.p; vi.bad = lw.vbad.p; vi.ntie = lw.vntie.p;
  vi.ties = lw.vties.p; vi.tierows = lw.vtierows.p; vi.piecemin = lw.vpiecemin.p;
  return vi;
}

// quantised transition table of binade e as the P2 lane pass wants it (tehmm_spec.hip.h:96-107):
// 64 * R_u(lt[f][j]) + (63 - f) * u, pads -inf.  false if an entry sits exactly between two grid points.
static bool quantised_table(const tehmm_model *m, int e, double *out) {
  const int N = m->N, NP = m->NP;
  const double u = std::ldexp(1.0, e - 52), M = std::ldexp(1.5, e), half_u = 0.5 * u;
  bool ok = true;
  for (int f = 0; f < NP; ++f)
    for (int j = 0; j < NP; ++j) {
      double v = -INFINITY;
      if (f < N && j < N) {
        const double z = m->h_lt[(size_t)f * N + j];
        volatile double zm = z + M;
        const double q = zm - M;
        if (std::fabs(z - q) == half_u) ok = false;
        v = 64.0 * q + (double)(63 - f) * u;
      }
      out[(size_t)TEHMM_TG(f, j, NP)] = v;
    }
  return ok;
}

// The same with one header block per output group for the segment-ratio terms (k_vit_lane<.., RATIO>): lt[o][o]
// of the group's four outputs (unrounded: the product with the ratio is rounded per position) and R_u(lt[0][0]).
static size_t quantised_table_size(const tehmm_model *m, bool ratio) {
  return (size_t)m->NP * m->NP + (ratio ? (size_t)(m->NP / 4) * 16 : 0);
}
static bool quantised_table_ratio(const tehmm_model *m, int e, double *out) {
  const int N = m->N, NP = m->NP;
  std::vector<double> plain((size_t)NP * NP);
  bool ok = quantised_table(m, e, plain.data());
  const double u = std::ldexp(1.0, e - 52), M = std::ldexp(1.5, e), half_u = 0.5 * u;
  const double z = m->h_lt[0];
  volatile double zm = z + M;
  const double q00 = zm - M;
  if (std::fabs(z - q00) == half_u) ok = false;
  for (int og = 0; og < NP / 4; ++og) {
    double *grp = out + (size_t)og * (NP * 4 + 16);
    for (int i = 0; i < 16; ++i) grp[i] = 0.0;
    for (int q = 0; q < 4; ++q) {
      const int o = 4 * og + q;
      grp[q] = o < N ? m->h_lt[(size_t)o * N + o] : 0.0;
    }
    grp[4] = q00;
    std::memcpy(grp + 16, plain.data() + (size_t)og * NP * 4, (size_t)NP * 4 * sizeof(double));
  }
  return ok;
}
// segment ratios on the chunk-parallel path need finite self-transitions (a -inf one makes the products NaN
// or -inf in ways the quantised arithmetic does not mirror; the sequential kernels handle those models)
static bool ratio_lane_ok(const tehmm_model *m) {
  for (int o = 0; o < m->N; ++o)
    if (!std::isfinite(m->h_lt[(size_t)o * m->N + o])) return false;
  return true;
}

// The chunk / item argument tables of the quantised pass live in two small device structs.  They hold pointers only, so
// they change when a workspace is reallocated, not from call to call: upload them when they differ from what the
// device has (a host -> device copy queued behind running kernels waits for a wave slot like any other blit).
static int vit_lane_upload_args(tehmm_batch *b, const VitChunks &vc, hipStream_t st) {
  LaneWork &lw = b->lw;
  const VitItems vi = lane_vit_items(lw);
  if (lw.up_valid && lw.d_vc.p && lw.d_vi.p && std::memcmp(&lw.up_vc, &vc, sizeof(vc)) == 0 &&
      std::memcmp(&lw.up_vi, &vi, sizeof(vi)) == 0)
    return TEHMM_OK;
  lw.hs_vc = vc;
  lw.hs_vi = vi;
  HIPCHK(lw.d_vc.fill_async(&lw.hs_vc, 1, st));
  HIPCHK(lw.d_vi.fill_async(&lw.hs_vi, 1, st));
  std::memcpy(&lw.up_vc, &vc, sizeof(vc));
  std::memcpy(&lw.up_vi, &vi, sizeof(vi));
  lw.up_valid = true;
  return TEHMM_OK;
}

// n_work_dev != nullptr: the number of work units is read on the device (tehmm_place.hip.h) and n_work is only the
// upper bound the grid is sized for; tabs / e0: the quantised tables and the binade of the first one
template <int NT>
static void launch_vit_lane(tehmm_batch *b, const tehmm_model *m, const IntervalTab &iv, const VitChunks &vc,
                            bool quant, bool ratio, int Wu, int n_work, const double *tabs, int e0, const int *n_work_dev,
                            hipStream_t st) {
  LaneWork &lw = b->lw;
  const LaneGeom lg = lane_geom(lw);
  lw.soft_ties = 0;
  if (n_work <= 0) return;
  // round 4: the outputs of a step split over the three waves of a workgroup (tehmm_lane3.hip.h); TEHMM_P2_SPLIT=0
  // keeps the one-wave kernel (which also serves the smallest models)
  if constexpr (NT >= 12) {
    const bool split = !(std::getenv("TEHMM_P2_SPLIT") && std::atoi(std::getenv("TEHMM_P2_SPLIT")) == 0);
    if (quant && split) {
      constexpr int NW = TEHMM_P2_NW;
      const size_t lds = Lane3Geom<NT, NW>::LDS_BYTES;
      if constexpr (NT <= TEHMM_RATIO_LANE_MAX) if (ratio) {
        allow_lds(k_vit_lane3<NT, NW, true>, lds);
        hipLaunchKernelGGL((k_vit_lane3<NT, NW, true>), dim3(n_work), dim3(64 * NW), lds, st, iv, lg,
                           (const VitChunks *)lw.d_vc.p, (const VitItems *)lw.d_vi.p, m->N, Wu, (const int *)lw.wk_g.p,
                           (const int *)lw.wk_e.p, n_work, tabs, e0, (const double *)lw.B.p, b->tb.p,
                           (const double *)b->ratios.p, (const int *)lw.wk_items.p, n_work_dev);
        return;
      }
      // soft ties (TEHMM_SOFT_TIES=0: every rounding tie ends a piece, as in rounds 2 and 3)
      const bool soft = !(std::getenv("TEHMM_SOFT_TIES") && std::atoi(std::getenv("TEHMM_SOFT_TIES")) == 0);
      lw.soft_ties = soft ? 1 : 0;
      allow_lds(k_vit_lane3<NT, NW, false>, lds);
      hipLaunchKernelGGL((k_vit_lane3<NT, NW, false>), dim3(n_work), dim3(64 * NW), lds, st, iv, lg,
                         (const VitChunks *)lw.d_vc.p, (const VitItems *)lw.d_vi.p, m->N, Wu, (const int *)lw.wk_g.p,
                         (const int *)lw.wk_e.p, n_work, tabs, e0, (const double *)lw.B.p, b->tb.p,
                         (const double *)nullptr, (const int *)lw.wk_items.p, n_work_dev, lw.soft_ties);
      return;
    }
  }
  const dim3 grid((n_work + 3) / 4);
  if constexpr (NT <= TEHMM_RATIO_LANE_MAX) if (quant && ratio) {
    hipLaunchKernelGGL((k_vit_lane<NT, true, true>), grid, dim3(256), 0, st, iv, lg, (const VitChunks *)lw.d_vc.p,
                       (const VitItems *)lw.d_vi.p, m->N, Wu,
                       (const int *)lw.wk_g.p, (const int *)lw.wk_e.p, n_work, tabs, e0,
                       (const double *)lw.B.p, b->tb.p, (const double *)b->ratios.p, (const int *)lw.wk_items.p, n_work_dev);
    return;
  }
  if (quant) {
    hipLaunchKernelGGL((k_vit_lane<NT, true>), grid, dim3(256), 0, st, iv, lg, (const VitChunks *)lw.d_vc.p,
                       (const VitItems *)lw.d_vi.p, m->N, Wu,
                       (const int *)lw.wk_g.p, (const int *)lw.wk_e.p, n_work, tabs, e0,
                       (const double *)lw.B.p, b->tb.p, (const double *)nullptr, (const int *)lw.wk_items.p, n_work_dev);
  }
}

// Quantised tables of every binade a score can reach, once per parameter version (device-side placement).
static int ensure_qtabs(tehmm_model *m, bool ratio) {
  const int w = ratio ? 1 : 0;
  if (m->q_version[w] == m->version && m->qall[w].p) return TEHMM_OK;
  const size_t tsz = quantised_table_size(m, ratio);
  std::vector<double> qt((size_t)TEHMM_PLACE_NE * tsz + 64, 0.0);      // (+ one block of padding: k_vit_lane3's prefetch)
  std::vector<int> ok((size_t)TEHMM_PLACE_NE, 0);
  for (int e = TEHMM_SPEC_MIN_E; e <= TEHMM_PLACE_EMAX; ++e) {
    double *dst = qt.data() + (size_t)(e - TEHMM_SPEC_MIN_E) * tsz;
    ok[(size_t)(e - TEHMM_SPEC_MIN_E)] = (ratio ? quantised_table_ratio(m, e, dst) : quantised_table(m, e, dst)) ? 1 : 0;
  }
  HIPCHK(m->qall[w].upload(qt.data(), qt.size()));
  HIPCHK(m->qok[w].upload(ok.data(), ok.size()));
  m->q_version[w] = m->version;
  return TEHMM_OK;
}

// Binade placement and the work list of the quantised pass on the device, in stream order behind the gain pass
// (tehmm_place.hip.h).  Returns the upper bound of work units the quantised pass is launched for.
static int launch_vit_place(tehmm_batch *b, const tehmm_model *m, const IntervalTab &iv, int CS, bool ratio, hipStream_t st,
                            int *n_work_max) {
  LaneWork &lw = b->lw;
  SpecWork &sw = b->sw;
  static const double rel = std::getenv("TEHMM_SPEC_MARGIN") ? std::atof(std::getenv("TEHMM_SPEC_MARGIN")) : 2e-5;
  static const int cross = !(std::getenv("TEHMM_SPEC_CROSS") && std::atoi(std::getenv("TEHMM_SPEC_CROSS")) == 0) ? 1 : 0;
  const LaneGeom lg = lane_geom(lw);
  hipLaunchKernelGGL(k_vit_place_chunks, dim3(std::max(1, b->n)), dim3(64), 0, st, iv, (const int64_t *)sw.first.p,
                     (const int64_t *)sw.t0.p, (const int64_t *)lw.ifirst.p, b->n, CS, lw.L, (const double *)lw.vgain.p,
                     (const int *)m->qok[ratio ? 1 : 0].p, rel, cross, sw.e.p, sw.gain.p);
  const int gb = (lw.n_groups + 255) / 256;
  hipLaunchKernelGGL(k_vit_place_count, dim3(std::max(1, gb)), dim3(256), 0, st, lg, (const int64_t *)sw.first.p,
                     (const int *)sw.e.p, CS, lw.place.p, lw.gclass.p);
  hipLaunchKernelGGL(k_vit_place_scan, dim3(1), dim3(64), 0, st, lw.place.p);
  hipLaunchKernelGGL(k_vit_place_fill, dim3(std::max(1, gb)), dim3(256), 0, st, lg, (const int64_t *)sw.first.p,
                     (const int *)sw.e.p, CS, lw.place.p, (const int *)lw.gclass.p, lw.wk_g.p, lw.wk_e.p, lw.wk_items.p);
  *n_work_max = 2 * lw.n_groups + TEHMM_PLACE_NE;
  return TEHMM_OK;
}

template <int NT>
static void launch_vit_stitch(tehmm_batch *b, const tehmm_model *m, const IntervalTab &iv, const VitChunks &vc,
                              hipStream_t st) {
  LaneWork &lw = b->lw;
  hipLaunchKernelGGL((k_vit_stitch<NT>), dim3((vc.n + 3) / 4), dim3(256), 0, st, iv, lane_geom(lw), vc,
                     lane_vit_items(lw), m->N, lw.soft_ties);
  const char *vl = std::getenv("TEHMM_VIT_RUNS");        // 0: one verification per chunk
  if (vl && std::atoi(vl) == 0)
    (void)hipMemsetAsync(vc.clink, 0, (size_t)vc.n * sizeof(int), st);
  else
    hipLaunchKernelGGL((k_vit_links<NT>), dim3((vc.n + 3) / 4), dim3(256), 0, st, iv, lane_geom(lw), vc,
                       lane_vit_items(lw), m->N);
  hipLaunchKernelGGL(k_vit_runs, dim3(std::max(1, b->n)), dim3(64), 0, st, vc, b->n);
}

static LaneGeom lane_geom(const LaneWork &lw) {
  LaneGeom lg;
  lg.item_iv = lw.item_iv.p; lg.item_t0 = lw.item_t0.p; lg.ifirst = lw.ifirst.p;
  lg.n_items = lw.n_items; lg.n_groups = lw.n_groups; lg.L = lw.L;
  return lg;
}

template <int NT>
static void launch_emis_lane(tehmm_batch *b, const tehmm_model *m, const IntervalTab &iv, const EmisTab &em,
                             bool want_log, bool want_lin, bool want_f32, hipStream_t st) {
  LaneWork &lw = b->lw;
  const size_t lds = (size_t)em.lds_rows * NT * sizeof(double);
  allow_lds(k_emis_lane<NT>, lds);
  hipLaunchKernelGGL((k_emis_lane<NT>), dim3((lw.n_groups + 3) / 4), dim3(256), lds, st, iv, em, lane_geom(lw),
                     m->N, want_log ? lw.B.p : (double *)nullptr, want_lin ? lw.BH.p : (double *)nullptr, lw.MS.p,
                     want_f32 ? lw.B32.p : (float *)nullptr);
}

// emission rows (fp64 log rows for the exact Viterbi pass) and P0 in one pass
template <int NT>
static void launch_emis_gain_lane(tehmm_batch *b, const tehmm_model *m, const IntervalTab &iv, const EmisTab &em,
                                  int CS, int Wu, bool ratio, hipStream_t st) {
  LaneWork &lw = b->lw;
  // round 4: the states of a row split over the three waves of a unit (tehmm_lane3.hip.h); TEHMM_EMIS_SPLIT=0 keeps
  // the one-wave kernel (which also serves the smallest models)
  if constexpr (NT >= 12) {
    // (measured, 36 states: 1.1 Mb 0.98 ms against 1.70; 100 Mb 17.4 against 14.7 -- three waves repeat the observation
    //  decode and the bookkeeping of a position, and a full GPU is short of vector issue slots, not of waves: the split
    //  form serves the batches that leave the one-wave kernel fewer than two waves per SIMD)
    const char *es = std::getenv("TEHMM_EMIS_SPLIT");
    const bool split = es ? std::atoi(es) != 0 : lw.n_groups < 2048;
    if (split) {
      constexpr int NW = TEHMM_P2_NW, NU = 2;
      const size_t lds3 = Emis3Geom<NT, NW, NU>::lds_bytes(em.lds_rows);
      const dim3 grid3((lw.n_groups + NU - 1) / NU);
      if constexpr (NT <= TEHMM_RATIO_LANE_MAX) if (ratio) {
        allow_lds(k_emis_gain_lane3<NT, NW, NU, true>, lds3);
        hipLaunchKernelGGL((k_emis_gain_lane3<NT, NW, NU, true>), grid3, dim3(64 * NW * NU), lds3, st, iv, em, lane_geom(lw),
                           m->N, CS, Wu, (const float *)m->ltP.p, lw.B.p, lw.vgain.p, (const double *)b->ratios.p);
        return;
      }
      allow_lds(k_emis_gain_lane3<NT, NW, NU, false>, lds3);
      hipLaunchKernelGGL((k_emis_gain_lane3<NT, NW, NU, false>), grid3, dim3(64 * NW * NU), lds3, st, iv, em, lane_geom(lw),
                         m->N, CS, Wu, (const float *)m->ltP.p, lw.B.p, lw.vgain.p, (const double *)nullptr);
      return;
    }
  }
  const size_t lds = (size_t)em.lds_rows * NT * sizeof(double);
  if constexpr (NT <= TEHMM_RATIO_LANE_MAX) if (ratio) {
    allow_lds(k_emis_gain_lane<NT, true>, lds);
    hipLaunchKernelGGL((k_emis_gain_lane<NT, true>), dim3((lw.n_groups + 3) / 4), dim3(256), lds, st, iv, em,
                       lane_geom(lw), m->N, CS, Wu, (const float *)m->ltP.p, lw.B.p, lw.vgain.p,
                       (const double *)b->ratios.p);
    return;
  }
  allow_lds(k_emis_gain_lane<NT>, lds);
  hipLaunchKernelGGL((k_emis_gain_lane<NT>), dim3((lw.n_groups + 3) / 4), dim3(256), lds, st, iv, em, lane_geom(lw),
                     m->N, CS, Wu, (const float *)m->ltP.p, lw.B.p, lw.vgain.p, (const double *)nullptr);
}

template <int NT>
static void launch_gain_lane(tehmm_batch *b, const tehmm_model *m, const IntervalTab &iv, int CS, int Wu,
                             hipStream_t st) {
  LaneWork &lw = b->lw;
  hipLaunchKernelGGL((k_vit_gain_lane<NT>), dim3((lw.n_groups + 3) / 4), dim3(256), 0, st, iv, lane_geom(lw), m->N,
                     CS, Wu, (const float *)m->ltP.p, (const float *)lw.B32.p, lw.vgain.p);
}

// Forward lane pass -> [forward links, runs, forward fix-up chain on the side stream]  ||  backward lane pass ->
// backward links, runs, backward fix-up chain; joined before the combine.  ev_mid is recorded between the
// lane passes and the backward fix-up chain (stage timing).
template <int NT>
static void launch_fb_lane(tehmm_batch *b, const tehmm_model *m, const IntervalTab &iv, const EmisTab &em_in,
                           const FbChunks &fc, int Wu, hipStream_t st, hipEvent_t ev_mid) {
  LaneWork &lw = b->lw;
  const LaneGeom lg = lane_geom(lw);
  const dim3 grid((lw.n_groups + 3) / 4);
  const dim3 gridm((unsigned)lw.n_groups);           // 4 tiles of 16 items per 256-thread block = one group
  const dim3 gridc((fc.n + 255) / 256);              // one thread per chunk
  const dim3 gridit((lw.n_items + 255) / 256);       // one thread per item
  const dim3 gridi(std::max(1, b->n));
  // TEHMM_LANE_MFMA: 0 = the VALU form (default), 1 = the fp64 matrix-core form
  const char *mfs = std::getenv("TEHMM_LANE_MFMA");
  const bool mf = mfs && std::atoi(mfs) != 0;
  const char *er = std::getenv("TEHMM_FB_RUNS");
  const int extend = (er && std::atoi(er) == 0) ? 0 : 1;
  const EmisTab em = b->n > 256 ? without_lds_tables(em_in) : em_in;
  size_t lds = ((size_t)2 * 64 * (NT + 1) + 2 * 64 + NT + (size_t)em.lds_rows * NT + 8) * sizeof(double);
  allow_lds(k_fb_fix<NT, 0, false, true>, lds);
  allow_lds(k_fb_fix<NT, 1, false, true>, lds);
  hipStream_t sS = b->sB;
  // ---- forward
  if (mf)
    hipLaunchKernelGGL((k_fb_mfma<NT, 0>), gridm, dim3(256), 0, st, iv, lg, m->N, fc.CS, Wu, m->A.p, lw.BH.p,
                       lw.MS.p, lw.AL.p, lw.pre_f.p, lw.end_f.p, lw.slog32.p);
  else
    hipLaunchKernelGGL((k_fb_lane<NT, 0>), grid, dim3(256), 0, st, iv, lg, m->N, fc.CS, Wu, m->AG.p, lw.BH.p,
                       lw.MS.p, lw.AL.p, lw.pre_f.p, lw.end_f.p, lw.slog32.p);
  (void)hipEventRecord(b->evX[0], st);
  (void)hipStreamWaitEvent(sS, b->evX[0], 0);
  hipLaunchKernelGGL((k_fb_itemlinks<NT>), gridit, dim3(256), 0, sS, lg, m->N, lw.pre_f.p, lw.end_f.p, lw.pre_b.p,
                     lw.end_b.p, lw.dl_f.p, lw.lr_f.p, lw.dl_b.p, 1);
  hipLaunchKernelGGL((k_fb_stitch<NT>), gridc, dim3(256), 0, sS, iv, lg, fc, m->N, lw.slog32.p, lw.end_b.p,
                     lw.dl_f.p, lw.lr_f.p, lw.dl_b.p, lw.ok_f.p, lw.ok_b.p, 1);
  hipLaunchKernelGGL(k_fb_runs, gridi, dim3(64), 0, sS, iv, fc, (const int *)lw.ok_f.p, (const int *)lw.ok_b.p,
                     extend, 1);
  hipLaunchKernelGGL((k_fb_fix<NT, 0, false, true>), dim3(b->n), dim3(128), lds, sS, iv, em, fc, m->N, m->A.p,
                     m->lt.p, m->pi.p, (const double *)nullptr, lw.AL.p, b->fwd_lp.p, b->dead.p,
                     (double *)nullptr, (int *)nullptr, 1, b->sw.stats.p, lg, (const int *)lw.ok_f.p);
  (void)hipEventRecord(b->evX[1], sS);
  // ---- backward
  if (mf)
    hipLaunchKernelGGL((k_fb_mfma<NT, 1>), gridm, dim3(256), 0, st, iv, lg, m->N, fc.CS, Wu, m->A.p, lw.BH.p,
                       lw.MS.p, lw.BE.p, lw.pre_b.p, lw.end_b.p, (double *)nullptr);
  else
    hipLaunchKernelGGL((k_fb_lane<NT, 1>), grid, dim3(256), 0, st, iv, lg, m->N, fc.CS, Wu, m->ATG.p, lw.BH.p,
                       lw.MS.p, lw.BE.p, lw.pre_b.p, lw.end_b.p, (double *)nullptr);
  hipLaunchKernelGGL((k_fb_itemlinks<NT>), gridit, dim3(256), 0, st, lg, m->N, lw.pre_f.p, lw.end_f.p, lw.pre_b.p,
                     lw.end_b.p, lw.dl_f.p, lw.lr_f.p, lw.dl_b.p, 2);
  hipLaunchKernelGGL((k_fb_stitch<NT>), gridc, dim3(256), 0, st, iv, lg, fc, m->N, lw.slog32.p, lw.end_b.p,
                     lw.dl_f.p, lw.lr_f.p, lw.dl_b.p, lw.ok_f.p, lw.ok_b.p, 2);
  hipLaunchKernelGGL(k_fb_runs, gridi, dim3(64), 0, st, iv, fc, (const int *)lw.ok_f.p, (const int *)lw.ok_b.p,
                     extend, 2);
  (void)hipEventRecord(ev_mid, st);
  hipLaunchKernelGGL((k_fb_fix<NT, 1, false, true>), dim3(b->n), dim3(128), lds, st, iv, em, fc, m->N, m->A.p,
                     m->lt.p, m->pi.p, (const double *)nullptr, lw.BE.p, b->fwd_lp.p, b->dead.p,
                     (double *)nullptr, (int *)nullptr, 1, b->sw.stats.p, lg, (const int *)lw.ok_b.p);
  (void)hipStreamWaitEvent(st, b->evX[1], 0);
}

// Fused posterior pipeline (tehmm_fused.hip.h), one stream: forward lane pass -> forward links / runs / exact
// chain (alpha' rows final) -> backward lane pass writing the posterior rows -> backward links / runs ->
// [ev_mid] -> backward exact chain (posterior rows of its exact blocks).  ev_fwd: end of the forward half.
// Index records of the fused passes (FusedTab::rixx): the track order of EmisStream's fixed schedule and, unless
// the records of this (batch, model layout, item length, warm-up) exist already, k_fused_rowindex on `st`.
// Depends on the observations only, so tehmm_eval_batch enqueues it ahead of the deferral behind the Viterbi
// passes: the 3 ms of a first evaluation hide next to the emission-row kernel.
template <int NT>
static int fused_prepare(tehmm_batch *b, const tehmm_model *m, const IntervalTab &iv, int Wu, hipStream_t st, FusedOrder &fo) {
  LaneWork &lw = b->lw;
  const LaneGeom lg = lane_geom(lw);
  std::memset(&fo, 0, sizeof(fo));
  fo.K = m->K;
  fo.KP = b->KP;
  fo.Wu = Wu;
  fo.zero_glb = m->R;
  fo.zero_lds = m->lds_zero;
  // the fixed schedule of EmisStream: at least NGS tracks from the global table, then at least NSLOT from LDS;
  // the padding entries (cnt = 0) always read the identity row
  {
    constexpr int NSLOT = EmisStream<NT, false>::NSLOT, NGS = EmisStream<NT, false>::NGS;
    int slot = 0;
    auto pad = [&]() { fo.order[slot] = 0; fo.base[slot] = 0; fo.cnt[slot] = 0; ++slot; };
    for (int k = 0; k < m->K; ++k)
      if (m->ldsbase[k] < 0) { fo.order[slot] = (unsigned char)k; fo.base[slot] = m->rowbase[k]; fo.cnt[slot] = m->rowcnt[k]; ++slot; }
    while (slot < NGS) pad();
    fo.n_glb = slot;
    for (int k = 0; k < m->K; ++k)
      if (m->ldsbase[k] >= 0) { fo.order[slot] = (unsigned char)k; fo.base[slot] = m->ldsbase[k]; fo.cnt[slot] = m->rowcnt[k]; ++slot; }
    while (slot - fo.n_glb < NSLOT) pad();
    fo.K = slot;
  }
  {
    static const int ok[] = {1, 2, 3, 4, 6, 8, 12, 24};          // divisors of TEHMM_FUSED_BLKW
    int need = (fo.K + 3) / 4;
    fo.FKW = 24;
    for (int d : ok) if (d >= need) { fo.FKW = d; break; }
  }
  fo.SB = TEHMM_FUSED_BLKW / fo.FKW;
  fo.NB = (lw.L + 2 * Wu + fo.SB - 1) / fo.SB;
  if (!lw.rix.p || lw.rix_model != m->uid || lw.rix_L != lw.L || lw.rix_Wu != Wu) {
    const size_t words = (size_t)lw.n_groups * 4 * fo.NB * TEHMM_FUSED_BLKW * 16;
    lw.rix_model = 0;                                   // (the key is set once the records exist)
    HIPCHK(lw.rix.ensure(words + 16));
    hipLaunchKernelGGL(k_fused_rowindex, dim3(grid_for((int64_t)lw.n_groups * 4 * fo.NB * fo.SB * 16, 256, 1 << 20)),
                       dim3(256), 0, st, iv, lg, fo, (const uint8_t *)b->obs.p, lw.rix.p);
    lw.rix_model = m->uid;
    lw.rix_L = lw.L;
    lw.rix_Wu = Wu;
  }
  return TEHMM_OK;
}

template <int NT>
static int launch_fused_fb(tehmm_batch *b, const tehmm_model *m, const IntervalTab &iv, const EmisTab &em_in,
                           const FbChunks &fc, int Wu, hipStream_t st, hipEvent_t ev_fwd, hipEvent_t ev_mid,
                           bool estep = false, int half = 0, hipEvent_t ev_fwdk = nullptr) {
  // half: 0 = the whole pipeline; 1 = the forward half only (pass, links, exact forward chain; ev_fwdk is recorded right
  // behind the pass itself), 2 = the backward half only -- tehmm_eval_batch puts the quantised Viterbi pass between them
  LaneWork &lw = b->lw;
  const LaneGeom lg = lane_geom(lw);
  const dim3 gridm((unsigned)lw.n_groups);           // 4 tiles of 16 items per 256-thread block = one group
  const dim3 gridc((fc.n + 255) / 256);              // one thread per chunk
  const dim3 gridit((lw.n_items + 255) / 256);       // one thread per item
  const dim3 gridi(std::max(1, b->n));
  const char *er = std::getenv("TEHMM_FB_RUNS");
  const int extend = (er && std::atoi(er) == 0) ? 0 : 1;
  FusedOrder fo;
  if (int rcp = fused_prepare<NT>(b, m, iv, Wu, st, fo)) return rcp;
  FusedTab ft;
  ft.rixx = lw.rix.p;
  ft.ptab = m->ptab.p;
  ft.ptab_lds = m->ptab_lds.p;
  ft.FKW = fo.FKW;
  ft.SB = fo.SB;
  ft.NB = fo.NB;
  ft.K = fo.K;
  ft.n_glb = fo.n_glb;
  ft.lds_rows = m->lds_rows;
  ft.normalize = m->normalize;
  const EmisTab emc = b->n > 256 ? without_lds_tables(em_in) : em_in;          // the chains' own emission rows
  const size_t lds_f = fused_lds_bytes<NT>(m->lds_rows, false), lds_b = fused_lds_bytes<NT>(m->lds_rows, true);
  const size_t lds_c = ((size_t)2 * 64 * (NT + 1) + 2 * 64 + NT + (size_t)emc.lds_rows * NT + 8) * sizeof(double);
  allow_lds(k_fb_fix<NT, 0, false, true, true>, lds_c);
  allow_lds(k_fb_fix<NT, 1, false, true, true>, lds_c);
#define TEHMM_FUSED_LAUNCH(LOG_)                                                                                     \
  do {                                                                                                              \
    allow_lds(k_fused_fwd<NT, LOG_>, lds_f);                                                                         \
    allow_lds(k_fused_bwd<NT, LOG_, true>, lds_b);                                                                   \
    if (half != 2) {                                                                                                \
    hipLaunchKernelGGL((k_fused_fwd<NT, LOG_>), gridm, dim3(256), lds_f, st, iv, ft, lg, m->N, fc.CS, Wu,            \
                       (const double *)m->A.p, lw.AL32.p, lw.chkf.p, lw.pre_f.p, lw.end_f.p, lw.slog32.p);           \
    if (ev_fwdk) (void)hipEventRecord(ev_fwdk, st);                                                                  \
    hipLaunchKernelGGL((k_fb_itemlinks<NT>), gridit, dim3(256), 0, st, lg, m->N, lw.pre_f.p, lw.end_f.p,             \
                       lw.pre_b.p, lw.end_b.p, lw.dl_f.p, lw.lr_f.p, lw.dl_b.p, 1);                                  \
    hipLaunchKernelGGL((k_fb_stitch<NT>), gridc, dim3(256), 0, st, iv, lg, fc, m->N, lw.slog32.p, lw.end_b.p,        \
                       lw.dl_f.p, lw.lr_f.p, lw.dl_b.p, lw.ok_f.p, lw.ok_b.p, 1);                                    \
    hipLaunchKernelGGL(k_fb_runs, gridi, dim3(64), 0, st, iv, fc, (const int *)lw.ok_f.p, (const int *)lw.ok_b.p,    \
                       extend, 1);                                                                                   \
    hipLaunchKernelGGL((k_fb_fix<NT, 0, false, true, true>), dim3(b->n), dim3(128), lds_c, st, iv, emc, fc, m->N,    \
                       m->A.p, m->lt.p, m->pi.p, (const double *)nullptr, (double *)nullptr, b->fwd_lp.p, b->dead.p, \
                       (double *)nullptr, (int *)nullptr, 1, b->sw.stats.p, lg, (const int *)lw.ok_f.p,              \
                       (const double *)lw.chkf.p, (double *)nullptr, lw.AL32.p, (const double *)lw.end_f.p);         \
    (void)hipEventRecord(ev_fwd, st);                                                                                \
    }                                                                                                               \
    if (half == 1) break;                                                                                           \
    if (estep) {                                                                                                    \
      allow_lds(k_fused_bwd<NT, LOG_, false, true>, lds_f);                                                          \
      hipLaunchKernelGGL((k_fused_bwd<NT, LOG_, false, true>), gridm, dim3(256), lds_f, st, iv, ft, lg, m->N, fc.CS, \
                         Wu, (const double *)m->A.p, (const float *)lw.AL32.p, (double *)nullptr, lw.pre_b.p,        \
                         lw.end_b.p, lw.chk.p, lw.GAM32.p, lw.WZ32.p);                                               \
    } else {                                                                                                        \
      hipLaunchKernelGGL((k_fused_bwd<NT, LOG_, true>), gridm, dim3(256), lds_b, st, iv, ft, lg, m->N, fc.CS,         \
                         Wu, (const double *)m->A.p, (const float *)lw.AL32.p, b->post.p, lw.pre_b.p, lw.end_b.p,    \
                         lw.chk.p, (float *)nullptr, (float *)nullptr, lw.sink.p);                                   \
    }                                                                                                               \
  } while (0)
  if (m->ptab_log) TEHMM_FUSED_LAUNCH(true);
  else TEHMM_FUSED_LAUNCH(false);
#undef TEHMM_FUSED_LAUNCH
  if (half == 1) {
    HIPCHK(hipGetLastError());
    return TEHMM_OK;
  }
  hipLaunchKernelGGL((k_fb_itemlinks<NT>), gridit, dim3(256), 0, st, lg, m->N, lw.pre_f.p, lw.end_f.p, lw.pre_b.p,
                     lw.end_b.p, lw.dl_f.p, lw.lr_f.p, lw.dl_b.p, 2);
  hipLaunchKernelGGL((k_fb_stitch<NT>), gridc, dim3(256), 0, st, iv, lg, fc, m->N, lw.slog32.p, lw.end_b.p,
                     lw.dl_f.p, lw.lr_f.p, lw.dl_b.p, lw.ok_f.p, lw.ok_b.p, 2);
  hipLaunchKernelGGL(k_fb_runs, gridi, dim3(64), 0, st, iv, fc, (const int *)lw.ok_f.p, (const int *)lw.ok_b.p,
                     extend, 2);
  (void)hipEventRecord(ev_mid, st);
  if (estep) {
    const size_t lds_e = lds_c + (size_t)64 * (NT + 1) * sizeof(double);
    allow_lds(k_fb_fix<NT, 1, false, true, true, true>, lds_e);
    hipLaunchKernelGGL((k_fb_fix<NT, 1, false, true, true, true>), dim3(b->n), dim3(128), lds_e, st, iv, emc, fc, m->N,
                       m->A.p, m->lt.p, m->pi.p, (const double *)nullptr, (double *)nullptr, b->fwd_lp.p, b->dead.p,
                       (double *)nullptr, (int *)nullptr, 1, b->sw.stats.p, lg, (const int *)lw.ok_b.p,
                       (const double *)lw.chk.p, (double *)nullptr, lw.AL32.p, (const double *)nullptr, lw.GAM32.p,
                       lw.WZ32.p);
    return TEHMM_OK;
  }
  hipLaunchKernelGGL((k_fb_fix<NT, 1, false, true, true>), dim3(b->n), dim3(128), lds_c, st, iv, emc, fc, m->N, m->A.p,
                     m->lt.p, m->pi.p, (const double *)nullptr, (double *)nullptr, b->fwd_lp.p, b->dead.p,
                     (double *)nullptr, (int *)nullptr, 1, b->sw.stats.p, lg, (const int *)lw.ok_b.p,
                     (const double *)lw.chk.p, b->post.p, lw.AL32.p, (const double *)nullptr);
  return TEHMM_OK;
}

template <int NT>
static void launch_combine_lane(tehmm_batch *b, const tehmm_model *m, const IntervalTab &iv, hipStream_t st) {
  LaneWork &lw = b->lw;
  hipLaunchKernelGGL((k_combine_lane<NT, true>), dim3((unsigned)lw.n_groups * (lw.L / 8)), dim3(256), 0, st, iv,
                     lane_geom(lw), m->N, lw.AL.p, lw.BE.p, b->post.p);
}

template <int NT>
static void launch_fb_spec(tehmm_batch *b, const tehmm_model *m, const IntervalTab &iv, const EmisTab &em,
                           const FbChunks &fc, hipStream_t st) {
  size_t lds = ((size_t)4 * 64 * (NT + 1) + 4 * 64) * sizeof(double);
  const int grid = std::min((fc.n + 3) / 4, 4096);
  allow_lds(k_fb_spec<NT, 0>, lds);
  allow_lds(k_fb_spec<NT, 1>, lds);
  hipLaunchKernelGGL((k_fb_spec<NT, 0>), dim3(grid), dim3(256), lds, st, iv, em, fc, m->N, m->A.p, b->post.p);
  hipLaunchKernelGGL((k_fb_spec<NT, 1>), dim3(grid), dim3(256), lds, st, iv, em, fc, m->N, m->A.p, b->beta.p);
}

template <int NT>
static void launch_fb_fix(tehmm_batch *b, const tehmm_model *m, const IntervalTab &iv, const EmisTab &em_in,
                          const FbChunks &fc, hipStream_t sF, hipStream_t sBk) {
  const EmisTab em = b->n > 256 ? without_lds_tables(em_in) : em_in;
  size_t lds = ((size_t)2 * 64 * (NT + 1) + 2 * 64 + NT + (size_t)em.lds_rows * NT + 8) * sizeof(double);
  allow_lds(k_fb_fix<NT, 0, false, false>, lds);
  allow_lds(k_fb_fix<NT, 1, false, false>, lds);
  hipLaunchKernelGGL((k_fb_fix<NT, 0, false, false>), dim3(b->n), dim3(128), lds, sF, iv, em, fc, m->N, m->A.p,
                     m->lt.p, m->pi.p, (const double *)nullptr, b->post.p, b->fwd_lp.p, b->dead.p,
                     (double *)nullptr, (int *)nullptr, 1, b->sw.stats.p, LaneGeom(), (const int *)nullptr);
  hipLaunchKernelGGL((k_fb_fix<NT, 1, false, false>), dim3(b->n), dim3(128), lds, sBk, iv, em, fc, m->N, m->A.p,
                     m->lt.p, m->pi.p, (const double *)nullptr, b->beta.p, b->fwd_lp.p, b->dead.p,
                     (double *)nullptr, (int *)nullptr, 1, b->sw.stats.p, LaneGeom(), (const int *)nullptr);
}

// -DTEHMM_DEV_NT=36: development builds that instantiate the fused kernels for one padded state
// count only (seconds instead of minutes to compile); never used for the shipped library.
#ifdef TEHMM_DEV_NT
#define TEHMM_NT_DISPATCH(NP_, CALL)                                                                \
  switch (NP_) {                                                                                    \
    case TEHMM_DEV_NT: CALL(TEHMM_DEV_NT); break;                                                   \
    default: break;                                                                                 \
  }
#else
#define TEHMM_NT_DISPATCH(NP_, CALL)                                                                \
  switch (NP_) {                                                                                    \
    case 4: CALL(4); break;   case 8: CALL(8); break;   case 12: CALL(12); break;                   \
    case 16: CALL(16); break; case 20: CALL(20); break; case 24: CALL(24); break;                   \
    case 28: CALL(28); break; case 32: CALL(32); break; case 36: CALL(36); break;                   \
    case 40: CALL(40); break; case 48: CALL(48); break; case 56: CALL(56); break;                   \
    case 64: CALL(64); break;                                                                       \
    default: break;                                                                                 \
  }
#endif



// ---- chunk-parallel posterior for 64 <= N <= 128 (tehmm_wide.hip.h) ---------------------------------------------
template <int NPW>
static void launch_wide_passes(tehmm_batch *b, const tehmm_model *m, const IntervalTab &iv, const LaneGeom &lg, int Wu,
                               hipStream_t st, hipEvent_t mid) {
  WideWork &w = b->ww;
  const size_t lds = WideGeom<NPW>::FRAG_BYTES;
  allow_lds(k_wide_fwd<NPW>, lds);
  allow_lds(k_wide_bwd<NPW>, lds);
  const dim3 grid((unsigned)std::max(1, w.n_groups));
  hipLaunchKernelGGL((k_wide_fwd<NPW>), grid, dim3(256), lds, st, iv, lg, m->N, m->NP, Wu, (const double *)m->A.p,
                     (const double *)m->pi.p, (const double *)w.E.p, (const double *)w.ms.p, w.AL.p, w.pre_f.p, w.end_f.p,
                     w.SL.p);
  (void)hipEventRecord(mid, st);
  hipLaunchKernelGGL((k_wide_bwd<NPW>), grid, dim3(256), lds, st, iv, lg, m->N, m->NP, Wu, (const double *)m->A.p,
                     (const double *)w.E.p, (const float *)w.AL.p, b->post.p, w.pre_b.p, w.end_b.p);
}

// the E-step form (tehmm_wide_estep.hip.h): the backward pass leaves gamma / wz rows instead of posteriors
template <int NPW>
static void launch_wide_passes_estep(tehmm_batch *b, const tehmm_model *m, const IntervalTab &iv, const LaneGeom &lg, int Wu,
                                     hipStream_t st, hipEvent_t mid) {
  WideWork &w = b->ww;
  const size_t lds = WideGeom<NPW>::FRAG_BYTES;
  allow_lds(k_wide_fwd<NPW>, lds);
  allow_lds(k_wide_bwd<NPW, true>, lds);
  const dim3 grid((unsigned)std::max(1, w.n_groups));
  hipLaunchKernelGGL((k_wide_fwd<NPW>), grid, dim3(256), lds, st, iv, lg, m->N, m->NP, Wu, (const double *)m->A.p,
                     (const double *)m->pi.p, (const double *)w.E.p, (const double *)w.ms.p, w.AL.p, w.pre_f.p, w.end_f.p,
                     w.SL.p);
  (void)hipEventRecord(mid, st);
  hipLaunchKernelGGL((k_wide_bwd<NPW, true>), grid, dim3(256), lds, st, iv, lg, m->N, m->NP, Wu, (const double *)m->A.p,
                     (const double *)w.E.p, (const float *)w.AL.p, (double *)nullptr, w.pre_b.p, w.end_b.p, w.GAM.p, w.WZ.p);
}

// One attempt of the chunk-parallel posterior with warm-up Wu, enqueued on st: passes, link checks, and the flags
// (impossible rows, failed links) on their way to pinned host memory.
static int wide_post_attempt(tehmm_batch *b, const tehmm_model *m, const IntervalTab &iv, int Wu, hipStream_t st, hipEvent_t mid,
                             bool estep = false) {
  WideWork &w = b->ww;
  LaneGeom lg;
  lg.item_iv = w.item_iv.p; lg.item_t0 = w.item_t0.p; lg.ifirst = w.ifirst.p;
  lg.n_items = w.n_items; lg.n_groups = w.n_groups; lg.L = w.L;
  const int NPW = w.NPW;
  if (estep) {
    switch (NPW) {
      case 16: launch_wide_passes_estep<16>(b, m, iv, lg, Wu, st, mid); break;
      case 32: launch_wide_passes_estep<32>(b, m, iv, lg, Wu, st, mid); break;
      case 48: launch_wide_passes_estep<48>(b, m, iv, lg, Wu, st, mid); break;
      case 64: launch_wide_passes_estep<64>(b, m, iv, lg, Wu, st, mid); break;
      case 80: launch_wide_passes_estep<80>(b, m, iv, lg, Wu, st, mid); break;
      case 96: launch_wide_passes_estep<96>(b, m, iv, lg, Wu, st, mid); break;
      case 112: launch_wide_passes_estep<112>(b, m, iv, lg, Wu, st, mid); break;
      default: launch_wide_passes_estep<128>(b, m, iv, lg, Wu, st, mid); break;
    }
  } else
  switch (NPW) {
    case 80: launch_wide_passes<80>(b, m, iv, lg, Wu, st, mid); break;
    case 96: launch_wide_passes<96>(b, m, iv, lg, Wu, st, mid); break;
    case 112: launch_wide_passes<112>(b, m, iv, lg, Wu, st, mid); break;
    default: launch_wide_passes<128>(b, m, iv, lg, Wu, st, mid); break;
  }
  // link tolerance (Hilbert distance between the vector an item arrives with and the one its neighbour left): 1e-8 -- the
  // alpha' rows between the passes are floats (6e-8 each, posteriors observed at 1.2e-7 of the reference, bar 1e-6), and an
  // item's error is the link's distance at its first position, smaller further in.  Round 3 asked for 1e-10: twice the
  // warm-up at 100 states for nothing a float row can show.
  double tol = 1e-8;
  if (const char *ts = std::getenv("TEHMM_WIDE_TOL")) tol = std::min(1e-6, std::max(1e-13, std::atof(ts)));
  (void)estep;
  hipLaunchKernelGGL(k_wide_links, dim3((w.n_items + 255) / 256), dim3(256), 0, st, iv, lg, m->N, NPW,
                     (const double *)w.pre_f.p, (const double *)w.end_f.p, (const double *)w.pre_b.p,
                     (const double *)w.end_b.p, w.lr.p, w.flags.p, tol);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(w.h_flags, w.flags.p, 4 * sizeof(int), hipMemcpyDeviceToHost, st));
  w.pp_Wu = Wu;
  w.pp_active = true;
  return TEHMM_OK;
}

// The attempt's verdict (waits for st): links that do not verify double the warm-up and run again, up to 1024
// positions.  *done = true: posteriors and forward log-likelihoods of the batch are in place; false: the caller runs
// the sequential kernels (a link that does not verify within the longest warm-up, impossible rows).
static int posterior_wide_finish(tehmm_batch *b, const tehmm_model *m, const IntervalTab &iv, hipStream_t st, hipEvent_t mid,
                                 bool *done) {
  *done = false;
  WideWork &w = b->ww;
  if (!w.pp_active) return TEHMM_OK;
  w.pp_active = false;
  constexpr int kWuMax = 1024;
  for (;;) {
    HIPCHK(hipStreamSynchronize(st));
    const int Wu = w.pp_Wu;
    if (std::getenv("TEHMM_SPEC_DEBUG"))
      std::fprintf(stderr, "[tehmm wide] NPW %d L %d Wu %d: impossible rows in %d items, failed links %d of %d items\n", w.NPW, w.L,
                   Wu, w.h_flags[0], w.h_flags[1], w.n_items);
    if (w.h_flags[0] > 0) return TEHMM_OK;             // impossible rows: the sequential kernels own their semantics
    if (w.h_flags[1] == 0) break;
    if (Wu >= kWuMax) return TEHMM_OK;                 // does not forget: sequential kernels
    HIPCHK(hipMemsetAsync(w.flags.p, 0, 4 * sizeof(int), st));
    if (int rc = wide_post_attempt(b, m, iv, std::min(kWuMax, 2 * Wu), st, mid)) return rc;
    w.pp_active = false;
  }
  w.wu_ok = w.pp_Wu;
  w.wu_model = m->uid;
  w.wu_version = m->version;
  LaneGeom lg;
  lg.item_iv = w.item_iv.p; lg.item_t0 = w.item_t0.p; lg.ifirst = w.ifirst.p;
  lg.n_items = w.n_items; lg.n_groups = w.n_groups; lg.L = w.L;
  hipLaunchKernelGGL(k_wide_loglik, dim3((b->n + 3) / 4), dim3(256), 0, st, iv, lg, m->N, w.NPW, (const double *)w.end_f.p,
                     (const double *)w.SL.p, (const double *)w.lr.p, b->fwd_lp.p);
  *done = true;
  return TEHMM_OK;
}

// emission rows of the item-parallel passes in the tile layout (k_wide_emis_tile): mode 0 tables, 1 from the log rows
// of the exact Viterbi, 2 fit, 3 fit with segment ratios
template <int NPW>
static void launch_wide_emis_npw(tehmm_batch *b, const tehmm_model *m, const IntervalTab &iv, const EmisTab &em_in, const LaneGeom &lg,
                                 int mode, const double *log_rows, hipStream_t st) {
  WideWork &w = b->ww;
  // this kernel's own LDS assignment: smallest tracks first while they fit (two workgroups per CU)
  EmisTab em = em_in;
  {
    // (tables + the redistribution buffers of the four waves within 76 KB)
    const size_t budget_rows = ((size_t)(76 * 1024) - (size_t)4 * 8 * wide_emis_tstride(NPW) * sizeof(double)) / ((size_t)m->NP * sizeof(double));
    std::vector<int> ord((size_t)m->K);
    std::iota(ord.begin(), ord.end(), 0);
    std::stable_sort(ord.begin(), ord.end(), [&](int a, int c) { return m->rowcnt[a] < m->rowcnt[c]; });
    for (int k = 0; k < TEHMM_MAX_TRACKS; ++k) em.ldsbase[k] = -1;
    int used = 0;
    for (int k : ord) {
      if ((size_t)(used + m->rowcnt[k] + 1) > budget_rows) break;
      em.ldsbase[k] = used;
      used += m->rowcnt[k];
    }
    em.lds_zero = used;
    em.lds_rows = used + 1;
  }
  const int64_t n_tiles = ((int64_t)w.n_items + 15) / 16, units = n_tiles * ((w.L + 15) / 16);
  const dim3 grid((unsigned)std::max<int64_t>(1, std::min<int64_t>((units + 3) / 4, 2048)));
  const size_t lds = ((size_t)(mode == 1 ? 0 : std::max(1, em.lds_rows)) * m->NP + NPW + (mode == 1 ? 0 : 4 * 8 * wide_emis_tstride(NPW))) * sizeof(double) +
                     (size_t)3 * m->K * sizeof(int) + 16;
#define EMIS(MODE_)                                                                                                   \
  do {                                                                                                                \
    allow_lds(k_wide_emis_tile<NPW, MODE_>, lds);                                                                     \
    hipLaunchKernelGGL((k_wide_emis_tile<NPW, MODE_>), grid, dim3(256), lds, st, iv, em, lg, m->N, m->NP,             \
                       (const double *)m->lt.p, (const double *)b->ratios.p, log_rows, w.E.p, w.ms.p, w.flags.p);     \
  } while (0)
  switch (mode) {
    case 0: EMIS(0); break;
    case 1: EMIS(1); break;
    case 2: EMIS(2); break;
    default: EMIS(3); break;
  }
#undef EMIS
}
static void launch_wide_emis(tehmm_batch *b, const tehmm_model *m, const IntervalTab &iv, const EmisTab &em, const LaneGeom &lg,
                             int mode, const double *log_rows, hipStream_t st) {
  switch (b->ww.NPW) {
    case 16: launch_wide_emis_npw<16>(b, m, iv, em, lg, mode, log_rows, st); break;
    case 32: launch_wide_emis_npw<32>(b, m, iv, em, lg, mode, log_rows, st); break;
    case 48: launch_wide_emis_npw<48>(b, m, iv, em, lg, mode, log_rows, st); break;
    case 64: launch_wide_emis_npw<64>(b, m, iv, em, lg, mode, log_rows, st); break;
    case 80: launch_wide_emis_npw<80>(b, m, iv, em, lg, mode, log_rows, st); break;
    case 96: launch_wide_emis_npw<96>(b, m, iv, em, lg, mode, log_rows, st); break;
    case 112: launch_wide_emis_npw<112>(b, m, iv, em, lg, mode, log_rows, st); break;
    default: launch_wide_emis_npw<128>(b, m, iv, em, lg, mode, log_rows, st); break;
  }
}

// Item geometry and workspaces of the item-parallel passes for (NPW, L); *fits = false: not enough device memory
// (nothing changed).  The alpha' rows are (re)allocated here: al_zeroed tells the E-step whether the slots no pass
// writes have been cleared since.
static int wide_geometry(tehmm_batch *b, int NPW, int L, bool *fits) {
  WideWork &w = b->ww;
  *fits = true;
  if (w.L != L || w.NPW != NPW || !w.item_iv.p) {
    {
      // emission rows (8 NPW bytes per position) + float alpha' rows (4 NPW): the sequential kernels need neither
      size_t free_b = 0, total_b = 0;
      HIPCHK(dev_mem_info(&free_b, &total_b));
      if ((double)b->total * NPW * 12.5 + (double)(b->total / L + b->n + 64) * NPW * 40.0 > 0.9 * (double)free_b) { *fits = false; return TEHMM_OK; }
    }
    std::vector<int> h_iv;
    std::vector<int64_t> h_t0, h_first((size_t)b->n + 1, 0);
    for (int i = 0; i < b->n; ++i) {
      h_first[(size_t)i] = (int64_t)h_iv.size();
      for (int64_t t0 = 0; t0 < b->h_len[(size_t)i]; t0 += L) {
        h_iv.push_back(i);
        h_t0.push_back(t0);
      }
    }
    h_first[(size_t)b->n] = (int64_t)h_iv.size();
    w.L = L;
    w.NPW = NPW;
    w.n_items = (int)h_iv.size();
    w.n_groups = (w.n_items + 63) / 64;
    w.wu_ok = 0;
    HIPCHK(w.item_iv.upload(h_iv.data(), h_iv.size()));
    HIPCHK(w.item_t0.upload(h_t0.data(), h_t0.size()));
    HIPCHK(w.ifirst.upload(h_first.data(), h_first.size()));
    const size_t ni = (size_t)std::max(1, w.n_groups) * 64;
    HIPCHK(w.E.alloc(ni * L * NPW));           // tile layout (wide_e_row): items padded to whole tiles
    HIPCHK(w.ms.alloc(ni * L));
    HIPCHK(w.AL.alloc(ni * L * (NPW / 4) * 4));
    for (DBuf<double> *d : {&w.pre_f, &w.end_f, &w.pre_b, &w.end_b}) HIPCHK(d->alloc(ni * NPW));
    HIPCHK(w.SL.alloc(ni));
    HIPCHK(w.lr.alloc(ni));
    HIPCHK(w.flags.alloc(4));
    w.al_zeroed = false;
    w.GAM.release();
    w.WZ.release();
  }
  return TEHMM_OK;
}

// Chunk-parallel posterior for 64 <= N <= 128: geometry, emission rows and the first attempt are ENQUEUED here
// (*pending = true; nothing enqueued otherwise: short batches, TEHMM_WIDE_CP=0); posterior_wide_finish delivers the
// verdict, so the exact Viterbi of the same evaluation can be enqueued in between and share the GPU with the passes.
static int posterior_wide_cp(tehmm_batch *b, const tehmm_model *m, const IntervalTab &iv, const EmisTab &em, hipStream_t st,
                             hipEvent_t mid, bool *pending, const double *log_rows = nullptr) {
  *pending = false;
  const char *ws = std::getenv("TEHMM_WIDE_CP");
  if (ws && std::atoi(ws) == 0) return TEHMM_OK;
  if (m->N < 64 || m->N > 128 || b->total < 4096) return TEHMM_OK;
  static const int sizes[] = {80, 96, 112, 128};
  int NPW = 128;
  for (int sz : sizes)
    if (m->N <= sz) { NPW = sz; break; }
  WideWork &w = b->ww;
  if (!w.h_flags) HIPCHK(hipHostMalloc((void **)&w.h_flags, 64, hipHostMallocDefault));
  // item length: enough items to give every SIMD a tile (1024 tiles of 16 items), 64 <= L <= 512, multiple of 32
  int L = (int)std::min<int64_t>(512, std::max<int64_t>(64, (b->total / (16 * 1024) + 31) & ~31));
  if (const char *ls = std::getenv("TEHMM_WIDE_SUB")) L = std::max(32, (std::atoi(ls) + 31) & ~31);
  {
    bool fits = true;
    if (int rcg = wide_geometry(b, NPW, L, &fits)) return rcg;
    if (!fits) return TEHMM_OK;
  }
  LaneGeom lg;
  lg.item_iv = w.item_iv.p; lg.item_t0 = w.item_t0.p; lg.ifirst = w.ifirst.p;
  lg.n_items = w.n_items; lg.n_groups = w.n_groups; lg.L = L;
  HIPCHK(hipMemsetAsync(w.flags.p, 0, 4 * sizeof(int), st));
  // (log_rows: the exact Viterbi of this evaluation has the log rows already (k_wide_logrows): no second gather)
  launch_wide_emis(b, m, iv, em, lg, log_rows ? 1 : 0, log_rows, st);
  // warm-up: 64 positions to begin with (it may exceed the item length: the passes read the emission rows of the
  // interval, not of the item), or what the last evaluation with this model needed
  constexpr int kWuMax = 1024;
  int Wu = 64;
  if (const char *wus = std::getenv("TEHMM_LANE_WARMUP")) Wu = std::min(kWuMax, std::max(1, std::atoi(wus)));
  else if (w.wu_ok > 0 && w.wu_model == m->uid && w.wu_version == m->version) Wu = w.wu_ok;
  if (int rc = wide_post_attempt(b, m, iv, Wu, st, mid)) return rc;
  *pending = true;
  return TEHMM_OK;
}


// ---- chunk-parallel exact Viterbi for 64 <= N <= 128 (tehmm_wide.hip.h) -------------------------------------------
// *done = true: traceback bytes, last states and scores of the batch are in place (the generic traceback follows);
// false: the caller runs the sequential kernels
static int wide_vit_warmup(int CS) {    // positions the quantised pass runs ahead of its chunk (<= chunk length)
  const char *s = std::getenv("TEHMM_WIDE_VIT_WARMUP");
  // (measured at 100 states, exact blocks of the chain per 480 kb: dense 3 300 at 64 and at 32 positions; sticky 0.995:
  //  1 397 at 64, 1 392 at 32, 1 728 at 16 -- a chunk that has not converged only costs the chain one 16-position block)
  return std::min(CS, std::max(0, s ? std::atoi(s) : 32));
}
static int viterbi_wide_cp(tehmm_batch *b, const tehmm_model *m, const IntervalTab &iv, const EmisTab &em, bool ratio,
                           hipStream_t st, hipEvent_t ev_spec, bool *done) {
  *done = false;
  const char *ws = std::getenv("TEHMM_WIDE_VIT");
  if (ws && std::atoi(ws) == 0) return TEHMM_OK;
  // chunks of 256 positions unless TEHMM_SPEC_CHUNK says otherwise: a chunk that crosses into the next binade is walked
  // exactly up to the crossing (half a chunk on average, seven crossings per 100 kb interval), an interval's first
  // chunk always (measured, 2 Mb in 20 intervals at 100 states: 1024 -> 55 ms, 512 -> 38, 256 -> 32, 128 -> 31)
  const int CS = std::getenv("TEHMM_SPEC_CHUNK") ? spec_chunk_size() : 256;
  if (m->N < 64 || m->N > 128 || CS < 64 || b->total < 2 * (int64_t)CS) return TEHMM_OK;
  if (use_wide() == false || !(m->NP / 4 <= TEHMM_WIDE_QM && wide_lds_bytes(em.lds_rows, m->NP) <= 160 * 1024)) return TEHMM_OK;
  int rc = spec_prepare(b, m, CS);
  if (rc) return rc;
  SpecWork &sw = b->sw;
  WideWork &w = b->ww;
  const int nc = sw.n_chunks;
  if (nc <= 0) return TEHMM_OK;
  if (!w.BL.p) {
    // workspaces: log rows (1 KB per position), two sets of traceback bytes, recorded rows -- if they do not fit next
    // to what the batch holds already, the sequential kernels (which need none of it) take the call
    size_t free_b = 0, total_b = 0;
    HIPCHK(dev_mem_info(&free_b, &total_b));
    const double need = (double)b->total_pad * (TEHMM_WIDE_S * 8.0 + 2.0 * b->TBW) * 1.25 +
                        (double)nc * ((sw.CS / TEHMM_VROW) + 2.0) * m->NP * 8.0 * 1.25;
    if (need > 0.9 * (double)free_b) return TEHMM_OK;
  }
  HIPCHK(w.BL.ensure((size_t)b->total_pad * TEHMM_WIDE_S + TEHMM_WIDE_S));
  HIPCHK(w.rows2.ensure((size_t)nc * (CS / TEHMM_VROW) * m->NP + 1));
  HIPCHK(w.pre.ensure((size_t)2 * nc * m->NP + 1));
  HIPCHK(w.tb2.ensure((size_t)(b->total_pad + 1) * b->TBW));
  HIPCHK(w.tb1.ensure((size_t)(b->total_pad + 1) * b->TBW));
  HIPCHK(w.ready.ensure((size_t)nc + 1));
  HIPCHK(w.sel_from.ensure((size_t)nc + 1));
  HIPCHK(w.sel_hyp.ensure((size_t)nc + 1));
  VitChunks vc;
  std::memset(&vc, 0, sizeof(vc));
  vc.iv = sw.iv.p; vc.t0 = sw.t0.p; vc.first = sw.first.p; vc.n = nc; vc.CS = CS;
  vc.e = sw.e.p; vc.gain = sw.gain.p; vc.ok = sw.ok.p; vc.wmin = sw.wmin.p; vc.rows = sw.rows.p;
  vc.ntie = sw.ntie.p; vc.ties = sw.ties.p; vc.tierows = sw.tierows.p; vc.segmin = sw.segmin.p;
  const EmisTab emg = without_lds_tables(em);
  hipLaunchKernelGGL(k_wide_logrows, dim3((nc + 3) / 4), dim3(256), 0, st, iv, emg, vc, m->N, w.BL.p);
  const size_t lds = (size_t)m->N * TEHMM_WIDE_S * sizeof(double) + 64 + TEHMM_WIDE_MAXTT * sizeof(int);
  // P0: plain gains of every chunk
  {
    const int nwg = (nc + 7) / 8;
    w.h_wkc.assign((size_t)nwg * 8, -1);
    for (int c = 0; c < nc; ++c) w.h_wkc[(size_t)c] = c;
    w.h_wke.assign((size_t)nwg, 0);
    HIPCHK(w.wk_c.ensure(w.h_wkc.size() + 8));
    HIPCHK(w.wk_e.ensure(w.h_wke.size() + 8));
    HIPCHK(hipMemcpyAsync(w.wk_c.p, w.h_wkc.data(), w.h_wkc.size() * sizeof(int), hipMemcpyHostToDevice, st));
    const char *pf = std::getenv("TEHMM_WIDE_P0F");
    if (!(pf && std::atoi(pf) == 0)) {
      // packed floats (k_vit_wide_gain): the gains only place the binades
      const size_t ldsg = (size_t)m->N * 64 * sizeof(float2);
      if (ratio) {
        allow_lds(k_vit_wide_gain<true>, ldsg);
        hipLaunchKernelGGL((k_vit_wide_gain<true>), dim3(nwg), dim3(512), ldsg, st, iv, vc, m->N, m->NP, (const double *)m->lt.p,
                           (const double *)w.BL.p, (const double *)b->ratios.p);
      } else {
        allow_lds(k_vit_wide_gain<false>, ldsg);
        hipLaunchKernelGGL((k_vit_wide_gain<false>), dim3(nwg), dim3(512), ldsg, st, iv, vc, m->N, m->NP, (const double *)m->lt.p,
                           (const double *)w.BL.p, (const double *)nullptr);
      }
    } else if (ratio) {
      allow_lds(k_vit_wide_spec<false, true>, lds);
      hipLaunchKernelGGL((k_vit_wide_spec<false, true>), dim3(nwg), dim3(512), lds, st, iv, vc, m->N, m->NP,
                         (const double *)m->lt.p, (const double *)w.BL.p, (const double *)b->ratios.p,
                         (const int *)w.wk_c.p, (const int *)w.wk_e.p, b->TBW, b->tb.p, w.tb2.p, w.rows2.p, wide_vit_warmup(CS), w.pre.p);
    } else {
      allow_lds(k_vit_wide_spec<false, false>, lds);
      hipLaunchKernelGGL((k_vit_wide_spec<false, false>), dim3(nwg), dim3(512), lds, st, iv, vc, m->N, m->NP,
                         (const double *)m->lt.p, (const double *)w.BL.p, (const double *)nullptr,
                         (const int *)w.wk_c.p, (const int *)w.wk_e.p, b->TBW, b->tb.p, w.tb2.p, w.rows2.p, wide_vit_warmup(CS), w.pre.p);
    }
    w.h_gain.resize((size_t)nc);
    HIPCHK(hipMemcpyAsync(w.h_gain.data(), sw.gain.p, (size_t)nc * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
  }
  // binades; work lists of P2: the speculated chunks sorted by binade, eight to a workgroup
  spec_assign_binades(b, w.h_gain, w.h_e);
  std::vector<int> order;
  for (int c = 0; c < nc; ++c)
    if (w.h_e[(size_t)c] != TEHMM_SPEC_NONE) order.push_back(c);
  // (binade, position in the interval): co-resident waves share one quantised table, and every interval's chain --
  // which may follow the pass while it runs -- finds its next chunk done early
  std::stable_sort(order.begin(), order.end(), [&](int a, int c2) {
    if (w.h_e[(size_t)a] != w.h_e[(size_t)c2]) return w.h_e[(size_t)a] < w.h_e[(size_t)c2];
    return sw.h_t0[(size_t)a] < sw.h_t0[(size_t)c2];
  });
  w.h_wkc.clear();
  w.h_wke.clear();
  for (size_t i = 0; i < order.size();) {
    const int e = w.h_e[(size_t)order[i]];
    w.h_wke.push_back(e);
    int k = 0;
    for (; k < 8 && i < order.size() && w.h_e[(size_t)order[i]] == e; ++k, ++i) w.h_wkc.push_back(order[i]);
    for (; k < 8; ++k) w.h_wkc.push_back(-1);
  }
  const int nwg2 = (int)w.h_wke.size();
  // every interval's head -- what lies before its first speculated chunk -- is walked by the exact chain on a side
  // stream WHILE the quantised pass runs (phase 1 of k_vit_wide_fix); the chain proper resumes behind it (phase 2)
  const char *hds = std::getenv("TEHMM_WIDE_HEAD");
  const bool head = !(hds && std::atoi(hds) == 0) && nwg2 > 0;
  // ... and with few enough intervals that their workgroups cannot keep the pass off the GPU, phase 2 follows the pass
  // WHILE it runs, waiting for each chunk's ready flag
  const char *pls = std::getenv("TEHMM_WIDE_POLL");
  const bool poll = head && b->n <= 64 && !(pls && std::atoi(pls) == 0);
  // polls (~2.7 us each) a chain may spend waiting: the pass needs ~1.4 us per chunk with the GPU to itself, allow tenfold
  const int spin_limit = (int)std::min<int64_t>(1 << 20, std::max<int64_t>(1 << 13, (int64_t)nc * 6));
  if (head) {
    w.h_hstop.assign((size_t)b->n, 0);
    for (int i = 0; i < b->n; ++i) {
      int64_t hs = b->h_len[(size_t)i];
      for (int64_t c = sw.h_first[i]; c < sw.h_first[i + 1]; ++c)
        if (w.h_e[(size_t)c] != TEHMM_SPEC_NONE) { hs = sw.h_t0[(size_t)c]; break; }
      w.h_hstop[(size_t)i] = hs;
    }
    HIPCHK(w.hstop.ensure((size_t)b->n + 1));
    HIPCHK(w.hvec.ensure((size_t)b->n * m->NP + 1));
    HIPCHK(w.hflag.ensure((size_t)b->n + 1));
    HIPCHK(hipMemcpyAsync(w.hstop.p, w.h_hstop.data(), (size_t)b->n * sizeof(int64_t), hipMemcpyHostToDevice, st));
  }
  HIPCHK(hipMemcpyAsync(sw.e.p, w.h_e.data(), (size_t)nc * sizeof(int), hipMemcpyHostToDevice, st));
  HIPCHK(hipMemsetAsync(sw.ok.p, 0, (size_t)nc * sizeof(int), st));
  HIPCHK(hipMemsetAsync(sw.ntie.p, 0, (size_t)nc * sizeof(int), st));
  HIPCHK(hipMemsetAsync(sw.stats.p, 0, 8 * sizeof(int), st));
  HIPCHK(hipMemsetAsync(w.sel_hyp.p, 0xff, ((size_t)nc + 1) * sizeof(int), st));
  HIPCHK(hipMemsetAsync(w.ready.p, 0, ((size_t)nc + 1) * sizeof(int), st));
  const size_t ldsf = wide_lds_bytes(em.lds_rows, m->NP);
  auto launch_chain = [&](int phase, hipStream_t s2) {
    if (ratio) {
      allow_lds(k_vit_wide_fix<true>, ldsf);
      hipLaunchKernelGGL((k_vit_wide_fix<true>), dim3(b->n), dim3(256), ldsf, s2, iv, em, vc, m->N, m->NP, (const double *)m->lt.p,
                         (const double *)m->pi.p, (const double *)b->ratios.p, b->TBW, b->tb.p, b->last_state.p, b->vit_lp.p,
                         sw.stats.p, (const double *)w.rows2.p, (const double *)w.BL.p, w.sel_from.p, w.sel_hyp.p,
                         (const double *)w.pre.p, phase, (const int64_t *)w.hstop.p, w.hvec.p, w.hflag.p,
                         (const int *)(phase == 2 && poll ? w.ready.p : nullptr), spin_limit);
    } else {
      allow_lds(k_vit_wide_fix<false>, ldsf);
      hipLaunchKernelGGL((k_vit_wide_fix<false>), dim3(b->n), dim3(256), ldsf, s2, iv, em, vc, m->N, m->NP, (const double *)m->lt.p,
                         (const double *)m->pi.p, (const double *)nullptr, b->TBW, b->tb.p, b->last_state.p, b->vit_lp.p,
                         sw.stats.p, (const double *)w.rows2.p, (const double *)w.BL.p, w.sel_from.p, w.sel_hyp.p,
                         (const double *)w.pre.p, phase, (const int64_t *)w.hstop.p, w.hvec.p, w.hflag.p,
                         (const int *)(phase == 2 && poll ? w.ready.p : nullptr), spin_limit);
    }
  };
  if (head) {
    (void)hipEventRecord(b->evX[0], st);
    (void)hipStreamWaitEvent(b->sB, b->evX[0], 0);
    launch_chain(1, b->sB);
    if (!poll) (void)hipEventRecord(b->evX[1], b->sB);
  }
  if (nwg2 > 0) {
    HIPCHK(w.wk_c.ensure(w.h_wkc.size() + 8));
    HIPCHK(w.wk_e.ensure(w.h_wke.size() + 8));
    HIPCHK(hipMemcpyAsync(w.wk_c.p, w.h_wkc.data(), w.h_wkc.size() * sizeof(int), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(w.wk_e.p, w.h_wke.data(), w.h_wke.size() * sizeof(int), hipMemcpyHostToDevice, st));
    if (ratio) {
      allow_lds(k_vit_wide_spec<true, true>, lds);
      hipLaunchKernelGGL((k_vit_wide_spec<true, true>), dim3(nwg2), dim3(512), lds, st, iv, vc, m->N, m->NP,
                         (const double *)m->lt.p, (const double *)w.BL.p, (const double *)b->ratios.p,
                         (const int *)w.wk_c.p, (const int *)w.wk_e.p, b->TBW, w.tb1.p, w.tb2.p, w.rows2.p, wide_vit_warmup(CS), w.pre.p,
                         w.ready.p);
    } else {
      allow_lds(k_vit_wide_spec<true, false>, lds);
      hipLaunchKernelGGL((k_vit_wide_spec<true, false>), dim3(nwg2), dim3(512), lds, st, iv, vc, m->N, m->NP,
                         (const double *)m->lt.p, (const double *)w.BL.p, (const double *)nullptr,
                         (const int *)w.wk_c.p, (const int *)w.wk_e.p, b->TBW, w.tb1.p, w.tb2.p, w.rows2.p, wide_vit_warmup(CS), w.pre.p,
                         w.ready.p);
    }
  }
  (void)hipEventRecord(ev_spec, st);                 // the throughput passes are behind: what follows is latency-bound
  // the exact chain
  if (poll) {
    launch_chain(2, b->sB);                          // behind the heads, next to the pass it follows
    (void)hipEventRecord(b->evX[1], b->sB);
    (void)hipStreamWaitEvent(st, b->evX[1], 0);
  } else if (head) {
    (void)hipStreamWaitEvent(st, b->evX[1], 0);
    launch_chain(2, st);
  } else {
    launch_chain(0, st);
  }
  hipLaunchKernelGGL(k_wide_tb_select, dim3(nc), dim3(256), 0, st, iv, vc, m->N, b->TBW, b->tb.p, (const uint8_t *)w.tb1.p,
                     (const uint8_t *)w.tb2.p, (const int64_t *)w.sel_from.p, (const int *)w.sel_hyp.p);
  if (std::getenv("TEHMM_SPEC_DEBUG")) {
    HIPCHK(hipStreamSynchronize(st));
    std::vector<int> hok((size_t)nc), hnt((size_t)nc), hsel((size_t)nc);
    HIPCHK(hipMemcpy(hok.data(), sw.ok.p, (size_t)nc * sizeof(int), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(hnt.data(), sw.ntie.p, (size_t)nc * sizeof(int), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(hsel.data(), w.sel_hyp.p, (size_t)nc * sizeof(int), hipMemcpyDeviceToHost));
    int n_spec = 0, n_ok = 0, n_sel[3] = {0, 0, 0};
    long ties = 0;
    for (int c = 0; c < nc; ++c) {
      if (w.h_e[(size_t)c] == TEHMM_SPEC_NONE) continue;
      ++n_spec;
      n_ok += hok[(size_t)c] != 0;
      ties += hnt[(size_t)c];
      ++n_sel[hsel[(size_t)c] < 0 ? 2 : hsel[(size_t)c]];
    }
    std::fprintf(stderr, "[tehmm wide vit] chunks %d, speculated %d, usable %d, tie positions %ld, adopted h0 %d h1 %d none %d\n", nc,
                 n_spec, n_ok, ties, n_sel[0], n_sel[1], n_sel[2]);
    {
      int st8[8];
      HIPCHK(hipMemcpy(st8, sw.stats.p, sizeof(st8), hipMemcpyDeviceToHost));
      std::fprintf(stderr, "  failed checks %d: not in binade %d, leaves binade %d, no constant with matching parity %d\n", st8[2], st8[3],
                   st8[4], st8[5]);
    }
    for (int e = 10; e < 30; ++e) {
      int ne = 0, nok = 0, nsel = 0;
      long nt = 0;
      for (int c = 0; c < nc; ++c)
        if (w.h_e[(size_t)c] == e) { ++ne; nok += hok[(size_t)c] != 0; nsel += hsel[(size_t)c] >= 0; nt += hnt[(size_t)c]; }
      if (ne) std::fprintf(stderr, "  binade %d: chunks %d usable %d adopted %d tie positions %ld\n", e, ne, nok, nsel, nt);
    }
  }
  *done = true;
  return TEHMM_OK;
}

// Forward / backward warm-up of the lane passes: TEHMM_LANE_WARMUP if set, else measured on this batch's own
// observations by k_fb_probe (96 windows per direction; tehmm_spec.hip.h) -- the longest forgetting time seen,
// plus a fifth, at least 32 and at most the item length.  TEHMM_LANE_PROBE=0: the round-2 constant 64.
template <int NT>
static void launch_fb_probe(const tehmm_model *m, const IntervalTab &iv, const EmisTab &emg, int n, int WMAX, const int *p_iv,
                            const int64_t *p_t0, int *steps, hipStream_t st) {
  const size_t lds = ((size_t)64 * (NT + 1) + 64) * sizeof(double);
  hipLaunchKernelGGL((k_fb_probe<NT, 0>), dim3(n), dim3(64), lds, st, iv, emg, m->N, (const double *)m->A.p, p_iv, p_t0, WMAX, steps);
  hipLaunchKernelGGL((k_fb_probe<NT, 1>), dim3(n), dim3(64), lds, st, iv, emg, m->N, (const double *)m->A.p, p_iv + n, p_t0 + n,
                     WMAX, steps + n);
}

static int fb_warmup(tehmm_batch *b, const tehmm_model *m, int LS, const IntervalTab &iv, const EmisTab &emg, int *out) {
  if (const char *wus = std::getenv("TEHMM_LANE_WARMUP")) {
    *out = std::min(LS, std::max(1, std::atoi(wus)));
    return TEHMM_OK;
  }
  *out = std::min(LS, 64);
  const char *pr = std::getenv("TEHMM_LANE_PROBE");
  if ((pr && std::atoi(pr) == 0) || m->NP > 64) return TEHMM_OK;
  LaneWork &lw = b->lw;
  if (lw.wu_model == m->uid && lw.wu_version == m->version && lw.wu_L == LS && lw.wu_val > 0) {
    *out = lw.wu_val;
    return TEHMM_OK;
  }
  const int WMAX = std::max(64, std::min(512, LS) & ~63);
  constexpr int NPR = 96;
  std::vector<int> cand;
  for (int i = 0; i < b->n; ++i)
    if (b->h_len[(size_t)i] >= 2 * (int64_t)WMAX + 128) cand.push_back(i);
  if (cand.empty()) return TEHMM_OK;
  std::vector<int> p_iv(2 * NPR);
  std::vector<int64_t> p_t0(2 * NPR);
  uint64_t rng = 0x9e3779b97f4a7c15ull ^ (uint64_t)b->total;
  auto next = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
  for (int k = 0; k < 2 * NPR; ++k) {
    const int id = cand[(size_t)(next() % cand.size())];
    const int64_t T = b->h_len[(size_t)id];
    const int64_t span = (T - 2 * WMAX - 64) / 64;                    // 64-aligned window starts
    const int64_t a = 64 + 64 * (int64_t)(next() % (uint64_t)std::max<int64_t>(1, span));
    p_iv[(size_t)k] = id;
    p_t0[(size_t)k] = k < NPR ? a : a + WMAX;                         // backward windows run down from their end
  }
  HIPCHK(lw.probe_iv.upload(p_iv.data(), p_iv.size()));
  HIPCHK(lw.probe_t0.upload(p_t0.data(), p_t0.size()));
  HIPCHK(lw.probe_steps.ensure(2 * NPR));
#define CALL(NT_) launch_fb_probe<NT_>(m, iv, emg, NPR, WMAX, lw.probe_iv.p, lw.probe_t0.p, lw.probe_steps.p, b->sP)
  TEHMM_NT_DISPATCH(m->NP, CALL)
#undef CALL
  HIPCHK(hipGetLastError());
  std::vector<int> steps(2 * NPR, 0);
  HIPCHK(hipMemcpyAsync(steps.data(), lw.probe_steps.p, steps.size() * sizeof(int), hipMemcpyDeviceToHost, b->sP));
  HIPCHK(hipStreamSynchronize(b->sP));
  int worst = 0;
  for (int v : steps) worst = std::max(worst, v);
  int wu = worst + worst / 5 + 8;
  wu = (wu + 7) & ~7;
  wu = std::min(LS, std::max(32, wu));
  lw.wu_model = m->uid;
  lw.wu_version = m->version;
  lw.wu_L = LS;
  lw.wu_val = wu;
  *out = wu;
  return TEHMM_OK;
}

int tehmm_eval_batch(tehmm_model_t *m, tehmm_batch_t *b, int flags, double *viterbi_logprob,
                     double *forward_logprob) {
  if (!m || !b) return fail(TEHMM_ERR_ARG, "tehmm_eval_batch: NULL handle");
  if (m->K != b->K) return fail(TEHMM_ERR_ARG, "tehmm_eval_batch: model/batch track count differ");
  if (!(flags & (TEHMM_EVAL_VITERBI | TEHMM_EVAL_POSTERIOR)))
    return fail(TEHMM_ERR_ARG, "tehmm_eval_batch: nothing to do");
  b->tnames.clear();
  b->tpairs.clear();
  b->tms.clear();
  if (b->n == 0 || b->total == 0) return TEHMM_OK;
#ifdef TEHMM_DEV_NT
  if (m->N < 64 && m->NP != TEHMM_DEV_NT)
    return fail(TEHMM_ERR_UNSUPPORTED, "development build: fused kernels exist for one padded state count only");
#endif
  int rc = ensure_workspace(b, m, flags);
  if (rc) return rc;
  const bool ratio = (flags & TEHMM_EVAL_USE_RATIOS) && b->has_ratios;
  IntervalTab iv;
  EmisTab em;
  fill_tabs(m, b, iv, em, false);   // decode / score_samples never apply ratios to emissions
  const int SPL = m->N <= 64 ? 1 : 2;
  const bool coop = m->N < 64;
  // Enqueue order: Viterbi speculation pass 0 -> the whole posterior pipeline (async on its own
  // streams) -> host binade assignment (needs pass 0) -> rest of the Viterbi pipeline.
  const int CS = spec_chunk_size();
  const bool spec_ok = coop && CS > 0 && m->NP <= 64 && b->total >= 2 * (int64_t)CS;
  const bool vit = flags & TEHMM_EVAL_VITERBI, postr = flags & TEHMM_EVAL_POSTERIOR;
  // segment ratios: the chunk-parallel Viterbi path takes them in its lane = item form only (finite
  // self-transitions, NP <= 36, the fused P0 pass); otherwise the sequential kernels
  bool vspec = vit && spec_ok && (!ratio || (ratio_lane_ok(m) && m->NP <= TEHMM_RATIO_LANE_MAX));
  const bool fspec = postr && spec_ok;
  if (vspec || fspec) {
    rc = spec_prepare(b, m, CS);
    if (rc) return rc;
  }
  SpecWork &sw = b->sw;
  LaneWork &lw = b->lw;
  const EmisTab emg = without_lds_tables(em);
  const int eV = 0, eP = 5;
  // lane = item geometry: the VALU lane passes (exact Viterbi, P0) hold 4 * NP VGPRs of state and stop at 36
  // padded states; the fused matrix-core forward / backward passes go up to 64
  const bool lane_vit_ok = m->NP <= 64;
  int LS = ((vspec && lane_vit_ok) || fspec) ? lane_sub_size(CS, b->total) : 0;
  // The lane = item Viterbi passes (TEHMM_LANE_VIT, default on) additionally need the fp64 log rows.
  const char *lvs = std::getenv("TEHMM_LANE_VIT");
  bool want_vlane = !(lvs && std::atoi(lvs) == 0);
  // TEHMM_FUSED=0 selects the round-1 posterior pipeline (emission rows, alpha' and beta' through HBM,
  // separate combine); default: the fused passes of tehmm_fused.hip.h
  const char *fus = std::getenv("TEHMM_FUSED");
  const bool fused_fb = !(fus && std::atoi(fus) == 0) && m->ptab.p != nullptr && m->K <= 78;   // (96 record entries incl. padding)
  if (LS > 0 && !b->lw.AL.p && !b->lw.AL32.p && !b->lw.B.p && !b->lw.B32.p) {
    // the item-interleaved buffers (8 * NP bytes per position each: alpha' -- plus, without the fused
    // passes, beta' and the linear emission rows --, the fp64 log rows of the exact Viterbi pass; 4 * NP
    // for the float rows of P0) must fit next to the results; otherwise do without the fp64 log rows, and
    // failing that stay with the [T][N] speculative passes
    size_t free_b = 0, total_b = 0;
    HIPCHK(dev_mem_info(&free_b, &total_b));
    const double per = (double)b->total * m->NP * 8.0;
    const double fb_units = fspec ? (fused_fb ? 0.7 : 3.2) : 0.0, p0_units = (vspec && !fused_fb) ? 0.5 : 0.0;
    b->lw.no_vlane = per * (fb_units + p0_units + 1.0) > 0.85 * (double)free_b;
    if (per * (fb_units + p0_units) > 0.85 * (double)free_b) LS = 0;
  }
  if (LS > 0 && want_vlane && !b->lw.no_vlane && !b->lw.B.p && (b->lw.AL.p || b->lw.AL32.p || b->lw.B32.p)) {
    // workspaces of an earlier call exist already: the fp64 (+ float) log rows must still fit
    size_t free_b = 0, total_b = 0;
    HIPCHK(dev_mem_info(&free_b, &total_b));
    if ((double)b->total * m->NP * 8.0 * 1.6 > 0.85 * (double)free_b) b->lw.no_vlane = true;
  }
  if (b->lw.no_vlane) want_vlane = false;
  if (ratio && vspec && !(LS > 0 && want_vlane && (fused_fb || !(fspec && LS > 0)))) {
    vspec = false;                                    // no lane passes for this call: sequential Viterbi with ratios
    if (!fspec) LS = 0;
  }
  const bool vlane = vspec && LS > 0 && want_vlane && lane_vit_ok, flane = fspec && LS > 0;
  // P0 (binade placement) as a packed-float lane pass over float emission rows; TEHMM_LANE_P0=0 keeps
  // the fp64 lane = state pass
  const char *lp0 = std::getenv("TEHMM_LANE_P0");
  const bool glane = vspec && LS > 0 && lane_vit_ok && (vlane || !(lp0 && std::atoi(lp0) == 0));
  int WuF = std::min(std::max(LS, 1), 64);                                   // forward / backward warm-up (set below)
  const char *wvs = std::getenv("TEHMM_LANE_WARMUP_VIT");
  const int WuV = std::min(LS, std::max(32, ((wvs ? std::atoi(wvs) : 32) + 31) & ~31));   // Viterbi warm-up (multiple of 32)
  VitChunks vc;
  std::vector<double> &gain = lw.hs_gain;
  if (vit) (void)hipEventRecord(b->ev[eV], b->sV);
  // emission rows and P0 in one pass (no float copy of the rows) unless the round-1 posterior pipeline, which
  // takes its linear rows from the row kernel, is selected
  const bool emis_gain = glane && !(flane && !fused_fb);
  if (vlane || flane || glane) {
    rc = lane_prepare(b, m, CS, LS, flane, vlane, glane, fused_fb, !emis_gain);
    if (rc) return rc;
  }
  if (flane) {
    rc = fb_warmup(b, m, LS, iv, emg, &WuF);
    if (rc) return rc;
  }
  // Scheduling.  The speculative Viterbi passes, the emission rows, the forward / backward lane passes
  // and the combine are throughput kernels that each fill the GPU; the fix-up chains and the traceback
  // are latency kernels on a few CUs.  With both requested, the Viterbi passes go first and the
  // posterior pipeline is released behind them (event), so that its wide kernels run next to the
  // Viterbi fix-up chain instead of competing with the passes that chain is waiting for (its emission
  // rows, which depend on nothing, are computed up front).
  const char *dfs = std::getenv("TEHMM_DEFER");
  // 0: no order, 1: behind the Viterbi passes, 2: behind the emission rows only.  Measured (ms per step, modes 1 / 2;
  // 35 states, bench geometry): 10 Mb 15.6 / 15.9, 20 Mb 21.8 / 21.7, 30 Mb 28.3 / 26.0, 50 Mb 45.1 / 41.2, 70 Mb 60.1 /
  // 56.3, 100 Mb 73.3 / 75.6 (mode 0: 76.0).  Below ~3 waves per SIMD the one-wave quantised pass and the emission
  // kernel leave tails that the posterior passes fill (mode 2); on a full GPU they only take from each other and the
  // exact chain loses its quiet partner (mode 1).
  // Mode 3 (fused passes): the FORWARD half runs from the start, beside the emission-row kernel and the host's binade
  // placement; the quantised pass waits for the forward pass itself and then has the GPU alone; the backward half
  // follows it, next to the exact chain.
  // Where ONE chain dominates (a single 10 Mb interval: 16.0 / 18.1) mode 1 keeps the posterior passes as its partner.
  int64_t longest = 0;
  for (int i = 0; i < b->n; ++i) longest = std::max<int64_t>(longest, b->h_len[(size_t)i]);
  // Round 4 (three-wave quantised pass at 168 registers, binade placement on the device; tools/defer_sweep.py, ms per
  // first evaluation, modes 0 / 1 / 2 / 3): 10 Mb 17.1 / 14.9 / 15.3 / 15.3, 20 Mb 22.5 / 20.1 / 20.5 / 20.4, 30 Mb
  // 25.8 / 27.4 / 26.7 / 25.3, 50 Mb 36.3 / 40.7 / 37.5 / 37.2, 70 Mb 51.6 / 55.4 / 53.5 / 48.1, 100 Mb 65.2 / 70.5 /
  // 67.7 / 65.1; one 10 Mb interval 16.7 / 15.9 / 17.2 / 16.8.  With the host out of the Viterbi pipeline the overlapped
  // orders win from 25 Mb up: mode 3 there, mode 1 below and where one chain dominates.
  const bool big = b->total >= (int64_t)25000000 && 4 * longest <= b->total;
  const int defer_mode = dfs ? std::atoi(dfs) : (big ? 3 : 1);
  const bool defer_post = vit && postr && vspec && (defer_mode == 1 || defer_mode == 3 || defer_mode == 5);
  const bool split_post = defer_post && (defer_mode == 3 || defer_mode == 5) && flane && fused_fb && vlane;
  if (postr) (void)hipEventRecord(b->ev[eP], b->sP);
  // Binade placement on the device (tehmm_place.hip.h; TEHMM_DEVICE_PLACE=0: the host path of rounds 1..3): everything
  // the quantised pass needs besides the gains -- zeroed flags, the argument structs, the tables of all binades -- is
  // put on the stream BEFORE the gain pass, so that nothing but four small kernels separates the two passes.
  const bool dev_place_on = !(std::getenv("TEHMM_DEVICE_PLACE") && std::atoi(std::getenv("TEHMM_DEVICE_PLACE")) == 0);
  const bool dev_place = dev_place_on && vlane && emis_gain;
  if (vspec) {
    vc.iv = sw.iv.p; vc.t0 = sw.t0.p; vc.first = sw.first.p; vc.n = sw.n_chunks; vc.CS = CS;
    vc.e = sw.e.p; vc.gain = sw.gain.p; vc.ok = sw.ok.p; vc.wmin = sw.wmin.p; vc.rows = sw.rows.p;
    vc.ntie = sw.ntie.p; vc.ties = sw.ties.p; vc.tierows = sw.tierows.p; vc.segmin = sw.segmin.p;
    vc.offend = sw.offend.p; vc.clink = sw.clink.p; vc.clk = sw.clk.p;
    vc.rtarget = sw.rtarget.p; vc.rsel = sw.rsel.p; vc.racc = sw.racc.p; vc.rmn = sw.rmn.p;
    vc.tsoft = sw.tsoft.p; vc.tpar = sw.tpar.p;
  }
  if (dev_place) {
    hipStream_t st = b->sV;
    rc = ensure_qtabs(m, ratio);
    if (rc) return rc;
    const size_t ng = (size_t)std::max(1, lw.n_groups);
    HIPCHK(lw.wk_g.ensure(2 * ng + TEHMM_PLACE_NE));
    HIPCHK(lw.wk_e.ensure(2 * ng + TEHMM_PLACE_NE));
    HIPCHK(lw.wk_items.ensure((ng + TEHMM_PLACE_NE) * 64));
    HIPCHK(lw.gclass.ensure(ng));
    HIPCHK(lw.place.ensure(1));
    HIPCHK(hipMemsetAsync(lw.place.p, 0, sizeof(PlaceCounts), st));
    HIPCHK(hipMemsetAsync(lw.wk_items.p, 0xff, (ng + TEHMM_PLACE_NE) * 64 * sizeof(int), st));
    HIPCHK(hipMemsetAsync(sw.ok.p, 0, (size_t)std::max(1, sw.n_chunks) * sizeof(int), st));
    HIPCHK(hipMemsetAsync(sw.stats.p, 0, 2 * sizeof(int), st));
    HIPCHK(hipMemsetAsync(lw.vbad.p, 0, ng * 64 * sizeof(int), st));
    HIPCHK(hipMemsetAsync(lw.vntie.p, 0, ng * 64 * sizeof(int), st));
    rc = vit_lane_upload_args(b, vc, st);
    if (rc) return rc;
  }
  // the emission-row / gain pass of the Viterbi pipeline (a lambda since round 4: mode 5 enqueues it BEHIND the forward pass)
  auto enqueue_emis_p0 = [&]() -> int {
  if (vlane || glane) {
      // emission rows of every position, once, item-interleaved (log rows for Viterbi, linear for fwd/bwd)
      hipStream_t st = b->sV;
      if (emis_gain) {
#define CALL(NT_) launch_emis_gain_lane<NT_>(b, m, iv, em, CS, WuV, ratio, st)
        TEHMM_NT_DISPATCH(m->NP, CALL)
#undef CALL
      } else {
#define CALL(NT_) launch_emis_lane<NT_>(b, m, iv, em, vlane, flane && !fused_fb, glane, st)
        TEHMM_NT_DISPATCH(m->NP, CALL)
#undef CALL
      }
      (void)hipEventRecord(b->ev[eV + 4], st);
      if (flane) {
        if (!fused_fb) {                                  // (the fused passes compute their own emission rows)
          (void)hipEventRecord(b->evX[0], st);
          (void)hipStreamWaitEvent(b->sP, b->evX[0], 0);
        }
        (void)hipEventRecord(b->ev[eP + 4], b->sP);
      }
    }
    if (vspec) {
      // chunk-parallel exact Viterbi: P0 (plain gains) -> binades -> P2 (quantised) -> fix-up chain
      hipStream_t st = b->sV;
      if (glane) {
        if (!emis_gain) {
#define CALL(NT_) launch_gain_lane<NT_>(b, m, iv, CS, WuV, st)
          TEHMM_NT_DISPATCH(m->NP, CALL)
#undef CALL
        }
        if (!dev_place) {
          gain.resize((size_t)std::max(1, lw.n_groups) * 64);
          HIPCHK(hipMemcpyAsync(gain.data(), lw.vgain.p, gain.size() * sizeof(double), hipMemcpyDeviceToHost, st));
        }
      } else {
#define CALL(NT_) launch_vit_spec<NT_>(b, m, iv, emg, vc, false, st)
        TEHMM_NT_DISPATCH(m->NP, CALL)
#undef CALL
        gain.resize((size_t)std::max(1, sw.n_chunks));
        HIPCHK(hipMemcpyAsync(gain.data(), sw.gain.p, gain.size() * sizeof(double), hipMemcpyDeviceToHost, st));
      }
    }
    return TEHMM_OK;
  };
  // mode 5 (round 4): as mode 3, but the emission + gain pass waits for the forward KERNEL: the forward pass runs alone
  // (12 ms instead of 17 beside the exact Viterbi chain, or 26 beside the emission kernel), then emission + gain pass,
  // quantised pass, and the backward half beside the exact chain and the traceback
  const bool f_first = split_post && defer_mode == 5;
  if (!f_first) {
    rc = enqueue_emis_p0();
    if (rc) return rc;
  }
  auto enqueue_emission = [&]() {
    if (flane && !fused_fb && !vlane && !glane) {
      // emission rows (linear domain) for the forward / backward lane passes
      hipStream_t st = b->sP;
#define CALL(NT_) launch_emis_lane<NT_>(b, m, iv, em, false, true, false, st)
      TEHMM_NT_DISPATCH(m->NP, CALL)
#undef CALL
      (void)hipEventRecord(b->ev[eP + 4], st);
    }
  };
  bool wide_cp = false;                             // the chunk-parallel posterior for 64 <= N <= 128 ran
  bool wide_vit = false;                            // ... and the chunk-parallel exact Viterbi
  bool wide_pending = false;                        // a chunk-parallel posterior attempt is in flight
  // half (fused passes only): 1 = the forward half now, 2 = the backward half; 0 = everything
  auto enqueue_posterior = [&](int half = 0) -> int {
    hipStream_t st = b->sP;
    if (half != 2) (void)hipEventRecord(b->ev[10], st);            // start of the passes (behind any deferral wait)
    else (void)hipEventRecord(b->ev[11], st);                      // ... of the backward half
    if (flane) {
      // lane = item passes (forward, backward, links), then the two sequential chains on the
      // item-interleaved rows, then the transposing combine
      FbChunks fc{};
      fc.iv = sw.iv.p; fc.t0 = sw.t0.p; fc.first = sw.first.p; fc.n = sw.n_chunks; fc.CS = CS;
      fc.scale = sw.scale.p; fc.wstart = sw.wstart.p;
      fc.link_f = lw.link_f.p; fc.glog_f = lw.glog_f.p; fc.link_b = lw.link_b.p; fc.runend_f = lw.runend_f.p;
      fc.pre_f = lw.cpre_f.p; fc.runstart_b = lw.runstart_b.p;
      if (half != 2) {
        (void)hipMemsetAsync(b->dead.p, 0, (size_t)(b->n + 1) * sizeof(int), st);
        (void)hipMemsetAsync(sw.stats.p + 2, 0, 4 * sizeof(int), st);
      }
      if (fused_fb) {
        int rcf = TEHMM_OK;
#define CALL(NT_) rcf = launch_fused_fb<NT_>(b, m, iv, em, fc, WuF, st, b->ev[eP + 3], b->ev[eP + 1], false, half, half == 1 ? b->evX[1] : nullptr)
        TEHMM_NT_DISPATCH(m->NP, CALL)
#undef CALL
        if (rcf) return rcf;
        if (half == 1) return TEHMM_OK;
      } else {
#define CALL(NT_) launch_fb_lane<NT_>(b, m, iv, em, fc, WuF, st, b->ev[eP + 3])
        TEHMM_NT_DISPATCH(m->NP, CALL)
#undef CALL
        (void)hipEventRecord(b->ev[eP + 1], st);
#define CALL(NT_) launch_combine_lane<NT_>(b, m, iv, st)
        TEHMM_NT_DISPATCH(m->NP, CALL)
#undef CALL
      }
      hipLaunchKernelGGL(k_poison_dead, dim3(64, std::min(b->n, 1024)), dim3(256), 0, st, iv, b->dead.p, m->N,
                         b->post.p, b->fwd_lp.p);
    } else if (fspec) {
      // chunk-parallel forward / backward: speculative rows from uniform starts, then the two
      // sequential chains (forward on this stream, backward on its own) with verified jumps
      rc = ensure_beta(b, m);
      if (rc) return rc;
      FbChunks fc{};
      fc.iv = sw.iv.p; fc.t0 = sw.t0.p; fc.first = sw.first.p; fc.n = sw.n_chunks; fc.CS = CS;
      fc.scale = sw.scale.p; fc.wstart = sw.wstart.p;
      (void)hipMemsetAsync(b->dead.p, 0, (size_t)(b->n + 1) * sizeof(int), st);
      (void)hipMemsetAsync(sw.stats.p + 2, 0, 4 * sizeof(int), st);
#define CALL(NT_) launch_fb_spec<NT_>(b, m, iv, emg, fc, st)
      TEHMM_NT_DISPATCH(m->NP, CALL)
#undef CALL
      (void)hipEventRecord(b->ev[eP + 3], st);
      (void)hipStreamWaitEvent(b->sB, b->ev[eP + 3], 0);
#define CALL(NT_) launch_fb_fix<NT_>(b, m, iv, em, fc, st, b->sB)
      TEHMM_NT_DISPATCH(m->NP, CALL)
#undef CALL
      (void)hipEventRecord(b->evX[1], b->sB);
      (void)hipStreamWaitEvent(st, b->evX[1], 0);
      (void)hipEventRecord(b->ev[eP + 1], st);
      hipLaunchKernelGGL((k_combine<true>), dim3(grid_for(b->total * 16, 256, 256 * 16)), dim3(256), 0, st,
                         b->total, m->N, b->post.p, b->beta.p);
      hipLaunchKernelGGL(k_poison_dead, dim3(64, std::min(b->n, 1024)), dim3(256), 0, st, iv, b->dead.p, m->N,
                         b->post.p, b->fwd_lp.p);
    } else if (coop) {
      rc = ensure_beta(b, m);
      if (rc) return rc;
      (void)hipMemsetAsync(b->dead.p, 0, (size_t)(b->n + 1) * sizeof(int), st);
#define CALL(NT_) launch_fb_coop<NT_>(b, m, iv, em, st)
      TEHMM_NT_DISPATCH(m->NP, CALL)
#undef CALL
      (void)hipEventRecord(b->ev[eP + 1], st);
      hipLaunchKernelGGL((k_combine<true>), dim3(grid_for(b->total * 16, 256, 256 * 16)), dim3(256), 0, st,
                         b->total, m->N, b->post.p, b->beta.p);
      hipLaunchKernelGGL(k_poison_dead, dim3(64, std::min(b->n, 1024)), dim3(256), 0, st, iv, b->dead.p, m->N, b->post.p,
                         b->fwd_lp.p);
    } else {
      // 64 <= N <= 128: the chunk-parallel passes are enqueued; their verdict is read behind the Viterbi enqueue
      // (finish_wide_posterior), so the two pipelines share the GPU
      rc = posterior_wide_cp(b, m, iv, em, st, b->ev[eP + 1], &wide_pending, wide_vit ? (const double *)b->ww.BL.p : nullptr);
      if (rc) return rc;
      if (!wide_pending) {
        if (SPL == 1) launch_posterior<1>(b, m, iv, em, st, b->ev[eP + 1]);
        else launch_posterior<2>(b, m, iv, em, st, b->ev[eP + 1]);
      }
    }
    (void)hipEventRecord(b->ev[eP + 2], st);
    return TEHMM_OK;
  };
  auto finish_wide_posterior = [&]() -> int {
    if (!wide_pending) return TEHMM_OK;
    wide_pending = false;
    hipStream_t st = b->sP;
    bool wide_done = false;
    int rcw = posterior_wide_finish(b, m, iv, st, b->ev[eP + 1], &wide_done);
    if (rcw) return rcw;
    wide_cp = wide_done;
    if (!wide_done) {
      if (SPL == 1) launch_posterior<1>(b, m, iv, em, st, b->ev[eP + 1]);
      else launch_posterior<2>(b, m, iv, em, st, b->ev[eP + 1]);
    }
    (void)hipEventRecord(b->ev[eP + 2], st);
    return TEHMM_OK;
  };
  if (postr && defer_post) enqueue_emission();      // the emission rows do not wait (17 ms next to P0)
  if (split_post) {
    rc = enqueue_posterior(1);
    if (rc) return rc;
    if (f_first) {
      (void)hipStreamWaitEvent(b->sV, b->evX[1], 0);       // the forward kernel is through
      rc = enqueue_emis_p0();
      if (rc) return rc;
    }
  } else if (postr && defer_post && flane && fused_fb) {
    // neither do the index records of the fused passes: they only read the observations
    int rcp = TEHMM_OK;
    FusedOrder fo_unused;
#define CALL(NT_) rcp = fused_prepare<NT_>(b, m, iv, WuF, b->sP, fo_unused)
    TEHMM_NT_DISPATCH(m->NP, CALL)
#undef CALL
    if (rcp) return rcp;
  }
  // 64 <= N <= 128 (tehmm_wide.hip.h), both results: the same order, arranged at the Viterbi enqueue below
  const bool wide_defer = vit && postr && !vspec && !coop && m->N >= 64 && defer_mode != 0;
  if (postr && !defer_post) {
    enqueue_emission();
    if (!wide_defer) {
      if (defer_mode == 2 && vit && vspec && (vlane || glane)) (void)hipStreamWaitEvent(b->sP, b->ev[eV + 4], 0);
      rc = enqueue_posterior();
      if (rc) return rc;
    }
  }
  // item gains of a lane P0 pass -> chunk gains (chunks the lanes did not run: see below)
  auto chunk_gains = [&](std::vector<double> &cgain) {
      cgain.assign((size_t)std::max(1, sw.n_chunks), 0.0);
      const int SUB = CS / LS;
      for (int c = 0; c < sw.n_chunks; ++c) {
        const int id = sw.h_iv[(size_t)c];
        const int64_t it0 = lw.h_first_item(id) + sw.h_t0[(size_t)c] / LS;
        double gsum = 0.0;
        const bool full = sw.h_t0[(size_t)c] + CS <= b->h_len[id];
        if (!full || c == sw.h_first[id]) gsum = std::nan("");
        else for (int k = 0; k < SUB; ++k) gsum += gain[(size_t)(it0 + k)];
        cgain[(size_t)c] = gsum;
      }
      // chunks the lanes did not run carry NaN; the prefix sums of spec_assign_binades still need a
      // number for them: use the mean gain per position of the interval's speculated chunks
      for (int i = 0; i < b->n; ++i) {
        double sum = 0.0;
        int cnt = 0;
        for (int64_t c = sw.h_first[i]; c < sw.h_first[i + 1]; ++c)
          if (cgain[(size_t)c] == cgain[(size_t)c]) { sum += cgain[(size_t)c]; ++cnt; }
        const double mean = cnt ? sum / cnt : std::nan("");
        for (int64_t c = sw.h_first[i]; c < sw.h_first[i + 1]; ++c)
          if (!(cgain[(size_t)c] == cgain[(size_t)c])) {
            const int64_t clen = std::min<int64_t>(CS, b->h_len[i] - sw.h_t0[(size_t)c]);
            cgain[(size_t)c] = (c == sw.h_first[i] || clen < CS) ? mean * (double)clen / (double)CS : std::nan("");
          }
      }
  };
  if (vit) {
    hipStream_t st = b->sV;
    if (vlane && !dev_place) {
      HIPCHK(hipStreamSynchronize(st));
      // item gains -> chunk gains -> binades; one P2 wave per (group, binade) pair
      std::vector<double> &cgain = lw.hs_cgain;
      chunk_gains(cgain);
      std::vector<int> &he = lw.hs_e;
      spec_assign_binades(b, cgain, he);
      // quantised tables of the binades in use
      int emin = INT_MAX, emax = INT_MIN;
      for (int c = 0; c < sw.n_chunks; ++c)
        if (he[(size_t)c] != TEHMM_SPEC_NONE) { emin = std::min(emin, he[(size_t)c]); emax = std::max(emax, he[(size_t)c]); }
      std::vector<int> &wk_g = lw.hs_wkg, &wk_e = lw.hs_wke;
      wk_g.clear();
      wk_e.clear();
      if (emin <= emax) {
        const size_t tsz = quantised_table_size(m, ratio);
        std::vector<double> &qt = lw.hs_qt;
        qt.assign((size_t)(emax - emin + 1) * tsz + 64, 0.0);      // (+ one block of padding: k_vit_lane3's prefetch)
        std::vector<char> eok((size_t)(emax - emin + 1), 1);
        for (int e = emin; e <= emax; ++e)
          eok[(size_t)(e - emin)] = ratio ? quantised_table_ratio(m, e, qt.data() + (size_t)(e - emin) * tsz)
                                          : quantised_table(m, e, qt.data() + (size_t)(e - emin) * tsz);
        for (int c = 0; c < sw.n_chunks; ++c)
          if (he[(size_t)c] != TEHMM_SPEC_NONE && !eok[(size_t)(he[(size_t)c] - emin)]) he[(size_t)c] = TEHMM_SPEC_NONE;
        HIPCHK(lw.qtabs.fill_async(qt.data(), qt.size(), st));
        // a group whose speculated items share one binade is one work unit; the items of the others (interval
        // heads, binade crossings) are pooled per binade and dealt out 64 to a wave
        std::vector<std::vector<int>> pool((size_t)(emax - emin + 1));
        for (int g = 0; g < lw.n_groups; ++g) {
          int e1 = TEHMM_SPEC_NONE;
          bool mixed = false;
          const int nl = std::min(64, lw.n_items - g * 64);
          for (int ln = 0; ln < nl; ++ln) {
            const size_t item = (size_t)g * 64 + ln;
            const int e = he[(size_t)(sw.h_first[lw.h_iv[item]] + lw.h_t0[item] / CS)];
            if (e == TEHMM_SPEC_NONE) continue;
            if (e1 == TEHMM_SPEC_NONE) e1 = e;
            else if (e != e1) mixed = true;
          }
          if (e1 == TEHMM_SPEC_NONE) continue;
          if (!mixed) { wk_g.push_back(g); wk_e.push_back(e1); continue; }
          for (int ln = 0; ln < nl; ++ln) {
            const size_t item = (size_t)g * 64 + ln;
            const int e = he[(size_t)(sw.h_first[lw.h_iv[item]] + lw.h_t0[item] / CS)];
            if (e != TEHMM_SPEC_NONE) pool[(size_t)(e - emin)].push_back((int)item);
          }
        }
        std::vector<int> &wki = lw.hs_wki;
        wki.clear();
        for (int e = emin; e <= emax; ++e) {
          const std::vector<int> &pl = pool[(size_t)(e - emin)];
          for (size_t i0 = 0; i0 < pl.size(); i0 += 64) {
            const int slot = (int)(wki.size() / 64);
            for (size_t i = 0; i < 64; ++i) wki.push_back(i0 + i < pl.size() ? pl[i0 + i] : -1);
            wk_g.push_back(-(1 + slot));
            wk_e.push_back(e);
          }
        }
        if (!wki.empty()) HIPCHK(lw.wk_items.fill_async(wki.data(), wki.size(), st));
      }
      {
        // waves of one binade next to each other: they share one quantised table in the scalar cache
        std::vector<int> ord(wk_g.size());
        std::iota(ord.begin(), ord.end(), 0);
        std::stable_sort(ord.begin(), ord.end(), [&](int x, int y) { return wk_e[(size_t)x] < wk_e[(size_t)y]; });
        std::vector<int> g2(ord.size()), e2(ord.size());
        for (size_t i = 0; i < ord.size(); ++i) { g2[i] = wk_g[(size_t)ord[i]]; e2[i] = wk_e[(size_t)ord[i]]; }
        wk_g.swap(g2);
        wk_e.swap(e2);
      }
      const int n_work = (int)wk_g.size();
      if (n_work > 0) {
        HIPCHK(lw.wk_g.fill_async(wk_g.data(), wk_g.size(), st));
        HIPCHK(lw.wk_e.fill_async(wk_e.data(), wk_e.size(), st));
      }
      HIPCHK(hipMemcpyAsync(sw.e.p, he.data(), he.size() * sizeof(int), hipMemcpyHostToDevice, st));
      HIPCHK(hipMemcpyAsync(sw.gain.p, cgain.data(), cgain.size() * sizeof(double), hipMemcpyHostToDevice, st));
      HIPCHK(hipMemsetAsync(sw.ok.p, 0, he.size() * sizeof(int), st));
      HIPCHK(hipMemsetAsync(sw.stats.p, 0, 2 * sizeof(int), st));
      HIPCHK(hipMemsetAsync(lw.vbad.p, 0, (size_t)std::max(1, lw.n_groups) * 64 * sizeof(int), st));
      HIPCHK(hipMemsetAsync(lw.vntie.p, 0, (size_t)std::max(1, lw.n_groups) * 64 * sizeof(int), st));
      rc = vit_lane_upload_args(b, vc, st);
      if (rc) return rc;
      if (split_post) (void)hipStreamWaitEvent(st, b->evX[1], 0);     // the forward pass itself is through
#define CALL(NT_) launch_vit_lane<NT_>(b, m, iv, vc, true, ratio, WuV, n_work, (const double *)lw.qtabs.p, emin, (const int *)nullptr, st)
      TEHMM_NT_DISPATCH(m->NP, CALL)
#undef CALL
    }
    if (vlane && dev_place) {
      int n_work_max = 0;
      rc = launch_vit_place(b, m, iv, CS, ratio, st, &n_work_max);
      if (rc) return rc;
      if (split_post) (void)hipStreamWaitEvent(st, b->evX[1], 0);     // the forward pass itself is through
#define CALL(NT_) launch_vit_lane<NT_>(b, m, iv, vc, true, ratio, WuV, n_work_max, (const double *)m->qall[ratio ? 1 : 0].p, TEHMM_SPEC_MIN_E, (const int *)&lw.place.p->n_work, st)
      TEHMM_NT_DISPATCH(m->NP, CALL)
#undef CALL
    }
    if (vlane) {
#define CALL(NT_) launch_vit_stitch<NT_>(b, m, iv, vc, st)
      TEHMM_NT_DISPATCH(m->NP, CALL)
#undef CALL
      (void)hipEventRecord(b->ev[eV + 3], st);
      if (defer_post) {
        (void)hipStreamWaitEvent(b->sP, b->ev[eV + 3], 0);
        rc = enqueue_posterior(split_post ? 2 : 0);
        if (rc) return rc;
      }
#define CALL(NT_) launch_vit_fix<NT_>(b, m, iv, em, vc, true, ratio, st)
      TEHMM_NT_DISPATCH(m->NP, CALL)
#undef CALL
    } else if (vspec) {
      HIPCHK(hipStreamSynchronize(st));
      std::vector<int> &he = lw.hs_e;
      if (glane) {
        std::vector<double> &cgain = lw.hs_cgain;
        chunk_gains(cgain);
        spec_assign_binades(b, cgain, he);
        HIPCHK(hipMemcpyAsync(sw.gain.p, cgain.data(), cgain.size() * sizeof(double), hipMemcpyHostToDevice, st));
      } else {
        spec_assign_binades(b, gain, he);
      }
      HIPCHK(hipMemcpyAsync(sw.e.p, he.data(), he.size() * sizeof(int), hipMemcpyHostToDevice, st));
      HIPCHK(hipMemsetAsync(sw.ok.p, 0, he.size() * sizeof(int), st));
      HIPCHK(hipMemsetAsync(sw.stats.p, 0, 2 * sizeof(int), st));
#define CALL(NT_) launch_vit_spec<NT_>(b, m, iv, emg, vc, true, st)
      TEHMM_NT_DISPATCH(m->NP, CALL)
#undef CALL
      (void)hipEventRecord(b->ev[eV + 3], st);
      if (defer_post) {
        (void)hipStreamWaitEvent(b->sP, b->ev[eV + 3], 0);
        rc = enqueue_posterior();
        if (rc) return rc;
      }
#define CALL(NT_) launch_vit_fix<NT_>(b, m, iv, em, vc, false, false, st)
      TEHMM_NT_DISPATCH(m->NP, CALL)
#undef CALL
    } else if (coop) {
#define CALL(NT_) launch_vit_coop<NT_>(b, m, iv, em, ratio, st)
      TEHMM_NT_DISPATCH(m->NP, CALL)
#undef CALL
    } else {
      rc = viterbi_wide_cp(b, m, iv, em, ratio, st, b->ev[eV + 3], &wide_vit);
      if (rc) return rc;
      if (!wide_vit) {
        if (SPL == 1) launch_viterbi<1>(b, m, iv, em, ratio, st);
        else launch_viterbi<2>(b, m, iv, em, ratio, st);
      }
      if (wide_defer) {
        // the posterior passes go behind the quantised Viterbi pass, next to the exact chain (few CUs, latency-bound)
        if (wide_vit) (void)hipStreamWaitEvent(b->sP, b->ev[eV + 3], 0);
        rc = enqueue_posterior();
        if (rc) return rc;
      }
    }
    (void)hipEventRecord(b->ev[eV + 1], st);
    allow_lds(k_tb_compose, 4 * TEHMM_TB_STAGE);
    allow_lds(k_tb_fill, 4 * TEHMM_TB_STAGE);
    if (b->n_chunks > 0)
      hipLaunchKernelGGL(k_tb_compose, dim3((b->n_chunks + 3) / 4), dim3(256), 4 * tb_stage_bytes(b->TBW), st, iv, b->d_chunk_iv.p,
                         b->d_chunk0.p, b->n_chunks, m->N, m->NP, b->TBW, b->tb.p, b->G.p);
    {
      // long intervals: two-level scan (tiles of 64 chunk maps composed in parallel)
      int64_t maxc = 0;
      for (int i = 0; i < b->n; ++i) maxc = std::max<int64_t>(maxc, b->h_chunk0[(size_t)i + 1] - b->h_chunk0[(size_t)i]);
      const int maxt = (int)((maxc + 63) / 64);
      if (maxt > 8 && maxt <= 65535) {          // (grid.y limit: longer intervals keep the one-level scan)
        const size_t ntile = (size_t)b->n_chunks / 64 + (size_t)b->n + 2;
        HIPCHK(b->Gg.ensure(ntile * m->NP));
        HIPCHK(b->tstate.ensure(ntile));
        hipLaunchKernelGGL(k_tb_group, dim3(b->n, maxt), dim3(64), 0, st, iv, b->d_chunk0.p, m->NP, (const uint8_t *)b->G.p,
                           b->Gg.p);
        hipLaunchKernelGGL(k_tb_scan_top, dim3(b->n), dim3(64), 0, st, iv, b->d_chunk0.p, m->NP, (const uint8_t *)b->Gg.p,
                           (const int *)b->last_state.p, b->tstate.p, b->paths.p);
        hipLaunchKernelGGL(k_tb_scan_tiles, dim3(b->n, maxt), dim3(64), 0, st, iv, b->d_chunk0.p, m->NP,
                           (const uint8_t *)b->G.p, (const uint8_t *)b->tstate.p, b->bstate.p);
      } else {
        hipLaunchKernelGGL(k_tb_scan, dim3(std::max(1, b->n)), dim3(64), 0, st, iv, b->d_chunk0.p, m->NP,
                           b->G.p, b->last_state.p, b->bstate.p, b->paths.p);
      }
    }
    if (b->n_chunks > 0)
      hipLaunchKernelGGL(k_tb_fill, dim3((b->n_chunks + 3) / 4), dim3(256), 4 * tb_stage_bytes(b->TBW), st, iv,
                         b->n_chunks, b->d_chunk_iv.p, b->d_chunk0.p, b->TBW, b->tb.p, b->bstate.p,
                         b->paths.p);
    (void)hipEventRecord(b->ev[eV + 2], st);
    if (vlane || glane) {
      b->tnames.push_back("emission_rows");
      b->tpairs.push_back({eV, eV + 4});
      b->tnames.push_back("viterbi_speculate");
      b->tpairs.push_back({eV + 4, eV + 3});
      b->tnames.push_back("viterbi");
      b->tpairs.push_back({eV + 3, eV + 1});
    } else if (vspec) {
      b->tnames.push_back("viterbi_speculate");
      b->tpairs.push_back({eV, eV + 3});
      b->tnames.push_back("viterbi");
      b->tpairs.push_back({eV + 3, eV + 1});
    } else {
      b->tnames.push_back("viterbi");
      b->tpairs.push_back({eV, eV + 1});
    }
    b->tnames.push_back("traceback");
    b->tpairs.push_back({eV + 1, eV + 2});
  }
  rc = finish_wide_posterior();
  if (rc) return rc;
  if (postr) {
    if (flane && fused_fb) {
      b->tnames.push_back("forward_pass");               // lane pass + links + exact forward chain
      b->tpairs.push_back({10, eP + 3});
      b->tnames.push_back("backward_posterior_pass");    // lane pass incl. the posterior rows + links
      b->tpairs.push_back({split_post ? 11 : eP + 3, eP + 1});
    } else if (flane) {
      if (!vlane && !glane) {
        b->tnames.push_back("emission_rows");
        b->tpairs.push_back({eP, eP + 4});
      }
      b->tnames.push_back("forward_backward_speculate");
      b->tpairs.push_back({10, eP + 3});
      b->tnames.push_back("forward_backward");
      b->tpairs.push_back({eP + 3, eP + 1});
    } else if (fspec) {
      b->tnames.push_back("forward_backward_speculate");
      b->tpairs.push_back({eP, eP + 3});
      b->tnames.push_back("forward_backward");
      b->tpairs.push_back({eP + 3, eP + 1});
    } else {
      b->tnames.push_back(coop ? "forward_backward" : "forward");
      b->tpairs.push_back({eP, eP + 1});
    }
    b->tnames.push_back(flane && fused_fb ? "backward_chain" : (coop ? "posterior_combine" : "backward_posterior"));
    b->tpairs.push_back({eP + 1, eP + 2});
  }
  HIPCHK(hipGetLastError());
  if (flags & TEHMM_EVAL_VITERBI) HIPCHK(hipStreamSynchronize(b->sV));
  if (flags & TEHMM_EVAL_POSTERIOR) HIPCHK(hipStreamSynchronize(b->sP));
  for (auto &pr : b->tpairs) {
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, b->ev[pr.first], b->ev[pr.second]);
    b->tms.push_back((double)ms);
  }
  if (flane && std::getenv("TEHMM_SPEC_DEBUG")) {
    const size_t nc = (size_t)sw.n_chunks;
    std::vector<int> re(nc), rb(nc), lf(nc), lb(nc), of(nc), ob(nc);
    HIPCHK(hipMemcpy(re.data(), lw.runend_f.p, nc * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(rb.data(), lw.runstart_b.p, nc * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(lf.data(), lw.link_f.p, nc * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(lb.data(), lw.link_b.p, nc * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(of.data(), lw.ok_f.p, nc * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(ob.data(), lw.ok_b.p, nc * 4, hipMemcpyDeviceToHost));
    for (size_t c = 0; c < nc && c < 48; ++c)
      std::fprintf(stderr, "[fb runs] c %zu ok_f %d link_f %d runend_f %d | ok_b %d link_b %d runstart_b %d\n", c, of[c],
                   lf[c], re[c], ob[c], lb[c], rb[c]);
  }
  if (fspec) {
    int st[4] = {0, 0, 0, 0};
    HIPCHK(hipMemcpy(st, b->sw.stats.p + 2, sizeof(st), hipMemcpyDeviceToHost));
    b->tnames.push_back("count:forward_exact_blocks");
    b->tms.push_back((double)st[0]);
    b->tnames.push_back("count:forward_chunk_jumps");
    b->tms.push_back((double)st[1]);
    b->tnames.push_back("count:backward_exact_blocks");
    b->tms.push_back((double)st[2]);
    b->tnames.push_back("count:backward_chunk_jumps");
    b->tms.push_back((double)st[3]);
  }
  if (wide_cp) {
    b->tnames.push_back("count:wide_chunk_parallel_warmup");
    b->tms.push_back((double)b->ww.wu_ok);
  }
  if (wide_vit) {
    int st2[2] = {0, 0};
    HIPCHK(hipMemcpy(st2, b->sw.stats.p, sizeof(st2), hipMemcpyDeviceToHost));
    b->tnames.push_back("count:viterbi_exact_blocks");
    b->tms.push_back((double)st2[0]);
    b->tnames.push_back("count:viterbi_chunk_jumps");
    b->tms.push_back((double)st2[1]);
  }
  if (vspec) {
    // counters (not times): 64-position blocks the exact chain ran / chunks it could jump over
    int st[4] = {0, 0, 0, 0};
    HIPCHK(hipMemcpy(st, b->sw.stats.p, sizeof(st), hipMemcpyDeviceToHost));
#ifdef TEHMM_CHAIN_PROF
    {
      int pr[4];
      HIPCHK(hipMemcpy(pr, b->sw.stats.p + 8, sizeof(pr), hipMemcpyDeviceToHost));
      HIPCHK(hipMemset(b->sw.stats.p + 8, 0, sizeof(pr)));
      std::fprintf(stderr, "[chain prof] summed over intervals, ms: prologue %.2f steps %.2f barrier %.2f\n", pr[0] * 1e-4,
                   pr[1] * 1e-4, pr[2] * 1e-4);
    }
#endif
    if (std::getenv("TEHMM_SPEC_DEBUG")) {
      std::vector<int> he((size_t)b->sw.n_chunks), hok((size_t)b->sw.n_chunks);
      HIPCHK(hipMemcpy(he.data(), b->sw.e.p, he.size() * sizeof(int), hipMemcpyDeviceToHost));
      HIPCHK(hipMemcpy(hok.data(), b->sw.ok.p, hok.size() * sizeof(int), hipMemcpyDeviceToHost));
      int na = 0, nok = 0;
      for (size_t i = 0; i < he.size(); ++i) {
        na += he[i] != TEHMM_SPEC_NONE;
        nok += he[i] != TEHMM_SPEC_NONE && hok[i];
      }
      std::fprintf(stderr, "[tehmm spec] chunks %d, speculated %d, usable %d, jumped %d, exact blocks %d\n",
                   b->sw.n_chunks, na, nok, st[1], st[0]);
      if (vlane) {
        // what ends the verified runs: ties inside chunks, chunks that do not link to their predecessor
        std::vector<int> hnt(he.size()), hcl(he.size());
        HIPCHK(hipMemcpy(hnt.data(), b->sw.ntie.p, hnt.size() * sizeof(int), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(hcl.data(), b->sw.clink.p, hcl.size() * sizeof(int), hipMemcpyDeviceToHost));
        long ties = 0, tchunks = 0, unlinked = 0, unl_samee = 0;
        for (size_t i = 0; i < he.size(); ++i) {
          if (he[i] == TEHMM_SPEC_NONE || !hok[i]) continue;
          ties += hnt[i];
          tchunks += hnt[i] > 0;
          if (!hcl[i]) {
            ++unlinked;
            if (i > 0 && he[i - 1] == he[i] && hok[i - 1]) ++unl_samee;
          }
        }
        std::fprintf(stderr, "[tehmm spec] ties %ld in %ld chunks, unlinked chunks %ld (%ld next to a usable chunk of the same binade)\n",
                     ties, tchunks, unlinked, unl_samee);
      }
    }
    b->tnames.push_back("count:viterbi_exact_blocks");
    b->tms.push_back((double)st[0]);
    b->tnames.push_back("count:viterbi_chunk_jumps");
    b->tms.push_back((double)st[1]);
  }
  if ((flags & TEHMM_EVAL_VITERBI) && viterbi_logprob)
    HIPCHK(hipMemcpy(viterbi_logprob, b->vit_lp.p, (size_t)b->n * sizeof(double), hipMemcpyDeviceToHost));
  if (flags & TEHMM_EVAL_POSTERIOR) {
    b->h_fwd_lp.resize((size_t)b->n);
    HIPCHK(hipMemcpy(b->h_fwd_lp.data(), b->fwd_lp.p, (size_t)b->n * sizeof(double), hipMemcpyDeviceToHost));
    if (forward_logprob) std::memcpy(forward_logprob, b->h_fwd_lp.data(), (size_t)b->n * sizeof(double));
  }
  return TEHMM_OK;
}

int tehmm_batch_get_interval_logprobs(tehmm_batch_t *b, double *out) {
  if (!b || !out) return fail(TEHMM_ERR_ARG, "tehmm_batch_get_interval_logprobs: bad argument");
  if (b->h_fwd_lp.size() != (size_t)b->n)
    return fail(TEHMM_ERR_ARG, "tehmm_batch_get_interval_logprobs: no forward result in this batch");
  if (b->n > 0) std::memcpy(out, b->h_fwd_lp.data(), (size_t)b->n * sizeof(double));
  return TEHMM_OK;
}

// ---- results over PCIe -------------------------------------------------------------------------------
// A pageable destination makes hipMemcpy stage through the runtime's own bounce buffers (measured round 2: 5.7 GB
// of posteriors at 15 GB/s).  Two ways past that:
//   * tehmm_host_alloc hands out PINNED host memory (cached in a small pool, because pinning is the slow part):
//     a D2H into it is one DMA at link speed;
//   * any other destination is served in 32 MB pieces through two pinned staging buffers -- the DMA of piece i + 1
//     runs while worker threads copy piece i to its destination.
namespace {
struct PinnedPool {
  std::mutex mu;
  std::vector<std::pair<void *, size_t>> free_blocks;      // cached (ptr, bytes)
  std::vector<std::pair<void *, size_t>> live;             // handed out
  size_t cached_bytes = 0;
} g_pinned;
constexpr size_t kPinnedCacheMax = (size_t)24 << 30;
constexpr size_t kStageBytes = (size_t)32 << 20;

bool is_pinned_host(const void *p) {
  hipPointerAttribute_t at;
  if (hipPointerGetAttributes(&at, p) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return at.type == hipMemoryTypeHost;
}

void threaded_copy(void *dst, const void *src, size_t bytes) {
  const int nt = bytes >= ((size_t)8 << 20) ? 4 : 1;
  if (nt == 1) { std::memcpy(dst, src, bytes); return; }
  std::vector<std::thread> th;
  const size_t per = ((bytes / nt) + 4095) & ~(size_t)4095;
  for (int i = 0; i < nt; ++i) {
    const size_t a = std::min(bytes, per * i), e = std::min(bytes, per * (i + 1));
    if (e > a) th.emplace_back([=]() { std::memcpy((char *)dst + a, (const char *)src + a, e - a); });
  }
  for (auto &t : th) t.join();
}

// device -> host of `bytes` bytes on stream `st` (which is synchronised before returning)
int d2h(void *dst, const void *src, size_t bytes, tehmm_batch *b) {
  if (bytes == 0) return TEHMM_OK;
  if (is_pinned_host(dst) || bytes < ((size_t)4 << 20)) {
    // (on the batch's own stream: the legacy default stream would serialise this transfer with every other thread's
    //  synchronous copy -- e.g. the observations of the next batch of engine.eval_stream)
    // In pieces of 64 MB, each waited for before the next is queued: the copy engine serves its queue in order, and a
    // multi-GB transfer queued whole kept every other thread's small copies -- the tables of the next batch, the
    // scalar results of its evaluation -- waiting until it was through (engine.eval_stream: no overlap at all).
    const size_t piece = (size_t)64 << 20;
    for (size_t off = 0; off < bytes; off += piece) {
      const size_t nb = std::min(piece, bytes - off);
      HIPCHK(hipMemcpyAsync((char *)dst + off, (const char *)src + off, nb, hipMemcpyDeviceToHost, b->sB));
      HIPCHK(hipStreamSynchronize(b->sB));
    }
    return TEHMM_OK;
  }
  for (int i = 0; i < 2; ++i)
    if (!b->stage[i]) HIPCHK(hipHostMalloc(&b->stage[i], kStageBytes, hipHostMallocDefault));
  hipStream_t st = b->sP;
  const size_t np = (bytes + kStageBytes - 1) / kStageBytes;
  size_t prev_bytes = 0;
  for (size_t i = 0; i <= np; ++i) {
    if (i < np) {
      const size_t off = i * kStageBytes, nb = std::min(kStageBytes, bytes - off);
      HIPCHK(hipMemcpyAsync(b->stage[i & 1], (const char *)src + off, nb, hipMemcpyDeviceToHost, st));
      (void)hipEventRecord(b->evX[i & 1], st);
      if (i > 0) {                                     // piece i - 1 landed before piece i was queued: copy it out
        HIPCHK(hipEventSynchronize(b->evX[(i - 1) & 1]));
        threaded_copy((char *)dst + (i - 1) * kStageBytes, b->stage[(i - 1) & 1], prev_bytes);
      }
      prev_bytes = nb;
    } else {
      HIPCHK(hipEventSynchronize(b->evX[(i - 1) & 1]));
      threaded_copy((char *)dst + (i - 1) * kStageBytes, b->stage[(i - 1) & 1], prev_bytes);
    }
  }
  return TEHMM_OK;
}
}  // namespace

int tehmm_host_alloc(size_t bytes, void **out) {
  if (!out || bytes == 0) return fail(TEHMM_ERR_ARG, "tehmm_host_alloc: bad argument");
  *out = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_pinned.mu);
    size_t best = SIZE_MAX;
    for (size_t i = 0; i < g_pinned.free_blocks.size(); ++i)
      if (g_pinned.free_blocks[i].second >= bytes && g_pinned.free_blocks[i].second <= 2 * bytes + (1 << 20) &&
          (best == SIZE_MAX || g_pinned.free_blocks[i].second < g_pinned.free_blocks[best].second))
        best = i;
    if (best != SIZE_MAX) {
      auto blk = g_pinned.free_blocks[best];
      g_pinned.free_blocks.erase(g_pinned.free_blocks.begin() + (long)best);
      g_pinned.cached_bytes -= blk.second;
      g_pinned.live.push_back(blk);
      *out = blk.first;
      return TEHMM_OK;
    }
  }
  void *p = nullptr;
  HIPCHK(hipHostMalloc(&p, bytes, hipHostMallocDefault));
  std::lock_guard<std::mutex> lk(g_pinned.mu);
  g_pinned.live.emplace_back(p, bytes);
  *out = p;
  return TEHMM_OK;
}

int tehmm_host_free(void *p) {
  if (!p) return TEHMM_OK;
  std::pair<void *, size_t> blk{nullptr, 0};
  {
    std::lock_guard<std::mutex> lk(g_pinned.mu);
    for (size_t i = 0; i < g_pinned.live.size(); ++i)
      if (g_pinned.live[i].first == p) {
        blk = g_pinned.live[i];
        g_pinned.live.erase(g_pinned.live.begin() + (long)i);
        break;
      }
    if (!blk.first) return fail(TEHMM_ERR_ARG, "tehmm_host_free: not a tehmm_host_alloc block");
    if (g_pinned.cached_bytes + blk.second <= kPinnedCacheMax) {
      g_pinned.free_blocks.push_back(blk);
      g_pinned.cached_bytes += blk.second;
      return TEHMM_OK;
    }
  }
  HIPCHK(hipHostFree(blk.first));
  return TEHMM_OK;
}

int tehmm_trim_pools(void) {
  (void)hipDeviceSynchronize();
  (void)dev_pool_trim();
  std::vector<std::pair<void *, size_t>> blocks;
  {
    std::lock_guard<std::mutex> lk(g_pinned.mu);
    blocks.swap(g_pinned.free_blocks);
    g_pinned.cached_bytes = 0;
  }
  for (auto &blk : blocks) (void)hipHostFree(blk.first);
  return TEHMM_OK;
}

int tehmm_batch_get_paths(tehmm_batch_t *b, int64_t row0, int64_t row1, int64_t *paths) {
  if (!b || !paths || row0 < 0 || row1 < row0 || row1 > b->total)
    return fail(TEHMM_ERR_ARG, "tehmm_batch_get_paths: bad argument");
  if (!b->paths.p) return fail(TEHMM_ERR_ARG, "tehmm_batch_get_paths: no Viterbi result in this batch");
  return d2h(paths, b->paths.p + row0, (size_t)(row1 - row0) * sizeof(int64_t), b);
}

int tehmm_batch_get_posteriors(tehmm_batch_t *b, int64_t row0, int64_t row1, double *post) {
  if (!b || !post || row0 < 0 || row1 < row0 || row1 > b->total)
    return fail(TEHMM_ERR_ARG, "tehmm_batch_get_posteriors: bad argument");
  if (!b->post.p) return fail(TEHMM_ERR_ARG, "tehmm_batch_get_posteriors: no posterior result in this batch");
  return d2h(post, b->post.p + (size_t)row0 * b->N, (size_t)(row1 - row0) * b->N * sizeof(double), b);
}

int tehmm_batch_device_ptrs(tehmm_batch_t *b, void **paths_i64, void **posteriors_f64) {
  if (!b) return fail(TEHMM_ERR_ARG, "tehmm_batch_device_ptrs: NULL handle");
  if (paths_i64) *paths_i64 = b->paths.p;
  if (posteriors_f64) *posteriors_f64 = b->post.p;
  return TEHMM_OK;
}

int tehmm_batch_last_timing(tehmm_batch_t *b, int max_entries, const char **names, double *ms) {
  if (!b) return 0;
  int n = (int)std::min<size_t>(b->tms.size(), (size_t)std::max(0, max_entries));
  for (int i = 0; i < n; ++i) {
    if (names) names[i] = b->tnames[i].c_str();
    if (ms) ms[i] = b->tms[i];
  }
  return n;
}

// ---- output reductions (bin/teHmmEval.py:238-275) ---------------------------------------------
int tehmm_batch_posterior_masksum(tehmm_batch_t *b, const double *mask, int64_t row0, int64_t row1, double *out) {
  if (!b || !mask || !out || row0 < 0 || row1 < row0 || row1 > b->total)
    return fail(TEHMM_ERR_ARG, "tehmm_batch_posterior_masksum: bad argument");
  if (!b->post.p) return fail(TEHMM_ERR_ARG, "tehmm_batch_posterior_masksum: no posterior result in this batch");
  if (row1 == row0) return TEHMM_OK;
  const int64_t rows = row1 - row0;
  DBuf<double> d_mask, d_out;
  HIPCHK(d_mask.upload(mask, (size_t)b->N));
  HIPCHK(d_out.alloc((size_t)rows));
  hipLaunchKernelGGL(k_post_masksum, dim3(grid_for(rows * 64, 256, 256 * 32)), dim3(256), 0, 0, rows, b->N,
                     (const double *)(b->post.p + (size_t)row0 * b->N), (const double *)d_mask.p, d_out.p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(out, d_out.p, (size_t)rows * sizeof(double), hipMemcpyDeviceToHost));
  return TEHMM_OK;
}

int tehmm_bed_coords(int64_t n_rows, int64_t table_start, int64_t table_end, const int64_t *segOffsets,
                     const int32_t *maskOffsets, int64_t n_mask, int64_t *starts, int64_t *ends) {
  if (n_rows < 0 || !starts || !ends || (maskOffsets && n_mask <= 0))
    return fail(TEHMM_ERR_ARG, "tehmm_bed_coords: bad argument");
  if (n_rows == 0) return TEHMM_OK;
  if (maskOffsets) {
    const int64_t dmax = segOffsets ? segOffsets[n_rows - 1] - segOffsets[0] : n_rows - 1;
    if (dmax >= n_mask) return fail(TEHMM_ERR_ARG, "tehmm_bed_coords: maskOffsets shorter than the table");
  }
  DBuf<int64_t> d_seg, d_s, d_e;
  DBuf<int32_t> d_m;
  if (segOffsets) HIPCHK(d_seg.upload(segOffsets, (size_t)n_rows));
  if (maskOffsets) HIPCHK(d_m.upload(maskOffsets, (size_t)n_mask));
  HIPCHK(d_s.alloc((size_t)n_rows));
  HIPCHK(d_e.alloc((size_t)n_rows));
  hipLaunchKernelGGL(k_bed_coords, dim3(grid_for(n_rows, 256)), dim3(256), 0, 0, n_rows, table_start, table_end,
                     (const int64_t *)d_seg.p, (const int32_t *)d_m.p, d_s.p, d_e.p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(starts, d_s.p, (size_t)n_rows * sizeof(int64_t), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(ends, d_e.p, (size_t)n_rows * sizeof(int64_t), hipMemcpyDeviceToHost));
  return TEHMM_OK;
}

// Host-side text writer of the per-row BED lines ("%s\t%d\t%d\t%s\n", teHmmEval.py:266-275): the
// reference formats every line in the interpreter, which dominates its wall-clock at 100 Mb.
// 4th column: names[states[i]] (or the integer state when names is NULL) -- or, when values is not NULL,
// values[i] printed the way Python 2 prints a numpy float64 ("%.12g", plus ".0" for integral values).
int tehmm_write_bed(const char *path, int append, const char *chrom, int64_t n, const int64_t *starts,
                    const int64_t *ends, const int64_t *states, int n_names, const char *const *names,
                    const double *values) {
  if (!path || !chrom || n < 0 || !starts || !ends || (!states && !values))
    return fail(TEHMM_ERR_ARG, "tehmm_write_bed: bad argument");
  FILE *f = std::fopen(path, append ? "a" : "w");
  if (!f) return fail(TEHMM_ERR_ARG, std::string("tehmm_write_bed: cannot open ") + path);
  std::string buf;
  buf.reserve(1 << 22);
  char num[64];
  const size_t clen = std::strlen(chrom);
  int rc = TEHMM_OK;
  for (int64_t i = 0; i < n && rc == TEHMM_OK; ++i) {
    buf.append(chrom, clen);
    int k = std::snprintf(num, sizeof(num), "\t%lld\t%lld\t", (long long)starts[i], (long long)ends[i]);
    buf.append(num, (size_t)k);
    if (values) {
      k = std::snprintf(num, sizeof(num), "%.12g", values[i]);
      bool plain = true;
      for (int q = 0; q < k; ++q) plain = plain && ((num[q] >= '0' && num[q] <= '9') || num[q] == '-');
      buf.append(num, (size_t)k);
      if (plain) buf.append(".0");
    } else if (names) {
      const int64_t st = states[i];
      if (st < 0 || st >= n_names || !names[st]) rc = fail(TEHMM_ERR_ARG, "tehmm_write_bed: state without a name");
      else buf.append(names[st]);
    } else {
      k = std::snprintf(num, sizeof(num), "%lld", (long long)states[i]);
      buf.append(num, (size_t)k);
    }
    buf.push_back('\n');
    if (buf.size() > (1u << 22) - 256) {
      if (std::fwrite(buf.data(), 1, buf.size(), f) != buf.size()) rc = fail(TEHMM_ERR_ARG, "tehmm_write_bed: write failed");
      buf.clear();
    }
  }
  if (rc == TEHMM_OK && !buf.empty() && std::fwrite(buf.data(), 1, buf.size(), f) != buf.size())
    rc = fail(TEHMM_ERR_ARG, "tehmm_write_bed: write failed");
  if (std::fclose(f) != 0 && rc == TEHMM_OK) rc = fail(TEHMM_ERR_ARG, "tehmm_write_bed: close failed");
  return rc;
}

// Array-level Viterbi: one interval, frame input, through the same kernel + traceback.
int tehmm_viterbi(int64_t T, int N, const double *pi, const double *lt, const double *segRatios,
                  const double *frame, int64_t *path, double *logprob) {
  if (T < 0 || N <= 0 || !pi || !lt || !frame || !path || !logprob)
    return fail(TEHMM_ERR_ARG, "tehmm_viterbi: bad argument");
  if (N > kMaxStates) return fail(TEHMM_ERR_UNSUPPORTED, "tehmm_viterbi: N > 128");
  if (T == 0) return TEHMM_OK;
  const int NP = pad_states(N);
  std::vector<double> hlt((size_t)NP * NP, -INFINITY), hpi(NP, -INFINITY);
  for (int i = 0; i < N; ++i) {
    hpi[i] = pi[i];
    for (int j = 0; j < N; ++j) hlt[(size_t)i * NP + j] = lt[(size_t)i * N + j];
  }
  const int64_t Tpad = (T + 63) & ~(int64_t)63;
  const int64_t nch64 = T > 1 ? (T - 1 + TEHMM_TB_CHUNK - 1) / TEHMM_TB_CHUNK : 0;
  if (nch64 > 0x7fffffff) return fail(TEHMM_ERR_UNSUPPORTED, "tehmm_viterbi: T too large");
  const int nch = (int)nch64;
  int64_t h_off[2] = {0, T}, h_pos0[2] = {0, Tpad}, h_len[1] = {T}, h_chunk0[2] = {0, nch};
  int h_order[1] = {0};
  std::vector<int> chunk_iv((size_t)nch, 0);
  DBuf<double> d_lt, d_pi, d_fr, d_r, d_lp;
  DBuf<int64_t> d_off, d_pos0, d_len, d_chunk0, d_paths;
  DBuf<int> d_order, d_chunk_iv, d_last;
  DBuf<uint8_t> d_tb;
  DBuf<uint8_t> d_G, d_bs;
  HIPCHK(d_lt.upload(hlt.data(), hlt.size()));
  HIPCHK(d_pi.upload(hpi.data(), hpi.size()));
  HIPCHK(d_fr.upload(frame, (size_t)T * N));
  if (segRatios) {
    std::vector<double> r((size_t)Tpad, 0.0);
    std::copy(segRatios, segRatios + T, r.begin());
    HIPCHK(d_r.upload(r.data(), r.size()));
  }
  HIPCHK(d_off.upload(h_off, 2));
  HIPCHK(d_pos0.upload(h_pos0, 2));
  HIPCHK(d_len.upload(h_len, 1));
  HIPCHK(d_chunk0.upload(h_chunk0, 2));
  HIPCHK(d_order.upload(h_order, 1));
  HIPCHK(d_chunk_iv.upload(chunk_iv.data(), chunk_iv.size()));
  HIPCHK(d_paths.alloc((size_t)T));
  HIPCHK(d_tb.alloc((size_t)(Tpad + 1) * NP));
  HIPCHK(d_G.alloc((size_t)(nch + 1) * NP));
  HIPCHK(d_bs.alloc((size_t)nch + 1));
  HIPCHK(d_last.alloc(1));
  HIPCHK(d_lp.alloc(1));
  IntervalTab iv;
  iv.order = d_order.p;
  iv.pos0 = d_pos0.p;
  iv.len = d_len.p;
  iv.out0 = d_off.p;
  iv.n = 1;
  EmisTab em;
  std::memset(&em, 0, sizeof(em));
  const int SPL = N <= 64 ? 1 : 2;
  size_t lds = ((size_t)N * NP + (TEHMM_PB + 2) * 64 * SPL) * sizeof(double);
#define VIT_LAUNCH(S_, R_)                                                                          \
  allow_lds(k_viterbi<S_, R_, true>, lds);                                                          \
  hipLaunchKernelGGL((k_viterbi<S_, R_, true>), dim3(1), dim3(64), lds, 0, iv, em, N, NP, d_lt.p,     \
                     d_pi.p, d_r.p, d_fr.p, NP, d_tb.p, d_last.p, d_lp.p)
  if (SPL == 1) {
    if (segRatios) { VIT_LAUNCH(1, true); } else { VIT_LAUNCH(1, false); }
  } else {
    if (segRatios) { VIT_LAUNCH(2, true); } else { VIT_LAUNCH(2, false); }
  }
#undef VIT_LAUNCH
  allow_lds(k_tb_compose, 4 * TEHMM_TB_STAGE);
  allow_lds(k_tb_fill, 4 * TEHMM_TB_STAGE);
  if (nch > 0)
    hipLaunchKernelGGL(k_tb_compose, dim3((nch + 3) / 4), dim3(256), 4 * tb_stage_bytes(NP), 0, iv, d_chunk_iv.p, d_chunk0.p,
                       nch, N, NP, NP, d_tb.p, d_G.p);
  hipLaunchKernelGGL(k_tb_scan, dim3(1), dim3(64), 0, 0, iv, d_chunk0.p, NP, d_G.p, d_last.p, d_bs.p,
                     d_paths.p);
  if (nch > 0)
    hipLaunchKernelGGL(k_tb_fill, dim3((nch + 3) / 4), dim3(256), 4 * tb_stage_bytes(NP), 0, iv, nch,
                       d_chunk_iv.p, d_chunk0.p, NP, d_tb.p, d_bs.p, d_paths.p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(path, d_paths.p, (size_t)T * sizeof(int64_t), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(logprob, d_lp.p, sizeof(double), hipMemcpyDeviceToHost));
  return TEHMM_OK;
}

// ---- fused Baum-Welch E-step ---------------------------------------------------------------
namespace {
constexpr int kEstepChunk = 512;

template <int NT>
void launch_estep(const tehmm_model *m, const IntervalTab &iv, const EmisTab &em, bool ratio, int n_iv,
                  int n_chunks, EstepWork &w, const double *tratios, hipStream_t st) {
  constexpr int CPB = NT <= 44 ? 64 : 32;
  const EmisTab emf = n_iv > 256 ? without_lds_tables(em) : em;
  size_t lds = ((size_t)4 * CPB * (NT + 1) + 4 * CPB + 5 * NT + (size_t)emf.lds_rows * NT) * sizeof(double);
  size_t lds2 = (size_t)std::max(1, m->lds_rows) * NT * sizeof(double) + (size_t)3 * m->K * sizeof(int) + 16;
  if (ratio) {
    allow_lds(k_fb_coop<NT, CPB, true>, lds);
    hipLaunchKernelGGL((k_fb_coop<NT, CPB, true>), dim3(n_iv), dim3(256), lds, st, iv, emf, m->N, m->A.p,
                       m->lt.p, m->pi.p, tratios, w.alpha.p, w.beta.p, w.fwd_lp.p, w.dead.p, w.wrows.p,
                       w.escale.p);
    allow_lds(k_estep_accum<NT, true>, lds2);
    hipLaunchKernelGGL((k_estep_accum<NT, true>), dim3((n_chunks + 3) / 4), dim3(256), lds2, st, iv, em,
                       m->N, kEstepChunk, w.chunk_iv.p, w.chunk_t0.p, n_chunks, w.alpha.p, w.beta.p,
                       w.wrows.p, w.escale.p, w.C, w.D, w.start, w.stat);
  } else {
    allow_lds(k_fb_coop<NT, CPB, false>, lds);
    hipLaunchKernelGGL((k_fb_coop<NT, CPB, false>), dim3(n_iv), dim3(256), lds, st, iv, emf, m->N, m->A.p,
                       m->lt.p, m->pi.p, (const double *)nullptr, w.alpha.p, w.beta.p, w.fwd_lp.p,
                       w.dead.p, w.wrows.p, w.escale.p);
    allow_lds(k_estep_accum<NT, false>, lds2);
    hipLaunchKernelGGL((k_estep_accum<NT, false>), dim3((n_chunks + 3) / 4), dim3(256), lds2, st, iv, em,
                       m->N, kEstepChunk, w.chunk_iv.p, w.chunk_t0.p, n_chunks, w.alpha.p, w.beta.p,
                       w.wrows.p, w.escale.p, w.C, w.D, w.start, w.stat);
  }
}
}  // namespace


// ---- chunk-parallel E-step on the fused posterior passes (tehmm_estep.hip.h) -----------------------------
// Track partition of the reduction kernels (tehmm_estep.hip.h): tracks of at most TEHMM_ESTEP_SMALL rows are
// packed into 16-row tiles for the one-hot product on the matrix cores, the others into LDS histogram groups of
// at most `cap_rows` rows (first fit, decreasing).
// Where the split lies (measured, config-4 model, 50 Mb, reduce stage): tracks of <= 9 rows on the matrix cores
// 21.6 ms, <= 21 rows 17.2, <= 31 (all ten small tracks) 16.3 -- a 16-row tile costs 12 matrix instructions per
// 16-item tile and step (~9 ns per row and CU), a track in LDS ~190-270 ns (nine ds_add_f64 at ~1 lane per cycle,
// more when items share a symbol), and the LDS kernel's workgroups do not share a CU with the 416-register waves
// of the one-hot kernel, so the two add up rather than overlap: break-even near 40 rows.
// TEHMM_ESTEP_SMALL overrides (largest track, in rows, that takes the one-hot product).
#define TEHMM_ESTEP_SMALL_DEFAULT 40
static int estep_small_threshold(const tehmm_model *) {
  if (const char *sm = std::getenv("TEHMM_ESTEP_SMALL")) return std::atoi(sm);
  return TEHMM_ESTEP_SMALL_DEFAULT;
}

static void estep_build_groups(const tehmm_model *m, int cap_rows, EstepGroups &eg) {
  std::memset(&eg, 0, sizeof(eg));
  const int small = estep_small_threshold(m);
  std::vector<int> big;
  int row = 0;
  for (int i = 0; i < TEHMM_ESTEP_MAXRT * 16; ++i) eg.rt_info[i] = -1;
  for (int k = 0; k < m->K; ++k) {
    if (m->rowcnt[k] <= small && row + m->rowcnt[k] <= TEHMM_ESTEP_MAXRT * 16) {
      for (int sy = 0; sy < m->rowcnt[k]; ++sy, ++row) {
        eg.rt_info[row] = (k & 255) | (sy << 8);
        eg.rt_grow[row] = m->rowbase[k] + sy;
      }
    } else {
      big.push_back(k);
    }
  }
  eg.n_rt = (row + 15) / 16;
  std::stable_sort(big.begin(), big.end(), [&](int a, int b2) { return m->rowcnt[a] > m->rowcnt[b2]; });
  std::vector<std::vector<int>> bins;
  std::vector<int> load;
  for (int k : big) {
    size_t g = 0;
    while (g < bins.size() && load[g] + m->rowcnt[k] > cap_rows) ++g;
    if (g == bins.size()) { bins.emplace_back(); load.push_back(0); }
    bins[g].push_back(k);
    load[g] += m->rowcnt[k];
  }
  int slot = 0;
  eg.n_lds = (int)bins.size();
  for (size_t g = 0; g < bins.size(); ++g) {
    eg.first[g] = slot;
    int lrow = 0;
    for (int k : bins[g]) {
      eg.info[slot] = (k & 127) | ((m->rowcnt[k] & 511) << 7) | (lrow << 16);
      eg.gbase[slot] = m->rowbase[k];
      lrow += m->rowcnt[k];
      ++slot;
    }
    eg.rows[g] = lrow;
  }
  eg.first[bins.size()] = slot;
}

static bool estep_fused_wanted(const tehmm_model *m, const tehmm_batch *b, bool ratio, int CS) {
  const char *s = std::getenv("TEHMM_ESTEP_FUSED");
  if (s && std::atoi(s) == 0) return false;
  if (ratio || m->N >= 64 || m->NP > 64 || CS <= 0 || b->total < 2 * (int64_t)CS) return false;
  if (!m->ptab.p || m->K > 78 || m->K > 127) return false;
  for (int k = 0; k < m->K; ++k)
    if (m->rowcnt[k] > 511) return false;
  return true;
}

template <int NT>
static int launch_estep_reduce(tehmm_batch *b, const tehmm_model *m, const IntervalTab &iv, double *dev_stats,
                               hipStream_t st) {
  LaneWork &lw = b->lw;
  EstepWork &w = b->ew;
  const LaneGeom lg = lane_geom(lw);
  const EstepGroups &eg = w.h_groups;
  const int64_t n_tiles = (int64_t)lw.n_groups * 4;
  double *gC = dev_stats + stats_off_C(m->NP), *gstart = dev_stats + stats_off_start(),
         *gstat = dev_stats + stats_off_stat(m->NP);
  // the three reductions are independent: the LDS-atomic histograms (LDS pipe) run next to the two matrix-core
  // products on their own streams.  Every writer owns a slot of a partial buffer; the fold kernels add the slots in
  // order (tehmm_estep.hip.h: reproducible sums).
  (void)hipEventRecord(b->evX[0], st);
  const int gxm = (int)std::max<int64_t>(1, std::min<int64_t>((n_tiles + 3) / 4, 512));
  const int cells_xi = NT * NT + NT;
  HIPCHK(w.part_xi.ensure((size_t)gxm * 4 * cells_xi));
  int gx = 0, gxh = 0, max_rows = 0, tot_rows = 0;
  if (eg.n_lds > 0) {
    for (int g = 0; g < eg.n_lds; ++g) { max_rows = std::max(max_rows, eg.rows[g]); tot_rows += eg.rows[g]; }
    const size_t lds = (size_t)max_rows * NT * sizeof(double) + (size_t)m->K * sizeof(int) + 16;
    allow_lds(k_estep_hist_lds<NT>, lds);
    gx = (int)std::max<int64_t>(1, std::min<int64_t>((n_tiles + 7) / 8, 256));
    HIPCHK(w.part_lds.ensure((size_t)gx * tot_rows * NT));
    // positions one workgroup can add into one cell: its tiles x 16 items x L
    const int64_t per_wg = ((n_tiles + (int64_t)gx * 8 - 1) / ((int64_t)gx * 8)) * 8 * 16 * (int64_t)lw.L;
    int shift = 62;
    while (shift > 20 && (double)per_wg * std::ldexp(1.0, shift) >= 9.2e18) --shift;
    (void)hipStreamWaitEvent(b->sV, b->evX[0], 0);
    hipLaunchKernelGGL((k_estep_hist_lds<NT>), dim3(gx, eg.n_lds), dim3(512), lds, b->sV, iv, lg,
                       (const EstepGroups *)w.d_groups.p, m->N, b->KP, (const uint8_t *)b->obs.p,
                       (const float *)lw.GAM32.p, w.part_lds.p, shift);
    hipLaunchKernelGGL(k_estep_fold_lds, dim3((max_rows * NT + 255) / 256, eg.n_lds), dim3(256), 0, b->sV,
                       (const double *)w.part_lds.p, gx, m->N, NT, (const EstepGroups *)w.d_groups.p, gstat);
    (void)hipEventRecord(b->ev[11], b->sV);
  }
  if (eg.n_rt > 0) {
    constexpr int RTG = EstepGeom<NT>::RTG;
    (void)hipStreamWaitEvent(b->sB, b->evX[0], 0);
    gxh = (int)std::max<int64_t>(1, std::min<int64_t>(n_tiles, 768));     // one tile per workgroup at a time, 3 per CU
    HIPCHK(w.part_rt.ensure((size_t)gxh * eg.n_rt * 16 * NT));
    hipLaunchKernelGGL((k_estep_hist_mfma<NT>), dim3(gxh, (eg.n_rt + RTG - 1) / RTG), dim3(TEHMM_ESTEP_HW * 64), 0, b->sB, iv, lg,
                       (const EstepGroups *)w.d_groups.p, m->N, b->KP, (const uint8_t *)b->obs.p,
                       (const float *)lw.GAM32.p, w.part_rt.p);
    hipLaunchKernelGGL(k_estep_fold_rows, dim3((eg.n_rt * 16 * NT + 255) / 256), dim3(256), 0, b->sB,
                       (const double *)w.part_rt.p, gxh, m->N, NT, (const EstepGroups *)w.d_groups.p, gstat);
    (void)hipEventRecord(b->evX[1], b->sB);
  }
  hipLaunchKernelGGL((k_estep_xi<NT>), dim3(gxm), dim3(256), 0, st, iv, lg, m->N, (const float *)lw.AL32.p,
                     (const float *)lw.GAM32.p, (const float *)lw.WZ32.p, w.part_xi.p);
  hipLaunchKernelGGL(k_estep_fold_xi, dim3((cells_xi + 255) / 256), dim3(256), 0, st, (const double *)w.part_xi.p, gxm * 4,
                     m->N, NT, gC, gstart);
  if (eg.n_lds > 0) (void)hipStreamWaitEvent(st, b->ev[11], 0);
  if (eg.n_rt > 0) (void)hipStreamWaitEvent(st, b->evX[1], 0);
  return TEHMM_OK;
}

// returns TEHMM_OK and *done = true when the fused path ran; *done = false: the caller falls back
static int estep_fused(tehmm_model_t *m, tehmm_batch_t *b, double *dev_stats, double *lp_out, int *dead_out, bool *done) {
  *done = false;
  const int CS = spec_chunk_size();
  if (!estep_fused_wanted(m, b, false, CS)) return TEHMM_OK;
  const int LS = lane_sub_size(CS, b->total);
  if (LS <= 0) return TEHMM_OK;
  LaneWork &lw = b->lw;
  EstepWork &w = b->ew;
  if (!(lw.AL32.p && lw.GAM32.p && lw.L == LS && lw.CS == CS && lw.NP == m->NP)) {
    // three float rows + index records per position must fit; otherwise the grouped sequential path
    size_t free_b = 0, total_b = 0;
    HIPCHK(dev_mem_info(&free_b, &total_b));
    const double per = (double)b->total * (3.0 * 32.0 * al32_pairs(m->NP) + 40.0) * 1.15;
    if (per > 0.85 * (double)free_b) return TEHMM_OK;
  }
  int rc = spec_prepare(b, m, CS);
  if (rc) return rc;
  rc = lane_prepare(b, m, CS, LS, true, false, false, true, false);
  if (rc) return rc;
  const size_t na = (size_t)std::max(1, lw.n_groups) * LS * 512 * al32_pairs(m->NP);
  if (!lw.GAM32.p) {
    HIPCHK(lw.GAM32.alloc(na));
    HIPCHK(lw.WZ32.alloc(na));
    HIPCHK(hipMemset(lw.GAM32.p, 0, na * sizeof(float)));      // (slots beyond an interval's end are read, never written)
    HIPCHK(hipMemset(lw.WZ32.p, 0, na * sizeof(float)));
  }
  if (!b->fwd_lp.p || !b->dead.p) {
    HIPCHK(b->fwd_lp.alloc((size_t)b->n + 1));
    HIPCHK(b->dead.alloc((size_t)b->n + 1));
    if (!b->first_good.p) HIPCHK(b->first_good.alloc((size_t)b->n + 1));
  }
  if (w.groups_model != m->uid || !w.d_groups.p) {
    const int cap_rows = (int)((160 * 1024 - (size_t)m->K * sizeof(int) - 64) / ((size_t)m->NP * sizeof(double)));
    estep_build_groups(m, cap_rows, w.h_groups);
    HIPCHK(w.d_groups.upload(&w.h_groups, 1));
    w.groups_model = m->uid;
  }
  IntervalTab iv;
  EmisTab em;
  fill_tabs(m, b, iv, em, false);
  SpecWork &sw = b->sw;
  int WuF = 64;
  rc = fb_warmup(b, m, LS, iv, without_lds_tables(em), &WuF);
  if (rc) return rc;
  FbChunks fc{};
  fc.iv = sw.iv.p; fc.t0 = sw.t0.p; fc.first = sw.first.p; fc.n = sw.n_chunks; fc.CS = CS;
  fc.scale = sw.scale.p; fc.wstart = sw.wstart.p;
  fc.link_f = lw.link_f.p; fc.glog_f = lw.glog_f.p; fc.link_b = lw.link_b.p; fc.runend_f = lw.runend_f.p;
  fc.pre_f = lw.cpre_f.p; fc.runstart_b = lw.runstart_b.p;
  hipStream_t st = b->sP;
  b->tnames.clear();
  b->tpairs.clear();
  b->tms.clear();
  (void)hipEventRecord(b->ev[10], st);
  (void)hipMemsetAsync(b->dead.p, 0, (size_t)(b->n + 1) * sizeof(int), st);
  (void)hipMemsetAsync(sw.stats.p + 2, 0, 4 * sizeof(int), st);
  int rcf = TEHMM_OK;
#define CALL(NT_) rcf = launch_fused_fb<NT_>(b, m, iv, em, fc, WuF, st, b->ev[8], b->ev[6], true)
  TEHMM_NT_DISPATCH(m->NP, CALL)
#undef CALL
  if (rcf) return rcf;
  (void)hipEventRecord(b->ev[7], st);
  int rcr = TEHMM_OK;
#define CALL(NT_) rcr = launch_estep_reduce<NT_>(b, m, iv, dev_stats, st)
  TEHMM_NT_DISPATCH(m->NP, CALL)
#undef CALL
  if (rcr) return rcr;
  (void)hipEventRecord(b->ev[9], st);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(st));
  static const struct { const char *name; int a, b2; } stages[] = {
      {"estep_forward_pass", 10, 8}, {"estep_backward_pass", 8, 6}, {"estep_backward_chain", 6, 7}, {"estep_reduce", 7, 9}};
  for (const auto &sg : stages) {
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, b->ev[sg.a], b->ev[sg.b2]);
    b->tnames.push_back(sg.name);
    b->tms.push_back((double)ms);
  }
  {
    int cnt[4] = {0, 0, 0, 0};
    HIPCHK(hipMemcpy(cnt, sw.stats.p + 2, sizeof(cnt), hipMemcpyDeviceToHost));
    static const char *names[4] = {"count:forward_exact_blocks", "count:forward_chunk_jumps",
                                   "count:backward_exact_blocks", "count:backward_chunk_jumps"};
    for (int i = 0; i < 4; ++i) {
      b->tnames.push_back(names[i]);
      b->tms.push_back((double)cnt[i]);
    }
  }
  std::vector<double> lp((size_t)b->n);
  std::vector<int> dead((size_t)b->n);
  HIPCHK(hipMemcpy(lp.data(), b->fwd_lp.p, (size_t)b->n * sizeof(double), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(dead.data(), b->dead.p, (size_t)b->n * sizeof(int), hipMemcpyDeviceToHost));
  double lp_total = 0.0;
  int dead_any = 0;
  for (int id = 0; id < b->n; ++id) {
    if (b->h_len[(size_t)id] <= 0) continue;
    lp_total += lp[(size_t)id];
    dead_any |= dead[(size_t)id];
    b->h_fwd_lp[(size_t)id] = dead[(size_t)id] ? std::nan("") : lp[(size_t)id];
  }
  *lp_out = lp_total;
  *dead_out = dead_any;
  *done = true;
  return TEHMM_OK;
}

// ---- E-step on the item-parallel passes of tehmm_wide.hip.h (tehmm_wide_estep.hip.h): 64 <= N <= 128 states, and
// segment ratios at any N <= 128 ---------------------------------------------------------------------------------
static int wide_estep_npw(int N) {
  static const int sizes[] = {16, 32, 48, 64, 80, 96, 112, 128};
  for (int sz : sizes)
    if (N <= sz) return sz;
  return 0;
}

static bool estep_wide_wanted(const tehmm_model *m, const tehmm_batch *b, bool ratio) {
  int mode = 1;                                     // 0: off, 1: where the fused passes do not apply, 2: everywhere
  if (const char *s = std::getenv("TEHMM_ESTEP_WIDE")) mode = std::atoi(s);
  if (mode == 0 || m->N > 128 || b->total < 1024) return false;
  if (mode == 1 && !(ratio || m->N >= 64)) return false;
  if (m->K > 255) return false;
  return true;
}

template <int NPW>
static int launch_wide_estep_reduce(tehmm_batch *b, const tehmm_model *m, const IntervalTab &iv, const LaneGeom &lg, bool ratio,
                                    double *dev_stats, hipStream_t st) {
  WideWork &w = b->ww;
  using G = WideEstepGeom<NPW>;
  const int64_t n_tiles = ((int64_t)w.n_items + 15) / 16;
  double *gC = dev_stats + stats_off_C(m->NP), *gD = dev_stats + stats_off_D(m->NP), *gstart = dev_stats + stats_off_start(),
         *gstat = dev_stats + stats_off_stat(m->NP);
  // work units = item tiles x SQ position ranges: at least ~4096 of them (a 2 Mb batch has 1000 item tiles; the GPU
  // 1024 SIMDs), ranges of 16 positions or more
  int SQ = 1;
  while (SQ < 8 && n_tiles * SQ < 4096 && w.L / (2 * SQ) >= 16 && w.L % (2 * SQ) == 0) SQ *= 2;
  if (const char *sq = std::getenv("TEHMM_WIDE_ESTEP_SQ")) {
    const int v = std::atoi(sq);
    if (v >= 1 && w.L % v == 0) SQ = v;
  }
  const int64_t n_units = n_tiles * SQ;
  // xi: two workgroups per CU at most; gamma product: one unit per workgroup at a time, partial buffer <= 128 MB
  const int gxm = (int)std::max<int64_t>(1, std::min<int64_t>((n_units + G::TPW - 1) / G::TPW, 512));
  HIPCHK(w.part_xi.ensure((size_t)gxm * G::TPW * NPW * NPW));
  const int nrt = w.h_rows.n_rt;
  const int64_t slot_rows = (int64_t)nrt * 16 * NPW * 8;
  // gamma product: four units per workgroup at a time (one per wave), up to three row tiles per wave; partial buffer <= 128 MB
  const int rtw = std::min(nrt, NPW <= 64 ? 4 : 3);      // (every group of row tiles reads the gamma rows once more)
  const int gxr = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>((n_units + 3) / 4, 256), std::max<int64_t>(64, ((int64_t)32 << 20) / slot_rows)));
  HIPCHK(w.part_rows.ensure((size_t)gxr * 4 * nrt * 16 * NPW));
  // the products are independent: the gamma product and the LDS histograms run on their own streams next to the xi
  // product (TEHMM_WIDE_ESTEP_SERIAL=1: one after the other, for profiling)
  const bool serial = std::getenv("TEHMM_WIDE_ESTEP_SERIAL") != nullptr;
  hipStream_t sB = serial ? st : b->sB, sV = serial ? st : b->sV;
  (void)hipEventRecord(b->evX[0], st);
  (void)hipStreamWaitEvent(sB, b->evX[0], 0);
  {
    const dim3 gr(gxr, (nrt + rtw - 1) / rtw);
#define ROWS(R_, W_)                                                                                                    \
    hipLaunchKernelGGL((k_wide_estep_rows<NPW, R_, W_>), gr, dim3(256), 0, sB, iv, lg, (const WideRows *)w.d_rows.p, b->KP, \
                       (const uint8_t *)b->obs.p, (const double *)(R_ ? b->ratios.p : nullptr), (const float *)w.GAM.p,  \
                       w.part_rows.p, SQ)
    if constexpr (NPW <= 64) {
      if (rtw == 4) { if (ratio) ROWS(true, 4); else ROWS(false, 4); }
    }
    if (rtw < 4) {
      if (ratio) { if (rtw == 1) ROWS(true, 1); else if (rtw == 2) ROWS(true, 2); else ROWS(true, 3); }
      else { if (rtw == 1) ROWS(false, 1); else if (rtw == 2) ROWS(false, 2); else ROWS(false, 3); }
    }
#undef ROWS
  }
  hipLaunchKernelGGL(k_wide_fold_rows, dim3((nrt * 16 * m->N + 255) / 256), dim3(256), 0, sB, (const double *)w.part_rows.p,
                     gxr * 4, m->N, NPW, m->NP, (const WideRows *)w.d_rows.p, gstat, gstart, gD);
  (void)hipEventRecord(b->evX[1], sB);
  if (w.h_lds.n_trk > 0) {
    // tracks with many symbols: fixed-point histograms privatised in LDS, on a third stream
    const int nsplit = w.lds_nsplit, HS = NPW / nsplit;
    int max_rows = 0, tot_rows = 0;
    for (int t = 0; t < w.h_lds.n_trk; ++t) { max_rows = std::max(max_rows, w.h_lds.rows[t]); tot_rows += w.h_lds.rows[t]; }
    const size_t lds = (size_t)max_rows * HS * sizeof(unsigned long long);
    const int gxl = (int)std::max<int64_t>(1, std::min<int64_t>((n_units + 7) / 8, 256));
    HIPCHK(w.part_lds.ensure((size_t)gxl * tot_rows * NPW));
    // the largest sum one workgroup can reach in a cell: its units x 16 items x L / SQ positions x the largest ratio
    const int64_t per_wg = ((n_units + (int64_t)gxl * 8 - 1) / ((int64_t)gxl * 8)) * 8 * 16 * (int64_t)(w.L / SQ);
    const double vmax = (double)per_wg * (ratio ? std::max(1.0, w.ratio_max) : 1.0);
    int shift = 62;
    while (shift > 20 && vmax * std::ldexp(1.0, shift) >= 9.2e18) --shift;
    (void)hipStreamWaitEvent(sV, b->evX[0], 0);
    const dim3 gl(gxl, w.h_lds.n_trk * nsplit);
#define LDSK(NS_, R_)                                                                                                  \
    do {                                                                                                               \
      allow_lds(k_wide_estep_hist_lds<NPW, NS_, R_>, lds);                                                             \
      hipLaunchKernelGGL((k_wide_estep_hist_lds<NPW, NS_, R_>), gl, dim3(512), lds, sV, iv, lg,                     \
                         (const WideLds *)w.d_lds.p, m->N, b->KP, (const uint8_t *)b->obs.p,                           \
                         (const double *)(R_ ? b->ratios.p : nullptr), (const float *)w.GAM.p, w.part_lds.p, shift, SQ); \
    } while (0)
    if (nsplit == 1) { if (ratio) LDSK(1, true); else LDSK(1, false); }
    else { if (ratio) LDSK(2, true); else LDSK(2, false); }
#undef LDSK
    hipLaunchKernelGGL(k_wide_fold_lds, dim3((max_rows * HS + 255) / 256, w.h_lds.n_trk * nsplit), dim3(256), 0, sV,
                       (const double *)w.part_lds.p, gxl, m->N, m->NP, HS, nsplit, (const WideLds *)w.d_lds.p, gstat);
    (void)hipEventRecord(b->ev[11], sV);
  }
  hipLaunchKernelGGL((k_wide_estep_xi<NPW>), dim3(gxm), dim3(256), 0, st, iv, lg, (const float *)w.AL.p, (const float *)w.WZ.p,
                     w.part_xi.p, SQ);
  hipLaunchKernelGGL(k_wide_fold_xi, dim3((m->N * m->N + 255) / 256), dim3(256), 0, st, (const double *)w.part_xi.p, gxm * G::TPW,
                     m->N, NPW, m->NP, gC);
  (void)hipStreamWaitEvent(st, b->evX[1], 0);
  if (w.h_lds.n_trk > 0) (void)hipStreamWaitEvent(st, b->ev[11], 0);
  return TEHMM_OK;
}

// returns TEHMM_OK and *done = true when the path ran; *done = false: the caller falls back
static int estep_wide(tehmm_model_t *m, tehmm_batch_t *b, bool ratio, double *dev_stats, double *lp_out, int *dead_out, bool *done) {
  *done = false;
  if (!estep_wide_wanted(m, b, ratio)) return TEHMM_OK;
  const int NPW = wide_estep_npw(m->N);
  if (NPW <= 0) return TEHMM_OK;
  WideWork &w = b->ww;
  if (!w.h_flags) HIPCHK(hipHostMalloc((void **)&w.h_flags, 64, hipHostMallocDefault));
  // item length: as posterior_wide_cp (enough items to give every SIMD a tile, 64 <= L <= 512, multiple of 32)
  int L = (int)std::min<int64_t>(512, std::max<int64_t>(64, (b->total / (16 * 1024) + 31) & ~31));
  if (const char *ls = std::getenv("TEHMM_WIDE_SUB")) L = std::max(32, (std::atoi(ls) + 31) & ~31);
  if (!(w.GAM.p && w.L == L && w.NPW == NPW)) {
    // emission rows + three float rows per position must fit
    size_t free_b = 0, total_b = 0;
    HIPCHK(dev_mem_info(&free_b, &total_b));
    if ((double)b->total * NPW * 22.0 + (double)(b->total / L + b->n + 64) * NPW * 40.0 > 0.85 * (double)free_b) return TEHMM_OK;
  }
  {
    bool fits = true;
    if (int rcg = wide_geometry(b, NPW, L, &fits)) return rcg;
    if (!fits) return TEHMM_OK;
  }
  const size_t na = (size_t)std::max(1, w.n_groups) * 64 * L * (NPW / 4) * 4;
  if (!w.GAM.p) {
    HIPCHK(w.GAM.alloc(na));
    HIPCHK(w.WZ.alloc(na));
    w.al_zeroed = false;
  }
  if (!w.al_zeroed) {       // slots no pass writes (beyond an item's end) are read by the reductions: zero once
    HIPCHK(hipMemset(w.AL.p, 0, na * sizeof(float)));
    HIPCHK(hipMemset(w.GAM.p, 0, na * sizeof(float)));
    HIPCHK(hipMemset(w.WZ.p, 0, na * sizeof(float)));
    w.al_zeroed = true;
  }
  if (!b->fwd_lp.p || !b->dead.p) {
    HIPCHK(b->fwd_lp.alloc((size_t)b->n + 1));
    HIPCHK(b->dead.alloc((size_t)b->n + 1));
    if (!b->first_good.p) HIPCHK(b->first_good.alloc((size_t)b->n + 1));
  }
  if (w.rows_model != m->uid || w.rows_npw != NPW || !w.d_rows.p) {
    // tracks of more than TEHMM_ESTEP_SMALL rows whose histogram [rows][NPW or NPW / 2] fits a CU's LDS go there,
    // the others (and START / DIAG) through the one-hot product
    WideRows &wr = w.h_rows;
    WideLds &wl = w.h_lds;
    std::memset(&wl, 0, sizeof(wl));
    for (int i = 0; i < TEHMM_ESTEP_MAXRT * 16; ++i) { wr.info[i] = -1; wr.grow[i] = 0; }
    const int small = estep_small_threshold(m);
    constexpr size_t kLdsCap = 150 * 1024;
    int big_rows = 0;
    for (int k = 0; k < m->K; ++k)
      if (m->rowcnt[k] > small) big_rows = std::max(big_rows, m->rowcnt[k]);
    w.lds_nsplit = (size_t)big_rows * NPW * 8 <= kLdsCap ? 1 : 2;
    int row = 0;
    wr.info[row++] = TEHMM_WIDE_ROW_START;
    wr.info[row++] = TEHMM_WIDE_ROW_DIAG;
    for (int k = 0; k < m->K; ++k) {
      const bool to_lds = m->rowcnt[k] > small && (size_t)m->rowcnt[k] * (NPW / w.lds_nsplit) * 8 <= kLdsCap;
      if (to_lds) {
        wl.col[wl.n_trk] = k;
        wl.rows[wl.n_trk] = m->rowcnt[k];
        wl.gbase[wl.n_trk] = m->rowbase[k];
        ++wl.n_trk;
        continue;
      }
      for (int sy = 0; sy < m->rowcnt[k]; ++sy, ++row) {
        if (row >= TEHMM_ESTEP_MAXRT * 16) return TEHMM_OK;        // (more rows than the row table holds: fall back)
        wr.info[row] = (k & 255) | (sy << 8);
        wr.grow[row] = m->rowbase[k] + sy;
      }
    }
    wr.n_rt = (row + 15) / 16;
    HIPCHK(w.d_rows.upload(&wr, 1));
    HIPCHK(w.d_lds.upload(&wl, 1));
    w.rows_model = m->uid;
    w.rows_npw = NPW;
  }
  if (ratio && w.h_lds.n_trk > 0 && w.ratio_max <= 0.0) {
    // the fixed-point scale of the LDS histograms needs the largest ratio of the batch (once per batch)
    if (!w.d_rmax.p) HIPCHK(w.d_rmax.alloc(1));
    HIPCHK(hipMemset(w.d_rmax.p, 0, sizeof(unsigned long long)));
    hipLaunchKernelGGL(k_ratio_max, dim3(grid_for(b->total_pad, 256, 2048)), dim3(256), 0, b->sP, (const double *)b->ratios.p,
                       b->total_pad, w.d_rmax.p);
    unsigned long long bits = 0;
    HIPCHK(hipMemcpyAsync(&bits, w.d_rmax.p, sizeof(bits), hipMemcpyDeviceToHost, b->sP));
    HIPCHK(hipStreamSynchronize(b->sP));
    double rmax = 0.0;
    std::memcpy(&rmax, &bits, sizeof(rmax));
    if (!(rmax < 1e12)) return TEHMM_OK;             // (inf / NaN ratios: the sequential kernels own their semantics)
    w.ratio_max = std::max(rmax, 1e-300);
  }
  IntervalTab iv;
  EmisTab em;
  fill_tabs(m, b, iv, em, ratio);     // fit applies the ratios to emissions too (basehmm.py:510)
  LaneGeom lg;
  lg.item_iv = w.item_iv.p; lg.item_t0 = w.item_t0.p; lg.ifirst = w.ifirst.p;
  lg.n_items = w.n_items; lg.n_groups = w.n_groups; lg.L = L;
  hipStream_t st = b->sP;
  b->tnames.clear();
  b->tpairs.clear();
  b->tms.clear();
  (void)hipEventRecord(b->ev[10], st);
  HIPCHK(hipMemsetAsync(w.flags.p, 0, 4 * sizeof(int), st));
  launch_wide_emis(b, m, iv, em, lg, ratio ? 3 : 2, nullptr, st);
  (void)hipEventRecord(b->ev[12], st);
  (void)hipEventRecord(b->ev[9], st);
  // warm-up: what the last E-step with this model handle needed (the parameters move a little per iteration; the links
  // are verified whatever the guess), 64 positions to begin with
  constexpr int kWuMax = 1024;
  int Wu = 64;
  if (const char *wus = std::getenv("TEHMM_LANE_WARMUP")) Wu = std::min(kWuMax, std::max(1, std::atoi(wus)));
  else if (w.wu_ok > 0 && w.wu_model == m->uid) Wu = w.wu_ok;
  int attempts = 0;
  for (;;) {
    if (int rc = wide_post_attempt(b, m, iv, Wu, st, b->ev[8], true)) return rc;
    w.pp_active = false;
    ++attempts;
    HIPCHK(hipStreamSynchronize(st));
    if (std::getenv("TEHMM_SPEC_DEBUG"))
      std::fprintf(stderr, "[tehmm wide estep] NPW %d L %d Wu %d: impossible rows in %d items, failed links %d of %d items\n", NPW,
                   L, Wu, w.h_flags[0], w.h_flags[1], w.n_items);
    if (w.h_flags[0] > 0) return TEHMM_OK;             // impossible rows: the sequential kernels own their semantics
    if (w.h_flags[1] == 0) break;
    if (Wu >= kWuMax) return TEHMM_OK;                 // does not forget: the caller falls back
    Wu = std::min(kWuMax, 2 * Wu);
    HIPCHK(hipMemsetAsync(w.flags.p, 0, 4 * sizeof(int), st));
    (void)hipEventRecord(b->ev[9], st);
  }
  w.wu_ok = Wu;
  w.wu_model = m->uid;
  w.wu_version = m->version;
  (void)hipEventRecord(b->ev[6], st);
  hipLaunchKernelGGL(k_wide_loglik, dim3((b->n + 3) / 4), dim3(256), 0, st, iv, lg, m->N, NPW, (const double *)w.end_f.p,
                     (const double *)w.SL.p, (const double *)w.lr.p, b->fwd_lp.p);
  int rcr = TEHMM_OK;
  switch (NPW) {
    case 16: rcr = launch_wide_estep_reduce<16>(b, m, iv, lg, ratio, dev_stats, st); break;
    case 32: rcr = launch_wide_estep_reduce<32>(b, m, iv, lg, ratio, dev_stats, st); break;
    case 48: rcr = launch_wide_estep_reduce<48>(b, m, iv, lg, ratio, dev_stats, st); break;
    case 64: rcr = launch_wide_estep_reduce<64>(b, m, iv, lg, ratio, dev_stats, st); break;
    case 80: rcr = launch_wide_estep_reduce<80>(b, m, iv, lg, ratio, dev_stats, st); break;
    case 96: rcr = launch_wide_estep_reduce<96>(b, m, iv, lg, ratio, dev_stats, st); break;
    case 112: rcr = launch_wide_estep_reduce<112>(b, m, iv, lg, ratio, dev_stats, st); break;
    default: rcr = launch_wide_estep_reduce<128>(b, m, iv, lg, ratio, dev_stats, st); break;
  }
  if (rcr) return rcr;
  (void)hipEventRecord(b->ev[7], st);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(st));
  // stage times (forward / backward: of the LAST attempt)
  static const struct { const char *name; int a, b2; } stages[] = {
      {"estep_emission_rows", 10, 12}, {"estep_forward_pass", 9, 8}, {"estep_backward_pass", 8, 6}, {"estep_reduce", 6, 7}};
  for (const auto &sg : stages) {
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, b->ev[sg.a], b->ev[sg.b2]);
    b->tnames.push_back(sg.name);
    b->tms.push_back((double)ms);
  }
  b->tnames.push_back("count:wide_estep_attempts");
  b->tms.push_back((double)attempts);
  b->tnames.push_back("count:wide_estep_warmup");
  b->tms.push_back((double)Wu);
  std::vector<double> lp((size_t)b->n);
  HIPCHK(hipMemcpy(lp.data(), b->fwd_lp.p, (size_t)b->n * sizeof(double), hipMemcpyDeviceToHost));
  double lp_total = 0.0;
  for (int id = 0; id < b->n; ++id) {
    if (b->h_len[(size_t)id] <= 0) continue;
    lp_total += lp[(size_t)id];
    b->h_fwd_lp[(size_t)id] = lp[(size_t)id];
  }
  *lp_out = lp_total;
  *dead_out = 0;
  *done = true;
  return TEHMM_OK;
}

// Raw statistics of every interval of the batch ADDED into dev_stats (device buffer of
// stats_size(NP, R) doubles, layout in tehmm_aux.hip.h); *dead_any: some interval met an impossible row
// after its first emittable one (the reference's lattices are NaN there).
static int estep_accumulate(tehmm_model_t *m, tehmm_batch_t *b, int use_ratios, double *dev_stats, double *lp_out,
                            int *dead_out) {
#ifdef TEHMM_DEV_NT
  if (m->N < 64 && m->NP != TEHMM_DEV_NT)
    return fail(TEHMM_ERR_UNSUPPORTED, "development build: fused kernels exist for one padded state count only");
#endif
  *lp_out = 0.0;
  *dead_out = 0;
  b->h_fwd_lp.assign((size_t)b->n, 0.0);
  if (b->n == 0 || b->total == 0) return TEHMM_OK;
  const bool ratio = use_ratios && b->has_ratios;
  const int N = m->N, NP = m->NP;
  {
    bool done = false;
    const char *wm = std::getenv("TEHMM_ESTEP_WIDE");
    const bool wide_first = wm && std::atoi(wm) == 2;
    if (!ratio && N < 64 && !wide_first) {
      int rcf = estep_fused(m, b, dev_stats, lp_out, dead_out, &done);
      if (rcf) return rcf;
      if (done) return TEHMM_OK;
    }
    int rcw = estep_wide(m, b, ratio, dev_stats, lp_out, dead_out, &done);
    if (rcw) return rcw;
    if (done) return TEHMM_OK;
    if (N >= 64)
      return fail(TEHMM_ERR_UNSUPPORTED, "E-step at N >= 64: the item-parallel passes do not apply to this batch (impossible "
                                         "emission rows, links that do not verify, or a batch below 1024 rows): use the array-level path");
  }
  // Intervals are processed in groups whose alpha / beta / w rows fit a fixed workspace
  // (3 x 8N + 4 bytes per position, 40 % of the free HBM): the 3 Gb training sets of config 4
  // never materialise whole-genome lattices.
  size_t free_b = 0, total_b = 0;
  (void)dev_mem_info(&free_b, &total_b);
  int64_t budget_bytes = (int64_t)((double)(free_b + (size_t)b->ew.rows_cap * (24 * N + 4)) * 0.4);
  if (budget_bytes < (4ll << 30)) budget_bytes = 4ll << 30;
  const int64_t budget_rows = std::max<int64_t>(budget_bytes / (24 * N + 4), 1);
  EstepWork &w = b->ew;
  if (w.N != N) {
    w.alpha.release();
    w.rows_cap = 0;
    w.N = N;
  }
  w.start = dev_stats + stats_off_start();
  w.C = dev_stats + stats_off_C(NP);
  w.D = dev_stats + stats_off_D(NP);
  w.stat = dev_stats + stats_off_stat(NP);
  IntervalTab iv;
  EmisTab em;
  fill_tabs(m, b, iv, em, ratio);     // fit applies the ratios to emissions too (basehmm.py:510)
  double lp_total = 0.0;
  int dead_any = 0;
  size_t pos_in_order = 0;
  while (pos_in_order < (size_t)b->n) {
    // next group: consecutive slots of the longest-first order
    std::vector<int> g_order;
    std::vector<int64_t> grow0((size_t)b->n, 0);
    std::vector<int> chunk_iv;
    std::vector<int64_t> chunk_t0;
    int64_t rows = 0;
    while (pos_in_order < (size_t)b->n) {
      const int id = b->h_order[pos_in_order];
      const int64_t T = b->h_len[id];
      if (!g_order.empty() && rows + T > budget_rows) break;
      g_order.push_back(id);
      grow0[id] = rows;
      for (int64_t t0 = 0; t0 < T; t0 += kEstepChunk) {
        chunk_iv.push_back(id);
        chunk_t0.push_back(t0);
      }
      rows += T;
      ++pos_in_order;
    }
    if (rows == 0) continue;
    if (rows > w.rows_cap) {
      HIPCHK(w.alpha.alloc((size_t)rows * N + 1));
      HIPCHK(w.beta.alloc((size_t)rows * N + 1));
      HIPCHK(w.wrows.alloc((size_t)rows * N + 1));
      HIPCHK(w.escale.alloc((size_t)rows + 1));
      w.rows_cap = rows;
    }
    if (b->n > w.n_cap) {
      HIPCHK(w.fwd_lp.alloc((size_t)b->n + 1));
      HIPCHK(w.dead.alloc((size_t)b->n + 1));
      w.n_cap = b->n;
    }
    HIPCHK(hipMemset(w.dead.p, 0, (size_t)(b->n + 1) * sizeof(int)));
    HIPCHK(hipMemset(w.escale.p, 0, ((size_t)rows + 1) * sizeof(int)));
    HIPCHK(w.order.upload(g_order.data(), g_order.size()));
    HIPCHK(w.grow0.upload(grow0.data(), grow0.size()));
    HIPCHK(w.chunk_iv.upload(chunk_iv.data(), chunk_iv.size()));
    HIPCHK(w.chunk_t0.upload(chunk_t0.data(), chunk_t0.size()));
    IntervalTab giv = iv;
    giv.order = w.order.p;
    giv.out0 = w.grow0.p;
    const int n_iv = (int)g_order.size(), n_chunks = (int)chunk_iv.size();
#define CALL(NT_) launch_estep<NT_>(m, giv, em, ratio, n_iv, n_chunks, w, b->ratios.p, b->sP)
    TEHMM_NT_DISPATCH(m->NP, CALL)
#undef CALL
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(b->sP));
    std::vector<double> lp((size_t)b->n);
    std::vector<int> dead((size_t)b->n);
    HIPCHK(hipMemcpy(lp.data(), w.fwd_lp.p, (size_t)b->n * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(dead.data(), w.dead.p, (size_t)b->n * sizeof(int), hipMemcpyDeviceToHost));
    for (int id : g_order) {
      lp_total += lp[(size_t)id];
      dead_any |= dead[(size_t)id];
      b->h_fwd_lp[(size_t)id] = dead[(size_t)id] ? std::nan("") : lp[(size_t)id];
    }
  }
  *lp_out = lp_total;
  *dead_out = dead_any;
  return TEHMM_OK;
}

int64_t tehmm_model_stats_size(const tehmm_model_t *m) { return m ? stats_size(m->NP, m->R) : 0; }

int tehmm_estep_batch_device(tehmm_model_t *m, tehmm_batch_t *b, int use_ratios, double *dev_stats,
                             double *logprob_sum) {
  if (!m || !b || !dev_stats || !logprob_sum)
    return fail(TEHMM_ERR_ARG, "tehmm_estep_batch_device: NULL argument");
  if (m->K != b->K) return fail(TEHMM_ERR_ARG, "tehmm_estep_batch_device: model/batch track count differ");
  if (m->N > 128) return fail(TEHMM_ERR_UNSUPPORTED, "tehmm_estep_batch_device: N > 128");
  double lp = 0.0;
  int dead = 0;
  int rc = estep_accumulate(m, b, use_ratios, dev_stats, &lp, &dead);
  if (rc) return rc;
  const double nan = std::nan("");
  // slots 0 / 1 of the buffer: log-likelihood sum (NaN poisons it, as the reference's NaN lattices would)
  // and sequence count -- they travel with the statistics through the all-reduce
  double head[2];
  HIPCHK(hipMemcpy(head, dev_stats, sizeof(head), hipMemcpyDeviceToHost));
  head[0] += dead ? nan : lp;
  head[1] += (double)b->n;
  HIPCHK(hipMemcpy(dev_stats, head, sizeof(head), hipMemcpyHostToDevice));
  if (dead) {      // the reference's statistics are NaN from such a sequence on: poison the buffer
    std::vector<double> bad((size_t)stats_size(m->NP, m->R), nan);
    bad[1] = head[1];
    HIPCHK(hipMemcpy(dev_stats, bad.data(), bad.size() * sizeof(double), hipMemcpyHostToDevice));
  }
  *logprob_sum = dead ? nan : lp;
  return TEHMM_OK;
}

int tehmm_estep_batch(tehmm_model_t *m, tehmm_batch_t *b, int use_ratios, double *start,
                      double *trans, double *obsStats, double *logprob_sum) {
  if (!m || !b || !start || !trans || !obsStats || !logprob_sum)
    return fail(TEHMM_ERR_ARG, "tehmm_estep_batch: NULL argument");
  if (m->K != b->K) return fail(TEHMM_ERR_ARG, "tehmm_estep_batch: model/batch track count differ");
  if (m->N > 128) return fail(TEHMM_ERR_UNSUPPORTED, "tehmm_estep_batch: N > 128");
  *logprob_sum = 0.0;
  if (b->n == 0 || b->total == 0) return TEHMM_OK;
  const int N = m->N, NP = m->NP, K = m->K, S = m->S;
  EstepWork &w = b->ew;
  const size_t ssz = (size_t)stats_size(NP, m->R);
  HIPCHK(w.statbuf.ensure(ssz));
  HIPCHK(hipMemset(w.statbuf.p, 0, ssz * sizeof(double)));
  double lp_total = 0.0;
  int dead_any = 0;
  int rc = estep_accumulate(m, b, use_ratios, w.statbuf.p, &lp_total, &dead_any);
  if (rc) return rc;
  std::vector<double> hs(ssz);
  HIPCHK(hipMemcpy(hs.data(), w.statbuf.p, ssz * sizeof(double), hipMemcpyDeviceToHost));
  const double *hS = hs.data() + stats_off_start(), *hC = hs.data() + stats_off_C(NP),
               *hD = hs.data() + stats_off_D(NP), *hst = hs.data() + stats_off_stat(NP);
  const double nan = std::nan("");
  const double invN = 1.0 / (double)N;
  for (int i = 0; i < N; ++i) {
    start[i] += dead_any ? nan : hS[i];
    for (int j = 0; j < N; ++j) {
      double v = std::exp(m->h_lt[(size_t)i * N + j]) * hC[(size_t)i * NP + j];
      if (i == j) v += hD[i];
      trans[(size_t)i * N + j] += dead_any ? nan : v * invN;
    }
  }
  for (int k = 0; k < K; ++k)
    for (int s2 = 0; s2 < m->rowcnt[k]; ++s2)
      for (int j = 0; j < N; ++j)
        obsStats[((size_t)k * N + j) * S + s2] += dead_any ? nan : hst[(size_t)(m->rowbase[k] + s2) * NP + j];
  *logprob_sum = dead_any ? nan : lp_total;
  return TEHMM_OK;
}

#include "tehmm_aux_host.inc"

// Diagnostic: cycle stamps of the last cooperative kernel (only in the -DTEHMM_STAMPS build).
int tehmm_debug_read_stamps(unsigned long long *out, int n) {
#ifdef TEHMM_STAMPS
  if (!out || n <= 0 || n > 2 * 4096 * 16) return fail(TEHMM_ERR_ARG, "tehmm_debug_read_stamps: bad argument");
  HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), (size_t)n * sizeof(unsigned long long)));
  return TEHMM_OK;
#else
  (void)out; (void)n;
  return fail(TEHMM_ERR_UNSUPPORTED, "tehmm_debug_read_stamps: library built without -DTEHMM_STAMPS");
#endif
}
