// glrm_multigpu.hip -- glrm_hip_multi_*: fit!(glrm, ProxGradParams) on N devices driven by ONE host process.
//
// The reference parallelises the half-steps over threads of one address space ("rows independent, then columns independent",
// src/algorithms/proxgrad_multithread.jl:118,163).  Here shard s = a contiguous block of rows + a contiguous block of columns on
// device_ids[s]; X and Y are replicated on every device; after the X half-step each device pushes the row block it updated to
// every peer, after the Y half-step its column block.  This is the entry point a single Julia process can
// ccall; the one-process-per-GPU host (lowrankmodels.jl_amd/fit.py::ShardedFit over torch.distributed / RCCL) drives the same
// step-level calls.
//
// Exchange algorithms
//   direct (default)  hipMemcpyPeerAsync of the 1/N slice from its owner to every peer, one copy stream per (source, destination)
//                     pair: xGMI is point-to-point (7 links per GPU), so all links of a GPU carry one slice at once and the
//                     exchange takes slice_bytes / link_bandwidth instead of the (N-1) hops of a ring.
//   rccl              ncclAllGather in place (equal blocks) or one ncclBroadcast per owner (ragged blocks) inside one group;
//                     librccl.so is loaded at run time, needs distinct devices.
// Each shard's step_* call is issued from its own host thread: the LDS-tiled column sweep reads back an active count per
// line-search round, and those round trips must not serialise across devices.
//
// The exchange is OFF the critical path between the half-steps (round 5).  The X half-step runs in x_chunks row chunks; the push of chunk
// j leaves while chunk j+1 is swept, and the Y half-step does not wait for the exchange to finish: glrm_hip_step_y_arrival gets the
// list (own rows -- no event --, then chunk 0 of every peer, chunk 1 of every peer, ...) with the copy event of each range, and the
// phase-aligned column passes launch each super-tile of X behind the events of the ranges it touches.  Partial sums are per (column,
// super-tile) and are added in super-tile order, so no bit depends on the arrival order (tests: test_multi_in_process.py,
// test_gpu_families.py).  The Y blocks (k x n / N: 0.08 ms per link at C4) are exchanged the old way.
//
// Link emulation (TEST BUILD ONLY: compiled with -DGLRM_HIP_TESTING into libglrm_hip_testing.so, never into the product library --
// lowrankmodels.jl_amd/build.py; one box, fewer GPUs than shards): GLRM_EXCHANGE_EMULATE_GBPS=<rate> makes every direct push occupy its (source,
// destination) link for bytes / rate counted from the moment the source range was complete -- a device-to-device copy on one GPU takes
// no such time.  Mechanism 1 (default where hipDeviceAttributeCanUseStreamWaitValue is set): the copy stream writes a start mark
// (hipStreamWriteValue64), copies, then waits (hipStreamWaitValue64) for a release value that a host timer thread posts bytes / rate
// after it saw the mark -- no CU is occupied by the delay.  Mechanism 2 (GLRM_EXCHANGE_EMULATE_MODE=2): a one-lane kernel that
// sleeps on wall_clock64 until the deadline (needs a free wave slot: persistent kernels that fill the register file delay it).
// GLRM_EXCHANGE_EMULATE_DILATE=<d> divides the rate; default = the largest number of shards on one device (d shards time-share the
// device, so compute is d times slower than on d devices and a transfer has to be as well to keep the proportion).
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <deque>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "glrm_engine.hpp"

namespace {

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// ---- RCCL, resolved at run time ---------------------------------------------------------------------------------------
struct Rccl {
  typedef void* comm_t;
  int (*CommInitAll)(comm_t*, int, const int*) = nullptr;
  int (*CommDestroy)(comm_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, comm_t, hipStream_t) = nullptr;
  int (*Broadcast)(const void*, void*, size_t, int, int, comm_t, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool ok = false;
  Rccl() {
#ifdef GLRM_HIP_TESTING // the test build may load a stand-in (tests/stubs/rccl_stub.cpp); the product library only ever opens librccl
    const char* path = getenv("GLRM_HIP_RCCL_LIB");
    void* lib = (path && *path) ? dlopen(path, RTLD_LAZY | RTLD_LOCAL) : nullptr;
#else
    void* lib = nullptr;
#endif
    if (!lib) lib = dlopen("librccl.so", RTLD_LAZY | RTLD_LOCAL);
    if (!lib) lib = dlopen("librccl.so.1", RTLD_LAZY | RTLD_LOCAL);
    if (!lib) return;
    CommInitAll = (decltype(CommInitAll))dlsym(lib, "ncclCommInitAll");
    CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
    AllGather = (decltype(AllGather))dlsym(lib, "ncclAllGather");
    Broadcast = (decltype(Broadcast))dlsym(lib, "ncclBroadcast");
    GroupStart = (decltype(GroupStart))dlsym(lib, "ncclGroupStart");
    GroupEnd = (decltype(GroupEnd))dlsym(lib, "ncclGroupEnd");
    GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
    ok = CommInitAll && CommDestroy && AllGather && Broadcast && GroupStart && GroupEnd;
  }
  static Rccl& get() { static Rccl r; return r; }
};
constexpr int NCCL_FLOAT64 = 8; // ncclFloat64 (rccl.h)

} // namespace

// One persistent host thread per shard beyond the first (shard 0 runs on the caller's thread): a half-step is ~8 calls per iteration
// and shard, so the threads are kept instead of being spawned per call.
struct ShardPool {
  std::vector<std::thread> th;
  std::mutex mu;
  std::condition_variable cv_go, cv_done;
  const std::function<void(int)>* job = nullptr;
  uint64_t epoch = 0;
  int pending = 0;
  bool quit = false;
  void start(int n) {
    for (int s = 1; s < n; ++s)
      th.emplace_back([this, s] {
        uint64_t seen = 0;
        for (;;) {
          const std::function<void(int)>* j;
          {
            std::unique_lock<std::mutex> lk(mu);
            cv_go.wait(lk, [&] { return quit || epoch != seen; });
            if (quit) return;
            seen = epoch;
            j = job;
          }
          (*j)(s);
          {
            std::lock_guard<std::mutex> lk(mu);
            if (--pending == 0) cv_done.notify_one();
          }
        }
      });
  }
  void run(const std::function<void(int)>& f) { // f(s) for s = 0..n-1, returns when all are done
    if (!th.empty()) {
      std::lock_guard<std::mutex> lk(mu);
      job = &f;
      pending = (int)th.size();
      ++epoch;
    }
    cv_go.notify_all();
    f(0);
    if (!th.empty()) {
      std::unique_lock<std::mutex> lk(mu);
      cv_done.wait(lk, [&] { return pending == 0; });
    }
  }
  ~ShardPool() {
    {
      std::lock_guard<std::mutex> lk(mu);
      quit = true;
    }
    cv_go.notify_all();
    for (auto& t : th) t.join();
  }
};

#ifdef GLRM_HIP_TESTING
// ---- link emulator ----------------------------------------------------------------------------------------------------
__global__ void link_mark_kernel(unsigned long long* t0) { *t0 = (unsigned long long)wall_clock64(); }
__global__ void link_delay_kernel(const unsigned long long* t0, unsigned long long ticks) {
  const unsigned long long start = *t0;
  while ((unsigned long long)wall_clock64() - start < ticks) __builtin_amdgcn_s_sleep(64);
}

struct LinkEmu {
  double bytes_per_s = 0.0; // per direction and link, already divided by the dilation; 0 = off
  double gbps = 0.0;
  int dilate = 1, mode = 0; // 1 stream wait-value + host timer, 2 delay kernels
  // mode 1: one (mark, release) pair of 8-byte signal words per (source, destination) link; tickets count the transfers of a link
  std::vector<uint64_t*> mark, release;
  std::vector<uint64_t> ticket;
  struct Job { int link; uint64_t ticket; double seconds; double due; bool started; };
  std::deque<Job> jobs; // per link in ticket order (one copy stream per link keeps them ordered on the device too)
  std::mutex mu;
  std::thread timer;
  std::atomic<bool> quit{false};
  // mode 2
  std::vector<unsigned long long*> t0;
  double wall_hz = 1e8;
  void timer_loop() {
    // a transfer may not be released before bytes / rate have passed since its mark; a job that never sees its mark (its stream was
    // torn down) is dropped by destroy, which posts every release value first
    for (;;) {
      if (quit.load(std::memory_order_acquire)) return;
      bool busy = false;
      {
        std::lock_guard<std::mutex> lk(mu);
        const double now = now_s();
        for (auto it = jobs.begin(); it != jobs.end();) {
          busy = true;
          if (!it->started && __atomic_load_n(mark[it->link], __ATOMIC_ACQUIRE) >= it->ticket) { it->started = true; it->due = now + it->seconds; }
          if (it->started && now >= it->due) {
            __atomic_store_n(release[it->link], it->ticket, __ATOMIC_RELEASE);
            it = jobs.erase(it);
          } else ++it;
        }
      }
      if (busy) std::this_thread::sleep_for(std::chrono::microseconds(20));
      else std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
  }
  void release_all() { // nothing may be left waiting on the device
    std::lock_guard<std::mutex> lk(mu);
    for (auto& j : jobs) __atomic_store_n(release[j.link], j.ticket, __ATOMIC_RELEASE);
    jobs.clear();
  }
};

#endif // GLRM_HIP_TESTING

struct glrm_multi {
  int n = 0;
  ShardPool pool;
  int64_t m = 0, nn = 0, d = 0;
  int k = 0, ld = 0;
  std::vector<int> dev;
  std::vector<glrm_handle*> sh;
  std::vector<hipStream_t> st;                       // compute stream of shard s (device dev[s])
  std::vector<double*> dX, dY, dObjCol, dObjRow;     // replica of shard s
  std::vector<std::vector<hipStream_t>> cs;          // cs[s][t]: copy stream on dev[s] for pushes s -> t
  std::vector<hipEvent_t> ev_done;                   // sweep of shard s finished (timing enabled: start of its exchange wait)
  std::vector<hipEvent_t> ev_ready;                  // every incoming block of shard s has arrived
  std::vector<std::vector<hipEvent_t>> ev_arr;       // ev_arr[s][t]: block of s landed on t
  std::vector<std::vector<std::vector<hipEvent_t>>> ev_chunk; // ev_chunk[s][t][j]: row chunk j of s's X block landed on t
  int arrival = 1;                                   // Y half-step consumes the X chunks in arrival order (glrm_hip_step_y_arrival)
#ifdef GLRM_HIP_TESTING
  LinkEmu emu;
#endif
  std::vector<double> wait_y0;                       // ms_wait_y of every shard at the start of the fit
  std::vector<int64_t> rbs, cbs, ybs;                // shard bounds: rows, columns, vectors of Y
  int exchange = 0;                                  // in use: 0 direct, 1 rccl
  std::vector<Rccl::comm_t> comms;
  int x_chunks = 1;
  bool dense = false;
  bool profile = false;                              // glrm_options.profile: account the exposed exchange time of every exchange
  double exchange_ms = 0.0;
  int64_t n_rx = 0, n_ry = 0;
};

namespace {

// Contiguous blocks balanced by observation count; equal-count blocks when they are within 2 % of that balance
// (lowrankmodels.jl_amd/fit.py::partition -- the two hosts shard alike).
std::vector<int64_t> partition(const int64_t* ptr, int64_t nseg, int parts) {
  std::vector<int64_t> eq((size_t)parts + 1);
  for (int i = 0; i <= parts; ++i) eq[i] = nseg * i / parts;
  if (!ptr) return eq;
  const int64_t nnz = ptr[nseg];
  if (nnz <= 0) return eq;
  if (nseg % parts == 0) {
    int64_t worst = 0;
    for (int i = 0; i < parts; ++i) worst = std::max(worst, ptr[eq[i + 1]] - ptr[eq[i]]);
    if ((double)worst <= 1.02 * (double)nnz / parts + 1) return eq;
  }
  std::vector<int64_t> b((size_t)parts + 1);
  for (int i = 0; i <= parts; ++i) {
    const double target = (double)nnz * i / parts;
    b[i] = std::lower_bound(ptr, ptr + nseg + 1, target, [](int64_t v, double t) { return (double)v < t; }) - ptr;
  }
  b[0] = 0;
  b[parts] = nseg;
  for (int i = 1; i <= parts; ++i) b[i] = std::max(b[i], b[i - 1]);
  return b;
}

// fn(s) for every shard, each on its own host thread (shard 0 on the caller's); first failure wins.
int run_all(glrm_multi* mh, const std::function<int(int)>& fn) {
  const int n = mh->n;
  std::vector<int> rc((size_t)n, 0);
  std::vector<std::string> msg((size_t)n);
  const std::function<void(int)> body = [&](int s) {
    rc[s] = fn(s);
    if (rc[s]) msg[s] = g_err; // thread-local message of the failing call
  };
  mh->pool.run(body);
  for (int s = 0; s < n; ++s)
    if (rc[s]) return fail(rc[s], "shard %d (device %d): %s", s, mh->dev[s], msg[s].c_str());
  return GLRM_OK;
}

#define COMMCK(expr)                                                                                            \
  do {                                                                                                          \
    hipError_t e_ = (expr);                                                                                     \
    if (e_ != hipSuccess) return fail(GLRM_ERR_COMM, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

#ifdef GLRM_HIP_TESTING
// Link emulation around the pushes of source `link` on its link stream c: before / after hooks
int emu_before(glrm_multi* mh, hipStream_t c, int link) {
  LinkEmu& e = mh->emu;
  if (e.bytes_per_s <= 0.0) return GLRM_OK;
  if (e.mode == 1) COMMCK(hipStreamWriteValue64(c, e.mark[link], e.ticket[link] + 1, 0));
  else hipLaunchKernelGGL(link_mark_kernel, dim3(1), dim3(1), 0, c, e.t0[link]);
  return GLRM_OK;
}
int emu_after(glrm_multi* mh, hipStream_t c, int link, size_t bytes) {
  LinkEmu& e = mh->emu;
  if (e.bytes_per_s <= 0.0) return GLRM_OK;
  const double seconds = (double)bytes / e.bytes_per_s;
  if (e.mode == 1) {
    const uint64_t tk = ++e.ticket[link];
    {
      std::lock_guard<std::mutex> lk(e.mu);
      e.jobs.push_back(LinkEmu::Job{link, tk, seconds, 0.0, false});
    }
    COMMCK(hipStreamWaitValue64(c, e.release[link], tk, hipStreamWaitValueGte));
  } else {
    hipLaunchKernelGGL(link_delay_kernel, dim3(1), dim3(1), 0, c, e.t0[link], (unsigned long long)(seconds * e.wall_hz));
    COMMCK(hipGetLastError());
  }
  return GLRM_OK;
}

#endif // GLRM_HIP_TESTING

// Shard s owns buf[bounds[s]*unit .. bounds[s+1]*unit) (doubles) restricted to the sub-range [lo_s, hi_s) of its block given by
// (cj, cn) (fractions j/C of the block for the pipelined X exchange; 0/1 = the whole block).  Phase 1 (push): after the work already
// queued on st[s], push that range to every destination and record ev_arr[s][t] (and, chunk_events: ev_chunk[s][t][cj]) behind it.
// Phase 2 (wait): st[t] waits for everything pushed to t so far.
int exchange_push(glrm_multi* mh, std::vector<double*>& buf, const std::vector<int64_t>& bounds, int64_t unit, int only_dst, int cj,
                  int cn, bool chunk_events = false) {
  const int n = mh->n;
  if (n == 1) return GLRM_OK;
  for (int s = 0; s < n; ++s) {
    COMMCK(hipSetDevice(mh->dev[s]));
    COMMCK(hipEventRecord(mh->ev_done[s], mh->st[s]));
  }
  if (mh->exchange == 1) { // RCCL: one grouped collective over all communicators (always to every rank; x_chunks is off here)
    Rccl& R = Rccl::get();
    bool equal = true;
    for (int s = 1; s < n; ++s) equal = equal && (bounds[s + 1] - bounds[s] == bounds[1] - bounds[0]);
    int rc = R.GroupStart();
    for (int s = 0; s < n && !rc; ++s) {
      if (equal) {
        const size_t cnt = (size_t)((bounds[1] - bounds[0]) * unit);
        if (cnt) rc = R.AllGather(buf[s] + bounds[s] * unit, buf[s], cnt, NCCL_FLOAT64, mh->comms[s], mh->st[s]);
      } else {
        for (int r = 0; r < n && !rc; ++r) {
          const size_t cnt = (size_t)((bounds[r + 1] - bounds[r]) * unit);
          if (cnt) rc = R.Broadcast(buf[s] + bounds[r] * unit, buf[s] + bounds[r] * unit, cnt, NCCL_FLOAT64, r, mh->comms[s], mh->st[s]);
        }
      }
    }
    const int rc2 = R.GroupEnd();
    if (rc || rc2) return fail(GLRM_ERR_COMM, "RCCL exchange failed: %s", R.GetErrorString ? R.GetErrorString(rc ? rc : rc2) : "?");
    return GLRM_OK;
  }
#ifdef GLRM_HIP_TESTING
  const bool emulate = mh->emu.bytes_per_s > 0.0;
#endif
  for (int s = 0; s < n; ++s) {
    const int64_t blk = bounds[s + 1] - bounds[s];
    const int64_t lo = bounds[s] + blk * cj / cn, hi = bounds[s] + blk * (cj + 1) / cn;
    const size_t bytes = (size_t)((hi - lo) * unit) * 8;
    COMMCK(hipSetDevice(mh->dev[s]));
#ifdef GLRM_HIP_TESTING
    if (emulate) {
      // ONE stream per source stands for its n - 1 links (they run in parallel and carry equal sizes, so they finish together):
      // mark, the copies, then the wait for bytes / rate since the mark, then the arrival events.  Few streams on purpose: a queue
      // that stands in a wait blocks every stream the runtime maps onto the same hardware queue (GPU_MAX_HW_QUEUES).
      hipStream_t c = mh->cs[s][(s + 1) % n];
      COMMCK(hipStreamWaitEvent(c, mh->ev_done[s], 0));
      int rc = bytes ? emu_before(mh, c, s) : GLRM_OK;
      if (rc) return rc;
      for (int d = 1; d < n && bytes; ++d) {
        const int t = (s + d) % n;
        if (only_dst >= 0 && t != only_dst) continue;
        if (mh->dev[s] == mh->dev[t]) COMMCK(hipMemcpyAsync(buf[t] + lo * unit, buf[s] + lo * unit, bytes, hipMemcpyDeviceToDevice, c));
        else COMMCK(hipMemcpyPeerAsync(buf[t] + lo * unit, mh->dev[t], buf[s] + lo * unit, mh->dev[s], bytes, c));
      }
      if (bytes && (rc = emu_after(mh, c, s, bytes))) return rc;
      for (int d = 1; d < n; ++d) {
        const int t = (s + d) % n;
        if (only_dst >= 0 && t != only_dst) continue;
        COMMCK(hipEventRecord(mh->ev_arr[s][t], c));
        if (chunk_events) COMMCK(hipEventRecord(mh->ev_chunk[s][t][cj], c));
      }
      continue;
    }
#endif
    for (int d = 1; d < n; ++d) { // destinations in ring order from s: no destination is everybody's first
      const int t = (s + d) % n;
      if (only_dst >= 0 && t != only_dst) continue;
      hipStream_t c = mh->cs[s][t];
      COMMCK(hipStreamWaitEvent(c, mh->ev_done[s], 0));
      if (bytes) {
        if (mh->dev[s] == mh->dev[t]) COMMCK(hipMemcpyAsync(buf[t] + lo * unit, buf[s] + lo * unit, bytes, hipMemcpyDeviceToDevice, c));
        else COMMCK(hipMemcpyPeerAsync(buf[t] + lo * unit, mh->dev[t], buf[s] + lo * unit, mh->dev[s], bytes, c));
      }
      COMMCK(hipEventRecord(mh->ev_arr[s][t], c));
      if (chunk_events) COMMCK(hipEventRecord(mh->ev_chunk[s][t][cj], c));
    }
  }
  return GLRM_OK;
}

int exchange_wait(glrm_multi* mh, int only_dst) {
  const int n = mh->n;
  if (n == 1 || mh->exchange == 1) return GLRM_OK; // RCCL collectives are ordered on the compute streams themselves
  for (int t = 0; t < n; ++t) {
    if (only_dst >= 0 && t != only_dst) continue;
    COMMCK(hipSetDevice(mh->dev[t]));
    for (int s = 0; s < n; ++s)
      if (s != t) COMMCK(hipStreamWaitEvent(mh->st[t], mh->ev_arr[s][t], 0));
    COMMCK(hipEventRecord(mh->ev_ready[t], mh->st[t]));
  }
  return GLRM_OK;
}

int exchange_all(glrm_multi* mh, std::vector<double*>& buf, const std::vector<int64_t>& bounds, int64_t unit, int only_dst = -1) {
  int rc = exchange_push(mh, buf, bounds, unit, only_dst, 0, 1);
  if (rc) return rc;
  return exchange_wait(mh, only_dst);
}

// Exposed exchange time: what the compute stream of a shard spent between the end of its own (last) sweep and the arrival of the
// last incoming block, max over shards.  Needs every stream drained, so it is only accounted with glrm_options.profile.
void account_exchange(glrm_multi* mh) {
  if (mh->n == 1 || mh->exchange == 1 || !mh->profile) return;
  for (int t = 0; t < mh->n; ++t)
    if (hipSetDevice(mh->dev[t]) != hipSuccess || hipStreamSynchronize(mh->st[t]) != hipSuccess) { (void)hipGetLastError(); return; }
  float worst = 0.f;
  for (int t = 0; t < mh->n; ++t) {
    float ms = 0.f;
    if (hipSetDevice(mh->dev[t]) == hipSuccess && hipEventElapsedTime(&ms, mh->ev_done[t], mh->ev_ready[t]) == hipSuccess) worst = std::max(worst, ms);
    else (void)hipGetLastError();
  }
  mh->exchange_ms += worst;
}

int sum_on_shard0(glrm_multi* mh, double* vec, int64_t cnt, double* out) { return glrm_hip_sum(mh->sh[0], vec, cnt, out); }

// loss + rx + ry at the replicated factors (objective(...), src/evaluate_fit.jl:4-23,91-104), fixed-order sums on shard 0
int multi_objective(glrm_multi* mh, double* out) {
  int rc;
  double loss = 0, px = 0, py = 0;
  if ((rc = run_all(mh, [&](int s) { return glrm_hip_col_losses(mh->sh[s]); }))) return rc;
  if ((rc = exchange_all(mh, mh->dObjCol, mh->cbs, 1, 0))) return rc;
  if ((rc = sum_on_shard0(mh, mh->dObjCol[0], mh->nn, &loss))) return rc;
  if ((rc = run_all(mh, [&](int s) { return glrm_hip_row_penalties(mh->sh[s]); }))) return rc;
  if ((rc = exchange_all(mh, mh->dObjRow, mh->rbs, 1, 0))) return rc;
  if ((rc = sum_on_shard0(mh, mh->dObjRow[0], mh->m, &px))) return rc;
  if ((rc = run_all(mh, [&](int s) { return glrm_hip_col_penalties(mh->sh[s]); }))) return rc;
  if ((rc = exchange_all(mh, mh->dObjCol, mh->cbs, 1, 0))) return rc;
  if ((rc = sum_on_shard0(mh, mh->dObjCol[0], mh->nn, &py))) return rc;
  *out = loss + (px + py);
  return GLRM_OK;
}

} // namespace

extern "C" void glrm_hip_multi_destroy(glrm_multi* mh) {
  if (!mh) return;
  int prev = 0;
  (void)hipGetDevice(&prev);
#ifdef GLRM_HIP_TESTING
  if (mh->emu.timer.joinable()) { // nothing may be left waiting for a release on the device
    mh->emu.release_all();
    mh->emu.quit.store(true, std::memory_order_release);
    mh->emu.timer.join();
  }
#endif
  for (int s = 0; s < (int)mh->sh.size(); ++s) {
    if (mh->sh[s]) glrm_hip_destroy(mh->sh[s]);
  }
  for (int s = 0; s < mh->n && s < (int)mh->dev.size(); ++s) {
    if (hipSetDevice(mh->dev[s]) != hipSuccess) continue;
    if (s < (int)mh->comms.size() && mh->comms[s]) (void)Rccl::get().CommDestroy(mh->comms[s]);
    for (double* p : {s < (int)mh->dX.size() ? mh->dX[s] : nullptr, s < (int)mh->dY.size() ? mh->dY[s] : nullptr,
                      s < (int)mh->dObjCol.size() ? mh->dObjCol[s] : nullptr, s < (int)mh->dObjRow.size() ? mh->dObjRow[s] : nullptr})
      if (p) (void)hipFree(p);
    if (s < (int)mh->cs.size())
      for (hipStream_t c : mh->cs[s])
        if (c) (void)hipStreamDestroy(c);
    if (s < (int)mh->ev_arr.size())
      for (hipEvent_t e : mh->ev_arr[s])
        if (e) (void)hipEventDestroy(e);
    if (s < (int)mh->ev_chunk.size())
      for (auto& v : mh->ev_chunk[s])
        for (hipEvent_t e : v)
          if (e) (void)hipEventDestroy(e);
#ifdef GLRM_HIP_TESTING
    if (s < (int)mh->emu.mark.size() && mh->emu.mark[s]) (void)hipFree(mh->emu.mark[s]);
    if (s < (int)mh->emu.release.size() && mh->emu.release[s]) (void)hipFree(mh->emu.release[s]);
    if (s < (int)mh->emu.t0.size() && mh->emu.t0[s]) (void)hipFree(mh->emu.t0[s]);
#endif
    if (s < (int)mh->ev_done.size() && mh->ev_done[s]) (void)hipEventDestroy(mh->ev_done[s]);
    if (s < (int)mh->ev_ready.size() && mh->ev_ready[s]) (void)hipEventDestroy(mh->ev_ready[s]);
    if (s < (int)mh->st.size() && mh->st[s]) (void)hipStreamDestroy(mh->st[s]);
  }
  (void)hipSetDevice(prev);
  delete mh;
}

static int multi_create_impl(glrm_multi* mh, const glrm_problem* p, const glrm_options* o, const glrm_multi_options* mo) {
  const int n = mh->n;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(GLRM_ERR_HIP, "no HIP device is visible (this engine has no CPU fallback)");
  mh->dev.resize(n);
  bool distinct = true;
  for (int s = 0; s < n; ++s) {
    mh->dev[s] = mo->device_ids ? mo->device_ids[s] : s;
    if (mh->dev[s] < 0 || mh->dev[s] >= ndev) return fail(GLRM_ERR_INVALID, "device_ids[%d] = %d out of range (%d devices)", s, mh->dev[s], ndev);
    for (int t = 0; t < s; ++t) distinct = distinct && mh->dev[t] != mh->dev[s];
  }
  mh->m = p->m; mh->nn = p->n; mh->k = p->k;
  mh->dense = p->dense_A != nullptr;
  mh->profile = o && o->profile;
  mh->rbs = partition(mh->dense ? nullptr : p->rowptr, p->m, n);
  mh->cbs = partition(mh->dense ? nullptr : p->colptr, p->n, n);
  // vectors of Y owned by each column (get_yidxs, src/losses.jl:76-93)
  std::vector<int64_t> ystart((size_t)p->n + 1, 0);
  for (int64_t f = 0; f < p->n; ++f) {
    const glrm_loss& l = p->n_losses == 1 ? p->losses[0] : p->losses[f];
    ystart[f + 1] = ystart[f] + (l.dim > 1 ? l.dim : 1);
  }
  mh->d = ystart[p->n];
  mh->ybs.resize((size_t)n + 1);
  for (int s = 0; s <= n; ++s) mh->ybs[s] = ystart[mh->cbs[s]];
  mh->n_rx = p->n_rx; mh->n_ry = p->n_ry;
  mh->sh.assign(n, nullptr); mh->st.assign(n, nullptr);
  mh->dX.assign(n, nullptr); mh->dY.assign(n, nullptr); mh->dObjCol.assign(n, nullptr); mh->dObjRow.assign(n, nullptr);
  mh->ev_done.assign(n, nullptr); mh->ev_ready.assign(n, nullptr);
  mh->cs.assign(n, std::vector<hipStream_t>(n, nullptr));
  mh->ev_arr.assign(n, std::vector<hipEvent_t>(n, nullptr));
  for (int s = 0; s < n; ++s) { // peer access (a no-op for shards that share a device)
    HIPCK(hipSetDevice(mh->dev[s]));
    for (int t = 0; t < n; ++t) {
      if (mh->dev[t] == mh->dev[s]) continue;
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, mh->dev[s], mh->dev[t]) == hipSuccess && can) {
        const hipError_t e = hipDeviceEnablePeerAccess(mh->dev[t], 0);
        if (e != hipSuccess) (void)hipGetLastError(); // already enabled is fine; hipMemcpyPeerAsync stages through the host otherwise
      }
    }
    HIPCK(hipStreamCreateWithFlags(&mh->st[s], hipStreamNonBlocking));
    HIPCK(hipEventCreate(&mh->ev_done[s]));
    HIPCK(hipEventCreate(&mh->ev_ready[s]));
    for (int t = 0; t < n; ++t) {
      if (t == s) continue;
      HIPCK(hipStreamCreateWithFlags(&mh->cs[s][t], hipStreamNonBlocking));
      HIPCK(hipEventCreateWithFlags(&mh->ev_arr[s][t], hipEventDisableTiming));
    }
  }
  // shard handles, created concurrently (each uploads its own slices)
  std::vector<std::vector<int64_t>> rp((size_t)n), cp((size_t)n);
  int rc = run_all(mh, [&](int s) -> int {
    glrm_problem q = *p;
    q.row_begin = mh->rbs[s]; q.row_end = mh->rbs[s + 1];
    q.col_begin = mh->cbs[s]; q.col_end = mh->cbs[s + 1];
    if (!mh->dense) {
      const int64_t r0 = p->rowptr[q.row_begin], c0 = p->colptr[q.col_begin];
      rp[s].resize((size_t)(q.row_end - q.row_begin) + 1);
      cp[s].resize((size_t)(q.col_end - q.col_begin) + 1);
      for (size_t i = 0; i < rp[s].size(); ++i) rp[s][i] = p->rowptr[q.row_begin + (int64_t)i] - r0;
      for (size_t i = 0; i < cp[s].size(); ++i) cp[s][i] = p->colptr[q.col_begin + (int64_t)i] - c0;
      q.rowptr = rp[s].data(); q.colidx = p->colidx ? p->colidx + r0 : nullptr; q.rowvals = p->rowvals ? p->rowvals + r0 : nullptr;
      q.colptr = cp[s].data(); q.rowidx = p->rowidx ? p->rowidx + c0 : nullptr; q.colvals = p->colvals ? p->colvals + c0 : nullptr;
    }
    if (p->n_rx != 1) { q.rx = p->rx + q.row_begin; q.n_rx = q.row_end - q.row_begin; }
    if (p->n_ry != 1) { q.ry = p->ry + q.col_begin; q.n_ry = q.col_end - q.col_begin; }
    glrm_options oo{};
    if (o) oo = *o;
    oo.device_id = mh->dev[s];
    oo.stream = (void*)mh->st[s];
    oo.caller_stream = 1;
    q.flags |= GLRM_PROBLEM_DEFER_SETUP; // the kernel families are chosen below, from the signature of the WHOLE problem
    return glrm_hip_create(&mh->sh[s], &q, &oo);
  });
  if (rc) return rc;
  glrm_signature whole{};
  for (int s = 0; s < n; ++s) {
    glrm_signature l{};
    if ((rc = glrm_hip_signature(mh->sh[s], &l))) return rc;
    whole.nnz_rows += l.nnz_rows; whole.nnz_cols += l.nnz_cols;
    whole.max_row_len = std::max(whole.max_row_len, l.max_row_len); whole.max_col_len = std::max(whole.max_col_len, l.max_col_len);
    whole.rows_unordered = std::max(whole.rows_unordered, l.rows_unordered); whole.cols_unordered = std::max(whole.cols_unordered, l.cols_unordered);
  }
  if ((rc = run_all(mh, [&](int s) -> int { return glrm_hip_finalize(mh->sh[s], &whole); }))) return rc;
  mh->ld = glrm_hip_factor_ld(mh->sh[0]);
  for (int s = 0; s < n; ++s) {
    HIPCK(hipSetDevice(mh->dev[s]));
    const size_t xb = (size_t)mh->ld * mh->m * 8, yb = (size_t)mh->ld * mh->d * 8;
    HIPCK(hipMalloc((void**)&mh->dX[s], xb));
    HIPCK(hipMalloc((void**)&mh->dY[s], yb));
    HIPCK(hipMalloc((void**)&mh->dObjCol[s], (size_t)mh->nn * 8));
    HIPCK(hipMalloc((void**)&mh->dObjRow[s], (size_t)mh->m * 8));
    HIPCK(hipMemsetAsync(mh->dX[s], 0, xb, mh->st[s]));
    HIPCK(hipMemsetAsync(mh->dY[s], 0, yb, mh->st[s]));
    HIPCK(hipMemsetAsync(mh->dObjCol[s], 0, (size_t)mh->nn * 8, mh->st[s]));
    HIPCK(hipMemsetAsync(mh->dObjRow[s], 0, (size_t)mh->m * 8, mh->st[s]));
    if ((rc = glrm_hip_bind_buffers(mh->sh[s], mh->dX[s], mh->dY[s], mh->dObjCol[s], mh->dObjRow[s]))) return rc;
  }
  mh->exchange = 0;
  const int want = env_int("GLRM_HIP_EXCHANGE_RCCL", mo->exchange);
  if (want == 1 && n > 1) {
    Rccl& R = Rccl::get();
    // RCCL wants one device per rank (the test build's GLRM_HIP_RCCL_ALLOW_SHARED=1 lifts that for the suite's stand-in library)
#ifdef GLRM_HIP_TESTING
    const bool shared_ok = env_int("GLRM_HIP_RCCL_ALLOW_SHARED", 0) != 0;
#else
    const bool shared_ok = false;
#endif
    if (R.ok && (distinct || shared_ok)) {
      mh->comms.assign(n, nullptr);
      const int e = R.CommInitAll(mh->comms.data(), n, mh->dev.data());
      if (e) return fail(GLRM_ERR_COMM, "ncclCommInitAll failed: %s", R.GetErrorString ? R.GetErrorString(e) : "?");
      mh->exchange = 1;
    } // else: direct path (shards sharing a device, or no librccl.so)
  }
  mh->x_chunks = (mo->x_chunks >= 2 && n > 1 && !mh->dense && mh->exchange == 0) ? mo->x_chunks : 1;
  // arrival order: the direct exchange of list problems (the dense path walks A, not X, super-tile by super-tile)
  mh->arrival = (n > 1 && !mh->dense && mh->exchange == 0 && env_int("GLRM_HIP_MULTI_ARRIVAL", mo->arrival == 2 ? 0 : 1)) ? 1 : 0;
  mh->ev_chunk.assign(n, std::vector<std::vector<hipEvent_t>>(n));
  if (mh->arrival)
    for (int s = 0; s < n; ++s) {
      HIPCK(hipSetDevice(mh->dev[s]));
      for (int t = 0; t < n; ++t) {
        if (t == s) continue;
        mh->ev_chunk[s][t].assign((size_t)mh->x_chunks, nullptr);
        for (int j = 0; j < mh->x_chunks; ++j) HIPCK(hipEventCreateWithFlags(&mh->ev_chunk[s][t][j], hipEventDisableTiming));
      }
    }
#ifdef GLRM_HIP_TESTING
  // link emulation (see the head of this file)
  const char* eg = getenv("GLRM_EXCHANGE_EMULATE_GBPS");
  const double gbps = (eg && *eg) ? atof(eg) : 0.0;
  if (gbps > 0.0 && n > 1 && mh->exchange == 0) {
    LinkEmu& e = mh->emu;
    int share = 1;
    for (int s = 0; s < n; ++s) {
      int c = 0;
      for (int t = 0; t < n; ++t) c += mh->dev[t] == mh->dev[s];
      share = std::max(share, c);
    }
    e.dilate = std::max(1, env_int("GLRM_EXCHANGE_EMULATE_DILATE", share));
    e.gbps = gbps;
    int can = 0;
    if (hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, mh->dev[0]) != hipSuccess) { (void)hipGetLastError(); can = 0; }
    e.mode = env_int("GLRM_EXCHANGE_EMULATE_MODE", can ? 1 : 2);
    if (e.mode != 1 && e.mode != 2) return fail(GLRM_ERR_INVALID, "GLRM_EXCHANGE_EMULATE_MODE must be 1 (stream wait-value) or 2 (delay kernels)");
    if (e.mode == 1 && !can) return fail(GLRM_ERR_UNSUPPORTED, "GLRM_EXCHANGE_EMULATE_MODE=1: the device does not support hipStreamWaitValue64");
    if (e.mode == 1) {
      e.mark.assign(n, nullptr); e.release.assign(n, nullptr); e.ticket.assign(n, 0);
      for (int s = 0; s < n; ++s) {
        HIPCK(hipSetDevice(mh->dev[s]));
        HIPCK(hipExtMallocWithFlags((void**)&e.mark[s], 8, hipMallocSignalMemory));
        HIPCK(hipExtMallocWithFlags((void**)&e.release[s], 8, hipMallocSignalMemory));
        *e.mark[s] = 0; *e.release[s] = 0; // signal memory is host-visible
      }
      e.timer = std::thread([mh] { mh->emu.timer_loop(); });
    } else {
      e.t0.assign(n, nullptr);
      int khz = 0;
      if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, mh->dev[0]) == hipSuccess && khz > 0) e.wall_hz = khz * 1e3;
      else (void)hipGetLastError();
      for (int s = 0; s < n; ++s) {
        HIPCK(hipSetDevice(mh->dev[s]));
        HIPCK(hipMalloc((void**)&e.t0[s], 8));
      }
    }
    e.bytes_per_s = gbps * 1e9 / e.dilate; // set last: the hooks are live from here
  }
#else
  if (getenv("GLRM_EXCHANGE_EMULATE_GBPS") && *getenv("GLRM_EXCHANGE_EMULATE_GBPS") && atof(getenv("GLRM_EXCHANGE_EMULATE_GBPS")) > 0.0)
    return fail(GLRM_ERR_UNSUPPORTED, "GLRM_EXCHANGE_EMULATE_GBPS: the link emulator lives in the test build (libglrm_hip_testing.so), not in this library");
#endif
  return GLRM_OK;
}

extern "C" int glrm_hip_multi_create(glrm_multi** out, const glrm_problem* p, const glrm_options* o, const glrm_multi_options* mo) {
  if (!out || !p || !mo) return fail(GLRM_ERR_INVALID, "NULL argument");
  *out = nullptr;
  if ((p->flags & GLRM_PROBLEM_ROWS_FROM_COLS) && !(p->flags & GLRM_PROBLEM_DEVICE_ARRAYS)) {
    // Omega is a sparse matrix's pattern, handed over as its column view only: every shard needs its rows' lists, so the row view is
    // built here, once, by a counting transpose on the host (columns in order => every row's list ascending), and the shards are cut
    // from both views as usual
    if (p->dense_A || p->rowptr || p->colidx || p->rowvals) return fail(GLRM_ERR_INVALID, "GLRM_PROBLEM_ROWS_FROM_COLS: rowptr / colidx / rowvals must be NULL");
    if (p->m <= 0 || p->n <= 0 || !p->colptr) return fail(GLRM_ERR_INVALID, "m, n must be positive and colptr given");
    // the structure of the column view is checked BEFORE anything is counted or written through it (a colptr that does not start
    // at 0, is not monotone or ends beyond the arrays made the fill loop read and write past the vectors; the single-device path checks first)
    if (p->colptr[0] != 0) return fail(GLRM_ERR_INVALID, "colptr[0] must be 0");
    for (int64_t f = 0; f < p->n; ++f)
      if (p->colptr[f + 1] < p->colptr[f]) return fail(GLRM_ERR_INVALID, "colptr is not monotone at %lld", (long long)f);
    const int64_t nz = p->colptr[p->n]; // >= 0 and >= every other entry from here on
    if (nz > 0 && (!p->rowidx || !p->colvals)) return fail(GLRM_ERR_INVALID, "index / value arrays are NULL");
    std::vector<int64_t> rowptr, fill;
    std::vector<int32_t> colidx;
    std::vector<double> rowvals;
    try {
      rowptr.assign((size_t)p->m + 1, 0); fill.resize((size_t)p->m);
      colidx.resize((size_t)nz); rowvals.resize((size_t)nz);
    } catch (const std::exception&) { // nothing may unwind across the extern "C" boundary
      return fail(GLRM_ERR_OOM, "out of host memory for the row view of %lld observations", (long long)nz);
    }
    for (int64_t t = 0; t < nz; ++t) {
      if (p->rowidx[t] < 0 || p->rowidx[t] >= p->m) return fail(GLRM_ERR_INVALID, "rowidx holds an index outside [0, m)");
      rowptr[(size_t)p->rowidx[t] + 1] += 1;
    }
    for (int64_t i = 0; i < p->m; ++i) { rowptr[i + 1] += rowptr[i]; fill[i] = rowptr[i]; }
    for (int64_t f = 0; f < p->n; ++f)
      for (int64_t t = p->colptr[f]; t < p->colptr[f + 1]; ++t) {
        const int64_t r = p->rowidx[t], at = fill[r]++;
        if (at >= rowptr[r + 1]) return fail(GLRM_ERR_INVALID, "the column view changed while it was read"); // (cannot happen on a checked, unchanged view)
        colidx[at] = (int32_t)f;
        rowvals[at] = p->colvals[t];
      }
    glrm_problem q = *p;
    q.flags &= ~GLRM_PROBLEM_ROWS_FROM_COLS;
    q.rowptr = rowptr.data(); q.colidx = colidx.data(); q.rowvals = rowvals.data();
    return glrm_hip_multi_create(out, &q, o, mo); // (create copies every slice to its device: the vectors may go afterwards)
  }
  if (mo->n_shards < 1 || mo->n_shards > 64) return fail(GLRM_ERR_INVALID, "n_shards must be in 1..64");
  if (mo->arrival < 0 || mo->arrival > 2) return fail(GLRM_ERR_INVALID, "glrm_multi_options.arrival must be 0, 1 or 2");
  if (p->flags & GLRM_PROBLEM_DEVICE_ARRAYS) return fail(GLRM_ERR_UNSUPPORTED, "glrm_hip_multi_create takes host arrays (it slices them per device)");
  if (p->m <= 0 || p->n <= 0 || p->k <= 0) return fail(GLRM_ERR_INVALID, "m, n, k must be positive");
  if (!(p->row_begin == 0 && p->row_end == p->m && p->col_begin == 0 && p->col_end == p->n))
    return fail(GLRM_ERR_INVALID, "glrm_hip_multi_create takes the whole problem (row/col ranges [0,m) x [0,n)); it shards it itself");
  if (!p->dense_A && (!p->rowptr || !p->colptr)) return fail(GLRM_ERR_INVALID, "rowptr / colptr are NULL");
  if (!p->losses || !(p->n_losses == 1 || p->n_losses == p->n) || !p->rx || !(p->n_rx == 1 || p->n_rx == p->m) || !p->ry ||
      !(p->n_ry == 1 || p->n_ry == p->n))
    return fail(GLRM_ERR_INVALID, "descriptor counts must be 1 or one per column / row");
  if (!p->dense_A) {
    if (p->rowptr[0] != 0 || p->colptr[0] != 0) return fail(GLRM_ERR_INVALID, "rowptr[0] / colptr[0] must be 0");
    for (int64_t s = 0; s < p->m; ++s) if (p->rowptr[s + 1] < p->rowptr[s]) return fail(GLRM_ERR_INVALID, "rowptr is not monotone at %lld", (long long)s);
    for (int64_t s = 0; s < p->n; ++s) if (p->colptr[s + 1] < p->colptr[s]) return fail(GLRM_ERR_INVALID, "colptr is not monotone at %lld", (long long)s);
  }
  int prev = 0;
  (void)hipGetDevice(&prev);
  glrm_multi* mh = new (std::nothrow) glrm_multi();
  if (!mh) return fail(GLRM_ERR_OOM, "out of host memory");
  mh->n = mo->n_shards;
  mh->pool.start(mh->n);
  const int rc = multi_create_impl(mh, p, o, mo);
  if (rc) {
    char keep[sizeof g_err];
    memcpy(keep, g_err, sizeof keep);
    glrm_hip_multi_destroy(mh);
    memcpy(g_err, keep, sizeof keep);
    (void)hipSetDevice(prev);
    return rc;
  }
  (void)hipSetDevice(prev);
  *out = mh;
  return GLRM_OK;
}

extern "C" int glrm_hip_multi_set_regularizers(glrm_multi* mh, const glrm_reg* rx, int64_t n_rx, const glrm_reg* ry, int64_t n_ry) {
  if (!mh || !rx || !ry) return fail(GLRM_ERR_INVALID, "NULL argument");
  if (n_rx != mh->n_rx || n_ry != mh->n_ry) return fail(GLRM_ERR_INVALID, "regularizer counts must match the create call");
  for (int s = 0; s < mh->n; ++s) {
    const glrm_reg* sx = n_rx == 1 ? rx : rx + mh->rbs[s];
    const glrm_reg* sy = n_ry == 1 ? ry : ry + mh->cbs[s];
    const int rc = glrm_hip_set_regularizers(mh->sh[s], sx, n_rx == 1 ? 1 : mh->rbs[s + 1] - mh->rbs[s], sy, n_ry == 1 ? 1 : mh->cbs[s + 1] - mh->cbs[s]);
    if (rc) return rc;
  }
  return GLRM_OK;
}

extern "C" int glrm_hip_multi_info(glrm_multi* mh, int64_t* row_bounds, int64_t* col_bounds, int32_t* exchange_used, double* exchange_ms) {
  if (!mh) return fail(GLRM_ERR_INVALID, "NULL handle");
  if (row_bounds) memcpy(row_bounds, mh->rbs.data(), ((size_t)mh->n + 1) * 8);
  if (col_bounds) memcpy(col_bounds, mh->cbs.data(), ((size_t)mh->n + 1) * 8);
  if (exchange_used) *exchange_used = mh->exchange;
  if (exchange_ms) *exchange_ms = mh->exchange_ms;
  return GLRM_OK;
}

extern "C" int glrm_hip_multi_fit(glrm_multi* mh, const glrm_params* prm, double* X, double* Y, double* objective, double* seconds,
                                  int64_t cap, int64_t* n_recorded) {
  if (!mh || !prm || !X || !Y || !objective || !seconds || !n_recorded) return fail(GLRM_ERR_INVALID, "NULL argument");
  if (prm->max_iter < 0 || cap < prm->max_iter + 1) return fail(GLRM_ERR_INVALID, "objective/seconds capacity must be >= max_iter+1");
  if (prm->inner_iter_X < 1 || prm->inner_iter_Y < 1) return fail(GLRM_ERR_INVALID, "inner iteration counts must be >= 1");
  double ynorm = 0.0; // norm(Y)==0 guard, proxgrad.jl:45-48
  for (int64_t i = 0; i < (int64_t)mh->k * mh->d; ++i) ynorm += Y[i] * Y[i];
  if (ynorm == 0.0) return fail(GLRM_ERR_INVALID, "Y is all zeros (the reference cannot start from Y == 0)");
  int prev = 0;
  (void)hipGetDevice(&prev);
  struct Restore { int d; ~Restore() { (void)hipSetDevice(d); } } restore{prev};
  int rc;
  mh->exchange_ms = 0.0;
  if ((rc = run_all(mh, [&](int s) {                                    // X = glrm.X; Y = glrm.Y (:43), alpharow / alphacol (:69-70)
         int r = glrm_hip_set_factors(mh->sh[s], X, Y);
         return r ? r : glrm_hip_reset_stepsizes(mh->sh[s], prm->stepsize);
       })))
    return rc;
  int64_t nnz_rows = 0;
  for (int s = 0; s < mh->n; ++s) nnz_rows += mh->sh[s]->nnz_r;
  // arrival order: what shard t's Y half-step is told about the rows of X -- its own block is there, then chunk 0 of every peer (ring
  // order from t: the order the peers push in), chunk 1 of every peer, ...; the events are re-recorded by every iteration's pushes
  const int n = mh->n, C = mh->x_chunks;
  const bool arr = mh->arrival != 0;
  std::vector<std::vector<glrm_arrival>> arrv((size_t)n);
  if (arr) {
    for (int t = 0; t < n; ++t) {
      arrv[t].push_back(glrm_arrival{mh->rbs[t], mh->rbs[t + 1], nullptr});
      for (int j = 0; j < C; ++j)
        for (int d = 1; d < n; ++d) {
          const int s = (t + n - d) % n; // s pushes to (s + d) % n = t as its d-th destination
          const int64_t blk = mh->rbs[s + 1] - mh->rbs[s];
          const int64_t lo = mh->rbs[s] + blk * j / C, hi = mh->rbs[s] + blk * (j + 1) / C;
          if (hi > lo) arrv[t].push_back(glrm_arrival{lo, hi, (void*)mh->ev_chunk[s][t][j]});
        }
    }
    mh->wait_y0.assign((size_t)n, 0.0);
    if (mh->profile)
      for (int s = 0; s < n; ++s) {
        glrm_kernel_stats ks{};
        if ((rc = glrm_hip_kernel_stats(mh->sh[s], &ks, 0))) return rc;
        mh->wait_y0[s] = ks.ms_wait_y;
      }
  }
  // a shard may not overwrite its block of X (Y) while a push of the previous iteration's exchanges still reads it: at the start of an
  // iteration its stream waits for its own outgoing copies (one copy stream per link keeps them in order, so the last event covers all;
  // they have long finished in practice -- a whole half-step lies in between; with link emulation they "finish" late)
  auto wait_outgoing = [&]() -> int {
    if (n == 1 || mh->exchange == 1) return GLRM_OK;
    for (int s = 0; s < n; ++s) {
      COMMCK(hipSetDevice(mh->dev[s]));
      for (int t = 0; t < n; ++t)
        if (t != s) COMMCK(hipStreamWaitEvent(mh->st[s], mh->ev_arr[s][t], 0));
    }
    return GLRM_OK;
  };
  const double scaled_abs_tol = prm->abs_tol * (double)nnz_rows;       // :72
  if ((rc = multi_objective(mh, &objective[0]))) return rc;           // :76
  seconds[0] = 0.0;
  int64_t nrec = 1;
  double t = now_s();
  for (int64_t i = 1; i <= prm->max_iter; ++i) {                       // :107
    if (prm->inner_iter_X > 1 || prm->inner_iter_Y > 1)
      if ((rc = run_all(mh, [&](int s) { return glrm_hip_reset_stepsizes(mh->sh[s], prm->stepsize); }))) return rc; // :112-115
    if ((rc = wait_outgoing())) return rc;
    for (int64_t in = 0; in + 1 < prm->inner_iter_X; ++in)              // inner sweeps touch own rows only: exchange once
      if ((rc = run_all(mh, [&](int s) { return glrm_hip_step_x(mh->sh[s], prm->min_stepsize); }))) return rc;
    if (C > 1 || arr) { // last inner sweep in row chunks: the push of chunk j overlaps the sweep of chunk j+1
      for (int j = 0; j < C; ++j) {
        if ((rc = run_all(mh, [&](int s) {
               if (C == 1) return glrm_hip_step_x(mh->sh[s], prm->min_stepsize);
               const int64_t ml = mh->rbs[s + 1] - mh->rbs[s];
               return glrm_hip_step_x_range(mh->sh[s], ml * j / C, ml * (j + 1) / C, prm->min_stepsize);
             })))
          return rc;
        if ((rc = exchange_push(mh, mh->dX, mh->rbs, mh->ld, -1, j, C, arr))) return rc;
      }
      if (!arr) { // the Y half-step needs all of X: wait here (with arrival order it waits block by block, inside step_y_arrival)
        if ((rc = exchange_wait(mh, -1))) return rc;
        account_exchange(mh);
      }
    } else {
      if ((rc = run_all(mh, [&](int s) { return glrm_hip_step_x(mh->sh[s], prm->min_stepsize); }))) return rc; // :117-158
      if ((rc = exchange_all(mh, mh->dX, mh->rbs, mh->ld))) return rc;
      account_exchange(mh);
    }
    for (int64_t in = 0; in < prm->inner_iter_Y; ++in)                  // :160-203
      if ((rc = run_all(mh, [&](int s) {
             if (arr && in == 0) return glrm_hip_step_y_arrival(mh->sh[s], prm->min_stepsize, arrv[s].data(), (int32_t)arrv[s].size());
             return glrm_hip_step_y(mh->sh[s], prm->min_stepsize);
           })))
        return rc;
    if ((rc = exchange_all(mh, mh->dY, mh->ybs, mh->ld))) return rc;
    account_exchange(mh);
    if ((rc = exchange_all(mh, mh->dObjCol, mh->cbs, 1, mh->exchange == 1 ? -1 : 0))) return rc;
    double obj = 0.0;
    if ((rc = sum_on_shard0(mh, mh->dObjCol[0], mh->nn, &obj))) return rc; // obj = sum(obj_by_col) :205 (synchronises shard 0)
    const double dt = now_s() - t;
    objective[nrec] = obj;
    seconds[nrec] = seconds[nrec - 1] + dt;                            // update_ch! (src/convergence.jl:22-26)
    ++nrec;
    t = now_s();
    const double dec = objective[nrec - 2] - obj;                      // :210
    if (i > 10 && (dec < scaled_abs_tol || dec / obj < prm->rel_tol)) break; // :211-213
  }
  for (int s = 0; s < mh->n; ++s)
    if ((rc = glrm_hip_synchronize(mh->sh[s]))) return rc;
  if (arr && mh->profile) { // what the Y half-steps spent in front of blocks of X that had not arrived yet
    double worst = 0.0;
    for (int s = 0; s < n; ++s) {
      glrm_kernel_stats ks{};
      if ((rc = glrm_hip_kernel_stats(mh->sh[s], &ks, 0))) return rc;
      worst = std::max(worst, ks.ms_wait_y - mh->wait_y0[s]);
    }
    mh->exchange_ms += worst;
  }
  if ((rc = glrm_hip_get_factors(mh->sh[0], X, Y))) return rc;
  *n_recorded = nrec;
  return GLRM_OK;
}
