// rccl_stub.cpp -- TEST INFRASTRUCTURE: a stand-in for librccl.so that runs the collectives of ONE process's communicators with plain
// HIP copies, so that the exchange = 1 (RCCL) branch of csrc/glrm_multigpu.hip -- ncclCommInitAll, one grouped ncclAllGather over all
// communicators (equal blocks, in place), one ncclBroadcast per owner inside a group (ragged blocks) -- can be driven on a box with one
// GPU (tests/test_gpu_rccl_stub.py; the library is pointed at it with GLRM_HIP_RCCL_LIB and GLRM_HIP_RCCL_ALLOW_SHARED=1).  It restates
// the SEMANTICS the engine relies on (rccl.h): calls made between ncclGroupStart and ncclGroupEnd are matched across the communicators
// of the clique and issued at ncclGroupEnd; every rank's receive is ordered behind every sender's stream; sendbuff may be
// recvbuff + rank * count (in place).  It says nothing about RCCL's performance or its transport.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

namespace {
struct Comm { int rank, nranks, device; std::vector<Comm*>* clique; };
struct Call { int kind; Comm* c; const void* send; void* recv; size_t count; int root; hipStream_t st; }; // kind 0 all-gather, 1 broadcast
thread_local int g_depth = 0;
thread_local std::vector<Call> g_calls;
int g_allgathers = 0, g_broadcasts = 0, g_groups = 0;
constexpr int ncclSuccess = 0, ncclInvalidArgument = 4, ncclUnhandledCudaError = 1;
constexpr int ncclFloat64 = 8;

int fail(const char* what) { fprintf(stderr, "rccl_stub: %s\n", what); return ncclInvalidArgument; }

// make stream `dst` wait for everything queued on stream `src` so far
int order_after(hipStream_t dst, int dst_dev, hipStream_t src, int src_dev) {
  hipEvent_t e;
  if (hipSetDevice(src_dev) != hipSuccess || hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return ncclUnhandledCudaError;
  if (hipEventRecord(e, src) != hipSuccess) return ncclUnhandledCudaError;
  if (hipSetDevice(dst_dev) != hipSuccess || hipStreamWaitEvent(dst, e, 0) != hipSuccess) return ncclUnhandledCudaError;
  (void)hipEventDestroy(e); // released once the wait has consumed it
  return ncclSuccess;
}

int run(std::vector<Call>& calls) {
  // match the calls of a collective across the clique in issue order per communicator: the i-th call of every rank belongs together
  if (calls.empty()) return ncclSuccess;
  const int n = calls[0].c->nranks;
  std::vector<std::vector<Call*>> per((size_t)n);
  for (auto& c : calls) per[(size_t)c.c->rank].push_back(&c);
  const size_t rounds = per[0].size();
  for (int r = 0; r < n; ++r)
    if (per[(size_t)r].size() != rounds) return fail("the communicators of the clique made different numbers of calls inside one group");
  for (size_t i = 0; i < rounds; ++i) {
    const Call& c0 = *per[0][i];
    for (int r = 0; r < n; ++r) {
      const Call& c = *per[(size_t)r][i];
      if (c.kind != c0.kind || c.count != c0.count || c.root != c0.root) return fail("mismatched collective across ranks");
    }
    // every receiver first waits for every sender's stream (the data it reads was produced there)
    for (int d = 0; d < n; ++d)
      for (int s = 0; s < n; ++s) {
        if (s == d) continue;
        const Call &cd = *per[(size_t)d][i], &cs = *per[(size_t)s][i];
        if (c0.kind == 1 && s != c0.root) continue;
        const int rc = order_after(cd.st, cd.c->device, cs.st, cs.c->device);
        if (rc) return rc;
      }
    for (int d = 0; d < n; ++d) {
      const Call& cd = *per[(size_t)d][i];
      if (hipSetDevice(cd.c->device) != hipSuccess) return ncclUnhandledCudaError;
      if (c0.kind == 0) { // all-gather: rank d's recv[s * count ..] = rank s's send
        ++g_allgathers;
        for (int s = 0; s < n; ++s) {
          const Call& cs = *per[(size_t)s][i];
          char* dst = (char*)cd.recv + (size_t)s * cd.count * 8;
          if ((const void*)dst == cs.send) continue; // in place
          if (hipMemcpyAsync(dst, cs.send, cd.count * 8, hipMemcpyDefault, cd.st) != hipSuccess) return ncclUnhandledCudaError;
        }
      } else {            // broadcast from root: rank d's recv = root's send
        ++g_broadcasts;
        const Call& cr = *per[(size_t)c0.root][i];
        if (cd.recv != cr.send && hipMemcpyAsync(cd.recv, cr.send, cd.count * 8, hipMemcpyDefault, cd.st) != hipSuccess) return ncclUnhandledCudaError;
      }
    }
    // a sender may not overwrite its block before every receiver has read it: order the senders behind the receivers' copies
    for (int s = 0; s < n; ++s)
      for (int d = 0; d < n; ++d) {
        if (s == d) continue;
        const Call &cd = *per[(size_t)d][i], &cs = *per[(size_t)s][i];
        const int rc = order_after(cs.st, cs.c->device, cd.st, cd.c->device);
        if (rc) return rc;
      }
  }
  return ncclSuccess;
}
} // namespace

extern "C" {
int ncclCommInitAll(void** comms, int ndev, const int* devlist) {
  if (!comms || ndev < 1) return ncclInvalidArgument;
  auto* clique = new std::vector<Comm*>();
  for (int r = 0; r < ndev; ++r) {
    Comm* c = new Comm{r, ndev, devlist ? devlist[r] : r, clique};
    clique->push_back(c);
    comms[r] = c;
  }
  return ncclSuccess;
}
int ncclCommDestroy(void* comm) { delete (Comm*)comm; return ncclSuccess; }
int ncclGroupStart() { ++g_depth; return ncclSuccess; }
int ncclGroupEnd() {
  if (g_depth <= 0) return fail("ncclGroupEnd without ncclGroupStart");
  if (--g_depth > 0) return ncclSuccess;
  ++g_groups;
  std::vector<Call> calls;
  calls.swap(g_calls);
  int prev = 0;
  (void)hipGetDevice(&prev);
  const int rc = run(calls);
  (void)hipSetDevice(prev);
  return rc;
}
int ncclAllGather(const void* send, void* recv, size_t count, int dtype, void* comm, hipStream_t st) {
  if (dtype != ncclFloat64 || !comm) return ncclInvalidArgument;
  if (g_depth == 0) return fail("this stub runs grouped collectives only (one process drives every communicator)");
  g_calls.push_back(Call{0, (Comm*)comm, send, recv, count, 0, st});
  return ncclSuccess;
}
int ncclBroadcast(const void* send, void* recv, size_t count, int dtype, int root, void* comm, hipStream_t st) {
  if (dtype != ncclFloat64 || !comm) return ncclInvalidArgument;
  if (g_depth == 0) return fail("this stub runs grouped collectives only");
  g_calls.push_back(Call{1, (Comm*)comm, send, recv, count, root, st});
  return ncclSuccess;
}
const char* ncclGetErrorString(int e) { return e == 0 ? "no error" : "rccl_stub error"; }
// test hook: how many collectives ran
void rccl_stub_counts(int* allgathers, int* broadcasts, int* groups) { *allgathers = g_allgathers; *broadcasts = g_broadcasts; *groups = g_groups; }
}
