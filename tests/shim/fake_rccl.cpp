// TEST INFRASTRUCTURE ONLY — an in-process stand-in for the nine RCCL entry points csrc/wg_comm.hip resolves
// with dlsym, so the W > 1 code path of the C-level all-to-all pipeline can run on a ONE-GPU box: every "rank"
// is a thread of the test process, all on the same device, and a send/recv pair becomes a device-to-device
// copy through a per-(dst, src) mailbox.  Selected with WGAMD_RCCL_LIBRARY=<this .so>; never loaded otherwise.
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <vector>

typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclInternalError = 3, ncclInvalidArgument = 4 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclInt32 = 2, ncclInt64 = 4 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
typedef struct { char internal[128]; } ncclUniqueId;

namespace {
struct message {
  const void* buf;
  size_t bytes;
  bool taken = false;
};
struct world {
  int size = 0;
  std::mutex m;
  std::condition_variable cv;
  std::vector<std::deque<message>> box;  // [dst * size + src]
  int barrier_count = 0, barrier_gen = 0;
  long long reduce_acc = 0, reduce_result = 0;
};
struct comm {
  world* w;
  int rank;
};
struct op {
  bool send;
  void* buf;
  size_t bytes;
  int peer;
  comm* c;
  hipStream_t stream;
};
std::mutex g_m;
std::map<uint64_t, world*> g_worlds;
uint64_t g_next_id = 1;
thread_local std::vector<op> t_ops;
thread_local int t_depth = 0;

size_t width(ncclDataType_t t) { return t == ncclInt8 ? 1 : t == ncclInt32 ? 4 : t == ncclInt64 ? 8 : 0; }

ncclResult_t flush()
{
  if (t_ops.empty()) return ncclSuccess;
  // 1. everything I send must be complete in memory before a peer copies it
  for (auto& o : t_ops)
    if (o.send && hipStreamSynchronize(o.stream) != hipSuccess) return ncclUnhandledCudaError;
  for (auto& o : t_ops) {
    if (!o.send) continue;
    world* w = o.c->w;
    std::lock_guard<std::mutex> lk(w->m);
    w->box[(size_t)o.peer * w->size + o.c->rank].push_back(message{o.buf, o.bytes});
    w->cv.notify_all();
  }
  // 2. take what was sent to me
  for (auto& o : t_ops) {
    if (o.send) continue;
    world* w = o.c->w;
    const void* src;
    {
      std::unique_lock<std::mutex> lk(w->m);
      auto& q = w->box[(size_t)o.c->rank * w->size + o.peer];
      w->cv.wait(lk, [&] { for (auto& m : q) if (!m.taken) return true; return false; });
      message* got = nullptr;
      for (auto& m : q) if (!m.taken) { got = &m; break; }
      if (got->bytes != o.bytes) return ncclInvalidArgument;
      src = got->buf;
    }
    if (hipMemcpyAsync(o.buf, src, o.bytes, hipMemcpyDeviceToDevice, o.stream) != hipSuccess) return ncclUnhandledCudaError;
    if (hipStreamSynchronize(o.stream) != hipSuccess) return ncclUnhandledCudaError;
    {
      std::lock_guard<std::mutex> lk(w->m);
      auto& q = w->box[(size_t)o.c->rank * w->size + o.peer];
      for (auto& m : q) if (!m.taken) { m.taken = true; break; }
      w->cv.notify_all();
    }
  }
  // 3. my send buffers may be reused once every receiver has copied them
  for (auto& o : t_ops) {
    if (!o.send) continue;
    world* w = o.c->w;
    std::unique_lock<std::mutex> lk(w->m);
    auto& q = w->box[(size_t)o.peer * w->size + o.c->rank];
    w->cv.wait(lk, [&] { return !q.empty() && q.front().taken; });
    q.pop_front();
  }
  t_ops.clear();
  return ncclSuccess;
}
}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id)
{
  std::lock_guard<std::mutex> lk(g_m);
  memset(id, 0, sizeof(*id));
  uint64_t v = g_next_id++;
  memcpy(id->internal, &v, sizeof(v));
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(comm** out, int nranks, ncclUniqueId id, int rank)
{
  uint64_t key;
  memcpy(&key, id.internal, sizeof(key));
  std::lock_guard<std::mutex> lk(g_m);
  world*& w = g_worlds[key];
  if (w == nullptr) {
    w       = new world;
    w->size = nranks;
    w->box.resize((size_t)nranks * nranks);
  }
  if (w->size != nranks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  *out = new comm{w, rank};
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(comm* c)
{
  delete c;  // worlds are leaked on purpose: test process
  return ncclSuccess;
}

ncclResult_t ncclGroupStart() { t_depth++; return ncclSuccess; }
ncclResult_t ncclGroupEnd() { return --t_depth == 0 ? flush() : ncclSuccess; }

ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t t, int peer, comm* c, hipStream_t s)
{
  t_ops.push_back(op{true, const_cast<void*>(buf), count * width(t), peer, c, s});
  return t_depth == 0 ? flush() : ncclSuccess;
}

ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t t, int peer, comm* c, hipStream_t s)
{
  t_ops.push_back(op{false, buf, count * width(t), peer, c, s});
  return t_depth == 0 ? flush() : ncclSuccess;
}

// only what wholememory_communicator_barrier needs: a 1 x int32 sum, which doubles as a rendezvous
ncclResult_t ncclAllReduce(const void* in, void* out, size_t count, ncclDataType_t t, ncclRedOp_t, comm* c, hipStream_t s)
{
  if (count != 1 || t != ncclInt32) return ncclInvalidArgument;
  int v = 0;
  if (hipMemcpyAsync(&v, in, 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
    return ncclUnhandledCudaError;
  world* w = c->w;
  long long total;
  {
    std::unique_lock<std::mutex> lk(w->m);
    int gen = w->barrier_gen;
    w->reduce_acc += v;
    if (++w->barrier_count == w->size) {
      w->reduce_result = w->reduce_acc;
      w->reduce_acc = 0;
      w->barrier_count = 0;
      w->barrier_gen++;
      w->cv.notify_all();
    } else {
      w->cv.wait(lk, [&] { return w->barrier_gen != gen; });
    }
    total = w->reduce_result;
  }
  int r = (int)total;
  if (hipMemcpyAsync(out, &r, 4, hipMemcpyHostToDevice, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
    return ncclUnhandledCudaError;
  return ncclSuccess;
}

ncclResult_t ncclCommCount(const comm* c, int* count)
{
  *count = c->w->size;
  return ncclSuccess;
}
const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "ok" : "fake-rccl error"; }

}  // extern "C"
