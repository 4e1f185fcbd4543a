// fake_rccl.cpp -- TEST INFRASTRUCTURE: a transport double for the handful of RCCL entry points
// libsqmc_gpu's in-library exchange uses (sqmc_gpu_comm_init / sqmc_gpu_shard_run), so that the
// exchange logic -- counts, offsets, grouping, the order of the collectives on two communicators --
// can run with SEVERAL ranks on ONE GPU, where real RCCL refuses ("duplicate GPU").  Ranks are
// processes on the same host; messages go device -> POSIX shared memory -> device, every call
// synchronises the stream it is given and blocks until the collective is complete (a stricter
// ordering than RCCL's, so anything that completes here cannot deadlock there for ordering
// reasons).  Loaded through SQMC_RCCL_LIB; never part of the product.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

namespace {
constexpr int MAXP = 8;
constexpr size_t SLOT = 1u << 20;          // per-rank slot of a collective (AllReduce / AllGather payloads are small)
constexpr size_t PAIR = 8u << 20;          // per ordered pair of ranks, one grouped send/recv
struct Region {
  std::atomic<unsigned long long> arrive;  // barrier counter
  unsigned long long plen[MAXP][MAXP];     // bytes of the message src -> dst of the current group
  char slot[MAXP][SLOT];
  char pair[1];                            // P * P * PAIR bytes follow
};
struct Pending { bool send; void *buf; size_t bytes; int peer; hipStream_t st; };
}
struct ncclComm {
  Region *r; int P, rank; unsigned long long nbar; std::string name; size_t bytes;
  char *pair(int src, int dst) { return r->pair + ((size_t)src * P + dst) * PAIR; }
  void barrier() {                          // all P ranks, sense by a monotonic counter
    nbar++;
    r->arrive.fetch_add(1);
    while (r->arrive.load() < nbar * (unsigned long long)P) usleep(20);
  }
};
static thread_local std::vector<Pending> g_group; static thread_local int g_depth = 0; static thread_local ncclComm *g_gcomm = nullptr;
static ncclComm *g_first = nullptr;       // the communicator libsqmc_gpu opens its groups on: a group without operations still meets its barriers
static size_t tsize(ncclDataType_t t) { return (t == ncclDouble || t == ncclUint64 || t == ncclInt64) ? 8 : (t == ncclUint32 || t == ncclInt32 || t == ncclFloat) ? 4 : 1; }

static ncclComm *open_region(const std::string &name, int P, int rank) {
  const size_t bytes = sizeof(Region) + (size_t)P * P * PAIR;
  int fd = shm_open(name.c_str(), O_CREAT | O_RDWR, 0600);
  if (fd < 0) return nullptr;
  if (ftruncate(fd, (off_t)bytes) != 0) { close(fd); return nullptr; }      // new pages read as zero: counters start at 0
  void *p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return nullptr;
  ncclComm *c = new ncclComm(); c->r = (Region *)p; c->P = P; c->rank = rank; c->nbar = 0; c->name = name; c->bytes = bytes;
  return c;
}

extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
  memset(id, 0, sizeof(*id));
  snprintf(id->internal, sizeof(id->internal), "/fakerccl_%d_%ld", (int)getpid(), (long)random());
  return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank) {
  if (nranks > MAXP) return ncclInvalidArgument;
  ncclComm *c = open_region(std::string(id.internal), nranks, rank);
  if (!c) return ncclSystemError;
  c->barrier();
  *comm = c;
  if (!g_first) g_first = c;
  return ncclSuccess;
}
ncclResult_t ncclCommSplit(ncclComm_t comm, int color, int key, ncclComm_t *out, ncclConfig_t *) {
  (void)key;
  ncclComm *c = open_region(comm->name + "_s" + std::to_string(color), comm->P, comm->rank);
  if (!c) return ncclSystemError;
  c->barrier();
  *out = c;
  return ncclSuccess;
}
ncclResult_t ncclCommCount(const ncclComm_t c, int *n) { *n = c->P; return ncclSuccess; }
ncclResult_t ncclCommDestroy(ncclComm_t c) {
  if (!c) return ncclSuccess;
  c->barrier();
  if (c == g_first) g_first = nullptr;
  if (c->rank == 0) shm_unlink(c->name.c_str());
  munmap(c->r, c->bytes);
  delete c;
  return ncclSuccess;
}
ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t c, hipStream_t st) {
  if (t != ncclDouble || op != ncclSum || count * 8 > SLOT) return ncclInvalidArgument;
  if (hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;
  if (hipMemcpy(c->r->slot[c->rank], send, count * 8, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
  c->barrier();
  std::vector<double> acc(count, 0.0);
  for (int q = 0; q < c->P; q++) { const double *v = (const double *)c->r->slot[q]; for (size_t i = 0; i < count; i++) acc[i] += v[i]; }   // rank order: same sums everywhere
  c->barrier();
  return hipMemcpy(recv, acc.data(), count * 8, hipMemcpyHostToDevice) == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
}
ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t t, ncclComm_t c, hipStream_t st) {
  const size_t b = count * tsize(t);
  if (b > SLOT) return ncclInvalidArgument;
  if (hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;
  if (hipMemcpy(c->r->slot[c->rank], send, b, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
  c->barrier();
  std::vector<char> all(b * c->P);
  for (int q = 0; q < c->P; q++) memcpy(all.data() + q * b, c->r->slot[q], b);
  c->barrier();
  return hipMemcpy(recv, all.data(), all.size(), hipMemcpyHostToDevice) == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
}
ncclResult_t ncclGroupStart() { g_depth++; return ncclSuccess; }
ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t st) {
  if (g_depth <= 0) return ncclInvalidUsage;           // only grouped point-to-point is modelled
  g_gcomm = c; g_group.push_back(Pending{true, (void *)buf, count * tsize(t), peer, st});
  return ncclSuccess;
}
ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t st) {
  if (g_depth <= 0) return ncclInvalidUsage;
  g_gcomm = c; g_group.push_back(Pending{false, buf, count * tsize(t), peer, st});
  return ncclSuccess;
}
// Every rank of the communicator passes through the group's two barriers, also one that posted nothing:
// libsqmc_gpu opens exactly one group per step on every rank.  A rank without a communicator in the group
// (no sends, no receives) cannot know which one to wait on, so sqmc's "group every step" is a precondition.
ncclResult_t ncclGroupEnd() {
  if (--g_depth > 0) return ncclSuccess;
  ncclComm *c = g_gcomm ? g_gcomm : g_first;
  ncclResult_t rc = ncclSuccess;
  if (c) {
    for (int q = 0; q < c->P; q++) c->r->plen[c->rank][q] = 0;
    for (const Pending &p : g_group) if (p.send) {
      if (p.bytes > PAIR) { rc = ncclInvalidArgument; continue; }
      hipStreamSynchronize(p.st);
      if (hipMemcpy(c->pair(c->rank, p.peer), p.buf, p.bytes, hipMemcpyDeviceToHost) != hipSuccess) rc = ncclUnhandledCudaError;
      c->r->plen[c->rank][p.peer] = p.bytes;
    }
    c->barrier();
    for (const Pending &p : g_group) if (!p.send) {
      if (c->r->plen[p.peer][c->rank] != p.bytes) { fprintf(stderr, "fake_rccl: rank %d expects %zu bytes from %d, which sent %llu\n", c->rank, p.bytes, p.peer, c->r->plen[p.peer][c->rank]); rc = ncclInvalidUsage; continue; }
      hipStreamSynchronize(p.st);
      if (hipMemcpy(p.buf, c->pair(p.peer, c->rank), p.bytes, hipMemcpyHostToDevice) != hipSuccess) rc = ncclUnhandledCudaError;
    }
    // a message nobody asked for is a protocol error of the caller too
    for (int q = 0; q < c->P; q++) if (c->r->plen[q][c->rank] != 0) {
      bool wanted = false; for (const Pending &p : g_group) if (!p.send && p.peer == q) wanted = true;
      if (!wanted) { fprintf(stderr, "fake_rccl: rank %d got %llu unexpected bytes from %d\n", c->rank, c->r->plen[q][c->rank], q); rc = ncclInvalidUsage; }
    }
    c->barrier();
  }
  g_group.clear(); g_gcomm = nullptr;
  return rc;
}
const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "fake_rccl error"; }
}
