// One-time weight broadcast over RCCL / xGMI, inside the library (SURVEY section 2.3 "C-bcast", section 8e).
//
// The reference is single-device (src/bin/sample/main.rs:131); the multi-GPU deployment of this engine is N independent
// replicas (one process per GPU, prompt i -> GPU i mod N) that need the packed weight arena of rank 0 exactly once.  No
// collective runs inside the sampling loop.
//
// Schedule: xGMI is point to point (7 links x ~153 GB/s per GPU), so a ring or tree broadcast is bound by ONE link of the
// root.  Instead the arena is cut into `world` equal pieces:
//   1. scatter   root sends piece r to rank r -- world-1 different peers, i.e. all of the root's links at once
//                (ncclSend / ncclRecv inside one group);
//   2. all-gather every rank contributes its piece, in place (ncclAllGather) -- every link of every GPU carries traffic;
//   3. tail      the < world * 256 bytes left over by the 256-byte piece granularity go with one small ncclBroadcast.
// bcast_plan() below is the pure description of that schedule (offsets / lengths per rank); tests/test_cpu_distributed.py
// executes the same plan over gloo on a weight-arena byte image, the engine executes it over RCCL.
//
// RCCL is bound at run time (dlopen librccl.so.1; the PyTorch ROCm wheel ships the same SONAME, in which case the process
// ends up with ONE RCCL): the CPU-only build container and single-GPU users never load it.
#include "../../include/sdxl_mi355.h"
#include "engine.h"

#include <dlfcn.h>

#include <cstring>
#include <mutex>

namespace sdxl {

// ---- the schedule, as data -------------------------------------------------------------------------------------------
void bcast_plan(size_t bytes, int world, int rank, size_t* piece_off, size_t* piece_len, size_t* tail_off, size_t* tail_len) {
  SDXL_REQUIRE(world >= 1 && rank >= 0 && rank < world, "bcast_plan: bad rank / world");
  const size_t piece = world > 1 ? (bytes / (size_t)world) / 256 * 256 : 0;   // equal, 256-byte aligned pieces
  *piece_len = piece;
  *piece_off = piece * (size_t)rank;
  *tail_off = piece * (size_t)world;
  *tail_len = bytes - *tail_off;
}

namespace {
// the few RCCL entry points this file needs (signatures of rccl.h 2.x; ncclUniqueId is 128 opaque bytes, passed by value)
struct UniqueId { char internal[128]; };
typedef void* Comm;
enum { kNcclInt8 = 0 };
struct Rccl {
  void* so = nullptr;
  int (*GetUniqueId)(UniqueId*) = nullptr;
  int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
  int (*CommDestroy)(Comm) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void*, size_t, int, int, Comm, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, Comm, hipStream_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, Comm, hipStream_t) = nullptr;
  int (*Broadcast)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;
std::once_flag g_rccl_once;
std::string g_rccl_err;

void load_rccl() {
  for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    g_rccl.so = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (g_rccl.so) break;
  }
  if (!g_rccl.so) { g_rccl_err = std::string("cannot load RCCL: ") + dlerror(); return; }
  auto sym = [&](const char* n) { void* p = dlsym(g_rccl.so, n); if (!p && g_rccl_err.empty()) g_rccl_err = std::string("RCCL symbol missing: ") + n; return p; };
  g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))sym("ncclGetUniqueId");
  g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))sym("ncclCommInitRank");
  g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))sym("ncclCommDestroy");
  g_rccl.GroupStart = (decltype(g_rccl.GroupStart))sym("ncclGroupStart");
  g_rccl.GroupEnd = (decltype(g_rccl.GroupEnd))sym("ncclGroupEnd");
  g_rccl.Send = (decltype(g_rccl.Send))sym("ncclSend");
  g_rccl.Recv = (decltype(g_rccl.Recv))sym("ncclRecv");
  g_rccl.AllGather = (decltype(g_rccl.AllGather))sym("ncclAllGather");
  g_rccl.Broadcast = (decltype(g_rccl.Broadcast))sym("ncclBroadcast");
  g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))sym("ncclGetErrorString");
}
const Rccl& rccl() {
  std::call_once(g_rccl_once, load_rccl);
  if (!g_rccl_err.empty()) throw Error(g_rccl_err);
  return g_rccl;
}
void check(int rc, const char* what) {
  if (rc != 0) throw Error(std::string(what) + " failed: " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "RCCL error"));
}
}  // namespace

struct CommImpl {
  Comm comm = nullptr;
  int rank = 0, world = 1, device = 0;
  hipStream_t stream = nullptr;
};

void comm_bcast(CommImpl& c, void* base, size_t bytes, int root, hipStream_t s) {
  SDXL_REQUIRE(root >= 0 && root < c.world, "bcast: bad root");
  if (c.world == 1 || bytes == 0) return;
  SDXL_REQUIRE(base != nullptr, "bcast: null buffer");
  const Rccl& r = rccl();
  char* b = static_cast<char*>(base);
  size_t poff, plen, toff, tlen;
  bcast_plan(bytes, c.world, c.rank, &poff, &plen, &toff, &tlen);
  if (plen > 0) {
    // 1. scatter: piece p from the root to rank p (the root keeps its own)
    check(r.GroupStart(), "ncclGroupStart");
    if (c.rank == root) {
      for (int p = 0; p < c.world; ++p)
        if (p != root) check(r.Send(b + plen * (size_t)p, plen, kNcclInt8, p, c.comm, s), "ncclSend");
    } else {
      check(r.Recv(b + poff, plen, kNcclInt8, root, c.comm, s), "ncclRecv");
    }
    check(r.GroupEnd(), "ncclGroupEnd");
    // 2. in-place all-gather of the pieces
    check(r.AllGather(b + poff, b, plen, kNcclInt8, c.comm, s), "ncclAllGather");
  }
  // 3. tail
  if (tlen > 0) check(r.Broadcast(b + toff, b + toff, tlen, kNcclInt8, root, c.comm, s), "ncclBroadcast");
}

}  // namespace sdxl

using namespace sdxl;

struct sdxl_comm { CommImpl c; };

namespace {
thread_local std::string g_comm_err;
}
extern "C" {

// sdxl_last_error() lives in capi.hip; the comm entry points report through it via this hook
void sdxl_set_last_error_(const char* msg);

#define COMM_BEGIN try {
#define COMM_END                                                                  \
  return SDXL_OK;                                                                 \
  } catch (const std::exception& e) { sdxl_set_last_error_(e.what()); return SDXL_ERR_RUNTIME; } \
  catch (...) { sdxl_set_last_error_("unknown error"); return SDXL_ERR_RUNTIME; }

int sdxl_bcast_plan(size_t bytes, int world, int rank, size_t* piece_off, size_t* piece_len, size_t* tail_off, size_t* tail_len) {
  COMM_BEGIN
  SDXL_REQUIRE(piece_off && piece_len && tail_off && tail_len, "null argument");
  bcast_plan(bytes, world, rank, piece_off, piece_len, tail_off, tail_len);
  COMM_END
}
int sdxl_comm_unique_id(void* id_out_128) {
  COMM_BEGIN
  SDXL_REQUIRE(id_out_128 != nullptr, "null argument");
  UniqueId id;
  check(rccl().GetUniqueId(&id), "ncclGetUniqueId");
  std::memcpy(id_out_128, id.internal, 128);
  COMM_END
}
int sdxl_comm_create(int device_id, int rank, int world, const void* id_128, sdxl_comm** out) {
  COMM_BEGIN
  SDXL_REQUIRE(out && id_128 && world >= 1 && rank >= 0 && rank < world, "bad argument");
  SDXL_HIP(hipSetDevice(device_id));
  sdxl_comm* h = new sdxl_comm();
  h->c.rank = rank; h->c.world = world; h->c.device = device_id;
  try {
    UniqueId id;
    std::memcpy(id.internal, id_128, 128);
    check(rccl().CommInitRank(&h->c.comm, world, id, rank), "ncclCommInitRank");
    SDXL_HIP(hipStreamCreateWithFlags(&h->c.stream, hipStreamNonBlocking));
  } catch (...) { delete h; throw; }
  *out = h;
  COMM_END
}
void sdxl_comm_destroy(sdxl_comm* c) {
  if (!c) return;
  if (c->c.comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->c.comm);
  if (c->c.stream) (void)hipStreamDestroy(c->c.stream);
  delete c;
}
int sdxl_bcast_buffer(sdxl_comm* c, void* stream, void* base_dev, size_t bytes, int root) {
  COMM_BEGIN
  SDXL_REQUIRE(c != nullptr, "null communicator");
  SDXL_HIP(hipSetDevice(c->c.device));
  hipStream_t s = stream ? (hipStream_t)stream : c->c.stream;
  comm_bcast(c->c, base_dev, bytes, root, s);
  SDXL_HIP(hipStreamSynchronize(s));
  COMM_END
}

}  // extern "C"
