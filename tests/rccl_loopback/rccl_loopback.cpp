// Test double for librccl (BOSSX_RCCL_LIB=<this library>): the "ranks" of a communicator are THREADS of one
// process, possibly all on one device.  It lets the native multi-GPU driver of libbossx.so
// (bossx_dist_init / _chain / _update / _allgather: include/bossx.h) run with world > 1 on a one-GPU box.
//
// Only what the driver calls: ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclAllReduce
// (int32 / int64 / float64, sum / max), ncclAllGather, ncclGetErrorString.  A collective here is synchronous:
// every rank drains its stream, copies its operand to the host, meets the others at a barrier, reduces all
// operands itself (rank order: the result is the same on every rank) and copies the result back — a valid, if
// slow, implementation of RCCL's stream-ordered semantics.  Test infrastructure: nothing in the product links it.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace {

struct Group {
    int world = 0;
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t generation = 0;
    std::vector<std::vector<char>> operand;     // per rank, host copy
    std::vector<size_t> bytes;                  // per rank: what it brought (must agree)
    int members = 0;

    // all ranks arrive, or 60 s pass (a rank died: the others must not hang the box)
    bool barrier() {
        std::unique_lock<std::mutex> lk(m);
        const uint64_t gen = generation;
        if (++arrived == world) { arrived = 0; ++generation; cv.notify_all(); return true; }
        return cv.wait_for(lk, std::chrono::seconds(60), [&] { return generation != gen; });
    }
};

std::mutex g_registry_m;
std::map<std::string, Group *> g_registry;
std::atomic<uint64_t> g_next_id{1};

}  // namespace

struct ncclComm {
    Group *group;
    int rank;
};

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    if (!id) return ncclInvalidArgument;
    memset(id, 0, sizeof(*id));
    const uint64_t n = g_next_id.fetch_add(1);
    memcpy(id->internal, "loopback", 8);
    memcpy(id->internal + 8, &n, sizeof(n));
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    const std::string key(id.internal, sizeof(id.internal));
    std::lock_guard<std::mutex> lk(g_registry_m);
    Group *&g = g_registry[key];
    if (!g) {
        g = new Group;
        g->world = nranks;
        g->operand.resize(size_t(nranks));
        g->bytes.assign(size_t(nranks), 0);
    }
    if (g->world != nranks) return ncclInvalidArgument;
    ++g->members;
    *comm = new ncclComm{g, rank};
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    if (!comm) return ncclSuccess;
    {
        std::lock_guard<std::mutex> lk(g_registry_m);
        if (--comm->group->members == 0) {
            for (auto it = g_registry.begin(); it != g_registry.end(); ++it)
                if (it->second == comm->group) { g_registry.erase(it); break; }
            delete comm->group;
        }
    }
    delete comm;
    return ncclSuccess;
}

const char *ncclGetErrorString(ncclResult_t r) {
    switch (r) {
    case ncclSuccess: return "no error";
    case ncclInvalidArgument: return "invalid argument (loopback)";
    case ncclUnhandledCudaError: return "HIP error (loopback)";
    case ncclInternalError: return "a rank did not arrive within 60 s (loopback)";
    default: return "error (loopback)";
    }
}

}  // extern "C"

namespace {

size_t type_size(ncclDataType_t dt) {
    switch (dt) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
    }
}

template <typename T>
void reduce_into(std::vector<char> &acc, const std::vector<char> &x, ncclRedOp_t op) {
    T *a = reinterpret_cast<T *>(acc.data());
    const T *b = reinterpret_cast<const T *>(x.data());
    const size_t n = acc.size() / sizeof(T);
    for (size_t i = 0; i < n; ++i) a[i] = op == ncclSum ? T(a[i] + b[i]) : (b[i] > a[i] ? b[i] : a[i]);
}

// operand -> host, barrier; `result` built by the caller's functor from all operands; barrier; result -> device
template <typename F>
ncclResult_t exchange(ncclComm_t comm, const void *send, size_t send_bytes, void *recv, size_t recv_bytes, hipStream_t stream, F build) {
    if (!comm || !send || !recv) return ncclInvalidArgument;
    Group &g = *comm->group;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    std::vector<char> &mine = g.operand[size_t(comm->rank)];
    mine.resize(send_bytes);
    if (hipMemcpy(mine.data(), send, send_bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    g.bytes[size_t(comm->rank)] = send_bytes;
    if (!g.barrier()) return ncclInternalError;
    for (int r = 0; r < g.world; ++r)
        if (g.bytes[size_t(r)] != send_bytes) return ncclInvalidArgument;      // the ranks disagree about the collective
    std::vector<char> result(recv_bytes);
    build(g, result);
    if (!g.barrier()) return ncclInternalError;                                 // everyone has read the operands
    if (hipMemcpy(recv, result.data(), recv_bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}

}  // namespace

extern "C" {

ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream) {
    const size_t ts = type_size(dt);
    if (!ts || (op != ncclSum && op != ncclMax)) return ncclInvalidArgument;
    if (dt != ncclInt32 && dt != ncclInt64 && dt != ncclFloat64) return ncclInvalidArgument;
    return exchange(comm, send, count * ts, recv, count * ts, stream, [&](Group &g, std::vector<char> &out) {
        out = g.operand[0];
        for (int r = 1; r < g.world; ++r) {
            if (dt == ncclInt32) reduce_into<int32_t>(out, g.operand[size_t(r)], op);
            else if (dt == ncclInt64) reduce_into<int64_t>(out, g.operand[size_t(r)], op);
            else reduce_into<double>(out, g.operand[size_t(r)], op);
        }
    });
}

ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t dt, ncclComm_t comm, hipStream_t stream) {
    const size_t ts = type_size(dt);
    if (!ts || !comm) return ncclInvalidArgument;
    const size_t bytes = count * ts;
    return exchange(comm, send, bytes, recv, bytes * size_t(comm->group->world), stream, [&](Group &g, std::vector<char> &out) {
        for (int r = 0; r < g.world; ++r) memcpy(out.data() + size_t(r) * bytes, g.operand[size_t(r)].data(), bytes);
    });
}

}  // extern "C"
