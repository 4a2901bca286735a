// tests/fake_rccl/fake_rccl.cpp -- TEST DOUBLE of librccl.so.1 (test infrastructure, not the product).
//
// RCCL refuses two ranks on one device, and the builder's GPU box has one: jb_multi.cpp's N > 1 code --
// jb_comm_init, the ncclAllGather of the slab lengths, the grouped ncclSend / ncclRecv of ragged slabs,
// root != 0, an empty rank, a failing rank -- would first run on the driver's 8-GPU node.  This library
// implements the nine entry points jb_multi.cpp binds (Rccl table there) for N PROCESSES SHARING ONE DEVICE, over
// a POSIX shared-memory segment keyed by the communicator id: device buffers are copied to / from the segment
// with hipMemcpy, chunks are handed over with sequence counters.  Calls are synchronous (the real library
// enqueues on the stream): good enough for a functional rehearsal, useless for timing.  Every wait is bounded.
// Selected with JB_RCCL_LIBRARY=<path to this .so> (jb_multi.cpp) -- never on the library search path.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {
constexpr int kMaxRanks = 8;
constexpr size_t kChunk = 1 << 20, kAgBytes = 4096;
constexpr double kTimeoutS = 120.0;

struct Chan {
    std::atomic<uint64_t> ready, ack; // chunks written by the sender / consumed by the receiver
    uint64_t nbytes;
    alignas(64) uint8_t data[kChunk];
};
struct Shm {
    std::atomic<uint32_t> arrived, bar_count, bar_gen;
    uint32_t nranks;
    alignas(64) uint8_t ag[kMaxRanks][kAgBytes];
    Chan chan[kMaxRanks][kMaxRanks]; // [src][dst]
};
struct Comm {
    int nranks = 0, rank = 0;
    Shm *shm = nullptr;
    uint64_t sent[kMaxRanks] = {}, rcvd[kMaxRanks] = {};
};
struct Op {
    bool send;
    void *buf;
    size_t bytes, done;
    int peer;
    Comm *c;
    hipStream_t stream;
};
thread_local int g_group = 0;
thread_local std::vector<Op> g_ops;

double now()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
size_t type_size(ncclDataType_t t)
{
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    default: return 8;
    }
}
bool barrier(Comm *c)
{
    Shm *s = c->shm;
    const uint32_t gen = s->bar_gen.load(std::memory_order_acquire);
    if (s->bar_count.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)c->nranks) {
        s->bar_count.store(0, std::memory_order_relaxed);
        s->bar_gen.fetch_add(1, std::memory_order_acq_rel);
        return true;
    }
    const double t0 = now();
    while (s->bar_gen.load(std::memory_order_acquire) == gen) {
        if (now() - t0 > kTimeoutS)
            return false;
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    return true;
}
ncclResult_t run_ops(std::vector<Op> &ops)
{
    for (Op &o : ops)
        if (hipStreamSynchronize(o.stream) != hipSuccess) // what the stream did before the call is done
            return ncclUnhandledCudaError;
    const double t0 = now();
    size_t left = 0;
    for (Op &o : ops)
        left += o.bytes ? 1 : 0;
    while (left) {
        bool progress = false;
        for (Op &o : ops) {
            if (o.done >= o.bytes)
                continue;
            Comm *c = o.c;
            if (o.send) {
                Chan &ch = c->shm->chan[c->rank][o.peer];
                if (ch.ack.load(std::memory_order_acquire) != c->sent[o.peer])
                    continue; // the previous chunk has not been taken yet
                const size_t n = std::min(kChunk, o.bytes - o.done);
                if (hipMemcpy(ch.data, (const uint8_t *)o.buf + o.done, n, hipMemcpyDeviceToHost) != hipSuccess)
                    return ncclUnhandledCudaError;
                ch.nbytes = n;
                ch.ready.store(++c->sent[o.peer], std::memory_order_release);
                o.done += n;
            } else {
                Chan &ch = c->shm->chan[o.peer][c->rank];
                if (ch.ready.load(std::memory_order_acquire) != c->rcvd[o.peer] + 1)
                    continue;
                const size_t n = (size_t)ch.nbytes;
                if (n > o.bytes - o.done)
                    return ncclInvalidArgument; // the peer sends more than this receive holds
                if (hipMemcpy((uint8_t *)o.buf + o.done, ch.data, n, hipMemcpyHostToDevice) != hipSuccess)
                    return ncclUnhandledCudaError;
                ch.ack.store(++c->rcvd[o.peer], std::memory_order_release);
                o.done += n;
            }
            progress = true;
            if (o.done >= o.bytes)
                left--;
        }
        if (!progress) {
            if (now() - t0 > kTimeoutS)
                return ncclSystemError;
            std::this_thread::sleep_for(std::chrono::microseconds(20));
        }
    }
    return ncclSuccess;
}
} // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    memset(id->internal, 0, sizeof id->internal);
    memcpy(id->internal, "JBFR", 4);
    int fd = open("/dev/urandom", O_RDONLY);
    if (fd < 0 || read(fd, id->internal + 4, 16) != 16) {
        if (fd >= 0)
            close(fd);
        return ncclSystemError;
    }
    close(fd);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *out, int nranks, ncclUniqueId id, int rank)
{
    if (!out || nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks || memcmp(id.internal, "JBFR", 4) != 0)
        return ncclInvalidArgument;
    char name[64] = "/jbfr_";
    for (int i = 0; i < 16; i++)
        snprintf(name + 6 + 2 * i, 3, "%02x", (unsigned)(uint8_t)id.internal[4 + i]);
    int fd = shm_open(name, O_CREAT | O_RDWR, 0600);
    if (fd < 0)
        return ncclSystemError;
    if (ftruncate(fd, (off_t)sizeof(Shm)) != 0) {
        close(fd);
        return ncclSystemError;
    }
    void *p = mmap(nullptr, sizeof(Shm), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED)
        return ncclSystemError;
    Comm *c = new Comm();
    c->nranks = nranks;
    c->rank = rank;
    c->shm = (Shm *)p; // zero-filled by ftruncate: counters start at 0
    c->shm->nranks = (uint32_t)nranks;
    c->shm->arrived.fetch_add(1, std::memory_order_acq_rel);
    const double t0 = now();
    while (c->shm->arrived.load(std::memory_order_acquire) < (uint32_t)nranks) {
        if (now() - t0 > kTimeoutS) {
            shm_unlink(name);
            munmap(p, sizeof(Shm));
            delete c;
            return ncclSystemError;
        }
        std::this_thread::sleep_for(std::chrono::microseconds(100));
    }
    if (rank == 0)
        shm_unlink(name); // everybody holds a mapping: the name can go (no leak if a process dies later)
    *out = (ncclComm_t)c;
    return ncclSuccess;
}

// the number of ranks that ARRIVED in the shared segment, not the number this rank asked for
ncclResult_t ncclCommCount(const ncclComm_t comm, int *count)
{
    const Comm *c = (const Comm *)comm;
    if (!c || !count)
        return ncclInvalidArgument;
    *count = (int)c->shm->arrived.load(std::memory_order_acquire);
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    Comm *c = (Comm *)comm;
    if (!c)
        return ncclSuccess;
    munmap(c->shm, sizeof(Shm));
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclGroupStart()
{
    g_group++;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd()
{
    if (g_group <= 0)
        return ncclInvalidUsage;
    if (--g_group > 0)
        return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(g_ops);
    return run_ops(ops);
}

ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t stream)
{
    Comm *c = (Comm *)comm;
    if (!c || peer < 0 || peer >= c->nranks || peer == c->rank)
        return ncclInvalidArgument;
    g_ops.push_back(Op{true, (void *)buf, count * type_size(t), 0, peer, c, stream});
    if (g_group)
        return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(g_ops);
    return run_ops(ops);
}

ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t stream)
{
    Comm *c = (Comm *)comm;
    if (!c || peer < 0 || peer >= c->nranks || peer == c->rank)
        return ncclInvalidArgument;
    g_ops.push_back(Op{false, buf, count * type_size(t), 0, peer, c, stream});
    if (g_group)
        return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(g_ops);
    return run_ops(ops);
}

ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t t, ncclComm_t comm,
                           hipStream_t stream)
{
    Comm *c = (Comm *)comm;
    const size_t nb = count * type_size(t);
    if (!c || nb > kAgBytes)
        return ncclInvalidArgument;
    if (hipStreamSynchronize(stream) != hipSuccess)
        return ncclUnhandledCudaError;
    if (hipMemcpy(c->shm->ag[c->rank], send, nb, hipMemcpyDeviceToHost) != hipSuccess)
        return ncclUnhandledCudaError;
    std::atomic_thread_fence(std::memory_order_release);
    if (!barrier(c))
        return ncclSystemError;
    std::atomic_thread_fence(std::memory_order_acquire);
    for (int p = 0; p < c->nranks; p++)
        if (hipMemcpy((uint8_t *)recv + (size_t)p * nb, c->shm->ag[p], nb, hipMemcpyHostToDevice) != hipSuccess)
            return ncclUnhandledCudaError;
    if (!barrier(c)) // nobody overwrites a slot another rank is still reading
        return ncclSystemError;
    return ncclSuccess;
}

const char *ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "fake_rccl: HIP call failed";
    case ncclSystemError: return "fake_rccl: system error or timeout (a peer never arrived)";
    case ncclInvalidArgument: return "fake_rccl: invalid argument";
    case ncclInvalidUsage: return "fake_rccl: invalid usage";
    default: return "fake_rccl: error";
    }
}

} // extern "C"
