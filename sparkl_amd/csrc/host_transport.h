// Host-staged stand-in for the RCCL entry points the slab-group loop uses (SPARKL_HIP_TRANSPORT=host).
//
// WHY.  sparkl_hip_group_substeps / _group_step exchange halos, flags, dt bounds, migration records and cut histograms through RCCL between
// ranks (the reference's counterpart: cuda_mpm_pipeline.rs:407-533, peer copies between contexts).  The GPU boxes this repository is tested
// on have ONE GPU, and RCCL refuses two ranks on one device -- so until round 5 the rank-asymmetric parts of that loop (bootstrap from the
// unique id, the calibration all-reduce of the overlap schedule, collective re-bin + migration + slab balance, the abort path) had only ever
// run with a world of one.  This file implements the SAME function table (struct Rccl in sparkl_hip.hip) on top of a POSIX shared-memory
// segment: device -> host copy, mailbox / reduction slots in shared memory, host -> device copy.  Two (or more) OS processes that share the
// one GPU then execute the identical C++ group loop with world > 1.  It is a TEST transport: every operation synchronises its stream and
// is executed on the host at ncclGroupEnd (or at once outside a group), which is a stronger ordering than RCCL's stream order, never a
// weaker one.  Nothing selects it unless the environment asks for it.
//
// Layout of the segment (name = the 128-byte "unique id"):
//   Header | world x ReduceSlot | world x world x Mailbox       (sparse: only touched pages exist)
// Mailbox (src -> dst): FIFO of chunks; the sender waits until the previous chunk was consumed, writes up to MAILBOX_CAP bytes and bumps
// `written`; the receiver consumes and bumps `read`.  Messages of one (src, dst) pair are matched in issue order, like ncclSend / ncclRecv.
// All-reduce number k of a communicator: every rank writes its operand to slot[rank].data[k & 1] and publishes seq = k; once all slots
// show k everyone reduces locally (two buffers suffice: nobody can publish k + 2 before everyone has read k).
// Every wait polls the shared abort flag and a deadline (SPARKL_HIP_COMM_TIMEOUT_S): a peer that died makes the call fail, not hang.
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

namespace hostcomm {

constexpr uint64_t MAGIC = 0x53504b484f535431ull;   // "SPKHOST1"
constexpr size_t MAILBOX_CAP = 4u << 20;
constexpr size_t REDUCE_CAP = 64u << 10;
constexpr int MAX_WORLD = 16;

struct Header {
    uint64_t magic;
    uint64_t world;
    uint64_t attached;
    uint64_t abort_flag;
    char pad[4096 - 32];
};
struct ReduceSlot {
    uint64_t seq;   // number of the last all-reduce this rank has published (1-based)
    char pad[56];
    unsigned char data[2][REDUCE_CAP];
};
struct Mailbox {
    uint64_t written, read;   // chunks
    uint64_t chunk_bytes;
    char pad[40];
    unsigned char data[MAILBOX_CAP];
};

struct Comm {
    int rank = 0, world = 1;
    unsigned char* base = nullptr;
    size_t bytes = 0;
    uint64_t reduce_seq = 0;
    uint64_t sent[MAX_WORLD] = {0}, received[MAX_WORLD] = {0};   // chunks this rank has written to / read from each peer's mailbox
    bool aborted = false;
    char name[128] = {0};
    Header* hdr() const { return reinterpret_cast<Header*>(base); }
    ReduceSlot* slot(int r) const { return reinterpret_cast<ReduceSlot*>(base + sizeof(Header)) + r; }
    Mailbox* box(int src, int dst) const { return reinterpret_cast<Mailbox*>(base + sizeof(Header) + (size_t)world * sizeof(ReduceSlot)) + ((size_t)src * world + dst); }
};
inline thread_local std::vector<Comm*> g_new_comms;
inline size_t segment_bytes(int world) { return sizeof(Header) + (size_t)world * sizeof(ReduceSlot) + (size_t)world * world * sizeof(Mailbox); }

inline uint64_t load(const uint64_t* p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
inline void store(uint64_t* p, uint64_t v) { __atomic_store_n(p, v, __ATOMIC_RELEASE); }

inline double timeout_s() {
    static const double t = getenv("SPARKL_HIP_COMM_TIMEOUT_S") ? atof(getenv("SPARKL_HIP_COMM_TIMEOUT_S")) : 300.0;
    return t;
}
struct Deadline {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    // false = give up (peer gone or communicator aborted)
    bool wait(const Comm* c) {
        if (load(&c->hdr()->abort_flag) || c->aborted) return false;
        if ((++spins & 63u) == 0) {
            const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (timeout_s() > 0.0 && el > timeout_s()) return false;
            if (el > 0.002) std::this_thread::sleep_for(std::chrono::microseconds(100));
        }
        return true;
    }
};

// ---- operations, queued between GroupStart and GroupEnd
struct Op {
    int kind;   // 0 send, 1 recv, 2 all-reduce
    Comm* comm;
    const void* src;
    void* dst;
    size_t bytes;
    int peer;
    ncclDataType_t type;
    ncclRedOp_t red;
    hipStream_t stream;
    std::vector<unsigned char> host;
    size_t off = 0;
    uint64_t seq = 0;   // all-reduce: its number on the communicator, taken when it is posted
    bool posted = false, done = false;
};
inline thread_local int g_depth = 0;
inline thread_local std::vector<Op> g_ops;

inline size_t type_bytes(ncclDataType_t t) {
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
        default: return 0;
    }
}
template <class T>
inline void reduce_typed(T* acc, const T* in, size_t n, ncclRedOp_t op) {
    for (size_t i = 0; i < n; ++i) {
        switch (op) {
            case ncclSum: acc[i] = (T)(acc[i] + in[i]); break;
            case ncclMax: acc[i] = in[i] > acc[i] ? in[i] : acc[i]; break;
            case ncclMin: acc[i] = in[i] < acc[i] ? in[i] : acc[i]; break;
            default: break;
        }
    }
}
inline bool reduce_into(void* acc, const void* in, size_t count, ncclDataType_t t, ncclRedOp_t op) {
    if (op != ncclSum && op != ncclMax && op != ncclMin) return false;
    switch (t) {
        case ncclUint8: reduce_typed((uint8_t*)acc, (const uint8_t*)in, count, op); return true;
        case ncclInt32: reduce_typed((int32_t*)acc, (const int32_t*)in, count, op); return true;
        case ncclUint32: reduce_typed((uint32_t*)acc, (const uint32_t*)in, count, op); return true;
        case ncclInt64: reduce_typed((int64_t*)acc, (const int64_t*)in, count, op); return true;
        case ncclUint64: reduce_typed((uint64_t*)acc, (const uint64_t*)in, count, op); return true;
        case ncclFloat32: reduce_typed((float*)acc, (const float*)in, count, op); return true;
        case ncclFloat64: reduce_typed((double*)acc, (const double*)in, count, op); return true;
        default: return false;
    }
}

inline ncclResult_t run_ops(std::vector<Op>& ops) {
    // 1. everything the streams have queued ahead of these operations has happened; operands come to the host
    for (Op& o : ops) {
        if (hipStreamSynchronize(o.stream) != hipSuccess) return ncclUnhandledCudaError;
        o.host.resize(o.bytes);
        if (o.kind != 1 && o.bytes && hipMemcpy(o.host.data(), o.src, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
        if (o.kind == 2 && o.bytes > REDUCE_CAP) return ncclInvalidArgument;
        if (o.bytes == 0) o.done = true;
    }
    // 2. progress loop: sends, receives and reductions of all local ranks advance together (a pair of local ranks, or a send and the
    //    matching receive of two processes, can only complete if neither blocks the other)
    Deadline dl;
    for (;;) {
        bool all = true, moved = false;
        for (size_t oi = 0; oi < ops.size(); ++oi) {
            Op& o = ops[oi];
            if (o.done) continue;
            Comm* c = o.comm;
            // issue order per (communicator, peer, kind), as ncclSend / ncclRecv match and as the two-buffer all-reduce needs: an operation
            // does not start before every earlier one of its kind to the same peer has finished (a skipped Send must not be overtaken by a
            // later one, a multi-chunk message must not interleave with the next, a second all-reduce must not post before the first has reduced)
            bool blocked = false;
            for (size_t e = 0; e < oi && !blocked; ++e) blocked = !ops[e].done && ops[e].comm == c && ops[e].kind == o.kind && ops[e].peer == o.peer;
            if (blocked) { all = false; continue; }
            if (o.kind == 0) {
                Mailbox* b = c->box(c->rank, o.peer);
                if (load(&b->read) == c->sent[o.peer]) {   // the previous chunk was consumed
                    const size_t m = std::min(MAILBOX_CAP, o.bytes - o.off);
                    memcpy(b->data, o.host.data() + o.off, m);
                    b->chunk_bytes = m;
                    store(&b->written, ++c->sent[o.peer]);
                    o.off += m;
                    moved = true;
                    if (o.off == o.bytes) o.done = true;
                }
            } else if (o.kind == 1) {
                Mailbox* b = c->box(o.peer, c->rank);
                if (load(&b->written) > c->received[o.peer]) {
                    const size_t m = (size_t)b->chunk_bytes;
                    if (m > o.bytes - o.off) return ncclInvalidArgument;   // the peer sent more than this receive expects
                    memcpy(o.host.data() + o.off, b->data, m);
                    store(&b->read, ++c->received[o.peer]);
                    o.off += m;
                    moved = true;
                    if (o.off == o.bytes) o.done = true;
                }
            } else {
                if (!o.posted) {
                    o.seq = ++c->reduce_seq;
                    memcpy(c->slot(c->rank)->data[o.seq & 1], o.host.data(), o.bytes);
                    store(&c->slot(c->rank)->seq, o.seq);
                    o.posted = true;
                    moved = true;
                }
                const uint64_t k = o.seq;
                bool ready = true;
                for (int r = 0; r < c->world; ++r) ready = ready && load(&c->slot(r)->seq) >= k;
                if (ready) {
                    memcpy(o.host.data(), c->slot(0)->data[k & 1], o.bytes);   // rank order: every rank forms the same sum bit for bit
                    for (int r = 1; r < c->world; ++r)
                        if (!reduce_into(o.host.data(), c->slot(r)->data[k & 1], o.bytes / type_bytes(o.type), o.type, o.red)) return ncclInvalidArgument;
                    o.done = true;
                    moved = true;
                }
            }
            all = all && o.done;
        }
        if (all) break;
        if (!moved && !dl.wait(ops[0].comm)) return ncclSystemError;
    }
    // 3. results go back to the device
    for (Op& o : ops)
        if (o.kind != 0 && o.bytes && hipMemcpy(o.dst, o.host.data(), o.bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}
// an error anywhere between GroupStart and GroupEnd (or CommAbort) leaves no half-open group behind on this thread: later groups would
// otherwise be queued and never executed
inline void reset_group_state() { g_depth = 0; g_ops.clear(); g_new_comms.clear(); }
inline ncclResult_t submit(Op&& o) {
    if (!o.comm || o.comm->aborted) { reset_group_state(); return ncclInvalidUsage; }
    g_ops.push_back(std::move(o));
    if (g_depth > 0) return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(g_ops);
    const ncclResult_t r = run_ops(ops);
    if (r != ncclSuccess) reset_group_state();
    return r;
}

// every rank has attached: rank 0 removes the name (every rank holds its mapping: nothing is left behind in /dev/shm whatever happens next)
inline ncclResult_t rendezvous(Comm* c) {
    Deadline dl;
    while (load(&c->hdr()->attached) < (uint64_t)c->world)
        if (!dl.wait(c)) return ncclSystemError;
    if (c->rank == 0) shm_unlink(c->name);
    return ncclSuccess;
}

// ---- the function table
inline ncclResult_t GetUniqueId(ncclUniqueId* id) {
    memset(id, 0, sizeof(*id));
    unsigned long long r = 0;
    if (FILE* f = fopen("/dev/urandom", "rb")) { if (fread(&r, sizeof(r), 1, f) != 1) r = 0; fclose(f); }
    snprintf(id->internal, sizeof(id->internal), "/spkhc_%d_%016llx", (int)getpid(), r);
    return ncclSuccess;
}
inline ncclResult_t CommInitRank(ncclComm_t* out, int world, ncclUniqueId id, int rank) {
    if (!out || world < 1 || world > MAX_WORLD || rank < 0 || rank >= world || id.internal[0] != '/') return ncclInvalidArgument;
    id.internal[sizeof(id.internal) - 1] = 0;
    const size_t bytes = segment_bytes(world);
    int fd = shm_open(id.internal, O_CREAT | O_RDWR, 0600);
    if (fd < 0) return ncclSystemError;
    if (ftruncate(fd, (off_t)bytes) != 0) { close(fd); return ncclSystemError; }
    void* m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return ncclSystemError;
    Comm* c = new Comm();
    c->rank = rank; c->world = world; c->base = (unsigned char*)m; c->bytes = bytes;
    store(&c->hdr()->world, (uint64_t)world);
    store(&c->hdr()->magic, MAGIC);
    __atomic_add_fetch(&c->hdr()->attached, 1, __ATOMIC_ACQ_REL);
    *out = reinterpret_cast<ncclComm_t>(c);
    snprintf(c->name, sizeof(c->name), "%s", id.internal);
    if (g_depth > 0) { g_new_comms.push_back(c); return ncclSuccess; }   // inside a group (several ranks of one thread): the rendezvous happens at GroupEnd
    return rendezvous(c);
}
inline ncclResult_t CommDestroy(ncclComm_t comm) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (!c) return ncclSuccess;
    if (c->base) munmap(c->base, c->bytes);
    delete c;
    return ncclSuccess;
}
inline ncclResult_t GroupStart() { ++g_depth; return ncclSuccess; }
inline ncclResult_t GroupEnd() {
    if (g_depth <= 0) return ncclInvalidUsage;
    if (--g_depth > 0) return ncclSuccess;
    {
        std::vector<Comm*> fresh;
        fresh.swap(g_new_comms);
        for (Comm* c : fresh) { const ncclResult_t r = rendezvous(c); if (r != ncclSuccess) { reset_group_state(); return r; } }
    }
    std::vector<Op> ops;
    ops.swap(g_ops);
    if (ops.empty()) return ncclSuccess;
    const ncclResult_t r = run_ops(ops);
    if (r != ncclSuccess) reset_group_state();
    return r;
}
inline ncclResult_t Send(const void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t st) {
    Op o{};
    o.kind = 0; o.comm = reinterpret_cast<Comm*>(comm); o.src = buf; o.dst = nullptr; o.bytes = count * type_bytes(t); o.peer = peer; o.type = t; o.red = ncclSum; o.stream = st;
    if (!o.comm || peer < 0 || peer >= o.comm->world || peer == o.comm->rank) { reset_group_state(); return ncclInvalidArgument; }
    return submit(std::move(o));
}
inline ncclResult_t Recv(void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t st) {
    Op o{};
    o.kind = 1; o.comm = reinterpret_cast<Comm*>(comm); o.src = nullptr; o.dst = buf; o.bytes = count * type_bytes(t); o.peer = peer; o.type = t; o.red = ncclSum; o.stream = st;
    if (!o.comm || peer < 0 || peer >= o.comm->world || peer == o.comm->rank) { reset_group_state(); return ncclInvalidArgument; }
    return submit(std::move(o));
}
inline ncclResult_t AllReduce(const void* src, void* dst, size_t count, ncclDataType_t t, ncclRedOp_t red, ncclComm_t comm, hipStream_t st) {
    Op o{};
    o.kind = 2; o.comm = reinterpret_cast<Comm*>(comm); o.src = src; o.dst = dst; o.bytes = count * type_bytes(t); o.peer = -1; o.type = t; o.red = red; o.stream = st;
    if (!o.comm || type_bytes(t) == 0) { reset_group_state(); return ncclInvalidArgument; }
    return submit(std::move(o));
}
inline const char* GetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "no error";
        case ncclUnhandledCudaError: return "host transport: HIP call failed";
        case ncclSystemError: return "host transport: peer gone, communicator aborted or timeout (SPARKL_HIP_COMM_TIMEOUT_S)";
        case ncclInvalidArgument: return "host transport: invalid argument";
        case ncclInvalidUsage: return "host transport: invalid usage";
        default: return "host transport: error";
    }
}
inline ncclResult_t CommAbort(ncclComm_t comm) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (!c) return ncclSuccess;
    c->aborted = true;
    reset_group_state();
    if (c->base) store(&c->hdr()->abort_flag, 1);   // (unlike ncclCommAbort this also wakes the peers; a KILLED rank cannot do it, they time out)
    return CommDestroy(comm);
}
inline ncclResult_t CommGetAsyncError(ncclComm_t comm, ncclResult_t* out) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    *out = (c && !c->aborted && !load(&c->hdr()->abort_flag)) ? ncclSuccess : ncclSystemError;
    return ncclSuccess;
}

}  // namespace hostcomm
