// frames_native.hip -- the native transport of include/ohevc_frames.h: the four callbacks of the frame-parallel decoder (one process per
// GPU) in C++ inside the product library.  Host code only.
//
// What travels per exchanged picture: the three sample planes exactly as the picture store lays them out
// (stride x height bytes, ohevc_pic_export / ohevc_pic_import) and one motion-field message = HEVCFrame.tab_mvf + 8 status bytes
// (byte 0 != 0: the owner failed on the picture).  Wires:
//   RCCL     one ncclGroup of four ncclBroadcast per picture on a stream of the transport's own: planes AND motion field are device
//            memory (the motion field is staged through a pinned host buffer on both sides), no TCP anywhere on the data path.  librccl
//            is loaded with dlopen when such a transport is created: libohevc_hip.so itself does not depend on it.
//   sockets  one TCP connection per pair of ranks and a worker thread that executes the queued broadcasts in order; host-staged.  For
//            ranks that share a GPU (RCCL refuses that) and for the tests.
// Every rank issues the same sequence of broadcasts (decoding order; root = index % world), so neither wire can deadlock.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include <arpa/inet.h>
#include <dlfcn.h>
#include <errno.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <unistd.h>
#include "common.hpp"
#include "ohevc_frames.h"

using namespace ohevc;

namespace {

// ---- the few RCCL entry points (rccl.h: ncclGetUniqueId :187, ncclCommInitRank :220, ncclCommDestroy :260, ncclBroadcast :591, ncclGroupStart/End :923)
struct NcclUniqueId { char internal[128]; };
typedef void *NcclComm;
constexpr int kNcclUint8 = 1;
struct Rccl {
    void *lib = nullptr;
    int (*GetUniqueId)(NcclUniqueId *) = nullptr;
    int (*CommInitRank)(NcclComm *, int, NcclUniqueId, int) = nullptr;
    int (*CommDestroy)(NcclComm) = nullptr;
    int (*Broadcast)(const void *, void *, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    int (*GroupStart)(void) = nullptr;
    int (*GroupEnd)(void) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool load()
    {
        for (const char *name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) if ((lib = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
        if (!lib) return false;
        GetUniqueId = reinterpret_cast<decltype(GetUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
        CommInitRank = reinterpret_cast<decltype(CommInitRank)>(dlsym(lib, "ncclCommInitRank"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
        Broadcast = reinterpret_cast<decltype(Broadcast)>(dlsym(lib, "ncclBroadcast"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(dlsym(lib, "ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(dlsym(lib, "ncclGroupEnd"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
        return GetUniqueId && CommInitRank && CommDestroy && Broadcast && GroupStart && GroupEnd && GetErrorString;
    }
};

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// ---- sockets wire: queued broadcasts executed in order by one thread
struct SockOp { unsigned char *buf; size_t bytes; int root; bool done; };
struct SockWire {
    int rank = 0, world = 1, timeout_s = 60;
    std::vector<int> fd;                   // fd[peer], -1 for self
    int listen_fd = -1;
    std::thread worker;
    std::mutex m;
    std::condition_variable cv;
    std::deque<SockOp *> queue;
    bool stop = false, broken = false;

    bool io_all(int f, unsigned char *p, size_t n, bool wr)
    {
        const double t0 = now_s();
        while (n) {
            pollfd pf = { f, (short)(wr ? POLLOUT : POLLIN), 0 };
            const int pr = poll(&pf, 1, 200);
            if (pr < 0 && errno != EINTR) return false;
            if (pr <= 0) { if (now_s() - t0 > timeout_s) return false; continue; }
            const ssize_t k = wr ? send(f, p, n, MSG_NOSIGNAL) : recv(f, p, n, 0);
            if (k < 0 && (errno == EINTR || errno == EAGAIN)) continue;
            if (k <= 0) return false;
            p += k; n -= (size_t)k;
        }
        return true;
    }
    void run()
    {
        for (;;) {
            SockOp *op;
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return stop || !queue.empty(); });
                if (queue.empty()) return;
                op = queue.front();
            }
            bool ok = !broken;
            if (ok) {
                if (op->root == rank) { for (int p = 0; p < world && ok; p++) if (p != rank) ok = io_all(fd[p], op->buf, op->bytes, true); }
                else ok = io_all(fd[op->root], op->buf, op->bytes, false);
            }
            {
                std::lock_guard<std::mutex> g(m);
                if (!ok) broken = true;
                op->done = true;
                queue.pop_front();
            }
            cv.notify_all();
        }
    }
    bool connect_all(const char *rendezvous)
    {
        std::string host(rendezvous ? rendezvous : "127.0.0.1:29700");
        int port = 29700;
        const size_t colon = host.rfind(':');
        if (colon != std::string::npos) { port = atoi(host.c_str() + colon + 1); host.resize(colon); }
        fd.assign((size_t)world, -1);
        listen_fd = socket(AF_INET, SOCK_STREAM, 0);
        if (listen_fd < 0) return false;
        int one = 1;
        setsockopt(listen_fd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
        sockaddr_in a = {};
        a.sin_family = AF_INET; a.sin_port = htons((uint16_t)(port + rank)); a.sin_addr.s_addr = htonl(INADDR_ANY);
        if (bind(listen_fd, reinterpret_cast<sockaddr *>(&a), sizeof(a)) != 0 || listen(listen_fd, world) != 0) return false;
        hostent *he = gethostbyname(host.c_str());
        if (!he) return false;
        // a rank connects to every lower rank and accepts from every higher one
        for (int p = 0; p < rank; p++) {
            const double t0 = now_s();
            for (;;) {
                const int s = socket(AF_INET, SOCK_STREAM, 0);
                sockaddr_in b = {};
                b.sin_family = AF_INET; b.sin_port = htons((uint16_t)(port + p));
                memcpy(&b.sin_addr, he->h_addr_list[0], sizeof(b.sin_addr));
                if (connect(s, reinterpret_cast<sockaddr *>(&b), sizeof(b)) == 0) {
                    setsockopt(s, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
                    const int32_t me = rank;
                    if (send(s, &me, sizeof(me), MSG_NOSIGNAL) != (ssize_t)sizeof(me)) { close(s); return false; }
                    fd[(size_t)p] = s;
                    break;
                }
                close(s);
                if (now_s() - t0 > timeout_s) return false;
                usleep(20000);
            }
        }
        for (int k = rank + 1; k < world; k++) {
            pollfd pf = { listen_fd, POLLIN, 0 };
            if (poll(&pf, 1, timeout_s * 1000) <= 0) return false;
            const int s = accept(listen_fd, nullptr, nullptr);
            if (s < 0) return false;
            setsockopt(s, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
            int32_t who = -1;
            if (recv(s, &who, sizeof(who), MSG_WAITALL) != (ssize_t)sizeof(who) || who <= rank || who >= world || fd[(size_t)who] >= 0) { close(s); return false; }
            fd[(size_t)who] = s;
        }
        worker = std::thread([this] { run(); });
        return true;
    }
    void post(SockOp *op)
    {
        { std::lock_guard<std::mutex> g(m); op->done = false; queue.push_back(op); }
        cv.notify_all();
    }
    bool wait(SockOp *op)
    {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return op->done; });
        return !broken;
    }
    void shutdown()
    {
        { std::lock_guard<std::mutex> g(m); stop = true; }
        cv.notify_all();
        if (worker.joinable()) worker.join();
        for (int &f : fd) if (f >= 0) { close(f); f = -1; }
        if (listen_fd >= 0) { close(listen_fd); listen_fd = -1; }
    }
};

struct Msg {
    int index = -1, root = 0;
    bool outgoing = false, planes_done = false, motion_done = false;
    size_t plane_bytes[3] = {0, 0, 0}, mvf_bytes = 0;
    void *d_plane[3] = {nullptr, nullptr, nullptr};
    void *d_mvf = nullptr;                     // RCCL: the motion-field message in device memory
    unsigned char *h_mvf = nullptr;            // pinned: the motion-field message, mvf_bytes + 8 status bytes
    unsigned char *h_plane[3] = {nullptr, nullptr, nullptr};      // sockets: host staging of the planes
    hipEvent_t ev = nullptr;                   // RCCL: the picture's collectives (and the copy into h_mvf) are done
    SockOp op[4];                              // sockets: planes 0..2, motion field
};

}  // namespace

struct ohevc_frames_transport {
    int rank = 0, world = 1, device = 0, wire = 0, timeout_s = 60;
    ohhip_frames_mode mode = {};
    ohevc_frames_stats stats = {};
    std::string rendezvous;
    Rccl rccl;
    NcclComm comm = nullptr;
    hipStream_t stream = nullptr;
    SockWire sock;
    std::map<int, Msg *> pending;              // subscribed pictures by decoding-order index
    std::deque<Msg *> outgoing;                // published pictures whose transfers may still be in flight
    std::vector<std::pair<size_t, void *>> dev_pool, host_pool;
    bool broken = false;

    void *dev_alloc(size_t n)
    {
        for (size_t i = 0; i < dev_pool.size(); i++)
            if (dev_pool[i].first == n) { void *p = dev_pool[i].second; dev_pool[i] = dev_pool.back(); dev_pool.pop_back(); return p; }
        void *p = nullptr;
        return hipMalloc(&p, n ? n : 1) == hipSuccess ? p : nullptr;
    }
    void dev_free(size_t n, void *p) { if (p) dev_pool.emplace_back(n, p); }
    unsigned char *host_alloc(size_t n)
    {
        for (size_t i = 0; i < host_pool.size(); i++)
            if (host_pool[i].first == n) { void *p = host_pool[i].second; host_pool[i] = host_pool.back(); host_pool.pop_back(); return static_cast<unsigned char *>(p); }
        void *p = nullptr;
        return hipHostMalloc(&p, n ? n : 1, hipHostMallocDefault) == hipSuccess ? static_cast<unsigned char *>(p) : nullptr;
    }
    void host_free(size_t n, void *p) { if (p) host_pool.emplace_back(n, p); }

    bool wait_event(hipEvent_t e)
    {
        const double t0 = now_s();
        for (;;) {
            const hipError_t q = hipEventQuery(e);
            if (q == hipSuccess) return true;
            if (q != hipErrorNotReady) { (void)hipGetLastError(); return false; }
            if (now_s() - t0 > timeout_s) return false;
            usleep(50);
        }
    }
    // the message's transfers are complete on this rank (outgoing: sent; incoming: arrived)
    bool complete(Msg *m)
    {
        if (wire == OHEVC_FRAMES_WIRE_RCCL) return wait_event(m->ev);
        bool ok = true;
        for (SockOp &o : m->op) if (o.buf) ok = sock.wait(&o) && ok;
        return ok;
    }
    void recycle(Msg *m)
    {
        for (int c = 0; c < 3; c++) { dev_free(m->plane_bytes[c], m->d_plane[c]); host_free(m->plane_bytes[c], m->h_plane[c]); m->d_plane[c] = nullptr; m->h_plane[c] = nullptr; }
        dev_free(m->mvf_bytes + 8, m->d_mvf); m->d_mvf = nullptr;
        host_free(m->mvf_bytes + 8, m->h_mvf); m->h_mvf = nullptr;
        if (m->ev) { (void)hipEventDestroy(m->ev); m->ev = nullptr; }
        delete m;
    }
    // staging of one picture: plane sizes from the store, buffers from the pools
    Msg *stage(int index, ohevc_ctx *ctx, int slot, size_t mvf_bytes, int root)
    {
        ohevc_plane pl[3];
        if (ohevc_pic_planes(ctx, slot, pl) != OHEVC_OK) return nullptr;
        Msg *m = new Msg();
        m->index = index; m->root = root; m->mvf_bytes = mvf_bytes;
        bool ok = true;
        for (int c = 0; c < 3; c++) {
            m->plane_bytes[c] = (size_t)pl[c].stride * pl[c].height;
            ok = ok && (m->d_plane[c] = dev_alloc(m->plane_bytes[c])) != nullptr;
            if (wire == OHEVC_FRAMES_WIRE_SOCKETS) ok = ok && (m->h_plane[c] = host_alloc(m->plane_bytes[c])) != nullptr;
        }
        ok = ok && (m->h_mvf = host_alloc(mvf_bytes + 8)) != nullptr;
        if (wire == OHEVC_FRAMES_WIRE_RCCL) {
            ok = ok && (m->d_mvf = dev_alloc(mvf_bytes + 8)) != nullptr;
            ok = ok && hipEventCreateWithFlags(&m->ev, hipEventDisableTiming) == hipSuccess;
        }
        if (!ok) { set_error("frames transport: staging of picture %d failed (out of memory?)", index); recycle(m); return nullptr; }
        return m;
    }
    // issue the picture's broadcasts (same sequence on every rank)
    bool post(Msg *m)
    {
        stats.bytes += (long long)(m->plane_bytes[0] + m->plane_bytes[1] + m->plane_bytes[2] + m->mvf_bytes + 8);
        if (wire == OHEVC_FRAMES_WIRE_RCCL) {
            int rc = rccl.GroupStart();
            for (int c = 0; c < 3 && rc == 0; c++) rc = rccl.Broadcast(m->d_plane[c], m->d_plane[c], m->plane_bytes[c], kNcclUint8, m->root, comm, stream);
            if (rc == 0) rc = rccl.Broadcast(m->d_mvf, m->d_mvf, m->mvf_bytes + 8, kNcclUint8, m->root, comm, stream);
            const int rc2 = rccl.GroupEnd();
            if (rc != 0 || rc2 != 0) { set_error("frames transport: ncclBroadcast failed: %s", rccl.GetErrorString(rc ? rc : rc2)); return false; }
            if (!m->outgoing && hipMemcpyAsync(m->h_mvf, m->d_mvf, m->mvf_bytes + 8, hipMemcpyDeviceToHost, stream) != hipSuccess) return false;
            return hipEventRecord(m->ev, stream) == hipSuccess;
        }
        for (int c = 0; c < 3; c++) { m->op[c] = SockOp{ m->h_plane[c], m->plane_bytes[c], m->root, false }; sock.post(&m->op[c]); }
        m->op[3] = SockOp{ m->h_mvf, m->mvf_bytes + 8, m->root, false };
        sock.post(&m->op[3]);
        return true;
    }
    void reap_outgoing(bool all)
    {
        while (!outgoing.empty()) {
            Msg *m = outgoing.front();
            if (!all) {
                bool done;
                if (wire == OHEVC_FRAMES_WIRE_RCCL) done = hipEventQuery(m->ev) == hipSuccess;
                else { std::lock_guard<std::mutex> g(sock.m); done = m->op[0].done && m->op[1].done && m->op[2].done && m->op[3].done; }
                if (!done) { (void)hipGetLastError(); break; }
            } else if (!complete(m)) {
                broken = true;
            }
            outgoing.pop_front();
            recycle(m);
        }
    }
};

// ------------------------------------------------------------------ the four callbacks (+ release)
static int cb_publish(void *user, int index, ohevc_ctx *ctx, int slot, const void *mvf, size_t mvf_bytes, int failed)
{
    ohevc_frames_transport *t = static_cast<ohevc_frames_transport *>(user);
    if (t->broken) return -1;
    (void)hipSetDevice(t->device);
    t->reap_outgoing(false);
    Msg *m = t->stage(index, ctx, slot, mvf_bytes, t->rank);
    if (!m) { t->broken = true; return -1; }
    m->outgoing = true;
    memset(m->h_mvf + mvf_bytes, 0, 8);
    if (failed || !mvf) {
        m->h_mvf[mvf_bytes] = 1;                              // the error mark; the payload is whatever the buffers hold
        t->stats.failed++;
    } else {
        memcpy(m->h_mvf, mvf, mvf_bytes);
        for (int c = 0; c < 3; c++) {
            if (ohevc_pic_export(ctx, slot, c, m->d_plane[c], m->plane_bytes[c]) != OHEVC_OK) { m->h_mvf[mvf_bytes] = 1; t->stats.failed++; break; }
            if (t->wire == OHEVC_FRAMES_WIRE_SOCKETS && hipMemcpy(m->h_plane[c], m->d_plane[c], m->plane_bytes[c], hipMemcpyDeviceToHost) != hipSuccess) { m->h_mvf[mvf_bytes] = 1; break; }
        }
    }
    if (t->wire == OHEVC_FRAMES_WIRE_RCCL && hipMemcpyAsync(m->d_mvf, m->h_mvf, mvf_bytes + 8, hipMemcpyHostToDevice, t->stream) != hipSuccess) { t->recycle(m); t->broken = true; return -1; }
    if (!t->post(m)) { t->recycle(m); t->broken = true; return -1; }
    t->outgoing.push_back(m);
    t->stats.published++;
    return 0;
}

static int cb_subscribe(void *user, int index, ohevc_ctx *ctx, int slot, size_t mvf_bytes)
{
    ohevc_frames_transport *t = static_cast<ohevc_frames_transport *>(user);
    if (t->broken) return -1;
    (void)hipSetDevice(t->device);
    Msg *m = t->stage(index, ctx, slot, mvf_bytes, index % t->world);
    if (!m) { t->broken = true; return -1; }
    if (!t->post(m)) { t->recycle(m); t->broken = true; return -1; }
    if (t->pending.count(index)) { t->complete(t->pending[index]); t->recycle(t->pending[index]); }
    t->pending[index] = m;
    t->stats.subscribed++;
    return 0;
}

static Msg *arrived(ohevc_frames_transport *t, int index)
{
    auto it = t->pending.find(index);
    if (it == t->pending.end()) { set_error("frames transport: picture %d was never subscribed to", index); return nullptr; }
    Msg *m = it->second;
    if (!t->complete(m)) { set_error("frames transport: picture %d did not arrive from rank %d within %d s", index, m->root, t->timeout_s); t->broken = true; return nullptr; }
    if (m->h_mvf[m->mvf_bytes] != 0) { set_error("frames transport: picture %d: its owner (rank %d) reported a decoding failure", index, m->root); return nullptr; }
    return m;
}

static void drop_if_consumed(ohevc_frames_transport *t, Msg *m)
{
    if (m->planes_done && m->motion_done) { t->pending.erase(m->index); t->recycle(m); }
}

static int cb_await_motion(void *user, int index, void *mvf, size_t mvf_bytes)
{
    ohevc_frames_transport *t = static_cast<ohevc_frames_transport *>(user);
    Msg *m = arrived(t, index);
    if (!m || mvf_bytes != m->mvf_bytes) return -1;
    memcpy(mvf, m->h_mvf, mvf_bytes);
    m->motion_done = true;
    t->stats.awaited_motion++;
    drop_if_consumed(t, m);
    return 0;
}

static int cb_await_planes(void *user, int index, ohevc_ctx *ctx, int slot)
{
    ohevc_frames_transport *t = static_cast<ohevc_frames_transport *>(user);
    Msg *m = arrived(t, index);
    if (!m) return -1;
    (void)hipSetDevice(t->device);
    for (int c = 0; c < 3; c++) {
        if (t->wire == OHEVC_FRAMES_WIRE_SOCKETS && hipMemcpy(m->d_plane[c], m->h_plane[c], m->plane_bytes[c], hipMemcpyHostToDevice) != hipSuccess) return -1;
        if (ohevc_pic_import(ctx, slot, c, m->d_plane[c], m->plane_bytes[c]) != OHEVC_OK) return -1;
    }
    for (int c = 0; c < 3; c++) {                              // the planes are in the store now; the motion field may still be asked for
        t->dev_free(m->plane_bytes[c], m->d_plane[c]); t->host_free(m->plane_bytes[c], m->h_plane[c]);
        m->d_plane[c] = nullptr; m->h_plane[c] = nullptr;
    }
    m->planes_done = true;
    t->stats.awaited_planes++;
    drop_if_consumed(t, m);
    return 0;
}

static int cb_release(void *user, int index)
{
    ohevc_frames_transport *t = static_cast<ohevc_frames_transport *>(user);
    auto it = t->pending.find(index);
    if (it == t->pending.end()) return 0;
    Msg *m = it->second;
    const bool ok = t->complete(m);
    t->pending.erase(it);
    t->recycle(m);
    t->stats.released++;
    return ok ? 0 : -1;
}

// ------------------------------------------------------------------ life cycle
extern "C" int ohevc_frames_transport_create(ohevc_frames_transport **out, int rank, int world, int device, int wire, const char *rendezvous, int timeout_s)
{
    OHEVC_REQUIRE(out != nullptr && world >= 1 && rank >= 0 && rank < world, "rank / world");
    OHEVC_REQUIRE(wire == OHEVC_FRAMES_WIRE_RCCL || wire == OHEVC_FRAMES_WIRE_SOCKETS, "unknown wire");
    ohevc_frames_transport *t = new ohevc_frames_transport();
    t->rank = rank; t->world = world; t->device = device; t->wire = wire; t->timeout_s = timeout_s > 0 ? timeout_s : 60;
    t->rendezvous = rendezvous ? rendezvous : "";
    t->mode = ohhip_frames_mode{ rank, world, t, cb_publish, cb_subscribe, cb_await_motion, cb_await_planes, cb_release };
    auto fail = [&](int rc) { ohevc_frames_transport_destroy(t); return rc; };
    if (hipSetDevice(device) != hipSuccess) { set_error("frames transport: no device %d", device); return fail(OHEVC_ERR_NODEV); }
    if (wire == OHEVC_FRAMES_WIRE_SOCKETS) {
        t->sock.rank = rank; t->sock.world = world; t->sock.timeout_s = t->timeout_s;
        if (world > 1 && !t->sock.connect_all(rendezvous)) { set_error("frames transport: connecting the ranks through %s failed", rendezvous ? rendezvous : "(default)"); return fail(OHEVC_ERR_STATE); }
        if (world == 1) t->sock.worker = std::thread([t] { t->sock.run(); });
        *out = t;
        return OHEVC_OK;
    }
    OHEVC_REQUIRE(rendezvous != nullptr && rendezvous[0], "the RCCL wire needs a rendezvous file path");
    if (!t->rccl.load()) { set_error("frames transport: librccl.so could not be loaded: %s", dlerror()); return fail(OHEVC_ERR_STATE); }
    if (hipStreamCreateWithFlags(&t->stream, hipStreamNonBlocking) != hipSuccess) return fail(OHEVC_ERR_HIP);
    NcclUniqueId id;
    memset(&id, 0, sizeof(id));
    if (rank == 0) {                                           // hand the id to the others through a file, written whole or not at all
        if (t->rccl.GetUniqueId(&id) != 0) { set_error("frames transport: ncclGetUniqueId failed"); return fail(OHEVC_ERR_STATE); }
        const std::string tmp = t->rendezvous + ".tmp";
        FILE *f = fopen(tmp.c_str(), "wb");
        if (!f || fwrite(&id, sizeof(id), 1, f) != 1) { if (f) fclose(f); set_error("frames transport: cannot write %s", tmp.c_str()); return fail(OHEVC_ERR_STATE); }
        fclose(f);
        if (rename(tmp.c_str(), t->rendezvous.c_str()) != 0) { set_error("frames transport: cannot publish %s", t->rendezvous.c_str()); return fail(OHEVC_ERR_STATE); }
    } else {
        const double t0 = now_s();
        for (;;) {
            FILE *f = fopen(t->rendezvous.c_str(), "rb");
            const bool ok = f && fread(&id, sizeof(id), 1, f) == 1;
            if (f) fclose(f);
            if (ok) break;
            if (now_s() - t0 > t->timeout_s) { set_error("frames transport: rank 0 never published %s", t->rendezvous.c_str()); return fail(OHEVC_ERR_STATE); }
            usleep(20000);
        }
    }
    const int rc = t->rccl.CommInitRank(&t->comm, world, id, rank);
    if (rc != 0) { set_error("frames transport: ncclCommInitRank failed: %s", t->rccl.GetErrorString(rc)); t->comm = nullptr; return fail(OHEVC_ERR_STATE); }
    *out = t;
    return OHEVC_OK;
}

extern "C" const ohhip_frames_mode *ohevc_frames_transport_mode(ohevc_frames_transport *t) { return t ? &t->mode : nullptr; }

extern "C" int ohevc_frames_transport_finish(ohevc_frames_transport *t)
{
    OHEVC_REQUIRE(t != nullptr, "null transport");
    (void)hipSetDevice(t->device);
    t->reap_outgoing(true);
    for (auto &kv : t->pending) { if (!t->complete(kv.second)) t->broken = true; t->recycle(kv.second); }
    t->pending.clear();
    if (t->stream) (void)hipStreamSynchronize(t->stream);
    if (t->broken) { set_error("frames transport: a transfer failed or timed out"); return OHEVC_ERR_STATE; }
    return OHEVC_OK;
}

extern "C" void ohevc_frames_transport_destroy(ohevc_frames_transport *t)
{
    if (!t) return;
    (void)hipSetDevice(t->device);
    if (t->wire == OHEVC_FRAMES_WIRE_SOCKETS) t->sock.shutdown();
    for (auto &kv : t->pending) t->recycle(kv.second);
    for (Msg *m : t->outgoing) t->recycle(m);
    if (t->stream) (void)hipStreamSynchronize(t->stream);
    if (t->comm) t->rccl.CommDestroy(t->comm);
    if (t->stream) (void)hipStreamDestroy(t->stream);
    for (auto &b : t->dev_pool) (void)hipFree(b.second);
    for (auto &b : t->host_pool) (void)hipHostFree(b.second);
    if (t->wire == OHEVC_FRAMES_WIRE_RCCL && t->rank == 0 && !t->rendezvous.empty()) (void)remove(t->rendezvous.c_str());
    if (t->rccl.lib) dlclose(t->rccl.lib);
    delete t;
}

extern "C" int ohevc_frames_transport_stats(ohevc_frames_transport *t, ohevc_frames_stats *out)
{
    OHEVC_REQUIRE(t != nullptr && out != nullptr, "null argument");
    *out = t->stats;
    return OHEVC_OK;
}
