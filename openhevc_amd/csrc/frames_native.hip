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
#include <memory>
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
#include "ohevc_debug.h"

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
    int (*CommCount)(NcclComm, int *) = nullptr;
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
        CommCount = reinterpret_cast<decltype(CommCount)>(dlsym(lib, "ncclCommCount"));      // rccl.h: ncclCommCount(comm, int *count)
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
        // The ranks of this wire share one host (ohevc_frames.h): listen on the rendezvous host's address only - loopback by default - not on
        // every interface, and let a peer in only if it presents the run's token (OHEVC_FRAMES_TOKEN when the launcher sets one for all
        // ranks, else derived from the rendezvous string: it keeps two runs on one machine apart, it is not a secret).
        hostent *he = gethostbyname(host.c_str());
        if (!he) return false;
        sockaddr_in a = {};
        a.sin_family = AF_INET; a.sin_port = htons((uint16_t)(port + rank));
        memcpy(&a.sin_addr, he->h_addr_list[0], sizeof(a.sin_addr));
        if (bind(listen_fd, reinterpret_cast<sockaddr *>(&a), sizeof(a)) != 0 || listen(listen_fd, world) != 0) return false;
        uint64_t token = 1469598103934665603ull;
        {
            const char *env = ohevc::config().frames_token;
            const std::string src = env && env[0] ? std::string(env) : std::string(rendezvous ? rendezvous : "127.0.0.1:29700");
            for (unsigned char ch : src) token = (token ^ ch) * 1099511628211ull;
        }
        struct Hello { int32_t rank; uint32_t pad; uint64_t token; };
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
                    const Hello me = { rank, 0u, token };
                    if (send(s, &me, sizeof(me), MSG_NOSIGNAL) != (ssize_t)sizeof(me)) { close(s); return false; }
                    fd[(size_t)p] = s;
                    break;
                }
                close(s);
                if (now_s() - t0 > timeout_s) return false;
                usleep(20000);
            }
        }
        const double t_accept = now_s();                   // one deadline for the whole rendezvous: a local peer that keeps connecting with bad
        for (int k = rank + 1; k < world;) {               // hellos must not hold this rank here for ever
            const double t0 = t_accept;
            if (now_s() - t0 > timeout_s) return false;
            pollfd pf = { listen_fd, POLLIN, 0 };
            if (poll(&pf, 1, timeout_s * 1000) <= 0) return false;
            const int s = accept(listen_fd, nullptr, nullptr);
            if (s < 0) return false;
            setsockopt(s, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
            // a connection that says nothing, the wrong thing or the wrong token is dropped and the slot stays open for the real peer
            Hello who = { -1, 0u, 0ull };
            pollfd hf = { s, POLLIN, 0 };
            const bool talks = poll(&hf, 1, 2000) > 0 && recv(s, &who, sizeof(who), MSG_WAITALL) == (ssize_t)sizeof(who);
            if (!talks || who.token != token || who.rank <= rank || who.rank >= world || fd[(size_t)who.rank] >= 0) {
                close(s);
                if (now_s() - t0 > timeout_s) return false;
                continue;
            }
            fd[(size_t)who.rank] = s;
            k++;
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

// A picture travels as 1 + nb messages, in this order on every rank: the motion-field message (HEVCFrame.tab_mvf + 8 status bytes - what a
// dependent picture needs FIRST, when its parse starts), then nb bands of CTU rows, each band = the three planes' row ranges.  The reference's
// frame threads publish a picture CTB row by CTB row (ff_thread_report_progress, hevc.c:2934-2937) and a dependent picture waits for the rows its
// motion vectors reach (hevc_await_progress, hevc.c:1951-1958: y0 + mv.y + nPbH + 9); here a band is the unit of both.
constexpr int kMaxBands = 8;
struct Msg {
    int index = -1, root = 0;
    bool outgoing = false, planes_done = false, motion_done = false;
    size_t plane_bytes[3] = {0, 0, 0}, mvf_bytes = 0;
    int nb = 1;                                // bands
    int row0[3][kMaxBands + 1] = {};           // band b of plane c = rows [row0[c][b], row0[c][b + 1])
    int stride[3] = {0, 0, 0};
    int luma_rows = 0;
    int bands_imported = 0;                    // subscriber: bands 0 .. bands_imported - 1 are in the picture store
    void *d_plane[3] = {nullptr, nullptr, nullptr};
    void *d_mvf = nullptr;                     // RCCL: the motion-field message in device memory
    unsigned char *h_mvf = nullptr;            // pinned: the motion-field message, mvf_bytes + 8 status bytes
    unsigned char *h_plane[3] = {nullptr, nullptr, nullptr};      // sockets: host staging of the planes
    hipEvent_t ev_mvf = nullptr;               // RCCL: the motion-field broadcast (and, on a subscriber, its copy into h_mvf) is done
    hipEvent_t ev_band[kMaxBands] = {};        // RCCL: band b's three broadcasts are done
    SockOp op_mvf;                             // sockets
    SockOp op_band[kMaxBands][3];
    size_t band_off(int c, int b) const { return (size_t)row0[c][b] * (size_t)stride[c]; }
    size_t band_bytes(int c, int b) const { return (size_t)(row0[c][b + 1] - row0[c][b]) * (size_t)stride[c]; }
    int band_of_luma_row(int row) const { int b = 0; while (b + 1 < nb && row >= row0[0][b + 1]) b++; return b; }
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
    int max_bands = kMaxBands;                 // 1: whole pictures (OHEVC_FRAMES_BANDS / ohevc_frames_transport_set_bands)

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
    // band b (b < 0: the motion-field message) of the picture is complete on this rank (outgoing: sent; incoming: arrived)
    bool part_complete(Msg *m, int b)
    {
        if (wire == OHEVC_FRAMES_WIRE_RCCL) return wait_event(b < 0 ? m->ev_mvf : m->ev_band[b]);
        if (b < 0) return sock.wait(&m->op_mvf);
        bool ok = true;
        for (int c = 0; c < 3; c++) if (m->op_band[b][c].buf) ok = sock.wait(&m->op_band[b][c]) && ok;
        return ok;
    }
    // ... all of it
    bool complete(Msg *m)
    {
        bool ok = part_complete(m, -1);
        for (int b = 0; b < m->nb; b++) ok = part_complete(m, b) && ok;
        return ok;
    }
    bool is_complete_now(Msg *m)
    {
        if (wire == OHEVC_FRAMES_WIRE_RCCL) {
            if (hipEventQuery(m->ev_band[m->nb - 1]) == hipSuccess) return true;
            (void)hipGetLastError();
            return false;
        }
        std::lock_guard<std::mutex> g(sock.m);
        bool done = m->op_mvf.done;
        for (int c = 0; c < 3; c++) done = done && (!m->op_band[m->nb - 1][c].buf || m->op_band[m->nb - 1][c].done);
        return done;                                               // (the wire is in order: the last band done = everything done)
    }
    void recycle(Msg *m)
    {
        for (int c = 0; c < 3; c++) { dev_free(m->plane_bytes[c], m->d_plane[c]); host_free(m->plane_bytes[c], m->h_plane[c]); m->d_plane[c] = nullptr; m->h_plane[c] = nullptr; }
        dev_free(m->mvf_bytes + 8, m->d_mvf); m->d_mvf = nullptr;
        host_free(m->mvf_bytes + 8, m->h_mvf); m->h_mvf = nullptr;
        if (m->ev_mvf) { (void)hipEventDestroy(m->ev_mvf); m->ev_mvf = nullptr; }
        for (hipEvent_t &e : m->ev_band) if (e) { (void)hipEventDestroy(e); e = nullptr; }
        delete m;
    }
    // staging of one picture: plane sizes and the band grid from the store, buffers from the pools
    Msg *stage(int index, ohevc_ctx *ctx, int slot, size_t mvf_bytes, int root)
    {
        ohevc_plane pl[3];
        if (ohevc_pic_planes(ctx, slot, pl) != OHEVC_OK) return nullptr;
        Msg *m = new Msg();
        m->index = index; m->root = root; m->mvf_bytes = mvf_bytes;
        // bands of whole 64-row CTU rows, at most kMaxBands of them (8K: 68 CTU rows -> 8 bands of 9 rows, 12 MB each at 10 bit); max_bands 1 =
        // the whole picture as one band
        const int ctu_rows = std::max(1, (pl[0].height + 63) / 64);
        const int limit = std::max(1, std::min(max_bands, kMaxBands));
        const int per_band = (ctu_rows + limit - 1) / limit;
        m->nb = (ctu_rows + per_band - 1) / per_band;
        m->luma_rows = pl[0].height;
        bool ok = true;
        for (int c = 0; c < 3; c++) {
            m->plane_bytes[c] = (size_t)pl[c].stride * pl[c].height;
            m->stride[c] = pl[c].stride;
            const int vs = (c && pl[0].height > pl[c].height) ? 1 : 0;
            for (int b = 0; b <= m->nb; b++) m->row0[c][b] = b == m->nb ? pl[c].height : std::min(pl[c].height, (b * per_band * 64) >> vs);
            ok = ok && (m->d_plane[c] = dev_alloc(m->plane_bytes[c])) != nullptr;
            if (wire == OHEVC_FRAMES_WIRE_SOCKETS) ok = ok && (m->h_plane[c] = host_alloc(m->plane_bytes[c])) != nullptr;
        }
        ok = ok && (m->h_mvf = host_alloc(mvf_bytes + 8)) != nullptr;
        if (wire == OHEVC_FRAMES_WIRE_RCCL) {
            ok = ok && (m->d_mvf = dev_alloc(mvf_bytes + 8)) != nullptr;
            ok = ok && hipEventCreateWithFlags(&m->ev_mvf, hipEventDisableTiming) == hipSuccess;
            for (int b = 0; b < m->nb; b++) ok = ok && hipEventCreateWithFlags(&m->ev_band[b], hipEventDisableTiming) == hipSuccess;
        }
        if (!ok) { set_error("frames transport: staging of picture %d failed (out of memory?)", index); recycle(m); return nullptr; }
        return m;
    }
    // issue the picture's motion-field message (same sequence on every rank: post_mvf, then post_band 0 .. nb - 1)
    bool post_mvf(Msg *m)
    {
        stats.bytes += (long long)(m->mvf_bytes + 8);
        if (wire == OHEVC_FRAMES_WIRE_RCCL) {
            const int rc = rccl.Broadcast(m->d_mvf, m->d_mvf, m->mvf_bytes + 8, kNcclUint8, m->root, comm, stream);
            if (rc != 0) { set_error("frames transport: ncclBroadcast failed: %s", rccl.GetErrorString(rc)); return false; }
            if (!m->outgoing && hipMemcpyAsync(m->h_mvf, m->d_mvf, m->mvf_bytes + 8, hipMemcpyDeviceToHost, stream) != hipSuccess) return false;
            return hipEventRecord(m->ev_mvf, stream) == hipSuccess;
        }
        m->op_mvf = SockOp{ m->h_mvf, m->mvf_bytes + 8, m->root, false };
        sock.post(&m->op_mvf);
        return true;
    }
    bool post_band(Msg *m, int b)
    {
        for (int c = 0; c < 3; c++) stats.bytes += (long long)m->band_bytes(c, b);
        if (wire == OHEVC_FRAMES_WIRE_RCCL) {
            int rc = rccl.GroupStart();
            for (int c = 0; c < 3 && rc == 0; c++) {
                if (!m->band_bytes(c, b)) continue;
                unsigned char *at = static_cast<unsigned char *>(m->d_plane[c]) + m->band_off(c, b);
                rc = rccl.Broadcast(at, at, m->band_bytes(c, b), kNcclUint8, m->root, comm, stream);
            }
            const int rc2 = rccl.GroupEnd();
            if (rc != 0 || rc2 != 0) { set_error("frames transport: ncclBroadcast failed: %s", rccl.GetErrorString(rc ? rc : rc2)); return false; }
            return hipEventRecord(m->ev_band[b], stream) == hipSuccess;
        }
        for (int c = 0; c < 3; c++) {
            m->op_band[b][c] = SockOp{ m->band_bytes(c, b) ? m->h_plane[c] + m->band_off(c, b) : nullptr, m->band_bytes(c, b), m->root, true };
            if (m->op_band[b][c].buf) sock.post(&m->op_band[b][c]);
        }
        return true;
    }
    void reap_outgoing(bool all)
    {
        while (!outgoing.empty()) {
            Msg *m = outgoing.front();
            if (!all) {
                if (!is_complete_now(m)) break;
            } else if (!complete(m)) {
                broken = true;
            }
            outgoing.pop_front();
            recycle(m);
        }
    }
    // subscriber: bands [bands_imported, upto] of the picture into the store, each as soon as it has arrived
    bool import_bands(Msg *m, ohevc_ctx *ctx, int slot, int upto)
    {
        for (int b = m->bands_imported; b <= upto; b++) {
            if (!part_complete(m, b)) { set_error("frames transport: band %d of picture %d did not arrive from rank %d within %d s", b, m->index, m->root, timeout_s); broken = true; return false; }
            int r0[3], nr[3];
            const void *base[3];
            for (int c = 0; c < 3; c++) {
                r0[c] = m->row0[c][b]; nr[c] = m->row0[c][b + 1] - r0[c]; base[c] = m->d_plane[c];
                if (wire == OHEVC_FRAMES_WIRE_SOCKETS && nr[c] &&
                    hipMemcpy(static_cast<unsigned char *>(m->d_plane[c]) + m->band_off(c, b), m->h_plane[c] + m->band_off(c, b), m->band_bytes(c, b), hipMemcpyHostToDevice) != hipSuccess) return false;
            }
            if (ohevc_pic_import_band(ctx, slot, r0, nr, base, b == 0) != OHEVC_OK) return false;      // three planes, one wait
            m->bands_imported = b + 1;
            stats.bands_imported++;
        }
        return true;
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
    // band b leaves the picture store (and, sockets, the device) while band b - 1 is on the wire.  The export of band 0 is the one that waits
    // for the picture's device work and learns whether it failed: it runs BEFORE the motion-field message, which carries the error mark.
    auto export_band = [&](int b) {
        int r0[3], nr[3];
        void *base[3];
        for (int c = 0; c < 3; c++) { r0[c] = m->row0[c][b]; nr[c] = m->row0[c][b + 1] - r0[c]; base[c] = m->d_plane[c]; }
        if (ohevc_pic_export_band(ctx, slot, r0, nr, base) != OHEVC_OK) return false;                    // three planes, one wait
        for (int c = 0; c < 3; c++)
            if (t->wire == OHEVC_FRAMES_WIRE_SOCKETS && nr[c] &&
                hipMemcpy(m->h_plane[c] + m->band_off(c, b), static_cast<unsigned char *>(m->d_plane[c]) + m->band_off(c, b), m->band_bytes(c, b), hipMemcpyDeviceToHost) != hipSuccess) return false;
        return true;
    };
    bool bad = failed || !mvf;
    if (!bad) {
        memcpy(m->h_mvf, mvf, mvf_bytes);
        bad = !export_band(0);
    }
    if (bad) {
        m->h_mvf[mvf_bytes] = 1;                              // the error mark; the payload is whatever the buffers hold
        t->stats.failed++;
    }
    auto give_up = [&] { t->recycle(m); t->broken = true; return -1; };
    if (t->wire == OHEVC_FRAMES_WIRE_RCCL && hipMemcpyAsync(m->d_mvf, m->h_mvf, mvf_bytes + 8, hipMemcpyHostToDevice, t->stream) != hipSuccess) return give_up();
    if (!t->post_mvf(m)) return give_up();
    for (int b = 0; b < m->nb; b++) {
        if (b && !bad && !export_band(b)) { set_error("frames transport: exporting band %d of picture %d failed", b, index); t->broken = true; }   // (cannot happen after band 0 worked: a device error)
        if (!t->post_band(m, b)) return give_up();             // every rank issues every band, whatever happened to the picture
    }
    t->outgoing.push_back(m);
    t->stats.published++;
    return t->broken ? -1 : 0;
}

static int cb_subscribe(void *user, int index, ohevc_ctx *ctx, int slot, size_t mvf_bytes)
{
    ohevc_frames_transport *t = static_cast<ohevc_frames_transport *>(user);
    if (t->broken) return -1;
    (void)hipSetDevice(t->device);
    Msg *m = t->stage(index, ctx, slot, mvf_bytes, index % t->world);
    if (!m) { t->broken = true; return -1; }
    bool ok = t->post_mvf(m);
    for (int b = 0; b < m->nb && ok; b++) ok = t->post_band(m, b);
    if (!ok) { t->recycle(m); t->broken = true; return -1; }
    if (t->pending.count(index)) { t->complete(t->pending[index]); t->recycle(t->pending[index]); }
    t->pending[index] = m;
    t->stats.subscribed++;
    return 0;
}

// the picture's first message - motion field + status - has arrived and says the owner succeeded
static Msg *arrived(ohevc_frames_transport *t, int index)
{
    auto it = t->pending.find(index);
    if (it == t->pending.end()) { set_error("frames transport: picture %d was never subscribed to", index); return nullptr; }
    Msg *m = it->second;
    if (!t->part_complete(m, -1)) { set_error("frames transport: picture %d did not arrive from rank %d within %d s", index, m->root, t->timeout_s); t->broken = true; return nullptr; }
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

// rows 0 .. last_luma_row of remote picture `index` into picture-store slot `slot` (bands already imported are not touched again); once the
// last band is in, the staging memory goes back to the pools
static int cb_await_rows(void *user, int index, ohevc_ctx *ctx, int slot, int last_luma_row)
{
    ohevc_frames_transport *t = static_cast<ohevc_frames_transport *>(user);
    Msg *m = arrived(t, index);
    if (!m) return -1;
    if (m->planes_done) return 1;
    (void)hipSetDevice(t->device);
    const int upto = last_luma_row < 0 || last_luma_row >= m->luma_rows ? m->nb - 1 : m->band_of_luma_row(last_luma_row);
    if (upto >= m->bands_imported) t->stats.awaited_planes++;
    if (!t->import_bands(m, ctx, slot, upto)) return -1;
    if (m->bands_imported == m->nb) {
        for (int c = 0; c < 3; c++) {                          // the planes are in the store now; the motion field may still be asked for
            t->dev_free(m->plane_bytes[c], m->d_plane[c]); t->host_free(m->plane_bytes[c], m->h_plane[c]);
            m->d_plane[c] = nullptr; m->h_plane[c] = nullptr;
        }
        m->planes_done = true;
        drop_if_consumed(t, m);
        return 1;                                              // the whole picture is in: the caller need not (and must not) ask again
    }
    return 0;
}

static int cb_await_planes(void *user, int index, ohevc_ctx *ctx, int slot) { return cb_await_rows(user, index, ctx, slot, -1) < 0 ? -1 : 0; }

static int cb_release(void *user, int index)
{
    ohevc_frames_transport *t = static_cast<ohevc_frames_transport *>(user);
    auto it = t->pending.find(index);
    if (it == t->pending.end()) return 0;
    Msg *m = it->second;
    const bool ok = t->complete(m);
    t->pending.erase(it);
    t->stats.released++;
    if (!ok) {
        // the transfer timed out: it may still land (a late RCCL write, a socket read in progress), so its buffers must never be handed to
        // another picture - they are leaked on purpose - and the transport is done for
        t->broken = true;
        set_error("frames transport: picture %d was still in flight after %d s when its buffer was released", index, t->timeout_s);
        return -1;
    }
    t->recycle(m);
    return 0;
}

// ------------------------------------------------------------------ RCCL rendezvous through the file system
// ncclCommInitRank needs the same ncclUniqueId on every rank and hangs - without a timeout - when a rank brings another one.  A file left
// behind by a crashed run (or a second run reusing the path before rank 0 has replaced it) must therefore never be mistaken for this run's:
//   rank r > 0   writes <path>.ready.<r> = a nonce of its own (16 random bytes), then polls <path> until it carries THAT nonce in slot r,
//                answers with <path>.ack.<r> = nonce ^ rank 0's run nonce, and only then calls ncclCommInitRank;
//   rank 0       removes a stale <path>, collects the nonces of all ready files, writes <path> = magic, run nonce, the nonces it saw, the id
//                (whole or not at all: tmp + rename) and waits for every ack to match; while it waits it re-reads the ready files - a ready
//                file it read too early (stale, from an earlier run) is replaced by its rank a moment later - and rewrites <path> if one
//                changed.  Every wait is bounded by timeout_s: a missing peer is an error, not a hang.
struct Nonce { unsigned char b[16]; };
static bool read_file(const std::string &path, void *dst, size_t n)
{
    FILE *f = fopen(path.c_str(), "rb");
    const bool ok = f && fread(dst, n, 1, f) == 1;
    if (f) fclose(f);
    return ok;
}
static bool write_file_atomically(const std::string &path, const void *src, size_t n)
{
    const std::string tmp = path + ".tmp" + std::to_string((long)getpid());
    FILE *f = fopen(tmp.c_str(), "wb");
    const bool ok = f && fwrite(src, n, 1, f) == 1;
    if (f) fclose(f);
    if (!ok || rename(tmp.c_str(), path.c_str()) != 0) { (void)remove(tmp.c_str()); return false; }
    return true;
}
static Nonce fresh_nonce()
{
    Nonce n;
    memset(&n, 0, sizeof(n));
    FILE *f = fopen("/dev/urandom", "rb");
    if (!f || fread(n.b, sizeof(n.b), 1, f) != 1) {            // no entropy source: pid, time and an address still tell two runs apart
        const unsigned long long v[2] = { (unsigned long long)getpid() * 0x9e3779b97f4a7c15ull ^ (unsigned long long)(now_s() * 1e9), (unsigned long long)(uintptr_t)&n };
        memcpy(n.b, v, sizeof(n.b));
    }
    if (f) fclose(f);
    return n;
}
static constexpr unsigned kRendezvousMagic = 0x6f687276u;
// `id` is filled by make_id on rank 0 (after the stale file is gone) and received on the others
template <class MakeId>
static bool rendezvous_exchange(const std::string &path, int rank, int world, int timeout_s, NcclUniqueId *id, MakeId make_id)
{
    const size_t file_bytes = 8 + 16 + 16 * (size_t)world + sizeof(NcclUniqueId);
    std::vector<unsigned char> blob(file_bytes);
    const double t0 = now_s();
    auto expired = [&] { return now_s() - t0 > timeout_s; };
    if (rank != 0) {
        const Nonce mine = fresh_nonce();
        if (!write_file_atomically(path + ".ready." + std::to_string(rank), &mine, sizeof(mine))) { set_error("frames transport: cannot write %s.ready.%d", path.c_str(), rank); return false; }
        for (;;) {
            unsigned magic = 0;
            if (read_file(path, blob.data(), file_bytes) && (memcpy(&magic, blob.data(), 4), magic == kRendezvousMagic) &&
                memcmp(blob.data() + 8 + 16 + 16 * (size_t)rank, mine.b, 16) == 0)
                break;
            if (expired()) { set_error("frames transport: rank 0 never published an id for this run in %s", path.c_str()); return false; }
            usleep(20000);
        }
        Nonce ack;
        for (int i = 0; i < 16; i++) ack.b[i] = mine.b[i] ^ blob[8 + (size_t)i];
        memcpy(id, blob.data() + 8 + 16 + 16 * (size_t)world, sizeof(*id));
        if (!write_file_atomically(path + ".ack." + std::to_string(rank), &ack, sizeof(ack))) { set_error("frames transport: cannot write %s.ack.%d", path.c_str(), rank); return false; }
        return true;
    }
    (void)remove(path.c_str());                                // whatever an earlier run left behind
    if (!make_id(id)) { set_error("frames transport: ncclGetUniqueId failed"); return false; }
    const Nonce run = fresh_nonce();
    std::vector<Nonce> seen((size_t)world), written((size_t)world);
    memset(seen.data(), 0, sizeof(Nonce) * (size_t)world);
    memset(written.data(), 0xff, sizeof(Nonce) * (size_t)world);
    bool published = false;
    for (;;) {
        bool all_ready = true, all_acked = published;
        for (int r = 1; r < world; r++) {
            Nonce n;
            if (!read_file(path + ".ready." + std::to_string(r), &n, sizeof(n))) { all_ready = false; continue; }
            seen[(size_t)r] = n;
        }
        if (all_ready && (!published || memcmp(seen.data(), written.data(), sizeof(Nonce) * (size_t)world) != 0)) {
            memcpy(blob.data(), &kRendezvousMagic, 4);
            memset(blob.data() + 4, 0, 4);
            memcpy(blob.data() + 8, run.b, 16);
            memcpy(blob.data() + 8 + 16, seen.data(), 16 * (size_t)world);
            memcpy(blob.data() + 8 + 16 + 16 * (size_t)world, id, sizeof(*id));
            if (!write_file_atomically(path, blob.data(), file_bytes)) { set_error("frames transport: cannot publish %s", path.c_str()); return false; }
            written = seen;
            published = true;
            all_acked = false;
        }
        if (published) {
            all_acked = true;
            for (int r = 1; r < world && all_acked; r++) {
                Nonce a, want;
                for (int i = 0; i < 16; i++) want.b[i] = written[(size_t)r].b[i] ^ run.b[i];
                all_acked = read_file(path + ".ack." + std::to_string(r), &a, sizeof(a)) && memcmp(a.b, want.b, 16) == 0;
            }
        }
        if (all_acked) return true;
        if (expired()) { set_error("frames transport: not every rank answered the rendezvous in %s within %d s", path.c_str(), timeout_s); (void)remove(path.c_str()); return false; }
        usleep(20000);
    }
}
static bool rendezvous_id(ohevc_frames_transport *t, NcclUniqueId *id)
{
    return rendezvous_exchange(t->rendezvous, t->rank, t->world, t->timeout_s, id, [t](NcclUniqueId *out) { return t->rccl.GetUniqueId(out) == 0; });
}
// the handshake alone, for tests without RCCL (ohevc_debug.h): rank 0 distributes the 128 bytes it is given, the others receive them
extern "C" int ohevc_debug_frames_rendezvous(const char *path, int rank, int world, int timeout_s, unsigned char id[128])
{
    OHEVC_REQUIRE(path != nullptr && id != nullptr && world >= 1 && rank >= 0 && rank < world, "bad argument");
    NcclUniqueId v;
    memcpy(&v, id, sizeof(v));
    const NcclUniqueId given = v;
    if (!rendezvous_exchange(path, rank, world, timeout_s, &v, [&given](NcclUniqueId *out) { *out = given; return true; })) return OHEVC_ERR_STATE;
    memcpy(id, &v, sizeof(v));
    // (the transport removes each rank's files after ncclCommInitRank, a collective: nobody still reads them.  Here rank 0, the last one out of
    // the handshake, clears the path for all.)
    if (rank == 0) {
        (void)remove(path);
        for (int r = 1; r < world; r++) { (void)remove((std::string(path) + ".ready." + std::to_string(r)).c_str()); (void)remove((std::string(path) + ".ack." + std::to_string(r)).c_str()); }
    }
    return OHEVC_OK;
}
static void rendezvous_cleanup(ohevc_frames_transport *t)
{
    if (t->rank == 0) { (void)remove(t->rendezvous.c_str()); return; }
    (void)remove((t->rendezvous + ".ready." + std::to_string(t->rank)).c_str());
    (void)remove((t->rendezvous + ".ack." + std::to_string(t->rank)).c_str());
}

// ------------------------------------------------------------------ life cycle
extern "C" int ohevc_frames_transport_create(ohevc_frames_transport **out, int rank, int world, int device, int wire, const char *rendezvous, int timeout_s)
{
    OHEVC_REQUIRE(out != nullptr && world >= 1 && rank >= 0 && rank < world, "rank / world");
    OHEVC_REQUIRE(wire == OHEVC_FRAMES_WIRE_RCCL || wire == OHEVC_FRAMES_WIRE_SOCKETS, "unknown wire");
    ohevc_frames_transport *t = new ohevc_frames_transport();
    t->rank = rank; t->world = world; t->device = device; t->wire = wire; t->timeout_s = timeout_s > 0 ? timeout_s : 60;
    t->rendezvous = rendezvous ? rendezvous : "";
    t->mode = ohhip_frames_mode{ sizeof(ohhip_frames_mode), rank, world, t, cb_publish, cb_subscribe, cb_await_motion, cb_await_planes, cb_release, cb_await_rows, 0 };
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
    if (world > 1 && !rendezvous_id(t, &id)) return fail(OHEVC_ERR_STATE);
    if (world == 1 && t->rccl.GetUniqueId(&id) != 0) { set_error("frames transport: ncclGetUniqueId failed"); return fail(OHEVC_ERR_STATE); }
    // ncclCommInitRank has no timeout of its own: a peer that died between the handshake and this call would hang the rank for ever.  It runs
    // on a helper thread; the caller waits for timeout_s and then gives up with an error (the helper, stuck inside RCCL, is left behind).
    struct Init { std::mutex m; std::condition_variable cv; bool done = false; int rc = 0; NcclComm comm = nullptr; };
    auto st = std::make_shared<Init>();
    {
        const Rccl r = t->rccl;
        std::thread([st, r, world, id, rank, device] {
            (void)hipSetDevice(device);
            NcclComm c = nullptr;
            const int rc = r.CommInitRank(&c, world, id, rank);
            std::lock_guard<std::mutex> g(st->m);
            st->rc = rc; st->comm = c; st->done = true;
            st->cv.notify_all();
        }).detach();
    }
    {
        std::unique_lock<std::mutex> lk(st->m);
        if (!st->cv.wait_for(lk, std::chrono::seconds(t->timeout_s), [&] { return st->done; })) {
            set_error("frames transport: ncclCommInitRank did not return within %d s (a peer died after the rendezvous?)", t->timeout_s);
            t->rccl.lib = nullptr;                             // (the helper still runs inside librccl: never unload it)
            return fail(OHEVC_ERR_STATE);
        }
    }
    if (st->rc != 0) { set_error("frames transport: ncclCommInitRank failed: %s", t->rccl.GetErrorString(st->rc)); t->comm = nullptr; return fail(OHEVC_ERR_STATE); }
    t->comm = st->comm;
    if (world > 1) rendezvous_cleanup(t);
    *out = t;
    return OHEVC_OK;
}

extern "C" int ohevc_frames_transport_set_bands(ohevc_frames_transport *t, int max_bands)
{
    OHEVC_REQUIRE(t != nullptr && max_bands >= 1, "bad argument");
    OHEVC_REQUIRE(t->pending.empty() && t->outgoing.empty(), "pictures are in flight: every rank must change the band grid at the same point of the stream");
    t->max_bands = std::min(max_bands, kMaxBands);
    return OHEVC_OK;
}

extern "C" int ohevc_frames_transport_set_ownership(ohevc_frames_transport *t, int per_idr_segment)
{
    OHEVC_REQUIRE(t != nullptr, "null transport");
    OHEVC_REQUIRE(t->pending.empty() && t->outgoing.empty(), "pictures are in flight");
    t->mode.segment_ownership = per_idr_segment != 0;
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

// One picture through the wire, on every rank at once: `root` sends the picture in its src_slot and `mvf_in`, every rank - root included -
// puts what arrived into its dst_slot and mvf_out.  The same steps as publish on the owner and subscribe + await_planes + await_motion on the
// others (stage, ohevc_pic_export, one group of four broadcasts, the wait, ohevc_pic_import); a start-up check of the wire, and - with
// world 1, where the decoder itself never exchanges anything - the only way to execute the RCCL calls on a single GPU.
extern "C" int ohevc_frames_transport_selftest(ohevc_frames_transport *t, ohevc_ctx *ctx, int src_slot, int dst_slot, int root, const void *mvf_in, void *mvf_out,
                                               size_t mvf_bytes)
{
    OHEVC_REQUIRE(t != nullptr && ctx != nullptr && root >= 0 && root < t->world && mvf_out != nullptr, "bad argument");
    OHEVC_REQUIRE(t->rank != root || mvf_in != nullptr, "the root needs a motion field to send");
    if (t->broken) { set_error("frames transport: broken"); return OHEVC_ERR_STATE; }
    (void)hipSetDevice(t->device);
    const bool me = t->rank == root;
    Msg *m = t->stage(-1, ctx, me ? src_slot : dst_slot, mvf_bytes, root);
    if (!m) return OHEVC_ERR_STATE;
    m->outgoing = false;                                       // (every rank copies the motion field back, the root too)
    int rc = OHEVC_OK;
    if (me) {
        memcpy(m->h_mvf, mvf_in, mvf_bytes);
        memset(m->h_mvf + mvf_bytes, 0, 8);
        for (int c = 0; c < 3 && rc == OHEVC_OK; c++) {
            rc = ohevc_pic_export(ctx, src_slot, c, m->d_plane[c], m->plane_bytes[c]);
            if (rc == OHEVC_OK && t->wire == OHEVC_FRAMES_WIRE_SOCKETS && hipMemcpy(m->h_plane[c], m->d_plane[c], m->plane_bytes[c], hipMemcpyDeviceToHost) != hipSuccess) rc = OHEVC_ERR_HIP;
        }
        if (rc == OHEVC_OK && t->wire == OHEVC_FRAMES_WIRE_RCCL && hipMemcpyAsync(m->d_mvf, m->h_mvf, mvf_bytes + 8, hipMemcpyHostToDevice, t->stream) != hipSuccess) rc = OHEVC_ERR_HIP;
        if (rc == OHEVC_OK && t->wire == OHEVC_FRAMES_WIRE_RCCL) {
            if (hipStreamSynchronize(t->stream) != hipSuccess) rc = OHEVC_ERR_HIP;     // (the copy above reads the host buffer asynchronously)
            memset(m->h_mvf, 0xee, mvf_bytes + 8);                                     // what comes back must come off the wire
        }
    }
    if (rc == OHEVC_OK && !t->post_mvf(m)) rc = OHEVC_ERR_STATE;
    for (int b = 0; b < m->nb && rc == OHEVC_OK; b++) if (!t->post_band(m, b)) rc = OHEVC_ERR_STATE;
    if (rc == OHEVC_OK && !t->complete(m)) { set_error("frames transport: the self-test picture did not arrive from rank %d within %d s", root, t->timeout_s); t->broken = true; return OHEVC_ERR_STATE; }
    for (int c = 0; c < 3 && rc == OHEVC_OK; c++) {
        if (t->wire == OHEVC_FRAMES_WIRE_SOCKETS && !me && hipMemcpy(m->d_plane[c], m->h_plane[c], m->plane_bytes[c], hipMemcpyHostToDevice) != hipSuccess) rc = OHEVC_ERR_HIP;
        for (int b = 0; b < m->nb && rc == OHEVC_OK; b++)      // band by band, like a subscriber's await_rows
            rc = ohevc_pic_import_rows(ctx, dst_slot, c, m->row0[c][b], m->row0[c][b + 1] - m->row0[c][b], m->d_plane[c], b == 0);
    }
    if (rc == OHEVC_OK) {
        if (m->h_mvf[mvf_bytes] != 0) { set_error("frames transport: the self-test message carries an error mark"); rc = OHEVC_ERR_STATE; }
        memcpy(mvf_out, m->h_mvf, mvf_bytes);
    }
    t->stats.bytes += 0;
    t->recycle(m);
    return rc;
}

extern "C" int ohevc_frames_transport_stats(ohevc_frames_transport *t, ohevc_frames_stats *out)
{
    OHEVC_REQUIRE(t != nullptr && out != nullptr, "null argument");
    *out = t->stats;
    out->wire_ranks = 0;
    if (t->wire == OHEVC_FRAMES_WIRE_RCCL) {
        int n = 0;
        if (t->comm && t->rccl.CommCount && t->rccl.CommCount(t->comm, &n) == 0) out->wire_ranks = n;
    } else {
        out->wire_ranks = 1;
        for (int f : t->sock.fd) out->wire_ranks += f >= 0;
    }
    return OHEVC_OK;
}
