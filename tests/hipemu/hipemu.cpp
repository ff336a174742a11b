// hipemu.cpp -- TEST-ONLY host emulation of the HIP execution model for the kernels under openhevc_amd/csrc/.
//
// A launch runs synchronously on the calling host thread, workgroup after workgroup.  Every lane of a workgroup is a
// fiber (ucontext); __syncthreads and the wave-level operations (wave barrier, shuffles, ballot, readfirstlane, the
// matrix-core and transposing-read collectives) are scheduling points: a wave's lanes run, in lane order, until each has
// reached its next such point, and are then released together.  That is one legal schedule of the GPU's, so a kernel
// that is correct on the device gives the same bytes here; HIPEMU_ORDER=reverse runs lanes and waves in the opposite
// order (a second legal schedule - results that differ between the two point at a missing barrier).
// Not emulated: spin-waits between workgroups (they run one after another), timing, LDS bank behaviour.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <sys/mman.h>
#include <ucontext.h>
#include <vector>

// AddressSanitizer build (make SAN=1): tell the runtime about every stack switch
#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#define HIPEMU_ASAN 1
extern "C" void __sanitizer_start_switch_fiber(void **fake_stack_save, const void *bottom, size_t size);
extern "C" void __sanitizer_finish_switch_fiber(void *fake_stack_save, const void **bottom_old, size_t *size_old);
#endif
#endif

namespace hipemu {

thread_local Lane g_lane;

enum { READY = 0, WAIT_WAVE = 1, WAIT_BLOCK = 2, DONE = 3 };
static const size_t kStackBytes = 256 << 10;

struct Fiber {
    ucontext_t ctx;
    void      *stack = nullptr;
    int        state = DONE;
    unsigned   or_calls = 0;
    void      *fake = nullptr;      // ASan's fake-stack handle while the fiber is switched out
};

struct Sched {
    ucontext_t                   main;
    std::vector<Fiber>           fibers;
    int                          nthreads = 0, cur = 0;
    dim3                         block;
    const std::function<void()> *body = nullptr;
    std::vector<uint64_t>        slots;
    std::vector<unsigned char>   gather;      // 16 bytes per lane
    bool                         reverse = false;
    bool                         running = false;
    int                          or_acc[2] = { 0, 0 };
    unsigned                     spins = 0;
    const void                  *main_bottom = nullptr;     // ASan: the scheduler's own stack, learnt at the first switch
    size_t                       main_size = 0;
    void                        *main_fake = nullptr;

    void ensure(int n)
    {
        if ((int)fibers.size() < n) fibers.resize(n);
        for (int i = 0; i < n; i++)
            if (!fibers[i].stack) {
                void *p = mmap(nullptr, kStackBytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_STACK, -1, 0);
                if (p == MAP_FAILED) { perror("hipemu: mmap of a fiber stack"); abort(); }
                fibers[i].stack = p;
            }
        if ((int)slots.size() < n) { slots.resize(n); gather.resize((size_t)n * 16); }
    }
};
static thread_local Sched *g_sched = nullptr;

static Sched &sched()
{
    if (!g_sched) {
        g_sched = new Sched();
        const char *o = getenv("HIPEMU_ORDER");
        g_sched->reverse = o && o[0] == 'r';
    }
    return *g_sched;
}

[[noreturn]] void unsupported(const char *what)
{
    fprintf(stderr, "hipemu: %s\n", what);
    abort();
}

void spin()
{
    if (++g_sched->spins > (1u << 22))
        unsupported("a workgroup keeps polling: it waits for a workgroup that runs after it (workgroups are emulated one after another, in launch order)");
}

static void set_lane(Sched &s, int flat)
{
    s.cur = flat;
    g_lane.flat = flat;
    g_lane.tid.x = flat % s.block.x;
    g_lane.tid.y = (flat / s.block.x) % s.block.y;
    g_lane.tid.z = flat / (s.block.x * s.block.y);
}

static void yield(int state)
{
    Sched &s = *g_sched;
    Fiber &f = s.fibers[s.cur];
    f.state = state;
#ifdef HIPEMU_ASAN
    __sanitizer_start_switch_fiber(state == DONE ? nullptr : &f.fake, s.main_bottom, s.main_size);
#endif
    swapcontext(&f.ctx, &s.main);
#ifdef HIPEMU_ASAN
    __sanitizer_finish_switch_fiber(f.fake, &s.main_bottom, &s.main_size);
#endif
}

static void fiber_entry()
{
    Sched &s = *g_sched;
#ifdef HIPEMU_ASAN
    __sanitizer_finish_switch_fiber(nullptr, &s.main_bottom, &s.main_size);
#endif
    (*s.body)();
    yield(DONE);
    abort();     // a finished fiber is never resumed
}

void sync_block() { yield(WAIT_BLOCK); }
void sync_wave() { yield(WAIT_WAVE); }

static void run_block(Sched &s)
{
    const int n = s.nthreads, nwaves = (n + 63) / 64;
    for (int i = 0; i < n; i++) {
        Fiber &f = s.fibers[i];
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack;
        f.ctx.uc_stack.ss_size = kStackBytes;
        f.ctx.uc_link = nullptr;
        makecontext(&f.ctx, fiber_entry, 0);
        f.state = READY;
        f.or_calls = 0;
    }
    for (;;) {
        for (int wi = 0; wi < nwaves; wi++) {
            const int w = s.reverse ? nwaves - 1 - wi : wi;
            const int lo = w * 64, hi = lo + 64 < n ? lo + 64 : n;
            for (;;) {
                for (int li = lo; li < hi; li++) {
                    const int l = s.reverse ? hi - 1 - (li - lo) : li;
                    if (s.fibers[l].state == READY) {
                        set_lane(s, l);
#ifdef HIPEMU_ASAN
                        __sanitizer_start_switch_fiber(&s.main_fake, s.fibers[l].stack, kStackBytes);
#endif
                        swapcontext(&s.main, &s.fibers[l].ctx);
#ifdef HIPEMU_ASAN
                        __sanitizer_finish_switch_fiber(s.main_fake, nullptr, nullptr);
#endif
                    }
                }
                bool any = false;
                for (int l = lo; l < hi; l++)
                    if (s.fibers[l].state == WAIT_WAVE) { s.fibers[l].state = READY; any = true; }
                if (!any) break;
            }
        }
        bool any = false;
        for (int l = 0; l < n; l++)
            if (s.fibers[l].state == WAIT_BLOCK) { s.fibers[l].state = READY; any = true; }
        if (!any) break;
    }
}

void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()> &body)
{
    (void)shmem;
    Sched &s = sched();
    if (s.running) unsupported("nested launch");
    const uint64_t n = (uint64_t)block.x * block.y * block.z;
    if (n == 0 || n > 1024) unsupported("workgroup size outside 1..1024");
    if ((uint64_t)grid.x * grid.y * grid.z == 0) return;
    s.running = true;
    s.spins = 0;
    s.nthreads = (int)n;
    s.block = block;
    s.body = &body;
    s.ensure((int)n);
    g_lane.bdim = block;
    g_lane.gdim = grid;
    for (unsigned bz = 0; bz < grid.z; bz++)
        for (unsigned by = 0; by < grid.y; by++)
            for (unsigned bx = 0; bx < grid.x; bx++) {
                g_lane.bid = dim3(bx, by, bz);
                run_block(s);
            }
    s.body = nullptr;
    s.running = false;
}

// __syncthreads_or: two alternating accumulators, so that a lane already in the next call cannot disturb this one's result
int block_or(int pred)
{
    Sched &s = *g_sched;
    // every lane of the block is in the same call (block-uniform control flow is the caller's contract), so the phase
    // a lane uses is a function of how many calls it has made; keep that count per lane in its slot-free fiber state
    Fiber &f = s.fibers[s.cur];
    const unsigned ph = f.or_calls++ & 1;
    if (pred) s.or_acc[ph] = 1;
    sync_block();
    const int r = s.or_acc[ph];
    sync_block();
    s.or_acc[ph] = 0;
    return r;
}

// ---- wave collectives: publish, wave-sync, read, wave-sync (the second sync keeps a fast lane's next publish off the slot)
uint64_t wave_exchange(uint64_t mine, int src_lane)
{
    Sched &s = *g_sched;
    const int me = s.cur, base = me & ~63;
    s.slots[me] = mine;
    sync_wave();
    const int src = base + (src_lane & 63);
    const uint64_t r = src < s.nthreads ? s.slots[src] : 0;
    sync_wave();
    return r;
}

uint64_t wave_ballot(bool pred)
{
    Sched &s = *g_sched;
    const int me = s.cur, base = me & ~63;
    s.slots[me] = pred ? 1 : 0;
    sync_wave();
    uint64_t r = 0;
    for (int l = base; l < base + 64 && l < s.nthreads; l++)
        if (s.fibers[l].state != DONE && s.slots[l]) r |= 1ull << (l - base);
    sync_wave();
    return r;
}

uint64_t wave_first(uint64_t mine)
{
    Sched &s = *g_sched;
    const int me = s.cur, base = me & ~63;
    s.slots[me] = mine;
    sync_wave();
    uint64_t r = mine;
    for (int l = base; l < base + 64 && l < s.nthreads; l++)
        if (s.fibers[l].state != DONE) { r = s.slots[l]; break; }
    sync_wave();
    return r;
}

void wave_gather(const void *mine, size_t bytes, void *all64)
{
    Sched &s = *g_sched;
    if (bytes > 16) unsupported("wave_gather of more than 16 bytes per lane");
    const int me = s.cur, base = me & ~63;
    memcpy(&s.gather[(size_t)me * 16], mine, bytes);
    sync_wave();
    for (int i = 0; i < 64; i++) {
        const int l = base + i;
        if (l < s.nthreads && s.fibers[l].state != DONE) memcpy((unsigned char *)all64 + i * bytes, &s.gather[(size_t)l * 16], bytes);
        else memset((unsigned char *)all64 + i * bytes, 0, bytes);
    }
    sync_wave();
}

}  // namespace hipemu

// ---------------------------------------------------------------- runtime API: host memory, ordering-only streams / events
struct hipemuStream { int unused; };
struct hipemuEvent { int unused; };

extern "C" {
const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "hipemu error"; }
hipError_t hipGetLastError(void) { return hipSuccess; }
hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : hipErrorInvalidValue; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int d)
{
    if (d != 0) return hipErrorInvalidValue;
    memset(p, 0, sizeof(*p));
    snprintf(p->name, sizeof(p->name), "hipemu (host emulation, tests only)");
    snprintf(p->gcnArchName, sizeof(p->gcnArchName), "gfx950:hipemu");
    p->multiProcessorCount = 256;
    p->totalGlobalMem = (size_t)8 << 30;
    p->warpSize = 64;
    return hipSuccess;
}
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
hipError_t hipMalloc(void **p, size_t n)
{
    void *q = nullptr;
    if (posix_memalign(&q, 256, n ? n : 1)) return hipErrorOutOfMemory;
    const char *poison = getenv("HIPEMU_POISON");
    memset(q, poison && poison[0] == '1' ? 0xA5 : 0, n);
    *p = q;
    return hipSuccess;
}
hipError_t hipFree(void *p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void **p, size_t n, unsigned) { return hipMalloc(p, n); }
hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
hipError_t hipHostRegister(void *, size_t, unsigned) { return hipSuccess; }
hipError_t hipHostUnregister(void *) { return hipSuccess; }
hipError_t hipMemcpy(void *dst, const void *src, size_t n, hipMemcpyKind) { memmove(dst, src, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, hipMemcpyKind, hipStream_t) { memmove(dst, src, n); return hipSuccess; }
hipError_t hipMemcpy2DAsync(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height, hipMemcpyKind, hipStream_t)
{
    for (size_t y = 0; y < height; y++) memmove((char *)dst + y * dpitch, (const char *)src + y * spitch, width);
    return hipSuccess;
}
hipError_t hipMemsetAsync(void *dst, int v, size_t n, hipStream_t) { memset(dst, v, n); return hipSuccess; }
hipError_t hipMemset(void *dst, int v, size_t n) { memset(dst, v, n); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = new hipemuStream(); return hipSuccess; }
hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned, int) { *s = new hipemuStream(); return hipSuccess; }
hipError_t hipExtStreamCreateWithCUMask(hipStream_t *s, uint32_t, const uint32_t *) { *s = new hipemuStream(); return hipSuccess; }
hipError_t hipDeviceGetStreamPriorityRange(int *least, int *greatest) { if (least) *least = 0; if (greatest) *greatest = -1; return hipSuccess; }
hipError_t hipStreamCreate(hipStream_t *s) { *s = new hipemuStream(); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = new hipemuEvent(); return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t *e) { *e = new hipemuEvent(); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
}
