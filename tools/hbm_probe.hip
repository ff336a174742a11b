// tools/hbm_probe.hip -- what this MI355X sustains on plain streaming patterns (measurement aid, not product code).
//
//   hipcc --offload-arch=gfx950 -O3 tools/hbm_probe.hip -o tools/hbm_probe && tools/hbm_probe [GiB-of-payload]
//
// Hand-written 16-byte-per-lane kernels over buffers far larger than the 256 MiB Infinity Cache, each timed with HIP events over
// several launches on rotating buffers (a launch never touches what the previous one left in the cache):
//   copy        1 read : 1 write, float4 per lane                      (the guide's 6.29 TB/s figure is this pattern)
//   read        read-only, 4 x 16 B in flight per lane, one dword written per workgroup
//   write       write-only
//   mix31       the 8-bit residual kernel's traffic: 32 B of int16 coefficients + 16 B of prediction read, 16 B written IN PLACE
//   mix21       the 16-bit one: 32 B coefficients + 32 B prediction read, 32 B written in place
//   mix31_nt    mix31 with non-temporal coefficient loads (the stream that is read exactly once)
// One JSON line per pattern: GB/s of total traffic (reads + writes) and the share of the 8 TB/s HBM3E peak.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <bool NT> __device__ __forceinline__ u32x4 ld16(const u32x4 *p)
{
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}

// every workgroup owns a contiguous span of `per_wg` 16-byte chunks; lanes stride through it, UNROLL accesses in flight
template <int UNROLL>
__global__ __launch_bounds__(256) void copy_kernel(u32x4 *__restrict__ dst, const u32x4 *__restrict__ src, size_t n)
{
    const size_t base = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x;
    u32x4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) if (base + (size_t)u * 256 < n) v[u] = src[base + (size_t)u * 256];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) if (base + (size_t)u * 256 < n) dst[base + (size_t)u * 256] = v[u];
}

template <int UNROLL>
__global__ __launch_bounds__(256) void read_kernel(unsigned *__restrict__ out, const u32x4 *__restrict__ src, size_t n)
{
    const size_t base = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x;
    unsigned acc = 0;
#pragma unroll
    for (int u = 0; u < UNROLL; u++)
        if (base + (size_t)u * 256 < n) { const u32x4 v = src[base + (size_t)u * 256]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[blockIdx.x] = acc;          // practically never: keeps the loads alive
}

template <int UNROLL>
__global__ __launch_bounds__(256) void write_kernel(u32x4 *__restrict__ dst, size_t n, unsigned seed)
{
    const size_t base = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x;
#pragma unroll
    for (int u = 0; u < UNROLL; u++)
        if (base + (size_t)u * 256 < n) dst[base + (size_t)u * 256] = u32x4{ seed, (unsigned)base, seed ^ u, 0u };
}

// residual-like mix: CPP 16-byte coefficient chunks per 16-byte pixel chunk (2 for 8-bit pixels, 1 for 16-bit), pixels updated in place
template <int CPP, bool NT>
__global__ __launch_bounds__(256) void mix_kernel(u32x4 *__restrict__ px, const u32x4 *__restrict__ coef, size_t npx)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npx) return;
    u32x4 c[CPP];
#pragma unroll
    for (int k = 0; k < CPP; k++) c[k] = ld16<NT>(coef + i * CPP + k);
    u32x4 p = px[i];
#pragma unroll
    for (int k = 0; k < CPP; k++) { p.x += c[k].x; p.y ^= c[k].y; p.z += c[k].z; p.w ^= c[k].w; }
    px[i] = p;
}

// the residual kernel's REAL pixel addressing: a plane 16384 samples wide, cut into tiles of SW x 32 samples; a group of SW lanes
// (one wave for SW = 64, the whole workgroup for SW = 256) owns a tile, reads its 4 * SW / 64 KB of coefficients linearly and its
// 32 rows as SW-byte pieces, 16384 bytes apart; pixels updated in place.  SW = 64: the wave-private strips of tu_idct_add_kernel.
template <int SW, bool NT>
__global__ __launch_bounds__(256) void mix_tiled_kernel(unsigned char *__restrict__ px, const u32x4 *__restrict__ coef, int tiles)
{
    constexpr int LANES = SW;                             // lanes per tile: SW * 32 samples / 16 per chunk / 2 chunks per lane
    constexpr int TPW = 256 / LANES;                      // tiles per workgroup
    const int t = blockIdx.x * TPW + threadIdx.x / LANES, l = threadIdx.x % LANES;
    if (t >= tiles) return;
    constexpr int CPR = SW / 16;                          // 16-byte chunks per tile row
    const int tiles_per_row = 16384 / SW;
    unsigned char *base = px + (size_t)(t / tiles_per_row) * 32 * 16384 + (size_t)(t % tiles_per_row) * SW;
    const u32x4 *cf = coef + (size_t)t * (SW * 32 * 2 / 16);
    u32x4 c[4];
#pragma unroll
    for (int k = 0; k < 4; k++) c[k] = ld16<NT>(cf + k * LANES + l);
    u32x4 p[2];
#pragma unroll
    for (int k = 0; k < 2; k++) p[k] = *reinterpret_cast<const u32x4 *>(base + (size_t)(l / CPR + k * (LANES / CPR)) * 16384 + (l % CPR) * 16);
#pragma unroll
    for (int k = 0; k < 2; k++) { p[k].x += c[2 * k].x; p[k].y ^= c[2 * k].y; p[k].z += c[2 * k + 1].z; p[k].w ^= c[2 * k + 1].w; }
#pragma unroll
    for (int k = 0; k < 2; k++) *reinterpret_cast<u32x4 *>(base + (size_t)(l / CPR + k * (LANES / CPR)) * 16384 + (l % CPR) * 16) = p[k];
}

struct Timer {
    hipEvent_t a, b;
    Timer() { CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b)); }
    template <typename F> double run(int reps, F f)
    {
        f(0); f(1);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(a, 0));
        for (int r = 0; r < reps; r++) f(r);
        CHECK(hipEventRecord(b, 0));
        CHECK(hipEventSynchronize(b));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, a, b));
        return ms * 1e-3 / reps;
    }
};

static void fill_random(void *p, size_t bytes)
{
    // device-side pseudo-random fill (xorshift of the index): no zero pages, no constant data (DVFS: low-entropy data clocks higher)
    const size_t n = bytes / 16;
    hipLaunchKernelGGL((write_kernel<4>), dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, 0, (u32x4 *)p, n, 0x9e3779b9u);
    CHECK(hipGetLastError());
}

int main(int argc, char **argv)
{
    const double gib = argc > 1 ? atof(argv[1]) : 2.0;
    const size_t bytes = (size_t)(gib * (1ull << 30)) & ~(size_t)4095;
    const int RING = 3;                                   // rotating buffer sets
    CHECK(hipSetDevice(0));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    std::vector<void *> a(RING), b(RING), c(RING);
    for (int r = 0; r < RING; r++) {
        CHECK(hipMalloc(&a[r], bytes)); CHECK(hipMalloc(&b[r], bytes)); CHECK(hipMalloc(&c[r], 2 * bytes));
        fill_random(a[r], bytes); fill_random(b[r], bytes); fill_random(c[r], 2 * bytes);
    }
    unsigned *sink;
    CHECK(hipMalloc((void **)&sink, 1 << 24));
    CHECK(hipDeviceSynchronize());
    const size_t n = bytes / 16;
    Timer t;
    const int reps = 12;
    auto report = [&](const char *name, double sec, double traffic, const char *note) {
        printf("{\"pattern\": \"%s\", \"GBps\": %.1f, \"frac_of_8TBps\": %.4f, \"ms\": %.4f, \"bytes_moved\": %.0f, \"note\": \"%s\", \"device\": \"%s\"}\n",
               name, traffic / sec / 1e9, traffic / sec / 8e12, sec * 1e3, traffic, note, prop.gcnArchName);
        fflush(stdout);
    };
    double s;
    s = t.run(reps, [&](int r) { hipLaunchKernelGGL((copy_kernel<4>), dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, 0, (u32x4 *)b[r % RING], (const u32x4 *)a[(r + 1) % RING], n); });
    report("copy", s, 2.0 * bytes, "1 read : 1 write, 16 B per lane, 4 in flight");
    s = t.run(reps, [&](int r) { hipLaunchKernelGGL((copy_kernel<1>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (u32x4 *)b[r % RING], (const u32x4 *)a[(r + 1) % RING], n); });
    report("copy_1", s, 2.0 * bytes, "1 read : 1 write, 16 B per lane, 1 in flight");
    s = t.run(reps, [&](int r) { hipLaunchKernelGGL((read_kernel<4>), dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, 0, sink, (const u32x4 *)a[r % RING], n); });
    report("read", s, 1.0 * bytes, "read-only");
    s = t.run(reps, [&](int r) { hipLaunchKernelGGL((write_kernel<4>), dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, 0, (u32x4 *)b[r % RING], n, (unsigned)r); });
    report("write", s, 1.0 * bytes, "write-only");
    // mix31: pixel buffer = bytes, coefficient buffer = 2 * bytes
    s = t.run(reps, [&](int r) { hipLaunchKernelGGL((mix_kernel<2, false>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (u32x4 *)a[r % RING], (const u32x4 *)c[(r + 1) % RING], n); });
    report("mix31", s, 4.0 * bytes, "8-bit residual pattern: 2 B coefficients + 1 B prediction read, 1 B written in place, per sample");
    s = t.run(reps, [&](int r) { hipLaunchKernelGGL((mix_kernel<2, true>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (u32x4 *)a[r % RING], (const u32x4 *)c[(r + 1) % RING], n); });
    report("mix31_nt", s, 4.0 * bytes, "same, non-temporal coefficient loads");
    s = t.run(reps, [&](int r) { hipLaunchKernelGGL((mix_kernel<1, false>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (u32x4 *)a[r % RING], (const u32x4 *)b[(r + 1) % RING], n); });
    report("mix21", s, 3.0 * bytes, "16-bit residual pattern: 2 B coefficients + 2 B prediction read, 2 B written in place, per sample");
    {   // tiled addressing: the pixel buffer as a 16384-wide plane
        const int tiles64 = (int)(bytes / (64 * 32)), tiles128 = (int)(bytes / (128 * 32)), tiles256 = (int)(bytes / (256 * 32));
        s = t.run(reps, [&](int r) { hipLaunchKernelGGL((mix_tiled_kernel<64, false>), dim3((unsigned)((tiles64 + 3) / 4)), dim3(256), 0, 0, (unsigned char *)a[r % RING], (const u32x4 *)c[(r + 1) % RING], tiles64); });
        report("mix31_tiled64", s, 4.0 * bytes, "8-bit residual pattern on a 16384-wide plane, 64 x 32 tiles (64-byte row pieces)");
        s = t.run(reps, [&](int r) { hipLaunchKernelGGL((mix_tiled_kernel<64, true>), dim3((unsigned)((tiles64 + 3) / 4)), dim3(256), 0, 0, (unsigned char *)a[r % RING], (const u32x4 *)c[(r + 1) % RING], tiles64); });
        report("mix31_tiled64_nt", s, 4.0 * bytes, "same, non-temporal coefficient loads");
        s = t.run(reps, [&](int r) { hipLaunchKernelGGL((mix_tiled_kernel<128, true>), dim3((unsigned)((tiles128 + 1) / 2)), dim3(256), 0, 0, (unsigned char *)a[r % RING], (const u32x4 *)c[(r + 1) % RING], tiles128); });
        report("mix31_tiled128_nt", s, 4.0 * bytes, "128 x 32 tiles (one full line per row), nt coefficients");
        s = t.run(reps, [&](int r) { hipLaunchKernelGGL((mix_tiled_kernel<256, true>), dim3((unsigned)tiles256), dim3(256), 0, 0, (unsigned char *)a[r % RING], (const u32x4 *)c[(r + 1) % RING], tiles256); });
        report("mix31_tiled256_nt", s, 4.0 * bytes, "256 x 32 tiles (two full lines per row), nt coefficients");
    }
    return 0;
}
