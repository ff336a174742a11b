// What the "one short workgroup per block of samples" structure of the SAO / MC / deblocking kernels can reach at most: workgroups that do
// nothing, and workgroups that copy 16 bytes per lane (address from blockIdx alone, or through a 32-byte job record as the kernels do),
// at 64 / 128 / 256 threads.  Build: hipcc --offload-arch=gfx950 -O3 tools/probes/dispatch_probe.hip -o tools/probes/dispatch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct Job { unsigned off, pad[7]; };

template <int T> __global__ __launch_bounds__(T) void k_empty(int *sink) { if (sink == nullptr && threadIdx.x == 9999) *sink = 1; }

template <int T, int MODE> __global__ __launch_bounds__(T) void k_copy(const uint4 *__restrict__ src, uint4 *__restrict__ dst, const Job *__restrict__ jobs)
{
    size_t base = (size_t)blockIdx.x * T;
    if (MODE == 1) base = jobs[blockIdx.x].off;            // scalar load of the job record first, like the kernels
    uint4 v = src[base + threadIdx.x];
    if (MODE == 2) { uint4 a = src[base + threadIdx.x + 4096], b = src[base + threadIdx.x + 8192]; v.x ^= a.x & b.y; v.y += a.w; }   // three loads per lane (SAO edge)
    v.x += 1;
    dst[base + threadIdx.x] = v;
}

// the SAO / deblocking access pattern: workgroup = one tile of a 2-D plane (tile_w bytes x tile_h rows, 16 bytes per lane, lanes row-major in the tile)
__global__ __launch_bounds__(256) void k_tile(const unsigned char *__restrict__ src, unsigned char *__restrict__ dst, int stride, int tiles_x, int tile_w, int tile_h, int xcd_contig, int ntiles)
{
    int t = blockIdx.x;
    if (xcd_contig) { const int per = (ntiles + 7) >> 3; t = (t & 7) * per + (t >> 3); if (t >= ntiles) return; }
    const int ty = t / tiles_x, tx = t - ty * tiles_x;
    const int pieces = tile_w >> 4, rows_per_pass = 256 / pieces;
    const int piece = threadIdx.x % pieces, row0 = threadIdx.x / pieces;
    for (int y = row0; y < tile_h; y += rows_per_pass) {
        const size_t off = (size_t)(ty * tile_h + y) * stride + (size_t)tx * tile_w + piece * 16;
        uint4 v = *reinterpret_cast<const uint4 *>(src + off);
        v.x += 1;
        *reinterpret_cast<uint4 *>(dst + off) = v;
    }
}

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <typename F> static float timeit(F f, int reps)
{
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int i = 0; i < 3; i++) f(i);
    CHECK(hipEventRecord(a));
    for (int i = 0; i < reps; i++) f(i);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main()
{
    const size_t bytes_per_set = 64ull << 20;              // one "launch" touches 64 MiB in, 64 MiB out
    const int nsets = 12;                                  // rotate through 1.5 GiB so nothing is L2 / MALL resident
    uint4 *src, *dst; Job *jobs;
    CHECK(hipMalloc(&src, bytes_per_set * nsets + (1 << 20))); CHECK(hipMalloc(&dst, bytes_per_set * nsets));
    CHECK(hipMemset(src, 1, bytes_per_set * nsets + (1 << 20))); CHECK(hipMemset(dst, 0, bytes_per_set * nsets));
    const size_t elems = bytes_per_set / 16;
    for (int T : { 64, 128, 256 }) {
        const int grid = (int)(elems / T);
        std::vector<Job> hj(grid); for (int i = 0; i < grid; i++) { hj[i].off = (unsigned)((size_t)i * T); }
        CHECK(hipMalloc(&jobs, sizeof(Job) * grid)); CHECK(hipMemcpy(jobs, hj.data(), sizeof(Job) * grid, hipMemcpyHostToDevice));
        auto run = [&](int mode, int i) {
            const uint4 *s = src + (size_t)(i % nsets) * elems; uint4 *d = dst + (size_t)(i % nsets) * elems;
            if (T == 64)  { if (mode == 0) k_copy<64, 0><<<grid, 64>>>(s, d, jobs); else if (mode == 1) k_copy<64, 1><<<grid, 64>>>(s, d, jobs); else if (mode == 2) k_copy<64, 2><<<grid, 64>>>(s, d, jobs); else k_empty<64><<<grid, 64>>>((int *)d); }
            if (T == 128) { if (mode == 0) k_copy<128, 0><<<grid, 128>>>(s, d, jobs); else if (mode == 1) k_copy<128, 1><<<grid, 128>>>(s, d, jobs); else if (mode == 2) k_copy<128, 2><<<grid, 128>>>(s, d, jobs); else k_empty<128><<<grid, 128>>>((int *)d); }
            if (T == 256) { if (mode == 0) k_copy<256, 0><<<grid, 256>>>(s, d, jobs); else if (mode == 1) k_copy<256, 1><<<grid, 256>>>(s, d, jobs); else if (mode == 2) k_copy<256, 2><<<grid, 256>>>(s, d, jobs); else k_empty<256><<<grid, 256>>>((int *)d); }
        };
        const char *names[4] = { "copy 16 B per lane, address from blockIdx", "copy 16 B per lane, address from a job record", "3 loads + 1 store per lane, address from blockIdx", "empty workgroups" };
        for (int mode : { 3, 0, 1, 2 }) {
            const float ms = timeit([&](int i) { run(mode, i); }, 40);
            const double gbs = mode == 3 ? 0.0 : 2.0 * bytes_per_set / (ms * 1e-3) / 1e9;
            printf("{\"threads\": %d, \"workgroups\": %d, \"kind\": \"%s\", \"ms\": %.4f, \"workgroups_per_us\": %.1f, \"waves_per_us\": %.1f, \"GBps\": %.0f, \"frac_of_8TBps\": %.3f}\n",
                   T, grid, names[mode], ms, grid / (ms * 1e3), grid * (T / 64) / (ms * 1e3), gbs, gbs / 8000.0);
        }
        CHECK(hipFree(jobs));
    }
    // 8 planes of 3840 x 2160 bytes (8-bit 4K luma) / 7680-byte rows (10-bit), tiles as the SAO kernel cuts them
    for (int row_bytes : { 3840, 7680 }) {
        const int H = 2160, planes = 8;
        for (int tw : { 64, 128, 256 }) {
            if (row_bytes % tw) continue;
            for (int xc : { 0, 1 }) {
                const int th = 4096 / tw > 64 ? 64 : 4096 / tw;        // 4 KiB per workgroup (one 64 x 64 block of 8-bit samples), at most 64 rows
                const int tiles_x = row_bytes / tw, tiles_y = (H * planes) / th, nt = tiles_x * tiles_y;
                const size_t plane_set = (size_t)row_bytes * H * planes;
                const int sets = (int)((bytes_per_set * nsets) / plane_set);
                const float ms = timeit([&](int i) {
                    const size_t o = (size_t)(i % sets) * plane_set;
                    k_tile<<<(nt + 7) / 8 * 8, 256>>>((const unsigned char *)src + o, (unsigned char *)dst + o, row_bytes, tiles_x, tw, th, xc, nt);
                }, 40);
                const double gbs = 2.0 * plane_set / (ms * 1e-3) / 1e9;
                printf("{\"kind\": \"2-D tile copy\", \"row_bytes\": %d, \"tile\": \"%d bytes x %d rows\", \"xcd_contiguous\": %d, \"workgroups\": %d, \"ms\": %.4f, \"GBps\": %.0f, \"frac_of_8TBps\": %.3f}\n",
                       row_bytes, tw, th, xc, nt, ms, gbs, gbs / 8000.0);
            }
        }
    }
    return 0;
}
