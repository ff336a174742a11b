/* host_smoke.c -- a plain C host (no Python, no torch) driving libohevc_hip.so, the way the reference's C
 * decoder would after linking the drop-in.  Allocates device buffers through the HIP C API, runs one small
 * batched 16x16 IDCT+add through the C ABI and compares with a scalar reimplementation of the same few lines
 * (DC-only input, where the transform is a constant: idct_dc, hevcdsp_template.c:303-316).
 * Build: gcc host_smoke.c -I../../include -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -L../../openhevc_amd -lohevc_hip -L/opt/rocm/lib -lamdhip64
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <hip/hip_runtime_api.h>
#include "ohevc_hip.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)

int main(void)
{
    enum { N = 16, NB = 40, W = 256, H = 48 };
    if (ohevc_device_count() < 1) { printf("no device\n"); return 3; }
    if (ohevc_set_device(0) != OHEVC_OK) { printf("set_device: %s\n", ohevc_last_error()); return 3; }
    static uint8_t plane[H][W], want[H][W];
    static int16_t coeffs[NB][N * N];
    static ohevc_tu_job jobs[NB];
    srand(7);
    for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) want[y][x] = plane[y][x] = rand() & 255;
    memset(coeffs, 0, sizeof(coeffs));
    for (int b = 0; b < NB; b++) {
        int dc = (rand() % 4001) - 2000;
        coeffs[b][0] = (int16_t)dc;
        jobs[b].x = (b % 16) * N; jobs[b].y = (b / 16) * N; jobs[b].plane = 0; jobs[b].coeff_off = b * N * N;
        /* a DC-only block: every residual sample is ((((dc*64+64)>>7)*64 + 2048) >> 12) after both stages */
        int s1 = (dc * 64 + 64) >> 7; if (s1 > 32767) s1 = 32767; if (s1 < -32768) s1 = -32768;
        int r = (s1 * 64 + 2048) >> 12;
        for (int y = 0; y < N; y++) for (int x = 0; x < N; x++) {
            int v = want[jobs[b].y + y][jobs[b].x + x] + r;
            want[jobs[b].y + y][jobs[b].x + x] = v < 0 ? 0 : v > 255 ? 255 : v;
        }
    }
    void *d_plane, *d_coeffs, *d_jobs;
    CHECK(hipMalloc(&d_plane, sizeof(plane))); CHECK(hipMalloc(&d_coeffs, sizeof(coeffs))); CHECK(hipMalloc(&d_jobs, sizeof(jobs)));
    CHECK(hipMemcpy(d_plane, plane, sizeof(plane), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_coeffs, coeffs, sizeof(coeffs), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_jobs, jobs, sizeof(jobs), hipMemcpyHostToDevice));
    ohevc_plane planes[3] = { { d_plane, W, W, H }, { 0 }, { 0 } };
    int rc = ohevc_dev_tu_batch(planes, 8, 4, OHEVC_TU_IDCT, (const ohevc_tu_job *)d_jobs, NB, (const int16_t *)d_coeffs, NULL);
    if (rc != OHEVC_OK) { printf("tu_batch rc=%d: %s\n", rc, ohevc_last_error()); return 4; }
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(plane, d_plane, sizeof(plane), hipMemcpyDeviceToHost));
    int bad = 0;
    for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) bad += plane[y][x] != want[y][x];
    printf("c host smoke: %s (%d mismatches), %s\n", bad ? "FAIL" : "ok", bad, ohevc_version());
    return bad ? 1 : 0;
}
