// runtime.hip -- library management entry points of libohevc_hip.so (error text, device selection).
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "common.hpp"

namespace ohevc {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace ohevc

namespace ohevc {
const Config &config()
{
    static const Config cfg = [] {
        Config c;
        auto env = [](const char *name) { const char *v = getenv(name); return v && v[0] ? v : nullptr; };
        if (const char *v = env("OHEVC_REF_WAIT_SECONDS")) if (atoi(v) > 0) c.ref_wait_seconds = atoi(v);
        if (const char *v = env("OHEVC_PREWARM_KIB")) c.prewarm_kib = atoi(v);
        if (const char *v = env("OHEVC_PICTURE_BATCH")) c.picture_batch = atoi(v);
        if (const char *v = env("OHEVC_PARK_THREADS")) if (atoi(v) >= 0 && atoi(v) <= 16) c.park_threads = atoi(v);
        if (const char *v = env("OHEVC_ISSUER_THREADS")) if (atoi(v) >= 1 && atoi(v) <= 16) c.issuer_threads = atoi(v);
        c.frames_token = env("OHEVC_FRAMES_TOKEN");
        if (const char *v = env("OHEVC_TRACE")) {
            // whole comma-separated words only, every occurrence tried ("ctbdebug,ctb" has both; the first strstr hit alone would miss "ctb")
            auto find = [v](const char *w, bool prefix) -> const char * {
                const size_t n = strlen(w);
                for (const char *p = strstr(v, w); p; p = strstr(p + 1, w))
                    if ((p == v || p[-1] == ',') && (prefix || p[n] == 0 || p[n] == ',')) return p;
                return nullptr;
            };
            auto has = [&](const char *w) { return find(w, false) != nullptr; };
            c.trace_order = has("order"); c.trace_timing = has("timing"); c.trace_ctb = has("ctb"); c.trace_levels = has("levels");
            c.trace_launches = has("launches"); c.trace_sao = has("sao"); c.trace_reg = has("reg"); c.trace_upload = has("upload"); c.trace_pin = has("pin"); c.profile_slots = has("slots"); c.ctb_debug = has("ctbdebug");
            if (const char *at = find("at=", true)) if (sscanf(at + 3, "%d:%d:%d", &c.trace_at[0], &c.trace_at[1], &c.trace_at[2]) != 3) c.trace_at[0] = -1;
        }
        return c;
    }();
    return cfg;
}
}  // namespace ohevc

extern "C" const char *ohevc_last_error(void) { return ohevc::g_err; }

extern "C" int ohevc_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int ohevc_set_device(int device)
{
    using namespace ohevc;
    OHEVC_HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    OHEVC_HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
        return OHEVC_ERR_NODEV;
    }
    return OHEVC_OK;
}

extern "C" const char *ohevc_version(void) { return "ohevc_hip 0.1 (gfx950)"; }

// ---- coefficients cross the bus compact, the kernels index a dense arena (include/ohevc_hip.h: ohevc_expand_rec) ------------------------------
// One wavefront per record; a lane writes 16-byte pieces of the dense block: coefficients where the record's rectangle covers them, zeros
// elsewhere.  A record is at most 1024 elements (a 32 x 32 block / a run of whole small blocks): two pieces per lane.  Pure HBM traffic:
// reads the compact stream once, writes the dense arena once (the H2D copy used to write exactly that much).
static __global__ __launch_bounds__(256) void expand_coeffs_kernel(const int16_t *__restrict__ compact, const ohevc_expand_rec *__restrict__ recs, int nrecs,
                                                                    int16_t *__restrict__ dense)
{
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (w >= nrecs) return;
    const ohevc_expand_rec e = recs[w];
    ohevc::u32x4 *out = reinterpret_cast<ohevc::u32x4 *>(dense + e.dst);                  // (dense offsets are whole blocks: 32-byte aligned)
    if (e.kind == 0) {
        const ohevc::u32x2 *in = reinterpret_cast<const ohevc::u32x2 *>(compact + e.src);  // (compact offsets are multiples of 4 elements: 8-byte aligned)
        for (unsigned k = lane; k < e.dims / 8u; k += 64u) {
            const ohevc::u32x2 a = in[2 * k], b = in[2 * k + 1];
            out[k] = ohevc::u32x4{ a.x, a.y, b.x, b.y };
        }
        return;
    }
    if (e.kind & 0x100u) {                                 // sub-block form: dims = one bit per 4x4 group of the region, the set ones travel (16 elements each)
        const unsigned log2n = e.kind & 0xffu, n = 1u << log2n, part = e.kind >> 9, mask = e.dims;
        const unsigned total = (log2n == 5u && part != 2u) ? 512u : n * n, gshift = log2n - 2u;      // elements this record writes; log2 of the groups per row
        const int16_t *in = compact + e.src;
        for (unsigned k = lane; k < total / 8u; k += 64u) {
            const unsigned row = (8u * k) >> log2n, col = (8u * k) & (n - 1u);
            const unsigned gi = ((row >> 2) << gshift) + (col >> 2);           // the piece = row (row & 3) of groups gi and gi + 1
            ohevc::u32x2 a = ohevc::u32x2{ 0u, 0u }, b = ohevc::u32x2{ 0u, 0u };
            if (gi < 32u) {
                const unsigned below = mask & ((1u << gi) - 1u), two = mask >> gi;
                const unsigned at = __builtin_popcount(below) * 16u + (row & 3u) * 4u;
                if (two & 1u) a = *reinterpret_cast<const ohevc::u32x2 *>(in + at);
                if (two & 2u) b = *reinterpret_cast<const ohevc::u32x2 *>(in + at + ((two & 1u) ? 16u : 0u));
            }
            out[k] = ohevc::u32x4{ a.x, a.y, b.x, b.y };
        }
        return;
    }
    const unsigned log2n = e.kind, n = 1u << log2n, cols = e.dims & 0xffu, rows = e.dims >> 8;
    for (unsigned k = lane; k < (n * n) / 8u; k += 64u) {
        const unsigned row = (8u * k) >> log2n, col = (8u * k) & (n - 1u);
        ohevc::u32x2 a = ohevc::u32x2{ 0u, 0u }, b = ohevc::u32x2{ 0u, 0u };
        if (row < rows) {
            const int16_t *src = compact + e.src + row * cols + col;
            if (col < cols) a = *reinterpret_cast<const ohevc::u32x2 *>(src);
            if (col + 4u < cols) b = *reinterpret_cast<const ohevc::u32x2 *>(src + 4);
        }
        out[k] = ohevc::u32x4{ a.x, a.y, b.x, b.y };
    }
}

extern "C" int ohevc_dev_expand_coeffs(const int16_t *compact, const ohevc_expand_rec *recs, int nrecs, int16_t *dense, void *stream)
{
    using namespace ohevc;
    OHEVC_REQUIRE(nrecs >= 0 && (nrecs == 0 || (compact != nullptr && recs != nullptr && dense != nullptr)), "null argument");
    if (nrecs == 0) return OHEVC_OK;
    hipLaunchKernelGGL(expand_coeffs_kernel, dim3((unsigned)((nrecs + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream), compact, recs, nrecs, dense);
    OHEVC_HIP_TRY(hipGetLastError());
    return OHEVC_OK;
}
