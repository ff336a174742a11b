// LD_PRELOAD sampling profiler: SIGPROF at 1 kHz of process CPU time, leaf + 3 callers per sample, resolved with dladdr at exit.
// For the CPU box only (the hooks recording without a device, DIAG_KIND=hipemu OHHIP_RECORD_ONLY=1): with the real HIP runtime loaded and 16
// decoding threads the process hung on the device box (backtrace() in a signal handler against the runtime's own threads) and took the
// visit's whole time limit with it - do not preload it there.
#define _GNU_SOURCE
#include <dlfcn.h>
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>
#include <ucontext.h>
#define MAXS (1 << 20)
static void *g_pc[MAXS][4];
static volatile int g_n;
static void on_prof(int sig, siginfo_t *si, void *uc)
{
    (void)sig; (void)si;
    int i = __atomic_fetch_add(&g_n, 1, __ATOMIC_RELAXED);
    if (i >= MAXS) return;
    void *bt[8];
    int n = backtrace(bt, 8);
    ucontext_t *u = (ucontext_t *)uc;
    g_pc[i][0] = (void *)u->uc_mcontext.gregs[REG_RIP];
    for (int k = 1; k < 4; k++) g_pc[i][k] = (k + 2 < n) ? bt[k + 2] : NULL;      // skip handler + sigreturn frames
}
__attribute__((constructor)) static void start(void)
{
    if (!getenv("SAMPLER_OUT")) return;
    void *bt[4]; backtrace(bt, 4);                                                 // load libgcc now, not in the handler
    struct sigaction sa; memset(&sa, 0, sizeof(sa));
    sa.sa_sigaction = on_prof; sa.sa_flags = SA_SIGINFO | SA_RESTART;
    sigaction(SIGPROF, &sa, NULL);
    struct itimerval it = { { 0, 1000 }, { 0, 1000 } };
    setitimer(ITIMER_PROF, &it, NULL);
}
__attribute__((destructor)) static void stop(void)
{
    const char *out = getenv("SAMPLER_OUT");
    if (!out) return;
    struct itimerval it = { { 0, 0 }, { 0, 0 } };
    setitimer(ITIMER_PROF, &it, NULL);
    FILE *f = fopen(out, "w");
    int n = g_n < MAXS ? g_n : MAXS;
    for (int i = 0; i < n; i++) {
        for (int k = 0; k < 4; k++) {
            Dl_info di; memset(&di, 0, sizeof(di));
            if (g_pc[i][k] && dladdr(g_pc[i][k], &di) && di.dli_fname)
                fprintf(f, "%s+0x%lx(%s)%s", strrchr(di.dli_fname, '/') ? strrchr(di.dli_fname, '/') + 1 : di.dli_fname,
                        (unsigned long)((char *)g_pc[i][k] - (char *)di.dli_fbase), di.dli_sname ? di.dli_sname : "?", k < 3 ? "\t" : "");
            else fprintf(f, "?%s", k < 3 ? "\t" : "");
        }
        fputc('\n', f);
    }
    fclose(f);
}
