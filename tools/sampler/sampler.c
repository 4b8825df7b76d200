// A per-thread PC sampler for the calling thread of an end-to-end run (tools/profile_host.py): a POSIX timer (wall clock: CPU-time timers tick at 4 ms)
// delivers SIGPROF to that thread every `period_us`; the handler stores the interrupted PC.
// Measurement aid only; nothing in the product links it.
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/syscall.h>
#include <time.h>
#include <ucontext.h>
#include <unistd.h>

#ifndef SIGEV_THREAD_ID
#define SIGEV_THREAD_ID 4
#endif
#ifndef sigev_notify_thread_id
#define sigev_notify_thread_id _sigev_un._tid
#endif

#define DEPTH 8  // frames kept per sample when SAMPLE_STACKS=1 (glibc backtrace: unwinds through .eh_frame, no frame pointers needed)
static int g_stacks;
static uint64_t* g_pcs;
static volatile size_t g_n, g_cap;
static timer_t g_timer;
static int g_on;

static void on_prof(int sig, siginfo_t* si, void* uc_) {
    (void)sig; (void)si;
    ucontext_t* uc = (ucontext_t*)uc_;
    size_t i = g_n;
    if (i < g_cap) {
        if (g_stacks) {
            void* fr[DEPTH + 3];
            int n = backtrace(fr, DEPTH + 3);  // [0] this handler, [1] the signal trampoline, [2] the interrupted PC, callers after it
            uint64_t* o = g_pcs + i * DEPTH;
            for (int k = 0; k < DEPTH; ++k) o[k] = k + 2 < n ? (uint64_t)fr[k + 2] : 0;
            o[0] = (uint64_t)uc->uc_mcontext.gregs[REG_RIP];
        } else
            g_pcs[i] = (uint64_t)uc->uc_mcontext.gregs[REG_RIP];
        g_n = i + 1;
    }
}

int sampler_start(int period_us, size_t cap) {
    if (g_on) return -1;
    g_stacks = getenv("SAMPLE_STACKS") != NULL;
    if (g_stacks) {
        void* warm[4];
        (void)backtrace(warm, 4);  // the first call loads the unwinder: not inside the handler
    }
    g_pcs = (uint64_t*)malloc(cap * sizeof(uint64_t) * (g_stacks ? DEPTH : 1));
    g_cap = cap; g_n = 0;
    struct sigaction sa; memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = on_prof; sa.sa_flags = SA_SIGINFO | SA_RESTART;
    sigemptyset(&sa.sa_mask);
    if (sigaction(SIGPROF, &sa, NULL)) return -2;
    struct sigevent ev; memset(&ev, 0, sizeof ev);
    ev.sigev_notify = SIGEV_THREAD_ID; ev.sigev_signo = SIGPROF;
    ev.sigev_notify_thread_id = (int)syscall(SYS_gettid);
    if (timer_create(CLOCK_MONOTONIC, &ev, &g_timer)) return -3;
    struct itimerspec its; memset(&its, 0, sizeof its);
    its.it_interval.tv_nsec = (long)period_us * 1000; its.it_value = its.it_interval;
    if (timer_settime(g_timer, 0, &its, NULL)) return -4;
    g_on = 1;
    return 0;
}

long sampler_stop(const char* path) {
    if (!g_on) return -1;
    struct itimerspec its; memset(&its, 0, sizeof its);
    timer_settime(g_timer, 0, &its, NULL);
    timer_delete(g_timer);
    signal(SIGPROF, SIG_IGN);
    g_on = 0;
    FILE* f = fopen(path, "w");
    if (!f) return -2;
    FILE* m = fopen("/proc/self/maps", "r");
    char line[1024];
    while (m && fgets(line, sizeof line, m))
        if (strstr(line, " r-xp ") || strstr(line, " r--p 00000000")) fprintf(f, "M %s", line);
    if (m) fclose(m);
    for (size_t i = 0; i < g_n; ++i) {
        if (!g_stacks) {
            fprintf(f, "S %llx\n", (unsigned long long)g_pcs[i]);
            continue;
        }
        fputs("S", f);
        for (int k = 0; k < DEPTH && g_pcs[i * DEPTH + k]; ++k) fprintf(f, " %llx", (unsigned long long)g_pcs[i * DEPTH + k]);
        fputs("\n", f);
    }
    fclose(f);
    long n = (long)g_n;
    free(g_pcs); g_pcs = NULL;
    return n;
}
