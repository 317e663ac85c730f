/* Test-side helper (never part of the product): prints the C stack of the faulting thread when the
 * process receives SIGSEGV / SIGBUS / SIGABRT / SIGFPE / SIGILL, then re-raises with the default
 * action.  tools/stress/stress_multi.py loads it with ctypes and calls crash_bt_install(tag).
 * Lines are "module(+offset)" pairs: resolve them with tools/stress/symbolize.py.
 *
 *   gcc -O1 -g -fPIC -shared tools/stress/crash_bt.c -o tools/stress/libcrashbt.so
 */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <sys/syscall.h>
#include <unistd.h>

static char g_tag[64] = "?";

static void put(const char* s) { (void)!write(2, s, strlen(s)); }

static void handler(int sig, siginfo_t* info, void* uctx)
{
    (void)uctx;
    char line[256];
    snprintf(line, sizeof line, "\n=== crash_bt [%s] pid %d tid %ld: signal %d (%s) fault address %p ===\n",
             g_tag, (int)getpid(), (long)syscall(SYS_gettid), sig,
             sig == SIGSEGV ? "SIGSEGV" : sig == SIGABRT ? "SIGABRT" : sig == SIGBUS ? "SIGBUS" : "other",
             info ? info->si_addr : (void*)0);
    put(line);
    void* frames[96];
    int n = backtrace(frames, 96);
    backtrace_symbols_fd(frames, n, 2);
    put("=== end of crash_bt ===\n");
    signal(sig, SIG_DFL);
    raise(sig);
}

int crash_bt_install(const char* tag)
{
    if (tag) { strncpy(g_tag, tag, sizeof g_tag - 1); g_tag[sizeof g_tag - 1] = 0; }
    void* warm[4];
    (void)backtrace(warm, 4);          /* loads libgcc now, not inside the handler */
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = handler;
    sa.sa_flags = SA_SIGINFO | SA_NODEFER;
    sigemptyset(&sa.sa_mask);
    int sigs[] = {SIGSEGV, SIGBUS, SIGABRT, SIGFPE, SIGILL};
    for (unsigned i = 0; i < sizeof sigs / sizeof sigs[0]; ++i)
        if (sigaction(sigs[i], &sa, 0)) return -1;
    return 0;
}
