/* abort_trace.c -- TEST INFRASTRUCTURE (loaded by tests/conftest.py and bench.py, never by the product).
 *
 * A fatal signal raised by native code (abort() from glibc's heap checks, from the HIP / ROCr runtimes' fault handlers, from
 * libstdc++'s terminate; SIGSEGV / SIGBUS from a wild pointer) ends a test run with nothing but Python's "Fatal Python error:
 * Aborted" -- faulthandler knows the Python frames only.  This handler writes the NATIVE backtrace of the thread that raised the
 * signal (which library called abort) to a descriptor duplicated from the real stderr at install time, then hands over to the
 * handler that was installed before it (faulthandler's, which dumps the Python frames and re-raises).
 * Round 4's one unexplained SIGABRT is why: its log could not say which runtime had aborted. */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

static struct sigaction g_prev[65];
static int g_fd = 2;
static volatile sig_atomic_t g_busy;

static void put(const char *s) { (void)!write(g_fd, s, strlen(s)); }

static void on_fatal(int sig, siginfo_t *si, void *ctx)
{
    if (!g_busy) {
        g_busy = 1;
        void *bt[48];
        put("\n[abort_trace] fatal signal ");
        put(sig == SIGABRT ? "SIGABRT" : sig == SIGSEGV ? "SIGSEGV" : sig == SIGBUS ? "SIGBUS" : sig == SIGILL ? "SIGILL" : "SIGFPE");
        put(": native frames of the raising thread (innermost first)\n");
        int n = backtrace(bt, 48);
        backtrace_symbols_fd(bt, n, g_fd);
        put("[abort_trace] end of native frames\n");
    }
    struct sigaction *p = &g_prev[sig];
    if ((p->sa_flags & SA_SIGINFO) && p->sa_sigaction) {
        p->sa_sigaction(sig, si, ctx); /* faulthandler: dumps the Python frames, restores ITS predecessor and re-raises */
        return;
    }
    if (!(p->sa_flags & SA_SIGINFO) && p->sa_handler != SIG_DFL && p->sa_handler != SIG_IGN) {
        p->sa_handler(sig);
        return;
    }
    signal(sig, SIG_DFL);
    raise(sig);
}

/* Installs the handler in front of whatever handles SIGABRT / SIGSEGV / SIGBUS / SIGILL / SIGFPE now.  `fd`: where to write
 * (duplicated, so a later redirection of fd 2 does not take the trace with it).  Returns 0. */
int abort_trace_install(int fd)
{
    static int installed;
    if (installed) return 0;
    int d = dup(fd);
    if (d >= 0) g_fd = d;
    void *warm[4];
    (void)backtrace(warm, 4); /* loads libgcc's unwinder now: no dlopen / malloc inside the handler */
    const int sigs[] = {SIGABRT, SIGSEGV, SIGBUS, SIGILL, SIGFPE};
    for (unsigned i = 0; i < sizeof sigs / sizeof *sigs; ++i) {
        struct sigaction sa;
        memset(&sa, 0, sizeof sa);
        sa.sa_sigaction = on_fatal;
        sa.sa_flags = SA_SIGINFO | SA_NODEFER | SA_ONSTACK;
        sigemptyset(&sa.sa_mask);
        if (sigaction(sigs[i], &sa, &g_prev[sigs[i]]) != 0) return -1;
    }
    installed = 1;
    return 0;
}
