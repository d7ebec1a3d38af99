// nus_fatal_trace.cpp -- nus_install_fatal_trace(): what a process that dies inside native code leaves behind.
//
// A fatal signal raised by native code (abort() from glibc's heap checks, from the HIP / ROCr runtimes' fault handlers, from
// libstdc++'s terminate; SIGSEGV / SIGBUS from a wild pointer) otherwise ends a run with the signal's name and nothing else.
// The handler writes, to a descriptor duplicated from `fd` at install time, using async-signal-safe calls only:
//   1. the NATIVE backtrace of the thread that raised the signal (which library called abort);
//   2. the host ranges this library has registered with the runtime, allocated as pinned memory or hinted (nus_ranges.hpp),
//      live ones and the last 128 events;
//   3. /proc/self/maps -- so that an address in the runtime's own last words ("Memory access fault by GPU node-2 ... on address
//      0x5b7cd2d8f000", round 5) can be placed: heap, a thread arena, an anonymous mapping, a file;
// then hands over to the handler that was installed before it (Python's faulthandler in the harness processes, which dumps the
// Python frames and re-raises; the default action otherwise).
// Off unless a process asks for it (tests/conftest.py, bench.py, __graft_entry__.smoke() do).
#include "../../include/nuscaler_hip.h"

#include "nus_ranges.hpp"

#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

namespace {

struct sigaction g_prev[65];
int g_fd = 2;
volatile sig_atomic_t g_busy = 0;
bool g_installed = false;

void put(const char *s) { (void)!write(g_fd, s, strlen(s)); }

void put_hex(uint64_t v)
{
    char buf[19] = "0x";
    int n = 2;
    bool lead = true;
    for (int shift = 60; shift >= 0; shift -= 4) {
        const unsigned d = (unsigned)((v >> shift) & 15);
        if (lead && d == 0 && shift != 0) continue;
        lead = false;
        buf[n++] = (char)(d < 10 ? '0' + d : 'a' + d - 10);
    }
    buf[n] = 0;
    put(buf);
}

void put_dec(uint64_t v)
{
    char buf[24];
    int n = 23;
    buf[n] = 0;
    do {
        buf[--n] = (char)('0' + v % 10);
        v /= 10;
    } while (v);
    put(buf + n);
}

const char *kind_name(uint32_t k)
{
    return k == nus::kRangePinned ? "pinned-by-caller (nus_host_pin)" : k == nus::kRangeHostAlloc ? "hipHostMalloc (library)" :
           k == nus::kRangeHugeHint ? "MADV_HUGEPAGE hint" : "?";
}

void put_record(const nus::RangeRecord &r, bool history)
{
    put("  #");
    put_dec(r.seq);
    put(" ");
    if (history) put(r.op ? "note   " : "forget ");
    put_hex(r.lo);
    put("-");
    put_hex(r.hi);
    put(" (");
    put_dec((uint64_t)(r.hi - r.lo));
    put(" bytes) ");
    put(kind_name(r.kind));
    put("\n");
}

void dump_ranges()
{
    static nus::RangeRecord recs[256]; // static: the handler may run on a small alternate stack
    size_t n = nus::range_live_snapshot(recs, 256);
    put("[nus_fatal_trace] host ranges the library holds now: ");
    put_dec(n);
    put(nus::range_overflowed() ? " (table overflowed earlier: incomplete)\n" : "\n");
    for (size_t i = 0; i < n; ++i) put_record(recs[i], false);
    n = nus::range_history_snapshot(recs, 128);
    put("[nus_fatal_trace] last ");
    put_dec(n);
    put(" range events, oldest first\n");
    for (size_t i = 0; i < n; ++i) put_record(recs[i], true);
}

void dump_maps()
{
    put("[nus_fatal_trace] /proc/self/maps\n");
    const int f = open("/proc/self/maps", O_RDONLY | O_CLOEXEC);
    if (f < 0) {
        put("  (cannot open)\n");
        return;
    }
    static char buf[1 << 16];
    for (;;) {
        const ssize_t got = read(f, buf, sizeof buf);
        if (got <= 0) break;
        for (ssize_t off = 0; off < got;) {
            const ssize_t w = write(g_fd, buf + off, (size_t)(got - off));
            if (w <= 0) {
                off = got;
                break;
            }
            off += w;
        }
    }
    close(f);
    put("[nus_fatal_trace] end of /proc/self/maps\n");
}

void on_fatal(int sig, siginfo_t *si, void *ctx)
{
    if (!g_busy) {
        g_busy = 1;
        void *bt[48];
        put("\n[nus_fatal_trace] fatal signal ");
        put(sig == SIGABRT ? "SIGABRT" : sig == SIGSEGV ? "SIGSEGV" : sig == SIGBUS ? "SIGBUS" : sig == SIGILL ? "SIGILL" : "SIGFPE");
        if ((sig == SIGSEGV || sig == SIGBUS) && si) {
            put(" at address ");
            put_hex((uint64_t)(uintptr_t)si->si_addr);
        }
        put(": native frames of the raising thread (innermost first)\n");
        const int n = backtrace(bt, 48);
        backtrace_symbols_fd(bt, n, g_fd);
        put("[nus_fatal_trace] end of native frames\n");
        dump_ranges();
        dump_maps();
    }
    struct sigaction *p = &g_prev[sig];
    if ((p->sa_flags & SA_SIGINFO) && p->sa_sigaction) {
        p->sa_sigaction(sig, si, ctx); // faulthandler: dumps the Python frames, restores ITS predecessor and re-raises
        return;
    }
    if (!(p->sa_flags & SA_SIGINFO) && p->sa_handler != SIG_DFL && p->sa_handler != SIG_IGN) {
        p->sa_handler(sig);
        return;
    }
    signal(sig, SIG_DFL);
    raise(sig);
}

} // namespace

extern "C" int nus_install_fatal_trace(int fd)
{
    if (g_installed) return NUS_OK;
    const int d = dup(fd);
    if (d >= 0) g_fd = d;
    void *warm[4];
    (void)backtrace(warm, 4); // loads libgcc's unwinder now: no dlopen / malloc inside the handler
    const int sigs[] = {SIGABRT, SIGSEGV, SIGBUS, SIGILL, SIGFPE};
    for (unsigned i = 0; i < sizeof sigs / sizeof *sigs; ++i) {
        struct sigaction sa;
        memset(&sa, 0, sizeof sa);
        sa.sa_sigaction = on_fatal;
        sa.sa_flags = SA_SIGINFO | SA_NODEFER | SA_ONSTACK;
        sigemptyset(&sa.sa_mask);
        if (sigaction(sigs[i], &sa, &g_prev[sigs[i]]) != 0) return NUS_ERR_INVALID_ARGUMENT;
    }
    g_installed = true;
    return NUS_OK;
}

extern "C" size_t nus_host_ranges(nus_host_range *out, size_t cap, int history)
{
    static_assert(sizeof(nus_host_range) == sizeof(nus::RangeRecord), "nus_host_range mirrors nus::RangeRecord");
    if (!out || cap == 0) return 0;
    nus::RangeRecord *r = reinterpret_cast<nus::RangeRecord *>(out);
    return history ? nus::range_history_snapshot(r, cap) : nus::range_live_snapshot(r, cap);
}
