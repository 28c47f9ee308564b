// Diagnostic aid (off unless USC_SEGV_BACKTRACE is set): on SIGSEGV print, with async-signal-safe calls only, the faulting
// address, the program counter and the /proc/self/maps line of the module the program counter lies in (enough for
// `llvm-addr2line` / `nm` afterwards), then try a libgcc backtrace, then hand over to the handler installed before (Python's
// faulthandler under pytest).  The handler runs on its own stack, so a stack overflow is reported too.
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>
#include <unistd.h>

namespace {
struct sigaction g_prev;

void put(const char* s) { (void)!write(2, s, strlen(s)); }
void put_hex(uint64_t v) {
    char b[19] = "0x0000000000000000";
    for (int i = 0; i < 16; i++) { const int d = (v >> (60 - 4 * i)) & 15; b[2 + i] = (char)(d < 10 ? '0' + d : 'a' + d - 10); }
    (void)!write(2, b, 18);
}
uint64_t parse_hex(const char*& p) {
    uint64_t v = 0;
    for (;; p++) {
        const char c = *p;
        if (c >= '0' && c <= '9') v = v * 16 + (uint64_t)(c - '0');
        else if (c >= 'a' && c <= 'f') v = v * 16 + (uint64_t)(c - 'a' + 10);
        else return v;
    }
}
// print the maps line(s) whose range holds `addr`
void put_mapping(uint64_t addr) {
    const int fd = open("/proc/self/maps", O_RDONLY);
    if (fd < 0) return;
    static char buf[1 << 16];
    static char line[512];
    int ll = 0;
    for (;;) {
        const ssize_t n = read(fd, buf, sizeof(buf));
        if (n <= 0) break;
        for (ssize_t i = 0; i < n; i++) {
            if (buf[i] != '\n' && ll < (int)sizeof(line) - 2) { line[ll++] = buf[i]; continue; }
            line[ll] = 0;
            const char* p = line;
            const uint64_t lo = parse_hex(p);
            if (*p == '-') { p++; const uint64_t hi = parse_hex(p); if (addr >= lo && addr < hi) { put("    "); put(line); put("\n"); } }
            ll = 0;
        }
    }
    close(fd);
}

void on_segv(int, siginfo_t* si, void* ctx) {
    const ucontext_t* uc = reinterpret_cast<const ucontext_t*>(ctx);
    const uint64_t pc = (uint64_t)uc->uc_mcontext.gregs[REG_RIP], sp = (uint64_t)uc->uc_mcontext.gregs[REG_RSP];
    put("\n[usc] SIGSEGV: fault address "); put_hex((uint64_t)si->si_addr);
    put("  pc "); put_hex(pc); put("  sp "); put_hex(sp); put("\n  pc is in:\n");
    put_mapping(pc);
    put("  sp is in:\n");
    put_mapping(sp);
    // return addresses found on the first 64 stack words that point into executable mappings are not resolved here; the libgcc
    // unwinder is tried last because it may fault itself on a damaged stack
    put("  backtrace:\n");
    void* frames[64];
    const int n = backtrace(frames, 64);
    backtrace_symbols_fd(frames, n, 2);
    sigaction(SIGSEGV, &g_prev, nullptr);          // returning re-executes the faulting instruction under the previous handler
}
}  // namespace

// (Re-)arm the handler: called when the library loads and again in front of the graph launches, because runtimes loaded later
// (HSA, Python's faulthandler) install their own handlers on top.  sigaltstack is per thread: armed for the calling thread.
extern "C" void usc_debug_arm_segv_backtrace(void) {
    static const bool on = getenv("USC_SEGV_BACKTRACE") != nullptr;
    if (!on) return;
    static thread_local char* alt = nullptr;
    if (!alt) {
        alt = (char*)malloc(1 << 18);
        stack_t ss;
        ss.ss_sp = alt; ss.ss_size = 1 << 18; ss.ss_flags = 0;
        sigaltstack(&ss, nullptr);
    }
    struct sigaction cur;
    sigaction(SIGSEGV, nullptr, &cur);
    if ((cur.sa_flags & SA_SIGINFO) && cur.sa_sigaction == on_segv) return;
    void* warm[2];
    (void)backtrace(warm, 2);                      // loads libgcc now, not inside the handler
    struct sigaction sa;
    sa.sa_sigaction = on_segv;
    sigemptyset(&sa.sa_mask);
    sa.sa_flags = SA_SIGINFO | SA_ONSTACK;
    sigaction(SIGSEGV, &sa, &g_prev);
}

namespace {
__attribute__((constructor)) void usc_arm_at_load() { usc_debug_arm_segv_backtrace(); }
}  // namespace
