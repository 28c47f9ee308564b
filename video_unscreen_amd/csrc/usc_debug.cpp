// Diagnostic aid (off unless USC_SEGV_BACKTRACE is set when the library loads): on SIGSEGV print the native call stack to
// stderr, then hand over to the handler that was installed before (Python's faulthandler under pytest).  Used to find which
// runtime call a crash sits in when the only evidence is a Python-level trace ending in a ctypes call.
#include <execinfo.h>
#include <signal.h>
#include <stdlib.h>
#include <unistd.h>

namespace {
struct sigaction g_prev;

void on_segv(int, siginfo_t*, void*) {
    void* frames[96];
    const int n = backtrace(frames, 96);
    static const char msg[] = "\n[usc] SIGSEGV - native backtrace of the faulting thread:\n";
    (void)!write(2, msg, sizeof(msg) - 1);
    backtrace_symbols_fd(frames, n, 2);
    sigaction(SIGSEGV, &g_prev, nullptr);          // returning re-executes the faulting instruction under the previous handler
}

}  // namespace

// (Re-)arm the handler: called when the library loads and again in front of the graph launches, because runtimes loaded later
// (HSA, Python's faulthandler) install their own handlers on top.
extern "C" void usc_debug_arm_segv_backtrace(void) {
    static const bool on = getenv("USC_SEGV_BACKTRACE") != nullptr;
    if (!on) return;
    struct sigaction cur;
    sigaction(SIGSEGV, nullptr, &cur);
    if ((cur.sa_flags & SA_SIGINFO) && cur.sa_sigaction == on_segv) return;
    void* warm[2];
    (void)backtrace(warm, 2);                      // loads libgcc now, not inside the handler
    static char* alt = nullptr;                    // the handler runs on its own stack: a stack overflow is a SIGSEGV too
    if (!alt) {
        alt = (char*)malloc(1 << 18);
        stack_t ss;
        ss.ss_sp = alt; ss.ss_size = 1 << 18; ss.ss_flags = 0;
        sigaltstack(&ss, nullptr);
    }
    struct sigaction sa;
    sa.sa_sigaction = on_segv;
    sigemptyset(&sa.sa_mask);
    sa.sa_flags = SA_SIGINFO | SA_ONSTACK;
    sigaction(SIGSEGV, &sa, &g_prev);
}

namespace {
__attribute__((constructor)) void usc_arm_at_load() { usc_debug_arm_segv_backtrace(); }
}  // namespace
