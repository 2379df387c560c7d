/* Developer aid: native backtrace of EVERY thread of a stuck process. Loaded into the process (ctypes.CDLL) it
 * installs a SIGUSR2 handler that prints the receiving thread's stack to stderr; tools/diag/dump_threads.py sends
 * the signal to each thread (tgkill). For boxes without gdb. */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <sys/syscall.h>
#include <unistd.h>

static void on_signal(int sig) {
    (void)sig;
    void* frames[64];
    const int n = backtrace(frames, 64);
    char head[64];
    const int len = snprintf(head, sizeof head, "\n== native stack of tid %ld\n", (long)syscall(SYS_gettid));
    if (write(2, head, (size_t)len) < 0) return;
    backtrace_symbols_fd(frames, n, 2);
}

__attribute__((constructor)) static void install(void) {
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_handler = on_signal;
    sa.sa_flags = SA_RESTART;
    sigaction(SIGUSR2, &sa, 0);
    void* warm[4];
    backtrace(warm, 4);                 /* loads libgcc's unwinder now, not inside the handler */
}
