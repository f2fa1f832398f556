/* debugging aid: LD_PRELOAD this to get a native backtrace on SIGABRT / SIGSEGV (no gdb on the GPU boxes) */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>
static void on_sig(int s)
{
    void *bt[64];
    int n = backtrace(bt, 64);
    const char msg[] = "\n== native backtrace ==\n";
    write(2, msg, sizeof msg - 1);
    backtrace_symbols_fd(bt, n, 2);
    signal(s, SIG_DFL);
    raise(s);
}
__attribute__((constructor)) static void init(void)
{
    signal(SIGABRT, on_sig);
    signal(SIGSEGV, on_sig);
}
