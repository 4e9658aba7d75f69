/* Diagnostics only (never loaded by the product or by a default bench run): an LD_PRELOAD interposer that counts
 * ioctl() calls per (thread, request) and their duration, so that `bench.py` can say whether -- and from which thread --
 * its timed window entered the kernel driver (KFD / amdgpu).
 *     gcc -O2 -shared -fPIC -o libioctl_trace.so ioctl_trace.c -ldl
 *     LD_PRELOAD=tools/microbench/libioctl_trace.so python bench.py ...   -> config.ioctl_calls / ioctl_max_us / ioctl_top */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>

static int (*real_ioctl)(int, unsigned long, void *);
static volatile uint64_t g_calls, g_total_ns, g_max_ns, g_max_req;
#define SLOTS 64
static struct { volatile uint64_t key, calls, ns; } g_tab[SLOTS];   /* key = tid << 32 | (request & 0xffffffff) */

static uint64_t now_ns(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
}

int ioctl(int fd, unsigned long req, ...) {
  va_list ap;
  va_start(ap, req);
  void *arg = va_arg(ap, void *);
  va_end(ap);
  if (!real_ioctl) real_ioctl = (int (*)(int, unsigned long, void *))dlsym(RTLD_NEXT, "ioctl");
  uint64_t t0 = now_ns();
  int r = real_ioctl(fd, req, arg);
  uint64_t dt = now_ns() - t0;
  __atomic_fetch_add(&g_calls, 1, __ATOMIC_RELAXED);
  __atomic_fetch_add(&g_total_ns, dt, __ATOMIC_RELAXED);
  if (dt > g_max_ns) { g_max_ns = dt; g_max_req = req; }
  uint64_t key = ((uint64_t)(uint32_t)syscall(SYS_gettid) << 32) | (uint32_t)req;
  for (unsigned i = 0, h = (unsigned)(key * 0x9E3779B97F4A7C15ull >> 58); i < SLOTS; ++i, h = (h + 1) % SLOTS) {
    uint64_t k = g_tab[h].key;
    if (k == 0) {
      uint64_t zero = 0;
      if (__atomic_compare_exchange_n(&g_tab[h].key, &zero, key, 0, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) k = key;
      else k = g_tab[h].key;
    }
    if (k == key) {
      __atomic_fetch_add(&g_tab[h].calls, 1, __ATOMIC_RELAXED);
      __atomic_fetch_add(&g_tab[h].ns, dt, __ATOMIC_RELAXED);
      break;
    }
  }
  return r;
}

void ioctl_trace_reset(void) {
  g_calls = g_total_ns = g_max_ns = g_max_req = 0;
  memset((void *)g_tab, 0, sizeof g_tab);
}
void ioctl_trace_read(uint64_t out[4]) { out[0] = g_calls; out[1] = g_total_ns; out[2] = g_max_ns; out[3] = g_max_req; }

/* "comm[tid] req=0x.. calls=.. us=..; ..." of the busiest (thread, request) pairs since the last reset */
int ioctl_trace_top(char *buf, int len) {
  int off = 0;
  for (int round = 0; round < 4 && off < len - 96; ++round) {
    int best = -1;
    for (int i = 0; i < SLOTS; ++i)
      if (g_tab[i].key && g_tab[i].calls && (best < 0 || g_tab[i].calls > g_tab[best].calls)) best = i;
    if (best < 0) break;
    unsigned tid = (unsigned)(g_tab[best].key >> 32);
    char path[64], comm[32] = "?";
    snprintf(path, sizeof path, "/proc/self/task/%u/comm", tid);
    FILE *f = fopen(path, "r");
    if (f) { if (fgets(comm, sizeof comm, f)) comm[strcspn(comm, "\n")] = 0; fclose(f); }
    off += snprintf(buf + off, len - off, "%s[%u] req=0x%x calls=%llu us=%.0f; ", comm, tid, (unsigned)g_tab[best].key,
                    (unsigned long long)g_tab[best].calls, g_tab[best].ns / 1e3);
    g_tab[best].calls = 0;
  }
  return off;
}
