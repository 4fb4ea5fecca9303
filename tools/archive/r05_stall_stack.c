/* tools/r05_stall_stack.c -- a watchdog for the one-in-ten slow step: a thread that, when the main thread has been
 * inside one timed step for longer than a limit, signals it; the handler writes the main thread's native stack
 * (module + offset per frame, resolvable on this image) and the time into a file.  Diagnosis only. */
#define _GNU_SOURCE
#include <execinfo.h>
#include <fcntl.h>
#include <pthread.h>
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

static volatile int64_t g_step_start_ns = 0; /* 0 = not inside a step */
static volatile int g_fired = 0;
static volatile int g_stop = 0;
static int g_fd = -1;
static int64_t g_limit_ns = 8000000;
static int64_t g_repeat_ns = 5000000;
static pthread_t g_main, g_dog;

static int64_t now_ns(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (int64_t)ts.tv_sec * 1000000000ll + ts.tv_nsec;
}

static void handler(int sig)
{
  (void)sig;
  void* frames[64];
  int n = backtrace(frames, 64);
  char head[96];
  int64_t t0 = g_step_start_ns;
  int len = snprintf(head, sizeof head, "--- sample at +%.2f ms of the step\n", t0 ? (now_ns() - t0) / 1e6 : -1.0);
  if (g_fd >= 0) {
    (void)!write(g_fd, head, len);
    backtrace_symbols_fd(frames, n, g_fd);
  }
}

static void* dog(void* arg)
{
  (void)arg;
  int64_t last = 0;
  while (!g_stop) {
    int64_t t0 = g_step_start_ns;
    int64_t t = now_ns();
    if (t0 && t - t0 > g_limit_ns && t - last > g_repeat_ns && g_fired < 12) {
      last = t;
      g_fired++;
      pthread_kill(g_main, SIGUSR2);
    }
    /* busy-ish poll: a sleeping watchdog would share the timer granularity under suspicion */
    for (volatile int i = 0; i < 2000; i++) {}
  }
  return 0;
}

int dog_start(const char* path, double limit_ms, double repeat_ms)
{
  g_fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
  g_limit_ns = (int64_t)(limit_ms * 1e6);
  g_repeat_ns = (int64_t)(repeat_ms * 1e6);
  g_main = pthread_self();
  void* warm[4];
  backtrace(warm, 4); /* loads libgcc outside the handler */
  struct sigaction sa;
  memset(&sa, 0, sizeof sa);
  sa.sa_handler = handler;
  sa.sa_flags = SA_RESTART;
  sigaction(SIGUSR2, &sa, 0);
  return pthread_create(&g_dog, 0, dog, 0);
}

void dog_step_begin(void) { g_fired = 0; g_step_start_ns = now_ns(); }
void dog_step_end(void) { g_step_start_ns = 0; }
void dog_stop(void) { g_stop = 1; pthread_join(g_dog, 0); if (g_fd >= 0) close(g_fd); }

/* how long does a 100 us sleep take here? (timer granularity of the box) */
double sleep_probe_us(int reps)
{
  struct timespec rq = {0, 100000};
  int64_t t0 = now_ns();
  for (int i = 0; i < reps; i++) nanosleep(&rq, 0);
  return (now_ns() - t0) / 1e3 / reps;
}

/* A thread that only reads the clock: every gap above `thresh_ms` between two reads (the thread was not running) is
 * recorded as (start in s on CLOCK_MONOTONIC, length in ms).  Runs until `seconds` are over; returns the count. */
int spin_gaps(double seconds, double thresh_ms, double* out, int cap)
{
  int64_t t_end = now_ns() + (int64_t)(seconds * 1e9);
  int64_t prev = now_ns();
  int n = 0;
  while (prev < t_end) {
    int64_t t = now_ns();
    if (t - prev > (int64_t)(thresh_ms * 1e6) && n < cap) {
      out[2 * n] = prev / 1e9;
      out[2 * n + 1] = (t - prev) / 1e6;
      n++;
    }
    prev = t;
  }
  return n;
}
