/* cpuprof.c -- a sampling CPU profiler to preload next to the shim (diagnostics only, not part of the product):
 *   gcc -O2 -fPIC -shared -o /tmp/libcpuprof.so tools/cpuprof.c -ldl
 *   BMH_CPU_PROFILE=/tmp/prof.txt LD_PRELOAD=/tmp/libcpuprof.so:.../libbwamem_hip_dropin.so bwa mem ...
 * Every thread gets its own CPU-time timer (a process-wide ITIMER_PROF mostly hits the main thread): it ticks once per
 * millisecond of CPU the thread burns and the tick records the thread's call stack.  At exit every frame is written as
 * "<object file> <offset> <symbol>" for tools/cpuprof_report.py to resolve. */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include <sys/syscall.h>
#include <sys/time.h>
#include <time.h>
#include <unistd.h>

enum { MAX_SAMPLES = 400000, DEPTH = 24 };
static void *g_frames[MAX_SAMPLES][DEPTH];
static unsigned char g_depth[MAX_SAMPLES];
static volatile int g_n;
static const char *g_out;

static void on_tick(int sig, siginfo_t *si, void *uc)
{
	int k;
	(void)sig, (void)si, (void)uc;
	k = __sync_fetch_and_add(&g_n, 1);
	if (k >= MAX_SAMPLES) return;
	g_depth[k] = (unsigned char)backtrace(g_frames[k], DEPTH);
}

static void arm_this_thread(void)
{
	struct sigevent sev;
	struct itimerspec its;
	clockid_t cid;
	timer_t t;
	memset(&sev, 0, sizeof(sev));
	sev.sigev_notify = SIGEV_THREAD_ID, sev.sigev_signo = SIGPROF;
	sev._sigev_un._tid = (pid_t)syscall(SYS_gettid);
	if (pthread_getcpuclockid(pthread_self(), &cid) || timer_create(cid, &sev, &t)) return;
	its.it_interval.tv_sec = 0, its.it_interval.tv_nsec = 1000000, its.it_value = its.it_interval;
	timer_settime(t, 0, &its, 0);
}

typedef struct { void *(*fn)(void *); void *arg; } start_t;
static void *thread_entry(void *p)
{
	start_t s = *(start_t *)p;
	free(p);
	if (g_out) arm_this_thread();
	return s.fn(s.arg);
}

int pthread_create(pthread_t *th, const pthread_attr_t *attr, void *(*fn)(void *), void *arg)
{
	static int (*real)(pthread_t *, const pthread_attr_t *, void *(*)(void *), void *);
	start_t *s;
	if (!real) real = (int (*)(pthread_t *, const pthread_attr_t *, void *(*)(void *), void *))dlsym(RTLD_NEXT, "pthread_create");
	if (!g_out) return real(th, attr, fn, arg);
	s = (start_t *)malloc(sizeof(*s));
	s->fn = fn, s->arg = arg;
	return real(th, attr, thread_entry, s);
}

__attribute__((constructor)) static void prof_start(void)
{
	struct sigaction sa;
	void *warm[4];
	g_out = getenv("BMH_CPU_PROFILE");
	if (!g_out) return;
	backtrace(warm, 4); /* loads the unwinder now, not inside the first signal */
	memset(&sa, 0, sizeof(sa));
	sa.sa_sigaction = on_tick, sa.sa_flags = SA_SIGINFO | SA_RESTART;
	sigaction(SIGPROF, &sa, 0);
	arm_this_thread();
}

__attribute__((destructor)) static void prof_stop(void)
{
	FILE *f;
	int k, j, n;
	if (!g_out) return;
	signal(SIGPROF, SIG_IGN);
	n = g_n < MAX_SAMPLES ? g_n : MAX_SAMPLES;
	if (!(f = fopen(g_out, "w"))) return;
	fprintf(f, "# %d samples (1 ms of process CPU time each)\n", n);
	for (k = 0; k < n; ++k) {
		for (j = 2; j < g_depth[k]; ++j) { /* 0, 1 = this handler and the signal trampoline */
			Dl_info di;
			if (dladdr(g_frames[k][j], &di) && di.dli_fname)
				fprintf(f, "%s %lx %s;", di.dli_fname, (unsigned long)((char *)g_frames[k][j] - (char *)di.dli_fbase), di.dli_sname ? di.dli_sname : "?");
			else fprintf(f, "? %lx ?;", (unsigned long)g_frames[k][j]);
		}
		fputc('\n', f);
	}
	fclose(f);
}
