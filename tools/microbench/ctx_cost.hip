// What does creating a library context cost?  Times the runtime calls bmh_ctx_create / bmh_ctx_reserve_* make, four rounds
// (the first one pays for initialising the runtime).   hipcc --offload-arch=gfx950 -O2 -o /tmp/ctx_cost tools/microbench/ctx_cost.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
	hipSetDevice(0);
	for (int round = 0; round < 4; ++round) {
		double t0 = now();
		hipStream_t s[3];
		for (int i = 0; i < 3; ++i) hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking);
		double t1 = now();
		hipEvent_t ev[40];
		for (int i = 0; i < 40; ++i) hipEventCreate(&ev[i]);
		double t2 = now();
		void *d_err, *h_err;
		hipMalloc(&d_err, 4);
		hipMemset(d_err, 0, 4);
		double t3 = now();
		hipHostMalloc(&h_err, 4, hipHostMallocDefault);
		double t4 = now();
		void *hp[2], *dp[4];
		for (int i = 0; i < 2; ++i) hipHostMalloc(&hp[i], 16 << 20, hipHostMallocDefault);
		double t5 = now();
		for (int i = 0; i < 4; ++i) hipMalloc(&dp[i], 16 << 20);
		double t6 = now();
		hipMemsetAsync(dp[0], 0, 1024, s[0]);
		hipStreamSynchronize(s[0]);
		double t7 = now();
		printf("round %d: 3 streams %.2f ms, 40 events %.2f ms, hipMalloc+memset(4 B) %.2f ms, hipHostMalloc(4 B) %.2f ms, 2 x hipHostMalloc(16 MB) %.2f ms, "
		       "4 x hipMalloc(16 MB) %.2f ms, first use of a stream %.2f ms\n", round, t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5, t7 - t6);
	}
	return 0;
}
