// Tuning only: which HIP call blocks the host when host-to-device copies on a copy stream run beside a long kernel on another stream
// and the kernel stream waits for the copy's event?  (round 5, pipelined ingest)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
__global__ void spin(float *p, int iters) {
    float v = p[threadIdx.x];
    for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
    p[threadIdx.x + blockIdx.x * blockDim.x] = v;
}
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0;   // 0: kernel waits for the copy event; 1: no copy event at all; 2: event recorded but not waited for; 3: like 0 without the host-side slot sync; 4: like 0, the slot sync as a hipEventQuery spin
    const size_t nb = 921600;
    unsigned char *h0, *d[3];
    float *buf;
    (void)hipHostMalloc(&h0, 2 * nb);
    for (int i = 0; i < 3; ++i) (void)hipMalloc(&d[i], 2 * nb);
    (void)hipMalloc(&buf, 256 * 1024 * 4);
    hipStream_t K, C;
    (void)hipStreamCreateWithFlags(&K, hipStreamNonBlocking);
    (void)hipStreamCreateWithFlags(&C, hipStreamNonBlocking);
    hipEvent_t copied[3], consumed[3];
    for (int i = 0; i < 3; ++i) { (void)hipEventCreateWithFlags(&copied[i], hipEventDisableTiming); (void)hipEventCreateWithFlags(&consumed[i], hipEventDisableTiming); }
    int iters = 20000;
    for (int rep = 0; rep < 3; ++rep) {   // calibrate the kernel to ~250 us
        double t = now();
        hipLaunchKernelGGL(spin, dim3(256), dim3(1024), 0, K, buf, iters);
        (void)hipStreamSynchronize(K);
        double us = now() - t;
        if (rep == 1) iters = (int)(iters * 250.0 / us);
        if (rep == 2) printf("kernel %.0f us\n", us);
    }
    double tm[6] = {0, 0, 0, 0, 0, 0};
    const int n = 200;
    double t0 = now();
    for (int i = 0; i < n; ++i) {
        const int s = i % 3;
        double a = now();
        if (i >= 3 && mode != 3 && mode != 4) (void)hipEventSynchronize(consumed[s]);
        if (mode == 4 && i >= 3) while (hipEventQuery(consumed[s]) == hipErrorNotReady) { }
        double b = now();
        (void)hipMemcpyAsync(d[s], h0, nb, hipMemcpyHostToDevice, C);
        (void)hipMemcpyAsync(d[s] + nb, h0 + nb, nb, hipMemcpyHostToDevice, C);
        double c = now();
        if (mode != 1) (void)hipEventRecord(copied[s], C);
        double e = now();
        if (mode == 0 || mode >= 3) (void)hipStreamWaitEvent(K, copied[s], 0);
        double f = now();
        hipLaunchKernelGGL(spin, dim3(256), dim3(1024), 0, K, buf, iters);
        double g = now();
        (void)hipEventRecord(consumed[s], K);
        double h = now();
        tm[0] += b - a; tm[1] += c - b; tm[2] += e - c; tm[3] += f - e; tm[4] += g - f; tm[5] += h - g;
    }
    double host = now() - t0;
    (void)hipDeviceSynchronize();
    double wall = now() - t0;
    printf("mode %d: wall %.1f us / step, host %.1f; eventSync %.1f memcpyAsync x2 %.1f recordCopied %.1f waitEvent %.1f launch %.1f recordConsumed %.1f\n", mode, wall / n, host / n,
           tm[0] / n, tm[1] / n, tm[2] / n, tm[3] / n, tm[4] / n, tm[5] / n);
    return 0;
}
