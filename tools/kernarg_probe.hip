// Measurement aid: does a large by-value kernel argument lengthen back-to-back dependent launches?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Big { double v[224]; int n; };                 // ~1.8 KB like the stage-1 arguments
struct Small { const Big* p; int n; };
__global__ void kbig(Big a, double* out) { if (threadIdx.x == 0) out[blockIdx.x] = a.v[blockIdx.x & 127] + a.v[200] + a.n; }
__global__ void ksmall(Small a, double* out) { if (threadIdx.x == 0) out[blockIdx.x] = a.p->v[blockIdx.x & 127] + a.p->v[200] + a.n; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    double* out; hipMalloc(&out, 8 * 4096);
    Big hb; for (int i = 0; i < 224; ++i) hb.v[i] = i; hb.n = 1;
    Big* db; hipMalloc(&db, sizeof(Big)); hipMemcpy(db, &hb, sizeof(Big), hipMemcpyHostToDevice);
    Small hs{db, 1};
    hipStream_t s; hipStreamCreate(&s);
    const int reps = 2000;
    for (int grid : {8, 448, 1200}) for (int mode = 0; mode < 2; ++mode) for (int warm = 0; warm < 2; ++warm) {
        hipStreamSynchronize(s);
        double t0 = now();
        for (int r = 0; r < reps; ++r) {
            if (mode == 0) hipLaunchKernelGGL(kbig, dim3(grid), dim3(256), 0, s, hb, out);
            else hipLaunchKernelGGL(ksmall, dim3(grid), dim3(256), 0, s, hs, out);
        }
        hipStreamSynchronize(s);
        if (warm) printf("grid %4d %s: %.2f us per launch\n", grid, mode ? "pointer to device-resident arguments" : "1.8 KB by value", (now() - t0) / reps * 1e6);
    }
    return 0;
}
