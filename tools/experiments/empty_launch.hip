// empty_launch.hip -- what does a kernel that returns at once cost behind another kernel, by grid / block / LDS size?
// (timing experiment of round 3: the second launch of the headline path; build: hipcc --offload-arch=gfx950 -O2 -o empty_launch empty_launch.hip)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void busy(float *p, int n) {
    float a = p[threadIdx.x];
    for (int i = 0; i < n; ++i) a = a * 1.0001f + 0.5f;
    p[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
__global__ void empty(const int *flag, float *p) {
    extern __shared__ float lds[];
    if (*flag) { lds[threadIdx.x] = 1.0f; p[threadIdx.x] = lds[threadIdx.x ^ 1]; }
}
int main() {
    float *p; int *flag;
    hipMalloc(&p, 1 << 24); hipMalloc(&flag, 4); hipMemset(flag, 0, 4); hipMemset(p, 0, 1 << 24);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int cfg[][3] = {{1, 64, 0}, {256, 256, 0}, {256, 1024, 0}, {256, 1024, 158000}, {128, 1024, 158000}, {64, 1024, 158000},
                          {512, 512, 75000}, {2048, 1024, 20000}, {8192, 1024, 20000}, {8192, 256, 20000}};
    for (auto &c : cfg) {
        hipFuncSetAttribute((const void *)empty, hipFuncAttributeMaxDynamicSharedMemorySize, c[2]);
        float best_pair = 1e9f, best_one = 1e9f;
        for (int rep = 0; rep < 20; ++rep) {
            hipEventRecord(e0);
            for (int k = 0; k < 50; ++k) hipLaunchKernelGGL(busy, dim3(256), dim3(256), 0, 0, p, 2000);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float t1; hipEventElapsedTime(&t1, e0, e1);
            hipEventRecord(e0);
            for (int k = 0; k < 50; ++k) {
                hipLaunchKernelGGL(busy, dim3(256), dim3(256), 0, 0, p, 2000);
                hipLaunchKernelGGL(empty, dim3(c[0]), dim3(c[1]), c[2], 0, flag, p);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float t2; hipEventElapsedTime(&t2, e0, e1);
            if (t1 < best_one) best_one = t1;
            if (t2 < best_pair) best_pair = t2;
        }
        printf("grid %5d x %4d threads, %6d B LDS: +%.2f us per empty launch (busy kernel alone %.2f us)\n", c[0], c[1], c[2],
               (best_pair - best_one) / 50 * 1e3, best_one / 50 * 1e3);
    }
    return 0;
}
