#include <hip/hip_runtime.h>
__global__ void k(int *o) {
    int x = threadIdx.x * 3 + 1;
    int a = __builtin_amdgcn_update_dpp(-1, x, 0x130, 0xf, 0xf, false);
    int b = __builtin_amdgcn_update_dpp(-1, x, 0x138, 0xf, 0xf, false);
    o[threadIdx.x * 2] = a; o[threadIdx.x * 2 + 1] = b;
}
int main() { int *d; hipMalloc(&d, 512); k<<<1, 64>>>(d); int h[128]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  for (int i : {0, 1, 31, 32, 62, 63}) printf("lane %d: x=%d next=%d prev=%d\n", i, i * 3 + 1, h[2 * i], h[2 * i + 1]); return 0; }
