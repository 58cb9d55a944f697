// LAB PROGRAM: which XCD does workgroup (x, y) of a 2-D grid run on?  hipcc -O3 --offload-arch=gfx950 -o build/xcc_probe tools/xcc_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_probe(uint32_t* out) {
    if (threadIdx.x == 0) out[blockIdx.y * gridDim.x + blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15;  // HW_REG_XCC_ID
}
int main() {
    for (auto dims : {std::pair<int, int>{960, 64}, {641, 64}, {24, 8}}) {
        const int gx = dims.first, gy = dims.second;
        uint32_t* d;
        hipMalloc(&d, gx * gy * 4);
        for (int rep = 0; rep < 2; rep++) {
            hipLaunchKernelGGL(k_probe, dim3(gx, gy), dim3(256), 0, 0, d);
            std::vector<uint32_t> h(gx * gy);
            hipMemcpy(h.data(), d, gx * gy * 4, hipMemcpyDeviceToHost);
            int ok_lin = 0, ok_x = 0;
            for (int i = 0; i < gx * gy; i++) ok_lin += h[i] == (uint32_t)(i % 8), ok_x += h[i] == (uint32_t)((i % gx) % 8);
            printf("grid %d x %d rep %d: xcc == linear %% 8 for %d of %d, xcc == x %% 8 for %d; first 24:", gx, gy, rep, ok_lin, gx * gy, ok_x);
            for (int i = 0; i < 24; i++) printf(" %u", h[i]);
            printf("\n");
        }
        hipFree(d);
    }
    return 0;
}
