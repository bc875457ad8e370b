// Probe: which CUs does a CU-masked stream (hipExtStreamCreateWithCUMask) really use on MI355X?  A grid of many short
// workgroups records (XCC_ID, SE, CU) of the CU it ran on; the host counts the distinct ones for several mask shapes, and
// times a fixed amount of ALU work (perfectly parallel over workgroups) as a measure of the CUs actually available.
// Build: hipcc --offload-arch=gfx950 -O3 -o cu_mask cu_mask.hip ; diagnostic only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
#include <vector>

__global__ void where(unsigned *out, int spin) {
    float x = threadIdx.x;
    for (int i = 0; i < spin; ++i) x = __builtin_fmaf(x, 1.0001f, 0.5f);
    if (threadIdx.x == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20);       // XCC_ID
        const unsigned hw = __builtin_amdgcn_s_getreg(((16 - 1) << 11) | (0 << 6) | 4);        // HW_ID: cu [11:8], sh [12], se [15:13]
        out[blockIdx.x] = (xcc << 16) | (hw & 0xff00) | (x == 12345.f);
    }
}

static void run(const char *name, const std::vector<int> &cus, unsigned *d) {
    uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int c : cus) mask[c / 32] |= 1u << (c % 32);
    hipStream_t s;
    if (hipExtStreamCreateWithCUMask(&s, 8, mask) != hipSuccess) { printf("%s: stream creation failed\n", name); return; }
    const int blocks = 8192;
    hipLaunchKernelGGL(where, dim3(blocks), dim3(256), 0, s, d, 20000);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, s);
    hipLaunchKernelGGL(where, dim3(blocks), dim3(256), 0, s, d, 20000);
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned> h(blocks);
    hipMemcpy(h.data(), d, blocks * 4, hipMemcpyDeviceToHost);
    std::set<unsigned> distinct, xccs;
    for (unsigned v : h) { distinct.insert(v & ~1u); xccs.insert(v >> 16); }
    printf("%-44s mask bits %3zu -> distinct CUs used %3zu on %zu XCDs, %8.1f us for the fixed work\n", name, cus.size(), distinct.size(), xccs.size(), ms * 1e3);
    hipStreamDestroy(s);
}

int main() {
    unsigned *d; hipMalloc(&d, 8192 * 4);
    std::vector<int> all, first28, first32, first64, every9, lo128, hi128, spread80;
    for (int c = 0; c < 256; ++c) {
        all.push_back(c);
        if (c < 28) first28.push_back(c);
        if (c < 32) first32.push_back(c);
        if (c < 64) first64.push_back(c);
        if (c % 9 == 0) every9.push_back(c);
        if (c < 128) lo128.push_back(c); else hi128.push_back(c);
        if ((c * 80) / 256 != ((c + 1) * 80) / 256) spread80.push_back(c);
    }
    run("all 256", all, d);
    run("bits 0..127", lo128, d);
    run("bits 128..255", hi128, d);
    run("bits 0..63", first64, d);
    run("bits 0..31", first32, d);
    run("bits 0..27", first28, d);
    run("every 9th bit (29 bits)", every9, d);
    run("80 bits spread evenly", spread80, d);
    return 0;
}
