// EXPERIMENT harness for csrc/lifter_chain.hip: the res-block chain on random operands, per-phase shader-clock stamps of wave 0 of block 0
// (RC_STAMP) and the launch time at a few batch sizes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DRC_STAMPS -I contextaware-poseformer_amd/csrc -I include -o tools/ab/res_chain tools/res_chain_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__device__ unsigned long long rc_stamps[64];
#define RC_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) rc_stamps[i] = __builtin_readcyclecounter(); } while (0)
#include "../contextaware-poseformer_amd/csrc/lifter_chain.hip"
using namespace capf;
int main(int argc, char** argv) {
    const int C = 128, NB = 4;
    std::vector<ResBlockW> blk(NB);
    auto dev = [&](size_t n, float scale, bool pack) {
        std::vector<float> h(n);
        for (auto& v : h) v = scale * ((rand() & 0xFFFF) / 32768.0f - 1.0f);
        if (pack) { unsigned short* p = reinterpret_cast<unsigned short*>(h.data()); for (size_t i = 0; i < 2 * n; ++i) p[i] = (unsigned short)(0x3000 + (rand() & 0x7FF)); }
        float* d; hipMalloc(&d, n * 4); hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice); return d;
    };
    auto packw = [&](int N, int K) {          // [N][K floats] of fp16 pairs, then [N] inverse scales
        float* d = dev((size_t)N * K + N, 1.f, true);
        std::vector<float> inv(N, 1.0f / 1024.f);
        hipMemcpy(d + (size_t)N * K, inv.data(), N * 4, hipMemcpyHostToDevice);
        return d;
    };
    for (int i = 0; i < NB; ++i)
        blk[i] = ResBlockW{dev(C, 1, false), dev(C, .1f, false), packw(3 * C, C), dev(3 * C, .1f, false), packw(C, C), dev(C, .1f, false),
                           dev(C, 1, false), dev(C, .1f, false), packw(2 * C, C), dev(2 * C, .1f, false), packw(C, 2 * C), dev(C, .1f, false)};
    for (int B : {1, 8, 64, 128, 512}) {
        const int rows = B * 85;
        float* X = dev((size_t)rows * C, 1.f, false);
        for (int i = 0; i < 3; ++i) launch_res_chain(X, rows, 5, 8, 1e-6f, blk.data(), NB, 0);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int reps = 50;
        hipEventRecord(e0);
        for (int i = 0; i < reps; ++i) launch_res_chain(X, rows, 5, 8, 1e-6f, blk.data(), NB, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long st[64];
        hipMemcpyFromSymbol(st, HIP_SYMBOL(rc_stamps), sizeof(st));
        printf("batch %4d: %7.1f us per launch;  block 0 wave 0, cycles: load X %llu |", B, ms * 1e3 / reps, st[1] - st[0]);
        const char* nm[] = {"ln1", "qkv", "qkv-ep", "attn", "proj", "ln2", "fc1", "fc2"};
        for (int k = 0; k < 8; ++k) printf(" %s %llu", nm[k], st[2 + k] - st[1 + k]);
        printf(" | whole launch (block 0) %llu\n", st[40] - st[0]);
        hipFree(X);
    }
    return 0;
}
