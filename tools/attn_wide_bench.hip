// Diagnostic: the wide-head fused attention kernel (csrc/attn.hip, k_attn_wide) alone on random codes, with the cycle stamps of the
// EDADM_STAMPS build.   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -DEDADM_STAMPS -I include tools/attn_wide_bench.hip -o tools/attn_wide_bench
#include "../eda-dm_amd/csrc/attn.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 100, N = argc > 2 ? atoi(argv[2]) : 1024, d = 384;
    const size_t n = (size_t)B * N * d;
    std::vector<__half> h(n);
    __half *q, *k, *v;
    int8_t* out;
    float* qp;
    hipMalloc(&q, n * 2); hipMalloc(&k, n * 2); hipMalloc(&v, n * 2); hipMalloc(&out, n); hipMalloc(&qp, 64);
    for (int t = 0; t < 3; ++t) {
        for (size_t i = 0; i < n; ++i) h[i] = __float2half((float)((rand() % 241) - 120));
        hipMemcpy(t == 0 ? q : t == 1 ? k : v, h.data(), n * 2, hipMemcpyHostToDevice);
    }
    const float qph[8] = {1.0f / 255.0f, 0.f, 255.f, 0.f, 0.037f, 131.f, 255.f, 0.f};
    hipMemcpy(qp, qph, sizeof(qph), hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const float alpha = 0.03f * 0.031f / sqrtf((float)d);
    auto run = [&]() {
        return edadm_attention_fused_f16(q, d, (int64_t)N * d, d, k, d, (int64_t)N * d, d, v, d, (int64_t)N * d, d, out, d, (int64_t)N * d, B, 1,
                                         N, N, d, alpha, qp, 0.029f / 255.0f, 2, qp + 4, nullptr);
    };
#ifdef EDADM_DIAG
    {
        const int e = getenv("ATTW_EXP") ? atoi(getenv("ATTW_EXP")) : 0;
        hipMemcpyToSymbol(HIP_SYMBOL(g_attw_exp), &e, sizeof(e));
    }
#endif
    int rc = run();
    hipDeviceSynchronize();
    unsigned long long z[16] = {0};
#ifdef EDADM_STAMPS
    hipMemcpyToSymbol(HIP_SYMBOL(g_attw_stamps), z, sizeof(z));
#endif
    hipEventRecord(e0);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) run();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const double flop = 3.0 * 2.0 * B * (double)N * N * d;
    printf("rc %d  B %d N %d d %d: %.3f ms per launch, %.1f TFLOP/s executed (3 products), %.1f algorithmic\n", rc, B, N, d, ms, flop / ms * 1e-9,
           flop / 1.5 / ms * 1e-9);
#ifdef EDADM_STAMPS
    hipMemcpyFromSymbol(z, HIP_SYMBOL(g_attw_stamps), sizeof(z));
    printf("walk 1 (%.0f block visits): lstore + gload %.0f  scores(next) + stats %.0f  barrier %.0f cycles per block\n", (double)z[5],
           z[0] / (double)z[5], z[1] / (double)z[5], z[2] / (double)z[5]);
    printf("walk 2 (%.0f block visits): lstore + gload %.0f  scores(next) + softmax %.0f  pack + PV %.0f  barrier %.0f cycles per block\n",
           (double)z[13], z[8] / (double)z[13], z[9] / (double)z[13], z[10] / (double)z[13], z[11] / (double)z[13]);
#endif
    return 0;
}
