// Timing experiments on the register-chained, panel-streamed fp1 + head kernel (encoder_reg.hip::fp_s_kernel) with synthetic data:
// compile with -DLSDM_X_NOBAR / -DLSDM_X_NOEPI / -DLSDM_X_NOFETCH to drop the panel barrier + commit, the ReLU + split epilogue,
// the weight fetch.  argv[1] = gather mode (0 random rows, 1 sequential rows, 2 one row per wave).
#include "../../lsdm_amd/csrc/encoder_reg.hip"
#include <cstdio>
#include <vector>
int main(int argc, char** argv) {
    const int dbg = argc > 1 ? atoi(argv[1]) : 0;
    const int nc = 576;
    const size_t npts = (size_t)nc * 1024;
    float *p2, *nnw, *out, *bias;
    int* nni;
    __bf16* blob;
    (void)hipMalloc(&p2, npts * 128 * 4); (void)hipMemset(p2, 0, npts * 128 * 4);
    (void)hipMalloc(&nnw, npts * 3 * 4); (void)hipMemset(nnw, 0, npts * 3 * 4);
    (void)hipMalloc(&out, npts * 3 * 4);
    (void)hipMalloc(&bias, 1024 * 4); (void)hipMemset(bias, 0, 1024 * 4);
    (void)hipMalloc(&blob, 17 * 16384); (void)hipMemset(blob, 0, 17 * 16384);
    std::vector<int> h(npts * 3);
    srand(1);
    for (auto& v : h) v = rand() % 1024;
    (void)hipMalloc(&nni, npts * 3 * 4); (void)hipMemcpy(nni, h.data(), npts * 3 * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        for (int i = 0; i < 5; ++i)
            hipLaunchKernelGGL((lsdm::fp_s_kernel<8, 2>), dim3(256), dim3(512), 0, 0, p2, nni, nnw, 1024, 1024, (long long)npts, blob, bias, out, dbg);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep == 2) printf("fp_s_kernel<8,2> gather mode %d: %.1f us per launch\n", dbg, ms * 1000 / 5);
    }
    return 0;
}
