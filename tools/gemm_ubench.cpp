// Standalone GEMM / conv micro-benchmark over the C ABI (no torch: starts in a second on a fresh GPU box).
//
//   hipcc -O2 --offload-arch=gfx950 tools/gemm_ubench.cpp -o tools/bin/gemm_ubench -ldl
//   tools/bin/gemm_ubench <lib.so> <case> ...      case = M,N,K,epi[,res]:cfg[/swz],cfg[/swz],...
//                                                   conv = cB,H,W,Cin,Cout,stride,up[,rowvec[,res]]:cfg,...
//
// For every case: operands are pseudo-random (the chip clocks by its power budget: zero-filled operands run ~20 % faster),
// every cfg is checked against the first cfg of the list (max |diff| relative to max |ref|) and timed in SUSTAINED mode
// (back-to-back launches over rotating weight copies, one event pair), rounds interleaved over the cfgs; prints min / median.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

typedef int (*gemm_fn)(const void*, const void*, void*, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, const void*, const void*,
                       int64_t, int, int, void*);
typedef int (*conv_fn)(const void*, const void*, void*, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, const void*,
                       const void*, int64_t, const void*, int, void*);
typedef int (*tune_fn)(const char*, int);
typedef const char* (*err_fn)(void);

__global__ void fill(uint16_t* p, size_t n, uint32_t seed, float scale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 2654435761u + seed;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
        const float f = ((float)(x >> 8) * (1.0f / 8388608.0f) - 1.0f) * scale;
        uint32_t u = __float_as_uint(f);
        u += 0x7fffu + ((u >> 16) & 1u);
        p[i] = (uint16_t)(u >> 16);
    }
}
__global__ void maxdiff(const uint16_t* a, const uint16_t* b, size_t n, float* out) {   // out[0] = max|a-b|, out[1] = max|b|
    float d = 0.f, m = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float x = __uint_as_float((uint32_t)a[i] << 16), y = __uint_as_float((uint32_t)b[i] << 16);
        d = fmaxf(d, fabsf(x - y));
        m = fmaxf(m, fabsf(y));
        if (x != x) d = 1e30f;
    }
    atomicMax((int*)out, __float_as_int(d));
    atomicMax((int*)out + 1, __float_as_int(m));
    unsigned nd = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) nd += a[i] != b[i];
    if (nd) atomicAdd((unsigned*)out + 2, nd);          // out[2] = number of elements that are not bit-equal
}

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s lib.so case...\n", argv[0]); return 2; }
    void* h = dlopen(argv[1], RTLD_NOW);
    if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
    gemm_fn ss_gemm = (gemm_fn)dlsym(h, "ss_gemm");
    conv_fn ss_conv = (conv_fn)dlsym(h, "ss_conv3x3");
    tune_fn set_tuning = (tune_fn)dlsym(h, "ss_set_tuning");
    err_fn last_error = (err_fn)dlsym(h, "ss_last_error");
    if (!ss_gemm || !ss_conv || !set_tuning) { fprintf(stderr, "missing symbols\n"); return 2; }
    if (const char* kn = getenv("UBENCH_KNOB")) {      // "name=int[,name=int...]": extra tuning knobs for the whole run
        std::string all = kn;
        size_t pos = 0;
        while (pos < all.size()) {
            size_t e = all.find(',', pos);
            if (e == std::string::npos) e = all.size();
            std::string kv = all.substr(pos, e - pos);
            const size_t eq = kv.find('=');
            if (eq != std::string::npos) set_tuning(kv.substr(0, eq).c_str(), atoi(kv.c_str() + eq + 1));
            pos = e + 1;
        }
    }
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float* dstat;
    CK(hipMalloc(&dstat, 16));
    const float wscale = getenv("UBENCH_WSCALE") ? (float)atof(getenv("UBENCH_WSCALE")) : 0.05f;
    const int nscreen = getenv("UBENCH_SCREEN") ? atoi(getenv("UBENCH_SCREEN")) : 1;   // race screen: repeat the check
    const int BF16 = 1;   // SS_BF16
    for (int a = 2; a < argc; ++a) {
        std::string arg = argv[a];
        const bool conv = arg[0] == 'c';
        const size_t colon = arg.find(':');
        std::vector<long> dims;
        {
            std::string d = arg.substr(conv ? 1 : 0, colon - (conv ? 1 : 0));
            char* p = &d[0];
            while (*p) { dims.push_back(strtol(p, &p, 10)); if (*p == ',') ++p; }
        }
        std::vector<std::pair<int, int>> cfgs;
        {
            std::string c = arg.substr(colon + 1);
            char* p = &c[0];
            while (*p) {
                int cfg = (int)strtol(p, &p, 10), swz = 8;
                if (*p == '/') { ++p; swz = (int)strtol(p, &p, 10); }
                cfgs.push_back({cfg, swz});
                if (*p == ',') ++p;
            }
        }
        int64_t M, N, K, B = 0, H = 0, W = 0, Cin = 0, Cout = 0, stride = 1, up = 0;
        int epi = 0, use_res = 0, use_rv = 0;
        size_t a_elems;
        if (conv) {
            B = dims[0]; H = dims[1]; W = dims[2]; Cin = dims[3]; Cout = dims[4]; stride = dims[5]; up = dims[6];
            use_rv = dims.size() > 7 ? (int)dims[7] : 0; use_res = dims.size() > 8 ? (int)dims[8] : 0;
            const int64_t Hin = up ? 2 * H : H, Win = up ? 2 * W : W;
            const int64_t Ho = (Hin + 2 - 3) / stride + 1, Wo = (Win + 2 - 3) / stride + 1;
            M = B * Ho * Wo; N = Cout; K = 9 * Cin;
            a_elems = (size_t)B * H * W * Cin;
        } else {
            M = dims[0]; N = dims[1]; K = dims[2]; epi = dims.size() > 3 ? (int)dims[3] : 0; use_res = dims.size() > 4 ? (int)dims[4] : 0;
            a_elems = (size_t)M * K;
        }
        const bool geglu = (epi & 16) != 0;    // SS_EPI_GEGLU_PAIR
        const int64_t ldc = geglu ? N / 2 : N;
        const size_t w_elems = (size_t)N * K, c_elems = (size_t)M * ldc;
        size_t nW = ((size_t)352 << 20) / (w_elems * 2);
        if (nW > 64) nW = 64;
        if (nW < 2) nW = 2;
        uint16_t *dA, *dW, *dC, *dRef, *dBias, *dRes;
        CK(hipMalloc(&dA, a_elems * 2));
        CK(hipMalloc(&dW, nW * w_elems * 2));
        CK(hipMalloc(&dC, c_elems * 2));
        CK(hipMalloc(&dRef, c_elems * 2));
        CK(hipMalloc(&dBias, (size_t)N * 2));
        CK(hipMalloc(&dRes, (c_elems + (size_t)(B + 1) * N) * 2));
        fill<<<2048, 256, 0, s>>>(dA, a_elems, 0x1234u, 1.0f);
        fill<<<2048, 256, 0, s>>>(dW, nW * w_elems, 0x9876u, wscale);
        fill<<<64, 256, 0, s>>>(dBias, (size_t)N, 0x55u, 0.5f);
        fill<<<2048, 256, 0, s>>>(dRes, c_elems + (size_t)(B + 1) * N, 0x77u, 1.0f);
        CK(hipStreamSynchronize(s));
        const int epi_full = epi | 1 | (use_res ? 4 : 0);    // bias always; SS_EPI_RESIDUAL = 4
        auto launch = [&](int cfg, int swz, const uint16_t* w, uint16_t* c) -> int {
            set_tuning("gemm_cfg", cfg);
            set_tuning("gemm_xcd_swizzle", swz);
            int rc;
            if (conv) rc = ss_conv(dA, w, c, B, H, W, Cin, Cout, stride, up, dBias, use_rv ? dRes : nullptr, use_rv ? Cout : 0, use_res ? dRes + (size_t)B * Cout : nullptr, BF16, s);
            else rc = ss_gemm(dA, w, c, M, N, K, K, K, ldc, dBias, use_res ? dRes : nullptr, ldc, epi_full, BF16, s);
            if (rc) { fprintf(stderr, "launch cfg %d failed: %d %s\n", cfg, rc, last_error ? last_error() : ""); exit(3); }
            return rc;
        };
        const double flops = 2.0 * (double)M * N * K;
        printf("case %s  (M=%ld N=%ld K=%ld, %zu weight copies)\n", arg.c_str(), (long)M, (long)N, (long)K, nW);
        // correctness vs the first cfg
        launch(cfgs[0].first, cfgs[0].second, dW, dRef);
        CK(hipStreamSynchronize(s));
        for (size_t i = 0; i < cfgs.size(); ++i) {
            float worst = 0.f, mref = 0.f;
            unsigned nbad = 0, bad_runs = 0;
            for (int rep = 0; rep < nscreen; ++rep) {
                CK(hipMemsetAsync(dC, 0xff, c_elems * 2, s));
                launch(cfgs[i].first, cfgs[i].second, dW, dC);
                CK(hipMemsetAsync(dstat, 0, 16, s));
                maxdiff<<<1024, 256, 0, s>>>(dC, dRef, c_elems, dstat);
                float st[4];
                CK(hipMemcpyAsync(st, dstat, 16, hipMemcpyDeviceToHost, s));
                CK(hipStreamSynchronize(s));
                unsigned nd;
                memcpy(&nd, &st[2], 4);
                worst = std::max(worst, st[0]); mref = st[1]; nbad = std::max(nbad, nd); bad_runs += nd != 0;
            }
            printf("  cfg %2d/%d  max|diff| %.3e  (max|ref| %.3e, rel %.2e)  not-bit-equal: max %u elements, %u of %d runs%s\n", cfgs[i].first,
                   cfgs[i].second, worst, mref, worst / (mref + 1e-30f), nbad, bad_runs, nscreen, worst / (mref + 1e-30f) > 2e-2f ? "   <<<<<< MISMATCH" : "");
        }
        if (getenv("UBENCH_PMC")) {   // counter collection: a few launches per cfg over rotating weights, no timing loop
            for (size_t i = 0; i < cfgs.size(); ++i)
                for (int r = 0; r < 3; ++r) launch(cfgs[i].first, cfgs[i].second, dW + (size_t)(r % nW) * w_elems, dC);
            CK(hipStreamSynchronize(s));
            CK(hipFree(dA)); CK(hipFree(dW)); CK(hipFree(dC)); CK(hipFree(dRef)); CK(hipFree(dBias)); CK(hipFree(dRes));
            continue;
        }
        // timing: 5 interleaved rounds
        const int R = (int)std::max<size_t>(nW + nW / 2, 12);
        std::vector<std::vector<float>> us(cfgs.size());
        for (int round = 0; round < 5; ++round)
            for (size_t i = 0; i < cfgs.size(); ++i) {
                launch(cfgs[i].first, cfgs[i].second, dW, dC);
                launch(cfgs[i].first, cfgs[i].second, dW + w_elems, dC);
                CK(hipEventRecord(e0, s));
                for (int r = 0; r < R; ++r) launch(cfgs[i].first, cfgs[i].second, dW + (size_t)(r % nW) * w_elems, dC);
                CK(hipEventRecord(e1, s));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                us[i].push_back(ms * 1e3f / R);
            }
        for (size_t i = 0; i < cfgs.size(); ++i) {
            std::sort(us[i].begin(), us[i].end());
            printf("  cfg %2d/%d  min %8.1f us (%7.1f TF)   median %8.1f us (%7.1f TF)\n", cfgs[i].first, cfgs[i].second, us[i][0],
                   flops / us[i][0] * 1e-6, us[i][2], flops / us[i][2] * 1e-6);
        }
        fflush(stdout);
        CK(hipFree(dA)); CK(hipFree(dW)); CK(hipFree(dC)); CK(hipFree(dRef)); CK(hipFree(dBias)); CK(hipFree(dRes));
    }
    return 0;
}
