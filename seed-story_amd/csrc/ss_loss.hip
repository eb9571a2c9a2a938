// Loss heads of the training-side forward (forward only — no backward kernels):
//   * token cross-entropy of LlamaForCausalLM.forward(labels=...)   (src/models_clm/modeling_llama_xformer.py:761-772:
//     logits[..., :-1, :] against labels[..., 1:], CrossEntropyLoss() = mean over the labels != -100);
//   * cosine_loss of ContinuousLVLM.forward                          (src/models_clm/models.py:13-17, 78);
//   * F.mse_loss(noise_pred.float(), noise.float()) of SDXLAdapter.forward (src/models_ipa/adapter_modules.py:339).
// All three are bandwidth-trivial next to the forwards that feed them; what matters is that they are DETERMINISTIC
// (fixed-order reductions, no atomics) and round where the reference's torch graph rounds.
#include "ss_common.h"

namespace ss {

// one block per row: loss[r] = logsumexp(logits[r, :]) - logits[r, label[r]] in fp32, valid[r] = 1; label == ignore_index
// -> loss 0, valid 0 (torch's nll_loss ignore_index).  log_softmax on a 16-bit tensor is computed in fp32 and ROUNDED to the
// tensor dtype before nll_loss picks the label's entry (torch CPU / CUDA kernels alike), so the 16-bit path rounds there.
template <typename T>
__global__ __launch_bounds__(256) void cross_entropy_rows_kernel(const T* __restrict__ logits, int64_t ld,
                                                                 const int64_t* __restrict__ labels, int64_t vocab,
                                                                 int64_t ignore_index, float* __restrict__ row_loss,
                                                                 float* __restrict__ row_valid) {
    __shared__ float red[16];
    const int64_t r = blockIdx.x;
    const int64_t lab = labels[r];
    if (lab == ignore_index || lab < 0 || lab >= vocab) {      // (out-of-range labels are rejected on the host)
        if (threadIdx.x == 0) { row_loss[r] = 0.f; row_valid[r] = 0.f; }
        return;
    }
    const T* x = logits + r * ld;
    float m = -INFINITY;
    for (int64_t i = threadIdx.x; i < vocab; i += 256) m = fmaxf(m, Tr<T>::ld(x + i));
    m = block_max(m, red);
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < vocab; i += 256) s += expf(Tr<T>::ld(x + i) - m);
    s = block_sum(s, red);
    if (threadIdx.x == 0) {
        const float lsm = Tr<T>::rnd(Tr<T>::ld(x + lab) - m - logf(s));   // log_softmax entry, rounded to T
        row_loss[r] = -lsm;
        row_valid[r] = 1.f;
    }
}

// one wave per row: val[r] = 1 - sum_k (t/|t|)_k (r/|r|)_k with every intermediate tensor of the reference rounded to T
// (norm, the normalised operands, the product, the row sum, 1 - sum): models.py:13-17
template <typename T>
__global__ __launch_bounds__(256) void cosine_rows_kernel(const T* __restrict__ rec, const T* __restrict__ tgt, int64_t rows,
                                                          int64_t dim, float* __restrict__ val) {
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= rows) return;
    const T* a = rec + r * dim;
    const T* b = tgt + r * dim;
    float na = 0.f, nb = 0.f;
    for (int64_t i = lane; i < dim; i += 64) {
        const float x = Tr<T>::ld(a + i), y = Tr<T>::ld(b + i);
        na = fmaf(x, x, na);
        nb = fmaf(y, y, nb);
    }
    na = Tr<T>::rnd(sqrtf(wave_sum(na)));
    nb = Tr<T>::rnd(sqrtf(wave_sum(nb)));
    float s = 0.f;
    for (int64_t i = lane; i < dim; i += 64) {
        const float x = Tr<T>::rnd(Tr<T>::ld(a + i) / na), y = Tr<T>::rnd(Tr<T>::ld(b + i) / nb);
        s += Tr<T>::rnd(y * x);
    }
    s = Tr<T>::rnd(wave_sum(s));
    if (lane == 0) val[r] = Tr<T>::rnd(1.0f - s);
}

// stage 1 of a fixed-order sum: block k sums f(i) over its contiguous slice [k * per, ...) -> part[k] (fp64)
//   MODE 0: f = vals[i] * (mask ? mask[i] : 1); part2[k] = sum of mask (or count)      (means of row losses)
//   MODE 1: f = (a[i] - b[i])^2 in fp32 of two T tensors                              (mse_loss(a.float(), b.float()))
template <typename T, int MODE>
__global__ __launch_bounds__(256) void sum_stage1_kernel(const void* __restrict__ pa, const void* __restrict__ pb, int64_t n,
                                                         int64_t per, double* __restrict__ part, double* __restrict__ part2) {
    __shared__ double sh[256], sh2[256];
    const int64_t i0 = (int64_t)blockIdx.x * per, i1 = i0 + per < n ? i0 + per : n;
    // thread t sums the contiguous sub-slice [i0 + t * q, ...): fixed order whatever the launch geometry
    const int64_t q = (per + 255) / 256;
    const int64_t j0 = i0 + (int64_t)threadIdx.x * q, j1 = j0 + q < i1 ? j0 + q : i1;
    double s = 0.0, c = 0.0;
    for (int64_t i = j0; i < j1; ++i) {
        if constexpr (MODE == 0) {
            const float m = pb ? ((const float*)pb)[i] : 1.f;
            s += (double)(((const float*)pa)[i] * m);
            c += (double)m;
        } else {
            const float d = Tr<T>::ld((const T*)pa + i) - Tr<T>::ld((const T*)pb + i);
            s += (double)(d * d);
            c += 1.0;
        }
    }
    sh[threadIdx.x] = s;
    sh2[threadIdx.x] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, b = 0.0;
        for (int k = 0; k < 256; ++k) { a += sh[k]; b += sh2[k]; }
        part[blockIdx.x] = a;
        part2[blockIdx.x] = b;
    }
}
// stage 2: out[0] = (sum of part) / (sum of part2) (0 when the denominator is 0: CrossEntropyLoss over no valid label is NaN
// in torch; the host decides what to do with an empty batch), out[1] = the denominator
__global__ void sum_stage2_kernel(const double* __restrict__ part, const double* __restrict__ part2, int nblk, float* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double a = 0.0, b = 0.0;
    for (int k = 0; k < nblk; ++k) { a += part[k]; b += part2[k]; }
    out[0] = b > 0.0 ? (float)(a / b) : 0.f;
    out[1] = (float)b;
}

static int sum_blocks(int64_t n) {
    int64_t nb = (n + 65535) / 65536;
    if (nb < 1) nb = 1;
    if (nb > 1024) nb = 1024;
    return (int)nb;
}

template <typename T>
int cross_entropy_launch(const void* logits, int64_t ld, const int64_t* labels, int64_t rows, int64_t vocab, int64_t ignore_index,
                         float* row_loss, float* row_valid, hipStream_t s) {
    if (rows == 0) return SS_OK;
    hipLaunchKernelGGL(cross_entropy_rows_kernel<T>, dim3((unsigned)rows), dim3(256), 0, s, (const T*)logits, ld, labels, vocab,
                       ignore_index, row_loss, row_valid);
    SS_LAUNCH_CHECK("cross_entropy_rows");
    return SS_OK;
}
template <typename T>
int cosine_launch(const void* rec, const void* tgt, int64_t rows, int64_t dim, float* val, hipStream_t s) {
    if (rows == 0) return SS_OK;
    hipLaunchKernelGGL(cosine_rows_kernel<T>, dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, s, (const T*)rec, (const T*)tgt, rows, dim,
                       val);
    SS_LAUNCH_CHECK("cosine_rows");
    return SS_OK;
}
template <typename T>
int mse_launch(const void* a, const void* b, int64_t n, double* ws, float* out, hipStream_t s) {
    const int nb = sum_blocks(n);
    const int64_t per = (n + nb - 1) / nb;
    hipLaunchKernelGGL((sum_stage1_kernel<T, 1>), dim3(nb), dim3(256), 0, s, a, b, n, per, ws, ws + nb);
    SS_LAUNCH_CHECK("mse_stage1");
    hipLaunchKernelGGL(sum_stage2_kernel, dim3(1), dim3(64), 0, s, (const double*)ws, (const double*)(ws + nb), nb, out);
    SS_LAUNCH_CHECK("mse_stage2");
    return SS_OK;
}

}  // namespace ss

extern "C" {

size_t ss_loss_workspace_bytes(int64_t n) { return (size_t)2 * ss::sum_blocks(n) * sizeof(double); }

int ss_cross_entropy_rows(const void* logits, int64_t ld, const int64_t* labels, int64_t rows, int64_t vocab, int64_t ignore_index,
                          float* row_loss, float* row_valid, int dtype, void* stream) {
    SS_REQUIRE(logits && labels && row_loss && row_valid && vocab > 0 && ld >= vocab, "cross_entropy_rows: bad arguments");
    return SS_DISPATCH(dtype, ss::cross_entropy_launch, logits, ld, labels, rows, vocab, ignore_index, row_loss, row_valid,
                       (hipStream_t)stream);
}

int ss_cosine_rows(const void* rec, const void* target, int64_t rows, int64_t dim, float* row_val, int dtype, void* stream) {
    SS_REQUIRE(rec && target && row_val && dim > 0, "cosine_rows: bad arguments");
    return SS_DISPATCH(dtype, ss::cosine_launch, rec, target, rows, dim, row_val, (hipStream_t)stream);
}

int ss_masked_mean(const float* vals, const float* mask, int64_t n, void* workspace, float* out2, void* stream) {
    SS_REQUIRE(vals && workspace && out2 && n >= 0, "masked_mean: bad arguments");
    const int nb = ss::sum_blocks(n);
    const int64_t per = (n + nb - 1) / nb;
    double* ws = (double*)workspace;
    hipLaunchKernelGGL((ss::sum_stage1_kernel<float, 0>), dim3(nb), dim3(256), 0, (hipStream_t)stream, (const void*)vals, (const void*)mask,
                       n, per, ws, ws + nb);
    SS_LAUNCH_CHECK("masked_mean_stage1");
    hipLaunchKernelGGL(ss::sum_stage2_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const double*)ws, (const double*)(ws + nb), nb,
                       out2);
    SS_LAUNCH_CHECK("masked_mean_stage2");
    return SS_OK;
}

int ss_mse(const void* a, const void* b, int64_t n, void* workspace, float* out2, int dtype, void* stream) {
    SS_REQUIRE(a && b && workspace && out2 && n > 0, "mse: bad arguments");
    return SS_DISPATCH(dtype, ss::mse_launch, a, b, n, (double*)workspace, out2, (hipStream_t)stream);
}

}  // extern "C"
