// AutoImageTokenGenerationProcessor + greedy argmax as one device routine.
//   reference: src/models_clm/generation.py:19-31 (processor) and transformers==4.34.0 greedy
//   search as driven from src/models_clm/models.py:146-153.
// Literal semantics, in the model dtype like the reference (the processor edits the fp16/bf16
// logits in place):
//   last id in img_ids[:-1]  -> scores[successor] = round_T(max(scores) + 10)
//   otherwise                -> scores[img_ids[1:]] = 0.0      (assignment of zero, NOT -inf)
//   token = first index of the maximum.
// Must be called by all threads of ONE block (any multiple of 64 threads <= 1024).
#pragma once
#include "ss_common.h"

namespace ss {

struct ArgMax { float v; int i; };

__device__ __forceinline__ ArgMax argmax_better(ArgMax a, ArgMax b) {
    return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a;
}

// block-wide arg max over logits[0..vocab); smem: >= 2*16 words
template <typename T>
__device__ __forceinline__ ArgMax block_argmax(const T* logits, int vocab, float* sv, int* si) {
    ArgMax best{-INFINITY, 0x7fffffff};
    for (int i = threadIdx.x; i < vocab; i += blockDim.x) {
        const float v = Tr<T>::ld(logits + i);
        if (v > best.v) { best.v = v; best.i = i; }  // strict: keeps the first index per thread
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ArgMax other{__shfl_xor(best.v, o, 64), __shfl_xor(best.i, o, 64)};
        best = argmax_better(best, other);
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) { sv[wid] = best.v; si[wid] = best.i; }
    __syncthreads();
    ArgMax r{sv[0], si[0]};
    for (int w = 1; w < nw; ++w) r = argmax_better(r, ArgMax{sv[w], si[w]});
    return r;
}

// Returns the greedy token (same value in every thread).  Edits `logits` in place.
template <typename T>
__device__ __forceinline__ int imgproc_argmax_block(T* logits, int vocab, int last_id, const int32_t* img_ids,
                                                    int n_img_ids, float* sv, int* si) {
    __shared__ int s_succ;
    if (threadIdx.x == 0) {
        int succ = -1;
        for (int j = 0; j + 1 < n_img_ids; ++j)
            if (img_ids[j] == last_id) { succ = img_ids[j + 1]; break; }  // list.index: first match
        s_succ = succ;
    }
    __syncthreads();
    const int succ = s_succ;
    if (succ >= 0) {
        const ArgMax mx = block_argmax<T>(logits, vocab, sv, si);
        if (threadIdx.x == 0) Tr<T>::st(logits + succ, mx.v + 10.0f);
    } else {
        for (int j = 1 + threadIdx.x; j < n_img_ids; j += blockDim.x) Tr<T>::st(logits + img_ids[j], 0.0f);
    }
    __threadfence_block();
    __syncthreads();
    return block_argmax<T>(logits, vocab, sv, si).i;
}

}  // namespace ss
