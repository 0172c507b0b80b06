// Multi-tensor Adam step for the training path (the reference: train.py:99-100 torch.optim.Adam(imagen.parameters(), lr),
// stepped in training.py:375-377).  ONE launch updates every parameter of the model: the tensors are described by a device-resident
// table, the grid is a list of fixed-size chunks (tensor index, offset).  Same update as torch.optim.Adam (no amsgrad):
//     m <- m + (g - m) (1 - beta1)            v <- beta2 v + (1 - beta2) g g            [g <- g + weight_decay p first, if any]
//     p <- p - (lr / (1 - beta1^t)) m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
// in that operation order and in fp32 like ATen's kernels; HBM-bound (16 bytes read, 12 written per element).
#include "common.hip.h"

namespace {

__global__ __launch_bounds__(256) void adam_kernel(const mi_adam_params a) {
    const int c = blockIdx.x;
    const mi_adam_tensor t = a.tensors[a.chunk_tensor[c]];
    const long long i0 = (long long)a.chunk_off[c] * a.chunk;
    const long long i1 = i0 + a.chunk < t.n ? i0 + a.chunk : t.n;
    const float gs = a.grad_scale ? *a.grad_scale : 1.0f;
    const float step_size = a.lr / a.bias_correction1, bc2_sqrt = sqrtf(a.bias_correction2);
    for (long long i = i0 + threadIdx.x; i < i1; i += 256) {
        float g = t.g[i] * gs, p = t.p[i], m = t.m[i], v = t.v[i];
        if (a.weight_decay != 0.0f) g = fmaf(a.weight_decay, p, g);
        m = fmaf(g - m, a.one_minus_beta1, m);                               // lerp_
        v = fmaf(g * a.one_minus_beta2, g, v * a.beta2);                      // mul_(beta2).addcmul_(g, g, 1 - beta2)
        const float denom = sqrtf(v) / bc2_sqrt + a.eps;
        p = p - step_size * (m / denom);                                      // addcdiv_(m, denom, -step_size)
        t.p[i] = p; t.m[i] = m; t.v[i] = v;
    }
}

}  // namespace

extern "C" int mi_adam_step(const mi_adam_params* a, void* stream) {
    if (!a || a->nchunks <= 0 || a->chunk <= 0 || !a->tensors || !a->chunk_tensor || !a->chunk_off) { mi_set_error("mi_adam_step: empty / missing tables"); return MI_ERR_INVALID; }
    if (!(a->bias_correction1 > 0.0f) || !(a->bias_correction2 > 0.0f)) { mi_set_error("mi_adam_step: bias corrections must be positive (step >= 1)"); return MI_ERR_INVALID; }
    hipLaunchKernelGGL(adam_kernel, dim3(a->nchunks), dim3(256), 0, (hipStream_t)stream, *a);
    return mi_check_launch("adam_kernel");
}
