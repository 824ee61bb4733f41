// Training glue of the reference's loop as two multi-tensor kernels:
//   torch.nn.utils.clip_grad_norm_(params, max_norm)  +  torch.optim.Adam(lr, betas, eps).step()
// (train_detector_3D_angle.py:337, 385-387).  torch runs these as ~10 foreach kernels plus a host read of the norm;
// here the global gradient norm is reduced on device, the clip coefficient is derived from it inside the update
// kernel (no host sync) and every parameter / gradient / moment element is touched once.
//
// Tensors are described by a device table (one entry per tensor) and a chunk table (one entry per 4096-element
// chunk -> tensor, offset), so one launch covers all 195 parameter tensors of a ResNet-50 detector.
// Roofline: HBM -- read g, p, m, v and write p, m, v (+ g when the clipped gradient is written back): 28-32 B per
// parameter, 36.6 M parameters => ~1.1 GB per step.
#include "common.h"

#define OPT_CHUNK 4096

struct OptTensor {          // device table entry
    float *p, *g, *m, *v;
    int64_t n;
};
struct OptChunk {
    int tensor;
    int offset_chunks;      // chunk index inside the tensor
};
static_assert(sizeof(OptTensor) == 40, "layout");

__global__ __launch_bounds__(256) void opt_sqnorm_kernel(const OptTensor *__restrict__ T, const OptChunk *__restrict__ C,
                                                         double *__restrict__ partial) {
    __shared__ double red[4];
    const OptChunk c = C[blockIdx.x];
    const OptTensor t = T[c.tensor];
    const int64_t base = (int64_t)c.offset_chunks * OPT_CHUNK;
    const int64_t end = base + OPT_CHUNK < t.n ? base + OPT_CHUNK : t.n;
    float s = 0.f;
    for (int64_t i = base + threadIdx.x; i < end; i += 256) { const float g = t.g[i]; s += g * g; }
    double d = wave_sum((double)s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = d;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// hp (device, optional): [lr, max_norm, beta1, beta2, eps, grad_scale] -- hyper-parameters a replayed hipGraph must be able to see
// change (a scheduler's lr) and the 1/world of a data-parallel gradient SUM, folded in here instead of a pass over the buffer.
#define HP_LR 0
#define HP_MAX_NORM 1
#define HP_BETA1 2
#define HP_BETA2 3
#define HP_EPS 4
#define HP_GSCALE 5

__global__ __launch_bounds__(1024) void opt_norm_final_kernel(const double *__restrict__ partial, int n, float *__restrict__ out,
                                                              const float *__restrict__ hp) {
    __shared__ double red[16];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 1024) s += partial[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int k = 0; k < 16; ++k) t += red[k];
        const double gs = hp != nullptr ? (double)hp[HP_GSCALE] : 1.0;
        out[0] = (float)(sqrt(t) * gs);                           // total L2 norm of the (scaled) gradients: clip_grad_norm_'s return value
    }
}

// Device-resident step counter (hipGraph capture: a replayed launch must not carry the step number as a baked-in kernel
// argument): one thread increments it, the update kernel derives the bias corrections from it in double, as the host does.
__global__ void opt_step_inc_kernel(int *__restrict__ step) { step[0] += 1; }

__global__ __launch_bounds__(256) void opt_adam_kernel(const OptTensor *__restrict__ T, const OptChunk *__restrict__ C,
                                                       const float *__restrict__ total_norm, float max_norm, float lr,
                                                       float beta1, float beta2, float eps, float bc1, float bc2_sqrt,
                                                       int write_clipped, const int *__restrict__ step_dev,
                                                       const float *__restrict__ hp) {
    float gscale = 1.0f;
    if (hp != nullptr) {                                          // wave-uniform scalar loads
        lr = hp[HP_LR]; max_norm = hp[HP_MAX_NORM]; beta1 = hp[HP_BETA1]; beta2 = hp[HP_BETA2]; eps = hp[HP_EPS]; gscale = hp[HP_GSCALE];
    }
    if (step_dev != nullptr) {                                    // wave-uniform: same value for every thread
        const int st = step_dev[0];
        bc1 = (float)(1.0 - pow((double)beta1, (double)st));
        bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)st));
    }
    const OptChunk c = C[blockIdx.x];
    const OptTensor t = T[c.tensor];
    const int64_t base = (int64_t)c.offset_chunks * OPT_CHUNK;
    const int64_t end = base + OPT_CHUNK < t.n ? base + OPT_CHUNK : t.n;
    float coef = 1.0f;
    if (max_norm > 0.f) {                                         // clip_coef = max_norm / (norm + 1e-6), clamped to 1
        coef = max_norm / (total_norm[0] + 1e-6f);
        coef = coef > 1.0f ? 1.0f : coef;
    }
    const float step = lr / bc1;
    coef *= gscale;                                               // (gscale == 1: bit-identical to the unscaled form)
    for (int64_t i = base + threadIdx.x; i < end; i += 256) {
        const float g = t.g[i] * coef;
        const float m = beta1 * t.m[i] + (1.0f - beta1) * g;      // exp_avg.lerp_(grad, 1 - beta1)
        const float v = beta2 * t.v[i] + (1.0f - beta2) * g * g;  // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
        const float denom = sqrtf(v) / bc2_sqrt + eps;
        t.p[i] -= step * (m / denom);
        t.m[i] = m;
        t.v[i] = v;
        if (write_clipped) t.g[i] = g;
    }
}

extern "C" int64_t rn_opt_workspace_bytes(int n_chunks) { return (int64_t)n_chunks * sizeof(double); }

static int opt_clip_adam_impl(const void *tensor_table, const void *chunk_table, int n_chunks, float max_norm, float lr,
                              float beta1, float beta2, float eps, int step, int *step_dev, int write_clipped, void *workspace,
                              float *total_norm, void *stream, const float *hp = nullptr);

extern "C" int rn_opt_clip_adam(const void *tensor_table, const void *chunk_table, int n_chunks, float max_norm, float lr,
                                float beta1, float beta2, float eps, int step, int write_clipped, void *workspace,
                                float *total_norm, void *stream) {
    if (step <= 0) return RN_EINVAL;
    return opt_clip_adam_impl(tensor_table, chunk_table, n_chunks, max_norm, lr, beta1, beta2, eps, step, nullptr, write_clipped,
                              workspace, total_norm, stream);
}

// Same step with the step number kept on the device: *step_dev is incremented first (start it at 0), then used.  No kernel
// argument changes from step to step, so the launch sequence can be captured once and replayed (hipGraph).
extern "C" int rn_opt_clip_adam_dev(const void *tensor_table, const void *chunk_table, int n_chunks, float max_norm, float lr,
                                    float beta1, float beta2, float eps, int *step_dev, int write_clipped, void *workspace,
                                    float *total_norm, void *stream) {
    if (step_dev == nullptr) return RN_EINVAL;
    return opt_clip_adam_impl(tensor_table, chunk_table, n_chunks, max_norm, lr, beta1, beta2, eps, 1, step_dev, write_clipped,
                              workspace, total_norm, stream);
}

// The hyper-parameters in DEVICE memory (hp: six floats, [lr, max_norm, beta1, beta2, eps, grad_scale]) as well as the step
// number: nothing in the launch sequence is baked into a captured graph, so a ReduceLROnPlateau step between two replays
// reaches the kernels (the caller rewrites hp with an asynchronous copy outside the graph).  grad_scale multiplies every
// gradient before the norm and the update: 1/world for a data-parallel gradient SUM (ddp.GradReducer(defer_scale=True)).
// The norm / clip launches always run (max_norm <= 0 in hp disables the clip inside the kernel).
extern "C" int rn_opt_clip_adam_hp(const void *tensor_table, const void *chunk_table, int n_chunks, const float *hp, int *step_dev,
                                   int write_clipped, void *workspace, float *total_norm, void *stream) {
    if (step_dev == nullptr || hp == nullptr) return RN_EINVAL;
    return opt_clip_adam_impl(tensor_table, chunk_table, n_chunks, 1.0f, 0.f, 0.f, 0.f, 0.f, 1, step_dev, write_clipped, workspace,
                              total_norm, stream, hp);
}

static int opt_clip_adam_impl(const void *tensor_table, const void *chunk_table, int n_chunks, float max_norm, float lr,
                              float beta1, float beta2, float eps, int step, int *step_dev, int write_clipped, void *workspace,
                              float *total_norm, void *stream, const float *hp) {
    if (n_chunks <= 0) return RN_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const OptTensor *T = reinterpret_cast<const OptTensor *>(tensor_table);
    const OptChunk *C = reinterpret_cast<const OptChunk *>(chunk_table);
    double *partial = reinterpret_cast<double *>(workspace);
    if (max_norm > 0.f) {
        hipLaunchKernelGGL(opt_sqnorm_kernel, dim3(n_chunks), dim3(256), 0, s, T, C, partial);
        hipLaunchKernelGGL(opt_norm_final_kernel, dim3(1), dim3(1024), 0, s, (const double *)partial, n_chunks, total_norm, hp);
    }
    if (step_dev != nullptr) hipLaunchKernelGGL(opt_step_inc_kernel, dim3(1), dim3(1), 0, s, step_dev);
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    hipLaunchKernelGGL(opt_adam_kernel, dim3(n_chunks), dim3(256), 0, s, T, C, (const float *)total_norm, max_norm, lr, beta1,
                       beta2, eps, (float)bc1, (float)sqrt(bc2), write_clipped, (const int *)step_dev, hp);
    RN_LAUNCH_CHECK();
    return RN_OK;
}
