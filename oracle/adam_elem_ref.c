/* oracle/adam_elem_ref.c — TEST INFRASTRUCTURE ONLY.
 * Plain-C restatement of one element of torch's CPU Adam step (torch/optim/adam.py _single_tensor_adam as run.py:720-721
 * configures it) with the L2 gradient of model/layer.py:96-112 folded in.  It mirrors, operation for operation and
 * rounding for rounding, the device routine adam_elem() in csrc/common.h; tests/test_host_logic.py pins it bit-for-bit
 * against torch.optim.Adam on the CPU, which in turn pins the rounding pattern the kernels use.
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC (see __graft_entry__.build). */
#include <math.h>
#include <stdint.h>

void adam_elem_ref(float* w, float* m, float* v, const float* g_in, int64_t n, float lerp_w, float beta2,
                   float one_minus_beta2, float eps, float weight_decay, float l2_twice, float step_size, float bc2_sqrt) {
    for (int64_t i = 0; i < n; ++i) {
        float g = g_in[i] + l2_twice * w[i];                 /* autograd: grad of sum(l2*w^2) added to the batch gradient */
        g = fmaf(w[i], weight_decay, g);                     /* grad.add(param, alpha=wd): ATen vec fmadd */
        m[i] = fmaf(lerp_w, g - m[i], m[i]);                 /* exp_avg.lerp_(grad, 1-beta1) */
        float vv = v[i] * beta2;                             /* exp_avg_sq.mul_(beta2) */
        vv = fmaf(one_minus_beta2 * g, g, vv);               /* .addcmul_(grad, grad, value=1-beta2): ATen vec fmadd((value*t1), t2, self)
                                                                — the variant that reproduces torch 2.10 CPU bits (tests/test_host_logic.py) */
        v[i] = vv;
        float denom = sqrtf(vv) / bc2_sqrt + eps;            /* (sqrt / bias_correction2_sqrt).add_(eps) */
        w[i] = w[i] + (-step_size * m[i]) / denom;           /* param.addcdiv_(exp_avg, denom, value=-step_size) */
    }
}
